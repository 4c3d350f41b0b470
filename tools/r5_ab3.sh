#!/bin/bash
# GEMM shape probe at the given M values for several libraries on one box.  tools/r5_ab3.sh <tag> "<libs>" "<M,M,...>"
set -u
TAG=${1:-r5ab3}; LIBS=${2:-"r4 new"}; export GEMM_MS=${3:-"512,1024"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for lib in $LIBS; do
  if [ $lib = new ]; then unset PCRL_HIP_LIB; else export PCRL_HIP_LIB=$PWD/_ab/$lib/libpcrl_hip.so; fi
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$lib -- python3 tools/r5_gemm_probe.py run $OUT/labels_$lib.txt > $OUT/run_$lib.log 2>&1
  python3 tools/r5_gemm_probe.py fold $OUT/trace_$lib $OUT/labels_$lib.txt > $OUT/table_$lib.txt 2>&1; rm -rf $OUT/trace_$lib
done
first=$(echo $LIBS | cut -d" " -f1)
cut -c1-28 $OUT/table_$first.txt > $OUT/cols.txt
for lib in $LIBS; do cut -c36-45 $OUT/table_$lib.txt | paste $OUT/cols.txt - > $OUT/cols2.txt; mv $OUT/cols2.txt $OUT/cols.txt; done
grep -v "K32 \|K256 \|K0 " $OUT/cols.txt
