#!/bin/bash
# Round 5 A/B on ONE box: round-4 library (_ab/r4) against the tree's.   tools/r5_ab1.sh <tag>
set -u
TAG=${1:-r5ab1}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for lib in r4 new; do
  if [ $lib = r4 ]; then export PCRL_HIP_LIB=$PWD/_ab/r4/libpcrl_hip.so; else unset PCRL_HIP_LIB; fi
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$lib -- python3 tools/r5_gemm_probe.py run $OUT/labels_$lib.txt > $OUT/run_$lib.log 2>&1
  python3 tools/r5_gemm_probe.py fold $OUT/trace_$lib $OUT/labels_$lib.txt > $OUT/table_$lib.txt 2>&1; rm -rf $OUT/trace_$lib
done
paste <(cut -c1-28,36-45 $OUT/table_r4.txt) <(cut -c36-45 $OUT/table_new.txt) | grep -v "K32 \|K256 " > $OUT/table_ab.txt; cat $OUT/table_ab.txt
for rep in 1 2; do
for spec in "k1 256" "k1 32" "k3 128"; do
  for lib in r4 new; do
    if [ $lib = r4 ]; then export PCRL_HIP_LIB=$PWD/_ab/r4/libpcrl_hip.so; else unset PCRL_HIP_LIB; fi
    set -- $spec
    python bench.py --workload $1 --batch $2 --steps 300 --warmup 30 --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/b.json 2> $OUT/b.err || tail -3 $OUT/b.err
    python - $OUT/b.json "$1 b$2 $lib" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"],4), "ms", round(d["value"],1))
PY
  done
done
done
