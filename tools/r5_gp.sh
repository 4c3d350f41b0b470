#!/bin/bash
# Round 5: GEMM shape probe under the kernel trace -> gpurun_out/gp/table.txt
export TMPDIR=/tmp
OUT=gpurun_out/gp; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/r5_gemm_probe.py run $OUT/labels.txt > $OUT/run.log 2>&1
echo "run rc=$?"; tail -3 $OUT/run.log
python3 tools/r5_gemm_probe.py fold $OUT/trace $OUT/labels.txt > $OUT/table.txt 2>&1
find $OUT/trace -name "*.csv" -size +20M -delete
cat $OUT/table.txt
