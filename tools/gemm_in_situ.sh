#!/bin/bash
# GEMM launches of the graph-replayed K1 step grouped by grid size (rocprofv3 kernel trace, rocpd database): in-situ durations,
# i.e. with the operands where the step leaves them (weights just rewritten by Adam), not the L2-hot loop of tools/bench_gemm.py.
export TMPDIR=/tmp
OUT=${1:-gpurun_out/gemm_trace}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -- python3 bench.py --steps 60 --warmup 30 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 tools/_gemm_trace_sum.py $OUT
find $OUT -name "*.db" -delete
