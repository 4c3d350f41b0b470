#!/bin/bash
# Kernel sequence (durations + gaps) of two consecutive steady-state K1 steps, from a rocprofv3 kernel trace of the graph-replayed step.
export TMPDIR=/tmp
OUT=${1:-gpurun_out/timeline}; shift
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -- python3 bench.py --steps 60 --warmup 30 --no-cpu-baseline --no-extra-workloads --no-experimental "$@" > $OUT/log.txt 2>&1
python3 tools/step_timeline.py $OUT bwdg_reduce > $OUT/timeline.txt
find $OUT -name "*.db" -delete
cat $OUT/timeline.txt
