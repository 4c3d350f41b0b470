#!/bin/bash
# Round 6: the idle device between two replays as a distribution over the trace (tools/step_timeline.py), K1 and its 32-cloud share, profiler on;
# and the same share unprofiled right after it, for the step time the kernel sums are compared with.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6l2; mkdir -p $OUT
bash tools/r3_timeline.sh $OUT/tl_k1 > $OUT/timeline_k1.txt 2>&1
bash tools/r3_timeline.sh $OUT/tl_b32 --batch 32 > $OUT/timeline_k1_b32.txt 2>&1
bash tools/r3_timeline.sh $OUT/tl_k3b128 --workload k3 --batch 128 > $OUT/timeline_k3_b128.txt 2>&1
for f in $OUT/timeline_*.txt; do echo $f; tail -n 2 $f; done
python3 bench.py --batch 32 --no-cpu-baseline --no-extra-workloads --no-experimental > $OUT/share_k1_b32.json 2> $OUT/share_k1_b32.err
python3 -c "import json;d=json.loads([l for l in open('$OUT/share_k1_b32.json') if l.startswith('{')][-1]);print('b32 unprofiled ms/step', d['ms_per_step'])"
