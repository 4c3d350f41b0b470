#!/bin/bash
# Where a rank's small share of the K1 batch goes (8 / 4 GPUs: 32 / 64 clouds): kernel sequence of the replayed step with durations
# and gaps, the head GEMMs at M = 32 / 64 stand-alone, and the K3 bench line with per-span min / median / max.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3s; mkdir -p $OUT
bash tools/r3_timeline.sh $OUT/tl_b32 --batch 32 > $OUT/timeline_b32.txt 2>&1
bash tools/r3_timeline.sh $OUT/tl_b64 --batch 64 > $OUT/timeline_b64.txt 2>&1
bash tools/r3_timeline.sh $OUT/tl_b256 > $OUT/timeline_b256.txt 2>&1
rm -rf $OUT/tl_b32 $OUT/tl_b64 $OUT/tl_b256
for m in 32 64; do GEMM_M=$m python tools/bench_gemm.py > $OUT/gemm_m$m.txt 2>&1; done
python bench.py --workload k3 --steps 40 --warmup 10 --no-cpu-baseline --no-extra-workloads > $OUT/bench_k3_spans.json 2> $OUT/bench_k3_spans.err
tail -50 $OUT/timeline_b32.txt
