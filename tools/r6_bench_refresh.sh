#!/bin/bash
# the four bench lines again with the final bench.py (tile counts, GEMM matrix-busy from the committed ledger), same protocol as tools/r6_measure.sh
set -u
OUT=gpurun_out/r6m2; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum pointcloud_rl_amd/libpcrl_hip.so | tee $OUT/libpcrl_hip.sha256
for wl in k1 k2 k3 k4; do
  steps=2000; warm=500
  [ "$wl" = "k3" ] && { steps=400; warm=100; }
  [ "$wl" = "k4" ] && { steps=200; warm=40; }
  extra="--no-extra-workloads"; [ "$wl" = "k1" ] && extra=""
  python bench.py --workload $wl --steps $steps --warmup $warm $extra > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "bench $wl rc=$?"
done
python bench.py --steps 20 --warmup 5 > $OUT/bench_k1_driver_style.json 2> /dev/null
for b in 128 64 32; do python bench.py --batch $b --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/share_k1_b$b.json 2>/dev/null; done
python bench.py --workload k3 --batch 128 --steps 1000 --warmup 200 --no-cpu-baseline --no-extra-workloads > $OUT/share_k3_b128.json 2>/dev/null
ls $OUT
