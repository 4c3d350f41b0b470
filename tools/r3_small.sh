#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_bwd_gpu.py -m gpu -q 2>&1 | tail -3
for cfg in "--B 256 --N 1024" "--B 128 --N 1024" "--B 64 --N 1024" "--B 32 --N 1024" "--B 128 --N 1200 --c1 128 --seg 1" "--B 16 --N 1024"; do
  for algo in 1 0; do echo "algo=$algo $cfg"; PCRL_BWD_ALGO=$algo python tools/bench_encoder.py $cfg --iters 30 2>&1 | grep encoder_bwd; done
done
bash tools/prof_encoder.sh --B 32 --N 1024 --iters 20 | grep bwdg_
bash tools/prof_encoder.sh --B 128 --N 1200 --c1 128 --seg 1 --iters 20 | grep bwdg_
