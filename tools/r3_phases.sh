#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_bwd_gpu.py tests/test_k2_fullsize_bf16_gpu.py -m gpu -q 2>&1 | tail -8
for skip in 0 16; do
  echo "== PCRL_BWDG_SKIP=$skip"; PCRL_BWDG_SKIP=$skip bash tools/prof_encoder.sh --B 256 --N 1024 --iters 20 2>&1 | grep "bwdg_"
done
for cfg in "--B 256 --N 1024" "--B 128 --N 1200 --c1 128 --seg 1" "--B 32 --N 1024" "--B 1024 --N 1200 --c1 128 --seg 1" "--B 512 --N 8192" "--B 256 --N 1024 --split"; do
  echo "$cfg"; python tools/bench_encoder.py $cfg --iters 30 2>&1 | grep encoder_bwd
done
bash tools/prof_encoder.sh --B 1024 --N 1200 --c1 128 --seg 1 --iters 10 | grep bwdg_
