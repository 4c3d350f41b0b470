#!/bin/bash
# Round 3: Gram-form encoder backward (PCRL_BWD_ALGO=1, default) against the round-2 kernels (PCRL_BWD_ALGO=0): parity tests, timings.
set -u
OUT=gpurun_out/r3c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_bwd_gpu.py -m gpu -q -k "not bf16 and not split" > $OUT/pytest_bwd_gram.log 2>&1; echo "gram pytest rc=$?"; tail -25 $OUT/pytest_bwd_gram.log
for cfg in "--B 256 --N 1024" "--B 128 --N 1024" "--B 32 --N 1024" "--B 128 --N 1200 --c1 128 --seg 1" "--B 1024 --N 1200 --c1 128 --seg 1" "--B 512 --N 8192"; do
  for nw in 4 8; do
    echo "gram nw=$nw $cfg"; PCRL_BWD_TILE_WAVES=$nw timeout 300 python tools/bench_encoder.py $cfg --iters 30 2>&1 | grep encoder_bwd
  done
done
PCRL_BWD_TILE_WAVES=4 bash tools/prof_encoder.sh --B 256 --N 1024 --iters 20
PCRL_BWD_TILE_WAVES=8 bash tools/prof_encoder.sh --B 256 --N 1024 --iters 20
PCRL_BWD_TILE_WAVES=4 bash tools/prof_encoder.sh --B 1024 --N 1200 --c1 128 --seg 1 --iters 10
PCRL_BWD_TILE_WAVES=4 bash tools/prof_encoder.sh --B 128 --N 1200 --c1 128 --seg 1 --iters 10
