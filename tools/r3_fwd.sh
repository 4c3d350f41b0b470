#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_encoder_fwd_gpu.py tests/test_k2_fullsize_bf16_gpu.py -m gpu -q 2>&1 | tail -3
for cfg in "--B 256 --N 1024" "--B 256 --N 1024 --bf16" "--B 512 --N 1200 --c1 128 --seg 1 --bf16" "--B 512 --N 1200 --c1 128 --seg 1" "--B 512 --N 8192" "--B 256 --N 1024 --split"; do
  echo "$cfg"; python tools/bench_encoder.py $cfg --iters 30 2>&1 | grep encoder_fwd
done
python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k1', d['value'], d['ms_per_step'], d['roofline']['frac'])"
python bench.py --workload k2 --steps 400 --warmup 100 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k2', d['value'], d['ms_per_step'], d['roofline']['frac'])"
