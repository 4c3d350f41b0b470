#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/v2; mkdir -p $OUT
python -m pytest tests/test_dense_tail_gpu.py tests/test_headtail_gpu.py -m gpu -x -q > $OUT/dense_tests.log 2>&1; echo "dense tests rc=$?"; tail -3 $OUT/dense_tests.log
python bench.py --steps 600 --warmup 200 --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/bench_k1.json 2> $OUT/bench_k1.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/v2/bench_k1.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
PY
PCRL_HIP_LIB=$PWD/_abship/r5/libpcrl_hip.so python bench.py --steps 600 --warmup 200 --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/bench_k1_r5.json 2> $OUT/bench_k1_r5.err; echo "bench r5 rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/v2/bench_k1_r5.json").read().strip().splitlines()[-1]); print("r5 lib:", d["value"], d["ms_per_step"])
PY
bash tools/r6_hunt2.sh 12
