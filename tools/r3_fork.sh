#!/bin/bash
# The forked prep branch: parity tests that touch it, then K1 / K3-share lines with and without it, and the step's timeline.
set -u
OUT=gpurun_out/${1:-r3g}; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_encoder_bwd_gpu.py tests/test_update_step_gpu.py tests/test_data_parallel_gpu.py -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for fork in 1 0; do
  PCRL_BWD_FORK=$fork python bench.py --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/k1_fork$fork.json 2> $OUT/k1_fork$fork.err
  PCRL_BWD_FORK=$fork python bench.py --no-cpu-baseline --no-experimental --no-extra-workloads --workload k3 --batch 128 --steps 1000 --warmup 200 > $OUT/k3s_fork$fork.json 2> $OUT/k3s_fork$fork.err
  python - $OUT/k1_fork$fork.json $OUT/k3s_fork$fork.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
done
bash tools/r3_timeline.sh $OUT/timeline > /dev/null; grep -n "prep\|points\|gemm\|fwd" $OUT/timeline/timeline.txt | tail -30
