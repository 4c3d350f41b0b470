#!/bin/bash
# Round-3 first visit: changed tests, the three bench launch forms, exchange cost captured vs segmented.
set -u
OUT=gpurun_out/r3a; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_data_parallel_gpu.py tests/test_aux_aug_acting_gpu.py tests/test_integration_stub_gpu.py tests/test_encoder_fwd_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver-style rc=$?"; tail -c 2500 $OUT/bench_driver.json
python bench.py --gpus 2 --backend gloo --share-gpu --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_2r.json 2> $OUT/bench_2r.err; echo "bench 2 ranks rc=$?"; tail -c 1500 $OUT/bench_2r.json; tail -5 $OUT/bench_2r.err
for cap in 1 0; do
for b in 256 32; do
  PCRL_CAPTURE_EXCHANGE=$cap python bench.py --single-rank-exchange --backend nccl --batch $b --steps 1000 --warmup 200 --no-cpu-baseline > $OUT/sre_cap${cap}_b$b.json 2> $OUT/sre_cap${cap}_b$b.err; echo "sre cap=$cap b=$b rc=$?"
  python - $OUT/sre_cap${cap}_b$b.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","ms_per_step_nocomm","comm_ms_per_step")}, d["config"].get("exchange"))
PY
done; done
