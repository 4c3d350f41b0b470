#!/bin/bash
# Final checks of round 4: full GPU suite, the default bench line as the driver runs it, DrQ / K4 side workloads in a two-rank run.
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r4f; mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_all.log
( time python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | grep real
python - $OUT/bench_driver.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic"], d.get("extras_error"))
for k,v in d.items():
    if isinstance(v,dict) and k.startswith(("config3","config4","config5","experimental","cpu_baseline")):
        print(k, round(v.get("value",0),3), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), (v.get("cpu_baseline") or {}).get("value"))
PY
for wl in k2 k4 k1; do
  echo "== two ranks sharing the GPU over gloo: $wl"
  timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --workload $wl --steps 6 --warmup 5 --no-cpu-baseline --no-extra-workloads --replay-capacity 512 2> $OUT/dp_$wl.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['exchange'], d['config']['batch_per_gpu'])" || tail -5 $OUT/dp_$wl.err
done
echo "== two ranks, default k1 line with its extras (gloo, shared GPU)"
( time timeout 900 python bench.py --gpus 2 --backend gloo --share-gpu --steps 6 --warmup 5 --replay-capacity 512 --extra-steps 6 2> $OUT/dp_extras.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], [k for k in d if k.startswith('config')], d.get('extras_error'))" ) 2>&1 | tail -5
