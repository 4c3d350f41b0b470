#!/bin/bash
# Round 4: a short GPU visit -- named tests + the driver-style line.   tools/r4_quick.sh <tag> "<pytest -k expression>"
set -u
TAG=${1:-r4q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "$2" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "driver-style rc=$?"; tail -5 $OUT/bench_driver.err
python - $OUT/bench_driver.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], {k:(v["launches"], round(v["avg_ms"]*1e3,1)) for k,v in d["kernels_ms"].items()})
for k,v in d.items():
    if isinstance(v,dict) and k.startswith(("config3","config4","config5","experimental","cpu_baseline")):
        print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","ms_per_step","dtype","cores","extrapolated")}, (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"))
print("extras_error", d.get("extras_error"))
PY
