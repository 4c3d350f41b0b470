#!/bin/bash
# Round 4: full GPU suite + the driver-style headline line.   tools/r4_suite.sh <tag>
set -u
TAG=${1:-r4a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_all.log; tail -30 $OUT/pytest_all.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "driver-style rc=$?"; tail -3 $OUT/bench_driver.err
python - $OUT/bench_driver.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], {k:(v["launches"], round(v["avg_ms"]*1e3,1)) for k,v in d["kernels_ms"].items()})
print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.items() if k.startswith(("config","experimental","extras","cpu_baseline"))})
PY
