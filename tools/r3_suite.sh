#!/bin/bash
# Full GPU suite + headline bench lines (driver-style and default protocol) + stand-alone backward timings.
set -u
TAG=${1:-r3d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_all.log; tail -12 $OUT/pytest_all.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "driver-style rc=$?"
python - $OUT/bench_driver.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], {k:(v["launches"], round(v["avg_ms"]*1e3,1)) for k,v in d["kernels_ms"].items()}, d.get("config4_k3",{}).get("value"), d.get("experimental_f32split",{}).get("value"))
PY
python bench.py --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/bench_k1.json 2> $OUT/bench_k1.err; echo "k1 rc=$?"; python -c "
import json,sys
d=json.loads(open('$OUT/bench_k1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
for cfg in "--B 256 --N 1024" "--B 128 --N 1200 --c1 128 --seg 1" "--B 32 --N 1024"; do python tools/bench_encoder.py $cfg --iters 30 2>&1 | grep encoder_bwd; done
bash tools/prof_encoder.sh --B 256 --N 1024 --iters 20
