#!/bin/bash
# Round 5: GEMM tests + the shape probe + the step rates the verdict names.   tools/r5_gemm_visit.sh <tag>
set -u
TAG=${1:-r5g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_dense_tail_gpu.py tests/test_headtail_gpu.py -q -x > $OUT/pytest_dense.log 2>&1; echo "pytest dense rc=$?"; tail -5 $OUT/pytest_dense.log
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/r5_gemm_probe.py run $OUT/labels.txt > $OUT/run.log 2>&1
python3 tools/r5_gemm_probe.py fold $OUT/trace $OUT/labels.txt > $OUT/table.txt 2>&1; rm -rf $OUT/trace
cat $OUT/table.txt
for spec in "k1 256" "k1 32" "k1 128" "k3 128"; do
  set -- $spec
  python bench.py --workload $1 --batch $2 --steps 300 --warmup 30 --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/bench_$1_b$2.json 2> $OUT/bench_$1_b$2.err || tail -3 $OUT/bench_$1_b$2.err
  python - $OUT/bench_$1_b$2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], {k:d[k] for k in ("value","ms_per_step")}, {k:(v["launches"], round(v["avg_ms"]*1e3,1)) for k,v in d.get("kernels_ms",{}).items()})
PY
done
