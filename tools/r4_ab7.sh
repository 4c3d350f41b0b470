#!/bin/bash
set -u
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "replay or update_parameters or graph or sampling" 2>&1 | tail -3
for wl in "k1:" "k3:--workload k3 --steps 100" "k4:--workload k4 --steps 60"; do
  name=${wl%%:*}; args=${wl#*:}
  python bench.py $args --warmup 30 ${args:+} --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name %.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']), {k:(v['launches'], round(v['avg_ms']*1e3,1)) for k,v in d['kernels_ms'].items() if 'replay' in k})"
done
