#!/bin/bash
# Same-box A/B of this round's launch cuts (environment switches of methods/fused.py), K1 and a rank's 32-cloud share of it.
set -u
export TMPDIR=/tmp
run() { python bench.py "$@" --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for cfg in "PCRL_TAIL_BWD=0 PCRL_ATTACH_COLSUM=0" "PCRL_TAIL_BWD=1 PCRL_ATTACH_COLSUM=0" "PCRL_TAIL_BWD=1 PCRL_ATTACH_COLSUM=1"; do
  echo "== $cfg (rep $rep)"
  echo -n " k1      "; env $cfg python bench.py --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))"
  echo -n " k1 b32  "; env $cfg python bench.py --batch 32 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))"
done
done
