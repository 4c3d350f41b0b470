#!/bin/bash
set -u
export TMPDIR=/tmp
true
one() { env $1 python bench.py $2 --warmup 30 --steps 300 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']), {k:(v['launches'], round(v['avg_ms']*1e3,1)) for k,v in d['kernels_ms'].items() if 'tail' in k})"; }
for rep in 1 2 3; do
for cfg in "PCRL_TAIL_SPLIT_MAX=0 PCRL_TAIL_BWD=0" "PCRL_TAIL_SPLIT_MAX=0 PCRL_TAIL_BWD=1" "PCRL_TAIL_SPLIT_MAX=512 PCRL_TAIL_BWD=0" "PCRL_TAIL_SPLIT_MAX=512 PCRL_TAIL_BWD=1"; do
  echo "== $cfg (rep $rep)"
  echo -n " k1      "; one "$cfg" ""
  echo -n " k1 b32  "; one "$cfg" "--batch 32"
done
done
