#!/bin/bash
# Same-box A/B: the row-split policy tails with all operand pieces of a phase in flight (PCRL_TAIL_PREFETCH=1, default) against the
# group-by-group loops (=0).  Initial training state (30 + 300 steps), alternating, three repetitions.
set -u
export TMPDIR=/tmp
python -m pytest tests/test_headtail_gpu.py tests/test_update_step_gpu.py -m gpu -q -x 2>&1 | tail -3
one() { env $1 python bench.py $2 --warmup 30 --steps 300 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']), {k:(v['launches'], round(v['avg_ms']*1e3,1)) for k,v in d['kernels_ms'].items() if 'tail' in k})"; }
for rep in 1 2 3; do
for cfg in "PCRL_TAIL_PREFETCH=0" "PCRL_TAIL_PREFETCH=1"; do
  echo "== $cfg (rep $rep)"
  echo -n " k1      "; one "$cfg" ""
  echo -n " k1 b32  "; one "$cfg" "--batch 32"
  echo -n " k3 b128 "; one "$cfg" "--workload k3 --batch 128"
  echo -n " k2      "; one "$cfg" "--workload k2"
done
done
