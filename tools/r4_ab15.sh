#!/bin/bash
# Same-box A/B on K2 (DrQ, A = 22): the forward policy tail's row-output limit with the row-split kernels that request all their operands
# at once -- 8 192 (default until now: both of K2's tails run as GEMM + head launch), 11 264 (the actor phase's 256 rows fused), 22 528
# (the critic phase's 512 rows too).  Initial state, alternating, three repetitions.
set -u
export TMPDIR=/tmp
one() { env $1 python bench.py $2 --warmup 30 --steps 300 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']), {k:(v['launches'], round(v['avg_ms']*1e3,1)) for k,v in d['kernels_ms'].items() if 'tail' in k or 'tanh' in k})"; }
for rep in 1 2 3; do
for cfg in "PCRL_TAIL_FWD_MAX_SPLIT=8192" "PCRL_TAIL_FWD_MAX_SPLIT=11264" "PCRL_TAIL_FWD_MAX_SPLIT=22528"; do
  echo "== $cfg (rep $rep)"
  echo -n " k2      "; one "$cfg" "--workload k2"
done
done
