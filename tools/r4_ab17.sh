#!/bin/bash
# Same-box A/B of the heads' weight-gradient GEMMs: 32x32 split-K tiles (PCRL_GEMM_WAVE_TILES_MIN=1000000000), a tile per wave
# (PCRL_GEMM_WAVE_PAIRS=0: cfg 2), two row blocks per wave from 8-byte loads (default: cfg 3).  Initial state, alternating, three repetitions.
set -u
export TMPDIR=/tmp
one() { env $1 python bench.py $2 --warmup 30 --steps 300 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']), {k:(v['launches'], round(v['avg_ms']*1e3,1)) for k,v in d['kernels_ms'].items() if 'gemm' in k})"; }
for rep in 1 2 3; do
for cfg in "PCRL_GEMM_WAVE_TILES_MIN=1000000000" "PCRL_GEMM_WAVE_PAIRS=0" "PCRL_GEMM_WAVE_PAIRS=1"; do
  echo "== $cfg (rep $rep)"
  echo -n " k1      "; one "$cfg" ""
  echo -n " k1 b32  "; one "$cfg" "--batch 32"
  echo -n " k2      "; one "$cfg" "--workload k2"
  echo -n " k3 b128 "; one "$cfg" "--workload k3 --batch 128"
  echo -n " k3      "; one "$cfg" "--workload k3 --steps 150"
done
done
