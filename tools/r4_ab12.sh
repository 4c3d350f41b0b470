#!/bin/bash
# Same-box A/B of builds / switches: base (_ab/libpcrl_hip_base.so), the step's last launch with all partials in flight only
# (_ab/libpcrl_hip_tailonly.so), and the shipped library with the merge + feature-head kernel at 4 waves (PCRL_MERGE_NT=256: weights requested
# before the keys, keys eight segments at a time) or 16 waves (one chunk of eight features per wave).  Initial state, alternating.
set -u
export TMPDIR=/tmp
one() { env $1 python bench.py $2 --warmup 30 --steps 300 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']), {k:(v['launches'], round(v['avg_ms']*1e3,1)) for k,v in d['kernels_ms'].items() if 'fwd' in k})"; }
for rep in 1 2 3; do
  for cfg in "PCRL_HIP_LIB=_ab/libpcrl_hip_base.so" "PCRL_HIP_LIB=_ab/libpcrl_hip_tailonly.so" "PCRL_MERGE_NT=256" "PCRL_MERGE_NT=1024"; do
    echo "== $cfg (rep $rep)"
    echo -n " k1      "; one "$cfg" ""
    echo -n " k1 b32  "; one "$cfg" "--batch 32"
    echo -n " k1 b64  "; one "$cfg" "--batch 64"
    echo -n " k3 b128 "; one "$cfg" "--workload k3 --batch 128"
  done
done
