#!/bin/bash
# Same-box A/B: the critic phase's re-pack as extra workgroups of the replay's sampling launch (PCRL_ENTRY_PACK=1, default) against a
# launch of its own (=0).  Initial training state (30 + 300 steps), alternating, three repetitions.
set -u
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "aux or update or data_parallel or k2_fullsize" 2>&1 | tail -3
one() { env $1 python bench.py $2 --warmup 30 --steps 300 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2 3; do
for cfg in "PCRL_ENTRY_PACK=0" "PCRL_ENTRY_PACK=1"; do
  echo "== $cfg (rep $rep)"
  echo -n " k1      "; one "$cfg" ""
  echo -n " k1 b32  "; one "$cfg" "--batch 32"
  echo -n " k3 b128 "; one "$cfg" "--workload k3 --batch 128"
  echo -n " k2      "; one "$cfg" "--workload k2"
done
done
bash tools/r3_timeline.sh gpurun_out/tl_k1 > /dev/null 2>&1; head -12 gpurun_out/tl_k1/timeline.txt; tail -1 gpurun_out/tl_k1/timeline.txt
