#!/bin/bash
# (Record of a round-4 experiment: PCRL_BWD_WGRAD_PERSIST selected a persistent-accumulator wgrad kernel that measured slower and was
# not kept -- DESIGN.md section 8; with the shipped library both settings run the same kernel.)
set -u
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "bwd or update_parameters or k2_full" 2>&1 | tail -3
one() { env $1 python bench.py $2 --warmup 30 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']), {k:(v['launches'], round(v['avg_ms']*1e3,1)) for k,v in d['kernels_ms'].items() if 'bwd' in k})"; }
for rep in 1 2; do
for cfg in "PCRL_BWD_WGRAD_PERSIST=0" "PCRL_BWD_WGRAD_PERSIST=1"; do
  echo "== $cfg (rep $rep)"
  echo -n " k2 "; one "$cfg" "--workload k2 --steps 300"
  echo -n " k3 "; one "$cfg" "--workload k3 --steps 100"
  echo -n " k4 "; one "$cfg" "--workload k4 --steps 60"
done
done
for cfg in "--B 1024 --N 1200 --c1 128 --seg 1" "--B 512 --N 1200 --c1 128 --seg 1" "--B 512 --N 8192"; do
  for pz in 0 1; do echo -n "persist=$pz $cfg: "; PCRL_BWD_WGRAD_PERSIST=$pz python tools/bench_encoder.py $cfg --iters 20 2>&1 | grep encoder_bwd; done
done
