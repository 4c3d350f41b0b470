#!/bin/bash
# Same-box A/B of two builds of the encoder backward: _ab/libpcrl_hip_base.so (the previous build) against the shipped library.  Used for
# (i) the wgrad kernel with the 10 upper blocks of the symmetric G and the longest-task-first schedule, (ii) the points kernel whose tile loop
# requests the next tile's point index / owned-channel word / point ahead of time.
set -u
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "bwd or update or fullsize or integration" 2>&1 | tail -3
for cfg in "--B 256 --N 1024" "--B 32 --N 1024" "--B 128 --N 1200 --c1 128 --seg 1" "--B 512 --N 1200 --c1 128 --seg 1" "--B 1024 --N 1200 --c1 128 --seg 1"; do
  for lib in "_ab/libpcrl_hip_base.so" ""; do
    echo -n "${lib:-shipped} $cfg: "; PCRL_HIP_LIB=$lib python tools/bench_encoder.py $cfg --iters 30 2>&1 | grep encoder_bwd
  done
done
one() { env PCRL_HIP_LIB=$1 python bench.py $2 --warmup 30 --steps 300 --no-cpu-baseline --no-experimental --no-extra-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']), {k:(v['launches'], round(v['avg_ms']*1e3,1)) for k,v in d['kernels_ms'].items() if 'bwd' in k})"; }
for rep in 1 2 3; do
  for lib in "_ab/libpcrl_hip_base.so" ""; do
    echo "== ${lib:-shipped} (rep $rep)"
    echo -n " k1      "; one "$lib" ""
    echo -n " k3 b128 "; one "$lib" "--workload k3 --batch 128"
    echo -n " k2      "; one "$lib" "--workload k2"
  done
done
