#!/bin/bash
# A/B of the round-2 (bf16) backward with whole-piece stores: parity tests on the new library, then K2 lines and stand-alone timings.
export TMPDIR=/tmp
NEW=_ab/libpcrl_hip_bf16w.so
PCRL_HIP_LIB=$NEW python -m pytest tests/test_encoder_bwd_gpu.py tests/test_k2_fullsize_bf16_gpu.py tests/test_update_step_gpu.py -q 2>&1 | tail -4
for l in "" $NEW; do
  echo "== lib=${l:-in-tree}"
  PCRL_HIP_LIB=$l python bench.py --workload k2 --no-cpu-baseline --no-experimental --no-extra-workloads --steps 1000 --warmup 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k2', d['value'], d['ms_per_step'], {k:(v['launches'], round(v['avg_ms']*1e3,1)) for k,v in d['kernels_ms'].items()})"
  for cfg in "--B 512 --N 1200 --c1 128 --seg 1 --bf16" "--B 256 --N 1024 --bf16" "--B 64 --N 1200 --c1 128 --seg 1 --bf16"; do echo -n "$cfg: "; PCRL_HIP_LIB=$l python tools/bench_encoder.py $cfg --iters 30 2>&1 | grep encoder_bwd; done
done
