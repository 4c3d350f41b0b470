#!/bin/bash
# Item 5 of VERDICT r3 (bf16 forward >= 0.25 of peak, or a measurement that says why not): the production bf16 forward next to a
# build with LayerNorm-2 + max-pool removed (-DPCRL_FWD_ABLATE_TAIL), same launch geometry.   Build part runs anywhere (hipcc
# cross-compiles); the timing part needs the GPU:  tools/r4_fwd_ablate.sh build | tools/r4_fwd_ablate.sh run
set -u
cd "$(dirname "$0")/.."
if [ "${1:-run}" = "build" ]; then
  mkdir -p _ab/build_ablate
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt \
      -Iinclude -Ipointcloud_rl_amd/csrc -Wall -Wno-unused-function -DPCRL_FWD_ABLATE_TAIL -c pointcloud_rl_amd/csrc/encoder_fwd.hip -o _ab/build_ablate/encoder_fwd.o || exit 1
  objs=$(ls pointcloud_rl_amd/csrc/build/*.o | grep -v encoder_fwd.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libpcrl_hip_ablate_tail.so $objs _ab/build_ablate/encoder_fwd.o || exit 1
  ls -la _ab/libpcrl_hip_ablate_tail.so; exit 0
fi
export TMPDIR=/tmp
for cfg in "--B 512 --N 1200 --c1 128 --seg 1 --bf16" "--B 256 --N 1024 --bf16" "--B 256 --N 1024"; do
  for lib in "" "_ab/libpcrl_hip_ablate_tail.so"; do
    echo "== $cfg  lib=${lib:-production}"
    PCRL_HIP_LIB=$lib python tools/bench_encoder.py $cfg --iters 40 --fwd-only 2>&1 | grep encoder_fwd
  done
done
