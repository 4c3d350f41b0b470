#!/bin/bash
# A/B of the encoder backward: the in-tree library vs _ab/libpcrl_hip_new.so (stand-alone launch timings), + the stamps build.
export TMPDIR=/tmp
for cfg in "--B 256 --N 1024" "--B 128 --N 1200 --c1 128 --seg 1" "--B 1024 --N 1200 --c1 128 --seg 1" "--B 32 --N 1024" "--B 512 --N 8192"; do
  for l in "" _ab/libpcrl_hip_new.so; do
    echo -n "$cfg lib=${l:-in-tree}: "; PCRL_HIP_LIB=$l python tools/bench_encoder.py $cfg --iters 30 2>&1 | grep encoder_bwd
  done
done
PCRL_HIP_LIB=_ab/libpcrl_hip_stamps.so python tools/bwdg_stamps.py --B 128 --N 1200 --c1 128 --seg 1 2>&1 | grep -v amdgpu.ids | head -11
PCRL_HIP_LIB=_ab/libpcrl_hip_stamps.so python tools/bwdg_stamps.py --B 256 2>&1 | grep -v amdgpu.ids | head -11
PCRL_HIP_LIB=_ab/libpcrl_hip_new.so python -m pytest tests/test_encoder_bwd_gpu.py -q 2>&1 | tail -3
