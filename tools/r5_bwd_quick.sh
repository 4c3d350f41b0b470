#!/bin/bash
# Round 5: encoder backward tests, then team kernel vs round-3/4 launches (tools/r5_bwd_ab.sh), then the team kernel's phase stamps.
timeout 600 python -m pytest tests/test_encoder_bwd_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/r5_bwd_ab.sh 2>&1 | grep -E "^==|fused_kernel|points_kernel|wgrad_kernel|reduce|sum of"
for B in 32 256; do PCRL_HIP_LIB=_ab/stamps/libpcrl_hip.so python tools/fused_stamps.py --B $B 2>&1 | tail -17; done
