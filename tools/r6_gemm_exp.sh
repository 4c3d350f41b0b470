#!/bin/bash
# Round 6: the GEMM shape table (rocprofv3 kernel durations) for the round-5 library, the tree's, and the planner experiments
set -u
OUT=gpurun_out/gx; mkdir -p $OUT; export TMPDIR=/tmp
export GEMM_MS=${GEMM_MS:-128,256,1024}
for arm in r5 new exp1 exp2 exp3; do
  unset PCRL_HIP_LIB PCRL_GEMM_EXP
  [ $arm = r5 ] && export PCRL_HIP_LIB=$PWD/_abship/r5/libpcrl_hip.so
  [ $arm = exp1 ] && export PCRL_GEMM_EXP=1
  [ $arm = exp2 ] && export PCRL_GEMM_EXP=2
  [ $arm = exp3 ] && export PCRL_GEMM_EXP=3
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$arm -- python3 tools/r5_gemm_probe.py run $OUT/labels_$arm.txt > $OUT/run_$arm.log 2>&1
  python3 tools/r5_gemm_probe.py fold $OUT/trace_$arm $OUT/labels_$arm.txt > $OUT/table_$arm.txt 2>&1; rm -rf $OUT/trace_$arm
done
python3 - <<'PY'
import re
arms=["r5","new","exp1","exp2","exp3"]
tabs={}
for a in arms:
    tabs[a]={}
    for l in open(f"gpurun_out/gx/table_{a}.txt"):
        m=re.match(r"(.{28}) median\s+([\d.]+)", l)
        if m: tabs[a][m.group(1).strip()]=float(m.group(2))
print("shape".ljust(28), *[a.rjust(7) for a in arms])
for k in tabs["new"]:
    if " K32" in k or " K256" in k: continue
    print(k.ljust(28), *[f"{tabs[a].get(k, float('nan')):7.2f}" for a in arms])
PY
