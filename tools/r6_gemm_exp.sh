#!/bin/bash
# Round 6: the GEMM shape table (rocprofv3 kernel durations) for library builds under _abship/ and planner experiments
#   tools/r6_gemm_exp.sh "lib[:exp] ..."     e.g. "r5 ga ga:1 gc gc:1"   (lib `tree` = the tree's own library)
set -u
OUT=gpurun_out/gx; mkdir -p $OUT; export TMPDIR=/tmp
export GEMM_MS=${GEMM_MS:-128,256}
ARMS=${1:-"r5 ga ga:1 gb gc gc:1 gd"}
n=0
for arm in $ARMS; do
  n=$((n+1))
  lib=${arm%%:*}; exp=0; [[ $arm == *:* ]] && exp=${arm#*:}
  tag=${n}_${lib}_e$exp
  unset PCRL_HIP_LIB PCRL_GEMM_EXP
  [ $lib != tree ] && export PCRL_HIP_LIB=$PWD/_abship/$lib/libpcrl_hip.so
  [ $exp != 0 ] && export PCRL_GEMM_EXP=$exp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$tag -- python3 tools/r5_gemm_probe.py run $OUT/labels_$tag.txt > $OUT/run_$tag.log 2>&1
  python3 tools/r5_gemm_probe.py fold $OUT/trace_$tag $OUT/labels_$tag.txt > $OUT/table_$tag.txt 2>&1; rm -rf $OUT/trace_$tag
done
python3 - $ARMS <<'PY'
import re, sys
arms=[f"{i+1}_" + (a.split(":")[0] + "_e" + (a.split(":")[1] if ":" in a else "0")) for i, a in enumerate(sys.argv[1:])]
tabs={}
for a in arms:
    tabs[a]={}
    for l in open(f"gpurun_out/gx/table_{a}.txt"):
        m=re.match(r"(.{28}) median\s+([\d.]+)", l)
        if m: tabs[a][m.group(1).strip()]=float(m.group(2))
print("shape".ljust(28), *[a.rjust(9) for a in arms])
for k in tabs[arms[0]]:
    if " K32" in k or " K256" in k: continue
    print(k.ljust(28), *[f"{tabs[a].get(k, float('nan')):9.2f}" for a in arms])
PY
