#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/v5; mkdir -p $OUT
bash tools/r6_gemm_exp.sh "r5 ga ga:1 gb gc gc:1 gd r5" 2>&1 | tail -40
B="--no-cpu-baseline --no-experimental --no-extra-workloads"
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"],4), "ms", round(d["value"],1), "gemm", {k: round(v,3) if isinstance(v,float) else v for k,v in (d.get("roofline_gemm") or {}).items() if k in ("us_per_step",)})
PY
}
for rep in 1 2; do
  for spec in "k1 256 600" "k3 1024 200" "k3 128 600"; do
    set -- $spec
    for arm in r5 ga ga:1 gc gc:1; do
      lib=${arm%%:*}; exp=0; [[ $arm == *:* ]] && exp=${arm#*:}
      unset PCRL_GEMM_EXP; export PCRL_HIP_LIB=$PWD/_abship/$lib/libpcrl_hip.so
      [ $exp != 0 ] && export PCRL_GEMM_EXP=$exp
      extra=""; [ $lib = r5 ] && extra="--set-fused publish_first=0"
      python bench.py --workload $1 --batch $2 --steps $3 --warmup 100 $B $extra > $OUT/b.json 2> $OUT/b.err || tail -3 $OUT/b.err
      show $OUT/b.json "$1 b$2 $arm rep$rep"
    done
  done
done
unset PCRL_HIP_LIB PCRL_GEMM_EXP
bash tools/r6_amdlog_hunt.sh 10
