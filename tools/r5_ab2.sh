#!/bin/bash
# Round 5 A/B on ONE box: libraries named on the command line (r4 = _ab/r4, r5a = _ab/r5a, new = the tree's) on the given workloads.
#   tools/r5_ab2.sh <tag> "<libs>" "<workload:batch ...>"
set -u
TAG=${1:-r5ab2}; LIBS=${2:-"r4 new"}; SPECS=${3:-"k2:256 k3:1024 k4:512"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for spec in $SPECS; do
  wl=${spec%%:*}; b=${spec##*:}
  for lib in $LIBS; do
    unset PCRL_GEMM_PATHS
    if [ $lib = new ]; then unset PCRL_HIP_LIB; elif [ $lib = newlegacy ]; then unset PCRL_HIP_LIB; export PCRL_GEMM_PATHS=legacy; else export PCRL_HIP_LIB=$PWD/_ab/$lib/libpcrl_hip.so; fi
    steps=300; [ $wl = k3 ] && [ $b = 1024 ] && steps=120; [ $wl = k4 ] && steps=80
    python bench.py --workload $wl --batch $b --steps $steps --warmup 30 --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/b.json 2> $OUT/b.err || tail -3 $OUT/b.err
    python - $OUT/b.json "$wl b$b $lib" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"],4), "ms", round(d["value"],1), {k:(v["launches"], round(v["avg_ms"]*1e3,1)) for k,v in d.get("kernels_ms",{}).items() if k in ("gemm","encoder_fwd","encoder_bwd")})
PY
  done
done
done
