#!/bin/bash
# Round 5: the K1 per-rank shares (32 / 64 clouds) and K3's share with the encoder backward's team kernel (default: small launches) against
# the points / wgrad / reduce launches for every size (PCRL_BWD_PATHS=legacy), same box, alternating.
#   bash tools/r5_ab4.sh [steps]
STEPS=${1:-400}
export TMPDIR=/tmp
mkdir -p gpurun_out/ab4
run() {
  local name=$1 wl=${2%%:*} bt=${2##*:}
  local extra=""; [ "$bt" != "0" ] && extra="--batch $bt"
  local env=""; [ "$name" = "launches" ] && env="PCRL_BWD_PATHS=legacy"
  env $env python bench.py --steps $STEPS --warmup 30 --workload $wl $extra 2>/dev/null | tail -1 > gpurun_out/ab4/${name}_${wl}_${bt}.json
  python - "$name" "$wl" "$bt" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab4/{sys.argv[1]}_{sys.argv[2]}_{sys.argv[3]}.json"))
print(f"{sys.argv[1]:10s} {sys.argv[2]}:{sys.argv[3]:4s} ms_per_step {d['ms_per_step']:.4f}  value {d['value']:.1f}")
PY
}
for rep in 1 2; do
  for cfg in k1:32 k1:64 k1:128 k3:128 k1:0; do
    run team $cfg
    run launches $cfg
  done
done
