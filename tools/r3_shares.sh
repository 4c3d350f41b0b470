#!/bin/bash
# A rank's share of the K1 step at 256 / 128 / 64 / 32 clouds (1 / 2 / 4 / 8 GPUs), graph-replayed, on this one GPU.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3sh}; mkdir -p $OUT
for b in 256 128 64 32; do
  python bench.py --batch $b --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/share_k1_b$b.json 2>/dev/null
  python - $OUT/share_k1_b$b.json $b <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("B", sys.argv[2], round(d["value"],1), "steps/s", round(d["ms_per_step"],4), "ms")
PY
done
