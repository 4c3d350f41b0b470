#!/bin/bash
# Round 6: is K1's points launch (66 us) a one-round launch with a few straggler tiles?  The launch deals 32-point tiles over 1 024 wave slots
# (256 CUs x 4 waves of 512 registers); batch sizes around 256 move the tile count across that line.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/tiles; mkdir -p $OUT
for b in 208 224 240 248 256 272 288; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$b -- python3 bench.py --batch $b --steps 60 --warmup 30 --no-cpu-baseline --no-experimental --no-extra-workloads --device-warmup-seconds 0 > $OUT/b$b.json 2> $OUT/b$b.err
  f=$(find $OUT/p$b -name "*kernel_stats.csv" | head -1)
  python3 - $f $OUT/b$b.json $b <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
get = lambda key: next((float(r["AverageNs"]) / 1e3 for r in rows if key in r["Name"]), float("nan"))
print(f"B={sys.argv[3]:>4}: tiles {d['step']['backward_tiles_per_rank']:>5} of {d['step']['backward_wave_slots']} slots, active/cloud {d['step']['active_points_per_cloud']:.1f} | "
      f"points {get('bwdg_points'):.1f} us  wgrad {get('bwdg_wgrad'):.1f}  reduce {get('bwdg_reduce'):.1f}  | step {d['ms_per_step']:.4f} ms")
PY
  rm -rf $OUT/p$b
done
