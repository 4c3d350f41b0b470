#!/bin/bash
# Round 5: stand-alone encoder backward, team kernel (encoder_bwd_fused.h) against the round-3/4 launches, per-kernel durations.
#   tools/r5_bwd_ab.sh            (K1 shapes at 32 / 64 / 128 / 256 clouds, K3's share and K3, K2's geometry)
export TMPDIR=/tmp
for cfg in "--B 32 --N 1024" "--B 64 --N 1024" "--B 128 --N 1024" "--B 256 --N 1024" "--B 128 --N 1200 --c1 128 --seg 1" "--B 512 --N 1200"; do
  for paths in fused legacy; do
    OUT=gpurun_out/bwd_ab/$(echo $cfg | tr -d ' -')_$paths
    rm -rf $OUT; mkdir -p $OUT
    if [ $paths = legacy ]; then export PCRL_BWD_PATHS=legacy; else export PCRL_BWD_PATHS=team; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_encoder.py $cfg --iters 30 > $OUT/log.txt 2>&1
    echo "== $cfg [$paths] $(tail -1 $OUT/log.txt)"
    python3 - "$OUT" <<'PY'
import csv, glob, sys
tot = 0.0
for row in csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])):
    if "bwd" in row["Name"]:
        print(f"   {row['Name'].replace('pcrl::', '').replace('void ', '')[:56]:56s} calls {row['Calls']:>4s}  avg {float(row['AverageNs']) / 1e3:8.1f} us")
        tot += float(row["AverageNs"]) / 1e3
print(f"   sum of the backward's kernels {tot:8.1f} us")
PY
    find $OUT -name "*.csv" ! -name "*kernel_stats.csv" -delete
  done
done
