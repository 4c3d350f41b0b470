#!/bin/bash
# Round 5: stand-alone backward (launch schedule of rounds 3-4) of two libraries on one box: _ab/prev against the tree's.
export TMPDIR=/tmp PCRL_BWD_PATHS=legacy
for rep in 1 2; do
for cfg in "--B 256 --N 1024" "--B 128 --N 1200 --c1 128 --seg 1" "--B 512 --N 1200"; do
  for lib in prev new; do
    if [ $lib = prev ]; then export PCRL_HIP_LIB=$PWD/_ab/prev/libpcrl_hip.so; else unset PCRL_HIP_LIB; fi
    OUT=gpurun_out/ab5/$(echo $cfg | tr -d ' -')_$lib; rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_encoder.py $cfg --iters 40 > $OUT/log.txt 2>&1
    python3 - "$OUT" "$cfg [$lib]" <<'PY'
import csv, glob, sys
rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]))}
pts = [v for k, v in rows.items() if "points_kernel" in k]; wg = [v for k, v in rows.items() if "wgrad_kernel" in k]
print(f"{sys.argv[2]:50s} points {pts[0]:7.1f} us  wgrad {wg[0]:7.1f} us")
PY
    find $OUT -name "*.csv" ! -name "*kernel_stats.csv" -delete
  done
done
done
