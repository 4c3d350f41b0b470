#!/bin/bash
# Per-kernel durations of the stand-alone encoder forward / backward (rocprofv3 --kernel-trace --stats of tools/bench_encoder.py).
#   tools/prof_encoder.sh [bench_encoder args...]
export TMPDIR=/tmp
OUT=gpurun_out/prof_encoder
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_encoder.py "$@" > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys
for row in list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])))[:8]:
    print(f"{row['Name'].replace('pcrl::', '').replace('void ', '')[:60]:60s} calls {row['Calls']:>4s}  avg {float(row['AverageNs']) / 1e3:8.1f} us")
PY
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" -delete
