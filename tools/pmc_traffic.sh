#!/bin/bash
# HBM traffic of encoder_fwd_kernel inside the update step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE need separate
# passes on gfx950, MI355X_MICROARCH.md) of the eager bench command, then tools/pmc_traffic.py -> gpurun_out/pmc/traffic_<wl>.json
#   tools/pmc_traffic.sh [workload]
export TMPDIR=/tmp
WL=${1:-k1}
OUT=gpurun_out/pmc
rm -rf $OUT/fetch_$WL $OUT/write_$WL; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$WL -- python3 bench.py --workload $WL --no-graphs --steps 20 --warmup 5 --no-cpu-baseline > $OUT/fetch_$WL.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$WL -- python3 bench.py --workload $WL --no-graphs --steps 20 --warmup 5 --no-cpu-baseline > $OUT/write_$WL.log 2>&1
python3 tools/pmc_traffic.py $OUT/fetch_$WL $OUT/write_$WL encoder_fwd_kernel $OUT/traffic_$WL.json
find $OUT -name "*.csv" -size +2M -delete
