#!/bin/bash
# Every kernel of the (eager) update step against both rooflines: three rocprofv3 passes of the same command -- MFMA-busy counters,
# FETCH_SIZE, WRITE_SIZE (separate passes on gfx950, MI355X_MICROARCH.md) -- folded per kernel by tools/step_ledger.py.
#   tools/step_ledger.sh [workload]  ->  gpurun_out/ledger/ledger_<wl>.md
export TMPDIR=/tmp
WL=${1:-k1}
OUT=gpurun_out/ledger; mkdir -p $OUT
CMD="bench.py --workload $WL --no-graphs --steps 20 --warmup 5 --no-cpu-baseline --no-experimental --no-extra-workloads --device-warmup-seconds 0"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf $OUT/${WL}_p$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${WL}_p$i -- python3 $CMD > $OUT/${WL}_p$i.log 2>&1
done
python3 tools/step_ledger.py $OUT/${WL}_p1 $OUT/${WL}_p2 $OUT/${WL}_p3 "$CMD" > $OUT/ledger_$WL.md
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/ledger_$WL.md
