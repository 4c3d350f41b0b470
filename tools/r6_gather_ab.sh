#!/bin/bash
# Round 6: the metrics gathered by the first workgroup of the published optimizer pass (one launch fewer): parity tests, then rider on / off, same library
set -u
export TMPDIR=/tmp
OUT=gpurun_out/gab; mkdir -p $OUT
python -m pytest tests/test_update_step_gpu.py tests/test_dense_tail_gpu.py tests/test_fullsize_parity_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
python -m pytest tests/test_data_parallel_gpu.py -m gpu -x -q -k "sharded or segmented or rccl_single or two_ranks_prints" > $OUT/dp.log 2>&1; echo "dp tests rc=$?"; tail -2 $OUT/dp.log
B="--no-cpu-baseline --no-experimental --no-extra-workloads"
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"],4), "ms", round(d["value"],1))
PY
}
for rep in 1 2 3; do
  for spec in "k1 256 600" "k1 32 1500" "k3 128 600"; do
    set -- $spec
    for arm in on off; do
      extra=""; [ $arm = off ] && extra="--set-fused gather_rider=0"
      python bench.py --workload $1 --batch $2 --steps $3 --warmup 100 $B $extra > $OUT/b.json 2> $OUT/b.err || tail -3 $OUT/b.err
      show $OUT/b.json "$1 b$2 gather-rider-$arm rep$rep"
    done
  done
done
