#!/bin/bash
# Round 6, last visit: the final configuration of the rehearsal (HSA_ENABLE_SDMA=0 + start lock): the data-parallel test file twice in the pytest
# context that hung with host-staged collectives, 40 consecutive launches, then the whole suite + the driver-style bench line.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6g; mkdir -p $OUT
for i in 1 2; do
  timeout 1200 python -m pytest tests/test_data_parallel_gpu.py -m gpu -x -q > $OUT/dp_$i.log 2>&1; echo "data-parallel file run $i rc=$?"; tail -2 $OUT/dp_$i.log | cut -c1-200
done
DRY="--steps 20 --warmup 5 --no-extra-workloads --no-cpu-baseline --replay-capacity 512"
fails=0; t0=$(date +%s)
for i in $(seq 1 40); do
  timeout 300 python bench.py --dry-run-ranks 8 $DRY > $OUT/F_$i.out 2> $OUT/F_$i.err; rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "rehearsal $i rc=$rc $(grep -m1 -o 'HSA_STATUS[A-Z_]*' $OUT/F_$i.err) $(grep -o 'ranks failed.*' $OUT/F_$i.err | tail -1)"; else rm -f $OUT/F_$i.err $OUT/F_$i.out; fi
done
echo "== bench --dry-run-ranks 8 (HSA_ENABLE_SDMA=0 + start lock), one box: $fails failed of 40 in $(( $(date +%s) - t0 )) s ==" | tee $OUT/final_loops.txt
make -C pointcloud_rl_amd/csrc > $OUT/make.log 2>&1; echo "make rc=$? ($(grep -c 'hipcc.*-c ' $OUT/make.log) objects recompiled on the box)"
sha256sum pointcloud_rl_amd/libpcrl_hip.so | tee $OUT/libpcrl_hip.sha256
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
echo "library the suite ran on: $(cat $OUT/libpcrl_hip.sha256)" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err; echo "driver-style bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6g/bench_driver_style.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["step"]["backward_tiles_per_rank"], d["step"]["frac"], d["roofline"]["frac"], d["roofline_gemm"]["frac"], d["roofline_gemm"]["matrix_busy"])
PY
