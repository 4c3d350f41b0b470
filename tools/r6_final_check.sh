#!/bin/bash
# Round 6, last visit: the whole GPU suite a second time on the shipped binary (flakiness), 30 consecutive locked rehearsals on ONE box, the soak run.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6f; mkdir -p $OUT
make -C pointcloud_rl_amd/csrc > $OUT/make.log 2>&1; echo "make rc=$? ($(grep -c 'hipcc.*-c ' $OUT/make.log) objects recompiled on the box)"
sha256sum pointcloud_rl_amd/libpcrl_hip.so | tee $OUT/libpcrl_hip.sha256
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
echo "library the suite ran on: $(cat $OUT/libpcrl_hip.sha256)" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
DRY="--steps 20 --warmup 5 --no-extra-workloads --no-cpu-baseline --replay-capacity 512"
fails=0
for i in $(seq 1 30); do
  timeout 300 python bench.py --dry-run-ranks 8 $DRY > $OUT/L_$i.out 2> $OUT/L_$i.err; rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "locked rehearsal $i rc=$rc $(grep -m1 -o 'HSA_STATUS[A-Z_]*' $OUT/L_$i.err)"; else rm -f $OUT/L_$i.err $OUT/L_$i.out; fi
done
echo "== bench --dry-run-ranks 8 (start lock on), one box: $fails failed of 30 ==" | tee $OUT/dry30.txt
timeout 1500 python tools/soak.py > $OUT/soak.txt 2>&1; echo "soak rc=$?"; tail -4 $OUT/soak.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err; echo "driver-style bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6f/bench_driver_style.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["step"]["backward_tiles_per_rank"], d["step"]["frac"], d["roofline"]["frac"], d["roofline_gemm"]["frac"], d["roofline_gemm"]["matrix_busy"])
PY
