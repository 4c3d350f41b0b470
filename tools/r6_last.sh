#!/bin/bash
# Round 6, closing visit on the shipped library (no rebuild): the step timelines with the corrected tool (tools/step_timeline.py now prints the
# median two-step span), 10 launches of the eight-rank rehearsal, and smoke().
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6l; mkdir -p $OUT
sha256sum pointcloud_rl_amd/libpcrl_hip.so | tee $OUT/libpcrl_hip.sha256
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/r3_timeline.sh $OUT/tl_k1 > $OUT/timeline_k1.txt 2>&1
bash tools/r3_timeline.sh $OUT/tl_b32 --batch 32 > $OUT/timeline_k1_b32.txt 2>&1
bash tools/r3_timeline.sh $OUT/tl_k3b128 --workload k3 --batch 128 > $OUT/timeline_k3_b128.txt 2>&1
tail -1 $OUT/timeline_k1.txt $OUT/timeline_k1_b32.txt $OUT/timeline_k3_b128.txt
grep -h "replay_gather" $OUT/timeline_k1_b32.txt
DRY="--steps 20 --warmup 5 --no-extra-workloads --no-cpu-baseline --replay-capacity 512"
fails=0; t0=$(date +%s)
for i in $(seq 1 ${LAUNCHES:-10}); do
  timeout 300 python bench.py --dry-run-ranks 8 $DRY > $OUT/F_$i.out 2> $OUT/F_$i.err; rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "rehearsal $i rc=$rc $(grep -m1 -o 'HSA_STATUS[A-Z_]*' $OUT/F_$i.err)"; else rm -f $OUT/F_$i.err $OUT/F_$i.out; fi
done
echo "== bench --dry-run-ranks 8 on $(cut -c1-8 $OUT/libpcrl_hip.sha256): $fails failed of ${LAUNCHES:-10} in $(( $(date +%s) - t0 )) s ==" | tee $OUT/final_loops.txt
