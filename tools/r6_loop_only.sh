#!/bin/bash
# N more consecutive launches of the eight-rank rehearsal on the shipped library (HSA_ENABLE_SDMA=0 + start lock: what bench.py sets itself).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6loop; mkdir -p $OUT
N=${1:-20}
sha256sum pointcloud_rl_amd/libpcrl_hip.so | tee $OUT/libpcrl_hip.sha256
DRY="--steps 20 --warmup 5 --no-extra-workloads --no-cpu-baseline --replay-capacity 512"
fails=0; t0=$(date +%s)
for i in $(seq 1 $N); do
  timeout 300 python bench.py --dry-run-ranks 8 $DRY > $OUT/F_$i.out 2> $OUT/F_$i.err; rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "rehearsal $i rc=$rc $(grep -m1 -o 'HSA_STATUS[A-Z_]*' $OUT/F_$i.err)"; else rm -f $OUT/F_$i.err $OUT/F_$i.out; fi
done
echo "== bench --dry-run-ranks 8 on $(cut -c1-8 $OUT/libpcrl_hip.sha256): $fails failed of $N in $(( $(date +%s) - t0 )) s ==" | tee $OUT/final_loops.txt
