#!/bin/bash
# One GPU-box visit that ENDS a round: the library is rebuilt from the tree's sources, the whole `-m gpu` suite runs on exactly that binary
# (kernel parity first, launcher rehearsals last: tests/conftest.py), and the binary's sha256 is written next to the log -- no csrc/ commit
# may follow this log.  Then the headline bench line and a kernel-trace profile of the same command.
#   tools/gpu_round.sh <tag> [workloads...]     (run through gpurun from the repo root; writes gpurun_out/<tag>/)
set -u
TAG=${1:-run}; shift || true
WLS=${@:-k1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
make -C pointcloud_rl_amd/csrc > $OUT/make.log 2>&1; echo "make rc=$? ($(grep -c hipcc $OUT/make.log) compile/link commands: 1 = only the link check, the shipped objects were current)"
sha256sum pointcloud_rl_amd/libpcrl_hip.so | tee $OUT/libpcrl_hip.sha256
git -C . rev-parse HEAD 2>/dev/null | tee $OUT/head.txt || true
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
echo "library the suite ran on: $(cat $OUT/libpcrl_hip.sha256)" >> $OUT/pytest.log
python3 tools/final_binary_json.py $OUT/pytest.log profiles/r06_gpu_suite.txt $OUT/final_binary.json || echo "suite not green: no final_binary.json"
for wl in $WLS; do
  steps=2000; warm=500
  [ "$wl" = "k3" ] && { steps=400; warm=100; }
  [ "$wl" = "k4" ] && { steps=200; warm=40; }
  python bench.py --workload $wl --steps $steps --warmup $warm > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "bench $wl rc=$?"
  tail -c 1500 $OUT/bench_$wl.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$wl -- python3 bench.py --workload $wl --steps 100 --warmup 30 --no-cpu-baseline > $OUT/prof_$wl.log 2>&1
  find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$wl.csv
  find $OUT/prof_$wl -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $OUT/prof_$wl -name "*.db" -delete
  head -25 $OUT/kernel_stats_$wl.csv
done
