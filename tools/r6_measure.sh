#!/bin/bash
# Round-6 measurement visit (gpurun_out/r6m/): the whole `-m gpu` suite on the final binary (its sha256 next to the log), bench lines + kernel
# statistics of K1..K4, the driver-style line, per-rank shares, PMC traffic of the encoder forward, step timelines, the K1 step ledger.
# tools/fold_r3m.py r06 r6m copies to profiles/.
set -u
OUT=gpurun_out/r6m; mkdir -p $OUT; export TMPDIR=/tmp
make -C pointcloud_rl_amd/csrc > $OUT/make.log 2>&1; echo "make rc=$? ($(grep -c 'hipcc.*-c ' $OUT/make.log) objects recompiled on the box)"
sha256sum pointcloud_rl_amd/libpcrl_hip.so | tee $OUT/libpcrl_hip.sha256
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
echo "library the suite ran on: $(cat $OUT/libpcrl_hip.sha256)" >> $OUT/pytest.log
python3 tools/final_binary_json.py $OUT/pytest.log profiles/r06_gpu_suite.txt $OUT/final_binary.json || echo "suite not green: no final_binary.json"
tail -4 $OUT/pytest.log
# counters first, folded into profiles/ ON THE BOX: the bench lines below then carry the HBM traffic / matrix-busy figures of THIS library
for wl in k1 k2 k4; do bash tools/pmc_traffic.sh $wl > $OUT/pmc_traffic_$wl.log 2>&1; cp gpurun_out/pmc/traffic_$wl.json profiles/r06_pmc_traffic_$wl.json; done
bash tools/step_ledger.sh k1 > $OUT/step_ledger_k1.md 2>&1
cp $OUT/step_ledger_k1.md profiles/r06_step_ledger_k1.md; python tools/gemm_busy_json.py profiles/r06_step_ledger_k1.md profiles/r06_gemm_matrix_busy.json > /dev/null; cp profiles/r06_gemm_matrix_busy.json $OUT/
for wl in k1 k2 k3 k4; do
  steps=2000; warm=500
  [ "$wl" = "k3" ] && { steps=400; warm=100; }
  [ "$wl" = "k4" ] && { steps=200; warm=40; }
  extra="--no-extra-workloads"; [ "$wl" = "k1" ] && extra=""
  python bench.py --workload $wl --steps $steps --warmup $warm $extra > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "bench $wl rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$wl -- python3 bench.py --workload $wl --steps 100 --warmup 30 --no-cpu-baseline --no-experimental --no-extra-workloads --device-warmup-seconds 0 > $OUT/prof_$wl.log 2>&1
  find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$wl.csv
  find $OUT/prof_$wl -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $OUT/prof_$wl -name "*.db" -delete
done
python bench.py --steps 20 --warmup 5 > $OUT/bench_k1_driver_style.json 2> /dev/null
for b in 128 64 32; do python bench.py --batch $b --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/share_k1_b$b.json 2>/dev/null; done
python bench.py --workload k3 --batch 128 --steps 1000 --warmup 200 --no-cpu-baseline --no-extra-workloads > $OUT/share_k3_b128.json 2>/dev/null
bash tools/r3_timeline.sh $OUT/tl_k1 > $OUT/timeline_k1.txt 2>&1
bash tools/r3_timeline.sh $OUT/tl_b32 --batch 32 > $OUT/timeline_k1_b32.txt 2>&1
bash tools/r3_timeline.sh $OUT/tl_k3b128 --workload k3 --batch 128 > $OUT/timeline_k3_b128.txt 2>&1
rm -rf $OUT/tl_*
ls $OUT
