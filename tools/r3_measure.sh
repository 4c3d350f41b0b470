#!/bin/bash
# Round-3 measurement visit: bench lines and kernel statistics of K1..K4, per-rank shares, the exchange schedule's cost over a
# one-rank RCCL group, PMC counters of the encoder kernels, the memory-shaped kernels.  Everything lands in gpurun_out/r3m/.
set -u
OUT=gpurun_out/r3m; mkdir -p $OUT; export TMPDIR=/tmp
for wl in k1 k2 k3 k4; do
  steps=2000; warm=500
  [ "$wl" = "k3" ] && { steps=400; warm=100; }
  [ "$wl" = "k4" ] && { steps=200; warm=40; }
  extra="--no-extra-workloads"; [ "$wl" = "k1" ] && extra=""
  python bench.py --workload $wl --steps $steps --warmup $warm $extra > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "bench $wl rc=$?"
  # kernel statistics of the same command without the device warm-up launches (they would share the encoder kernel's row)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$wl -- python3 bench.py --workload $wl --steps 100 --warmup 30 --no-cpu-baseline --no-experimental --no-extra-workloads --device-warmup-seconds 0 > $OUT/prof_$wl.log 2>&1
  find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$wl.csv
  find $OUT/prof_$wl -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $OUT/prof_$wl -name "*.db" -delete
done
python bench.py --steps 20 --warmup 5 > $OUT/bench_k1_driver_style.json 2> /dev/null
# one rank's share of a multi-GPU run, on this one GPU
for b in 128 64 32; do python bench.py --batch $b --no-cpu-baseline --no-experimental --no-extra-workloads > $OUT/share_k1_b$b.json 2>/dev/null; done
python bench.py --workload k3 --batch 128 --steps 1000 --warmup 200 --no-cpu-baseline --no-extra-workloads > $OUT/share_k3_b128.json 2>/dev/null
# the data-parallel schedule over a one-rank RCCL group: all-reduces captured in the step's graph (default) / eager between segments
for cap in 1 0; do
  PCRL_CAPTURE_EXCHANGE=$cap python bench.py --single-rank-exchange --backend nccl --steps 1000 --warmup 200 --no-cpu-baseline > $OUT/sre_cap${cap}_k1.json 2> $OUT/sre_cap${cap}_k1.err
  PCRL_CAPTURE_EXCHANGE=$cap python bench.py --single-rank-exchange --backend nccl --workload k3 --batch 128 --steps 1000 --warmup 200 --no-cpu-baseline > $OUT/sre_cap${cap}_k3b128.json 2> $OUT/sre_cap${cap}_k3b128.err
done
# PMC counters of the encoder kernels (stand-alone launches; counters in their own passes)
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 tools/bench_encoder.py --B 256 --N 1024 --iters 10 > $OUT/pmc$i.log 2>&1
  for k in encoder_fwd_kernel encoder_bwdg_points_kernel encoder_bwdg_wgrad_kernel; do echo "== $k [$set]"; python3 tools/pmc_kernel_summary.py $OUT/pmc$i $k; done
  find $OUT/pmc$i -name "*.csv" -size +1M -delete; find $OUT/pmc$i -name "*.db" -delete
done > $OUT/pmc_summary.txt 2>&1
bash tools/pmc_traffic.sh k1 > $OUT/pmc_traffic_k1.log 2>&1
bash tools/pmc_traffic.sh k4 > $OUT/pmc_traffic_k4.log 2>&1
python tools/bench_membound.py --md $OUT/membound.md > $OUT/membound.log 2>&1
python tools/bench_acting.py > $OUT/acting.log 2>&1
ls $OUT
