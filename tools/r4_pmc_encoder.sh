#!/bin/bash
# PMC counters of the encoder kernels, stand-alone launches, counters in their own passes (round 4's build) -> gpurun_out/r4m/pmc_summary.txt
set -u
OUT=gpurun_out/r4m; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 tools/bench_encoder.py --B 256 --N 1024 --iters 10 > $OUT/pmc$i.log 2>&1
  for k in encoder_fwd_kernel encoder_bwdg_points_kernel encoder_bwdg_wgrad_kernel encoder_bwdg_prep_kernel encoder_bwdg_reduce_kernel; do echo "== $k [$set]"; python3 tools/pmc_kernel_summary.py $OUT/pmc$i $k; done
  find $OUT/pmc$i -name "*.csv" -delete; find $OUT/pmc$i -name "*.db" -delete
done > $OUT/pmc_summary.txt 2>&1
tail -40 $OUT/pmc_summary.txt
