#!/bin/bash
# Counters of one head-GEMM shape on the round-5 kernel (M = 256): what the waves wait for.  -> gpurun_out/gemm_pmc5/summary.txt
export TMPDIR=/tmp
OUT=gpurun_out/gemm_pmc5; rm -rf $OUT; mkdir -p $OUT
for shape in "fwd1 1024->1024 x2" "dW1 | dh1 x2"; do
  tag=$(echo "$shape" | tr -c 'a-zA-Z0-9\n' '_')
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    GEMM_M=256 timeout 75 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${tag}_$i -- python3 tools/probes/gemm_one.py "$shape" 20 > $OUT/${tag}_$i.log 2>&1
    echo "== $shape [$set]"; python3 tools/pmc_kernel_summary.py $OUT/${tag}_$i gemm_f32_
  done
done > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
