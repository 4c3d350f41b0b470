#!/bin/bash
# Cache-side counters of one head-GEMM shape on the 32x32 split-K tiles (M = 256): L2 requests / hits / misses, the texture
# addresser's busy cycles, L1 accesses and its requests to L2 -- what bounds the k loop.  -> gpurun_out/gemm_pmc/summary.txt
export TMPDIR=/tmp
OUT=gpurun_out/gemm_pmc; rm -rf $OUT; mkdir -p $OUT
for shape in "dh1 x2" "dW1 x2"; do
  tag=$(echo "$shape" | tr -c 'a-zA-Z0-9\n' '_')
  i=0
  # (a pass with TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum aborted inside rocprofv3 and hung until the
  # box's limit: every pass runs under its own timeout, the TA counters are left out)
  for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    GEMM_M=256 timeout 75 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${tag}_$i -- python3 tools/probes/gemm_one.py "$shape" 20 1073741824 > $OUT/${tag}_$i.log 2>&1
    echo "== $shape [$set]"; python3 tools/pmc_kernel_summary.py $OUT/${tag}_$i gemm_f32_kernel
  done
  f=$(find $OUT/${tag}_1 -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(sys.argv[1])) if "gemm_f32_kernel" in r["Kernel_Name"]]
print("mean duration under the counters: %.1f us over %d launches" % (sum(d)/len(d), len(d)))
PY
done > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
