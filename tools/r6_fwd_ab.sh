#!/bin/bash
# Round 6: the forward kernel's explicit LDS window bases (no scratch reload inside the conv2 MFMA stream) against the library without them, same box
set -u
export TMPDIR=/tmp
OUT=gpurun_out/fwdab; mkdir -p $OUT
python -m pytest tests/test_encoder_fwd_gpu.py -m gpu -x -q > $OUT/fwd_tests.log 2>&1; echo "forward tests rc=$?"; tail -2 $OUT/fwd_tests.log
B="--no-cpu-baseline --no-experimental --no-extra-workloads"
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"],4), "ms", round(d["value"],1), "fwd frac", round(d["roofline"]["frac"],4), "fwd ms", round(d["roofline"]["avg_launch_ms"],4))
PY
}
for rep in 1 2 3; do
  for spec in "k1 256 600" "k3 1024 200" "k4 512 100"; do
    set -- $spec
    for arm in base new; do
      unset PCRL_HIP_LIB; [ $arm = base ] && export PCRL_HIP_LIB=$PWD/_abship/base/libpcrl_hip.so
      python bench.py --workload $1 --batch $2 --steps $3 --warmup 100 $B > $OUT/b.json 2> $OUT/b.err || tail -3 $OUT/b.err
      show $OUT/b.json "$1 b$2 $arm rep$rep"
    done
  done
done
