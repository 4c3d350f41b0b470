#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/v4; mkdir -p $OUT
bash tools/r6_gemm_exp.sh 2>&1 | tail -60
# name the faulting kernel: the runtime's own launch log on unlocked rehearsals (kept only for failing runs)
DRY="--steps 20 --warmup 5 --no-extra-workloads --no-cpu-baseline --replay-capacity 512 --start-lock 0"
for i in $(seq 1 10); do
  AMD_LOG_LEVEL=3 timeout 300 python bench.py --dry-run-ranks 8 $DRY > $OUT/amdlog_$i.out 2> $OUT/amdlog_$i.err; rc=$?
  if [ $rc -ne 0 ] && grep -q HSA_STATUS $OUT/amdlog_$i.err; then
    echo "amdlog run $i FAILED: $(grep -m1 -o 'HSA_STATUS[A-Z_]*' $OUT/amdlog_$i.err)"
    python3 - $OUT/amdlog_$i.err $OUT/amdlog_fail_$i.txt <<'PY'
import re, sys, collections
lines = open(sys.argv[1], errors="replace").read().splitlines()
# log lines carry [pid N tid 0x...]; keep, per pid, the last kernel launches; report the pid whose log ends first / holds the abort
by = collections.defaultdict(list)
abort_at = None
for n, l in enumerate(lines):
    m = re.search(r"pid[: ]+(\d+)", l)
    if "aborting with error" in l:
        abort_at = n
    if m and ("ShaderName" in l or "hipLaunchKernel" in l or "hipModuleLaunch" in l or "KernelExecution" in l or "hipMemcpy" in l or "hipMemset" in l):
        by[m.group(1)].append((n, l[:400]))
out = [f"total lines {len(lines)}, abort line {abort_at}", lines[abort_at][:400] if abort_at is not None else "no abort line"]
ctx = lines[max(0, (abort_at or 0) - 60):(abort_at or 0) + 5]
out += ["---- 60 lines before the abort ----"] + [c[:300] for c in ctx]
for pid, ls in by.items():
    last = [x for x in ls if abort_at is None or x[0] <= abort_at][-12:]
    out += [f"---- pid {pid}: {len(ls)} launch-type lines; last before the abort ----"] + [f"{n}: {l}" for n, l in last]
open(sys.argv[2], "w").write("\n".join(out) + "\n")
PY
    head -c 6000 $OUT/amdlog_fail_$i.txt | head -90
  else
    echo "amdlog run $i rc=$rc"
  fi
  rm -f $OUT/amdlog_$i.err $OUT/amdlog_$i.out
done
