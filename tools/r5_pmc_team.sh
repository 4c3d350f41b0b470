export TMPDIR=/tmp
for paths in team legacy; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  OUT=gpurun_out/pmc_team/${paths}_$ctr; rm -rf $OUT; mkdir -p $OUT
  PCRL_BWD_PATHS=$paths rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT -- python3 tools/bench_encoder.py --B 32 --N 1024 --iters 10 > $OUT/log.txt 2>&1
  python3 - $OUT $paths $ctr <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "bwd" in r["Kernel_Name"]: acc[r["Kernel_Name"].replace("pcrl::", "").replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
print(sys.argv[2], sys.argv[3], {k: round(sum(v[len(v)//2:]) / len(v[len(v)//2:]) / 1024, 2) for k, v in acc.items()}, "MB (KB counter / 1024)")
PY
done
done
