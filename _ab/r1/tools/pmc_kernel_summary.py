"""Mean per-launch value of every collected counter for kernels matching a substring (rocprofv3 --pmc ... --output-format csv).
    python tools/pmc_kernel_summary.py <dir> <kernel substring>"""
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if sys.argv[2] in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:36s} {sum(v) / len(v):16.1f}  (n={len(v)})")
