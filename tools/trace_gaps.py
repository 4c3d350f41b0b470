"""Gaps between consecutive kernels of the replayed update step (rocprofv3 --kernel-trace csv): where the device waits for the host
or for a cross-stream edge.   python tools/trace_gaps.py <dir with *_kernel_trace.csv> [min gap us]"""
import csv, glob, os, sys
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "--list" else 8.0
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(({"name": r["Kernel_Name"].split("(")[0][-48:], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]), "q": r.get("Queue_Id", "")}
               for r in csv.DictReader(open(f))), key=lambda r: r["s"])
# steady state: the last replay_gather_kernel-delimited steps before the eager pass; take steps 60..80 of the trace
starts = [i for i, r in enumerate(rows) if "replay_gather" in r["name"]]
lo, hi = starts[len(starts) // 2], starts[len(starts) // 2 + 2]
print(f"{len(starts)} steps in the trace; steps {len(starts)//2} and {len(starts)//2+1}: {(rows[hi]['s'] - rows[lo]['s'])/2e3:.1f} us per step")
if "--list" in sys.argv:
    for i in range(lo, hi):
        r = rows[i]
        print(f"  {(r['s'] - rows[lo]['s'])/1e3:8.1f} us  +{(r['e'] - r['s'])/1e3:6.1f} us  gap {(r['s'] - rows[i-1]['e'])/1e3:5.1f}  {r['name']}")
busy_end = rows[lo]["e"]
for i in range(lo + 1, hi + 1):
    r = rows[i]
    gap = (r["s"] - busy_end) / 1e3
    if gap > thr:
        print(f"  gap {gap:6.1f} us before {r['name']}  (queue {r['q']}; previous: {rows[i-1]['name']} on queue {rows[i-1]['q']})")
    busy_end = max(busy_end, r["e"])
