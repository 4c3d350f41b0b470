"""Per-launch HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes on
gfx950: MI355X_MICROARCH.md, rocprofv3 PMC slots).  FETCH_SIZE is doubled (gfx950 reports half of a 16 B/lane streaming
read, same guide, HBM section); WRITE_SIZE is taken as reported (uncalibrated).
    python tools/pmc_traffic.py <fetch dir> <write dir> <kernel substring> <out.json>"""
import csv
import glob
import json
import sys


def mean_counter(d, kernel, counter):
    vals = []
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch_kb, n_f = mean_counter(sys.argv[1], sys.argv[3], "FETCH_SIZE")
write_kb, n_w = mean_counter(sys.argv[2], sys.argv[3], "WRITE_SIZE")
out = {"kernel": sys.argv[3], "launches_sampled": [n_f, n_w], "FETCH_SIZE_KB_reported": fetch_kb, "WRITE_SIZE_KB_reported": write_kb,
       "fetch_bytes_corrected_x2": None if fetch_kb is None else 2 * 1024 * fetch_kb, "write_bytes": None if write_kb is None else 1024 * write_kb}
if fetch_kb is not None and write_kb is not None:
    out["hbm_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
