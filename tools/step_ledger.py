"""Per-kernel ledger of one update step from three rocprofv3 --pmc passes (tools/step_ledger.sh): launches per step, mean duration,
share of the step's kernel time, matrix-pipe busy fraction and HBM-side traffic -- each kernel against the roofline that bounds it.
    python tools/step_ledger.py <dir MFMA pass> <dir FETCH pass> <dir WRITE pass> "<command>"
Counter conventions (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the SIMDs; GRBM_GUI_ACTIVE summed over the
8 XCDs; FETCH_SIZE / WRITE_SIZE in KB, FETCH doubled (gfx950 reports half of a 16 B/lane streaming read), WRITE as reported; both count
fabric-side requests (Infinity-Cache hits included), i.e. they bound HBM traffic from above."""
import collections, csv, glob, re, sys

SIMDS, XCDS, HBM_PEAK = 1024, 8, 8.0e12


def short(name):
    n = re.sub(r"^void ", "", name).replace("pcrl::", "")
    return re.sub(r"\(.*$", "", n)[:64]


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"])
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return acc, dur


def mean(v):
    return sum(v) / len(v) if v else float("nan")


mf, dur = counters(sys.argv[1])
fe, _ = counters(sys.argv[2])
wr, _ = counters(sys.argv[3])
# one replay gather per update step (the metrics' gather rides the published Adam pass since round 6: adam_gather_kernel)
steps = len(dur.get("replay_gather_kernel", [])) or len(dur.get("gather_scalars_kernel", [])) or 1
total = sum(sum(v) for v in dur.values())
print(f"# Every kernel of the update step against both rooflines\n\n`{sys.argv[4]}` under `rocprofv3 --kernel-trace --pmc <set>`, three passes "
      f"(SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 | FETCH_SIZE | WRITE_SIZE); {steps} eager steps "
      f"(durations are those of the counter pass: serialised dispatches, a few per cent above the replayed graph's).  Matrix busy = "
      f"BUSY_CYCLES / ({SIMDS} SIMDs x GRBM_GUI_ACTIVE / {XCDS}); traffic: FETCH_SIZE and WRITE_SIZE as reported, fabric side (Infinity-Cache hits included: an upper bound of HBM "
      f"bytes); the rate columns are (FETCH + WRITE) / t and (2 FETCH + WRITE) / t -- gfx950 reports half of a 16 B/lane streaming read, "
      f"other access widths are uncalibrated (MI355X_MICROARCH.md), so the truth lies between the two; the last column is the larger one against 8 TB/s.\n")
print("| kernel | launches / step | mean us | share of kernel time | matrix pipe busy | FETCH_SIZE | WRITE_SIZE | (F + W) / t | (2 F + W) / t | of 8 TB/s |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    if sum(dur[k]) / total < 0.002:
        continue
    us = mean(dur[k])
    grbm, busy = mean(mf[k]["GRBM_GUI_ACTIVE"]), mean(mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
    frac = busy / (SIMDS * grbm / XCDS) if grbm == grbm and grbm > 0 else float("nan")
    f_kb, w_kb = mean(fe.get(k, {}).get("FETCH_SIZE", [])), mean(wr.get(k, {}).get("WRITE_SIZE", []))
    lo, hi = (f_kb + w_kb) * 1024 / (us * 1e-6), (2 * f_kb + w_kb) * 1024 / (us * 1e-6)
    print(f"| `{k}` | {len(dur[k]) / steps:.2f} | {us:.1f} | {100 * sum(dur[k]) / total:.1f} % | {100 * frac:.1f} % | {f_kb * 1024 / 1e6:.2f} MB | "
          f"{w_kb * 1024 / 1e6:.2f} MB | {lo / 1e12:.2f} TB/s | {hi / 1e12:.2f} TB/s | {100 * hi / HBM_PEAK:.1f} % |")
print(f"\nkernel time per step: {total / steps:.0f} us")
