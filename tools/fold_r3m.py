"""Copies what tools/r3_measure.sh / tools/r4_measure.sh left under gpurun_out/<dir> (and gpurun_out/pmc) into profiles/ (tracked),
named per round.    python tools/fold_r3m.py [round prefix, default r03] [source dir under gpurun_out, default r3m]"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
srcdir = sys.argv[2] if len(sys.argv) > 2 else "r3m"
src, dst = os.path.join(ROOT, "gpurun_out", srcdir), os.path.join(ROOT, "profiles")


def last_json_line(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return lines[-1] if lines else None


pairs = [(f"bench_{wl}.json", f"{rnd}_bench_{wl}.json") for wl in ("k1", "k2", "k3", "k4")]
pairs += [("bench_k1_driver_style.json", f"{rnd}_bench_k1_driver_style_20steps.json")]
pairs += [(f"share_k1_b{b}.json", f"{rnd}_share_k1_b{b}.json") for b in (128, 64, 32)] + [("share_k3_b128.json", f"{rnd}_share_k3_b128.json")]
pairs += [(f"sre_cap{c}_{w}.json", f"{rnd}_sre_cap{c}_{w}.json") for c in (0, 1) for w in ("k1", "k3b128")]
for a, b in pairs:
    p = os.path.join(src, a)
    if os.path.exists(p) and last_json_line(p):
        open(os.path.join(dst, b), "w").write(last_json_line(p) + "\n")
        d = json.loads(last_json_line(p))
        print(f"{b:44s} {d['value']:9.1f} {d['unit']}  {d['ms_per_step']:.4f} ms")
for wl in ("k1", "k2", "k3", "k4"):
    p = os.path.join(src, f"kernel_stats_{wl}.csv")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{rnd}_kernel_stats_{wl}.csv"))
for a, b in (("pmc_summary.txt", f"{rnd}_pmc_encoder_kernels_raw.txt"), ("membound.md", f"{rnd}_membound_kernels.md")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for name in sorted(os.listdir(src)) if os.path.isdir(src) else []:
    if name.startswith("timeline_") and name.endswith(".txt"):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{rnd}_{name}"))
for wl in ("k1", "k2", "k4"):
    p = os.path.join(ROOT, "gpurun_out", "pmc", f"traffic_{wl}.json")
    if os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(src):
        shutil.copy(p, os.path.join(dst, f"{rnd}_pmc_traffic_{wl}.json"))
if srcdir == "r3m":
    subprocess.call([sys.executable, os.path.join(ROOT, "tools", "fold_profiles.py"), "r3m", rnd])
