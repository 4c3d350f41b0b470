"""Copies what a GPU-box visit left under gpurun_out/ into profiles/ (tracked): kernel statistics and bench lines of
tools/gpu_round.sh <tag>, PMC traffic of tools/pmc_traffic.sh.   python tools/fold_profiles.py <tag> [round prefix, default r02]"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r02")
src = os.path.join(ROOT, "gpurun_out", tag)
for wl in ("k1", "k2", "k3", "k4"):
    for name, dst in ((f"kernel_stats_{wl}.csv", f"{rnd}_kernel_stats_{wl}.csv"), (f"bench_{wl}.json", f"{rnd}_bench_{wl}.json")):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(ROOT, "profiles", dst))
    t = os.path.join(ROOT, "gpurun_out", "pmc", f"traffic_{wl}.json")
    b = os.path.join(src, f"bench_{wl}.json")
    if os.path.exists(t) and os.path.exists(b):
        d, bench = json.load(open(t)), json.load(open(b))
        d["command"] = (f"tools/pmc_traffic.sh {wl}: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv "
                        f"-- python3 bench.py --workload {wl} --no-graphs --steps 20 --warmup 5 --no-cpu-baseline")
        d["algorithmic_bytes_per_launch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
        d["note"] = ("in situ (inside the eager update step), averaged over the step's encoder launches (the merged s|s' launch and the actor's "
                     "s launch); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a 16 B/lane streaming read), WRITE_SIZE as reported")
        json.dump(d, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{wl}.json"), "w"), indent=1)
        print(wl, "traffic / algorithmic =", round(d["hbm_bytes_per_launch"] / d["algorithmic_bytes_per_launch"], 3), "sha", d["kernel_source_sha"])
