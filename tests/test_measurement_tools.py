"""The folding tools behind profiles/ (tools/step_timeline.py, tools/step_ledger.py, tools/gemm_busy_json.py) on synthetic traces: the numbers
bench.py's `roofline_gemm.matrix_busy` and DESIGN.md's gap figures come from must not depend on a kernel's name surviving a refactor (round 6:
the metrics' gather became a rider of the optimizer pass and the ledger counted one step instead of twenty-five)."""
import csv
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STEP = [("pcrl::replay_gather_kernel(pcrl::GatherParams, int, pcrl::PackJob)", 8.0),
        ("void pcrl::encoder_fwd_kernel<3, 64, 128, 256, false, false>(pcrl::FwdParams)", 60.0),
        ("void pcrl::gemm_fam_kernel<4u, 4>(pcrl::GemmGroup)", 6.0),
        ("void pcrl::gemm_fam_kernel<24u, 4>(pcrl::GemmGroup)", 12.0),
        ("void pcrl::encoder_bwdg_reduce_kernel<0>(float const*, int, int, int)", 9.0),
        ("pcrl::adam_gather_kernel(pcrl::AdamParams, pcrl::ScalarListParams)", 10.0)]


def run(tool, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *map(str, args)], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_timeline_prints_the_median_span_and_the_gap_distribution(tmp_path):
    db = sqlite3.connect(str(tmp_path / "trace.db"))
    db.execute("create table kernels(name, start, end)")
    t = 0
    for step in range(21):
        t += 5_000_000 if step == 10 else 20_000 if step % 2 else 10_000          # a synchronisation in the middle of the trace; two host gaps
        for name, us in STEP:
            db.execute("insert into kernels values (?,?,?)", (name, t, t + int(us * 1e3)))
            t += int(us * 1e3)
    db.commit()
    db.close()
    out = run("step_timeline.py", tmp_path, "bwdg_reduce")
    lines = out.strip().splitlines()
    assert lines[-2].startswith("two steps: 240.0 us")                              # 2 x 105 us of kernels + a 10 and a 20 us gap: not the 5 s span
    assert "longest 5" in lines[-2]
    assert "over 19 steps" in lines[-1] and "min 10.0" in lines[-1] and "median 20.0" in lines[-1]
    assert sum("gap=  10.0" in l or "gap=  20.0" in l for l in lines) == 2


def test_ledger_counts_steps_by_the_sampling_launch(tmp_path):
    def counter_pass(d, counters):
        os.makedirs(d)
        with open(os.path.join(d, "x_counter_collection.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            w.writeheader()
            disp, t = 0, 0
            for _ in range(5):
                for name, us in STEP:
                    disp += 1
                    for c, v in counters(name, us).items():
                        w.writerow(dict(Dispatch_Id=disp, Kernel_Name=name, Counter_Name=c, Counter_Value=v, Start_Timestamp=t, End_Timestamp=t + int(us * 1e3)))
                    t += int(us * 1e3)
    grbm = lambda us: us * 2100.0 * 8                                               # GRBM_GUI_ACTIVE summed over the 8 XCDs
    counter_pass(str(tmp_path / "p1"), lambda n, us: dict(SQ_VALU_MFMA_BUSY_CYCLES=(0.5 if "24u" in n else 0.25 if "gemm" in n else 0.0) * 1024 * grbm(us) / 8,
                                                           GRBM_GUI_ACTIVE=grbm(us)))
    counter_pass(str(tmp_path / "p2"), lambda n, us: dict(FETCH_SIZE=1000.0))
    counter_pass(str(tmp_path / "p3"), lambda n, us: dict(WRITE_SIZE=500.0))
    md = run("step_ledger.py", tmp_path / "p1", tmp_path / "p2", tmp_path / "p3", "synthetic")
    assert "5 eager steps" in md and "kernel time per step: 105 us" in md
    rows = {l.split("|")[1].strip(" `"): [c.strip() for c in l.split("|")] for l in md.splitlines() if l.startswith("| `")}
    assert rows["gemm_fam_kernel<24u, 4>"][2] == "1.00" and rows["gemm_fam_kernel<24u, 4>"][5] == "50.0 %"
    assert rows["adam_gather_kernel"][2] == "1.00"
    ledger = tmp_path / "ledger.md"
    ledger.write_text(md)
    run("gemm_busy_json.py", ledger, tmp_path / "busy.json")
    busy = json.loads((tmp_path / "busy.json").read_text())
    assert abs(busy["us_per_step"] - 18.0) < 1e-6                                   # the step's GEMM launches: 6 + 12 us
    assert abs(busy["matrix_busy"] - (6 * 0.25 + 12 * 0.5) / 18) < 1e-6             # weighted by the launches' time
    import bench
    assert busy["kernel_source_sha"] == bench.gemm_source_sha()


def test_the_committed_matrix_busy_figure_is_per_step_and_of_the_shipped_sources():
    import bench
    busy = json.load(open(os.path.join(ROOT, "profiles", "r06_gemm_matrix_busy.json")))
    assert busy["kernel_source_sha"] == bench.gemm_source_sha()
    assert 50.0 < busy["us_per_step"] < 400.0                                       # nine launches of a K1 step, not a whole trace
    assert all(r["launches_per_step"] <= 4.0 for r in busy["rows"])


def test_no_kernel_source_changed_after_the_last_whole_gpu_suite():
    """The final-binary rule: the library sources in the tree are the ones the last whole `-m gpu` suite ran on.  Whoever
    changes csrc/ or include/pcrl.h re-runs tools/gpu_round.sh (rebuild, whole suite, sha256) and updates profiles/*_final_binary.json."""
    import glob
    import bench
    rec = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_binary.json")))[-1]))
    assert rec["library_source_sha256"] == bench.library_source_sha(), \
        "csrc/ or include/pcrl.h changed after the last whole GPU suite: run tools/gpu_round.sh on an MI355X and record the new identity"
    assert rec["library_sha256"][:8] in open(os.path.join(ROOT, rec["suite_log"])).read()
    assert " passed" in rec["suite"] and "failed" not in rec["suite"]
