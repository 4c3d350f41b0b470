"""profiles/<round>_gemm_matrix_busy.json from a step ledger (tools/step_ledger.sh): the time-weighted matrix-pipe busy fraction over the head
GEMM launches (`gemm_fam_kernel<...>` rows), tagged with the hash of the GEMM sources so that bench.py reports it only for the kernels it
was measured on.    python tools/gemm_busy_json.py gpurun_out/ledger/ledger_k1.md profiles/r06_gemm_matrix_busy.json"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

rows = []
for l in open(sys.argv[1]):
    if l.startswith("| `") and "gemm_f" in l:
        c = [x.strip() for x in l.strip().strip("|").split("|")]
        rows.append(dict(kernel=c[0].strip("`"), launches_per_step=float(c[1]), mean_us=float(c[2]), busy=float(c[4].rstrip(" %")) / 100.0))
t = sum(r["launches_per_step"] * r["mean_us"] for r in rows)
out = {"kernel_source_sha": bench.gemm_source_sha(), "source": sys.argv[1], "us_per_step": t,
       "matrix_busy": sum(r["launches_per_step"] * r["mean_us"] * r["busy"] for r in rows) / t if t else None, "rows": rows,
       "note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) per launch, weighted by the launches' time in an eager K1 step "
               "(counter pass: serialised dispatches)"}
open(sys.argv[2], "w").write(json.dumps(out, indent=1) + "\n")
print(json.dumps({k: v for k, v in out.items() if k != "rows"}))
