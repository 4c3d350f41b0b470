"""Summarise a rocprofv3 results .db (rocpd sqlite) into CSV: per-kernel calls / total / average / share.

    python tools/rocpd_summary.py gpurun_out/prof/k1_results.db > profiles/r01_kernel_trace_k1.csv
"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in rows:
    print(f"\"{name[:110]}\",{calls},{total:.1f},{avg:.2f},{pct:.2f}")
print(f"\"TOTAL\",{sum(r[1] for r in rows)},{sum(r[2] for r in rows):.1f},,100.0")
