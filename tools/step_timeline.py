"""Print the kernel sequence of one steady-state update step from a rocprofv3 rocpd database.
    python tools/step_timeline.py <dir-with-.db> [marker-kernel-substring]"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name,start,end from kernels order by start"))
# steps are delimited by the critic Adam kernel following encoder_bwd_reduce; take the span between two packs far into the run
idx = [i for i, r in enumerate(rows) if (sys.argv[2] if len(sys.argv) > 2 else "encoder_bwd_reduce") in r[0]]
lo, hi = idx[len(idx) // 2], idx[len(idx) // 2 + 2]          # two consecutive steps (one with, one without the actor update)
prev = rows[lo][2]
tot = 0
for n, s, e in rows[lo + 1:hi + 1]:
    print(f"{n[:70]:70s} dur={(e - s) / 1e3:7.1f} gap={(s - prev) / 1e3:6.1f}")
    prev = e
print("two steps: %.1f us" % ((rows[hi][2] - rows[lo][2]) / 1e3))
