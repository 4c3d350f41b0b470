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
# two consecutive steps (one with, one without the actor update): of all such spans the one of median length, so that a span which holds a host
# synchronisation (the end of bench.py's warm-up or timed region, an eager step before the graph exists) is never the one printed
spans = sorted((rows[idx[i + 2]][2] - rows[idx[i]][2], i) for i in range(len(idx) - 2))
i = spans[len(spans) // 2][1]
lo, hi = idx[i], idx[i + 2]
prev = rows[lo][2]
tot = 0
for n, s, e in rows[lo + 1:hi + 1]:
    print(f"{n[:70]:70s} dur={(e - s) / 1e3:7.1f} gap={(s - prev) / 1e3:6.1f}")
    prev = e
print("two steps: %.1f us (median of %d two-step spans; shortest %.1f, longest %.1f)" % ((rows[hi][2] - rows[lo][2]) / 1e3, len(spans), spans[0][0] / 1e3, spans[-1][0] / 1e3))
# the idle device in front of each step's first launch (the host's turn-around between two graph replays), over every step of the trace
first = [i for i, r in enumerate(rows) if i and "replay_gather_kernel" in r[0]]
gaps = sorted((rows[i][1] - rows[i - 1][2]) / 1e3 for i in first)
gaps = [g for g in gaps if g < 200.0]                       # not the synchronisations of the warm-up / timed-region boundaries or eager steps' host work
if gaps:
    q = lambda f: gaps[min(len(gaps) - 1, int(f * len(gaps)))]
    print("idle in front of a step's first launch over %d steps (under the profiler): min %.1f, p10 %.1f, median %.1f, p90 %.1f us" % (len(gaps), gaps[0], q(0.1), q(0.5), q(0.9)))
