"""Host time between two replayed steps: from the moment the host has seen step n's metrics (SAC._await_flag returns) to
the moment it starts waiting for step n + 1's (everything `update_parameters` and its caller do in between, the graph launch
included), and the part of it that is the graph launch.  python tools/probes/host_gap.py [k1|k2|k3|k4] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from pointcloud_rl_amd.methods import sac as sac_mod

name = sys.argv[1] if len(sys.argv) > 1 else "k1"
wl = dict(bench.WORKLOADS[name])
if len(sys.argv) > 2:
    wl["B"] = int(sys.argv[2])
dev = torch.device("cuda:0")
agent, _ = bench.build_agent(wl, wl["B"], dev)
memory = bench.device_ring(wl, wl.get("capacity", 2048), 0, dev)
agent.train(); agent.enable_graphs()
u = 0
for _ in range(40):
    u += 1; agent.update_parameters(memory, u)
torch.cuda.synchronize()

marks = []
orig_await = sac_mod.SAC._await_flag
orig_replay = torch.cuda.CUDAGraph.replay


def await_flag(view, n):
    t0 = time.perf_counter()
    out = orig_await(view, n)
    marks.append(("spin", t0, time.perf_counter()))
    return out


def replay(self):
    t0 = time.perf_counter()
    orig_replay(self)
    marks.append(("launch", t0, time.perf_counter()))


sac_mod.SAC._await_flag = staticmethod(await_flag)
torch.cuda.CUDAGraph.replay = replay
n = 600
t0 = time.perf_counter()
for _ in range(n):
    u += 1; agent.update_parameters(memory, u)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / n * 1e6
spins = [m for m in marks if m[0] == "spin"]
launches = [m for m in marks if m[0] == "launch"]
assert len(spins) == n and len(launches) == n, (len(spins), len(launches))
med = lambda v: sorted(v)[len(v) // 2] * 1e6
between = [spins[i + 1][1] - spins[i][2] for i in range(n - 1)]
before = [launches[i + 1][1] - spins[i][2] for i in range(n - 1)]
launch = [m[2] - m[1] for m in launches]
after = [spins[i][1] - launches[i][2] for i in range(n)]
spin = [m[2] - m[1] for m in spins]
print(f"{name} B={wl['B']}: {total:.1f} us per step; host between 'metrics seen' and 'waiting again': {med(between):.1f} us "
      f"= {med(before):.1f} before the graph launch + {med(launch):.1f} in hipGraphLaunch + {med(after):.1f} after it; spinning {med(spin):.1f} us")
