"""Where the host's time goes in one graph-replayed `update_parameters` call (SAC._replay_fast), per phase, at a rank's share of K1:
    entry -> hipGraphLaunch (the checks, the sample count, the sentinel fill) | the launch call itself | the spin for the metrics | the dictionary
and what is left between the metrics' arrival and the NEXT call's launch returning -- the host's turn-around the device waits through
(minus the optimizer pass that runs under it).      python tools/host_phases.py [batch=32] [steps=3000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloud_rl_amd import configs
from pointcloud_rl_amd.methods import build_agent
from pointcloud_rl_amd.methods.sac import SAC
from pointcloud_rl_amd.replay import DeviceReplay
from pointcloud_rl_amd.synthetic import make_batch_np

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dev = torch.device("cuda:0")
cfg = configs.sac_dmc(6, 6, B)
cfg["env_params"] = configs.env_params({"xyz": [3, 1024], "rgb": [3, 1024]}, 6)
torch.manual_seed(0)
agent = build_agent(cfg).to(dev)
mem = DeviceReplay(2048, device=dev, seed=1)
for lo in range(0, 2048, 512):
    mem.push_batch(make_batch_np(512, 1024, 6, seed=lo))
agent.train(); agent.enable_graphs()
for u in range(1, 41):
    agent.update_parameters(mem, u)
torch.cuda.synchronize()
now = time.perf_counter_ns
marks = {}


class Timed:
    def __init__(self, g):
        self.g = g

    def replay(self):
        marks["launch0"] = now()
        self.g.replay()
        marks["launch1"] = now()


fast = agent.__dict__["_fast"]
for key, (segments, names, view, n, sampler) in list(fast["entries"].items()):
    fast["entries"][key] = ([(Timed(g), meta) for g, meta in segments], names, view, n, sampler)
spin = SAC._await_flag


def timed_spin(view, n):
    out = spin(view, n)
    marks["seen"] = now()
    return out


agent._await_flag = timed_spin
rows = []
prev_seen = None
u = 40
for _ in range(STEPS):
    u += 1
    t0 = now()
    agent.update_parameters(mem, u)
    t1 = now()
    if "launch0" in marks and "seen" in marks:
        rows.append((marks["launch0"] - t0, marks["launch1"] - marks["launch0"], marks["seen"] - marks["launch1"], t1 - marks["seen"], t1 - t0,
                     (marks["launch1"] - prev_seen) if prev_seen else 0))
        prev_seen = marks["seen"]
    marks.clear()
a = np.array(rows[10:], dtype=np.float64) / 1e3
med = np.median(a, axis=0)
print(f"K1 share of {B} clouds, {len(a)} graph-replayed calls (medians, us; the timers themselves cost ~0.1 us each):")
for name, v in zip(("entry -> launch call", "hipGraphLaunch (CUDAGraph.replay)", "spin for the metrics", "metrics seen -> return (dictionary)", "whole call",
                    "metrics seen -> next launch call returned (turn-around)"), med):
    print(f"  {name:58s} {v:8.2f}")
print(f"  mean whole call {a[:, 4].mean():.2f} us  ->  {1e6 / a[:, 4].mean():.0f} calls/s (loop overhead excluded)")
