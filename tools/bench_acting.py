"""Latency of the acting path (agent(obs, mode=...), reference module_utils.py:147-159): the fused path of methods/acting.py against
the module tree (PointNet kernel + eager ATen heads), observation already on the device.  python tools/bench_acting.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloud_rl_amd import configs
from pointcloud_rl_amd.methods import build_agent
from pointcloud_rl_amd.synthetic import make_obs_np

dev = torch.device("cuda", 0)
CASES = [("K0 dmc_walker layout: N=1536, C=9 (xyz+rgb+frame one-hot), A=6", dict(N=1536, pos_encoding=3), 9, 6, 0, "sac_dmc"),
         ("K1 shape: N=1024, C=6, A=6", dict(N=1024), 6, 6, 0, "sac_dmc"),
         ("ManiSkill shape: N=1200, C=7, S=68, A=22", dict(N=1200, seg=1, agent=68), 7, 22, 68, "sac_maniskill")]
print(f"{'case':62s} {'B':>3s} {'mode':>8s} {'graph us':>9s} {'fused us':>9s} {'modules us':>10s}")
for name, kw, C, A, S, cfgname in CASES:
    kw = dict(kw)
    N = kw.pop("N")
    cfg = configs.sac_dmc(C, A, 256) if cfgname == "sac_dmc" else configs.sac_maniskill(C, A, S, 256)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(dev).eval()
    for B in (1, 4, 16):
        g = np.random.RandomState(B)
        obs = {k: torch.from_numpy(v).to(dev) for k, v in make_obs_np(g, B, N, **kw).items()}
        for mode in ("eval", "explore"):
            res = []
            for fused, graphs in ((True, True), (True, False), (False, False)):
                agent.use_fused_acting = fused
                agent.__dict__.pop("_fused_actor", None)
                agent._use_graphs = graphs
                for _ in range(20):
                    agent(obs, mode=mode)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 300
                for _ in range(n):
                    a = agent(obs, mode=mode)
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / n * 1e6)
            print(f"{name:62s} {B:3d} {mode:>8s} {res[0]:9.1f} {res[1]:9.1f} {res[2]:10.1f}")
