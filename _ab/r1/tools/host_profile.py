"""Host-side cost of update_parameters per step (cProfile over 500 graph-replayed K1 steps).  python tools/host_profile.py"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloud_rl_amd import configs
from pointcloud_rl_amd.methods import build_agent
from pointcloud_rl_amd.replay import DeviceReplay
from pointcloud_rl_amd.synthetic import make_batch_np
dev = torch.device("cuda:0")
cfg = configs.sac_dmc(6, 6, 256)
cfg["env_params"] = configs.env_params({"xyz": [3, 1024], "rgb": [3, 1024]}, 6)
torch.manual_seed(0)
agent = build_agent(cfg).to(dev)
mem = DeviceReplay(2048, device=dev, seed=1)
for lo in range(0, 2048, 512):
    mem.push_batch(make_batch_np(512, 1024, 6, seed=lo))
agent.train(); agent.enable_graphs()
for u in range(1, 31):
    agent.update_parameters(mem, u)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for u in range(31, 531):
    agent.update_parameters(mem, u)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:5000])
