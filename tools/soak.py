"""Soak run: 20 000 graph-replayed SAC steps (K1 shape), 8 000 DrQ steps (rotation + scale + jitter) and 20 000 SAC steps at a rank's 32-cloud
share (the encoder backward's team kernel) against a device replay whose ring keeps changing; every returned metric and every parameter must
stay finite.    python tools/soak.py"""
import sys, math, time, torch
sys.path.insert(0, '.')
import bench
from pointcloud_rl_amd import configs
from pointcloud_rl_amd.methods import build_agent
from pointcloud_rl_amd.replay import DeviceReplay
from pointcloud_rl_amd.synthetic import make_batch_np
dev = torch.device('cuda', 0)
for name, cfgf, kw, B, N, A, steps in [('k1', configs.sac_dmc, dict(pcd_channels=6, action_dim=6, batch_size=256), 256, 1024, 6, 20000),
                                        ('k2-like', configs.drq_dmc, dict(pcd_channels=6, action_dim=6, batch_size=128, obs_aug=[configs.ROT_SCALE, configs.JITTER]), 128, 1024, 6, 8000),
                                        ('k1 share', configs.sac_dmc, dict(pcd_channels=6, action_dim=6, batch_size=32), 32, 1024, 6, 20000)]:
    cfg = cfgf(**kw)
    cfg['env_params'] = configs.env_params({'xyz': [3, N], 'rgb': [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(dev)
    mem = DeviceReplay(1024, device=dev, seed=3)
    mem.push_batch(make_batch_np(1024, N, A, seed=5))
    agent.enable_graphs()
    t0 = time.time(); bad = 0
    for u in range(1, steps + 1):
        r = agent.update_parameters(mem, u)
        if u % 997 == 0:
            mem.push_batch(make_batch_np(64, N, A, seed=u))       # the ring keeps changing between replays
        if not all(math.isfinite(v) for v in r.values()):
            bad += 1
    torch.cuda.synchronize()
    print(name, 'steps', steps, 'non-finite returns', bad, 'last', {k: round(v, 4) for k, v in r.items()}, 'steps/s', round(steps / (time.time() - t0), 1), flush=True)
    assert bad == 0
    for n_, p_ in agent.named_parameters():
        assert torch.isfinite(p_).all(), n_
print('soak ok')
