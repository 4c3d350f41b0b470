"""world_size=2 `gloo` test (CPU) of the data-parallel gradient exchange used by the agents:
flat-buffer SUM all-reduce with the 1/world factor handed to the optimizer, identical replicas
after broadcast, and batch sharding."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointcloud_rl_amd.methods.sac import FlatBuffer
    from pointcloud_rl_amd.utils.dist import allreduce_sum_, broadcast_parameters_, shard_slice, world_size
    torch.manual_seed(rank)                                   # replicas start different ...
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    broadcast_parameters_(net)                                # ... and are made identical (DDP's constructor broadcast)
    fb = FlatBuffer(list(net.named_parameters()))
    x = torch.arange(40.0).reshape(8, 5)[shard_slice(8, rank, world)]      # each rank: its shard of the global batch
    fb.zero_grad()
    (net(x).pow(2).sum() / 8).backward()                      # loss = mean over the GLOBAL batch of per-sample terms
    scale = allreduce_sum_(fb.grad)
    assert world_size() == world and scale == 1.0 / world
    # `no exchange` switch (agent.to_normal()): nothing happens, scale 1
    g = fb.grad.clone()
    assert allreduce_sum_(g, enabled=False) == 1.0 and torch.equal(g, fb.grad)
    out[rank] = (fb.data.clone(), fb.grad.clone())
    dist.destroy_process_group()


def test_flat_gradient_allreduce_matches_large_batch():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (w0, g0), (w1, g1) = out[0], out[1]
    assert torch.equal(w0, w1) and torch.equal(g0, g1)        # identical replicas, identical summed gradients
    # single-process reference on the whole batch with rank 0's weights
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    x = torch.arange(40.0).reshape(8, 5)
    (net(x).pow(2).sum() / 8).backward()
    ref = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4)) for p in net.parameters()])
    assert torch.allclose(g0, ref, rtol=1e-5, atol=1e-6)      # sum of shard gradients == large-batch gradient


def test_shard_slices_partition_the_batch():
    from pointcloud_rl_amd.utils.dist import shard_slice
    idx = list(range(256))
    parts = [idx[shard_slice(256, r, 8)] for r in range(8)]
    assert sum(parts, []) == idx and all(len(p) == 32 for p in parts)


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    cfg = configs.sac_dmc(6, 4, 8, head_hidden=32)
    cfg["env_params"] = configs.env_params({"xyz": [3, 16], "rgb": [3, 16]}, 4)
    torch.manual_seed(100 + rank)             # the reference's driver seeds torch with seed + rank (run_rl.py:263)
    agent = build_agent(cfg)
    with torch.no_grad():
        agent.log_alpha.fill_(float(rank))    # differs across ranks until to_ddp
    before = torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).clone()
    agent.to_ddp(device_ids=["cuda"])         # reference: DDP's constructor broadcasts rank 0's weights (module_utils.py:322-343)
    after = torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).clone()
    tgt = torch.cat([p.detach().reshape(-1) for p in agent.target_critic.parameters()]).clone()
    out[rank] = (before, after, tgt, agent.is_data_parallel())
    agent.to_normal()
    assert not agent.is_data_parallel()
    agent.recover_ddp()
    assert agent.is_data_parallel()
    dist.destroy_process_group()


def test_to_ddp_broadcasts_rank0_weights_like_ddp_does():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(world, port, out), nprocs=world, join=True)
    (b0, a0, t0, dp0), (b1, a1, t1, dp1) = out[0], out[1]
    assert dp0 and dp1
    assert not torch.equal(b0, b1)                            # different seeds: different replicas before
    assert torch.equal(a0, b0)                                # rank 0 keeps its weights
    assert torch.equal(a1, a0) and torch.equal(t1, t0)        # every parameter (incl. target critic, log_alpha) now rank 0's
