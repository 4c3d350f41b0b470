"""Headline benchmark: SAC gradient steps/sec (encoder + update) on B=256, N=1024 points.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one `agent.update_parameters(memory, updates)` (reference pyrl/methods/mfrl/sac.py:103),
on BASELINE.json config 2 ("K1"): synthetic replay B=256, N=1024, C=6 (xyz f32 + rgb u8, planar),
A=6, nets of configs/mfrl/sac/dm_control/pn.py, fp32.  Inputs are resident in HBM before the timed
region.  With N > 1 the batch of 256 is sharded over the ranks (256/N clouds each), parameters are
replicated and the flat gradient buffers are all-reduced over RCCL after each backward: strong
scaling, value = global gradient steps per second.

Defaults: 500 warm-up + 2000 timed steps (about 2.5 s on one GPU): the rate keeps climbing for the first ~1000 steps after
start-up (986 steps/s timed right after 20 warm-up steps, 1016-1037 once warm), and a training run is 10^5-10^6 updates.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the fused encoder forward),
timed with HIP events inside the timed region; `cpu_baseline` is the op-for-op PyTorch-CPU
restatement of the reference (oracle/torch_ref.py) timed on this box's host cores, rank 0, N=1 only.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (B, N, pcd channels extras, action_dim, agent_dim, config builder)
    "k1": dict(B=256, N=1024, A=6, S=0, obs_kw={}, cfg="sac_dmc", desc="SAC PointNet, synthetic replay B=256 N=1024 C=6 (BASELINE config 2)"),
    "k2": dict(B=256, N=1200, A=22, S=68, obs_kw=dict(seg=1), cfg="drq_maniskill_bf16",
               desc="DrQ pn_jitter, PointNet [128,128,256], B=256 x 2 augmentations, N=1200 C=7, jitter fused into the encoder load, "
                    "bf16 conv1/conv2 with fp32 accumulate (BASELINE config 3)"),
    "k3": dict(B=1024, N=1200, A=22, S=68, obs_kw=dict(seg=1), cfg="sac_maniskill",
               desc="SAC PointNet, ManiSkill shape B=1024 N=1200 C=7 (BASELINE config 4)"),
    "k4": dict(B=512, N=8192, A=6, S=0, obs_kw={}, cfg="sac_dmc", capacity=1024,
               desc="SAC PointNet, large-N stress B=512 N=8192 C=6, clouds split over workgroups + two-stage pool (BASELINE config 5)"),
}


def kernel_source_sha():
    """Hash of the sources the dominant kernel (encoder_fwd_kernel) is compiled from; tags profiles/*_pmc_traffic_*.json."""
    h = hashlib.sha256()
    for name in ("encoder_fwd.hip", "encoder_common.h", "common.h"):
        h.update(open(os.path.join(ROOT, "pointcloud_rl_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", default="k1", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-experimental", action="store_true", help="skip the extra (non-headline) run with the split-precision encoder forward")
    ap.add_argument("--no-graphs", action="store_true", help="run the step eagerly instead of replaying hipGraphs")
    ap.add_argument("--device-warmup-seconds", type=float, default=0.5, help="encoder-forward launches (no parameter update, the training "
                    "state stays where the W warm-up steps left it) right before the timed region")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the cpu_baseline leg (0: every CPU the box grants this process; the "
                    "reference ships torch.set_num_threads(1), pyrl/utils/meta/__init__.py:38-49)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for "
                    "exercising the multi-process path with several ranks on one GPU)")
    ap.add_argument("--replay", default="device", choices=["device", "fixed", "host"], help="device: sample every step from a device-resident "
                    "ring of synthetic transitions; fixed: the same resident batch every step; host: the batch comes from pinned host "
                    "memory every step (PCIe-inclusive rate, reported in DESIGN.md, never the headline value)")
    ap.add_argument("--replay-capacity", type=int, default=0, help="transitions in the device ring (0: 2048, or the workload's own default)")
    ap.add_argument("--batch", type=int, default=0, help="analysis only: override the global batch size (the JSON line then is NOT the "
                    "BASELINE metric; used to look at the per-GPU share of a multi-GPU run on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--single-rank-exchange", action="store_true", help="debug: with one rank, still create the process group and run the "
                    "data-parallel schedule (segmented graphs + async all-reduces over a one-rank group): exercises RCCL on a one-GPU box; "
                    "the line is tagged and is not the headline")
    ap.add_argument("--encoder-dtype", default=None, choices=["f32", "bf16", "f32split"], help="override the workload's encoder arithmetic; "
                    "f32split = the EXPERIMENTAL three-term bf16 split of the fp32 contractions (~1e-6 of fp32, not bit-comparable): the JSON "
                    "line then says dtype f32split and is not the headline configuration")
    return ap.parse_args()


def build_agent(wl, batch_per_rank, device, encoder_dtype=None):
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent as _build
    C = 6 + wl["obs_kw"].get("seg", 0) + wl["obs_kw"].get("pos_encoding", 0)
    if wl["cfg"] == "sac_dmc":
        cfg = configs.sac_dmc(C, wl["A"], batch_per_rank)
    elif wl["cfg"] == "drq_maniskill_bf16":
        cfg = configs.drq_maniskill(C, wl["A"], wl["S"], batch_per_rank, encoder_dtype="bf16")
    else:
        cfg = configs.sac_maniskill(C, wl["A"], wl["S"], batch_per_rank)
    obs_shape = {"xyz": [3, wl["N"]], "rgb": [3, wl["N"]]}
    cfg["env_params"] = configs.env_params(obs_shape, wl["A"])
    if encoder_dtype is not None:
        cfg["actor_cfg"]["nn_cfg"]["visual_nn_cfg"]["compute_dtype"] = encoder_dtype
    torch.manual_seed(0)                       # random-init weights of the named architecture
    return _build(cfg).to(device), C


def usable_cpus():
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU quota (a GPU box shows 256 logical
    CPUs but grants e.g. cpu.max = 1600000/100000 = 16; torch's default of 128 threads on 16 CPUs only adds contention)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(agent, wl, steps, threads=0, sample_batch=0):
    """The reference's update step restated op for op (six encoder passes, permute LayerNorm, per-tensor
    Adam groups), timed on the host: 1 warm-up + `steps` timed steps.  sample_batch > 0 bounds the work: the step is
    timed on the first `sample_batch` transitions of the batch and the rate is scaled by sample_batch / B (> 99 % of
    the reference's step is per-cloud encoder work, BASELINE.md section 2, so the step time is linear in the batch)."""
    from oracle import torch_ref
    from pointcloud_rl_amd.synthetic import make_batch_np
    params = {n: p.detach().cpu().clone() for n, p in agent.named_parameters()}
    ref = torch_ref.RefAgent(params, kind="sac", gamma=agent.gamma, reward_scale=agent.reward_scale, alpha=0.1,
                             target_entropy=agent.target_entropy, actor_update_interval=agent.actor_update_interval,
                             target_update_interval=agent.target_update_interval,
                             update_coeff=agent.update_coeff["default"], mirror_redundancy=True)
    Bs = min(sample_batch, wl["B"]) if sample_batch else wl["B"]
    batch = make_batch_np(Bs, wl["N"], wl["A"], seed=1, agent=wl["S"], **wl["obs_kw"])
    tb = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v)) for k, v in batch.items()}
    torch.set_num_threads(threads if threads else usable_cpus())
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    eps = lambda: [torch.randn(Bs, wl["A"], generator=g), torch.randn(Bs, wl["A"], generator=g)]
    ref.update_parameters(tb, 1, eps())
    t0 = time.perf_counter()
    for u in range(2, 2 + steps):
        ref.update_parameters(tb, u, eps())
    dt = (time.perf_counter() - t0) / steps * (wl["B"] / Bs)
    what = f"{steps} full update steps (B={wl['B']}, N={wl['N']})" if Bs == wl["B"] else \
        f"{steps} update steps on a {Bs}-transition slice of the B={wl['B']}, N={wl['N']} batch, time scaled by {wl['B']}/{Bs}"
    return {"value": 1.0 / dt, "unit": "gradient steps/s", "cores": cores, "kind": "port",
            "sample": f"{what} after 1 warm-up, torch {torch.__version__} CPU, {cores} threads "
                      f"(box grants {usable_cpus()} of {os.cpu_count()} logical CPUs)"}


def main():
    args = parse()
    wl = dict(WORKLOADS[args.workload])
    if not args.replay_capacity:
        args.replay_capacity = wl.get("capacity", 2048)
    if args.batch:
        wl["B"], wl["desc"] = args.batch, wl["desc"] + f" [batch overridden to {args.batch}]"
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if args.gpus > 1 and world == 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist_on = world > 1
    if args.single_rank_exchange and world == 1:
        os.environ["PCRL_EXCHANGE_SINGLE_RANK"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist_on = True
    if dist_on:
        if rank != 0:       # only rank 0 reports; RCCL's banners (C stdio, flushed at exit) of the other ranks must not follow its line
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        torch.distributed.init_process_group(args.backend, rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
    assert wl["B"] % world == 0, "batch must divide over the ranks"
    b_rank = wl["B"] // world

    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    agent, C = build_agent(wl, b_rank, device, args.encoder_dtype)
    if dist_on:
        agent.to_ddp(device_ids=["cuda"])                 # broadcasts rank 0's weights (as DDP's constructor does) and turns the exchange on
    if args.replay == "device":
        # device-resident replay (pointcloud_rl_amd/replay.py): every rank owns a ring of synthetic transitions and each
        # step samples its share of the batch from it (uniform with replacement, as OneStepTransition does) -- sampling is
        # part of the timed step, the ring is resident in HBM before the timed region
        from pointcloud_rl_amd.replay import DeviceReplay
        from pointcloud_rl_amd.synthetic import make_batch_np
        memory = DeviceReplay(args.replay_capacity, device=device, seed=1 + rank)
        for lo in range(0, args.replay_capacity, 512):
            memory.push_batch(make_batch_np(min(512, args.replay_capacity - lo), wl["N"], wl["A"], seed=1 + 1000 * rank + lo, agent=wl["S"], **wl["obs_kw"]))
    elif args.replay == "host":
        # the reference's arrangement: the sampled batch sits in host memory and crosses PCIe inside update_parameters
        # (`memory.sample(...).to_torch(device=..., non_blocking=True)`, sac.py:104); pinned here, pageable in the reference
        full = SyntheticReplay(wl["B"], wl["N"], wl["A"], seed=1, agent=wl["S"], **wl["obs_kw"])
        from pointcloud_rl_amd.utils.dist import shard_slice
        sl = shard_slice(wl["B"], rank, world)
        pin = lambda a: torch.from_numpy(a[sl].copy()).pin_memory()
        memory = SyntheticReplay.__new__(SyntheticReplay)
        memory.batch_np = None
        memory.batch = {k: ({kk: pin(vv) for kk, vv in v.items()} if isinstance(v, dict) else pin(v)) for k, v in full.batch_np.items()}
    else:
        # one fixed batch: every rank generates the global batch with the same seed and keeps its shard resident in HBM
        full = SyntheticReplay(wl["B"], wl["N"], wl["A"], seed=1, agent=wl["S"], **wl["obs_kw"])
        from pointcloud_rl_amd.utils.dist import shard_slice
        sl = shard_slice(wl["B"], rank, world)
        shard = {k: ({kk: vv[sl] for kk, vv in v.items()} if isinstance(v, dict) else v[sl]) for k, v in full.batch_np.items()}
        memory = SyntheticReplay.__new__(SyntheticReplay)
        from pointcloud_rl_amd.utils.torch_utils import to_torch
        memory.batch_np, memory.batch = shard, to_torch(shard, device=device)
    agent.train()
    if not args.no_graphs:
        agent.enable_graphs()

    def sync():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    updates = 0
    for _ in range(args.warmup):
        updates += 1
        agent.update_parameters(memory, updates)
    if not args.no_graphs:
        # both step variants (with / without the actor + target update) must have been captured before the clock starts:
        # a capture inside the timed region would be timed as a step (extra untimed steps are harmless)
        for _ in range(12):
            if len(getattr(agent, "_graphs", {})) >= 2:
                break
            updates += 1
            agent.update_parameters(memory, updates)
    # Device warm-up without touching the training state: a fresh process comes out of graph capture (synchronisations, an idle
    # device at its lowest P-state a second ago) -- half a second of encoder forward launches on the current batch, no update.
    if args.device_warmup_seconds > 0 and args.replay == "device":
        vis = memory.sample(b_rank).to_torch(device=device)["obs"]
        vis = {k: v for k, v in vis.items() if k in ("xyz", "rgb", "seg", "pos_encoding")}
        t_w = time.perf_counter()
        with torch.no_grad():
            while time.perf_counter() - t_w < args.device_warmup_seconds:
                for _ in range(20):
                    agent.encoder.encode_raw(vis)
                torch.cuda.synchronize()
    graphed = bool(getattr(agent, "_graphs", None))
    n_graph_variants = len(getattr(agent, "_graphs", {}) or {})
    if not graphed:
        hip.TIMER = hip.KernelTimer()          # eager: HIP events around every C-ABI launch inside the timed region
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        updates += 1
        agent.update_parameters(memory, updates)
    sync()
    elapsed = time.perf_counter() - t0
    nocomm_ms = None
    if dist_on:
        # the same step with the gradient exchange switched off (every rank trains on its shard alone): what is left of
        # ms_per_step is compute, the difference is what the (overlapped) all-reduces still cost
        agent.to_normal()
        n2 = max(args.steps // 2, 2)
        for _ in range(12):
            updates += 1
            agent.update_parameters(memory, updates)
        sync()
        t1 = time.perf_counter()
        for _ in range(n2):
            updates += 1
            agent.update_parameters(memory, updates)
        sync()
        nocomm_ms = (time.perf_counter() - t1) / n2 * 1e3
        agent.recover_ddp()
    if graphed:
        # Launches replayed from a hipGraph cannot be bracketed by host-recorded events, so the per-kernel
        # durations come from an eager pass over the same batch and weights right after the timed region
        # (same kernels, same launch geometry; rocprofv3 --kernel-trace of this command sees both passes).
        agent.enable_graphs(False)
        hip.TIMER = hip.KernelTimer()
        timed_eager_steps = min(args.steps, 40) // 2 * 2 or 2          # an even count: actor steps are every second one
        for _ in range(timed_eager_steps):
            updates += 1
            agent.update_parameters(memory, updates)
        sync()
    timer, hip.TIMER = hip.TIMER, None
    if dist_on:
        t = torch.tensor([elapsed, nocomm_ms], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed, nocomm_ms = float(t[0].item()), float(t[1].item())

    if rank == 0:
        spans = timer.summary()
        n_fwd, ms_fwd = spans.get("encoder_fwd", (0, float("nan")))
        f_pt = 2.0 * (C * agent.encoder.mlp_spec[0] + agent.encoder.mlp_spec[0] * agent.encoder.mlp_spec[1] +
                      agent.encoder.mlp_spec[1] * agent.encoder.mlp_spec[2])
        # the launches of one step have different cloud counts only for DrQ; for SAC every launch encodes b_rank clouds
        is_bf16 = getattr(agent.encoder, "compute_dtype", "f32") == "bf16"
        num_aug = getattr(agent, "num_aug", 1)
        # clouds per encoder launch: a step encodes s' and s (b * num_aug clouds each; ONE launch when the replay stages them back
        # to back) and, every second step, s again for the actor (b clouds) -- divided by the launches the timer counted
        steps_timed = timed_eager_steps if graphed else args.steps
        clouds_per_launch = b_rank * (2 * num_aug + 0.5) * steps_timed / max(n_fwd, 1)
        flops_per_launch = f_pt * clouds_per_launch * wl["N"]
        achieved = flops_per_launch / (ms_fwd * 1e-3) / 1e12
        peak = 2500.0 if is_bf16 else 157.3        # dense MFMA peaks of MI355X_MICROARCH.md (bf16 / fp32)
        traffic, traffic_src = None, None
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{args.workload}.json")))
        if cands and not args.batch and not dist_on:
            # HBM bytes per encoder_fwd launch from the committed rocprofv3 PMC passes of this same command
            # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; tools/pmc_traffic.py) -- counters cannot be read in-process.
            # The file names the hash of the kernel's sources it was measured on: a figure from another kernel is not reported.
            tj = json.load(open(cands[-1]))
            if tj.get("kernel_source_sha") == kernel_source_sha():
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), os.path.relpath(cands[-1], ROOT)
            else:
                traffic_src = f"{os.path.relpath(cands[-1], ROOT)} is stale (measured on kernel sources {tj.get('kernel_source_sha')}, " \
                              f"now {kernel_source_sha()}): not reported"
        out = {
            "metric": "SAC gradient steps/sec (encoder+update) on B=256, N=1024 pts" if args.workload == "k1" else f"SAC gradient steps/sec ({args.workload})",
            "value": args.steps / elapsed, "unit": "gradient steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "f32split": "f32split"}.get(getattr(agent.encoder, "compute_dtype", "f32"), "f32"), "data": "synthetic",
            "config": {"workload": wl["desc"], "global_batch": wl["B"], "points": wl["N"], "channels": C, "action_dim": wl["A"],
                       "parallelism": f"dp{world}", "batch_per_gpu": b_rank,
                       "hip_graphs": graphed, "device_warmup_seconds": args.device_warmup_seconds, "replay": args.replay + (f" ring of {args.replay_capacity} transitions, B sampled per step" if args.replay == "device" else " batch")},
            "roofline": {"kernel": "encoder_fwd_kernel", "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(clouds_per_launch * (wl["N"] * (12 + 3 + (C - 6)) + 8 * agent.encoder.mlp_spec[2])), "launches": n_fwd, "avg_launch_ms": ms_fwd,
                         "timed_with": "HIP events on the launch stream" + (", eager pass after the graph-replayed timed region" if graphed else ", inside the timed region"),
                         "algorithmic_flops_per_launch": flops_per_launch},
            "kernels_ms": {k: {"launches": n, "avg_ms": ms} for k, (n, ms) in spans.items()},
        }
        if nocomm_ms is not None:
            out["ms_per_step_nocomm"] = nocomm_ms
            out["comm_ms_per_step"] = elapsed / args.steps * 1e3 - nocomm_ms
        if dist_on and world == 1:
            out["debug"] = f"single-rank exchange over backend {args.backend}: data-parallel schedule with a one-rank process group"
        if not dist_on and not args.no_cpu_baseline and wl["cfg"].startswith("sac"):
            # bounded samples (about 10-30 s of CPU work each): every granted core on a slice of the batch that takes a few
            # seconds per step, and the reference's shipped single-thread setting (pyrl/utils/meta/__init__.py:38-49) on a
            # smaller slice
            points = wl["B"] * wl["N"]
            all_cores_b = wl["B"] if points <= 300_000 else max(8, int(wl["B"] * 300_000 / points))
            out["cpu_baseline"] = cpu_baseline(agent, wl, args.cpu_steps, args.cpu_threads, sample_batch=all_cores_b)
            if not args.cpu_threads:
                one_b = max(4, int(wl["B"] * 24_000 / points))
                out["cpu_baseline_1thread"] = cpu_baseline(agent, wl, 2, 1, sample_batch=one_b)
        if (not dist_on and not args.batch and args.encoder_dtype is None and not args.no_experimental and out["dtype"] == "f32"
                and args.replay == "device"):
            # Not the headline: the same step with the EXPERIMENTAL split-precision encoder forward (three-term bf16 split of the
            # fp32 contractions, within ~3e-6 of the exact kernel, argmax exact on the reference fixtures)
            del agent
            torch.cuda.empty_cache()
            agent2, _ = build_agent(wl, b_rank, device, "f32split")
            agent2.train()
            if not args.no_graphs:
                agent2.enable_graphs()
            u2 = 0
            for _ in range(max(args.warmup // 2, 30)):
                u2 += 1
                agent2.update_parameters(memory, u2)
            n2 = max(args.steps // 2, 10)
            sync()
            t2 = time.perf_counter()
            for _ in range(n2):
                u2 += 1
                agent2.update_parameters(memory, u2)
            sync()
            dt2 = time.perf_counter() - t2
            out["experimental_f32split"] = {"value": n2 / dt2, "unit": "gradient steps/s", "ms_per_step": dt2 / n2 * 1e3, "steps": n2,
                                            "note": "encoder conv1/conv2 and the backward data-gradient GEMMs as three-term bf16 splits (pcrl_encoder_{fwd,bwd}_f32split); "
                                                    "opt-in, not the reported value"}
        line = json.dumps(out)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        # The JSON line must be the LAST thing on stdout: RCCL prints its version banner through C stdio, which a pipe buffers
        # until exit -- after a Python print.  Flush C stdio first, then print.
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
