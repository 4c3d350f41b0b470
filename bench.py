"""Headline benchmark: SAC gradient steps/sec (encoder + update) on B=256, N=1024 points.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: this process starts the N ranks itself, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one `agent.update_parameters(memory, updates)` (reference pyrl/methods/mfrl/sac.py:103),
on BASELINE.json config 2 ("K1"): synthetic replay B=256, N=1024, C=6 (xyz f32 + rgb u8, planar),
A=6, nets of configs/mfrl/sac/dm_control/pn.py, fp32.  Inputs are resident in HBM before the timed
region.  With N > 1 the batch of 256 is sharded over the ranks (256/N clouds each), parameters are
replicated and the flat gradient buffers are all-reduced over RCCL after each backward: strong
scaling, value = global gradient steps per second.

Defaults: 500 warm-up + 2000 timed steps (about 2.5 s on one GPU); a training run is 10^5-10^6 updates.  Every default K1 line also
carries, as extra top-level objects that are never the reported value, `config4_k3` (the same ranks on BASELINE config 4: B 1024,
N 1200, C 7 -- the shape north_star's strong-scaling target is stated for), `config3_k2` (config 3: DrQ, jitter + scale fused into the
bf16 encoder load; with its own cpu_baseline at N = 1) and `config5_k4` (config 5: B 512, N 8192), each with its own `roofline`
object, and, at N = 1, `experimental_f32split`.

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
    "k2": dict(B=256, N=1200, A=22, S=68, obs_kw=dict(seg=1), cfg="drq_maniskill_bf16", aug="rot_scale+jitter",
               desc="DrQ, PointNet [128,128,256], B=256 x 2 augmentations, N=1200 C=7, GlobalRotScaleTrans (rotation + per-axis scale) and "
                    "RandomJitterPoints fused into the encoder load, bf16 conv1/conv2 with fp32 accumulate (BASELINE config 3: jitter+scale)"),
    "k2j": dict(B=256, N=1200, A=22, S=68, obs_kw=dict(seg=1), cfg="drq_maniskill_bf16", aug="jitter",
                desc="DrQ pn_jitter (the reference's configs/mfrl/drq/maniskill/pn_jitter.py as shipped: jitter only), B=256 x 2 augmentations, "
                     "N=1200 C=7, bf16 conv1/conv2 with fp32 accumulate"),
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


def library_source_sha():
    """Hash of everything libpcrl_hip.so is compiled from (csrc/*.hip, csrc/*.h, csrc/Makefile, include/pcrl.h): the identity of the
    shipped library that survives a rebuild (hipcc's objects are not bit-reproducible).  profiles/*_final_binary.json records it next to the
    log of the last whole `-m gpu` suite; tests/test_measurement_tools.py holds the tree to it (no kernel change after the last suite)."""
    import glob
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "pointcloud_rl_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))) + [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "pcrl.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


def gemm_source_sha():
    """Hash of the sources the head GEMM kernels are compiled from; tags profiles/*_gemm_matrix_busy.json."""
    h = hashlib.sha256()
    for name in ("dense.hip", "dense_wtile.h", "common.h"):
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
    ap.add_argument("--no-extra-workloads", action="store_true", help="skip the short extra runs of BASELINE configs 4, 3 and 5 (K3: B=1024, N=1200, C=7, "
                    "the shape north_star's strong-scaling target is stated for; K2: DrQ bf16 jitter+scale; K4: B=512, N=8192) that every default "
                    "k1 line carries as `config4_k3`, `config3_k2`, `config5_k4`")
    ap.add_argument("--extra-steps", type=int, default=100)
    ap.add_argument("--extra-timeout", type=float, default=300.0, help="seconds the extra objects may take before the headline line is printed without them")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="seconds the launcher (plain `python bench.py --gpus N`) waits for its ranks")
    ap.add_argument("--set-fused", action="append", default=[], metavar="ATTR=VALUE", help="analysis only: set an attribute of the fused step "
                    "(methods/fused.py: e.g. publish_first=0) before the graphs are captured -- same-box A/B of a kept switch against its other arm")
    ap.add_argument("--start-lock", type=int, default=1, help="ranks SHARING one GPU (--dry-run-ranks / --share-gpu) bring their GPU context up one after "
                    "the other (a file lock around device initialisation, agent construction and the first synchronisation) -- next to "
                    "HSA_ENABLE_SDMA=0, which is what removes the multi-process fault of the rehearsal (profiles/r06_dry_run_loop.md); 0 = off. "
                    "Real multi-GPU runs (one device per rank) take neither")
    ap.add_argument("--dry-run-ranks", type=int, default=0, help="rehearsal of the multi-GPU run on a ONE-GPU box: the launcher starts this many ranks "
                    "exactly as `--gpus N` does (rendezvous on 127.0.0.1, to_ddp's broadcast, the sharded replay rings, the data-parallel step "
                    "schedule, the comm / no-comm timing, the extra workloads behind their watchdog, rank 0's line last on stdout), but the ranks "
                    "share cuda:0 for compute and exchange over gloo (host threads) instead of RCCL; the line is tagged `dry_run` and is not a "
                    "measurement")
    return ap.parse_args()


def build_agent(wl, batch_per_rank, device, encoder_dtype=None):
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent as _build
    C = 6 + wl["obs_kw"].get("seg", 0) + wl["obs_kw"].get("pos_encoding", 0)
    if wl["cfg"] == "sac_dmc":
        cfg = configs.sac_dmc(C, wl["A"], batch_per_rank)
    elif wl["cfg"] == "drq_maniskill_bf16":
        # the fused load applies the matrix first, then the jitter: the list is in that order (augmentations.py)
        aug = [configs.ROT_SCALE, configs.JITTER] if wl.get("aug") == "rot_scale+jitter" else configs.JITTER
        cfg = configs.drq_maniskill(C, wl["A"], wl["S"], batch_per_rank, obs_aug=aug, encoder_dtype="bf16")
    else:
        cfg = configs.sac_maniskill(C, wl["A"], wl["S"], batch_per_rank)
    obs_shape = {"xyz": [3, wl["N"]], "rgb": [3, wl["N"]]}
    cfg["env_params"] = configs.env_params(obs_shape, wl["A"])
    if encoder_dtype is not None:
        cfg["actor_cfg"]["nn_cfg"]["visual_nn_cfg"]["compute_dtype"] = encoder_dtype
    torch.manual_seed(0)                       # random-init weights of the named architecture
    return _build(cfg).to(device), C


def device_ring(wl, capacity, rank, device):
    """Device-resident replay (pointcloud_rl_amd/replay.py): every rank owns a ring of synthetic transitions and each step
    samples its share of the batch from it (uniform with replacement, as OneStepTransition does) -- sampling is part of the
    timed step, the ring is resident in HBM before the timed region."""
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    memory = DeviceReplay(capacity, device=device, seed=1 + rank)
    for lo in range(0, capacity, 512):
        memory.push_batch(make_batch_np(min(512, capacity - lo), wl["N"], wl["A"], seed=1 + 1000 * rank + lo, agent=wl["S"], **wl["obs_kw"]))
    return memory


def encoder_roofline(agent, wl, C, b_rank, n_fwd, ms_fwd, steps_timed, workload, graphed, use_traffic):
    """`roofline` object of the dominant kernel (the fused encoder forward) from its HIP-event spans: algorithmic FLOPs per average
    launch (F_pt = 2 (C c1 + c1 c2 + c2 c3) per point) / average launch duration, against the dense MFMA
    peak of the arithmetic it contracts in (MI355X_MICROARCH.md: fp32 157.3, bf16 2 500 TFLOP/s)."""
    spec = agent.encoder.mlp_spec
    f_pt = 2.0 * (C * spec[0] + spec[0] * spec[1] + spec[1] * spec[2])
    is_bf16 = getattr(agent.encoder, "compute_dtype", "f32") == "bf16"
    num_aug = getattr(agent, "num_aug", 1)
    # clouds per encoder launch: a step encodes s' and s (b * num_aug clouds each; ONE launch when the replay stages them back
    # to back) and, every second step, s again for the actor (b clouds) -- divided by the launches the timer counted
    clouds_per_launch = b_rank * (2 * num_aug + 0.5) * steps_timed / max(n_fwd, 1)
    flops_per_launch = f_pt * clouds_per_launch * wl["N"]
    achieved = flops_per_launch / (ms_fwd * 1e-3) / 1e12
    peak = 2500.0 if is_bf16 else 157.3
    traffic, traffic_src = None, None
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{workload}.json")))
    if cands and use_traffic:
        # HBM bytes per encoder_fwd launch from the committed rocprofv3 PMC passes of this same command
        # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; tools/pmc_traffic.py) -- counters cannot be read in-process.
        # The file names the hash of the kernel's sources it was measured on: a figure from another kernel is not reported.
        tj = json.load(open(cands[-1]))
        if tj.get("kernel_source_sha") == kernel_source_sha():
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), os.path.relpath(cands[-1], ROOT)
        else:
            traffic_src = f"{os.path.relpath(cands[-1], ROOT)} is stale (measured on kernel sources {tj.get('kernel_source_sha')}, " \
                          f"now {kernel_source_sha()}): not reported"
    alg_bytes = int(clouds_per_launch * (wl["N"] * (12 + 3 + (C - 6)) + 8 * spec[2]))
    out = {"kernel": "encoder_fwd_kernel", "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
           "frac": achieved / peak, "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
           "algorithmic_bytes_per_launch": alg_bytes, "launches": n_fwd, "avg_launch_ms": ms_fwd,
           "timed_with": "HIP events on the launch stream" + (", eager pass after the graph-replayed timed region" if graphed else ", inside the timed region"),
           "algorithmic_flops_per_launch": flops_per_launch}
    if is_bf16:
        # The bf16 kernel is NOT bound by the matrix pipe: measured (profiles/r04_bf16_fwd_ceiling.md, profiles/r03_fwd_stamps.md) a SIMD spends
        # 4.1 k of its 16.5 k cycles per 32-point tile in MFMAs and 12.4 k in ~1 900 vector / LDS instructions on the accumulators it holds
        # (LayerNorm-2 + the max-pool against the LDS keys: ~8.5 per conv2 output) -- with those two phases removed the same launch reaches
        # 0.275 of the bf16 peak, the best redesign by instruction count ~0.22.  `frac` stays achieved / dense bf16 MFMA peak for comparison.
        out.update(bound="valu", mfma_frac=achieved / peak,
                   valu={"vector_or_lds_instructions_per_conv2_output": 8.5, "matrix_cycles_per_tile": 4100, "vector_cycles_per_tile": 12400,
                         "frac_with_layernorm2_and_pool_removed": 0.275, "source": "profiles/r04_bf16_fwd_ceiling.md, profiles/r03_fwd_stamps.md"},
                   bound_note="vector-instruction bound (fp32 LayerNorm / ReLU / pool on 256 accumulators per point), not MFMA-bound: "
                              "the fraction of the dense bf16 MFMA peak is reported for comparison only")
    if traffic and traffic > 2 * alg_bytes:
        out["traffic_note"] = (f"counter traffic is {traffic / alg_bytes:.1f}x the algorithmic bytes: inside the step this launch's counters also see the "
                               "write-back of lines its predecessors (Adam, re-pack, replay gather) left dirty in L2 and the feature head's weight reads; "
                               "launched back to back the same kernel moves ~1.6x (DESIGN.md section 4.1).  At < 0.2 TB/s it is two orders of magnitude "
                               "under the HBM roofline either way")
    return out


def feature_dim(agent):
    fm = getattr(agent.encoder, "final_mlp", None)
    return fm[0].out_features if fm is not None else agent.encoder.mlp_spec[-1]


def active_points_per_cloud(agent, tiles=False):
    """Points per cloud the encoder backward of the last eager step actually visited: the distinct argmax positions of the LIVE channels
    (pooled value > 0: the backward's prep launch drops the others) of the gradient-carrying pass, read from the step's own tensors; None
    when the step kept none.  tiles=True: (mean points per cloud, total 32-point tiles = sum over the clouds of ceil(points / 32))."""
    fused = getattr(agent, "_fused", None)
    am, pooled = getattr(fused, "last_argmax", None), getattr(fused, "last_pooled", None)
    if am is None or am.numel() == 0:
        return (None, None) if tiles else None
    am = am.reshape(am.shape[0], -1).long()
    if pooled is not None and pooled.shape == am.shape:
        am = torch.where(pooled > 0, am, torch.full_like(am, 1 << 40))       # dead channels: one sentinel position, subtracted below
        srt = am.sort(dim=1).values
        n = (srt[:, 1:] != srt[:, :-1]).sum(1) + 1 - (srt[:, -1] == (1 << 40)).long()
    else:
        srt = am.sort(dim=1).values
        n = (srt[:, 1:] != srt[:, :-1]).sum(1) + 1
    mean = float(n.float().mean().item())
    return (mean, int(((n + 31) // 32).sum().item())) if tiles else mean


def step_roofline(wl, C, spec, b_global, num_aug, A, S, ms_per_step, is_bf16, head_hidden=1024, feat=None, active_pts=None, bwd_tiles=None):
    """Whole-step figure next to the dominant kernel's.  `frac` = EXECUTED FLOPs / step time / the dense MFMA peak of the encoder's
    arithmetic (<= 1 by construction); `useful_tflops` = SURVEY.md section 8(d)'s CANONICAL FLOPs (dense backward: 4.5 point-passes per
    point, heads (2.5 F_a + 11 F_q) per sample) / step time -- the rate of useful work, which exceeds the executed rate because the
    max-pool's gradient reaches only the argmax points of a cloud.
    Executed = 2.5 forward point-passes + heads + the Gram-form backward on the ACTIVE points, per active point:
    recompute conv0 / conv1 2 (C c1 + c1 c2), q = Mc h1 2 c2^2, dH0 = W1^T dz1 2 c1 c2, the 10 of 16 blocks of G 1.25 c2^2, dW1 2 c1 c2,
    dW0 2 C c1; per cloud the owned channels' rows 6 c2 c3 (two dot-product passes and the dW2 rows)."""
    c1, c2, c3 = spec
    f_pt = 2.0 * (C * c1 + c1 * c2 + c2 * c3)
    P = b_global * num_aug * wl["N"]
    enc = (4.0 * P + 0.5 * b_global * wl["N"]) * f_pt
    F = feat if feat is not None else 50
    d_a = F + S
    f_a = 2.0 * (d_a * head_hidden + head_hidden ** 2 + head_hidden * 2 * A)
    f_q = 2.0 * ((d_a + A) * head_hidden + head_hidden ** 2 + head_hidden)
    heads = (2.5 * f_a + 11 * f_q) * b_global * num_aug
    peak = 2500.0 if is_bf16 else 157.3
    per_s = 1.0 / (ms_per_step * 1e-3) / 1e12
    fwd = (2.0 * P + 0.5 * b_global * wl["N"]) * f_pt
    n_act = min(active_pts if active_pts is not None else float(c3), float(wl["N"]), float(c3))
    f_bwd_pt = 2.0 * (C * c1 + c1 * c2) + 2.0 * c2 * c2 + 2.0 * c1 * c2 + 1.25 * c2 * c2 + 2.0 * c1 * c2 + 2.0 * C * c1
    bwd = b_global * num_aug * (n_act * f_bwd_pt + 6.0 * c2 * c3)
    executed = fwd + heads + bwd
    return {"executed_gflop_per_step": executed / 1e9, "achieved": executed * per_s, "unit": "TFLOP/s", "peak": peak, "frac": executed * per_s / peak,
            "backward_tiles_per_rank": bwd_tiles, "backward_wave_slots": 4 * 256,
            "active_points_per_cloud": n_act, "active_points_source": "distinct argmax positions of the step's gradient-carrying pass" if active_pts is not None
            else "upper bound c3 (not measured on this run)",
            "encoder_fwd_gflop": fwd / 1e9, "encoder_bwd_gflop": bwd / 1e9, "heads_gflop": heads / 1e9,
            "canonical_gflop_per_step": (enc + heads) / 1e9, "useful_tflops": (enc + heads) * per_s, "useful_frac": (enc + heads) * per_s / peak,
            "note": "`frac` counts the FLOPs the kernels execute (forward passes, heads, the Gram-form backward on the active points: DESIGN.md 4.2); "
                    "`useful_tflops` / `useful_frac` price SURVEY.md 8(d)'s canonical dense-backward FLOPs at the same step time and may exceed the peak "
                    "(the max-pool's gradient reaches <= c3 points of a cloud): a rate of useful work, not pipe utilisation"}


def gemm_roofline(timer, steps_timed):
    """Second `roofline` object: the head GEMM launches (gemm_fam_kernel<...>, csrc/dense.hip) -- the kernel family furthest below its roofline
    with a real share of the step.  FLOPs = 2 M N K of every problem launched under the timer, time = the launches' HIP-event spans."""
    spans = timer.summary()
    names = [k for k in spans if k == "gemm" or k.startswith("gemm ")]
    n = sum(spans[k][0] for k in names)
    total_ms = sum(spans[k][0] * spans[k][1] for k in names)
    flops = timer.flops.get("gemm", 0.0)
    if not n or total_ms <= 0:
        return None
    achieved = flops / (total_ms * 1e-3) / 1e12
    busy, busy_src = None, None
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_matrix_busy.json")))
    if cands:
        bj = json.load(open(cands[-1]))
        if bj.get("kernel_source_sha") == gemm_source_sha():
            busy, busy_src = bj.get("matrix_busy"), os.path.relpath(cands[-1], ROOT)
        else:
            busy_src = f"{os.path.relpath(cands[-1], ROOT)} is stale (other kernel sources): not reported"
    return {"kernel": "gemm_fam_kernel<families, waves>", "bound": "mfma", "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3,
            "launches_per_step": n / max(steps_timed, 1), "us_per_step": total_ms * 1e3 / max(steps_timed, 1), "gflop_per_step": flops / 1e9 / max(steps_timed, 1),
            "matrix_busy": busy, "matrix_busy_source": busy_src, "timed_with": "HIP events on the launch stream, eager pass"}


def side_rate(name, rank, world, device, dist_on, steps, warmup, encoder_dtype=None, memory=None, graphs=True, roofline=False,
              cpu_steps=0, cpu_threads=0):
    """A short, separately reported run of another workload (or of another encoder arithmetic) with the same protocol as the
    headline -- device ring, hipGraph replay, barrier + synchronize on both sides, MAX over ranks -- for the extra objects of the
    JSON line.  Never the reported `value`."""
    wl = WORKLOADS[name]
    assert wl["B"] % world == 0
    b_rank = wl["B"] // world
    agent, C = build_agent(wl, b_rank, device, encoder_dtype)
    if dist_on:
        agent.to_ddp(device_ids=["cuda"])
    if memory is None:
        memory = device_ring(wl, wl.get("capacity", 2048), rank, device)
    agent.train()
    if graphs:
        agent.enable_graphs()

    def sync():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    u = 0
    for _ in range(warmup):
        u += 1
        agent.update_parameters(memory, u)
    for _ in range(12):
        if not graphs or len(getattr(agent, "_graphs", {})) >= 2:
            break
        u += 1
        agent.update_parameters(memory, u)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        u += 1
        agent.update_parameters(memory, u)
    sync()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    out = {"value": steps / dt, "unit": "gradient steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
           "global_batch": wl["B"], "batch_per_gpu": b_rank, "points": wl["N"], "n_gpus": world, "scaling": "strong",
           "exchange": exchange_mode(agent, dist_on), "dtype": {"bf16": "bf16", "f32split": "f32split"}.get(getattr(agent.encoder, "compute_dtype", "f32"), "f32")}
    if roofline:
        # per-kernel spans from an eager pass over the same ring and weights (a replayed graph's launches cannot be bracketed by
        # host-recorded events); every rank runs it (the step's collectives are inside), rank 0's spans are reported
        from pointcloud_rl_amd import hip
        if graphs:
            agent.enable_graphs(False)
        for _ in range(2):
            u += 1
            agent.update_parameters(memory, u)
        sync()
        hip.TIMER = hip.KernelTimer()
        eager_steps = 6
        for _ in range(eager_steps):
            u += 1
            agent.update_parameters(memory, u)
        sync()
        timer, hip.TIMER = hip.TIMER, None
        spans = timer.summary()
        n_fwd, ms_fwd = spans.get("encoder_fwd", (0, float("nan")))
        out["roofline"] = encoder_roofline(agent, wl, C, b_rank, n_fwd, ms_fwd, eager_steps, name, graphs, use_traffic=not dist_on)
        out["kernels_ms"] = {k: {"launches": n, "avg_ms": ms} for k, (n, ms) in spans.items()}
        g = gemm_roofline(timer, eager_steps)
        if g:
            out["roofline_gemm"] = g
    out["step"] = step_roofline(wl, C, agent.encoder.mlp_spec, wl["B"], getattr(agent, "num_aug", 1), wl["A"], wl["S"], out["ms_per_step"],
                                getattr(agent.encoder, "compute_dtype", "f32") == "bf16", feat=feature_dim(agent),
                                **(dict(zip(("active_pts", "bwd_tiles"), active_points_per_cloud(agent, tiles=True))) if roofline else {}))
    if cpu_steps and rank == 0:
        points = wl["B"] * wl["N"] * getattr(agent, "num_aug", 1)
        out["cpu_baseline"] = cpu_baseline(agent, wl, cpu_steps, cpu_threads, sample_batch=wl["B"] if points <= 300_000 else max(8, int(wl["B"] * 300_000 / points)),
                                           warmup_batch=4)
    del agent, memory
    torch.cuda.empty_cache()
    return out


def exchange_mode(agent, dist_on):
    """How the data-parallel step ran its all-reduces: nodes of the step's one hipGraph, or eager between per-segment graphs."""
    if not dist_on:
        return None
    graphs = getattr(agent, "_graphs", None) or {}
    if not graphs:
        return "eager step, asynchronous all-reduces"
    return "captured in the step's hipGraph" if all(len(segs) == 1 for segs, _, _ in graphs.values()) else "eager all-reduces between per-segment hipGraphs"


def exchange_probe(dist_on):
    """Verdict of the one-time probe that decides whether the all-reduces may live inside the step's hipGraph (utils/dist.py:
    capture_exchange): "works", "failed: <why>" (the run then used eager all-reduces between per-segment graphs), "not run" (a backend
    whose collectives cannot be captured, e.g. gloo; or PCRL_CAPTURE_EXCHANGE=0), "skipped"."""
    if not dist_on:
        return None
    from pointcloud_rl_amd.utils.dist import probe_verdict
    return probe_verdict()


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


def cpu_baseline(agent, wl, steps, threads=0, sample_batch=0, warmup_batch=0):
    """The reference's update step restated op for op (six encoder passes, permute LayerNorm, per-tensor
    Adam groups), timed on the host: 1 warm-up + `steps` timed steps.  sample_batch > 0 bounds the work: the step is
    timed on the first `sample_batch` transitions of the batch and the rate is scaled by sample_batch / B -- the result is then
    tagged "extrapolated" (> 99 % of the reference's step is per-cloud encoder work, BASELINE.md section 2, but a small slice
    runs at another cache / thread efficiency than the full batch).  warmup_batch > 0: the untimed warm-up step (lazy
    initialisation, allocator) runs on that many transitions instead of the whole sample."""
    from oracle import torch_ref
    from pointcloud_rl_amd.synthetic import make_batch_np
    params = {n: p.detach().cpu().clone() for n, p in agent.named_parameters()}
    kind = "drq" if wl["cfg"].startswith("drq") else "sac"
    num_aug = getattr(agent, "num_aug", 1) if kind == "drq" else 1
    ref = torch_ref.RefAgent(params, kind=kind, gamma=agent.gamma, reward_scale=agent.reward_scale, alpha=0.1,
                             target_entropy=agent.target_entropy, actor_update_interval=agent.actor_update_interval,
                             target_update_interval=agent.target_update_interval, num_aug=max(num_aug, 1),
                             update_coeff=agent.update_coeff["default"], mirror_redundancy=True)
    Bs = min(sample_batch, wl["B"]) if sample_batch else wl["B"]
    batch = make_batch_np(Bs, wl["N"], wl["A"], seed=1, agent=wl["S"], **wl["obs_kw"])
    as_t = lambda b: {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v)) for k, v in b.items()}
    tb = as_t(batch)
    torch.set_num_threads(threads if threads else usable_cpus())
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    eps = lambda n: [torch.randn(n * num_aug, wl["A"], generator=g), torch.randn(n, wl["A"], generator=g)]
    # DrQ: the augmentation draws are part of the reference's step (CPU generators, drq.py:52-60; pcd_aug.py:178-196, 316-322), so they
    # are made inside the timed loop: jitter noise for obs and next_obs, and the rotation / scale matrices when the workload has them
    aug = None
    if kind == "drq":
        from pointcloud_rl_amd.augmentations import GlobalRotScaleTrans
        from pointcloud_rl_amd import configs
        rst = GlobalRotScaleTrans(**{k: v for k, v in configs.ROT_SCALE.items() if k != "type"}) if wl.get("aug") == "rot_scale+jitter" else None

        def aug(n):
            rows = n * num_aug
            jit = [torch.empty(rows, 3, wl["N"]).uniform_(-0.01, 0.01, generator=g) for _ in range(2)]
            aff = [rst.sample_matrix(rows, "cpu") for _ in range(2)] if rst is not None else None
            return dict(jitter_list=jit, affine_list=aff)
    draws = (lambda n: aug(n)) if aug is not None else (lambda n: {})
    if warmup_batch and warmup_batch < Bs:
        cut = lambda v: v[:warmup_batch]
        wb = {k: ({kk: cut(vv) for kk, vv in v.items()} if isinstance(v, dict) else cut(v)) for k, v in tb.items()}
        ref.update_parameters(wb, 2, eps(warmup_batch), **draws(warmup_batch))           # an even count: the actor / target branch is warmed up too
    else:
        ref.update_parameters(tb, 1, eps(Bs), **draws(Bs))
    t0 = time.perf_counter()
    for u in range(2, 2 + steps):
        ref.update_parameters(tb, u, eps(Bs), **draws(Bs))
    dt = (time.perf_counter() - t0) / steps * (wl["B"] / Bs)
    what = f"{steps} full {kind.upper() if kind == 'sac' else 'DrQ'} update step(s) (B={wl['B']}" + (f" x {num_aug} augmentations" if num_aug > 1 else "") + f", N={wl['N']})" if Bs == wl["B"] else \
        f"{steps} update steps on a {Bs}-transition slice of the B={wl['B']}, N={wl['N']} batch, time scaled by {wl['B']}/{Bs}"
    warm = f"1 warm-up step on {warmup_batch} transitions" if (warmup_batch and warmup_batch < Bs) else "1 warm-up"
    return {"value": 1.0 / dt, "unit": "gradient steps/s", "cores": cores, "kind": "port", "extrapolated": Bs != wl["B"],
            "sample": f"{what} after {warm}, torch {torch.__version__} CPU, {cores} threads "
                      f"(box grants {usable_cpus()} of {os.cpu_count()} logical CPUs)"}


RC_EXCHANGE_IN_GRAPH = 75      # exit code of a rank whose step failed with the all-reduces captured in its hipGraph (launcher: retry segmented)


def exchange_in_graph_failure(err):
    """Is `err` the failure the launcher's one retry exists for?  A data-parallel run over RCCL whose exchanging step was
    captured whole (PCRL_CAPTURE_EXCHANGE != 0) and then either published no metrics (a collective inside the replayed graph
    hangs) or raised from the graph launch / RCCL.  Anything else -- out of memory, assertions, argument errors -- is not."""
    if os.environ.get("PCRL_CAPTURE_EXCHANGE", "1") == "0" or int(os.environ.get("WORLD_SIZE", "1")) < 2:
        return False
    if not isinstance(err, RuntimeError) or isinstance(err, (AssertionError, NotImplementedError)):
        return False
    text = str(err).lower()
    if "out of memory" in text:
        return False
    return any(k in text for k in ("did not publish its metrics", "without publishing its metrics", "nccl", "rccl", "hipgraph", "graph", "captur"))


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of THIS process -- one per GPU, each
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in its environment, exactly what torch.distributed.run
    hands them -- wait for all of them, forward rank 0's JSON line as the last line of stdout and return non-zero when any rank
    failed.  The parent makes no HIP call (importing torch does not initialise the device), and nothing is exec'ed.
    If the ranks fail with the gradient all-reduces captured in the step's hipGraph (the default over RCCL), they are started
    once more with PCRL_CAPTURE_EXCHANGE=0 (eager all-reduces between per-segment graphs); the line then says so."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]

    def attempt(extra_env):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                    HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **extra_env)
        base.setdefault("OMP_NUM_THREADS", "1")
        if args.share_gpu or args.dry_run_ranks:
            base.setdefault("HSA_ENABLE_SDMA", "0")       # ranks sharing one device: see main()
        base.setdefault("PCRL_STEP_TIMEOUT_S", "90")          # a step that publishes no metrics for 90 s ends its rank (and so the attempt)
        procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)),
                                  stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)) for r in range(n)]
        chunks = []
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        deadline, failed = time.time() + args.launch_timeout, []
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):      # a rank died: its peers would wait in a collective until RCCL's timeout
                break
            if time.time() > deadline:
                failed.append(f"no result after {args.launch_timeout:.0f} s")
                break
            time.sleep(0.05)
        for r, p in enumerate(procs):
            if p.poll() is None:              # still alive after a peer failed / after the deadline: stop exactly this child
                p.kill()
                p.wait()
                failed.append(f"rank {r} stopped")
            elif p.returncode != 0:
                failed.append(f"rank {r} rc {p.returncode}")
        reader.join(timeout=10)
        lines = [l for l in (chunks[0] if chunks else "").splitlines() if l.strip()]
        json_lines = [l for l in lines if l.startswith("{")]
        if not failed and len(json_lines) != 1:
            failed.append("rank 0 printed no result line")
        return failed, [l for l in lines if not l.startswith("{")], (json_lines[-1] if json_lines else None)

    failed, chatter, line = attempt({})
    note = None
    # Repeated ONLY on the dedicated signal: a rank left with RC_EXCHANGE_IN_GRAPH, which `__main__` uses when a step whose
    # all-reduces are nodes of its hipGraph failed while replaying or published nothing (the one thing a one-GPU box cannot
    # show).  Out of memory, an assertion, a missing result line or the deadline fail the run with the ranks' stderr above.
    if any(f.endswith(f"rc {RC_EXCHANGE_IN_GRAPH}") for f in failed) and args.backend == "nccl" and os.environ.get("PCRL_CAPTURE_EXCHANGE", "1") != "0":
        print(f"bench.py launcher: a rank reported a failure of the captured all-reduces ({failed}); once more with PCRL_CAPTURE_EXCHANGE=0", file=sys.stderr)
        note = f"first attempt (all-reduces captured in the step's hipGraph) failed: {failed}; this line is the PCRL_CAPTURE_EXCHANGE=0 run"
        failed, chatter, line = attempt({"PCRL_CAPTURE_EXCHANGE": "0"})
    for l in chatter:
        print(l)
    if failed:
        print(f"bench.py launcher: {n} ranks failed: {failed}", file=sys.stderr)
        return 1
    if note:
        d = json.loads(line)
        d["launcher_note"] = note
        line = json.dumps(d)
    print(line, flush=True)
    return 0


def main():
    args = parse()
    if os.environ.get("PCRL_BENCH_DUMP_AFTER_S"):
        # debugging a launch that does not come back: every thread's Python stack of THIS process to stderr after so many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PCRL_BENCH_DUMP_AFTER_S"]), exit=False)
    if args.dry_run_ranks:
        args.gpus, args.backend, args.share_gpu = args.dry_run_ranks, "gloo", True
    if args.share_gpu:
        # Ranks that SHARE one device copy through the compute queues, not the SDMA engines (read by the HIP runtime when it initialises,
        # i.e. below: nothing has touched the device yet).  With the engines, eight processes on one MI355X lose a rank to
        # HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION in 18 of 70 launches -- the victim is inside a host-to-device copy of its start-up, it has not
        # launched a kernel of this library -- and hang in a pageable copy when the collectives are staged through host memory; without
        # them 0 of 30 unlocked launches fail (profiles/r06_dry_run_loop.md).  A real multi-GPU run (a device per rank) keeps the engines.
        os.environ.setdefault("HSA_ENABLE_SDMA", "0")
    wl = dict(WORKLOADS[args.workload])
    if not args.replay_capacity:
        args.replay_capacity = wl.get("capacity", 2048)
    if args.batch:
        wl["B"], wl["desc"] = args.batch, wl["desc"] + f" [batch overridden to {args.batch}]"
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (the reference spawns its own ranks the same
        # way, pyrl/apis/run_rl.py:495-502) -- it never touches the GPU, the ranks are its children
        raise SystemExit(launch_ranks(args))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    if args.share_gpu:
        local_rank = 0
    lock_start = bool(args.share_gpu and world > 1 and args.start_lock)
    if not lock_start:
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
    start_lock = None
    if lock_start:
        # (the rendezvous above is gloo's, on the host; from here to the first synchronisation one rank at a time touches the device)
        import fcntl
        start_lock = open(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"pcrl_bench_start_{os.environ.get('MASTER_PORT', '0')}.lock"), "w")
        fcntl.flock(start_lock, fcntl.LOCK_EX)
        torch.cuda.set_device(local_rank)
    agent, C = build_agent(wl, b_rank, device, args.encoder_dtype)
    if start_lock is not None:
        torch.zeros(1 << 16, device=device).add_(1.0).sum().item()      # torch's first fill / elementwise / reduction kernels and the first copy back
        torch.cuda.synchronize()
        fcntl.flock(start_lock, fcntl.LOCK_UN)
        start_lock.close()
    if dist_on:
        agent.to_ddp(device_ids=["cuda"])                 # broadcasts rank 0's weights (as DDP's constructor does) and turns the exchange on
    if args.replay == "device":
        memory = device_ring(wl, args.replay_capacity, rank, device)
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
    if args.set_fused:
        agent._prepare()
        for item in args.set_fused:
            k, v = item.split("=", 1)
            assert agent._fused is not None and hasattr(agent._fused, k), f"--set-fused: the fused step has no attribute {k}"
            setattr(agent._fused, k, type(getattr(agent._fused, k))(int(v)) if isinstance(getattr(agent._fused, k), (bool, int)) else float(v))
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
    exch_mode = exchange_mode(agent, dist_on)
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
        for _ in range(2):          # untimed: the first eager launch of a kernel loads its code object (one 1-95 ms span otherwise)
            updates += 1
            agent.update_parameters(memory, updates)
        sync()
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
        spans, span_detail = timer.summary(), timer.detail()
        n_fwd, ms_fwd = spans.get("encoder_fwd", (0, float("nan")))
        steps_timed = timed_eager_steps if graphed else args.steps
        roof = encoder_roofline(agent, wl, C, b_rank, n_fwd, ms_fwd, steps_timed, args.workload, graphed, use_traffic=not args.batch and not dist_on)
        out = {
            "metric": "SAC gradient steps/sec (encoder+update) on B=256, N=1024 pts" if args.workload == "k1" else f"SAC gradient steps/sec ({args.workload})",
            "value": args.steps / elapsed, "unit": "gradient steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "f32split": "f32split"}.get(getattr(agent.encoder, "compute_dtype", "f32"), "f32"), "data": "synthetic",
            "config": {"workload": wl["desc"], "global_batch": wl["B"], "points": wl["N"], "channels": C, "action_dim": wl["A"],
                       "parallelism": f"dp{world}", "batch_per_gpu": b_rank,
                       "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if dist_on else None,
                       "rccl_ranks": torch.distributed.get_world_size() if dist_on else 1, "exchange": exch_mode,
                       "exchange_probe": exchange_probe(dist_on),
                       "hip_graphs": graphed, "device_warmup_seconds": args.device_warmup_seconds, "replay": args.replay + (f" ring of {args.replay_capacity} transitions, B sampled per step" if args.replay == "device" else " batch")},
            "roofline": roof,
            "step": step_roofline(wl, C, agent.encoder.mlp_spec, wl["B"], getattr(agent, "num_aug", 1), wl["A"], wl["S"], elapsed / args.steps * 1e3,
                                  getattr(agent.encoder, "compute_dtype", "f32") == "bf16", feat=feature_dim(agent),
                                  **dict(zip(("active_pts", "bwd_tiles"), active_points_per_cloud(agent, tiles=True)))),
            "roofline_gemm": gemm_roofline(timer, steps_timed),
            "kernels_ms": {k: dict({"launches": n, "avg_ms": ms}, **span_detail.get(k, {})) for k, (n, ms) in spans.items()},
        }
        if nocomm_ms is not None:
            out["ms_per_step_nocomm"] = nocomm_ms
            out["comm_ms_per_step"] = elapsed / args.steps * 1e3 - nocomm_ms
        if args.dry_run_ranks:
            out["dry_run"] = (f"{world} ranks sharing cuda:0 over gloo (host-side all-reduces): a rehearsal of launcher, rendezvous, to_ddp, the "
                              "data-parallel schedule and the extras -- not a measurement; HSA_ENABLE_SDMA=" + os.environ.get("HSA_ENABLE_SDMA", "unset")
                              + (", contexts brought up one after the other" if lock_start else ""))
        if dist_on and world == 1:
            out["debug"] = f"single-rank exchange over backend {args.backend}: data-parallel schedule with a one-rank process group"
        if not dist_on and not args.no_cpu_baseline:
            # bounded samples (about 10-30 s of CPU work each): every granted core on a slice of the batch that takes a few
            # seconds per step, and the reference's shipped single-thread setting (pyrl/utils/meta/__init__.py:38-49) on a
            # smaller slice
            points = wl["B"] * wl["N"] * getattr(agent, "num_aug", 1)
            all_cores_b = wl["B"] if points <= 300_000 else max(8, int(wl["B"] * 300_000 / points))
            out["cpu_baseline"] = cpu_baseline(agent, wl, args.cpu_steps, args.cpu_threads, sample_batch=all_cores_b)
            if not args.cpu_threads:
                # the reference's shipped setting (one thread): ONE timed step on the same sample as the all-cores leg -- for K1 the
                # full batch (8-9 s), no slice scaling
                out["cpu_baseline_1thread"] = cpu_baseline(agent, wl, 1, 1, sample_batch=all_cores_b, warmup_batch=4)
        line = json.dumps(out)
    # ---- extra objects (never the reported value), every rank takes part: the collectives of a data-parallel run are inside ----
    plain_k1 = args.workload == "k1" and not args.batch and args.encoder_dtype is None and args.replay == "device" and not args.single_rank_exchange
    extras = {}
    # The extras must never cost the headline line: if they do not finish in time (a collective that hangs in a shape the headline
    # did not exercise), every rank leaves on its own timer and rank 0 prints the line it already has.
    import threading

    def give_up():
        if rank == 0:
            import ctypes
            d = json.loads(line)
            d["extras_error"] = f"extra workloads did not finish within {args.extra_timeout:.0f} s; skipped"
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
            print(json.dumps(d), flush=True)
        os._exit(0)
    if dist_on:
        torch.distributed.barrier()        # every rank starts its watchdog at the same point
    timer_x = threading.Timer(args.extra_timeout, give_up)
    timer_x.daemon = True
    timer_x.start()
    try:
        held = {"agent": agent, "memory": memory}      # handed over: the extras free the headline's agent / ring before building theirs
        agent = memory = None
        run_extras(args, extras, plain_k1, rank, world, device, dist_on, held)
    except BaseException as err:           # an extra must never cost the headline: report it and print the line already built
        import traceback
        traceback.print_exc()
        timer_x.cancel()
        if rank == 0:
            import ctypes
            d = json.loads(line)
            d.update(extras)
            d["extras_error"] = f"{type(err).__name__}: {str(err).splitlines()[0] if str(err) else ''}"[:300]
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
            print(json.dumps(d), flush=True)
        os._exit(0)                        # peers still inside a collective leave on their own watchdog
    timer_x.cancel()
    if rank == 0:
        if extras:
            out.update(extras)
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


def run_extras(args, extras, plain_k1, rank, world, device, dist_on, held):
    """The extra objects of a default K1 line (never the reported value); fills `extras` as it goes."""
    memory = held.get("memory")
    held.pop("agent", None)
    if plain_k1 and not dist_on and not args.no_experimental:
        # the same step with the EXPERIMENTAL split-precision encoder forward (three-term bf16 split of the fp32 contractions,
        # within ~3e-6 of the exact kernel, argmax exact on the reference fixtures)
        torch.cuda.empty_cache()
        extras["experimental_f32split"] = dict(
            side_rate("k1", rank, world, device, dist_on, max(args.steps // 2, 10), max(args.warmup // 2, 30), "f32split", memory=memory, graphs=not args.no_graphs),
            note="encoder conv1/conv2 and the backward data-gradient GEMMs as three-term bf16 splits (pcrl_encoder_{fwd,bwd}_f32split); opt-in, not the reported value")
    if plain_k1 and not args.no_extra_workloads and WORKLOADS["k3"]["B"] % world == 0:
        # BASELINE config 4 on the same ranks: the shape the >= 6x strong-scaling target of north_star is stated for.  Each
        # `--gpus N` line carries it, so the scaling of K3 can be read off the driver's own N = 1, 2, 4, 8 runs.
        memory = None
        held.clear()
        torch.cuda.empty_cache()
        extras["config4_k3"] = dict(side_rate("k3", rank, world, device, dist_on, args.extra_steps, 30, graphs=not args.no_graphs, roofline=True),
                                    workload=WORKLOADS["k3"]["desc"])
        # BASELINE configs 3 and 5 as worded, same protocol, each with its own roofline object (and, at N = 1, config 3's CPU
        # baseline: the DrQ restatement on a bounded slice) -- every BASELINE config is in the driver's line
        for key, name in (("config3_k2", "k2"), ("config5_k4", "k4")):
            if WORKLOADS[name]["B"] % world:
                continue
            torch.cuda.empty_cache()
            extras[key] = dict(side_rate(name, rank, world, device, dist_on, args.extra_steps, 30, graphs=not args.no_graphs, roofline=True,
                                         cpu_steps=(2 if (name == "k2" and not dist_on and not args.no_cpu_baseline) else 0), cpu_threads=args.cpu_threads),
                               workload=WORKLOADS[name]["desc"])


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as err:
        # a rank that failed must END (its peers wait for it in a collective, the launcher watches exit codes): no interpreter /
        # process-group teardown that could itself wait for a wedged stream
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(RC_EXCHANGE_IN_GRAPH if exchange_in_graph_failure(err) else 1)
