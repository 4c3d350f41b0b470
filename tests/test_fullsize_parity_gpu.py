"""Whole-step parity at the sizes BASELINE.json names (not the small golden fixtures): the HIP update step vs the
PyTorch-CPU restatement of the reference's `update_parameters` (oracle/torch_ref.py, itself pinned to fixtures
captured from the reference: tests/test_oracle_golden.py) on identical replay batches with the policy / jitter noise
injected.  Compared after every update: every returned metric, the encoder argmax is implied by the gradients, the
gradient of every parameter BEFORE the optimizer step (the flat gradient buffers), and all parameters after it.

ReLU decisions.  A hidden unit whose pre-activation is within rounding of zero is switched on or off by the summation
order, and that switches a whole sample's contribution to the layer's weight gradient (1/B of its scale: measured 1.7e-2
of max|g| at B = 128 for ONE unit of 2 x 128 x 1024).  With ~10^6 gradient-carrying hidden units per step such units
exist in every full-size step, so the restatement is run with the HIP step's own decisions for the three gradient-
carrying head passes (oracle/torch_ref.py::linear_mlp, masks=) and the decisions are compared separately: every
disagreement must sit on a pre-activation below 1e-5 in the restatement, and their number is reported.  (The encoder's
per-point ReLUs are not injected; a switched unit there is one (point, channel) of ~5 x 10^6 and moves a gradient by
~1e-4 of its scale, which is the encoder tolerance below.)

Protocol: every update starts from the restatement's parameters (after update 1 the agent's parameters, target network
included, are overwritten with the restatement's; the Adam moments stay the agent's own).  Free-running, the two
trajectories separate for a reason that has nothing to do with kernel accuracy: Adam's first steps move every entry by
~lr whatever its gradient's size, a 1e-7 difference decides the sign of a ~1e-8 gradient, the parameters then differ by
1e-5 on a handful of entries, and in the next forward that flips a few ReLUs of the 256 x 1024 hidden units, each flip
changing a weight-gradient row by 1/256 of its scale (measured: 6e-3 of max|g| on the actor's first layer at update 2,
against 5e-7 at update 1).  Free-running agreement over several updates is what the small golden fixtures captured from
the reference check (test_update_step_gpu.py).

north_star asks for 1e-5 in fp32.  What is measured (maximum over all quantities of a case) is written to
gpurun_out/parity_fullsize_<case>.json and summarised in profiles/r02_parity_errors.md; the asserts below are the
measured maxima with head-room of about 2x, and they state per quantity where 1e-5 absolute is not the right yardstick:
  * metrics: relative to max(1, |ref|);
  * gradients: relative to the tensor's largest |entry| (a sum over up to 256 x 8192 points of fp32 products in a
    different association order than ATen's cannot agree to 1e-5 of each tiny entry);
  * parameters: absolute -- Adam's first steps move every entry by ~lr = 1e-3 whatever the gradient's size
    (update = lr * g / (|g| + 1e-8)), so an entry whose gradient is ~1e-8 turns a 1e-10 gradient difference into a
    1e-5 parameter difference; the fraction of such entries is reported.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Memory:
    def __init__(self, batch):
        self.batch = batch

    def sample(self, batch_size):
        return self

    def to_torch(self, device=None, non_blocking=False):
        from pointcloud_rl_amd.utils.torch_utils import to_torch
        return to_torch(self.batch, device=device, non_blocking=non_blocking)


CASES = {
    # BASELINE config 2 (K1): the bench workload itself, heads 1024 wide
    "k1_sac_dmc_b256_n1024": dict(kind="sac", cfg="sac_dmc", B=256, N=1024, A=6, S=0, obs_kw={}),
    # BASELINE config 4 (K3), one rank's share: ManiSkill nets [128,128,256]->128, C=7, S=68, A=22
    "k3_sac_maniskill_b128_n1200": dict(kind="sac", cfg="sac_maniskill", B=128, N=1200, A=22, S=68, obs_kw=dict(seg=1)),
    # BASELINE config 5 (K4), one rank's share of B=512 over 8 GPUs: N=8192 -> split clouds / two-stage pool in the
    # forward, 256 tiles per cloud; backward and update at that N
    "k4_sac_dmc_b64_n8192": dict(kind="sac", cfg="sac_dmc", B=64, N=8192, A=6, S=0, obs_kw={}),
    # BASELINE config 3 (K2) in fp32 (the bf16 kernels have no reference counterpart, see test_encoder_*_gpu.py):
    # DrQ, 2 augmentations, fused jitter, 64 x 2 clouds
    "k2_drq_maniskill_b64x2_n1200_f32": dict(kind="drq", cfg="drq_maniskill", B=64, N=1200, A=22, S=68, obs_kw=dict(seg=1)),
    # the EXPERIMENTAL split-precision encoder (compute_dtype="f32split": conv1 / conv2 and the backward's data-gradient GEMMs
    # as three-term bf16 splits) on the K1 workload, same tolerances as the exact kernels
    "k1_sac_dmc_b256_n1024_f32split": dict(kind="sac", cfg="sac_dmc", B=256, N=1024, A=6, S=0, obs_kw={}, encoder_dtype="f32split"),
}

# measured maxima (MI355X, round 2) x ~2-3; see profiles/r02_parity_errors.md
#   metrics 1.8e-5; head gradients 1.2e-5 (2e-6 in the cases without an encoder event upstream); encoder gradients 9e-7
#   when no discrete event falls into the step (k2, k4), 1.3e-4 (k1) / 1.1e-3 (k3, B = 128) when one does -- a switched
#   per-point ReLU or a near-tie of the max-pool decided differently by ATen's summation order: one (cloud, channel)
#   contribution of B x 256 that add incoherently, i.e. ~1 / sqrt(B x 256) of the gradient's scale; parameters: 3.7e-4
#   worst entry (bounded by 2 lr = 2e-3: Adam's sign-like first steps on an entry whose gradient is ~0), 2.6e-5 of the
#   entries beyond 1e-5; argmax: equal to ATen's except at value gaps below 1e-6 (bit-exact against the C oracle, which
#   sums in the kernel's order: test_encoder_fwd_gpu.py).
TOL = dict(metric_rel=3e-5, head_grad_rel_to_max=3e-5, encoder_grad_rel_to_max=3e-3, param_abs=2.1e-3, param_frac_over_1e5=1e-4,
           flip_max_preact=2e-5, argmax_gap=1e-6)


def _build(case, dev):
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    C = 6 + case["obs_kw"].get("seg", 0)
    if case["cfg"] == "sac_dmc":
        cfg = configs.sac_dmc(C, case["A"], case["B"])
    elif case["cfg"] == "sac_maniskill":
        cfg = configs.sac_maniskill(C, case["A"], case["S"], case["B"])
    else:
        cfg = configs.drq_maniskill(C, case["A"], case["S"], case["B"])
    cfg["env_params"] = configs.env_params({"xyz": [3, case["N"]], "rgb": [3, case["N"]]}, case["A"])
    if case.get("encoder_dtype"):
        cfg["actor_cfg"]["nn_cfg"]["visual_nn_cfg"]["compute_dtype"] = case["encoder_dtype"]
    torch.manual_seed(0)
    return build_agent(cfg)


def _flat_grads(agent, which):
    fb = agent._flat[which]
    return {n: g.detach().cpu().numpy().copy() for n, g in zip(fb.names, fb.views(fb.grad))}


@pytest.mark.parametrize("name", list(CASES))
def test_full_size_update_matches_cpu_restatement(cuda, name):
    from oracle import torch_ref
    from pointcloud_rl_amd.synthetic import make_batch_np
    case = CASES[name]
    B, N, A, S = case["B"], case["N"], case["A"], case["S"]
    agent = _build(case, cuda)
    params = {n: p.detach().clone() for n, p in agent.named_parameters()}
    ref = torch_ref.RefAgent(params, kind=case["kind"], gamma=agent.gamma, reward_scale=agent.reward_scale, alpha=0.1,
                             target_entropy=agent.target_entropy, update_coeff=agent.update_coeff["default"],
                             num_aug=getattr(agent, "num_aug", 2), mirror_redundancy=False)
    agent = agent.to(cuda)
    assert agent.use_fused_step
    g = torch.Generator().manual_seed(5)
    num_aug = getattr(agent, "num_aug", 1) if case["kind"] == "drq" else 1
    worst = dict(metric_rel=0.0, head_grad_rel_to_max=0.0, encoder_grad_rel_to_max=0.0, param_abs=0.0, param_frac_over_1e5=0.0,
                 flip_max_preact=0.0, flips=0, argmax_gap=0.0, argmax_differs=0)
    detail = {}
    for u in (1, 2):
        batch_np = make_batch_np(B, N, A, seed=10 + u, agent=S, **case["obs_kw"])
        cpu_batch = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v))
                     for k, v in batch_np.items()}
        eps = [torch.randn(B * num_aug, A, generator=g)] + ([torch.randn(B, A, generator=g)] if u % 2 == 0 else [])
        jit = [torch.empty(B * num_aug, 3, N).uniform_(-0.01, 0.01, generator=g) for _ in range(2)] if case["kind"] == "drq" else None
        agent.actor.head.noise_override = [e.to(cuda) for e in eps]
        if jit is not None:
            agent.obs_aug[0].noise_override = [j.to(cuda) for j in jit]
        pre = None
        if case["kind"] == "sac":          # the restatement's pre-pool features with the parameters this update starts from
            with torch.no_grad():
                pre = torch_ref.pointnet_prepool(ref.P, {k: v for k, v in cpu_batch["obs"].items() if k not in ("agent", "state")})
        got = agent.update_parameters(Memory(batch_np), u)
        if pre is not None:
            mine = agent._fused.last_argmax.cpu().long()
            vals, theirs = pre.max(-1)
            differs = mine != theirs
            worst["argmax_differs"] += int(differs.sum())
            if differs.any():            # the point this implementation picked must hold (to rounding) the same maximum
                gap = (vals - pre.gather(-1, mine[..., None])[..., 0])[differs]
                worst["argmax_gap"] = max(worst["argmax_gap"], float(gap.abs().max()))
            del pre
        assert agent._fused is not None, "the fused HIP step must be the one under test"
        masks = agent._fused.relu_decisions(B * num_aug, B if u % 2 == 0 else None)
        masks = {k: [[m.cpu() for m in head] for head in v] if k != "pi" else [m.cpu() for m in v] for k, v in masks.items()}
        want = ref.update_parameters(cpu_batch, u, eps, jit, relu_masks=masks)
        for n_bad, z_bad in ref.flips:
            worst["flips"] += n_bad
            worst["flip_max_preact"] = max(worst["flip_max_preact"], z_bad)
        assert got.keys() == want.keys()
        for k, v in want.items():
            err = abs(got[k] - v) / max(1.0, abs(v))
            detail[f"u{u}/metric/{k}"] = err
            worst["metric_rel"] = max(worst["metric_rel"], err)
        # gradients before the optimizer step
        sets = [("critic", ref.last_grads["critic"])] + ([("actor", ref.last_grads["actor"])] if u % 2 == 0 else [])
        for which, ref_grads in sets:
            mine = _flat_grads(agent, which)
            for n, gm in mine.items():
                ref_name = ("critic." + n) if (which == "critic" and n.startswith("values.") and "visual_nn" not in n) else None
                if which == "critic" and "visual_nn" in n:
                    ref_name = "actor.backbone.visual_nn." + n.split("visual_nn.", 1)[1]
                if which == "actor":
                    ref_name = "actor." + n
                gr = ref_grads[ref_name].numpy()
                scale = max(float(np.abs(gr).max()), 1e-12)
                err = float(np.abs(gm - gr).max()) / scale
                detail[f"u{u}/grad/{which}/{n}"] = err
                key = "encoder_grad_rel_to_max" if "visual_nn" in n else "head_grad_rel_to_max"
                worst[key] = max(worst[key], err)
        # parameters after this update's optimizer steps (and Polyak), then continue from the restatement's
        n_over = n_all = 0
        with torch.no_grad():
            for n, p in agent.named_parameters():
                err = np.abs(p.detach().cpu().numpy() - ref.P[n].detach().numpy())
                detail[f"u{u}/param/{n}"] = float(err.max())
                worst["param_abs"] = max(worst["param_abs"], float(err.max()))
                n_over += int((err > 1e-5).sum())
                n_all += err.size
                p.copy_(ref.P[n].detach().to(p.device))
        agent.encoder.invalidate_packed()
        worst["param_frac_over_1e5"] = max(worst["param_frac_over_1e5"], n_over / n_all)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"parity_fullsize_{name}.json"), "w") as f:
        json.dump(dict(case=name, worst=worst, detail=detail), f, indent=1)
    print(name, worst)
    for k, tol in TOL.items():
        assert worst[k] <= tol, (name, k, worst[k], tol, sorted(detail.items(), key=lambda kv: -kv[1])[:5])
