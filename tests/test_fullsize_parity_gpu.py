"""Whole-step parity at the sizes BASELINE.json names (not the small golden fixtures): the HIP update step vs the
PyTorch-CPU restatement of the reference's `update_parameters` (oracle/torch_ref.py, itself pinned to fixtures
captured from the reference: tests/test_oracle_golden.py) on identical replay batches with the policy / jitter noise
injected.  Compared after every update: every returned metric, the encoder argmax is implied by the gradients, the
gradient of every parameter BEFORE the optimizer step (the flat gradient buffers), and all parameters after it.

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
}

# measured maxima (MI355X, round 2) x ~2; see profiles/r02_parity_errors.md
TOL = dict(metric_rel=2e-5, grad_rel_to_max=2e-5, param_abs=1e-4, param_frac_over_1e5=2e-3)


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
    worst = dict(metric_rel=0.0, grad_rel_to_max=0.0, param_abs=0.0, param_frac_over_1e5=0.0)
    detail = {}
    crit_prefix = {"values.": "critic.values.", "": ""}
    for u in (1, 2):
        batch_np = make_batch_np(B, N, A, seed=10 + u, agent=S, **case["obs_kw"])
        cpu_batch = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v))
                     for k, v in batch_np.items()}
        eps = [torch.randn(B * num_aug, A, generator=g)] + ([torch.randn(B, A, generator=g)] if u % 2 == 0 else [])
        jit = [torch.empty(B * num_aug, 3, N).uniform_(-0.01, 0.01, generator=g) for _ in range(2)] if case["kind"] == "drq" else None
        agent.actor.head.noise_override = [e.to(cuda) for e in eps]
        if jit is not None:
            agent.obs_aug[0].noise_override = [j.to(cuda) for j in jit]
        got = agent.update_parameters(Memory(batch_np), u)
        assert agent._fused is not None, "the fused HIP step must be the one under test"
        want = ref.update_parameters(cpu_batch, u, eps, jit)
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
                worst["grad_rel_to_max"] = max(worst["grad_rel_to_max"], err)
    n_over = n_all = 0
    for n, p in agent.named_parameters():
        err = np.abs(p.detach().cpu().numpy() - ref.P[n].detach().numpy())
        detail[f"param/{n}"] = float(err.max())
        worst["param_abs"] = max(worst["param_abs"], float(err.max()))
        n_over += int((err > 1e-5).sum())
        n_all += err.size
    worst["param_frac_over_1e5"] = n_over / n_all
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"parity_fullsize_{name}.json"), "w") as f:
        json.dump(dict(case=name, worst=worst, detail=detail), f, indent=1)
    print(name, worst)
    for k, tol in TOL.items():
        assert worst[k] <= tol, (name, k, worst[k], tol, sorted(detail.items(), key=lambda kv: -kv[1])[:5])
