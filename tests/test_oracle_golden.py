"""Pins the oracle: C restatement and PyTorch restatement vs golden vectors captured from the
reference itself (tools/gen_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle, torch_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENC_FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "encoder_*.npz")))
STEP_FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "sac_*.npz")) + glob.glob(os.path.join(GOLDEN, "drq_*.npz")))


def enc_weights(d, prefix="w/"):
    g = lambda k: d[prefix + k]
    return dict(w0=g("conv.mlp.conv0.weight")[..., 0], b0=g("conv.mlp.conv0.bias"), w1=g("conv.mlp.conv1.weight")[..., 0],
                g1=g("conv.mlp.norm1.weight"), be1=g("conv.mlp.norm1.bias"), w2=g("conv.mlp.conv2.weight")[..., 0],
                g2=g("conv.mlp.norm2.weight"), be2=g("conv.mlp.norm2.bias"))


def test_fixtures_present():
    assert len(ENC_FIXTURES) >= 3 and len(STEP_FIXTURES) >= 3


@pytest.mark.parametrize("path", ENC_FIXTURES, ids=os.path.basename)
def test_c_oracle_encoder_matches_reference(path):
    d = np.load(path)
    obs = {k[4:]: d[k] for k in d.files if k.startswith("obs/")}
    pooled, argmax = c_oracle.encoder_fwd(c_oracle.preprocess(obs), enc_weights(d))
    assert float(d["min_live_top2_gap"]) > 1e-5          # fixture has no near-ties, so indices must agree exactly
    assert np.array_equal(argmax, d["argmax"])
    np.testing.assert_allclose(pooled, d["pooled"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("path", ENC_FIXTURES, ids=os.path.basename)
def test_torch_ref_encoder_matches_reference(path):
    d = np.load(path)
    obs = {k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("obs/")}
    P = {torch_ref.ENC + k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w/")}
    pre = torch_ref.pointnet_prepool(P, obs)
    vals, idx = pre.max(-1)
    assert np.array_equal(idx.numpy(), d["argmax"])
    np.testing.assert_allclose(vals.numpy(), d["pooled"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(torch_ref.pointnet_forward(P, obs).numpy(), d["feature"], rtol=0, atol=1e-5)


def make_ref_agent(d, **kw):
    gamma, reward_scale, alpha, target_entropy, aui, tui, num_aug = [float(x) for x in d["meta/hyper"]]
    kind = "drq" if str(d["meta/agent_type"]) == "DrQ" else "sac"
    return torch_ref.RefAgent(torch_ref.params_from_fixture(d), kind=kind, gamma=gamma, reward_scale=reward_scale, alpha=alpha,
                              target_entropy=target_entropy, actor_update_interval=int(aui), target_update_interval=int(tui),
                              update_coeff=float(d["meta/update_coeff_default"]), num_aug=int(num_aug),
                              svea=bool(d["meta/svea"]) if "meta/svea" in d.files else False, **kw)


def fixture_grad_name(tag, name):
    """Name under which gen_golden.py stored a gradient (module-relative, as the module's named_parameters())."""
    if tag == "critic":
        if name.startswith(torch_ref.ENC):
            return "values.0.backbone.visual_nn." + name[len(torch_ref.ENC):]
        return name[len("critic."):]
    if tag == "actor":
        return name[len("actor."):]
    return name


def fixture_draws(d, u, prefix):
    out, i = [], 0
    while f"u{u}/{prefix}{i}" in d.files:
        out.append(torch.from_numpy(d[f"u{u}/{prefix}{i}"]))
        i += 1
    return out


@pytest.mark.parametrize("path", STEP_FIXTURES, ids=os.path.basename)
@pytest.mark.parametrize("mirror", [True, False], ids=["six-encodes", "dedup"])
def test_torch_ref_update_matches_reference(path, mirror):
    d = np.load(path)
    torch.set_num_threads(4)
    agent = make_ref_agent(d, mirror_redundancy=mirror)
    n_updates = int(d["meta/dims"][4])
    for u in range(1, n_updates + 1):
        agent.encoder_passes[0] = 0
        batch = torch_ref.batch_from_fixture(d, u)
        if "meta/use_episode_dones" in d.files and bool(d["meta/use_episode_dones"]):
            batch["dones"] = batch["episode_dones"]              # sac.py:106-107
        ret = agent.update_parameters(batch, u, fixture_draws(d, u, "eps"), fixture_draws(d, u, "jitter"))
        if mirror:
            assert agent.encoder_passes[0] == int(d[f"u{u}/n_encoder_passes"])
        for k, v in ret.items():
            ref = float(d[f"u{u}/ret/{k.split('/', 1)[1]}"])
            assert abs(v - ref) <= 2e-5 * max(1.0, abs(ref)), (u, k, v, ref)
        for tag, gd in agent.last_grads.items():
            for name, g in gd.items():
                key = f"u{u}/grad_{tag}/{fixture_grad_name(tag, name)}"
                if u <= 2:
                    np.testing.assert_allclose(g.numpy(), d[key], rtol=1e-4, atol=2e-6, err_msg=key)
        for name, p in agent.P.items():
            s = d[f"u{u}/paramsum/{name}"]
            got = np.array([float(p.detach().double().sum()), float(p.detach().double().abs().sum())])
            np.testing.assert_allclose(got, s, rtol=1e-5, atol=1e-5, err_msg=f"u{u} {name}")
        if u == 2:
            for name, p in agent.P.items():
                np.testing.assert_allclose(p.detach().numpy(), d[f"u2/param/{name}"], rtol=0, atol=1e-5, err_msg=name)


def test_step_fixture_grad_keys_are_checked():
    # guard: the name mapping above must actually hit the stored gradients
    d = np.load(STEP_FIXTURES[-1])
    keys = [k for k in d.files if k.startswith("u2/grad_")]
    assert any("visual_nn" in k for k in keys) and any(k.startswith("u2/grad_actor/") for k in keys)
