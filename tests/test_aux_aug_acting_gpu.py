"""GPU tests: stand-alone segmented max / augmentation kernels, the fused augmentations of the encoder
(explicit noise, affine, in-kernel Philox statistics) and the acting path."""
import numpy as np
import pytest
import torch

from helpers import make_encoder_weights, make_obs

pytestmark = pytest.mark.gpu


def test_segmax_matches_oracle_and_torch(cuda):
    from oracle import c_oracle
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(0)
    for B, c, N in [(3, 7, 1000), (2, 5, 33), (4, 16, 1024)]:
        x = g.randn(B, c, N).astype(np.float32)
        x[0, 0, :] = 0.0                         # all-equal row -> index 0
        x[0, 1, 5] = x[0, 1, 9] = 9.0            # exact tie -> first index
        x[1, 2, 7] = np.nan                      # NaN wins
        x[1, 2, 3] = np.inf
        ref_v, ref_i = c_oracle.segmax(x)
        tv, ti = torch.from_numpy(x).max(-1)
        assert np.array_equal(ref_i, ti.numpy().astype(np.int32))
        v, i = hip.segmax_fwd(torch.from_numpy(x).to(cuda))
        assert np.array_equal(i.cpu().numpy(), ref_i)
        assert np.array_equal(np.isnan(v.cpu().numpy()), np.isnan(ref_v))
        m = ~np.isnan(ref_v)
        assert np.array_equal(v.cpu().numpy()[m], ref_v[m])
        go = g.randn(B, c).astype(np.float32)
        dx = hip.segmax_bwd(torch.from_numpy(go).to(cuda), i, N).cpu().numpy()
        want = np.zeros_like(x)
        np.put_along_axis(want, ref_i[..., None].astype(np.int64), go[..., None], axis=-1)
        assert np.array_equal(dx, want)


def test_augment_xyz_kernel_and_fused_affine_jitter(cuda):
    """GlobalRotScaleTrans (rot + per-axis scale + translation) followed by explicit jitter: the stand-alone
    kernel, the fused encoder load and the reference formula (apply_rot_trans: R x + t) agree."""
    from oracle import c_oracle
    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd.augmentations import GlobalRotScaleTrans
    B, N = 5, 150
    obs = make_obs(B, N, seed=21)
    torch.manual_seed(3)
    t = GlobalRotScaleTrans(main_key="xyz", req_keys=["xyz"], rot_range=[-0.5, 0.5], rot_axis="y",
                            scale_ratio_range=[0.8, 1.2], translation_range=[0.1, 0.2, 0.3], shift_height=True)
    mat = t.sample_matrix(B, cuda)
    noise = torch.from_numpy(np.random.RandomState(1).uniform(-0.01, 0.01, (B, 3, N)).astype(np.float32)).to(cuda)
    xyz = torch.from_numpy(obs["xyz"]).to(cuda)
    out = hip.augment_xyz(xyz, affine=mat, jitter_noise=noise)
    m = mat.cpu()
    want = torch.einsum("bin,bji->bjn", torch.from_numpy(obs["xyz"]), m[:, :, :3]) + m[:, :, 3:] + noise.cpu()
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), atol=1e-6)
    # fused into the encoder: encode(raw obs + aug desc) == encode(materialised augmented obs), bit for bit
    w = make_encoder_weights(6, 64, 128, 256, seed=2)
    wt = {k: torch.from_numpy(v).to(cuda) for k, v in w.items()}
    ew, _ = hip.make_encoder_weights(wt["w0"], wt["b0"], wt["w1"], wt["g1"], wt["be1"], wt["w2"], wt["g2"], wt["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(6, 64, 128, 256) // 4, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    rgb = torch.from_numpy(obs["rgb"]).to(cuda)
    d1, k1 = hip.make_cloud_desc({"xyz": xyz, "rgb": rgb})
    p_fused, a_fused = hip.encoder_fwd(d1, ew, packed, aug=hip.make_aug_desc(affine=mat, jitter_noise=noise))
    d2, k2 = hip.make_cloud_desc({"xyz": out, "rgb": rgb})
    p_mat, a_mat = hip.encoder_fwd(d2, ew, packed)
    assert torch.equal(p_fused, p_mat) and torch.equal(a_fused, a_mat)
    pooled_ref, arg_ref = c_oracle.encoder_fwd(c_oracle.preprocess({"xyz": out.cpu().numpy(), "rgb": obs["rgb"]}), w)
    assert np.array_equal(a_mat.cpu().numpy(), arg_ref) and np.array_equal(p_mat.cpu().numpy().view(np.uint32), pooled_ref.view(np.uint32))


def test_philox_jitter_statistics_and_streams(cuda):
    """In-kernel Philox4x32-10 jitter: range, first two moments, independence across calls / axes / clouds,
    reproducibility for a fixed (seed, offset), and row remapping for strided sub-batches."""
    from pointcloud_rl_amd import hip
    B, N = 64, 2048
    xyz = torch.zeros(B, 3, N, device=cuda)
    a = hip.augment_xyz(xyz, jitter_range=[-0.01, 0.01], seed=11, offset=0)
    a2 = hip.augment_xyz(xyz, jitter_range=[-0.01, 0.01], seed=11, offset=0)
    b = hip.augment_xyz(xyz, jitter_range=[-0.01, 0.01], seed=11, offset=1)
    c = hip.augment_xyz(xyz, jitter_range=[-0.01, 0.01], seed=12, offset=0)
    assert torch.equal(a, a2) and not torch.equal(a, b) and not torch.equal(a, c)
    v = a.cpu().numpy().astype(np.float64)
    assert v.min() >= -0.01 and v.max() < 0.01
    n = v.size
    assert abs(v.mean()) < 4 * (0.02 / np.sqrt(12)) / np.sqrt(n)
    assert abs(v.var() - 0.02 ** 2 / 12) < 0.02 * 0.02 ** 2 / 12
    u = (v + 0.01) / 0.02
    hist = np.histogram(u, bins=32, range=(0, 1))[0]
    assert np.abs(hist - n / 32).max() < 6 * np.sqrt(n / 32)
    corr = lambda p, q: abs(np.corrcoef(p.ravel(), q.ravel())[0, 1])
    assert corr(v[:, 0], v[:, 1]) < 0.01 and corr(v[:-1], v[1:]) < 0.01 and corr(v[:, :, :-1], v[:, :, 1:]) < 0.01
    assert corr(v, b.cpu().numpy()) < 0.01
    # the device-side offset slot is what a hipGraph replay reads
    off = torch.tensor([1], dtype=torch.int64, device=cuda)
    assert torch.equal(hip.augment_xyz(xyz, jitter_range=[-0.01, 0.01], seed=11, offset=0, offset_tensor=off), b)
    # cloud b of a strided sub-batch (every 2nd cloud) sees the noise row 2b of the full batch
    sub = hip.augment_xyz(torch.zeros(B // 2, 3, N, device=cuda), jitter_range=[-0.01, 0.01], seed=11, offset=0, row_mul=2, row_add=0)
    assert torch.equal(sub, a[::2])


def test_drq_eager_philox_step_runs_and_differs_between_calls(cuda):
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    cfg = configs.drq_dmc(6, 6, 8, head_hidden=64)
    cfg["env_params"] = configs.env_params({"xyz": [3, 128], "rgb": [3, 128]}, 6)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    agent.enable_graphs(warmup=1)
    mem = SyntheticReplay(8, 128, 6, seed=2, device=cuda)
    rets = [agent.update_parameters(mem, u) for u in range(1, 9)]
    assert all(np.isfinite(list(r.values())).all() for r in rets)
    assert len(agent._graphs) == 2 and agent._fused is not None
    # same batch every step, so identical consecutive critic losses would mean the replayed graph re-used its noise
    losses = [r["drq/critic_loss"] for r in rets]
    assert len(set(np.round(losses, 7))) == len(losses)


def test_acting_path(cuda):
    """BaseAgent.forward (reference module_utils.py:147-159): numpy obs in, actions out; (actions, None) for
    rnn_mode="with_states" as Rollout/Evaluation call it (rollout.py:91-97, evaluation.py:167-168)."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    cfg = configs.sac_maniskill(7, 8, 10, 4, head_hidden=64)
    cfg["env_params"] = configs.env_params({"xyz": [3, 90], "rgb": [3, 90]}, 8)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda).eval()
    obs = make_obs(2, 90, seed=4, seg=1)
    obs["agent"] = np.random.RandomState(0).randn(2, 10).astype(np.float32)
    with agent.no_sync(mode="actor"):
        actions, states = agent(obs, rnn_mode="with_states")
    assert states is None and actions.shape == (2, 8) and actions.abs().max() <= 1.0
    mean_a = agent(obs, mode="eval")
    assert torch.equal(mean_a, agent(obs, mode="eval"))                 # deterministic evaluation mode
    # against the oracle's restatement of the same modules
    from oracle import torch_ref
    P = {n: p.detach().cpu() for n, p in agent.named_parameters()}
    tobs = {k: torch.from_numpy(v) for k, v in obs.items()}
    feat, _ = torch_ref.visuomotor(P, "actor.backbone.final_mlp.mlp.", tobs)
    want = torch.tanh(feat[:, :8]) * P["actor.head.scale"] + P["actor.head.bias"]
    np.testing.assert_allclose(mean_a.cpu().numpy(), want.numpy(), atol=1e-5)
