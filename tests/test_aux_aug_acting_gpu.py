"""GPU tests: stand-alone segmented max / augmentation kernels, the fused augmentations of the encoder
(explicit noise, affine, in-kernel Philox statistics) and the acting path."""
import os

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


def test_device_replay_sampling_matches_numpy_take(cuda):
    """push_batch (with wrap-around) + sample == the reference's numpy ring + take with the same RandomState."""
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    cap, N, A = 40, 33, 5
    mem = DeviceReplay(cap, device=cuda, seed=7, host_rng=True)
    ring = None
    pos = count = 0
    for i, n in enumerate((16, 16, 16)):                 # third push wraps
        items = make_batch_np(n, N, A, seed=i, seg=1, agent=3)
        mem.push_batch(items)
        flat = {"obs/" + k: v for k, v in items["obs"].items()} | {"next_obs/" + k: v for k, v in items["next_obs"].items()} | \
               {k: v for k, v in items.items() if not isinstance(v, dict)}
        if ring is None:
            ring = {k: np.zeros((cap,) + v.shape[1:], v.dtype) for k, v in flat.items()}
        for j in range(n):
            for k, v in flat.items():
                ring[k][(pos + j) % cap] = v[j]
        pos, count = (pos + n) % cap, count + n
    assert len(mem) == cap
    rs = np.random.RandomState(7)
    first_ptrs = None
    for _ in range(3):
        batch = mem.sample(12).to_torch(device=cuda)
        idx = rs.randint(0, cap, 12)
        assert batch.persistent
        ptrs = (batch["obs"]["xyz"].data_ptr(), batch["rewards"].data_ptr())
        first_ptrs = first_ptrs or ptrs
        assert ptrs == first_ptrs                         # staging keeps its addresses
        for k, v in ring.items():
            node = batch
            for part in k.split("/"):
                node = node[part]
            np.testing.assert_array_equal(node.cpu().numpy(), v[idx], err_msg=k)
        assert batch["is_valid"].all() and batch["worker_indices"].shape == (12, 1)
    # rows drawn in the kernel: in range, uniform, different on every call, and the gathered data are those rows
    dev = DeviceReplay(cap, device=cuda, seed=11)
    dev.push_batch(make_batch_np(25, N, A, seed=3, seg=1, agent=3))        # partially filled ring: only rows [0, 25)
    seen, prev = np.zeros(25), None
    for _ in range(40):
        batch = dev.sample(64).to_torch(device=cuda)
        idx = dev.last_indices(64).cpu().numpy()
        assert idx.min() >= 0 and idx.max() < 25 and (prev is None or (idx != prev).any())
        np.testing.assert_array_equal(batch["obs"]["xyz"].cpu().numpy(), dev.storage["obs/xyz"].cpu().numpy()[idx])
        np.testing.assert_array_equal(batch["actions"].cpu().numpy(), dev.storage["actions"].cpu().numpy()[idx])
        seen += np.bincount(idx, minlength=25)
        prev = idx
    assert seen.min() > 0.6 * seen.mean() and seen.max() < 1.4 * seen.mean()      # 2560 draws over 25 rows
    with pytest.raises(RuntimeError):
        DeviceReplay(8, device="cpu")


def test_update_from_device_replay_reads_staging_in_place(cuda):
    """With graphs on, the agent adopts the replay's staging tensors as graph inputs (no copies) and every replay sees the new sample."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    B, N, A = 8, 64, 4
    cfg = configs.sac_dmc(6, A, B, head_hidden=64)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    mem = DeviceReplay(64, device=cuda, seed=3)
    mem.push_batch(make_batch_np(64, N, A, seed=5))
    agent.enable_graphs(warmup=1)
    rets = [agent.update_parameters(mem, u) for u in range(1, 9)]
    assert agent._graphs and agent._static_batch["obs"]["xyz"].data_ptr() == mem._staging[B][0]["obs/xyz"].data_ptr()
    assert all(np.isfinite(list(r.values())).all() for r in rets)
    assert len({round(r["sac/q_target"], 6) for r in rets[4:]}) > 1      # different samples -> different statistics


def test_policy_noise_in_kernel_is_standard_normal_and_advances(cuda):
    from pointcloud_rl_amd import hip
    B, A = 4096, 6
    feat = torch.zeros(B, 2 * A, device=cuda)            # mean 0, log_std 0 -> u = eps
    scale, bias = torch.ones(A, device=cuda), torch.zeros(A, device=cuda)
    step = torch.zeros(1, dtype=torch.int32, device=cuda)
    draws = []
    for s, d in ((0, 0), (0, 1), (1, 0)):
        step.fill_(s)
        eps, act, nlp = torch.empty(B, A, device=cuda), torch.empty(B, A, device=cuda), torch.empty(B, device=cuda)
        hip.tanh_gaussian_sample_fwd(feat, 2 * A, 1234, step, d, eps, scale, bias, B, A, -10.0, 2.0, 1e-6, act, A, nlp)
        np.testing.assert_allclose(act.cpu().numpy(), np.tanh(eps.cpu().numpy()), atol=1e-6)
        draws.append(eps.cpu().numpy().ravel())
    for e in draws:
        assert abs(e.mean()) < 0.03 and abs(e.std() - 1) < 0.03 and abs((e ** 3).mean()) < 0.08 and abs((e ** 4).mean() - 3) < 0.25
    assert abs(np.corrcoef(draws[0], draws[1])[0, 1]) < 0.03 and abs(np.corrcoef(draws[0], draws[2])[0, 1]) < 0.03


def test_random_downsample_is_an_index_on_the_point_load(cuda):
    """RandomDownSample (pcd_aug.py:231-268): encoding the stored cloud through the shared index equals encoding the
    sliced tensors the reference would build -- forward bit for bit (argmax counts subsampled positions), backward too."""
    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd.augmentations import RandomDownSample
    from pointcloud_rl_amd.networks.pointnet import materialize
    B, N = 6, 300
    obs = make_obs(B, N, seed=8, seg=1)
    dobs = {k: torch.from_numpy(v).to(cuda) for k, v in obs.items()}
    torch.manual_seed(5)
    aug = RandomDownSample(main_key="xyz", req_keys=["xyz", "rgb", "seg"], drop_ratio=0.3, fixed_ratio=True)
    out = aug(dobs)
    index = out.aug["point_index"]
    assert index.dtype == torch.int32 and index.numel() == N - int(N * 0.3) and index.unique().numel() == index.numel()
    sliced = materialize(out)
    assert sliced["xyz"].shape == (B, 3, index.numel()) and torch.equal(sliced["rgb"], dobs["rgb"][..., index.long()])
    w = make_encoder_weights(7, 128, 128, 256, seed=4)
    wt = {k: torch.from_numpy(v).to(cuda) for k, v in w.items()}
    ew, _ = hip.make_encoder_weights(wt["w0"], wt["b0"], wt["w1"], wt["g1"], wt["be1"], wt["w2"], wt["g2"], wt["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(7, 128, 128, 256) // 4, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    d1, k1 = hip.make_cloud_desc(dobs)
    a1 = hip.make_aug_desc(**out.aug)
    p1, i1 = hip.encoder_fwd(d1, ew, packed, aug=a1)
    d2, k2 = hip.make_cloud_desc({k: v.contiguous() for k, v in sliced.items()})
    p2, i2 = hip.encoder_fwd(d2, ew, packed)
    assert torch.equal(p1, p2) and torch.equal(i1, i2) and int(i1.max()) < index.numel()
    g = torch.randn_like(p1)
    assert torch.equal(hip.encoder_bwd(d1, ew, packed, i1, g, aug=a1), hip.encoder_bwd(d2, ew, packed, i2, g))
    # random drop count (fixed_ratio=False) and max_num_points variants
    var = RandomDownSample(main_key="xyz", req_keys=["xyz", "rgb", "seg"], drop_ratio=0.3, fixed_ratio=False)(dobs).aug
    assert var["point_index"].numel() == N and N - int(N * 0.3) < int(var["point_count"]) <= N       # the count travels by pointer
    np.random.seed(0)
    k_cpu = RandomDownSample(main_key="xyz", req_keys=["xyz", "rgb", "seg"], drop_ratio=0.3, fixed_ratio=False)(
        {k: v.cpu() for k, v in dobs.items()}).aug["point_index"].numel()
    np.random.seed(0)
    assert k_cpu == N - np.random.randint(int(N * 0.3))               # on CPU tensors: the reference's own numpy draw, index sliced
    assert RandomDownSample(main_key="xyz", req_keys=["xyz", "rgb", "seg"], max_num_points=100)(dobs).aug["point_index"].numel() == 100
    with pytest.raises(NotImplementedError):
        RandomDownSample(main_key="xyz", req_keys=["xyz"], max_num_points=100)(dobs)       # rgb / seg would keep all points
    with pytest.raises(RuntimeError):
        hip.encoder_fwd(d1, ew, packed, aug=hip.make_aug_desc(point_index=torch.zeros(N + 1, dtype=torch.int32, device=cuda)))


def test_drq_step_with_random_downsample(cuda):
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    cfg = configs.drq_dmc(6, 6, 8, head_hidden=64, obs_aug=dict(type="RandomDownSample", main_key="xyz", req_keys=["xyz", "rgb"],
                                                                   drop_ratio=0.3, fixed_ratio=True))
    cfg["env_params"] = configs.env_params({"xyz": [3, 128], "rgb": [3, 128]}, 6)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    agent.enable_graphs(warmup=1)
    mem = SyntheticReplay(8, 128, 6, seed=2, device=cuda)
    rets = [agent.update_parameters(mem, u) for u in range(1, 9)]
    assert all(np.isfinite(list(r.values())).all() for r in rets)
    assert agent._graphs and agent._fused is not None
    losses = [r["drq/critic_loss"] for r in rets[4:]]
    assert len(set(np.round(losses, 7))) == len(losses)        # every replay draws a new subset


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("B,N,count", [(6, 300, 211), (6, 300, 300), (2, 1200, 841), (2, 1200, 1), (3, 64, 33), (300, 96, 70)])
def test_random_count_downsample_equals_the_sliced_cloud(cuda, dtype, B, N, count):
    """RandomDownSample(fixed_ratio=False) on the GPU: the full permutation + a device-resident count (include/pcrl.h n_index_ptr)
    encodes exactly like the cloud sliced to the first `count` positions -- pooled values, argmax and every gradient bit for bit;
    few clouds (split over workgroups: segments past the count stay empty) and many, counts that are not tile multiples."""
    from pointcloud_rl_amd import hip
    obs = make_obs(B, N, seed=B + N, seg=1)
    dobs = {k: torch.from_numpy(v).to(cuda) for k, v in obs.items()}
    w = make_encoder_weights(7, 64, 128, 256, seed=4)
    wt = {k: torch.from_numpy(v).to(cuda) for k, v in w.items()}
    ew, _ = hip.make_encoder_weights(wt["w0"], wt["b0"], wt["w1"], wt["g1"], wt["be1"], wt["w2"], wt["g2"], wt["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(7, 64, 128, 256) // 4, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    perm = torch.from_numpy(np.random.RandomState(count).permutation(N).astype(np.int32)).to(cuda)
    cnt = torch.full((1,), count, dtype=torch.int32, device=cuda)
    d1, _k = hip.make_cloud_desc(dobs)
    a_var = hip.make_aug_desc(point_index=perm, point_count=cnt, jitter_range=(-0.01, 0.01), seed=3, offset=5)
    p1, i1 = hip.encoder_fwd(d1, ew, packed, aug=a_var, bf16=dtype == "bf16")
    a_cut = hip.make_aug_desc(point_index=perm[:count].contiguous(), jitter_range=(-0.01, 0.01), seed=3, offset=5)
    if count == N:                 # same Philox counters only when the logical N agrees; otherwise compare without jitter
        p2, i2 = hip.encoder_fwd(d1, ew, packed, aug=a_cut, bf16=dtype == "bf16")
        assert torch.equal(p1, p2) and torch.equal(i1, i2)
    a_var = hip.make_aug_desc(point_index=perm, point_count=cnt)
    a_cut = hip.make_aug_desc(point_index=perm[:count].contiguous())
    p1, i1 = hip.encoder_fwd(d1, ew, packed, aug=a_var, bf16=dtype == "bf16")
    p2, i2 = hip.encoder_fwd(d1, ew, packed, aug=a_cut, bf16=dtype == "bf16")
    assert torch.equal(p1, p2) and torch.equal(i1, i2) and int(i1.max()) < count
    g = torch.randn_like(p1)
    assert torch.equal(hip.encoder_bwd(d1, ew, packed, i1, g, aug=a_var, pooled=p1, bf16=dtype == "bf16"),
                       hip.encoder_bwd(d1, ew, packed, i2, g, aug=a_cut, pooled=p2, bf16=dtype == "bf16"))
    # the count is read at run time: the same descriptor, a new value
    cnt.fill_(max(1, count // 2))
    p3, i3 = hip.encoder_fwd(d1, ew, packed, aug=a_var, bf16=dtype == "bf16")
    p4, i4 = hip.encoder_fwd(d1, ew, packed, aug=hip.make_aug_desc(point_index=perm[:max(1, count // 2)].contiguous()), bf16=dtype == "bf16")
    assert torch.equal(p3, p4) and torch.equal(i3, i4)
    with pytest.raises(RuntimeError):
        hip.encoder_fwd(d1, ew, packed, aug=_aug_with_count_only(hip, cnt))


def _aug_with_count_only(hip, cnt):
    aug = hip.make_aug_desc(jitter_range=(-0.01, 0.01), seed=1)
    aug.n_index_ptr = cnt.data_ptr()
    return aug


def test_drq_reference_pn_dropout_config_follows_the_count_under_graph_replay(cuda):
    """configs/mfrl/drq/{dm_control,maniskill}/pn_dropout.py ship RandomDownSample(drop_ratio=0.3, fixed_ratio=False): the number of kept
    points changes every call (pcd_aug.py:244-246).  The obs_aug dict is the reference's, verbatim; the step is captured and replayed,
    and the replays must see different counts, all in (N - int(0.3 N), N]."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    N = 128
    obs_aug = dict(type="RandomDownSample", main_key="xyz", req_keys=["xyz", "rgb", "pos_encoding"], drop_ratio=0.3, fixed_ratio=False)
    cfg = configs.drq_dmc(6, 6, 8, head_hidden=64, obs_aug=obs_aug)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, 6)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    agent.enable_graphs(warmup=1)
    mem = SyntheticReplay(8, N, 6, seed=2, device=cuda)
    aug = agent.obs_aug.transforms[0]
    rets, counts = [], []
    for u in range(1, 15):
        rets.append(agent.update_parameters(mem, u))
        counts.append(int(aug.last_count))
    assert agent._graphs and agent._fused is not None and getattr(agent, "_use_graphs", False)
    assert all(np.isfinite(list(r.values())).all() for r in rets)
    replayed = counts[4:]                                       # both (actor / critic-only) graphs are captured by then
    assert len(set(replayed)) >= 3, counts
    assert all(N - int(0.3 * N) < c <= N for c in counts), counts
    losses = [r["drq/critic_loss"] for r in rets[4:]]
    assert len(set(np.round(losses, 7))) == len(losses)


def test_device_replay_without_replacement_walks_the_reference_epoch_order(cuda):
    """with_replacement=False: the index stream equals SamplingStrategy.get_index's (sampling_strategy.py:32-48) for the same
    seed -- shuffled epoch order, re-shuffled when exhausted, re-drawn after a push -- and every epoch visits each row once."""
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    cap, N, A, B = 24, 16, 3, 5
    mem = DeviceReplay(cap, device=cuda, seed=3, with_replacement=False)
    mem.push_batch(make_batch_np(cap, N, A, seed=1))
    rs = np.random.RandomState(3)
    items = np.arange(cap); rs.shuffle(items); pos = 0
    seen = []
    for _ in range(9):                      # 4 batches per epoch (drop_last), then a re-shuffle
        if pos + B > cap:
            rs.shuffle(items); pos = 0
        want = items[pos:pos + B]; pos += B
        batch = mem.sample(B).to_torch(device=cuda)
        np.testing.assert_array_equal(mem.last_indices(B).cpu().numpy(), want)
        np.testing.assert_array_equal(batch["rewards"].cpu().numpy(), mem.storage["rewards"].cpu().numpy()[want])
        seen.append(want)
    assert len(np.unique(np.concatenate(seen[:4]))) == 4 * B
    assert mem.sample(cap + 1, auto_restart=False) is None      # cannot be served without a restart


@pytest.mark.parametrize("mode", ["with", "without"])
def test_device_replay_reproduces_the_reference_replay_memory(cuda, mode):
    """tests/golden/ref_replay.npz was produced by the reference's ReplayMemory + OneStepTransition (tools/gen_golden_replay.py):
    the same pushes (the third wraps around the ring) and the same seed give the same ten sampled batches, key by key."""
    from pointcloud_rl_amd.replay import DeviceReplay
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_replay.npz"))
    mem = DeviceReplay(40, device=cuda, seed=7, with_replacement=(mode == "with"), host_rng=True)
    for i in range(3):
        items = {}
        for k in z.files:
            if k.startswith(f"push{i}/"):
                node, parts = items, k.split("/")[1:]
                for part in parts[:-1]:
                    node = node.setdefault(part, {})
                node[parts[-1]] = z[k]
        mem.push_batch(items)
    assert [len(mem), mem.position] == z[f"{mode}/len_position"].tolist()
    for s in range(10):
        batch = mem.sample(6).to_torch(device=cuda)
        keys = [k for k in z.files if k.startswith(f"{mode}/sample{s}/")]
        assert keys
        for k in keys:
            node = batch
            for part in k.split("/")[2:]:
                node = node[part]
            np.testing.assert_array_equal(node.cpu().numpy(), z[k], err_msg=k)


@pytest.mark.parametrize("tag,kw", [("ratio", dict(drop_ratio=0.3, fixed_ratio=True)), ("maxpts", dict(max_num_points=17))])
def test_random_downsample_matches_the_reference_class(cuda, tag, kw):
    """tests/golden/ref_downsample.npz: what the reference's RandomDownSample returned (tools/gen_golden_downsample.py).  With the
    reference's index injected, this class keeps the same number of points and `materialize` yields the same tensors for every key."""
    from pointcloud_rl_amd.augmentations import RandomDownSample
    from pointcloud_rl_amd.networks.pointnet import materialize
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_downsample.npz"))
    obs = {k: torch.from_numpy(z[f"in/{k}"]).to(cuda) for k in ("xyz", "rgb", "seg")}
    aug = RandomDownSample(main_key="xyz", req_keys=["xyz", "rgb", "seg"], **kw)
    torch.manual_seed(0)
    assert aug(obs).aug["point_index"].numel() == int(z[f"{tag}/n"])          # same count from the same keywords
    aug.index_override = [torch.from_numpy(z[f"{tag}/index"])]
    got = materialize(aug(obs))
    for k in ("xyz", "rgb", "seg"):
        np.testing.assert_array_equal(got[k].cpu().numpy(), z[f"{tag}/out/{k}"], err_msg=k)


def test_acting_matches_the_reference_agent_loaded_from_its_checkpoint(cuda):
    """The reference agent that wrote tests/golden/ref_sac_dmc_small.ckpt also acted on a small observation
    (tools/gen_golden_checkpoint.py -> ref_sac_dmc_small_acting.npz).  Loading its checkpoint here and acting on the same
    observation gives the same actions (1e-5) in the deterministic modes."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.utils.checkpoint import load_checkpoint
    gold = os.path.join(os.path.dirname(__file__), "golden")
    cfg = configs.sac_dmc(6, 6, 8, head_hidden=64)
    cfg["env_params"] = configs.env_params({"xyz": [3, 64], "rgb": [3, 64]}, 6)
    agent = build_agent(cfg)
    load_checkpoint(agent, os.path.join(gold, "ref_sac_dmc_small.ckpt"), map_location="cpu", strict=True)
    agent = agent.to(cuda).eval()
    z = np.load(os.path.join(gold, "ref_sac_dmc_small_acting.npz"))
    obs = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith("obs/")}
    for mode in ("eval", "mean"):
        got = agent(obs, mode=mode)
        np.testing.assert_allclose(got.cpu().numpy(), z[mode], atol=1e-5, rtol=0, err_msg=mode)
    acts, states = agent(obs, mode="eval", rnn_mode="with_states")
    assert states is None and torch.equal(acts, agent(obs, mode="eval"))


def test_device_replay_batches_are_encoded_in_one_launch_with_the_same_result(cuda):
    """DeviceReplay stages obs/<key> and next_obs/<key> as the halves of one allocation; the fused step then encodes s and s'
    in ONE launch.  Same metrics and parameters (bitwise) as the two-launch path on the same data."""
    import torch
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent, fused
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    B, N, A = 16, 128, 6

    def run(merge):
        cfg = configs.sac_dmc(6, A, B, head_hidden=64)
        cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
        torch.manual_seed(0)
        agent = build_agent(cfg).to(cuda)
        mem = DeviceReplay(64, device=cuda, seed=3)
        mem.push_batch(make_batch_np(64, N, A, seed=5))
        orig = fused._adjacent_halves
        seen = []
        fused._adjacent_halves = (lambda a, b: seen.append(orig(a, b)) or seen[-1]) if merge else (lambda a, b: None)
        try:
            rets = [agent.update_parameters(mem, u) for u in range(1, 5)]
        finally:
            fused._adjacent_halves = orig
        if merge:
            assert all(s is not None and s["xyz"].shape[0] == 2 * B for s in seen)
        return rets, {n: p.detach().clone() for n, p in agent.named_parameters()}

    (r1, p1), (r2, p2) = run(True), run(False)
    assert r1 == r2
    for n in p1:
        assert torch.equal(p1[n], p2[n]), n


def test_fused_acting_path_equals_the_module_tree(cuda):
    """methods/acting.py (seven launches) against the actor's module tree on the same observation: mean action to 1e-5 (fp32
    GEMMs in a different order), sampled action with the same injected noise, (actions, None) for rnn_mode="with_states"."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.methods.acting import FusedActor
    cfg = configs.sac_maniskill(7, 8, 10, 4, head_hidden=256)
    cfg["env_params"] = configs.env_params({"xyz": [3, 150], "rgb": [3, 150]}, 8)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda).eval()
    assert FusedActor.supported(agent.actor)
    obs = make_obs(3, 150, seed=4, seg=1)
    obs["agent"] = np.random.RandomState(0).randn(3, 10).astype(np.float32)
    eps = torch.randn(3, 8)
    out = {}
    for fused in (True, False):
        agent.use_fused_acting = fused
        agent.__dict__.pop("_fused_actor", None)
        mean_a = agent(obs, mode="eval")
        agent.actor.head.noise_override = [eps.to(cuda)]
        samp, states = agent(obs, mode="explore", rnn_mode="with_states")
        assert states is None and not agent.actor.head.noise_override
        out[fused] = (mean_a, samp)
        assert ("_fused_actor" in agent.__dict__) == fused
    for a, b in zip(out[True], out[False]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-5, rtol=0)
    assert not torch.equal(out[True][0], out[True][1])


def test_state_driven_sampling_launch_equals_the_explicit_one_and_counts_itself(cuda):
    """pcrl_replay_sample_gather_state reads (draw, size) from device memory and advances draw itself: same rows as the launch that
    gets them as arguments, call after call, also when the valid size changes between calls."""
    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    cap, N, A, B = 96, 32, 4, 48
    mem = DeviceReplay(cap, device=cuda, seed=21)
    mem.push_batch(make_batch_np(40, N, A, seed=3))
    ref_idx = torch.zeros(B, dtype=torch.int32, device=cuda)
    for call in range(6):
        if call == 3:
            mem.push_batch(make_batch_np(30, N, A, seed=4))                 # size 40 -> 70
        assert mem.state[:3].tolist() == [call, len(mem), 0] and int(mem.state[3:].abs().sum()) == 0
        mem.sample(B)
        flat, _, idx, _, segs = mem._stage(B)
        got = {k: v.clone() for k, v in flat.items()}
        hip.replay_sample_gather(segs, B, len(mem), cap, mem.seed, call, ref_idx)
        assert torch.equal(idx, ref_idx) and int(idx.max()) < len(mem)
        for k, v in flat.items():
            assert torch.equal(v, got[k]), k
    assert mem.state[:3].tolist() == [6, 70, 0] and int(mem.state[3:].abs().sum()) == 0 and mem.draws == 6


def test_pack_job_rides_on_the_replay_sampling_launch(cuda):
    """pcrl_encoder_pack_attach_to_gather: the encoder's re-pack and the column-gather jobs attached to it run as extra workgroups of the
    next replay sampling launch -- same image, same columns, same sampled rows as the separate launches; the flush afterwards is a no-op;
    a job no sampling launch takes is run by the flush; a dropped job never runs."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import make_encoder_weights
    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    cap, N, A, B, H, K0, col0 = 64, 96, 6, 24, 1024, 56, 50
    w = {k: torch.from_numpy(v).to(cuda) for k, v in make_encoder_weights(6, 64, 128, 256, seed=3).items()}
    ew, _ = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
    n_packed = hip.encoder_packed_bytes(6, 64, 128, 256) // 4
    want = torch.empty(n_packed, device=cuda)
    hip.encoder_pack_weights(ew, want)
    W0 = torch.randn(2, H, K0, device=cuda)
    want_cols = W0[:, :, col0:col0 + A].permute(0, 2, 1).contiguous()

    def replay():
        mem = DeviceReplay(cap, device=cuda, seed=21)
        mem.push_batch(make_batch_np(40, N, A, seed=3))
        return mem
    plain, riding = replay(), replay()
    for call in range(3):
        plain.sample(B)
        packed = torch.full((n_packed,), float("nan"), device=cuda)
        cols = torch.full((2, A, H), float("nan"), device=cuda)
        hip.pack_attach_cols([(W0, H * K0, 2, H, K0, col0, A, cols)])
        hip.encoder_pack_attach_to_gather(ew, packed)
        torch.cuda.synchronize()
        assert torch.isnan(packed).all() and torch.isnan(cols).all()          # nothing has run yet
        riding.sample(B)                                                       # ... this launch carries both
        torch.cuda.synchronize()
        assert torch.equal(packed, want) and torch.equal(cols, want_cols)
        fa, fb = plain._stage(B)[0], riding._stage(B)[0]
        for k in fa:
            assert torch.equal(fa[k], fb[k]), k
        assert plain.state.tolist() == riding.state.tolist()
        packed.fill_(float("nan"))
        hip.encoder_pack_flush_pending()                                       # taken already: no launch
        hip.pack_flush_cols()
        torch.cuda.synchronize()
        assert torch.isnan(packed).all()
    packed = torch.full((n_packed,), float("nan"), device=cuda)
    hip.encoder_pack_attach_to_gather(ew, packed)
    with pytest.raises(RuntimeError, match="already pending"):
        hip.encoder_pack_attach_to_gather(ew, packed)
    hip.encoder_pack_flush_pending()                                           # no sampling launch came: a launch of its own
    torch.cuda.synchronize()
    assert torch.equal(packed, want)
    packed.fill_(float("nan"))
    hip.encoder_pack_attach_to_gather(ew, packed)
    hip.encoder_pack_drop_pending()
    riding.sample(B)
    hip.encoder_pack_flush_pending()
    torch.cuda.synchronize()
    assert torch.isnan(packed).all()


@pytest.mark.parametrize("head_hidden,A,entry_pack", [(64, 4, "1"), (1024, 6, "1"), (1024, 6, "0")])
def test_sampling_inside_the_captured_step_equals_the_eager_run(cuda, monkeypatch, head_hidden, A, entry_pack):
    """With a DeviceReplay that draws its rows on the device, the sampling launch is the first node of the captured step and
    the metrics come back through the pinned mirror's flag (no copy node, no stream synchronisation): every returned metric and
    every parameter equal the eager run's bit for bit, and the replay counted every call.  The sampling launch also carries the critic
    phase's re-pack (heads 1 024 wide, A <= 8: with the target heads' action-column image; FusedStep.entry_pack = False: the separate launches)."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    B, N, steps = 8, 64, 12

    def run(graphs):
        cfg = configs.sac_dmc(6, A, B, head_hidden=head_hidden)
        cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
        torch.manual_seed(0)
        agent = build_agent(cfg).to(cuda)
        agent._prepare()
        agent._fused.entry_pack = entry_pack == "1"
        mem = DeviceReplay(64, device=cuda, seed=3)
        mem.push_batch(make_batch_np(48, N, A, seed=5))
        if graphs:
            agent.enable_graphs(warmup=1)
        rets = []
        for u in range(1, steps + 1):
            if u == 7:
                mem.push_batch(make_batch_np(16, N, A, seed=6))            # the ring grows between two replays
            rets.append(agent.update_parameters(mem, u))
        torch.cuda.synchronize()
        return agent, mem, rets

    eager, mem_e, rets_e = run(False)
    graph, mem_g, rets_g = run(True)
    assert len(graph._graphs) == 2 and all(s is mem_g for s in graph._graph_sampler.values()) and len(graph._graph_flag) == 2
    assert mem_g.draws == mem_e.draws == steps and mem_g.state[:3].tolist() == mem_e.state[:3].tolist() == [steps, 64, 0]
    for ra, rb in zip(rets_e, rets_g):
        assert ra == rb
    for (n, p), (_, q) in zip(eager.named_parameters(), graph.named_parameters()):
        assert torch.equal(p, q), n


@pytest.mark.parametrize("tag,kw", [("h3_with", dict(horizon=3, with_replacement=True)), ("h3_without", dict(horizon=3, with_replacement=False)),
                                    ("episode_with", dict(horizon=-1, with_replacement=True))])
def test_device_replay_tstep_sampling_reproduces_the_reference(cuda, tag, kw):
    """T-step sampling (sampling_strategy.py:105-246: [B, H] blocks of consecutive transitions of one worker's episode, whole
    episodes padded for horizon -1) on the device ring: every sampled key and the validity mask equal what the reference's
    ReplayMemory returned for the same pushes and seed (tests/golden/ref_replay_tstep.npz)."""
    import os
    from pointcloud_rl_amd.replay import DeviceReplay
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_replay_tstep.npz"))
    mem = DeviceReplay(48, device=cuda, sampling_cfg=dict(type="TStepTransition", seed=11, **kw))
    for i in range(4):
        items = {}
        for k in d.files:
            if k.startswith(f"push{i}/"):
                parts = k.split("/")[1:]
                node = items
                for part in parts[:-1]:
                    node = node.setdefault(part, {})
                node[parts[-1]] = d[k]
        mem.push_batch(items)
    assert len(mem) == int(d[f"{tag}/len_units"][0]) and len(mem.tstep) == int(d[f"{tag}/len_units"][1])
    for s_ in range(6):
        batch = mem.sample(5).to_torch(device=cuda)
        keys = [k[len(f"{tag}/sample{s_}/"):] for k in d.files if k.startswith(f"{tag}/sample{s_}/")]
        assert keys
        for k in keys:
            node = batch
            for part in k.split("/"):
                node = node[part]
            want = d[f"{tag}/sample{s_}/{k}"]
            assert tuple(node.shape) == want.shape, (k, tuple(node.shape), want.shape)
            assert np.array_equal(node.cpu().numpy(), want), (tag, s_, k)


def _swap_agent_and_rings(cuda, B, N, A):
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    cfg = configs.sac_dmc(6, A, B, head_hidden=64)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    rings = []
    for seed in (3, 4):
        mem = DeviceReplay(64, device=cuda, seed=seed)
        mem.push_batch(make_batch_np(64, N, A, seed=10 + seed))
        rings.append(mem)
    return agent, rings


def _consumed_rows_match(agent, mem, B):
    """The observations the captured step reads are the rows the replay's latest sampling launch drew."""
    idx = mem.last_indices(B).long()
    xyz = agent._static_batch["obs"]["xyz"]
    return torch.equal(xyz, mem.storage["obs/xyz"][idx]) and torch.equal(agent._static_batch["rewards"], mem.storage["rewards"][idx])


def test_swapping_the_device_replay_recaptures_on_the_new_staging_tensors(cuda):
    """An agent captured with DeviceReplay A and then fed DeviceReplay B must read B's staging tensors (the captured sampling
    launch writes those): every step after the swap consumes the rows B drew, q_target keeps varying, and the run equals an eager
    agent fed the same sequence bit for bit."""
    B, N, A = 8, 64, 4
    schedule = [0] * 6 + [1] * 8 + [0] * 4

    def run(graphs):
        agent, rings = _swap_agent_and_rings(cuda, B, N, A)
        if graphs:
            agent.enable_graphs(warmup=1)
        rets = []
        for u, which in enumerate(schedule, 1):
            rets.append(agent.update_parameters(rings[which], u))
            if graphs and agent._graphs:
                torch.cuda.synchronize()
                assert _consumed_rows_match(agent, rings[which], B), u
        return agent, rings, rets

    eager, _, rets_e = run(False)
    graph, rings_g, rets_g = run(True)
    assert [r_["sac/q_target"] for r_ in rets_e] == [r_["sac/q_target"] for r_ in rets_g]
    assert len({r_["sac/q_target"] for r_ in rets_g[6:14]}) == 8           # not one frozen batch
    assert rings_g[0].draws == 10 and rings_g[1].draws == 8
    for (n, p), (_, q) in zip(eager.named_parameters(), graph.named_parameters()):
        assert torch.equal(p, q), n


def test_synthetic_then_device_replay_does_not_freeze_the_batch(cuda):
    """First calls with a non-persistent memory (the static batch is made of clones), then a DeviceReplay: the variants captured
    later must not put the sampling launch in front of a step that reads the clones -- either the static batch is re-made from the
    replay's staging tensors or the batch is copied in every step; both ways the consumed rows are the drawn rows."""
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    B, N, A = 8, 64, 4
    agent, rings = _swap_agent_and_rings(cuda, B, N, A)
    agent.enable_graphs(warmup=1)
    syn = SyntheticReplay(B, N, A, seed=9, device=cuda)
    for u in range(1, 4):                                  # eager warm-up + the first capture, on clones
        agent.update_parameters(syn, u)
    seen = []
    for u in range(4, 16):
        seen.append(agent.update_parameters(rings[0], u)["sac/q_target"])
        torch.cuda.synchronize()
        assert _consumed_rows_match(agent, rings[0], B), u
    assert len(set(seen)) == len(seen)


def test_drq_jitter_counter_follows_the_replay_that_feeds_the_step(cuda, monkeypatch):
    """DrQ's fused jitter reads its Philox offset from the replay's device draw counter (`DeviceReplay.state[0]`), an ADDRESS
    baked into the captured encoder launches.  Swapping the replay must drop the graphs captured with the old address (else the
    noise would repeat on every step: the old replay's counter no longer advances); without the sampling launch in the graph
    (agent.graph_sampling = False) the shared counter is not used at all.  Both ways the graph-replayed run equals the eager run fed
    the same sequence bit for bit."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    B, N, A = 8, 64, 4
    schedule = [0] * 6 + [1] * 6 + [0] * 4

    def run(graphs, sampling):
        cfg = configs.drq_dmc(6, A, B, head_hidden=64, obs_aug=dict(configs.JITTER, seed=5))
        cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
        torch.manual_seed(0)
        agent = build_agent(cfg).to(cuda)
        agent.graph_sampling = sampling == "1"
        rings = []
        for seed in (3, 4):
            mem = DeviceReplay(64, device=cuda, seed=seed)
            mem.push_batch(make_batch_np(64, N, A, seed=10 + seed))
            rings.append(mem)
        if graphs:
            agent.enable_graphs(warmup=1)
        rets, ptrs = [], []
        for u, which in enumerate(schedule, 1):
            rets.append(agent.update_parameters(rings[which], u))
            ptrs.append(agent.__dict__.get("_jitter_counter_ptr"))
        return agent, rings, rets, ptrs

    for sampling in ("1", "0"):
        eager, _, rets_e, _ = run(False, sampling)
        graph, rings, rets_g, ptrs = run(True, sampling)
        if sampling == "1":
            assert ptrs[5] == rings[0].state.data_ptr() and ptrs[11] == rings[1].state.data_ptr() and ptrs[-1] == rings[0].state.data_ptr()
        else:
            assert set(ptrs) == {None}
        assert [r_["drq/critic_loss"] for r_ in rets_e] == [r_["drq/critic_loss"] for r_ in rets_g], sampling
        for (n, p), (_, q) in zip(eager.named_parameters(), graph.named_parameters()):
            assert torch.equal(p, q), (sampling, n)


def test_sampling_state_shorter_than_the_batch_is_refused(cuda):
    """pcrl_replay_sample_gather_state is told how many words `state` holds and returns PCRL_E_ARG below 3 + B."""
    import ctypes
    from pointcloud_rl_amd import _lib, hip
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    mem = DeviceReplay(32, device=cuda, seed=1)
    mem.push_batch(make_batch_np(32, 16, 4, seed=2))
    B = 8
    _, _, idx, _, segs = mem._stage(B)
    lib = _lib.lib()
    rc = lib.pcrl_replay_sample_gather_state(segs, len(segs), B, ctypes.c_int64(32), ctypes.c_uint64(1), ctypes.c_void_p(mem.state.data_ptr()),
                                             ctypes.c_int64(3 + B - 1), ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(hip.raw_stream()))
    assert rc == -1 and b"state" in lib.pcrl_last_error()          # PCRL_E_ARG
    torch.cuda.synchronize()
    assert mem.state[0].item() == 0


def test_graph_replayed_acting_equals_the_eager_launches_and_follows_the_parameters(cuda):
    """After agent.enable_graphs() the acting path replays one hipGraph per (mode, observation signature): the mean action equals
    the eager launches bit for bit, also after the parameters were updated in place by training steps (the graph re-packs the
    encoder weights itself), every call returns its own tensor, sampled actions differ from call to call, and an observation of
    another shape gets its own graph."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    B, N, A = 8, 96, 6
    cfg = configs.sac_dmc(6, A, B, head_hidden=256)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    mem = DeviceReplay(64, device=cuda, seed=1)
    mem.push_batch(make_batch_np(64, N, A, seed=2))
    agent.update_parameters(mem, 1)                       # parameters now live in the flat buffers
    agent.enable_graphs(warmup=1)
    obs = {k: torch.from_numpy(v).to(cuda) for k, v in make_obs(3, N, seed=5).items()}
    obs5 = {k: torch.from_numpy(v).to(cuda) for k, v in make_obs(5, N, seed=6).items()}

    def eager(o, mode="eval"):
        fa = agent._fused_actor
        fa.use_graphs = False
        try:
            return agent(o, mode=mode)
        finally:
            fa.use_graphs = True

    outs = [agent(obs, mode="eval") for _ in range(5)]
    fa = agent._fused_actor
    assert fa.use_graphs and len(fa.graphs) == 1
    ref = eager(obs)
    assert all(torch.equal(o, ref) for o in outs) and len({o.data_ptr() for o in outs}) == len(outs)
    for u in range(2, 8):                                 # training moves encoder, feature head and policy weights in place
        agent.update_parameters(mem, u)
    after = agent(obs, mode="eval")
    assert torch.equal(after, eager(obs)) and not torch.equal(after, ref)
    for _ in range(4):
        got5 = agent(obs5, mode="eval")
    assert len(fa.graphs) == 2 and torch.equal(got5, eager(obs5)) and torch.equal(agent(obs, mode="eval"), after)
    samples = [agent(obs, mode="explore") for _ in range(6)]
    assert len(fa.graphs) == 3 and all(not torch.equal(samples[i], samples[i + 1]) for i in range(3, 5))
    assert all(bool(torch.isfinite(s_).all()) and float(s_.abs().max()) <= 1.0 + 1e-6 for s_ in samples)


def test_jitter_shares_the_steps_device_counter(cuda):
    """Inside an update step fed by a device-sampling replay the jitter calls take the replay's draw counter as their Philox offset
    (no per-call counter launch) and differ by seed; outside a step every call advances its own counter as before."""
    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd.augmentations import RandomJitterPoints
    aug = RandomJitterPoints(main_key="xyz", req_keys=["xyz"], jitter_range=[-0.01, 0.01], seed=5)
    obs = {"xyz": torch.randn(4, 3, 64, device=cuda)}
    shared = torch.full((3,), 7, dtype=torch.int64, device=cuda)[:1]
    aug.begin_step(shared)
    a, b = aug(obs), aug(obs)
    assert a.aug["offset_tensor"].data_ptr() == shared.data_ptr() == b.aug["offset_tensor"].data_ptr()
    assert a.aug["seed"] != b.aug["seed"] and a.aug["seed"] != aug.seed
    xa = hip.augment_xyz(obs["xyz"], jitter_range=aug.jitter_range, seed=a.aug["seed"], offset=0, offset_tensor=shared)
    xb = hip.augment_xyz(obs["xyz"], jitter_range=aug.jitter_range, seed=b.aug["seed"], offset=0, offset_tensor=shared)
    assert not torch.equal(xa, xb)                                   # two calls of one step: independent noise
    shared += 1                                                      # the next step's sampling launch
    xa2 = hip.augment_xyz(obs["xyz"], jitter_range=aug.jitter_range, seed=a.aug["seed"], offset=0, offset_tensor=shared)
    assert not torch.equal(xa, xa2)                                  # the same (captured) call, next step: fresh noise
    aug.begin_step(None)
    c, d = aug(obs), aug(obs)
    assert c.aug["seed"] == aug.seed == d.aug["seed"]
    assert int(d.aug["offset_tensor"]) == int(c.aug["offset_tensor"]) + 1



def test_affine_sample_kernel_draws_the_reference_s_matrix_family(cuda):
    """pcrl_affine_sample_f32 (GlobalRotScaleTrans's draw as one launch): every matrix has the structure the reference builds
    (pcd_aug.py:178-196) -- diag(s) R_axis(angle) with angle / s_i inside their ranges, translation (u - 0.5) * 2 * range, zero for
    the LAST cloud unless shift_height, identity block without a rotation range -- the draws have the uniform law's first two
    moments, are reproducible for a fixed (seed, offset), change with either, and follow a device-side offset slot."""
    from pointcloud_rl_amd import hip
    B = 4096
    rot, sc, tr = [-0.15, 0.15], [0.95, 1.05], [0.1, 0.2, 0.3]
    draw = lambda **kw: hip.affine_sample(torch.empty(B, 3, 4, device=cuda), **dict(dict(rot_axis=2, rot_range=rot, scale_range=sc, translation_range=tr,
                                                                                        shift_height=False, seed=9, offset=0), **kw))
    m = draw()
    assert torch.equal(m, draw()) and not torch.equal(m, draw(offset=1)) and not torch.equal(m, draw(seed=10))
    off = torch.tensor([1], dtype=torch.int64, device=cuda)
    assert torch.equal(draw(offset_tensor=off), draw(offset=1))
    v = m.double().cpu().numpy()
    s = np.linalg.norm(v[:, :, :3], axis=2)                                   # row norms of diag(s) R = s_i
    assert s.min() >= 0.95 - 1e-6 and s.max() <= 1.05 + 1e-6
    R = v[:, :, :3] / s[:, :, None]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5
    assert np.abs(R[:, 2, 2] - 1).max() < 1e-6 and np.abs(R[:, 2, :2]).max() < 1e-6 and np.abs(R[:, :2, 2]).max() < 1e-6      # about z
    ang = np.arctan2(R[:, 1, 0], R[:, 0, 0])
    assert ang.min() >= -0.15 - 1e-6 and ang.max() <= 0.15 + 1e-6
    for x, lo, hi in [(ang, *rot)] + [(s[:, i], *sc) for i in range(3)] + [(v[:-1, i, 3], -tr[i], tr[i]) for i in range(3)]:
        w = hi - lo
        assert abs(x.mean() - (lo + hi) / 2) < 5 * (w / np.sqrt(12)) / np.sqrt(len(x))
        assert abs(x.var() - w * w / 12) < 0.08 * w * w / 12
    assert np.abs(np.corrcoef(np.stack([ang, s[:, 0], s[:, 1], s[:, 2], v[:, 0, 3]]))[np.triu_indices(5, 1)]).max() < 0.06
    assert np.all(v[-1, :, 3] == 0) and np.any(draw(shift_height=True)[-1, :, 3].cpu().numpy() != 0)       # delta_xyz[-1] = 0
    ident = draw(rot_range=None, translation_range=None)                      # the scale has nothing to act on, as in the reference
    assert torch.equal(ident[:, :, :3], torch.eye(3, device=cuda).expand(B, 3, 3)) and torch.all(ident[:, :, 3] == 0)
    # two draws in one launch (pcrl_affine_sample_pair_f32) = the two single launches
    second = torch.full((B, 3, 4), float("nan"), device=cuda)
    first = draw(second=(second, 10, 3))
    assert torch.equal(first, m) and torch.equal(second, draw(seed=10, offset=3))
    second.fill_(float("nan"))
    draw(offset_tensor=off, second=(second, 10, 77))                          # a device-side offset slot serves both draws
    assert torch.equal(second, draw(seed=10, offset=1))


def test_drq_jitter_plus_scale_step_matches_the_restatement_and_draws_fresh_matrices(cuda):
    """BASELINE config 3 as worded ("jitter+scale aug fused into encoder kernel"): DrQ with obs_aug = [GlobalRotScaleTrans(rotation +
    per-axis scale), RandomJitterPoints].  (a) With the matrices and the noise injected on both sides the fused HIP step equals the
    CPU restatement of the reference's step (apply_rot_trans, then the jitter).  (b) Fed by a device replay and replayed from a
    hipGraph, the matrices come from ONE launch per call keyed by the step's device counter: fresh every step, no ATen launches."""
    from oracle import torch_ref
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.augmentations import GlobalRotScaleTrans, RandomJitterPoints
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.replay import DeviceReplay
    from pointcloud_rl_amd.synthetic import make_batch_np
    B, N, A = 8, 96, 4
    cfg = configs.drq_dmc(6, A, B, head_hidden=64, obs_aug=[configs.ROT_SCALE, configs.JITTER])
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg)
    assert [type(t) for t in agent.obs_aug.transforms] == [GlobalRotScaleTrans, RandomJitterPoints]
    params = {n: p.detach().clone() for n, p in agent.named_parameters()}
    ref = torch_ref.RefAgent(params, kind="drq", gamma=agent.gamma, alpha=0.1, target_entropy=agent.target_entropy,
                             update_coeff=agent.update_coeff["default"], num_aug=2, mirror_redundancy=False)
    agent = agent.to(cuda)
    g = torch.Generator().manual_seed(4)
    rst = agent.obs_aug[0]

    class Mem:
        def __init__(self, b):
            self.b = b

        def sample(self, n):
            return self

        def to_torch(self, device=None, non_blocking=False):
            from pointcloud_rl_amd.utils.torch_utils import to_torch
            return to_torch(self.b, device=device)
    for u in (1, 2):
        batch_np = make_batch_np(B, N, A, seed=30 + u)
        cpu_batch = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v)) for k, v in batch_np.items()}
        eps = [torch.randn(2 * B, A, generator=g)] + ([torch.randn(B, A, generator=g)] if u % 2 == 0 else [])
        jit = [torch.empty(2 * B, 3, N).uniform_(-0.01, 0.01, generator=g) for _ in range(2)]
        aff = [rst.sample_matrix(2 * B, "cpu") for _ in range(2)]
        assert all(float(m[:, :, 3].abs().max()) == 0 for m in aff)                      # translation_range=None
        agent.actor.head.noise_override = [e.to(cuda) for e in eps]
        agent.obs_aug[0].matrix_override = [m.to(cuda) for m in aff]
        agent.obs_aug[1].noise_override = [j.to(cuda) for j in jit]
        got = agent.update_parameters(Mem(batch_np), u)
        assert agent._fused is not None
        want = ref.update_parameters(cpu_batch, u, eps, jit, affine_list=aff)
        for k, v in want.items():
            assert abs(got[k] - v) <= 5e-5 * max(1.0, abs(v)), (u, k, got[k], v)
    for n, p in agent.named_parameters():
        err = (p.detach().cpu() - ref.P[n].detach()).abs()
        assert (err <= 1e-5).float().mean() >= 0.999 and err.max() <= 2e-4, (n, float(err.max()))
    # (b) device replay + hipGraph: one affine_sample launch per augmentation call, fresh matrices on every replay
    mem = DeviceReplay(64, device=cuda, seed=3)
    mem.push_batch(make_batch_np(64, N, A, seed=40))
    agent.enable_graphs(warmup=1)
    seen, losses = [], []
    for u in range(3, 13):
        losses.append(agent.update_parameters(mem, u)["drq/critic_loss"])
        torch.cuda.synchronize()
        seen.append(torch.stack([m.clone() for m in rst._mats.values()]))
    assert len(rst._mats) == 2 and agent._graphs                                        # obs and next_obs: two persistent buffers
    for a, b in zip(seen[3:-1], seen[4:]):
        assert not torch.equal(a, b)                                                     # replayed launches drew again
    assert not torch.equal(seen[-1][0], seen[-1][1])                                     # the two calls of a step differ
    # the first call's launch draws for both calls of the step (pair_draws = False: a launch per call): same matrices, same steps
    def replayed_losses(pair):
        torch.manual_seed(0)
        ag = build_agent(cfg).to(cuda)
        ag.obs_aug[0].pair_draws = pair
        mem2 = DeviceReplay(64, device=cuda, seed=3)
        mem2.push_batch(make_batch_np(64, N, A, seed=40))
        ag.enable_graphs(warmup=1)
        out = [ag.update_parameters(mem2, u) for u in range(1, 9)]
        torch.cuda.synchronize()
        return out, [m.clone() for m in ag.obs_aug[0]._mats.values()]
    (la, ma), (lb, mb) = replayed_losses(True), replayed_losses(False)
    assert la == lb and len(ma) == len(mb) == 2 and all(torch.equal(x, y) for x, y in zip(ma, mb))
    assert np.isfinite(losses).all() and len(set(np.round(losses, 7))) == len(losses)
