"""GPU parity: fused HIP encoder forward (through the C ABI) vs the CPU oracle, bit for bit."""
import numpy as np
import pytest
import torch

from helpers import make_encoder_weights, make_obs

pytestmark = pytest.mark.gpu


def _run_hip(obs_np, w_np, dev, eps=1e-6, interleaved=False, aug=None, bf16=False):
    from pointcloud_rl_amd import hip
    wt = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in w_np.items()}
    ew, keep_w = hip.make_encoder_weights(wt["w0"], wt["b0"], wt["w1"], wt["g1"], wt["be1"], wt["w2"], wt["g2"], wt["be2"], eps)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=dev)
    hip.encoder_pack_weights(ew, packed)
    if interleaved:
        pts = torch.from_numpy(obs_np).to(dev)
        desc, keep = hip.make_interleaved_desc(pts)
    else:
        obs = {k: torch.from_numpy(v).to(dev) for k, v in obs_np.items()}
        desc, keep = hip.make_cloud_desc(obs)
    pooled, argmax = hip.encoder_fwd(desc, ew, packed, aug=aug, bf16=bf16)
    torch.cuda.synchronize()
    return pooled.cpu().numpy(), argmax.cpu().numpy()


def _check(obs, w, dev, **kw):
    from oracle import c_oracle
    pooled_ref, arg_ref = c_oracle.encoder_fwd(c_oracle.preprocess(obs), w)
    pooled, argmax = _run_hip(obs, w, dev, **kw)
    assert np.array_equal(argmax, arg_ref), f"argmax mismatches: {(argmax != arg_ref).sum()} of {arg_ref.size}"
    assert np.array_equal(pooled.view(np.uint32), pooled_ref.view(np.uint32)), \
        f"max abs diff {np.abs(pooled - pooled_ref).max()}"


@pytest.mark.parametrize("B,N,C_extra,c1", [
    (4, 64, dict(), 64),                       # one partial workgroup, full tiles
    (3, 100, dict(), 64),                      # ragged N (last tile partly masked)
    (2, 1, dict(), 64),                        # single point per cloud
    (5, 257, dict(pos_encoding=3), 64),        # K0 layout: C = 9
    (4, 200, dict(seg=1), 128),                # ManiSkill nets: C = 7, c1 = 128
    (2, 96, dict(rgb=False), 64),              # xyz only (C = 3 -> one zero-padded k-step)
    (3, 300, dict(), 32),                      # mlp_spec [32, 64, 128] (configs/mfrl/sac/dm_control/pn_motivating.py:29)
    (2, 1100, dict(pos_encoding=3), 32),       # the same nets, C = 9, more than 8 tiles per wave
])
def test_fwd_matches_oracle_small(cuda, B, N, C_extra, c1):
    obs = make_obs(B, N, seed=B * 1000 + N, **C_extra)
    C = sum(v.shape[1] for v in obs.values())
    c2, c3 = (64, 128) if c1 == 32 else (128, 256)
    w = make_encoder_weights(C, c1, c2, c3, seed=N)
    _check(obs, w, cuda)


def test_fwd_split_cloud_two_stage(cuda):
    # B < #CUs with many tiles: clouds are split over several workgroups and merged by the second stage
    obs = make_obs(2, 2048 + 17, seed=7)
    w = make_encoder_weights(6, 64, 128, 256, seed=3)
    _check(obs, w, cuda)


def test_fwd_k1_shape_batch_slice(cuda):
    # K1 launch geometry (B=256, N=1024) -- the oracle checks a slice of the batch to stay within seconds
    from oracle import c_oracle
    obs = make_obs(256, 1024, seed=1)
    w = make_encoder_weights(6, 64, 128, 256, seed=0)
    pooled, argmax = _run_hip(obs, w, cuda)
    sel = [0, 1, 127, 255]
    sub = {k: v[sel] for k, v in obs.items()}
    pooled_ref, arg_ref = c_oracle.encoder_fwd(c_oracle.preprocess(sub), w)
    assert np.array_equal(argmax[sel], arg_ref)
    assert np.array_equal(pooled[sel].view(np.uint32), pooled_ref.view(np.uint32))
    # size-independent property: permuting the points of a cloud permutes argmax and keeps pooled
    perm = np.random.RandomState(5).permutation(1024)
    obs_p = {k: np.ascontiguousarray(v[:, :, perm]) for k, v in obs.items()}
    pooled_p, argmax_p = _run_hip(obs_p, w, cuda)
    assert np.array_equal(pooled_p.view(np.uint32), pooled.view(np.uint32))
    gap = perm[argmax_p] != argmax          # may differ only where the maximum is attained twice (exact ties)
    assert gap.mean() < 0.2


def test_fwd_ties_dead_channels_and_duplicates(cuda):
    from oracle import c_oracle
    obs = make_obs(3, 160, seed=11)
    # duplicate points: exact ties everywhere -> first index must win
    for k in obs:
        obs[k][:, :, 80:] = obs[k][:, :, :80]
    w = make_encoder_weights(6, 64, 128, 256, seed=4)
    w["g2"][:17] = 0.0
    w["be2"][:17] = -1.0                          # ReLU-dead channels: all zeros -> index 0
    pooled_ref, arg_ref = c_oracle.encoder_fwd(c_oracle.preprocess(obs), w)
    assert (arg_ref < 80).all() and (arg_ref[:, :17] == 0).all() and (pooled_ref[:, :17] == 0).all()
    pooled, argmax = _run_hip(obs, w, cuda)
    assert np.array_equal(argmax, arg_ref)
    assert np.array_equal(pooled.view(np.uint32), pooled_ref.view(np.uint32))


def test_fwd_nan_point_wins(cuda):
    from oracle import c_oracle
    obs = make_obs(2, 70, seed=13)
    obs["xyz"][1, 0, 37] = np.nan
    obs["xyz"][1, 2, 50] = np.nan
    w = make_encoder_weights(6, 64, 128, 256, seed=5)
    pooled_ref, arg_ref = c_oracle.encoder_fwd(c_oracle.preprocess(obs), w)
    assert np.isnan(pooled_ref[1]).all() and (arg_ref[1] == 37).all()
    pooled, argmax = _run_hip(obs, w, cuda)
    assert np.array_equal(argmax, arg_ref)
    assert np.isnan(pooled[1]).all()
    assert np.array_equal(pooled[0].view(np.uint32), pooled_ref[0].view(np.uint32))


@pytest.mark.parametrize("B,N,C_extra", [(3, 100, dict()), (2, 1, dict()), (260, 70, dict()), (4, 300, dict(pos_encoding=3)), (2, 2100, dict())])
def test_fwd_wide_class_default_mlp_spec_matches_oracle(cuda, B, N, C_extra):
    """mlp_spec = [64, 128, 1024], PointNet's class default (pointnet.py:81): the wide kernel (three conv2 passes in 256-channel chunks)
    is bit-identical to the C oracle -- values and first-index argmax -- like the narrow shapes; ragged N, a single point, B above the
    number of CUs, C = 9, clouds split over workgroups (second-stage merge)."""
    obs = make_obs(B, N, seed=B * 100 + N, **C_extra)
    C = sum(v.shape[1] for v in obs.values())
    w = make_encoder_weights(C, 64, 128, 1024, seed=N + 3)
    _check(obs, w, cuda)


def test_fwd_wide_ties_dead_channels_and_nan(cuda):
    from oracle import c_oracle
    obs = make_obs(3, 96, seed=23)
    for k in obs:
        obs[k][:, :, 48:] = obs[k][:, :, :48]          # duplicate points: exact ties -> first index
    obs["xyz"][2, 1, 20] = np.nan                      # a NaN point wins every channel of its cloud
    w = make_encoder_weights(6, 64, 128, 1024, seed=6)
    w["g2"][100:140] = 0.0
    w["be2"][100:140] = -1.0                           # ReLU-dead channels: value 0, index 0
    pooled_ref, arg_ref = c_oracle.encoder_fwd(c_oracle.preprocess(obs), w)
    pooled, argmax = _run_hip(obs, w, cuda)
    assert np.array_equal(argmax, arg_ref)
    assert (arg_ref[:2] < 48).all() and (pooled_ref[:2, 100:140] == 0).all() and np.isnan(pooled[2]).all() and (argmax[2] == 20).all()
    assert np.array_equal(pooled[:2].view(np.uint32), pooled_ref[:2].view(np.uint32))


def test_fwd_interleaved_bnc_layout(cuda):
    # BASELINE.json's synthetic [B, N, C] f32 layout, read through strides
    from oracle import c_oracle
    g = np.random.RandomState(3)
    pts = g.randn(6, 300, 6).astype(np.float32)
    w = make_encoder_weights(6, 64, 128, 256, seed=8)
    pooled_ref, arg_ref = c_oracle.encoder_fwd(np.ascontiguousarray(pts.transpose(0, 2, 1)), w)
    pooled, argmax = _run_hip(pts, w, cuda, interleaved=True)
    assert np.array_equal(argmax, arg_ref)
    assert np.array_equal(pooled.view(np.uint32), pooled_ref.view(np.uint32))


def test_fwd_bad_arguments_raise(cuda):
    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd._lib import PcrlError
    with pytest.raises(PcrlError):
        hip.encoder_packed_bytes(6, 64, 128, 512)       # not one of the built (c1, c2, c3) shapes
    with pytest.raises(PcrlError):
        hip.encoder_packed_bytes(6, 128, 128, 1024)     # the wide last layer is built for the class default only
    assert hip.encoder_packed_bytes(6, 64, 128, 1024) > 0


@pytest.mark.parametrize("name,B,N,extra,c1", [
    ("K0 dmc_walker_walk: 3 frames x 512 points, xyz+rgb+pos_encoding", 4, 1536, dict(pos_encoding=3), 64),
    ("K2/K3 ManiSkill MoveBucket: N=1200, xyz+rgb+seg, nets [128,128,256], 128 clouds per GPU", 128, 1200, dict(seg=1), 128),
    ("K4 large-N stress: N=8192, 64 clouds per GPU, two-stage pool", 64, 8192, dict(), 64),
])
def test_fwd_baseline_config_shapes(cuda, name, B, N, extra, c1):
    """BASELINE.json configs 1, 3/4 and 5 at their full per-GPU sizes: a slice of the batch against the oracle (bit-exact),
    and the whole batch through size-independent properties (batch order invariance, point-permutation invariance)."""
    from oracle import c_oracle
    obs = make_obs(B, N, seed=11, **extra)
    C = sum(v.shape[1] for v in obs.values())
    w = make_encoder_weights(C, c1, 128, 256, seed=2)
    pooled, argmax = _run_hip(obs, w, cuda)
    sel = sorted({0, B // 2, B - 1})
    sub = {k: v[sel] for k, v in obs.items()}
    pooled_ref, arg_ref = c_oracle.encoder_fwd(c_oracle.preprocess(sub), w)
    assert np.array_equal(argmax[sel], arg_ref), name
    assert np.array_equal(pooled[sel].view(np.uint32), pooled_ref.view(np.uint32)), name
    assert argmax.min() >= 0 and argmax.max() < N
    # clouds are independent: reversing the batch reverses the outputs bit for bit
    rev = {k: np.ascontiguousarray(v[::-1]) for k, v in obs.items()}
    pooled_r, argmax_r = _run_hip(rev, w, cuda)
    assert np.array_equal(pooled_r[::-1].view(np.uint32), pooled.view(np.uint32)) and np.array_equal(argmax_r[::-1], argmax)
    # the pool is symmetric: permuting the points keeps the pooled values bit for bit
    perm = np.random.RandomState(3).permutation(N)
    obs_p = {k: np.ascontiguousarray(v[:, :, perm]) for k, v in obs.items()}
    pooled_p, argmax_p = _run_hip(obs_p, w, cuda)
    assert np.array_equal(pooled_p.view(np.uint32), pooled.view(np.uint32))
    assert argmax_p.min() >= 0 and argmax_p.max() < N


def _bf16_reference(obs, w, eps=1e-6):
    """torch emulation of pcrl_encoder_fwd_bf16's rounding points: bf16 weights and bf16 layer inputs for conv1 / conv2,
    fp32 accumulation and fp32 everything else."""
    import torch
    import torch.nn.functional as F
    from oracle import c_oracle
    x = torch.from_numpy(c_oracle.preprocess(obs))                      # [B, C, N]
    t = {k: torch.from_numpy(v) for k, v in w.items()}
    bf = lambda a: a.to(torch.bfloat16).to(torch.float32)
    h0 = F.relu(torch.einsum("oc,bcn->bon", t["w0"], x) + t["b0"][None, :, None])
    z1 = torch.einsum("oc,bcn->bon", bf(t["w1"]), bf(h0))
    h1 = F.relu(F.layer_norm(z1.permute(0, 2, 1), (z1.shape[1],), t["g1"], t["be1"], eps).permute(0, 2, 1))
    z2 = torch.einsum("oc,bcn->bon", bf(t["w2"]), bf(h1))
    h2 = F.relu(F.layer_norm(z2.permute(0, 2, 1), (z2.shape[1],), t["g2"], t["be2"], eps).permute(0, 2, 1))
    val, idx = h2.max(-1)
    return val.numpy(), idx.numpy(), h2.numpy()


@pytest.mark.parametrize("B,N,extra,c1", [(6, 300, dict(), 64), (4, 1200, dict(seg=1), 128), (2, 4100, dict(), 64)])
def test_fwd_bf16_matches_rounding_emulation(cuda, B, N, extra, c1):
    """Mixed-precision forward (BASELINE config 3).  Tolerances: |pooled - emulation| <= 3e-2 (outputs are O(1) LayerNorm
    values; a bf16 rounding tie that falls the other way moves one product by 2^-8 relative), argmax equal for >= 95 % of
    the channels and, where it differs, the emulation's value at the reported point is within 3e-2 of its maximum."""
    obs = make_obs(B, N, seed=17, **extra)
    C = sum(v.shape[1] for v in obs.values())
    w = make_encoder_weights(C, c1, 128, 256, seed=5)
    pooled, argmax = _run_hip(obs, w, cuda, bf16=True)
    val, idx, h2 = _bf16_reference(obs, w)
    np.testing.assert_allclose(pooled, val, atol=3e-2, rtol=0)
    assert (argmax == idx).mean() >= 0.95
    at_reported = np.take_along_axis(h2, argmax[:, :, None].astype(np.int64), axis=2)[:, :, 0]
    assert np.abs(at_reported - val).max() <= 3e-2
    # and it is a different function from the fp32 kernel only by rounding
    pooled32, _ = _run_hip(obs, w, cuda)
    assert 1e-5 < np.abs(pooled - pooled32).max() < 0.15


def test_virtual_repeat_equals_materialised_repeat(cuda):
    """pcrl_cloud_desc.row_div (DrQ's repeat without the copies): cloud b reads stored cloud b // 2 with its own jitter row --
    pooled, argmax and the backward's gradients are bitwise those of the repeat_interleave'd batch."""
    import torch
    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd.networks.pointnet import AugmentedObs
    B, N, rep = 5, 300, 2
    obs_np = make_obs(B, N, seed=21, seg=1)
    w = {k: torch.from_numpy(v).to(cuda) for k, v in make_encoder_weights(7, 128, 128, 256, seed=4).items()}
    ew, _ = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    obs = {k: torch.from_numpy(v).to(cuda) for k, v in obs_np.items()}
    noise = torch.empty(B * rep, 3, N, device=cuda).uniform_(-0.01, 0.01)
    gp = torch.randn(B * rep, 256, device=cuda)

    def run(o):
        desc, keep = hip.make_cloud_desc(o)
        aug = hip.make_aug_desc(jitter_noise=noise)
        pooled, argmax = hip.encoder_fwd(desc, ew, packed, aug=aug)
        grads = hip.encoder_bwd(desc, ew, packed, argmax, gp, aug=aug, pooled=pooled)
        return pooled, argmax, grads

    virt = AugmentedObs(obs)
    virt.repeat = rep
    mat = {k: torch.repeat_interleave(v, rep, dim=0) for k, v in obs.items()}
    a, b = run(virt), run(mat)
    assert a[0].shape == (B * rep, 256)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert not torch.equal(a[0][0], a[0][1])          # the two augmentations of a sample differ (their jitter rows do)


def _fixture_launch(cuda, fixture, **mode):
    import os
    from pointcloud_rl_amd import hip
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture + ".npz"))
    obs = {k[4:]: torch.from_numpy(d[k]).to(cuda) for k in d.files if k.startswith("obs/")}
    g = lambda k: torch.from_numpy(np.ascontiguousarray(d["w/" + k])).to(cuda)
    w0, w1, w2 = g("conv.mlp.conv0.weight")[..., 0].contiguous(), g("conv.mlp.conv1.weight")[..., 0].contiguous(), g("conv.mlp.conv2.weight")[..., 0].contiguous()
    ew, keep = hip.make_encoder_weights(w0, g("conv.mlp.conv0.bias"), w1, g("conv.mlp.norm1.weight"), g("conv.mlp.norm1.bias"), w2,
                                        g("conv.mlp.norm2.weight"), g("conv.mlp.norm2.bias"), 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    desc, keep2 = hip.make_cloud_desc(obs)
    pooled, argmax = hip.encoder_fwd(desc, ew, packed, **mode)
    torch.cuda.synchronize()
    return d, pooled.cpu().numpy(), argmax.cpu().numpy()


@pytest.mark.parametrize("fixture", ["encoder_dmc_c6", "encoder_dmc_c9_posenc", "encoder_maniskill_c7", "encoder_dmc_motivating_c6",
                                     "encoder_classdefault_c6"])
def test_fwd_f32_on_the_reference_fixtures(cuda, fixture):
    """The DEFAULT exact-fp32 kernel (pcrl_encoder_fwd_f32) directly against what the reference itself computed (fixtures captured
    from /root/reference by tools/gen_golden.py: pointnet.py:148-151 `self.conv(feature)` + `feature.max(-1)` on torch CPU): argmax
    bit-exact on every (cloud, channel), pooled values within north_star's 1e-5."""
    d, pooled, argmax = _fixture_launch(cuda, fixture)
    assert np.array_equal(argmax, d["argmax"]), f"{(argmax != d['argmax']).sum()} of {argmax.size} argmax entries differ"
    np.testing.assert_allclose(pooled, d["pooled"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("fixture", ["encoder_dmc_c6", "encoder_dmc_c9_posenc", "encoder_maniskill_c7", "encoder_dmc_motivating_c6"])
def test_split_precision_forward_on_the_reference_fixtures(cuda, fixture):
    """EXPERIMENTAL pcrl_encoder_fwd_f32split (three-term bf16 split of the fp32 contractions) on the encoder fixtures captured
    from the reference: pooled within 1e-5 of the reference's, argmax exact (the fixtures' smallest top-2 gap is > 1e-5)."""
    import os
    import torch
    from pointcloud_rl_amd import hip
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture + ".npz"))
    obs = {k[4:]: torch.from_numpy(d[k]).to(cuda) for k in d.files if k.startswith("obs/")}
    g = lambda k: torch.from_numpy(np.ascontiguousarray(d["w/" + k])).to(cuda)
    w0, w1, w2 = g("conv.mlp.conv0.weight")[..., 0].contiguous(), g("conv.mlp.conv1.weight")[..., 0].contiguous(), g("conv.mlp.conv2.weight")[..., 0].contiguous()
    ew, keep = hip.make_encoder_weights(w0, g("conv.mlp.conv0.bias"), w1, g("conv.mlp.norm1.weight"), g("conv.mlp.norm1.bias"), w2,
                                        g("conv.mlp.norm2.weight"), g("conv.mlp.norm2.bias"), 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    desc, keep2 = hip.make_cloud_desc(obs)
    pooled, argmax = hip.encoder_fwd(desc, ew, packed, split=True)
    assert np.array_equal(argmax.cpu().numpy(), d["argmax"])
    np.testing.assert_allclose(pooled.cpu().numpy(), d["pooled"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("fixture", ["encoder_dmc_c6", "encoder_dmc_c9_posenc", "encoder_maniskill_c7", "encoder_dmc_motivating_c6"])
def test_bf16_forward_on_the_reference_fixtures(cuda, fixture):
    """Mixed-precision forward (BASELINE config 3) against the fp32 REFERENCE's own outputs (fixtures captured from it): bf16
    operands with fp32 accumulation cannot match fp32 to 1e-5; what it does is measured here and bounded -- max |pooled - reference|
    1.1e-2 ... 1.7e-2 on values of magnitude ~4 (0.2-0.4 %), mean 2e-3, argmax equal on 98.9-99.6 % of the channels (a max-pool
    picks between points whose fp32 values differ by less than the bf16 rounding) -- with 1.5x margin."""
    import os
    import torch
    from pointcloud_rl_amd import hip
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture + ".npz"))
    obs = {k[4:]: torch.from_numpy(d[k]).to(cuda) for k in d.files if k.startswith("obs/")}
    g = lambda k: torch.from_numpy(np.ascontiguousarray(d["w/" + k])).to(cuda)
    w0, w1, w2 = g("conv.mlp.conv0.weight")[..., 0].contiguous(), g("conv.mlp.conv1.weight")[..., 0].contiguous(), g("conv.mlp.conv2.weight")[..., 0].contiguous()
    ew, keep = hip.make_encoder_weights(w0, g("conv.mlp.conv0.bias"), w1, g("conv.mlp.norm1.weight"), g("conv.mlp.norm1.bias"), w2,
                                        g("conv.mlp.norm2.weight"), g("conv.mlp.norm2.bias"), 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    desc, keep2 = hip.make_cloud_desc(obs)
    pooled, argmax = hip.encoder_fwd(desc, ew, packed, bf16=True)
    diff = np.abs(pooled.cpu().numpy() - d["pooled"])
    assert diff.max() <= 2.5e-2 and diff.mean() <= 4e-3, (diff.max(), diff.mean())
    assert (argmax.cpu().numpy() == d["argmax"]).mean() >= 0.98


def test_split_precision_forward_k1_shape_against_the_exact_kernel(cuda):
    """K1 shape (256 x 1024): |pooled - exact fp32| <= 1e-5 everywhere; argmax may differ only where the two candidates' values
    are within 1e-6 (measured: 0 of 65 536 entries on this data, max |diff| 3.1e-6)."""
    import torch
    from pointcloud_rl_amd import hip
    obs_np = make_obs(256, 1024, seed=1)
    w = {k: torch.from_numpy(v).to(cuda) for k, v in make_encoder_weights(6, 64, 128, 256, seed=0).items()}
    ew, _ = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    obs = {k: torch.from_numpy(v).to(cuda) for k, v in obs_np.items()}
    desc, keep = hip.make_cloud_desc(obs)
    p0, a0 = hip.encoder_fwd(desc, ew, packed)
    p1, a1 = hip.encoder_fwd(desc, ew, packed, split=True)
    diff = (p0 - p1).abs()
    assert float(diff.max()) <= 1e-5
    moved = a0 != a1
    assert int(moved.sum()) <= 8 and (not moved.any() or float(diff[moved].max()) <= 1e-6)
    p2, a2 = hip.encoder_fwd(desc, ew, packed, split=True)
    assert torch.equal(p1, p2) and torch.equal(a1, a2)              # deterministic


@pytest.mark.parametrize("B,N,c1,F,S", [
    (6, 2048, 64, 50, 0),         # B < #CUs: clouds split over workgroups -> the merge launch carries the head
    (300, 96, 64, 50, 5),         # B > #CUs: one workgroup per cloud, several clouds per workgroup; robot-state columns passed through
    (40, 200, 128, 128, 9),       # ManiSkill nets: F = 128 (two features per lane in the LayerNorm)
    (9, 64, 32, 50, 0),           # mlp_spec [32, 64, 128]
])
@pytest.mark.parametrize("mode", ["f32", "bf16", "f32split"])
def test_feature_head_epilogue_matches_linear_layernorm(cuda, B, N, c1, F, S, mode):
    """pcrl_encoder_fwd_head_*: PointNet.final_mlp (Linear + LayerNorm, pointnet.py:152-153) applied by the encoder launch to two
    ranges of its clouds, with destinations at column offsets, saved xhat / rstd and pass-through columns -- against torch's
    F.linear + F.layer_norm of the launch's own pooled output (1e-5: the dot products sum in a different order than torch's)."""
    import torch.nn.functional as Fn
    from pointcloud_rl_amd import hip
    c2, c3 = (64, 128) if c1 == 32 else (128, 256)
    obs_np = make_obs(B, N, seed=B + N)
    C = sum(v.shape[1] for v in obs_np.values())
    wt = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cuda) for k, v in make_encoder_weights(C, c1, c2, c3).items()}
    ew, _ = hip.make_encoder_weights(wt["w0"], wt["b0"], wt["w1"], wt["g1"], wt["be1"], wt["w2"], wt["g2"], wt["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    desc, keep = hip.make_cloud_desc({k: torch.from_numpy(v).to(cuda) for k, v in obs_np.items()})
    g = torch.Generator().manual_seed(B)
    W = (torch.randn(F, c3, generator=g) / c3 ** 0.5).to(cuda)
    b, gam, bet = (torch.randn(F, generator=g).to(cuda) * 0.1, (1 + 0.3 * torch.randn(F, generator=g)).to(cuda), torch.randn(F, generator=g).to(cuda) * 0.2)
    B0 = B // 3                                         # range 0 = clouds [0, B0), range 1 = clouds [B0, B)
    ld0, ld1 = F + S + 3, F + 2
    X0, X0b, X1 = torch.zeros(B0, ld0, device=cuda), torch.zeros(B0, ld1, device=cuda), torch.zeros(B - B0, ld0, device=cuda)
    xhat1, rstd1 = torch.zeros(B - B0, F, device=cuda), torch.zeros(B - B0, device=cuda)
    state = torch.randn(B - B0, max(S, 1), generator=g).to(cuda)
    jobs = [(0, dict(M=B0, dsts=[(X0, 1, ld0), (X0b, 2, ld1)])),
            (B0, dict(M=B - B0, dsts=[(X1, 0, ld0)], xhat=xhat1, rstd=rstd1, cats=[(state, X1, F, ld0)] if S else []))]
    head = hip.make_feature_head(W, b, gam, bet, F, 1e-5, jobs)
    pooled, argmax = hip.encoder_fwd(desc, ew, packed, bf16=mode == "bf16", split=mode == "f32split", head=head)
    plain, plain_arg = hip.encoder_fwd(desc, ew, packed, bf16=mode == "bf16", split=mode == "f32split")
    assert torch.equal(pooled, plain) and torch.equal(argmax, plain_arg)             # the encoder's own outputs are untouched
    y = Fn.linear(pooled, W, b)
    ref = Fn.layer_norm(y, (F,), gam, bet, 1e-5)
    assert (X0[:, 1:1 + F] - ref[:B0]).abs().max() <= 1e-5 and (X0b[:, 2:2 + F] - ref[:B0]).abs().max() <= 1e-5
    assert (X1[:, :F] - ref[B0:]).abs().max() <= 1e-5
    assert float(X0[:, 0].abs().max()) == 0.0 and float(X0[:, 1 + F:].abs().max()) == 0.0      # nothing outside the destination columns
    mean, var = y[B0:].mean(-1, keepdim=True), y[B0:].var(-1, unbiased=False, keepdim=True)
    assert (xhat1 - (y[B0:] - mean) / torch.sqrt(var + 1e-5)).abs().max() <= 1e-5
    assert (rstd1 - 1 / torch.sqrt(var[:, 0] + 1e-5)).abs().max() <= 1e-4 * float(rstd1.abs().max())
    if S:
        assert torch.equal(X1[:, F:F + S], state) and float(X1[:, F + S:].abs().max()) == 0.0
