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
])
def test_fwd_matches_oracle_small(cuda, B, N, C_extra, c1):
    obs = make_obs(B, N, seed=B * 1000 + N, **C_extra)
    C = sum(v.shape[1] for v in obs.values())
    w = make_encoder_weights(C, c1, 128, 256, seed=N)
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
        hip.encoder_packed_bytes(6, 64, 128, 1024)      # c3 = 1024 not supported by the fused kernel


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
