"""GPU parity: HIP encoder backward (C ABI) vs autograd through the PyTorch-CPU restatement."""
import numpy as np
import pytest
import torch

from helpers import make_encoder_weights, make_obs

pytestmark = pytest.mark.gpu

NAMES = {"conv0.weight": "w0", "conv0.bias": "b0", "conv1.weight": "w1", "norm1.weight": "g1", "norm1.bias": "be1",
         "conv2.weight": "w2", "norm2.weight": "g2", "norm2.bias": "be2"}


def torch_reference_grads(obs_np, w_np, gpool_np, jitter=None, route=None):
    """route: int argmax [B, c3] to send the gradient through instead of torch's own max (a 1-ulp near-tie between two
    points is decided by the summation order; see profiles/r02_parity_errors.md)."""
    from oracle import torch_ref
    P = {}
    for ref_name, k in NAMES.items():
        t = torch.from_numpy(np.ascontiguousarray(w_np[k])).clone()
        if t.ndim == 2:
            t = t[..., None]
        P[torch_ref.ENC + "conv.mlp." + ref_name] = t.requires_grad_(True)
    obs = {k: torch.from_numpy(v) for k, v in obs_np.items()}
    if jitter is not None:
        obs["xyz"] = obs["xyz"] + torch.from_numpy(jitter)
    pre = torch_ref.pointnet_prepool(P, obs)
    pooled, idx = pre.max(-1)
    routed = pooled if route is None else pre.gather(-1, torch.from_numpy(route).long()[..., None])[..., 0]
    (routed * torch.from_numpy(gpool_np)).sum().backward()
    return {n: P[torch_ref.ENC + "conv.mlp." + n].grad.numpy() for n in NAMES}, idx.numpy().astype(np.int32), pooled.detach().numpy()


def hip_grads(obs_np, w_np, gpool_np, dev, jitter=None, with_pooled=True):
    from pointcloud_rl_amd import hip
    wt = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in w_np.items()}
    ew, keep_w = hip.make_encoder_weights(wt["w0"], wt["b0"], wt["w1"], wt["g1"], wt["be1"], wt["w2"], wt["g2"], wt["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=dev)
    hip.encoder_pack_weights(ew, packed)
    obs = {k: torch.from_numpy(v).to(dev) for k, v in obs_np.items()}
    desc, keep = hip.make_cloud_desc(obs)
    aug = None
    if jitter is not None:
        jt = torch.from_numpy(jitter).to(dev)
        aug = hip.make_aug_desc(jitter_noise=jt)
    pooled, argmax = hip.encoder_fwd(desc, ew, packed, aug=aug)
    flat, n_act = hip.encoder_bwd(desc, ew, packed, argmax, torch.from_numpy(gpool_np).to(dev), aug=aug, want_n_active=True,
                                  pooled=pooled if with_pooled else None)
    views = {k: v.cpu().numpy() for k, v in hip.encoder_grad_views(flat, ew).items()}
    torch.cuda.synchronize()
    return views, argmax.cpu().numpy(), pooled.cpu().numpy(), n_act.cpu().numpy()


def assert_grads_close(got, ref):
    for name in NAMES:
        g, r = got[name].reshape(-1), ref[name].reshape(-1)
        scale = max(np.abs(r).max(), 1e-6)
        err = np.abs(g - r).max() / scale
        assert err < 2e-5, f"{name}: max rel-to-max err {err:.3e} (scale {scale:.3e})"


@pytest.fixture(params=["auto", "launches", "team"])
def bwd_path(request):
    """The Gram-form backward's two launch schedules (include/pcrl.h: pcrl_encoder_bwd_set_fused): by size (the default), the points / wgrad /
    reduce launches for every size, the team kernel for every size it is built for."""
    from pointcloud_rl_amd import hip
    hip.encoder_bwd_set_fused({"auto": 1, "launches": 0, "team": 2}[request.param])
    yield request.param
    hip.encoder_bwd_set_fused(1)


@pytest.mark.parametrize("B,N,extra,c1", [
    (3, 64, dict(), 64),                 # n_active <= 64: two tiles per cloud at most
    (2, 400, dict(), 64),                # many active points, several waves busy
    (4, 37, dict(), 64),                 # ragged N < 64
    (2, 300, dict(pos_encoding=3), 64),  # C = 9
    (3, 250, dict(seg=1), 128),          # ManiSkill nets: C = 7, c1 = 128
    (1, 1, dict(), 64),                  # a single point owns every channel
    (3, 260, dict(), 32),                # mlp_spec [32, 64, 128] (pn_motivating configs): tile mode (B < #CUs)
    (300, 140, dict(), 32),              # the same nets, one workgroup per cloud (B >= #CUs)
    (260, 130, dict(seg=1), 128),        # cloud mode with the ManiSkill nets
])
def test_bwd_matches_torch_autograd(cuda, bwd_path, B, N, extra, c1):
    obs = make_obs(B, N, seed=17 * B + N, **extra)
    C = sum(v.shape[1] for v in obs.values())
    c2, c3 = (64, 128) if c1 == 32 else (128, 256)
    w = make_encoder_weights(C, c1, c2, c3, seed=N + 1)
    gpool = np.random.RandomState(N).randn(B, c3).astype(np.float32)
    ref, idx_ref, pooled_ref = torch_reference_grads(obs, w, gpool)
    got, idx, pooled, n_act = hip_grads(obs, w, gpool, cuda)
    np.testing.assert_allclose(pooled, pooled_ref, atol=1e-5, rtol=0)
    if not np.array_equal(idx, idx_ref):
        # ATen's summation order decided a near-tie the other way (a few of B x c3 entries at the larger sizes): the two
        # candidates must hold the same maximum to rounding; the gradient is then compared along this kernel's routing
        assert (idx != idx_ref).mean() < 1e-3
        ref, _, _ = torch_reference_grads(obs, w, gpool, route=idx)
    # n_active = the points that receive gradient: the distinct argmax points of the LIVE channels (a channel the forward left at
    # zero passes none; the round-2 kernels -- a call without the forward's pooled values -- count the points of all channels)
    live_pts = [len(np.unique(r[m])) for r, m in zip(idx, pooled > 0)]
    assert np.array_equal(n_act, live_pts) or np.array_equal(n_act, [len(np.unique(r)) for r in idx])
    assert_grads_close(got, ref)
    # the same without the forward's pooled values (dense search for the owned channels)
    got_dense, _, _, _ = hip_grads(obs, w, gpool, cuda, with_pooled=False)
    assert_grads_close(got_dense, ref)


def test_bwd_schedule_follows_the_launch_size(cuda):
    """include/pcrl.h: pcrl_encoder_bwd_set_fused(1), the default -- the team kernel for launches of at most two 32-point tiles per CU (8 B <=
    2 x #CUs possible tiles), the points / wgrad / reduce launches beyond; a call without the forward's pooled values keeps the round-2 kernels;
    the shapes the team kernel is not built for (c2 = 64) keep the launches."""
    from pointcloud_rl_amd import hip
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    small, large = max(1, n_cu // 4), n_cu // 4 + 1
    seen = {}
    for B, c1 in ((small, 64), (large, 64), (3, 32)):
        obs = make_obs(B, 70, seed=B)
        c2, c3 = (64, 128) if c1 == 32 else (128, 256)
        w = make_encoder_weights(6, c1, c2, c3, seed=1)
        gpool = np.random.RandomState(B).randn(B, c3).astype(np.float32)
        hip_grads(obs, w, gpool, cuda)
        seen[(B, c1)] = hip.encoder_bwd_last_schedule()
    assert seen == {(small, 64): 3, (large, 64): 2, (3, 32): 2}, seen
    hip_grads(make_obs(2, 70, seed=1), make_encoder_weights(6, 64, 128, 256, seed=1), np.zeros((2, 256), np.float32), cuda, with_pooled=False)
    assert hip.encoder_bwd_last_schedule() == 1
    for mode, want in ((0, 2), (2, 3), (1, 3)):
        hip.encoder_bwd_set_fused(mode)
        hip_grads(make_obs(2, 70, seed=1), make_encoder_weights(6, 64, 128, 256, seed=1), np.zeros((2, 256), np.float32), cuda)
        assert hip.encoder_bwd_last_schedule() == want, (mode, want)


def test_bwd_with_jitter_noise(cuda, bwd_path):
    obs = make_obs(3, 200, seed=5)
    w = make_encoder_weights(6, 64, 128, 256, seed=9)
    gpool = np.random.RandomState(1).randn(3, 256).astype(np.float32)
    jitter = np.random.RandomState(2).uniform(-0.01, 0.01, (3, 3, 200)).astype(np.float32)
    ref, idx_ref, _ = torch_reference_grads(obs, w, gpool, jitter)
    got, idx, _, _ = hip_grads(obs, w, gpool, cuda, jitter)
    assert np.array_equal(idx, idx_ref)
    assert_grads_close(got, ref)


def test_bwd_is_deterministic_and_linear(cuda, bwd_path):
    # size-independent properties at the K1 launch geometry: bitwise reproducible, linear in grad_pooled
    obs = make_obs(256, 1024, seed=1)
    w = make_encoder_weights(6, 64, 128, 256, seed=0)
    g1 = np.random.RandomState(3).randn(256, 256).astype(np.float32)
    g2 = np.random.RandomState(4).randn(256, 256).astype(np.float32)
    a, _, _, n_act = hip_grads(obs, w, g1, cuda)
    a2, _, _, _ = hip_grads(obs, w, g1, cuda)
    b, _, _, _ = hip_grads(obs, w, g2, cuda)
    c, _, _, _ = hip_grads(obs, w, (g1 + g2), cuda)
    assert (n_act >= 1).all() and (n_act <= 256).all()
    for name in NAMES:
        assert np.array_equal(a[name], a2[name]), name
        scale = np.abs(c[name]).max()
        assert np.abs(a[name] + b[name] - c[name]).max() <= 2e-5 * scale, name
    # a slice of the batch against the oracle (per-cloud partial sums are independent)
    sel = slice(0, 3)
    obs_s = {k: v[sel] for k, v in obs.items()}
    ref, _, _ = torch_reference_grads(obs_s, w, g1[sel])
    got, _, _, _ = hip_grads(obs_s, w, g1[sel], cuda)
    assert_grads_close(got, ref)


@pytest.mark.parametrize("B,N,extra,c1,c3", [(5, 300, dict(), 64, 256), (130, 200, dict(seg=1), 128, 256), (3, 150, dict(), 64, 1024)])
def test_bwd_in_two_calls_equals_one_call(cuda, bwd_path, B, N, extra, c1, c3):
    """pcrl_encoder_bwd_prepare_f32 (on another stream, before grad_pooled exists) + pcrl_encoder_bwd_prepared_f32 ==
    pcrl_encoder_bwd_f32, bit for bit; the prepare call refuses what the Gram form does not cover."""
    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd._lib import PcrlError
    obs_np = make_obs(B, N, seed=3 * B + N, **extra)
    C = sum(v.shape[1] for v in obs_np.values())
    w_np = make_encoder_weights(C, c1, 128, c3, seed=N)
    wt = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cuda) for k, v in w_np.items()}
    ew, keep_w = hip.make_encoder_weights(wt["w0"], wt["b0"], wt["w1"], wt["g1"], wt["be1"], wt["w2"], wt["g2"], wt["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    obs = {k: torch.from_numpy(v).to(cuda) for k, v in obs_np.items()}
    desc, keep = hip.make_cloud_desc(obs)
    pooled, argmax = hip.encoder_fwd(desc, ew, packed)
    gpool = torch.from_numpy(np.random.RandomState(N).randn(B, c3).astype(np.float32)).to(cuda)
    one, n_one = hip.encoder_bwd(desc, ew, packed, argmax, gpool, want_n_active=True, pooled=pooled)
    import ctypes
    need = ctypes.c_size_t()
    hip.check(hip.lib().pcrl_encoder_bwd_workspace_bytes(B, ew.c_in, ew.c1, ew.c2, ew.c3, ctypes.byref(need)))
    ws = torch.zeros(need.value, dtype=torch.uint8, device=cuda)
    side, main = torch.cuda.Stream(device=cuda), torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        hip.encoder_bwd_prepare(desc, ew, packed, argmax, pooled, ws)
    main.wait_stream(side)
    two, n_two = hip.encoder_bwd(desc, ew, packed, argmax, gpool, want_n_active=True, pooled=pooled, workspace=ws, prepared=True)
    torch.cuda.synchronize()
    assert torch.equal(one, two) and torch.equal(n_one, n_two)
    with pytest.raises(PcrlError):          # the round-2 kernels (no pooled values) have no two-call form
        hip.check(hip.lib().pcrl_encoder_bwd_prepare_f32(ctypes.byref(desc), None, ctypes.byref(ew), hip._ptr(packed), hip._ptr(argmax), None,
                                                         hip._ptr(ws), ctypes.c_size_t(ws.numel()), hip._stream()))
    with pytest.raises(PcrlError):
        hip.check(hip.lib().pcrl_encoder_bwd_prepared_f32(ctypes.byref(desc), None, ctypes.byref(ew), hip._ptr(packed), hip._ptr(argmax),
                                                          hip._ptr(gpool), None, hip._ptr(two), None, hip._ptr(ws), ctypes.c_size_t(ws.numel()),
                                                          hip._stream()))


def _bf16_reference_grads(obs_np, w_np, gpool_np, eps=1e-6):
    """Autograd through the rounding emulation of the mixed-precision forward, roundings straight-through."""
    import torch.nn.functional as F
    from oracle import c_oracle
    x = torch.from_numpy(c_oracle.preprocess(obs_np))
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(True) for k, v in w_np.items()}
    st = lambda a: a + (a.to(torch.bfloat16).to(torch.float32) - a).detach()
    h0 = F.relu(torch.einsum("oc,bcn->bon", t["w0"], x) + t["b0"][None, :, None])
    z1 = torch.einsum("oc,bcn->bon", st(t["w1"]), st(h0))
    h1 = F.relu(F.layer_norm(z1.permute(0, 2, 1), (z1.shape[1],), t["g1"], t["be1"], eps).permute(0, 2, 1))
    z2 = torch.einsum("oc,bcn->bon", st(t["w2"]), st(h1))
    h2 = F.relu(F.layer_norm(z2.permute(0, 2, 1), (z2.shape[1],), t["g2"], t["be2"], eps).permute(0, 2, 1))
    return t, h2


@pytest.mark.parametrize("B,N,extra,c1", [(3, 200, dict(), 64), (2, 1200, dict(seg=1), 128)])
def test_bwd_bf16_matches_autograd_of_the_rounding_emulation(cuda, bwd_path, B, N, extra, c1):
    """Mixed-precision backward: gradients w.r.t. the fp32 master weights of the function the bf16 forward computes.
    The kernel's own argmax is used to pick the pooled points of the emulation (a rounding tie may move an argmax, which
    is a different -- equally valid -- subgradient).  Tolerance: 3e-2 of each tensor's largest gradient entry."""
    from pointcloud_rl_amd import hip
    obs = make_obs(B, N, seed=23, **extra)
    C = sum(v.shape[1] for v in obs.values())
    w = make_encoder_weights(C, c1, 128, 256, seed=6)
    gpool = np.random.RandomState(N).randn(B, 256).astype(np.float32)
    wt = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cuda) for k, v in w.items()}
    ew, keep_w = hip.make_encoder_weights(wt["w0"], wt["b0"], wt["w1"], wt["g1"], wt["be1"], wt["w2"], wt["g2"], wt["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    desc, keep = hip.make_cloud_desc({k: torch.from_numpy(v).to(cuda) for k, v in obs.items()})
    pooled, argmax = hip.encoder_fwd(desc, ew, packed, bf16=True)
    flat = hip.encoder_bwd(desc, ew, packed, argmax, torch.from_numpy(gpool).to(cuda), bf16=True)
    got = {k: v.cpu().numpy() for k, v in hip.encoder_grad_views(flat, ew).items()}
    t, h2 = _bf16_reference_grads(obs, w, gpool)
    picked = torch.gather(h2, 2, argmax.cpu().long()[:, :, None])[:, :, 0]
    np.testing.assert_allclose(pooled.cpu().numpy(), picked.detach().numpy(), atol=3e-2, rtol=0)
    (picked * torch.from_numpy(gpool)).sum().backward()
    for name, k in NAMES.items():
        g, r = got[name].reshape(-1), t[k].grad.numpy().reshape(-1)
        scale = max(np.abs(r).max(), 1e-6)
        assert np.abs(g - r).max() / scale < 3e-2, f"{name}: {np.abs(g - r).max() / scale:.3e}"
    # bitwise reproducible
    assert torch.equal(flat, hip.encoder_bwd(desc, ew, packed, argmax, torch.from_numpy(gpool).to(cuda), bf16=True))
    # With the forward's pooled values (what the agents pass) the bf16 mode runs the fp32 Gram-form backward at the bf16 forward's
    # routing: the fp32 gradient along that argmax (a call without pooled values keeps the kernels checked above)
    if True:
        flat_g = hip.encoder_bwd(desc, ew, packed, argmax, torch.from_numpy(gpool).to(cuda), bf16=True, pooled=pooled)
        got_g = {k: v.cpu().numpy() for k, v in hip.encoder_grad_views(flat_g, ew).items()}
        ref32, _, _ = torch_reference_grads(obs, w, gpool, route=argmax.cpu().numpy())
        for name in NAMES:
            g, r = got_g[name].reshape(-1), ref32[name].reshape(-1)
            scale = max(np.abs(r).max(), 1e-6)
            assert np.abs(g - r).max() / scale < 2e-4, f"{name}: {np.abs(g - r).max() / scale:.3e} against fp32 autograd at the bf16 routing"


def test_bwd_zero_gamma_falls_back_to_the_dense_path(cuda):
    """norm2.weight == 0 on some channels: xhat cannot be recovered from pooled there, the kernel must take the dense path for
    the cloud and still match autograd."""
    obs = make_obs(2, 150, seed=31)
    w = make_encoder_weights(6, 64, 128, 256, seed=7)
    w["g2"] = w["g2"].copy(); w["g2"][::5] = 0.0
    w["be2"] = np.abs(w["be2"]) + 0.1                     # y = beta > 0 on those channels: they are live
    gpool = np.random.RandomState(5).randn(2, 256).astype(np.float32)
    ref, idx_ref, _ = torch_reference_grads(obs, w, gpool)
    got, idx, _, _ = hip_grads(obs, w, gpool, cuda)
    assert np.array_equal(idx, idx_ref)
    assert_grads_close(got, ref)


def test_bwd_cloud_whose_channels_are_all_dead(cuda, bwd_path):
    """norm2.bias far below zero on every channel: the forward leaves every channel at zero (argmax = point 0, torch's first index),
    no point receives gradient -- every gradient is exactly zero, n_active = 0 -- and a batch that mixes such a cloud (a cloud of
    identical points has zero variance, LayerNorm outputs = beta) with ordinary ones still matches autograd."""
    obs = make_obs(3, 90, seed=51)
    w = make_encoder_weights(6, 64, 128, 256, seed=8)
    w_dead = dict(w, be2=np.full(256, -10.0, np.float32))
    gpool = np.random.RandomState(7).randn(3, 256).astype(np.float32)
    got, idx, pooled, n_act = hip_grads(obs, w_dead, gpool, cuda)
    assert (pooled == 0).all() and (idx == 0).all() and (n_act == 0).all()
    for name, g in got.items():
        assert np.isfinite(g).all() and (g == 0).all(), name
    # one degenerate cloud among ordinary ones: all its points identical -> xhat = 0 everywhere -> y = beta per channel
    obs2 = {k: v.copy() for k, v in obs.items()}
    for k in obs2:
        obs2[k][1] = obs2[k][1][:, :1]
    ref, idx_ref, _ = torch_reference_grads(obs2, w, gpool)
    got2, idx2, _, _ = hip_grads(obs2, w, gpool, cuda)
    assert np.array_equal(idx2, idx_ref) and (idx2[1] == 0).all()
    assert_grads_close(got2, ref)


def test_bwd_small_gamma_large_beta_stays_accurate(cuda):
    """norm2 with |beta| >> |gamma| (xhat = (y - beta) / gamma would lose its digits by cancellation) and with tiny gamma:
    the per-channel shortcut must not be taken; gradients still match autograd at the usual tolerance."""
    obs = make_obs(3, 200, seed=41)
    w = make_encoder_weights(6, 64, 128, 256, seed=9)
    g2, be2 = w["g2"].copy(), w["be2"].copy()
    g2[::3] = 1e-2 * np.sign(g2[::3])
    be2[::3] = 5.0                                         # y ~ 5 +- 1e-2 * xhat: live, cancellation if reconstructed
    g2[1::7] = 2e-4
    w["g2"], w["be2"] = g2, be2
    gpool = np.random.RandomState(6).randn(3, 256).astype(np.float32)
    ref, idx_ref, _ = torch_reference_grads(obs, w, gpool)
    got, idx, _, _ = hip_grads(obs, w, gpool, cuda)
    # exact ties are possible on nearly-constant channels: compare gradients only where argmax agrees everywhere
    assert np.array_equal(idx, idx_ref)
    assert_grads_close(got, ref)


@pytest.mark.parametrize("offset,tol", [(2.0, 2e-5), (20.0, 5e-4)])
def test_bwd_conv2_columns_with_a_large_common_component(cuda, bwd_path, offset, tol):
    """LayerNorm-2's variance in the Gram form comes from h1 . Mc h1 with the CENTRED Gram image Mc = M - s s^T / C3 (round 4; the
    round-3 form E z^2 - mu^2 cancelled when every column of W2 carries a large common offset: z2's channel mean is then >> its
    spread).  Reference: float64 autograd along the HIP forward's routing; the bound is the usual 2e-5 of each tensor's largest entry
    at offset 2 and 5e-4 at offset 20, where fp32's own z - mean cancellation (forward and backward alike, |z| ~ 600 against a spread
    of ~1) is what is left -- the uncentred variance had an ABSOLUTE error of ~0.05 there, i.e. rstd2 off by several per cent."""
    from oracle import torch_ref
    B, N = 4, 260
    obs = make_obs(B, N, seed=91)
    w = make_encoder_weights(6, 64, 128, 256, seed=13)
    rng = np.random.RandomState(3)
    w["w2"] = (w["w2"] + offset * rng.choice([-1.0, 1.0], size=(1, w["w2"].shape[1]))).astype(np.float32)   # column j shifted by +-offset
    gpool = rng.randn(B, 256).astype(np.float32)
    got, argmax, pooled, _ = hip_grads(obs, w, gpool, cuda)
    P = {}
    for ref_name, k in NAMES.items():
        t = torch.from_numpy(np.ascontiguousarray(w[k])).double()
        P[torch_ref.ENC + "conv.mlp." + ref_name] = (t[..., None] if t.ndim == 2 else t).requires_grad_(True)
    obs_t = {k: torch.from_numpy(v) for k, v in obs.items()}
    pre = torch_ref.pointnet_prepool(P, {k: (v.double() if v.dtype == torch.float32 else v) for k, v in obs_t.items()})
    routed = pre.gather(-1, torch.from_numpy(argmax).long()[..., None])[..., 0]
    assert float((pre.max(-1)[0] - routed).abs().max()) < 50 * tol      # the routing is the maximum up to fp32 rounding of this ill-conditioned layer
    (routed * torch.from_numpy(gpool).double()).sum().backward()
    ref = {n: P[torch_ref.ENC + "conv.mlp." + n].grad.float().numpy() for n in NAMES}
    worst = {n: float(np.abs(got[n].reshape(-1) - ref[n].reshape(-1)).max() / max(np.abs(ref[n]).max(), 1e-6)) for n in NAMES}
    print(f"offset {offset}: worst rel-to-max error per tensor", {k: f"{v:.2e}" for k, v in worst.items()})
    assert max(worst.values()) < tol, worst


@pytest.mark.parametrize("B,N,extra,c1", [(3, 260, dict(), 64), (4, 300, dict(seg=1), 128), (2, 200, dict(), 32), (260, 120, dict(), 64)])
def test_split_precision_backward_matches_torch_autograd(cuda, B, N, extra, c1):
    """EXPERIMENTAL pcrl_encoder_{fwd,bwd}_f32split: gradients against fp32 autograd of the restatement at the exact kernel's
    tolerance (2e-5 of each tensor's largest entry); bitwise reproducible."""
    from pointcloud_rl_amd import hip
    obs = make_obs(B, N, seed=3 * B + N, **extra)
    C = sum(v.shape[1] for v in obs.values())
    c2, c3 = (64, 128) if c1 == 32 else (128, 256)
    w = make_encoder_weights(C, c1, c2, c3, seed=N + 2)
    gpool = np.random.RandomState(N).randn(B, c3).astype(np.float32)
    wt = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cuda) for k, v in w.items()}
    ew, keep_w = hip.make_encoder_weights(wt["w0"], wt["b0"], wt["w1"], wt["g1"], wt["be1"], wt["w2"], wt["g2"], wt["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    desc, keep = hip.make_cloud_desc({k: torch.from_numpy(v).to(cuda) for k, v in obs.items()})
    pooled, argmax = hip.encoder_fwd(desc, ew, packed, split=True)
    gp = torch.from_numpy(gpool).to(cuda)
    flat = hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=pooled, split=True)
    ref, idx_ref, pooled_ref = torch_reference_grads(obs, w, gpool, route=argmax.cpu().numpy())
    np.testing.assert_allclose(pooled.cpu().numpy(), pooled_ref, atol=1e-5, rtol=0)
    assert (argmax.cpu().numpy() != idx_ref).mean() < 1e-3
    got = {k: v.cpu().numpy() for k, v in hip.encoder_grad_views(flat, ew).items()}
    assert_grads_close(got, ref)
    assert torch.equal(flat, hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=pooled, split=True))


@pytest.mark.parametrize("B,N,extra", [(3, 90, dict()), (2, 1500, dict()), (2, 40, dict(pos_encoding=3)), (260, 60, dict())])
def test_bwd_wide_class_default_mlp_spec_matches_torch_autograd(cuda, B, N, extra):
    """mlp_spec = [64, 128, 1024] (PointNet's class default, pointnet.py:81): the Gram-form backward needs only W2^T W2 [128 x 128]
    whatever c3 is; up to 1 024 gradient-carrying points per cloud (32 tiles), two-byte slots.  Same tolerance as the narrow shapes."""
    obs = make_obs(B, N, seed=31 * B + N, **extra)
    C = sum(v.shape[1] for v in obs.values())
    w = make_encoder_weights(C, 64, 128, 1024, seed=N + 5)
    gpool = np.random.RandomState(N).randn(B, 1024).astype(np.float32)
    ref, idx_ref, pooled_ref = torch_reference_grads(obs, w, gpool)
    got, idx, pooled, n_act = hip_grads(obs, w, gpool, cuda)
    np.testing.assert_allclose(pooled, pooled_ref, atol=1e-5, rtol=0)
    if not np.array_equal(idx, idx_ref):
        assert (idx != idx_ref).mean() < 1e-3
        ref, _, _ = torch_reference_grads(obs, w, gpool, route=idx)
    assert np.array_equal(n_act, [len(np.unique(r[m])) for r, m in zip(idx, pooled > 0)])
    assert_grads_close(got, ref)
    again, _, _, _ = hip_grads(obs, w, gpool, cuda)
    for name in NAMES:
        assert np.array_equal(got[name], again[name]), name          # bitwise run to run
