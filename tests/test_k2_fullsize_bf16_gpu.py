"""BASELINE config 3 (K2) at its FULL launch geometry -- 256 samples x 2 augmentations = 512 clouds of N = 1200 points, C = 7,
nets [128, 128, 256], bf16 conv1 / conv2 with fp32 accumulation, jitter fused into the load -- through size-independent
properties (the rounding emulation of tests/test_encoder_{fwd,bwd}_gpu.py is checked on a slice; 512 x 1200 x 256 activations do
not fit a CPU test):
  * batch-order invariance: a cloud's result does not depend on which workgroup / wave / tile position computed it;
  * point-permutation invariance: the pooled VALUES are bitwise those of the permuted cloud, the argmax follows the permutation;
  * the virtual repeat (row_div) with per-row jitter equals the materialised repeat;
  * a slice of the launch equals the same clouds launched alone, and that slice is within the emulation's tolerance;
  * backward at the same geometry: bitwise reproducible, linear in the upstream gradient, equal on a slice to the slice's own
    launch (per-cloud partial sums are independent); there it is the fp32 gradient at the bf16 forward's routing (2e-4) and within
    8e-2 of autograd through the emulation."""
import numpy as np
import pytest
import torch

from helpers import make_encoder_weights, make_obs
from test_encoder_bwd_gpu import NAMES, _bf16_reference_grads
from test_encoder_fwd_gpu import _bf16_reference

pytestmark = pytest.mark.gpu

B, REP, N, C1 = 256, 2, 1200, 128


def _setup(cuda):
    from pointcloud_rl_amd import hip
    obs_np = make_obs(B, N, seed=77, seg=1)
    w_np = make_encoder_weights(7, C1, 128, 256, seed=8)
    w = {k: torch.from_numpy(v).to(cuda) for k, v in w_np.items()}
    ew, keep = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    noise = torch.empty(B * REP, 3, N, device=cuda).uniform_(-0.01, 0.01, generator=torch.Generator(device=cuda).manual_seed(3))
    return hip, obs_np, w_np, ew, packed, noise, (keep, w)


def _fwd(hip, obs, ew, packed, noise, repeat=1):
    from pointcloud_rl_amd.networks.pointnet import AugmentedObs
    o = AugmentedObs(obs)
    o.repeat = repeat
    desc, keep = hip.make_cloud_desc(o)
    aug = hip.make_aug_desc(jitter_noise=noise) if noise is not None else None
    pooled, argmax = hip.encoder_fwd(desc, ew, packed, aug=aug, bf16=True)
    return pooled, argmax, desc, aug, keep


def test_k2_forward_full_size_properties(cuda):
    hip, obs_np, w_np, ew, packed, noise, keep_w = _setup(cuda)
    obs = {k: torch.from_numpy(v).to(cuda) for k, v in obs_np.items()}
    pooled, argmax, *_ = _fwd(hip, obs, ew, packed, noise, repeat=REP)            # the K2 launch: 512 virtual clouds
    assert pooled.shape == (B * REP, 256) and bool(torch.isfinite(pooled).all())
    assert int(argmax.min()) >= 0 and int(argmax.max()) < N
    # the materialised repeat (what the reference does, drq.py:52-60) gives the same bits
    mat = {k: torch.repeat_interleave(v, REP, dim=0) for k, v in obs.items()}
    p_mat, a_mat, *_ = _fwd(hip, mat, ew, packed, noise)
    assert torch.equal(pooled, p_mat) and torch.equal(argmax, a_mat)
    # batch order: reverse the clouds (every cloud lands on another workgroup / launch position)
    rev = {k: v.flip(0).contiguous() for k, v in mat.items()}
    p_rev, a_rev, *_ = _fwd(hip, rev, ew, packed, noise.flip(0).contiguous())
    assert torch.equal(p_rev.flip(0), pooled) and torch.equal(a_rev.flip(0), argmax)
    # point permutation: same values bitwise; the reported point is the permuted position of a point holding the maximum
    perm = torch.from_numpy(np.random.RandomState(4).permutation(N)).to(cuda)
    per = {k: v[:, :, perm].contiguous() for k, v in mat.items()}
    p_per, a_per, *_ = _fwd(hip, per, ew, packed, noise[:, :, perm].contiguous())
    assert torch.equal(p_per, pooled)
    # ... except on exact ties, where the first index wins in either order: ReLU-dead channels (every point 0 -> index 0) and
    # channels whose maximum is attained by two points
    same_point = (perm[a_per.long()] == argmax.long()) | (pooled == 0)
    assert float(same_point.float().mean()) >= 0.999
    # a slice of the launch == the slice launched alone, and the slice is within the rounding emulation's tolerance
    sel = slice(40, 46)
    sl = {k: v[sel].contiguous() for k, v in mat.items()}
    p_sl, a_sl, *_ = _fwd(hip, sl, ew, packed, noise[sel].contiguous())
    assert torch.equal(p_sl, pooled[sel]) and torch.equal(a_sl, argmax[sel])
    obs_sl = {k: np.repeat(v, REP, axis=0)[sel] for k, v in obs_np.items()}
    obs_sl["xyz"] = obs_sl["xyz"] + noise[sel].cpu().numpy()
    val, idx, h2 = _bf16_reference(obs_sl, w_np)
    np.testing.assert_allclose(p_sl.cpu().numpy(), val, atol=3e-2, rtol=0)
    assert (a_sl.cpu().numpy() == idx).mean() >= 0.95


def test_k2_backward_full_size_properties(cuda):
    hip, obs_np, w_np, ew, packed, noise, keep_w = _setup(cuda)
    obs = {k: torch.from_numpy(v).to(cuda) for k, v in obs_np.items()}
    pooled, argmax, desc, aug, keep = _fwd(hip, obs, ew, packed, noise, repeat=REP)
    g = torch.Generator(device=cuda).manual_seed(9)
    g1, g2 = (torch.randn(B * REP, 256, device=cuda, generator=g) for _ in range(2))
    run = lambda gp: hip.encoder_bwd(desc, ew, packed, argmax, gp, aug=aug, pooled=pooled, bf16=True).clone()
    a, a_again, b, c = run(g1), run(g1), run(g2), run(g1 + g2)
    assert torch.equal(a, a_again)                                              # fixed-order reductions: bitwise run to run
    assert bool(torch.isfinite(a).all())
    for name, v in hip.encoder_grad_views(c, ew).items():
        lin = hip.encoder_grad_views(a, ew)[name] + hip.encoder_grad_views(b, ew)[name]
        # linear in the upstream gradient up to the bf16 roundings inside the data-gradient GEMMs (the round-2 kernels round the
        # upstream gradient as it enters W2^T dz2; the Gram form has no such rounding and is linear to fp32 accuracy)
        assert float((lin - v).abs().max()) <= 2e-2 * float(v.abs().max()), name
    # a slice: its own launch, against autograd through the rounding emulation
    sel = slice(100, 104)
    mat = {k: torch.repeat_interleave(v, REP, dim=0)[sel].contiguous() for k, v in obs.items()}
    p_sl, a_sl, d_sl, aug_sl, keep_sl = _fwd(hip, mat, ew, packed, noise[sel].contiguous())
    assert torch.equal(p_sl, pooled[sel]) and torch.equal(a_sl, argmax[sel])
    flat = hip.encoder_bwd(d_sl, ew, packed, a_sl, g1[sel].contiguous(), aug=aug_sl, pooled=p_sl, bf16=True)
    got = {k: v.cpu().numpy() for k, v in hip.encoder_grad_views(flat, ew).items()}
    obs_sl = {k: np.repeat(v, REP, axis=0)[sel] for k, v in obs_np.items()}
    obs_sl["xyz"] = obs_sl["xyz"] + noise[sel].cpu().numpy()
    t, h2 = _bf16_reference_grads(obs_sl, w_np, None)
    picked = torch.gather(h2, 2, a_sl.cpu().long()[:, :, None])[:, :, 0]
    (picked * g1[sel].cpu()).sum().backward()
    # The bf16 mode's backward is (given the forward's pooled values) the fp32 Gram-form kernel at the bf16 forward's routing (argmax,
    # pooled > 0): the EXACT fp32 gradient at that routing (measured < 5e-5 of each tensor's largest entry), which differs from autograd
    # through the rounding emulation by what bf16 operands change in the two recomputed layers (measured up to 5.8e-2, on norm1.bias /
    # conv0.bias).  The round-2 bf16 kernels (a call without pooled values) are the mirror image: within 3e-3 of the emulation, up to 5.8e-2
    # from the fp32 gradient.
    from test_encoder_bwd_gpu import torch_reference_grads
    gram = True
    ref32, _, _ = torch_reference_grads(obs_sl, w_np, g1[sel].cpu().numpy(), route=a_sl.cpu().numpy())
    worst = {}
    for name, k in NAMES.items():
        r = t[k].grad.numpy().reshape(-1)
        err = np.abs(got[name].reshape(-1) - r).max() / max(np.abs(r).max(), 1e-6)
        r32 = ref32[name].reshape(-1)
        err32 = np.abs(got[name].reshape(-1) - r32).max() / max(np.abs(r32).max(), 1e-6)
        worst[name] = (round(float(err), 5), round(float(err32), 5))
    print("K2 slice, encoder gradient error (vs bf16 emulation, vs fp32 at the same routing):", worst)
    for name, (err, err32) in worst.items():
        if gram:
            assert err32 < 2e-4, f"{name}: {err32:.3e} of the largest entry against fp32 autograd at the same routing"
            assert err < 8e-2, f"{name}: {err:.3e} of the largest entry against autograd through the bf16 emulation"
        else:
            assert err < 3e-2, f"{name}: {err:.3e} of the largest entry against autograd through the bf16 emulation"


def test_k2_step_as_benched_against_the_fp32_restatement(cuda):
    """What bench.py times as `config3_k2`, asserted as a STEP at its full size: DrQ, 256 samples x 2 augmentations = 512 clouds of
    N = 1200, C = 7, nets [128, 128, 256], bf16 conv1 / conv2, obs_aug = [GlobalRotScaleTrans(rotation + per-axis scale),
    RandomJitterPoints], ManiSkill heads (S = 68, A = 22) -- against the CPU restatement of the reference's fp32 DrQ step on the SAME
    256 transitions with the matrices, the jitter noise and the policy noise injected on both sides.  Bounds: those of
    test_update_step_gpu.py::test_drq_mixed_precision_step_against_the_reference_fixture (losses / Q statistics 1e-2 relative, floor 1;
    gradient norms 6 %) -- what bf16 operands in two of three encoder layers do to an fp32 step."""
    import bench
    from oracle import torch_ref
    from pointcloud_rl_amd.augmentations import GlobalRotScaleTrans, RandomJitterPoints
    from pointcloud_rl_amd.synthetic import make_batch_np
    wl = bench.WORKLOADS["k2"]
    Bs, Np, A, S = wl["B"], wl["N"], wl["A"], wl["S"]
    assert (Bs, Np, wl["aug"]) == (256, 1200, "rot_scale+jitter")
    agent, C = bench.build_agent(wl, Bs, torch.device("cpu"))
    assert C == 7 and agent.encoder.compute_dtype == "bf16"
    assert [type(t) for t in agent.obs_aug.transforms] == [GlobalRotScaleTrans, RandomJitterPoints]
    params = {n: p.detach().clone() for n, p in agent.named_parameters()}
    ref = torch_ref.RefAgent(params, kind="drq", gamma=agent.gamma, reward_scale=agent.reward_scale, alpha=0.1, target_entropy=agent.target_entropy,
                             update_coeff=agent.update_coeff["default"], num_aug=agent.num_aug, mirror_redundancy=False)
    agent = agent.to(cuda)
    rst = agent.obs_aug[0]
    g = torch.Generator().manual_seed(12)

    class Mem:
        def __init__(self, b):
            self.b = b

        def sample(self, n):
            return self

        def to_torch(self, device=None, non_blocking=False):
            from pointcloud_rl_amd.utils.torch_utils import to_torch
            return to_torch(self.b, device=device)
    worst, worst_g = 0.0, 0.0
    for u in (1, 2):
        batch_np = make_batch_np(Bs, Np, A, seed=50 + u, agent=S, **wl["obs_kw"])
        cpu_batch = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v)) for k, v in batch_np.items()}
        eps = [torch.randn(2 * Bs, A, generator=g)] + ([torch.randn(Bs, A, generator=g)] if u % 2 == 0 else [])
        jit = [torch.empty(2 * Bs, 3, Np).uniform_(-0.01, 0.01, generator=g) for _ in range(2)]
        aff = [rst.sample_matrix(2 * Bs, "cpu") for _ in range(2)]
        agent.actor.head.noise_override = [e.to(cuda) for e in eps]
        rst.matrix_override = [m.to(cuda) for m in aff]
        agent.obs_aug[1].noise_override = [j.to(cuda) for j in jit]
        got = agent.update_parameters(Mem(batch_np), u)
        assert agent._fused is not None, "the fused HIP step is what the bench times"
        want = ref.update_parameters(cpu_batch, u, eps, jit, affine_list=aff)
        assert got.keys() == want.keys()
        for k, v in want.items():
            if k.endswith("_grad"):
                worst_g = max(worst_g, abs(got[k] - v) / max(abs(v), 1e-6))
            else:
                worst = max(worst, abs(got[k] - v) / max(1.0, abs(v)))
    print(f"K2 step as benched vs the fp32 restatement: worst metric {worst:.2e}, worst gradient norm {worst_g:.2e}")
    assert worst <= 1e-2 and worst_g <= 0.06, (worst, worst_g)


@pytest.mark.parametrize("tag,Bc,Nc,C,c1", [("config4 one GPU", 1024, 1200, 7, 128), ("config5 one GPU", 512, 8192, 6, 64)])
def test_one_gpu_launch_geometries_the_bench_times_are_batch_order_invariant(cuda, tag, Bc, Nc, C, c1):
    """bench.py's `config4_k3` (1 024 clouds of 1 200 points, C = 7) and `config5_k4` (512 x 8 192, C = 6) time launch geometries that the
    parity tests cover at one rank's share (128 / 64 clouds); at the full one-GPU geometry the fp32 forward and backward are checked
    through what does not depend on the size: a cloud's pooled values and argmax are bitwise the same wherever it sits in the batch
    (reversed order: another workgroup, another position in the persistent loop), a slice launched alone gives the same bits, and
    the backward is bitwise reproducible and equals, restricted to a slice's clouds, the slice's own backward summed in cloud order
    only up to the fixed-order reduce (compared to 1e-5 of each tensor's largest entry)."""
    from pointcloud_rl_amd import hip
    obs_np = make_obs(Bc, Nc, seed=5, seg=C - 6)
    obs = {k: torch.from_numpy(v).to(cuda) for k, v in obs_np.items()}
    w_np = make_encoder_weights(C, c1, 128, 256, seed=9)
    w = {k: torch.from_numpy(v).to(cuda) for k, v in w_np.items()}
    ew, keep = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)

    def fwd(o):
        desc, k = hip.make_cloud_desc(o)
        pooled, argmax = hip.encoder_fwd(desc, ew, packed)
        return pooled, argmax, desc, k
    pooled, argmax, desc, k0 = fwd(obs)
    assert pooled.shape == (Bc, 256) and bool(torch.isfinite(pooled).all()) and int(argmax.max()) < Nc
    rev = {k: v.flip(0).contiguous() for k, v in obs.items()}
    p_rev, a_rev, d_rev, k1 = fwd(rev)
    assert torch.equal(p_rev.flip(0), pooled) and torch.equal(a_rev.flip(0), argmax)
    sel = slice(Bc // 2 - 3, Bc // 2 + 5)
    sl = {k: v[sel].contiguous() for k, v in obs.items()}
    p_sl, a_sl, d_sl, k2 = fwd(sl)
    assert torch.equal(p_sl, pooled[sel]) and torch.equal(a_sl, argmax[sel])
    gp = torch.randn(Bc, 256, device=cuda, generator=torch.Generator(device=cuda).manual_seed(2))
    run = lambda: hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=pooled).clone()
    ga, gb = run(), run()
    assert torch.equal(ga, gb) and bool(torch.isfinite(ga).all())
    # gradient of the reversed batch with the reversed upstream gradient: the same sum over clouds in another order
    g_rev = hip.encoder_bwd(d_rev, ew, packed, a_rev, gp.flip(0).contiguous(), pooled=p_rev)
    for name, v in hip.encoder_grad_views(ga, ew).items():
        r = hip.encoder_grad_views(g_rev, ew)[name]
        assert float((v - r).abs().max()) <= 1e-5 * float(v.abs().max()), (tag, name)
    # a slice's own backward == the full launch's with every other cloud's upstream gradient zeroed
    gp0 = torch.zeros_like(gp)
    gp0[sel] = gp[sel]
    g_full0 = hip.encoder_bwd(desc, ew, packed, argmax, gp0, pooled=pooled).clone()
    g_slice = hip.encoder_bwd(d_sl, ew, packed, a_sl, gp[sel].contiguous(), pooled=p_sl)
    for name, v in hip.encoder_grad_views(g_slice, ew).items():
        r = hip.encoder_grad_views(g_full0, ew)[name]
        assert float((v - r).abs().max()) <= 1e-5 * max(float(v.abs().max()), 1e-12), (tag, name)
