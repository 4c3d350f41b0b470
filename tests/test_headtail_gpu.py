"""Head-tail kernels (csrc/headtail.hip) against the same math in PyTorch on the CPU: the Q heads' last Linear + TD target /
critic loss + first backward stage, its actor-phase twin, the policy's last Linear + squashed-Gaussian head, and the
column-sum launch that finishes their per-workgroup partials (reference: mlp.py:97-100, sac.py:125-195, drq.py:76-103,
gaussian.py:83-87)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("M,H,group,rd_div", [(256, 1024, 1, 1), (128, 1024, 2, 2), (52, 512, 4, 1), (7, 256, 1, 1)])
def test_q_tail_critic(cuda, M, H, group, rd_div):
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(M + H)
    h2 = np.maximum(g.randn(2, M, H), 0).astype(np.float32)
    h2t = np.maximum(g.randn(2, M, H), 0).astype(np.float32)
    w2, b2 = (g.randn(2, H) / 32).astype(np.float32), g.randn(2).astype(np.float32)
    w2t, b2t = (g.randn(2, H) / 32).astype(np.float32), g.randn(2).astype(np.float32)
    nlp = g.randn(M).astype(np.float32)
    r = g.randn(M // rd_div).astype(np.float32)
    done = g.rand(M // rd_div) < 0.2
    log_alpha, gamma, rs = np.float32(math.log(0.1)), 0.99, 0.7
    # ---- reference ----
    h2_t, w_t, b_t = torch.from_numpy(h2), torch.from_numpy(w2).requires_grad_(True), torch.from_numpy(b2).requires_grad_(True)
    h2_t.requires_grad_(True)
    q = torch.einsum("hmk,hk->mh", h2_t, w_t) + b_t
    qn = torch.einsum("hmk,hk->mh", torch.from_numpy(h2t), torch.from_numpy(w2t)) + torch.from_numpy(b2t)
    alpha = float(torch.tensor(log_alpha).exp())
    rr = torch.from_numpy(r).repeat_interleave(rd_div)[:, None]
    dd = torch.from_numpy(done).repeat_interleave(rd_div).float()[:, None]
    y = rr * rs + (1 - dd) * gamma * (qn.min(-1, keepdim=True).values + alpha * torch.from_numpy(nlp)[:, None])
    if group > 1:
        y = y.reshape(M // group, group).mean(1, keepdim=True).repeat_interleave(group, 0)
    yy = y.repeat(1, 2).detach()
    loss = F.mse_loss(q, yy) * 2
    loss.backward()
    # ---- kernel ----
    H2, H2T = T(h2, cuda), T(h2t, cuda)
    WB = T(np.concatenate([np.concatenate([w2[h], b2[h:h + 1], np.zeros(3, np.float32)]) for h in range(2)]), cuda)      # head stride H + 4
    WBT = T(np.concatenate([np.concatenate([w2t[h], b2t[h:h + 1], np.zeros(3, np.float32)]) for h in range(2)]), cuda)
    n_part, n_stat = hip.q_tail_workspace_floats(M, H)
    part, stat = torch.zeros(n_part, device=cuda), torch.zeros(n_stat, device=cuda)
    qo, yo, dq, dh2 = torch.empty(M, 2, device=cuda), torch.empty(M, device=cuda), torch.empty(M, 2, device=cuda), torch.empty(2, M, H, device=cuda)
    hip.q_tail_critic(H2, M * H, WB, WB[H:], H + 4, H2T, M * H, WBT, WBT[H:], H + 4, T(nlp, cuda), T(r, cuda), T(done.astype(np.uint8), cuda), rd_div,
                      torch.tensor([log_alpha], device=cuda), gamma, rs, False, group, M, H, qo, yo, dq, dh2, part, stat)
    n_wg, Hp = (M + 3) // 4, H + 4
    dW, db, st = torch.empty(2, H, device=cuda), torch.empty(2, device=cuda), torch.empty(4, device=cuda)
    jobs = []
    for h in range(2):
        jobs.append((part.data_ptr() + 4 * h * Hp, 2 * Hp, n_wg, H, dW[h].data_ptr(), 1.0, 0))
        jobs.append((part.data_ptr() + 4 * (h * Hp + H), 2 * Hp, n_wg, 1, db[h:].data_ptr(), 1.0, 0))
    for k, (sc, op) in enumerate(((1.0 / M, 0), (1.0, 1), (1.0 / M, 0), (1.0 / M, 0))):
        jobs.append((stat.data_ptr() + 4 * k, 4, n_wg, 1, st[k:].data_ptr(), sc, op))
    hip.colsum_jobs(jobs)
    np.testing.assert_allclose(qo.cpu().numpy(), q.detach().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(yo.cpu().numpy(), y[:, 0].numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(dh2.cpu().numpy(), (h2_t.grad * (torch.from_numpy(h2) > 0)).numpy(), atol=1e-7, rtol=2e-4)
    np.testing.assert_allclose(dW.cpu().numpy(), w_t.grad.numpy(), atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), b_t.grad.numpy(), atol=2e-6, rtol=2e-4)
    want = [loss.item(), (q - yy).abs().max().item(), q.min(-1).values.mean().item(), yy.mean().item()]
    np.testing.assert_allclose(st.cpu().numpy(), want, rtol=3e-5, atol=2e-6)


@pytest.mark.parametrize("M,H", [(256, 1024), (33, 512)])
def test_q_tail_actor_and_finalize(cuda, M, H):
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(M)
    h2 = np.maximum(g.randn(2, M, H), 0).astype(np.float32)
    w2, b2 = (g.randn(2, H) / 32).astype(np.float32), g.randn(2).astype(np.float32)
    nlp = g.randn(M).astype(np.float32)
    log_alpha, target_entropy = np.float32(math.log(0.2)), -6.0
    h2_t = torch.from_numpy(h2).requires_grad_(True)
    q = torch.einsum("hmk,hk->mh", h2_t, torch.from_numpy(w2)) + torch.from_numpy(b2)
    nl = torch.from_numpy(nlp).requires_grad_(True)
    alpha = float(torch.tensor(log_alpha).exp())
    ent = nl.mean()
    aloss = -(q.min(-1, keepdim=True).values.mean() + alpha * ent)
    aloss.backward()
    la = torch.tensor([log_alpha], requires_grad=True)
    alpha_loss = la.exp() * (ent.detach() - target_entropy)
    alpha_loss.backward()
    WB = T(np.concatenate([np.concatenate([w2[h], b2[h:h + 1], np.zeros(3, np.float32)]) for h in range(2)]), cuda)
    _, n_stat = hip.q_tail_workspace_floats(M, H)
    stat = torch.zeros(n_stat, device=cuda)
    qo, dq, dh2, dn = torch.empty(M, 2, device=cuda), torch.empty(M, 2, device=cuda), torch.empty(2, M, H, device=cuda), torch.empty(1, device=cuda)
    LA = torch.tensor([log_alpha], device=cuda)
    hip.q_tail_actor(T(h2, cuda), M * H, WB, WB[H:], H + 4, T(nlp, cuda), LA, M, H, qo, dq, dh2, dn, stat)
    ag, st = torch.empty(1, device=cuda), torch.empty(3, device=cuda)
    hip.actor_finalize(stat, M, LA, target_entropy, ag, st)
    np.testing.assert_allclose(qo.cpu().numpy(), q.detach().numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(dh2.cpu().numpy(), (h2_t.grad * (torch.from_numpy(h2) > 0)).numpy(), atol=1e-8, rtol=2e-4)
    np.testing.assert_allclose(dn.item(), nl.grad[0].item(), rtol=1e-6)
    np.testing.assert_allclose(ag.item(), la.grad.item(), rtol=2e-5)
    np.testing.assert_allclose(st.cpu().numpy(), [aloss.item(), ent.item(), alpha_loss.item()], rtol=3e-5, atol=2e-6)


@pytest.mark.parametrize("M,H,A,sampled", [(256, 1024, 6, False), (130, 1024, 22, False), (40, 256, 3, True), (64, 1024, 12, False),
                                            (48, 1024, 30, False)])
def test_policy_tail_fwd(cuda, M, H, A, sampled):
    from oracle import torch_ref
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(A + M)
    h2 = np.maximum(g.randn(M, H), 0).astype(np.float32)
    w2, b2 = (g.randn(2 * A, H) / 32).astype(np.float32), (0.1 * g.randn(2 * A)).astype(np.float32)
    eps = g.randn(M, A).astype(np.float32)
    scale, bias = g.uniform(0.5, 2.0, A).astype(np.float32), g.uniform(-0.5, 0.5, A).astype(np.float32)
    feat = torch.empty(M, 2 * A, device=cuda)
    act, act2 = torch.empty(M, A, device=cuda), torch.zeros(M, A + 8, device=cuda)
    nlp, saved, eps_out = torch.empty(M, device=cuda), torch.empty(M, 2 * A, device=cuda), torch.empty(M, A, device=cuda)
    step = torch.zeros(1, dtype=torch.int32, device=cuda)
    hip.policy_tail_fwd(T(h2, cuda), M, H, T(w2, cuda), T(b2, cuda), A, None if sampled else T(eps, cuda), 1234, step, 1, eps_out, T(scale, cuda),
                        T(bias, cuda), -10.0, 2.0, 1e-6, feat, act, A, nlp, saved, action2_ptr=act2.data_ptr() + 4 * 5, ld_action2=A + 8)
    used = eps_out.cpu()
    if not sampled:
        assert torch.equal(used, torch.from_numpy(eps))
    else:
        assert abs(float(used.mean())) < 0.3 and 0.7 < float(used.std()) < 1.3
    ref_feat = F.linear(torch.from_numpy(h2), torch.from_numpy(w2), torch.from_numpy(b2))
    np.testing.assert_allclose(feat.cpu().numpy(), ref_feat.numpy(), atol=3e-5, rtol=1e-5)
    a_ref, nlp_ref = torch_ref.tanh_gaussian(feat.cpu(), used, torch.from_numpy(scale), torch.from_numpy(bias))
    np.testing.assert_allclose(act.cpu().numpy(), a_ref.numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(act2[:, 5:5 + A].cpu().numpy(), a_ref.numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(nlp.cpu().numpy(), nlp_ref[:, 0].numpy(), atol=2e-4, rtol=2e-5)


@pytest.mark.parametrize("M,H,A,K0,col0", [(256, 1024, 6, 56, 50), (130, 512, 22, 220, 196), (7, 256, 3, 12, 8), (130, 1024, 22, 220, 196),
                                           (64, 1024, 12, 120, 100), (48, 1024, 30, 100, 64)])
def test_policy_tail_bwd_equals_the_three_launches_it_replaces(cuda, M, H, A, K0, col0):
    """pcrl_policy_tail_bwd_f32 (+ the column gather riding on pcrl_q_tail_actor_cols_f32) against autograd of the same chain on the CPU
    -- d_action = sum_h dh1_h W0_h[:, action columns], TanhGaussianHead's backward, dh2 = (d_feat W2) (.) [h2 > 0] -- and against the
    launches it replaces (the d_action GEMM's result fed to pcrl_tanh_gaussian_bwd_f32); the folded actor_finalize equals the
    stand-alone one bit for bit."""
    from oracle import torch_ref
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(M + A)
    dh1 = (g.randn(2, M, H) * (g.rand(2, M, H) < 0.5) / 64).astype(np.float32)
    w0 = (g.randn(2, H, K0) / 8).astype(np.float32)
    h2q = np.maximum(g.randn(2, M, H), 0).astype(np.float32)
    w2q, b2q = (g.randn(2, H) / 32).astype(np.float32), g.randn(2).astype(np.float32)
    nlp = g.randn(M).astype(np.float32)
    feat_in = (0.5 * g.randn(M, 2 * A)).astype(np.float32)
    feat_in[:, A:] = np.clip(feat_in[:, A:], -12, 3)                      # some log_std outside the clamp: zero gradient there
    eps = g.randn(M, A).astype(np.float32)
    scale, bias = g.uniform(0.5, 2.0, A).astype(np.float32), g.uniform(-0.5, 0.5, A).astype(np.float32)
    h2p = np.maximum(g.randn(M, H), 0).astype(np.float32)
    w2p = (g.randn(2 * A, H) / 32).astype(np.float32)
    log_alpha, target_entropy = np.float32(math.log(0.2)), -float(A)
    LA = torch.tensor([log_alpha], device=cuda)
    # forward pieces the backward reads: saved = tanh(u) | std (from the forward head kernel)
    feat = T(feat_in, cuda)
    act, nl, saved = torch.empty(M, A, device=cuda), torch.empty(M, device=cuda), torch.empty(M, 2 * A, device=cuda)
    hip.tanh_gaussian_fwd(feat, 2 * A, T(eps, cuda), T(scale, cuda), T(bias, cuda), M, A, -10.0, 2.0, 1e-6, act, A, nl, saved)
    # the q tail (actor mode) with the column gather riding along
    WB = T(np.concatenate([np.concatenate([w2q[h], b2q[h:h + 1], np.zeros(3, np.float32)]) for h in range(2)]), cuda)
    _, n_stat = hip.q_tail_workspace_floats(M, H)
    stat = torch.zeros(n_stat, device=cuda)
    qo, dq, dh2q, dn = torch.empty(M, 2, device=cuda), torch.empty(M, 2, device=cuda), torch.empty(2, M, H, device=cuda), torch.empty(1, device=cuda)
    W0 = T(w0, cuda)
    cols = torch.full((2, A, H), float("nan"), device=cuda)
    hip.q_tail_actor_cols(T(h2q, cuda), M * H, WB, WB[H:], H + 4, T(nlp, cuda), LA, M, H, qo, dq, dh2q, dn, stat, W0, H * K0, K0, col0, A, cols)
    assert torch.equal(cols, W0[:, :, col0:col0 + A].permute(0, 2, 1).contiguous())
    qo2, dq2, dh2q2, dn2, stat2 = torch.empty_like(qo), torch.empty_like(dq), torch.empty_like(dh2q), torch.empty_like(dn), torch.zeros_like(stat)
    hip.q_tail_actor(T(h2q, cuda), M * H, WB, WB[H:], H + 4, T(nlp, cuda), LA, M, H, qo2, dq2, dh2q2, dn2, stat2)
    assert torch.equal(qo, qo2) and torch.equal(dh2q, dh2q2) and torch.equal(stat, stat2) and torch.equal(dn, dn2)
    # ---- the fused backward tail ----
    DH1 = T(dh1, cuda)
    dfeat, dh2 = torch.empty(M, 2 * A, device=cuda), torch.full((M, H), float("nan"), device=cuda)
    ag, st = torch.empty(1, device=cuda), torch.empty(3, device=cuda)
    hip.policy_tail_bwd(DH1, M * H, cols, A * H, M, H, A, feat, 2 * A, T(eps, cuda), saved, T(scale, cuda), -10.0, 2.0, 1e-6, dn, dfeat, 2 * A,
                        T(h2p, cuda), T(w2p, cuda), dh2, finalize=(stat, LA, target_entropy, ag, st))
    # the launches it replaces
    d_act = torch.einsum("hmk,hkj->mj", DH1.double(), W0[:, :, col0:col0 + A].double()).float()
    dfeat_ref = torch.empty(M, 2 * A, device=cuda)
    hip.tanh_gaussian_bwd(feat, 2 * A, T(eps, cuda), saved, T(scale, cuda), M, A, -10.0, 2.0, 1e-6, d_act.data_ptr(), None, A, dn, dfeat_ref, 2 * A)
    tol = dict(atol=2e-6 * float(dfeat_ref.abs().max()), rtol=2e-4)
    np.testing.assert_allclose(dfeat.cpu().numpy(), dfeat_ref.cpu().numpy(), **tol)
    dh2_ref = (dfeat_ref.double() @ T(w2p, cuda).double()).float() * (T(h2p, cuda) > 0)
    np.testing.assert_allclose(dh2.cpu().numpy(), dh2_ref.cpu().numpy(), atol=2e-6 * float(dh2_ref.abs().max()), rtol=2e-4)
    ag2, st2 = torch.empty(1, device=cuda), torch.empty(3, device=cuda)
    hip.actor_finalize(stat, M, LA, target_entropy, ag2, st2)
    assert torch.equal(ag, ag2) and torch.equal(st, st2)
    # and autograd of the reference's head on the CPU (gaussian.py:83-87; distributions.py:89,116-127) for d(mean | log_std)
    f_t = torch.from_numpy(feat_in).requires_grad_(True)
    a_ref, nlp_ref = torch_ref.tanh_gaussian(f_t, torch.from_numpy(eps), torch.from_numpy(scale), torch.from_numpy(bias))
    (a_ref * d_act.cpu()).sum().backward(retain_graph=True)
    (nlp_ref[:, 0] * float(dn.item())).sum().backward()
    np.testing.assert_allclose(dfeat.cpu().numpy(), f_t.grad.numpy(), atol=5e-6 * float(f_t.grad.abs().max()), rtol=5e-4)


def test_colsum_jobs_attached_to_the_encoder_backward_equal_the_stand_alone_launch(cuda):
    """pcrl_encoder_bwd_attach_colsum: the jobs run in extra workgroups of the next backward's reduce launch -- same results bit for
    bit as pcrl_colsum_jobs_f32, the encoder gradient untouched, and the attachment is consumed by that one call."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import make_encoder_weights, make_obs
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(5)
    part = T(g.randn(37, 2, 300).astype(np.float32), cuda)
    outs = [torch.full((300,), float("nan"), device=cuda) for _ in range(4)]
    jobs = lambda o: [(part.data_ptr(), 600, 37, 300, o[0].data_ptr(), 1.0, 0), (part.data_ptr() + 4 * 300, 600, 37, 257, o[1].data_ptr(), 0.5, 1)]
    hip.colsum_jobs(jobs(outs[0:2]))
    obs = make_obs(5, 200, seed=2)
    w = {k: T(v, cuda) for k, v in make_encoder_weights(6, 64, 128, 256, seed=3).items()}
    ew, _ = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(6, 64, 128, 256) // 4, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    desc, keep = hip.make_cloud_desc({k: T(v, cuda) for k, v in obs.items()})
    pooled, argmax = hip.encoder_fwd(desc, ew, packed)
    gp = torch.randn(5, 256, device=cuda)
    plain = hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=pooled).clone()
    hip.colsum_jobs(jobs(outs[2:4]), attach_to_encoder_bwd=True)
    torch.cuda.synchronize()
    assert torch.isnan(outs[2]).all()                                  # nothing has run yet
    with_jobs = hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=pooled).clone()
    assert torch.equal(with_jobs, plain)
    assert torch.equal(outs[2], outs[0]) and torch.equal(outs[3][:257], outs[1][:257]) and torch.isnan(outs[3][257:]).all()
    outs[2].fill_(float("nan"))
    hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=pooled)      # the attachment was consumed: a plain backward again
    torch.cuda.synchronize()
    assert torch.isnan(outs[2]).all()
    # the round-2 kernels (no pooled values given) run the attached jobs as a launch of their own
    hip.colsum_jobs(jobs(outs[2:4]), attach_to_encoder_bwd=True)
    hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=None)
    assert torch.equal(outs[2], outs[0])


@pytest.mark.parametrize("M,A,K0,col0", [(256, 6, 56, 50), (40, 22, 220, 196), (33, 3, 40, 30), (64, 12, 120, 100)])
def test_policy_tail_fold_finishes_the_q_heads_first_layer(cuda, M, A, K0, col0):
    """pcrl_policy_tail_fwd_fold_f32: besides the policy head, h1[h] = relu(pre[h] + action W0_h[:, action columns]^T) for the Q heads --
    against relu([feature | state | action] W0^T + b0) computed whole in float64; the column image comes from a gather job riding on a
    pack launch (pcrl_encoder_pack_attach_cols) and, when no pack follows, from pcrl_encoder_pack_flush_cols; without the fold the
    same launch returns exactly the plain tail's outputs."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import make_encoder_weights
    from pointcloud_rl_amd import hip
    H = 1024
    g = np.random.RandomState(M + A)
    h2 = np.maximum(g.randn(M, H), 0).astype(np.float32)
    w2, b2 = (g.randn(2 * A, H) / 32).astype(np.float32), (0.1 * g.randn(2 * A)).astype(np.float32)
    eps = g.randn(M, A).astype(np.float32)
    scale, bias = g.uniform(0.5, 2.0, A).astype(np.float32), g.uniform(-0.5, 0.5, A).astype(np.float32)
    w0 = (g.randn(2, H, K0) / 8).astype(np.float32)
    b0 = (0.1 * g.randn(2, H)).astype(np.float32)
    xq = g.randn(M, K0).astype(np.float32)                       # [feature | state | (action columns: overwritten by the tail)]
    W0, XQ = T(w0, cuda), T(xq, cuda)
    # the column image: riding on a pack launch, and through the flush
    cols = torch.full((2, A, H), float("nan"), device=cuda)
    hip.pack_attach_cols([(W0, H * K0, 2, H, K0, col0, A, cols)])
    w = {k: T(v, cuda) for k, v in make_encoder_weights(6, 64, 128, 256, seed=3).items()}
    ew, _ = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(6, 64, 128, 256) // 4, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    want_cols = W0[:, :, col0:col0 + A].permute(0, 2, 1).contiguous()
    assert torch.equal(cols, want_cols)
    ref_packed = torch.empty_like(packed)
    hip.encoder_pack_weights(ew, ref_packed)                     # nothing attached any more: a plain pack, same image
    assert torch.equal(packed, ref_packed)
    cols2 = torch.full((2, A, H), float("nan"), device=cuda)
    hip.pack_attach_cols([(W0, H * K0, 2, H, K0, col0, A, cols2)])
    hip.pack_flush_cols()
    assert torch.equal(cols2, want_cols)
    hip.pack_flush_cols()                                        # nothing pending: no-op
    # pre = [feature | state] W0[:, :col0]^T + b0 (what the GEMM ahead of the tail leaves)
    pre = (torch.einsum("mk,hnk->hmn", XQ[:, :col0].double(), W0[:, :, :col0].double()) + T(b0, cuda).double()[:, None, :]).float().contiguous()
    outs = {}
    for fold in (False, True):
        feat = torch.empty(M, 2 * A, device=cuda)
        act, act2 = torch.empty(M, A, device=cuda), XQ.clone()
        nlp, saved, eps_out = torch.empty(M, device=cuda), torch.empty(M, 2 * A, device=cuda), torch.empty(M, A, device=cuda)
        h1 = pre.clone()
        step = torch.zeros(1, dtype=torch.int32, device=cuda)
        hip.policy_tail_fwd(T(h2, cuda), M, H, T(w2, cuda), T(b2, cuda), A, T(eps, cuda), 7, step, 1, eps_out, T(scale, cuda), T(bias, cuda), -10.0, 2.0,
                            1e-6, feat, act, A, nlp, saved, action2_ptr=act2.data_ptr() + 4 * col0, ld_action2=K0,
                            fold=(h1, M * H, cols, A * H, 2, h1, M * H) if fold else None)
        outs[fold] = (feat, act, nlp, saved, act2, h1)
    for a_, b_ in zip(outs[False][:5], outs[True][:5]):
        assert torch.equal(a_, b_)
    act2, h1 = outs[True][4], outs[True][5]
    assert torch.equal(act2[:, col0:col0 + A], outs[True][1])
    z = torch.einsum("mk,hnk->hmn", act2[:, :col0 + A].double(), W0[:, :, :col0 + A].double()) + T(b0, cuda).double()[:, None, :]
    want = torch.relu(z).float()
    np.testing.assert_allclose(h1.cpu().numpy(), want.cpu().numpy(), atol=2e-5, rtol=1e-5)
    assert torch.equal(outs[False][5], pre)                      # without the fold the buffer is left alone
