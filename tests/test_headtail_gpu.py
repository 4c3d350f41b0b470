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


@pytest.mark.parametrize("M,H,A,sampled", [(256, 1024, 6, False), (130, 1024, 22, False), (40, 256, 3, True)])
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
