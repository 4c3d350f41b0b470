"""GPU parity of the dense-head and update-tail kernels (C ABI) vs the same math in PyTorch on the CPU."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(params=["auto", "tile64", "tile32"])
def gemm_path(request):
    """Run a GEMM test on the default tile selection (round 5: wave-private staged tiles / weight-gradient panels wherever the operand
    layout allows), with the LDS-staged 64x64 tiles forced wherever a problem is eligible, and on the paths of rounds 1-4 only
    (32x32 split-K tiles, a tile per wave)."""
    from pointcloud_rl_amd import hip
    prev = hip.gemm_set_tile64_min({"auto": -1, "tile64": 1, "tile32": 1 << 30}[request.param])
    yield request.param
    hip.gemm_set_tile64_min(prev)


@pytest.mark.parametrize("M,K,N,relu", [(256, 56, 1024, True), (256, 1024, 1024, True), (256, 1024, 12, False), (37, 50, 70, True), (5, 3, 1, False),
                                        (128, 1024, 1024, True), (200, 196, 1000, True), (75, 1000, 130, False), (64, 64, 64, False)])
def test_linear_forward(cuda, gemm_path, M, K, N, relu):
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(M + K + N)
    x, w, b = g.randn(M, K).astype(np.float32), (g.randn(N, K) / np.sqrt(K)).astype(np.float32), g.randn(N).astype(np.float32)
    X, W, Bv = T(x, cuda), T(w, cuda), T(b, cuda)
    Y = torch.empty(M, N, device=cuda)
    hip.gemm(X, W, Y, M, N, K, (K, 1), (1, K), N, bias=Bv, relu=relu)
    ref = F.linear(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b))
    ref = F.relu(ref) if relu else ref
    np.testing.assert_allclose(Y.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("M,K,N", [(256, 56, 1024), (256, 1024, 1024), (100, 196, 520)])
def test_linear_backward_and_batched(cuda, gemm_path, M, K, N):
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(0)
    H = 2
    x = g.randn(H, M, K).astype(np.float32)
    act = np.maximum(g.randn(H, M, K), 0).astype(np.float32)           # the layer input as a ReLU output (mask)
    w = (g.randn(H, N, K) / np.sqrt(K)).astype(np.float32)
    dy = g.randn(H, M, N).astype(np.float32)
    X, ACT, W, DY = T(x, cuda), T(act, cuda), T(w, cuda), T(dy, cuda)
    # data gradient with the ReLU mask of the layer input, both heads in one launch
    DX = torch.empty(H, M, K, device=cuda)
    hip.gemm(DY, W, DX, M, K, N, (N, 1), (K, 1), K, mask=ACT, ld_mask=K, batch=H, batch_strides=(M * N, N * K, M * K, 0, M * K))
    ref_dx = torch.einsum("hmn,hnk->hmk", torch.from_numpy(dy), torch.from_numpy(w)) * (torch.from_numpy(act) > 0)
    np.testing.assert_allclose(DX.cpu().numpy(), ref_dx.numpy(), atol=1e-4, rtol=1e-5)
    # weight + bias gradient: dW|db = dy^T [x | 1]
    DWB = torch.empty(H, N, K + 1, device=cuda)
    hip.gemm(DY, X, DWB, N, K + 1, M, (1, N), (K, 1), K + 1, ones_col=K, batch=H, batch_strides=(M * N, M * K, N * (K + 1), 0, 0))
    ref_dw = torch.einsum("hmn,hmk->hnk", torch.from_numpy(dy), torch.from_numpy(x))
    ref_db = torch.from_numpy(dy).sum(1)
    np.testing.assert_allclose(DWB[:, :, :K].cpu().numpy(), ref_dw.numpy(), atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(DWB[:, :, K].cpu().numpy(), ref_db.numpy(), atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("path", ["auto", "legacy"])
@pytest.mark.parametrize("M,N,K,kind", [(1000, 1100, 200, "epilogue"), (1024, 1025, 256, "ones"), (1500, 1030, 33, "accumulate"),
                                        (1001, 1100, 200, "epilogue"), (1023, 1025, 250, "ones"), (1501, 1030, 40, "accumulate"),
                                        (1024, 1025, 32, "ones"), (64, 129, 100, "ones"), (2048, 1024, 512, "accumulate")])
def test_weight_gradient_paths_with_every_epilogue(cuda, path, M, N, K, kind):
    """Weight-gradient-shaped problems (both operands contiguous along their row index, K = batch).  Default: LDS panels of 64 x 64 /
    64 x 128 outputs (dense_wtile.h: gemm_wgrad_panel) when M and the real column count are multiples of 4; otherwise, and with the
    rounds-1-4 paths selected, a tile per WAVE from 1 536 tiles of 32 x 32 (dense.hip cfg 2; cfg 3 -- two row blocks per wave from
    8-byte loads -- when A is contiguous along an EVEN number of rows).  Ragged M / N / K, bias + ReLU + mask, the ones column with
    its separate destination, accumulation into C -- against float64."""
    from pointcloud_rl_amd import hip
    prev = hip.gemm_set_tile64_min(-1 if path == "auto" else 1 << 30)
    try:
        _weight_gradient_case(cuda, path, M, N, K, kind)
    finally:
        hip.gemm_set_tile64_min(prev)


def _weight_gradient_case(cuda, path, M, N, K, kind):
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(M + N + K)
    H = 2
    a = g.randn(H, K, M).astype(np.float32)              # A row-contiguous (stride 1 along m): the weight gradient's dY^T
    b = g.randn(H, K, N).astype(np.float32)              # B row-contiguous
    A_, B_ = T(a, cuda), T(b, cuda)
    ref = np.matmul(a.astype(np.float64).transpose(0, 2, 1), b.astype(np.float64))
    if kind == "epilogue":
        bias, mask = g.randn(H, N).astype(np.float32), (g.rand(H, M, N) < 0.6).astype(np.float32)
        C = torch.full((H, M, N), float("nan"), device=cuda)
        plan = hip.gemm_plan([hip.gemm_desc(A_, B_, C, M, N, K, (1, M), (N, 1), N, batch=H, batch_strides=(K * M, K * N, M * N, N, M * N))])
        assert plan[0][0] == (5 if path == "auto" and M % 4 == 0 and N % 4 == 0 and min(M, N) >= 256 else 3 if M % 2 == 0 else 2), plan
        hip.gemm(A_, B_, C, M, N, K, (1, M), (N, 1), N, bias=T(bias, cuda), mask=T(mask, cuda), ld_mask=N, relu=True, batch=H,
                 batch_strides=(K * M, K * N, M * N, N, M * N))
        want = np.maximum(ref + bias[:, None, :], 0) * mask
        np.testing.assert_allclose(C.cpu().numpy(), want, atol=2e-4, rtol=1e-5)
    elif kind == "ones":
        # column N - 1 of B does not exist: it stands for a column of ones (bias gradient) and lands in its own vector.  B holds the N - 1
        # real columns (as the heads' activations do: the panels need 16-byte aligned rows)
        b1 = np.ascontiguousarray(b[:, :, :N - 1])
        B1 = T(b1, cuda)
        C = torch.full((H, M, N), float("nan"), device=cuda)
        ones = torch.full((H, M), float("nan"), device=cuda)
        d = hip.gemm_desc(A_, B1, C, M, N, K, (1, M), (N - 1, 1), N, ones_col=N - 1, batch=H, batch_strides=(K * M, K * (N - 1), M * N, 0, 0),
                          c_ones=ones, c_ones_batch_stride=M)
        if path == "auto":
            assert (hip.gemm_plan([d])[0][0] == 5) == (M % 4 == 0 and (N - 1) % 4 == 0 and min(M, N - 1) >= 256), hip.gemm_plan([d])
        hip.gemm_group([d])
        np.testing.assert_allclose(C[:, :, :N - 1].cpu().numpy(), ref[:, :, :N - 1], atol=2e-4, rtol=1e-5)
        np.testing.assert_allclose(ones.cpu().numpy(), a.astype(np.float64).sum(1), atol=2e-4, rtol=1e-5)
        assert torch.isnan(C[:, :, N - 1]).all()
    else:
        c0 = g.randn(H, M, N).astype(np.float32)
        C = T(c0, cuda)
        hip.gemm(A_, B_, C, M, N, K, (1, M), (N, 1), N, accumulate=True, batch=H, batch_strides=(K * M, K * N, M * N, 0, 0))
        np.testing.assert_allclose(C.cpu().numpy(), c0 + ref, atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("kind", ["forward", "data-gradient"])
@pytest.mark.parametrize("M,N,K,H,want_shape", [(32, 1024, 1024, 1, 0), (32, 1024, 1024, 4, 1), (64, 1024, 1024, 4, 2), (128, 1024, 1024, 1, 1),
                                                (256, 1024, 1024, 1, 2), (256, 1024, 1024, 2, 3), (37, 50, 1000, 2, 0), (5, 3, 512, 1, 0),
                                                (300, 1030, 516, 1, None), (256, 50, 1024, 2, 0), (100, 520, 772, 3, None), (128, 1024, 1024, 4, 3)])
def test_wave_private_staged_tiles(cuda, kind, M, N, K, H, want_shape):
    """dense_wtile.h::gemm_wtile: every tile shape (16 x 16 ... 32 x 64 outputs, v_mfma_f32_16x16x4_f32 and 32x32x2 blocks), forward-shaped
    (A, B k-contiguous) and data-gradient-shaped (B contiguous along n), ragged M / N / K (K % 4 == 0, K >= 512), heads batched; bias + ReLU resp.
    the ReLU mask of the layer input, and accumulation -- against float64.  The launch plan must say the shape is reached."""
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(M + N + K + H)
    a = g.randn(H, M, K).astype(np.float32)
    A_ = T(a, cuda)
    if kind == "forward":
        w, b = (g.randn(H, N, K) / np.sqrt(K)).astype(np.float32), g.randn(H, N).astype(np.float32)
        W_, B_ = T(w, cuda), T(b, cuda)
        C = torch.full((H, M, N), float("nan"), device=cuda)
        d = hip.gemm_desc(A_, W_, C, M, N, K, (K, 1), (1, K), N, bias=B_, relu=True, batch=H, batch_strides=(M * K, N * K, M * N, N, 0))
        want = np.maximum(np.matmul(a.astype(np.float64), w.astype(np.float64).transpose(0, 2, 1)) + b[:, None, :], 0)
    else:
        w = (g.randn(H, K, N) / np.sqrt(K)).astype(np.float32)
        mask = np.maximum(g.randn(H, M, N), 0).astype(np.float32)
        c0 = g.randn(H, M, N).astype(np.float32)
        W_, MK, C = T(w, cuda), T(mask, cuda), T(c0, cuda)
        d = hip.gemm_desc(A_, W_, C, M, N, K, (K, 1), (N, 1), N, mask=MK, ld_mask=N, accumulate=True, batch=H, batch_strides=(M * K, K * N, M * N, 0, M * N))
        want = c0 + np.matmul(a.astype(np.float64), w.astype(np.float64)) * (mask > 0)
    path, shape, wgs = hip.gemm_plan([d])[0]
    assert path == 4, (path, shape, wgs)
    if want_shape is not None:
        assert shape == min(want_shape, 2 if kind == "data-gradient" else 3), (shape, wgs)
    hip.gemm_group([d])
    np.testing.assert_allclose(C.cpu().numpy(), want, atol=3e-5 * np.sqrt(K / 128), rtol=1e-5)


def test_grouped_launch_mixes_the_round5_paths(cuda):
    """dW | db (panels) + dx (wave tiles, row-contiguous B) + a forward layer (wave tiles) + a short-K problem (32 x 32 split-K) in ONE
    launch == the same problems launched one by one, bit for bit (a problem's tile path does not depend on its neighbours)."""
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(11)
    M, K, N = 128, 1024, 1024
    dy, x, w = g.randn(2, M, N).astype(np.float32), g.randn(2, M, K).astype(np.float32), (g.randn(2, N, K) / 32).astype(np.float32)
    x2, w2 = g.randn(70, 56).astype(np.float32), g.randn(33, 56).astype(np.float32)
    DY, X, W, X2, W2 = T(dy, cuda), T(x, cuda), T(w, cuda), T(x2, cuda), T(w2, cuda)

    def descs(dwb, db, dx, y2, y3):
        return [hip.gemm_desc(DY, X, dwb, N, K + 1, M, (1, N), (K, 1), K, ones_col=K, c_ones=db, c_ones_batch_stride=N, batch=2, batch_strides=(M * N, M * K, N * K, 0, 0)),
                hip.gemm_desc(DY, W, dx, M, K, N, (N, 1), (K, 1), K, batch=2, batch_strides=(M * N, N * K, M * K, 0, 0)),
                hip.gemm_desc(X2, W2, y2, 70, 33, 56, (56, 1), (1, 56), 33, relu=True),
                hip.gemm_desc(X, W, y3, M, N, K, (K, 1), (1, K), N, batch=2, batch_strides=(M * K, N * K, M * N, 0, 0))]
    shapes = [(2, N, K), (2, N), (2, M, K), (70, 33), (2, M, N)]
    outs_a = [torch.zeros(*sh, device=cuda) for sh in shapes]
    outs_b = [torch.zeros(*sh, device=cuda) for sh in shapes]
    assert [pl[0] for pl in hip.gemm_plan(descs(*outs_a))] == [5, 4, 0, 4]
    hip.gemm_group(descs(*outs_a))
    for d in descs(*outs_b):
        hip.gemm_group([d])
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    f64 = np.float64
    np.testing.assert_allclose(outs_a[0].cpu().numpy(), np.matmul(dy.astype(f64).transpose(0, 2, 1), x.astype(f64)), atol=3e-4, rtol=1e-5)
    np.testing.assert_allclose(outs_a[1].cpu().numpy(), dy.astype(f64).sum(1), atol=3e-4, rtol=1e-5)
    np.testing.assert_allclose(outs_a[2].cpu().numpy(), np.matmul(dy.astype(f64), w.astype(f64)), atol=3e-4, rtol=1e-5)
    np.testing.assert_allclose(outs_a[4].cpu().numpy(), np.matmul(x.astype(f64), w.astype(f64).transpose(0, 2, 1)), atol=3e-4, rtol=1e-5)


def test_layernorm_rows(cuda):
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(1)
    M, Fd = 77, 50
    x, gam, bet = g.randn(M, Fd).astype(np.float32), g.uniform(0.5, 1.5, Fd).astype(np.float32), g.randn(Fd).astype(np.float32)
    dy0, dy1 = g.randn(M, 64).astype(np.float32), g.randn(M, 64).astype(np.float32)
    X, G, Bt = T(x, cuda), T(gam, cuda), T(bet, cuda)
    out_a, out_b = torch.zeros(M, 56, device=cuda), torch.zeros(M, 64, device=cuda)
    xhat, rstd = torch.empty(M, Fd, device=cuda), torch.empty(M, device=cuda)
    hip.layernorm_rows_fwd(X, Fd, G, Bt, M, Fd, 1e-5, [(out_a, 0, 56), (out_b, 0, 64)], xhat, rstd)
    xt = torch.from_numpy(x).requires_grad_(True)
    gt, bt = torch.from_numpy(gam).requires_grad_(True), torch.from_numpy(bet).requires_grad_(True)
    ref = F.layer_norm(xt, (Fd,), gt, bt, 1e-5)
    np.testing.assert_allclose(out_a[:, :Fd].cpu().numpy(), ref.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(out_b[:, :Fd].cpu().numpy(), ref.detach().numpy(), atol=2e-6)
    assert (out_a[:, Fd:] == 0).all()
    D0, D1 = T(dy0, cuda), T(dy1, cuda)
    dx, dg, db = torch.empty(M, Fd, device=cuda), torch.empty(Fd, device=cuda), torch.empty(Fd, device=cuda)
    ws = torch.empty(((M + 3) // 4) * 2 * Fd * 4, dtype=torch.uint8, device=cuda)
    hip.layernorm_rows_bwd(D0.data_ptr(), D1.data_ptr(), 64, xhat, rstd, G, M, Fd, dx, Fd, dg, db, ws)
    ref.backward(torch.from_numpy(dy0[:, :Fd] + dy1[:, :Fd]))
    np.testing.assert_allclose(dx.cpu().numpy(), xt.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(dg.cpu().numpy(), gt.grad.numpy(), atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), bt.grad.numpy(), atol=1e-4)


def test_tanh_gaussian_forward_backward(cuda):
    from oracle import torch_ref
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(2)
    B, A = 130, 6
    feat = (g.randn(B, 2 * A) * 1.5).astype(np.float32)
    feat[0, A] = -30.0          # below the clamp: zero gradient to log_std
    feat[1, A + 1] = 5.0        # above the clamp
    eps = g.randn(B, A).astype(np.float32)
    scale, bias = g.uniform(0.5, 2.0, A).astype(np.float32), g.randn(A).astype(np.float32)
    da = g.randn(B, A).astype(np.float32)
    d_nlp = np.float32(-0.013)
    ft = torch.from_numpy(feat).requires_grad_(True)
    a_ref, nlp_ref = torch_ref.tanh_gaussian(ft, torch.from_numpy(eps), torch.from_numpy(scale), torch.from_numpy(bias))
    ((a_ref * torch.from_numpy(da)).sum() + (nlp_ref * float(d_nlp)).sum()).backward()
    Fd, E, S, Bi = T(feat, cuda), T(eps, cuda), T(scale, cuda), T(bias, cuda)
    act, act2 = torch.empty(B, A, device=cuda), torch.zeros(B, 20, device=cuda)
    nlp, saved = torch.empty(B, device=cuda), torch.empty(B, 2 * A, device=cuda)
    hip.tanh_gaussian_fwd(Fd, 2 * A, E, S, Bi, B, A, -10.0, 2.0, 1e-6, act, A, nlp, saved, action2_ptr=act2.data_ptr() + 4 * 14, ld_action2=20)
    np.testing.assert_allclose(act.cpu().numpy(), a_ref.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(act2[:, 14:].cpu().numpy(), a_ref.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(nlp.cpu().numpy(), nlp_ref.detach().numpy()[:, 0], atol=2e-5, rtol=1e-5)
    dfeat = torch.empty(B, 2 * A, device=cuda)
    DA, DN = T(da, cuda), torch.tensor([d_nlp], device=cuda)     # raw pointers below: keep the tensors alive
    hip.tanh_gaussian_bwd(Fd, 2 * A, E, saved, S, B, A, -10.0, 2.0, 1e-6, DA.data_ptr(), None, A, DN, dfeat, 2 * A)
    torch.cuda.synchronize()
    got, want = dfeat.cpu().numpy(), ft.grad.numpy()
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=2e-4)
    assert got[0, A] == 0.0 and got[1, A + 1] == 0.0


@pytest.mark.parametrize("group", [1, 2])
def test_critic_and_actor_loss(cuda, group):
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(3)
    B, H = 256 * group, 2
    qn, q = g.randn(B, H).astype(np.float32), g.randn(B, H).astype(np.float32)
    nlp, r = g.randn(B).astype(np.float32), g.randn(B).astype(np.float32)
    done = g.rand(B) < 0.2
    log_alpha = np.float32(math.log(0.1))
    gamma, rs = 0.99, 1.0 if group > 1 else 0.7
    qt = torch.from_numpy(q).requires_grad_(True)
    alpha = float(torch.tensor(log_alpha).exp())
    y = torch.from_numpy(r)[:, None] * rs + (1 - torch.from_numpy(done).float()[:, None]) * gamma * (
        torch.from_numpy(qn).min(-1, keepdim=True).values + alpha * torch.from_numpy(nlp)[:, None])
    if group > 1:
        y = y.reshape(B // group, group).mean(1, keepdim=True).repeat_interleave(group, 0)
    yy = y.repeat(1, H)
    loss = F.mse_loss(qt, yy) * H
    loss.backward()
    out_y, dq, stats = torch.empty(B, device=cuda), torch.empty(B, H, device=cuda), torch.empty(4, device=cuda)
    hip.sac_critic_loss(T(qn, cuda), H, T(nlp, cuda), T(r, cuda), T(done.astype(np.uint8), cuda), torch.tensor([log_alpha], device=cuda),
                        gamma, rs, False, group, T(q, cuda), H, B, H, out_y, dq, H, stats)
    np.testing.assert_allclose(out_y.cpu().numpy(), y[:, 0].numpy(), atol=2e-6)
    np.testing.assert_allclose(dq.cpu().numpy(), qt.grad.numpy(), atol=1e-7, rtol=1e-5)
    want = [loss.item(), (qt - yy).abs().max().item(), qt.min(-1).values.mean().item(), yy.mean().item()]
    np.testing.assert_allclose(stats.cpu().numpy(), want, rtol=2e-5, atol=1e-6)
    if group > 1:
        # rewards / dones stored once per sample, read by row // group (DrQ without the repeat_interleave): same bits when
        # the per-row arrays are the per-sample arrays repeated
        r_s, done_s = g.randn(B // group).astype(np.float32), g.rand(B // group) < 0.2
        outs = []
        for rr, dd, div in ((np.repeat(r_s, group), np.repeat(done_s, group), 1), (r_s, done_s, group)):
            oy, odq, ost = torch.empty(B, device=cuda), torch.empty(B, H, device=cuda), torch.empty(4, device=cuda)
            hip.sac_critic_loss(T(qn, cuda), H, T(nlp, cuda), T(rr, cuda), T(dd.astype(np.uint8), cuda), torch.tensor([log_alpha], device=cuda),
                                gamma, rs, False, group, T(q, cuda), H, B, H, oy, odq, H, ost, rd_row_div=div)
            outs.append((oy, odq, ost))
        for x, y_ in zip(*outs):
            assert torch.equal(x, y_)
    # actor / alpha loss
    qp = torch.from_numpy(q).requires_grad_(True)
    nl = torch.from_numpy(nlp).requires_grad_(True)
    ent = nl.mean()
    aloss = -(qp.min(-1, keepdim=True).values.mean() + alpha * ent)
    aloss.backward()
    la = torch.tensor([log_alpha], requires_grad=True)
    alpha_loss = la.exp() * (ent.detach() - (-6.0))
    alpha_loss.backward()
    dq2, dn, ag, st = torch.empty(B, H, device=cuda), torch.empty(1, device=cuda), torch.empty(1, device=cuda), torch.empty(3, device=cuda)
    hip.sac_actor_loss(T(q, cuda), H, T(nlp, cuda), torch.tensor([log_alpha], device=cuda), -6.0, B, H, dq2, H, dn, ag, st)
    np.testing.assert_allclose(dq2.cpu().numpy(), qp.grad.numpy(), atol=1e-9)
    np.testing.assert_allclose(dn.item(), nl.grad[0].item(), rtol=1e-6)
    np.testing.assert_allclose(ag.item(), la.grad.item(), rtol=1e-5)
    np.testing.assert_allclose(st.cpu().numpy(), [aloss.item(), ent.item(), alpha_loss.item()], rtol=2e-5, atol=1e-6)


def test_gemm_group_matches_single_launches(cuda, gemm_path):
    """dW|db, dx and two unrelated shapes in one launch give the same bits as separate launches."""
    if gemm_path == "auto":
        pytest.skip("a group and its single launches may pick different tile shapes (different summation order)")
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(5)
    M, K, N = 256, 1024, 1024
    dy, x, w = g.randn(M, N).astype(np.float32), g.randn(M, K).astype(np.float32), (g.randn(N, K) / 32).astype(np.float32)
    x2, w2 = g.randn(70, 50).astype(np.float32), g.randn(33, 50).astype(np.float32)
    DY, X, W, X2, W2 = T(dy, cuda), T(x, cuda), T(w, cuda), T(x2, cuda), T(w2, cuda)

    def descs(dwb, dx, y2, y3):
        return [hip.gemm_desc(DY, X, dwb, N, K + 1, M, (1, N), (K, 1), K + 1, ones_col=K),
                hip.gemm_desc(DY, W, dx, M, K, N, (N, 1), (K, 1), K),
                hip.gemm_desc(X2, W2, y2, 70, 33, 50, (50, 1), (1, 50), 33, relu=True),
                hip.gemm_desc(X, W, y3, M, N, K, (K, 1), (1, K), N)]
    outs_a = [torch.zeros(N, K + 1, device=cuda), torch.zeros(M, K, device=cuda), torch.zeros(70, 33, device=cuda), torch.zeros(M, N, device=cuda)]
    outs_b = [torch.zeros_like(o) for o in outs_a]
    hip.gemm_group(descs(*outs_a))
    for d in descs(*outs_b):
        hip.gemm_group([d])
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    np.testing.assert_allclose(outs_a[0][:, :K].cpu().numpy(), dy.T @ x, atol=3e-4, rtol=1e-5)
    np.testing.assert_allclose(outs_a[0][:, K].cpu().numpy(), dy.sum(0), atol=3e-4, rtol=1e-5)
    np.testing.assert_allclose(outs_a[1].cpu().numpy(), dy @ w, atol=3e-4, rtol=1e-5)
    np.testing.assert_allclose(outs_a[2].cpu().numpy(), np.maximum(x2 @ w2.T, 0), atol=2e-5, rtol=1e-5)
    with pytest.raises(RuntimeError):
        hip.gemm_group(descs(*outs_a) + descs(*outs_a)[:1])          # more than 4 problems


def test_layernorm_rows_multi_job_with_pass_through_columns(cuda):
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(3)
    Fd = 50
    G, Bt = T(g.randn(Fd).astype(np.float32), cuda), T(g.randn(Fd).astype(np.float32), cuda)
    jobs, wants = [], []
    for M in (37, 256):
        x, st, ac = g.randn(M, Fd).astype(np.float32), g.randn(M, 5).astype(np.float32), g.randn(M, 6).astype(np.float32)
        X, ST, AC = T(x, cuda), T(st, cuda), T(ac, cuda)
        dst = torch.zeros(M, 64, device=cuda)
        jobs.append(dict(x=X, ldx=Fd, M=M, dsts=[(dst, 0, 64)], cats=[(ST, dst, Fd, 64), (AC, dst, Fd + 5, 64)], keep=(X, ST, AC, dst)))
        ref = F.layer_norm(torch.from_numpy(x), (Fd,), G.cpu(), Bt.cpu(), 1e-5)
        wants.append(np.concatenate([ref.numpy(), st, ac, np.zeros((M, 64 - Fd - 11), np.float32)], 1))
    # a job whose pass-through sources hold one row per SAMPLE and are read by row // 2 (DrQ's virtual repeat)
    M = 64
    x, st, ac = g.randn(M, Fd).astype(np.float32), g.randn(M // 2, 5).astype(np.float32), g.randn(M // 2, 6).astype(np.float32)
    X, ST, AC = T(x, cuda), T(st, cuda), T(ac, cuda)
    dst = torch.zeros(M, 64, device=cuda)
    jobs.append(dict(x=X, ldx=Fd, M=M, dsts=[(dst, 0, 64)], cats=[(ST, dst, Fd, 64, 2), (AC, dst, Fd + 5, 64, 2)], keep=(X, ST, AC, dst)))
    ref = F.layer_norm(torch.from_numpy(x), (Fd,), G.cpu(), Bt.cpu(), 1e-5)
    wants.append(np.concatenate([ref.numpy(), np.repeat(st, 2, 0), np.repeat(ac, 2, 0), np.zeros((M, 64 - Fd - 11), np.float32)], 1))
    hip.layernorm_rows_fwd_multi(jobs, G, Bt, Fd, 1e-5)
    for job, want in zip(jobs, wants):
        np.testing.assert_allclose(job["keep"][3].cpu().numpy(), want, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("defer", [False, True], ids=["finalize-now", "finalize-in-gather"])
def test_fused_adam_polyak_gradnorm_matches_torch(cuda, defer):
    """pcrl_adam_step_f32 == torch.optim.Adam (+ soft_update on a parameter range, + grad 2-norm, + device step count),
    with the pass's second half either launched right away or deferred into pcrl_gather_scalars_f32."""
    from pointcloud_rl_amd import hip
    g = np.random.RandomState(2)
    n, t0, t1, tau = 10_007, 4_000, 10_007, 0.01
    w0 = g.randn(n).astype(np.float32)
    p_ref = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    opt = torch.optim.Adam([p_ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    tgt_ref = torch.from_numpy(g.randn(t1 - t0).astype(np.float32))
    P, M, V = T(w0, cuda), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    TG = tgt_ref.clone().to(cuda)
    step, norm = torch.zeros(1, dtype=torch.int32, device=cuda), torch.zeros(1, device=cuda)
    ws = torch.empty(hip.adam_workspace_bytes(n), dtype=torch.uint8, device=cuda)
    seen = torch.zeros(2, device=cuda)
    for it in range(3):
        grad = g.randn(n).astype(np.float32) * (10.0 ** -it)
        p_ref.grad = torch.from_numpy(grad * 0.5)
        opt.step()
        tgt_ref = tgt_ref * (1 - tau) + p_ref.detach()[t0:t1] * tau
        pend = hip.adam_step(P, T(grad, cuda), M, V, 1e-3, 0.9, 0.999, 1e-8, 0.5, step, norm, ws, target=TG, target_begin=t0,
                             target_end=t1, tau=tau, defer=defer)
        if defer:
            assert pend is not None and int(step.item()) == it         # the count advances only when the pass is finished
            hip.gather_scalars([(norm, seen[0:], False), (P[5:], seen[1:], True)], pending=[pend])
            assert abs(float(seen[0]) - float(norm)) == 0 and abs(float(seen[1]) - np.exp(float(P[5]))) < 1e-6
        assert int(step.item()) == it + 1
        np.testing.assert_allclose(float(norm), np.linalg.norm(grad * 0.5), rtol=1e-5)
        np.testing.assert_allclose(P.cpu().numpy(), p_ref.detach().numpy(), atol=2e-6, rtol=0)
        np.testing.assert_allclose(TG.cpu().numpy(), tgt_ref.numpy(), atol=2e-6, rtol=0)
