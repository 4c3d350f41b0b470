"""Times the GEMM shapes of the K1 update step (HIP events, L2-warm steady state).  python tools/bench_gemm.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_rl_amd import hip

dev = torch.device("cuda", 0)
M, H = int(os.environ.get("GEMM_M", "256")), 1024


def t(*shape):
    return torch.randn(*shape, device=dev)


def shapes():
    X56, W0, h, W1, W2 = t(2, M, 56), t(2, H, 56), t(2, M, H), t(2, H, H), t(2, 1, H)
    yield "fwd0 56->1024 x2", [hip.gemm_desc(X56, W0, t(2, M, H), M, H, 56, (56, 1), (1, 56), H, relu=True, batch=2, batch_strides=(M * 56, H * 56, M * H, 0, 0))]
    yield "fwd1 1024->1024 x2", [hip.gemm_desc(h, W1, t(2, M, H), M, H, H, (H, 1), (1, H), H, relu=True, batch=2, batch_strides=(M * H, H * H, M * H, 0, 0))]
    yield "fwd1 1024->1024 x1", [hip.gemm_desc(h, W1, t(1, M, H), M, H, H, (H, 1), (1, H), H, relu=True)]
    yield "fwd1 x2 | x2 (online+target)", [hip.gemm_desc(h, W1, t(2, M, H), M, H, H, (H, 1), (1, H), H, relu=True, batch=2, batch_strides=(M * H, H * H, M * H, 0, 0)),
                                           hip.gemm_desc(t(2, M, H), t(2, H, H), t(2, M, H), M, H, H, (H, 1), (1, H), H, relu=True, batch=2, batch_strides=(M * H, H * H, M * H, 0, 0))]
    yield "fwd2 1024->1 x2", [hip.gemm_desc(h, W2, t(M, 2), M, 1, H, (H, 1), (1, H), 2, batch=2, batch_strides=(M * H, H, 1, 0, 0))]
    yield "fwd2 1024->12 x1", [hip.gemm_desc(h, t(12, H), t(M, 12), M, 12, H, (H, 1), (1, H), 12)]
    dW = hip.gemm_desc(h, h, t(2, H, H + 1), H, H + 1, M, (1, H), (H, 1), H + 1, ones_col=H, batch=2, batch_strides=(M * H, M * H, H * (H + 1), 0, 0))
    dX = hip.gemm_desc(h, W1, t(2, M, H), M, H, H, (H, 1), (H, 1), H, mask=h, ld_mask=H, batch=2, batch_strides=(M * H, H * H, M * H, 0, M * H))
    yield "dW1 x2", [dW]
    yield "dh1 x2", [dX]
    yield "dW1 | dh1 x2", [dW, dX]
    yield "dh1 | dW1 x2 (long K first)", [dX, dW]
    dW1 = hip.gemm_desc(h, h, t(1, H, H + 1), H, H + 1, M, (1, H), (H, 1), H + 1, ones_col=H)
    dX1 = hip.gemm_desc(h, W1, t(1, M, H), M, H, H, (H, 1), (H, 1), H, mask=h, ld_mask=H)
    yield "dW1 x1", [dW1]
    yield "dh1 x1", [dX1]
    yield "dW1 | dh1 x1", [dW1, dX1]
    yield "dh1 | dW1 x1", [dX1, dW1]
    yield "dX0 1024->50 x2", [hip.gemm_desc(h, W0, t(2, M, 52), M, 50, H, (H, 1), (56, 1), 52, batch=2, batch_strides=(M * H, H * 56, M * 52, 0, 0))]
    yield "dW0 x2", [hip.gemm_desc(h, X56, t(2, H, 57), H, 57, M, (1, H), (56, 1), 57, ones_col=56, batch=2, batch_strides=(M * H, M * 56, H * 57, 0, 0))]
    yield "feat 256->50", [hip.gemm_desc(t(M, 256), t(50, 256), t(M, 50), M, 50, 256, (256, 1), (1, 256), 50)]
    yield "K=1 dh2 x2", [hip.gemm_desc(t(M, 2), W2, t(2, M, H), M, H, 1, (2, 1), (H, 1), H, mask=h, ld_mask=H, batch=2, batch_strides=(1, H, M * H, 0, M * H))]


def time_group(descs, n=200):
    for _ in range(5):
        hip.gemm_group(descs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        hip.gemm_group(descs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'shape':32s} {'32x32 split-K':>22s} {'64x64 LDS tiles':>22s} {'default':>22s}")
for name, descs in shapes():
    flops = sum(2.0 * d.M * d.N * d.K * d.batch for d in descs)
    cols = []
    for min_tiles in (1 << 30, 1, None):
        prev = hip.gemm_set_tile64_min(-1 if min_tiles is None else min_tiles)
        us = time_group(descs)
        if min_tiles is not None:
            hip.gemm_set_tile64_min(prev)
        cols.append(f"{us:7.1f} us {flops / us / 1e6:6.1f} TF/s")
    print(f"{name:32s} " + " ".join(f"{c:>22s}" for c in cols))
