"""Micro-benchmark of the fused encoder forward (dev tool; bench.py is the contract benchmark)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import make_encoder_weights, make_obs
from pointcloud_rl_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256); ap.add_argument("--N", type=int, default=1024)
ap.add_argument("--c1", type=int, default=64); ap.add_argument("--seg", type=int, default=0)
ap.add_argument("--iters", type=int, default=50); ap.add_argument("--bf16", action="store_true"); ap.add_argument("--split", action="store_true"); ap.add_argument("--no-pooled", action="store_true"); ap.add_argument("--fwd-only", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
obs_np = make_obs(a.B, a.N, seed=1, seg=a.seg)
C = sum(v.shape[1] for v in obs_np.values())
w = {k: torch.from_numpy(v).to(dev) for k, v in make_encoder_weights(C, a.c1, 128, 256).items()}
ew, _ = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=dev)
hip.encoder_pack_weights(ew, packed)
obs = {k: torch.from_numpy(v).to(dev) for k, v in obs_np.items()}
desc, keep = hip.make_cloud_desc(obs)
for _ in range(5): hip.encoder_fwd(desc, ew, packed, bf16=a.bf16, split=a.split)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters): hip.encoder_fwd(desc, ew, packed, bf16=a.bf16, split=a.split)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
flop = 2.0 * (C * a.c1 + a.c1 * 128 + 128 * 256) * a.B * a.N
print(f"encoder_fwd B={a.B} N={a.N} C={C} c1={a.c1}: {ms*1e3:.1f} us  {flop/ms/1e9:.1f} TFLOP/s ({flop/ms/1e9/157.3*100:.1f}% of 157.3 fp32 MFMA peak)")

if a.fwd_only:
    sys.exit(0)
if a.split:
    p0, a0 = hip.encoder_fwd(desc, ew, packed)
    p1, a1 = hip.encoder_fwd(desc, ew, packed, split=True)
    d = (p0 - p1).abs()
    print(f"split vs fp32: max |pooled diff| {float(d.max()):.3e}, mean {float(d.mean()):.3e}, argmax differs on {int((a0 != a1).sum())} of {a0.numel()} "
          f"(largest value gap there {float((p0 - p1).abs()[a0 != a1].max()) if (a0 != a1).any() else 0.0:.3e})")
# ---- backward (sparse exact backward through the max-pool) ----
pooled, argmax = hip.encoder_fwd(desc, ew, packed, bf16=a.bf16, split=a.split)
gp = torch.randn_like(pooled)
import ctypes
need = ctypes.c_size_t()
hip.check(hip.lib().pcrl_encoder_bwd_workspace_bytes(a.B, ew.c_in, ew.c1, ew.c2, ew.c3, ctypes.byref(need)))
ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
out = torch.empty(hip.encoder_num_grads(ew), device=dev)
for _ in range(5): hip.encoder_bwd(desc, ew, packed, argmax, gp, workspace=ws, out=out, bf16=a.bf16, split=a.split, pooled=None if a.no_pooled else pooled)
torch.cuda.synchronize()
e0.record()
for _ in range(a.iters): hip.encoder_bwd(desc, ew, packed, argmax, gp, workspace=ws, out=out, bf16=a.bf16, split=a.split, pooled=None if a.no_pooled else pooled)
e1.record(); torch.cuda.synchronize()
print(f"encoder_bwd B={a.B} N={a.N}: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us (points + wgrad + reduce kernels)")
