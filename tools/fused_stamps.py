"""Per-phase shader-clock cycles of the team kernel of the encoder backward (csrc/encoder_bwd_fused.h), every wave of every tile, from a
library whose encoder_bwd_gram_f32.hip was built with -DPCRL_BWDG_STAMPS:
    PCRL_HIP_LIB=_ab/stamps/libpcrl_hip.so python tools/fused_stamps.py --B 256"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import make_encoder_weights, make_obs
from pointcloud_rl_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256); ap.add_argument("--N", type=int, default=1024); ap.add_argument("--c1", type=int, default=64)
ap.add_argument("--seg", type=int, default=0)
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
pooled, argmax = hip.encoder_fwd(desc, ew, packed)
gp = torch.randn_like(pooled)
for _ in range(3):
    flat, n_act = hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=pooled, want_n_active=True)
torch.cuda.synchronize()
n_tiles = min(int(((n_act.cpu().numpy() + 31) // 32).sum()), 8192)
buf = (ctypes.c_ulonglong * (96 * n_tiles))()
hip.check(hip.lib().pcrl_debug_fused_stamps(buf, n_tiles))
full = np.frombuffer(buf, dtype=np.uint64).reshape(n_tiles, 4, 24).astype(np.int64)
st = full[:, :, :15]
names = ["conv0 (the point arrived with the previous tile)", "conv1 (block)", "LN1 partial statistics -> B1", "h0, x | 1 transposes; statistics",
         "xhat, h1, transpose -> B3", "owned channels, first pass", "q = Mc h1, B4, second pass, next tile's loads", "dH1, LN1 backward sums",
         "G blocks, v / u", "wait at B5", "dz1, transpose -> B6", "dH0 (waves < c1/32), dW1 blocks", "wait at B7", "dW0 blocks"]
d = np.diff(st, axis=2)
tot = st[:, :, 14] - st[:, :, 0]
print(f"B={a.B} c1={a.c1}: {n_tiles} tiles; tile (stamp 0 -> 14) median {np.median(tot):.0f} cycles (min {tot.min()}, max {tot.max()})")
for i, n in enumerate(names):
    per = "  ".join(f"{np.median(d[:, wv, i]):7.0f}" for wv in range(4))
    print(f"  {n:52s} waves: {per}   ({100 * np.median(d[:, :, i]) / np.median(tot):4.1f} %)")
# inside the longest phase (stamps 15 .. 17): q done -> B4 -> second pass done -> next tile's point loads issued
for a_, b_, n in ((6, 15, 'q = Mc h1 + its share of h1.q'), (15, 16, 'wait at B4'), (16, 17, 'second pass (rows of dW2, norm2 sums)'), (17, 7, "next tile's point loads")):
    print(f"    {n:50s} waves: " + "  ".join(f"{np.median(full[:, wv, b_] - full[:, wv, a_]):7.0f}" for wv in range(4)))
grid = int(os.environ.get("FUSED_GRID", "256"))
if n_tiles > grid:
    gap = st[grid:n_tiles, 0, 0] - st[:n_tiles - grid, 0, 14]
    print(f"  between a workgroup's consecutive tiles (stamp 14 -> next stamp 0): median {np.median(gap):.0f}  p90 {np.percentile(gap, 90):.0f}")
