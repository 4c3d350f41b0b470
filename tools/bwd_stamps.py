"""Per-phase shader-clock cycles of one tile's chain in the encoder backward (tile-mode kernel), from a library built with
-DPCRL_BWD_STAMPS on encoder_bwd_f32.hip (see encoder_bwd_impl.h):   PCRL_HIP_LIB=<that .so> python tools/bwd_stamps.py --B 32"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import make_encoder_weights, make_obs
from pointcloud_rl_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32); ap.add_argument("--N", type=int, default=1024); ap.add_argument("--c1", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda:0")
obs_np = make_obs(a.B, a.N, seed=1)
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
    hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=pooled)
torch.cuda.synchronize()
n_tiles = int(sum((len(set(r.tolist())) + 31) // 32 for r in argmax.cpu()))
buf = (ctypes.c_ulonglong * (8 * n_tiles))()
hip.check(hip.lib().pcrl_debug_bwd_stamps(buf, n_tiles))
st = np.frombuffer(buf, dtype=np.uint64).reshape(n_tiles, 8).astype(np.int64)
d = np.diff(st, axis=1)
names = ["conv0 + point load", "conv1 + LN1 (+ xhat1 / h1 stores)", "conv2 + LN2", "pool / LN2 backward (+ dz2 stores)", "dH1 = W2^T dz2 (MFMA)",
         "LN1 backward (+ dz1 stores)", "dH0 = W1^T dz1 (+ dz0 stores)"]
tot = st[:, 7] - st[:, 0]
print(f"B={a.B}: {n_tiles} tiles, chain median {np.median(tot):.0f} cycles (min {tot.min()}, max {tot.max()})")
for i, n in enumerate(names):
    print(f"  {n:40s} {np.median(d[:, i]):9.0f} cycles  {100 * np.median(d[:, i]) / np.median(tot):5.1f} %")
