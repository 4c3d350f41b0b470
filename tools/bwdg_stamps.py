"""Per-phase shader-clock cycles of the Gram-form encoder backward (points kernel: one tile's chain; wgrad kernel: one wave's task
list), from a library built with -DPCRL_BWDG_STAMPS on encoder_bwd_gram_f32.hip (see encoder_bwd_gram.h):
    PCRL_HIP_LIB=_ab/libpcrl_hip_stamps.so python tools/bwdg_stamps.py --B 256"""
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
n_tiles = int(((n_act.cpu().numpy() + 31) // 32).sum())
n_tiles = min(n_tiles, 16384)
n_waves = min(a.B * 8, 4096)
tb = (ctypes.c_ulonglong * (12 * n_tiles))(); wb = (ctypes.c_ulonglong * (8 * n_waves))()
hip.check(hip.lib().pcrl_debug_bwdg_stamps(tb, n_tiles, wb, n_waves))
st = np.frombuffer(tb, dtype=np.uint64).reshape(n_tiles, 12).astype(np.int64)[:, :9]
d = np.diff(st, axis=1)
names = ["point load + conv0 (+ x, h0 stores)", "conv1 + LN1 (+ h1 stores)", "mu = s.h1", "loop over the owned channels", "q = M h1, rstd2",
         "dH1", "LN1 backward (+ dz1 stores)", "dH0 = W1^T dz1 (+ dz0 stores)"]
tot = st[:, 8] - st[:, 0]
st9 = np.frombuffer(tb, dtype=np.uint64).reshape(n_tiles, 12).astype(np.int64)[:, 9]
print(f"  dH0: MFMA part (to the last MFMA's issue) median {np.median(st9 - st[:, 7]):.0f}, dz0 stores {np.median(st[:, 8] - st9):.0f}")
print(f"B={a.B}: {n_tiles} tiles (mean active points {float(n_act.float().mean()):.0f}), chain median {np.median(tot):.0f} cycles (min {tot.min()}, max {tot.max()})")
for i, n in enumerate(names):
    print(f"  {n:40s} median {np.median(d[:, i]):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f} cycles  {100 * np.median(d[:, i]) / np.median(tot):5.1f} %")
# between two tiles of one wave (item -> item + 4 * 256): the loop header (cloud search, n_act / act / own loads) and whatever the
# last stores of the previous tile hold up
stride = 4 * 256
if n_tiles > stride:
    gap = st[stride:n_tiles, 0] - st[:n_tiles - stride, 8]
    print(f"  between a wave's consecutive tiles (end of chain -> first stamp of the next): median {np.median(gap):.0f}  p90 {np.percentile(gap, 90):.0f} cycles")
ws = np.frombuffer(wb, dtype=np.uint64).reshape(n_waves, 8).astype(np.int64)[:, :7]
dw = np.diff(ws, axis=1)
wn = ["norm1 sums + staging (to the barrier)", "dW1 blocks", "G block pairs", "(v, u) / dW0 blocks", "wait at the barrier", "S rows"]
wt = ws[:, 6] - ws[:, 0]
print(f"wgrad: wave chain median {np.median(wt):.0f} cycles (max {wt.max()})")
for i, n in enumerate(wn):
    print(f"  {n:40s} median {np.median(dw[:, i]):9.0f}  max {dw[:, i].max():9.0f} cycles")
for wv in range(8):
    sel = dw[wv::8]
    print(f"  wave {wv}: " + "  ".join(f"{np.median(sel[:, i]):7.0f}" for i in range(6)))
