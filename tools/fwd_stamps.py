"""Per-phase shader-clock cycles of the encoder forward's tile loop, from a library whose encoder_fwd.hip was built with
-DPCRL_FWD_STAMPS:    PCRL_HIP_LIB=_ab/libpcrl_hip_fstamps.so python tools/fwd_stamps.py [--bf16] [--c1 128 --N 1200 --seg 1]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import make_encoder_weights, make_obs
from pointcloud_rl_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256); ap.add_argument("--N", type=int, default=1024); ap.add_argument("--c1", type=int, default=64)
ap.add_argument("--seg", type=int, default=0); ap.add_argument("--bf16", action="store_true"); ap.add_argument("--split", action="store_true")
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
for _ in range(3):
    hip.encoder_fwd(desc, ew, packed, bf16=a.bf16, split=a.split)
torch.cuda.synchronize()
rows = 256 * 8 * 8
buf = (ctypes.c_ulonglong * (8 * rows))()
hip.check(hip.lib().pcrl_debug_fwd_stamps(buf, rows))
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8, 8).astype(np.int64)     # [wg][wave][tile round][stamp]
tiles = (a.N + 31) // 32
rounds = (tiles + 7) // 8
names = ["point load + conv0", "conv1 (MFMA)", "LayerNorm-1 + ReLU", "conv2 (MFMA)", "LayerNorm-2", "max-pool"]
for r in range(min(rounds, 8)):
    sel = st[:min(a.B, 256), :, r, :7]
    sel = sel[(sel[..., 6] > sel[..., 0]) & (sel[..., 0] > 0)]
    if not len(sel):
        continue
    d = np.diff(sel, axis=1)
    tot = sel[:, 6] - sel[:, 0]
    print(f"tile round {r}: {len(sel)} tiles, median {np.median(tot):.0f} cycles: " + "  ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(names)))
first, last = st[:min(a.B, 256), :, 0, 0], st[:min(a.B, 256), :, rounds - 1, 6]
ok = (first > 0) & (last > first)
print(f"first stamp -> last stamp per wave: median {np.median((last - first)[ok]):.0f} cycles")
