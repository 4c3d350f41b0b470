"""Dump the operand pieces (workspace head) the Gram-form backward leaves for one small batch: A/B of two library builds.
   PCRL_HIP_LIB=... python tools/probes/dump_pieces.py out.npy [--c1 128 --seg 1 --pos 0]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from helpers import make_encoder_weights, make_obs
from pointcloud_rl_amd import hip
ap = argparse.ArgumentParser(); ap.add_argument("out"); ap.add_argument("--c1", type=int, default=64); ap.add_argument("--seg", type=int, default=0)
ap.add_argument("--pos", type=int, default=0); ap.add_argument("--B", type=int, default=2); ap.add_argument("--N", type=int, default=300)
a = ap.parse_args()
dev = torch.device("cuda:0")
extra = {}
if a.seg: extra["seg"] = a.seg
if a.pos: extra["pos_encoding"] = a.pos
obs_np = make_obs(a.B, a.N, seed=7, **extra)
C = sum(v.shape[1] for v in obs_np.values())
w = {k: torch.from_numpy(v).to(dev) for k, v in make_encoder_weights(C, a.c1, 128, 256, seed=3).items()}
ew, _ = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=dev)
hip.encoder_pack_weights(ew, packed)
obs = {k: torch.from_numpy(v).to(dev) for k, v in obs_np.items()}
desc, keep = hip.make_cloud_desc(obs)
pooled, argmax = hip.encoder_fwd(desc, ew, packed)
gp = torch.from_numpy(np.random.RandomState(5).randn(a.B, 256).astype(np.float32)).to(dev)
need = ctypes.c_size_t()
hip.check(hip.lib().pcrl_encoder_bwd_workspace_bytes(a.B, ew.c_in, ew.c1, ew.c2, ew.c3, ctypes.byref(need)))
outs = []
for rep in range(2):
    ws = torch.zeros(need.value, dtype=torch.uint8, device=dev)
    flat, n_act = hip.encoder_bwd(desc, ew, packed, argmax, gp, pooled=pooled, want_n_active=True, workspace=ws)
    torch.cuda.synchronize()
    MB1, MB2, NP = a.c1 // 32, 4, 32
    total = (2 * MB2 + 2 * MB1 + 1) * NP * 256
    outs.append(ws[:4 * total * a.B].view(torch.float32).cpu().numpy().reshape(a.B, -1).copy())
    outs.append(flat.cpu().numpy().copy())
print("n_act", n_act.cpu().numpy(), "pieces equal between two runs:", np.array_equal(outs[0], outs[2]), "grads equal:", np.array_equal(outs[1], outs[3]))
np.save(a.out, outs[0]); np.save(a.out.replace(".npy", "_g.npy"), outs[1])
