"""HBM-roofline report for the memory-shaped kernels: stand-alone segmented max-pool,
xyz augmentation, fused Adam(+Polyak), replay gather.  Achieved GB/s = algorithmic bytes / HIP-event time,
against the 8 TB/s HBM3E peak of MI355X_MICROARCH.md.      python tools/bench_membound.py [--md out.md]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloud_rl_amd import hip
from pointcloud_rl_amd.replay import DeviceReplay
from pointcloud_rl_amd.synthetic import make_batch_np

dev = torch.device("cuda", 0)
PEAK = 8000.0


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md")
    args = ap.parse_args()
    rows = []

    def report(name, nbytes, us, note):
        gbs = nbytes / us / 1e3
        rows.append((name, nbytes / 1e6, us, gbs, gbs / PEAK, note))
        print(f"{name:44s} {nbytes / 1e6:9.1f} MB {us:9.1f} us {gbs:8.0f} GB/s  {100 * gbs / PEAK:5.1f}% of HBM peak   {note}")

    # stand-alone max-pool over a materialised feature tensor [B, c3, N] (K4 shape per GPU: 64 clouds x 256 ch x 8192 pts)
    for B, C, N in ((64, 256, 8192), (256, 256, 1024)):
        x = torch.randn(B, C, N, device=dev)
        us = timeit(lambda: hip.segmax_fwd(x))
        report(f"segmax_fwd [{B},{C},{N}]", x.numel() * 4 + B * C * 8, us, "read 4 B/elem, write value+index")
        g = torch.randn(B, C, device=dev)
        _, idx = hip.segmax_fwd(x)
        us = timeit(lambda: hip.segmax_bwd(g, idx, N))
        report(f"segmax_bwd [{B},{C},{N}]", x.numel() * 4 + B * C * 8, us, "write 4 B/elem (zero fill + scatter)")
        del x
    # xyz augmentation (jitter drawn in the kernel + affine), out of place: 12 B read + 12 B written per point
    B, N = 2048, 1200
    xyz = torch.randn(B, 3, N, device=dev)
    out = torch.empty_like(xyz)
    aff = torch.eye(3, 4, device=dev).repeat(B, 1, 1).contiguous()
    us = timeit(lambda: hip.augment_xyz(xyz, out=out, jitter_range=(-0.01, 0.01), seed=1, offset=0, affine=aff))
    report(f"augment_xyz jitter+affine [{B},3,{N}]", B * N * 24, us, "24 B/point")
    us = timeit(lambda: hip.augment_xyz(xyz, out=xyz, jitter_range=(-0.01, 0.01), seed=1, offset=0))
    report(f"augment_xyz jitter in place [{B},3,{N}]", B * N * 24, us, "12 B read + 12 B written per point")
    # fused Adam + grad norm (+ Polyak on the Q-head range): 28 B/param (+12 B/param on the target range)
    for n, tgt in ((2_273_112, True), (64_000_000, False)):
        p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
        v.abs_()
        step, norm = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
        ws = torch.empty(hip.adam_workspace_bytes(n), dtype=torch.uint8, device=dev)
        t = torch.randn(n, device=dev) if tgt else None
        us = timeit(lambda: hip.adam_step(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 1.0, step, norm, ws, target=t, target_begin=0,
                                          target_end=n if tgt else 0, tau=0.01))
        report(f"adam_step n={n}{' +polyak' if tgt else ''}", n * (28 + (12 if tgt else 0)), us, "p,g,m,v read; p,m,v written" + ("; target r/w + p" if tgt else ""))
    # replay gather: 2 x 15 B/point read + written per sampled transition
    B, N, cap = 256, 1024, 2048
    mem = DeviceReplay(cap, device=dev, seed=0)
    for lo in range(0, cap, 512):
        mem.push_batch(make_batch_np(512, N, 6, seed=lo))
    us = timeit(lambda: mem.sample(B))
    row = sum(v[0].numel() * v.element_size() for v in mem.storage.values())
    report(f"replay sample+gather B={B} N={N}", 2 * B * row, us, f"{row} B/transition read + written (host-launch bound)")
    if args.md:
        with open(args.md, "w") as f:
            f.write("| kernel | algorithmic MB | us | GB/s | fraction of 8 TB/s | bytes counted |\n|---|---|---|---|---|---|\n")
            for name, mb, us, gbs, frac, note in rows:
                f.write(f"| {name} | {mb:.1f} | {us:.1f} | {gbs:.0f} | {frac:.3f} | {note} |\n")


if __name__ == "__main__":
    main()
