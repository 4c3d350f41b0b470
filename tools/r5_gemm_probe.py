"""Round 5: kernel durations (rocprofv3 kernel trace, not event averages) of the heads' 1 024-wide GEMM shapes against M and K.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp -- python3 tools/r5_gemm_probe.py run gpurun_out/gp/labels.txt
    python3 tools/r5_gemm_probe.py fold gpurun_out/gp gpurun_out/gp/labels.txt

Every shape is launched REPS times back to back; the fold step groups the trace's gemm_f32_kernel rows in start order.  K = 0 is the
kernel's fixed cost (dispatch, descriptor, split-K reduce, epilogue), the slope in K what the k loop costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 30
H = 1024


def shapes(torch, hip, dev):
    def t(*s):
        return torch.randn(*s, device=dev)
    for M in [int(x) for x in os.environ.get("GEMM_MS", "32,64,128,256").split(",")]:
        h4, W4 = t(4, M, H), t(4, H, H)
        for heads in (1, 2, 4):
            for K in (0, 32, 256, 1024):
                yield f"fwd1 M{M} x{heads} K{K}", [hip.gemm_desc(h4, W4, t(heads, M, H), M, H, K, (H, 1), (1, H), H, relu=True, batch=heads,
                                                                 batch_strides=(M * H, H * H, M * H, 0, 0))]
        two = [hip.gemm_desc(h4[2 * z:], W4[2 * z:], t(2, M, H), M, H, H, (H, 1), (1, H), H, relu=True, batch=2, batch_strides=(M * H, H * H, M * H, 0, 0)) for z in range(2)]
        yield f"fwd1 M{M} 2p x2", two
        h, W1 = t(2, M, H), t(2, H, H)
        dW = hip.gemm_desc(h, h, t(2, H, H + 1), H, H + 1, M, (1, H), (H, 1), H + 1, ones_col=H, batch=2, batch_strides=(M * H, M * H, H * (H + 1), 0, 0))
        dX = hip.gemm_desc(h, W1, t(2, M, H), M, H, H, (H, 1), (H, 1), H, mask=h, ld_mask=H, batch=2, batch_strides=(M * H, H * H, M * H, 0, M * H))
        yield f"dh1 M{M} x2", [dX]
        yield f"dW1 M{M} x2", [dW]
        yield f"dW1|dh1 M{M} x2", [dW, dX]
        dW1 = hip.gemm_desc(h, h, t(1, H, H + 1), H, H + 1, M, (1, H), (H, 1), H + 1, ones_col=H)
        dX1 = hip.gemm_desc(h, W1, t(1, M, H), M, H, H, (H, 1), (H, 1), H, mask=h, ld_mask=H)
        yield f"dh1 M{M} x1", [dX1]
        yield f"dW1|dh1 M{M} x1", [dW1, dX1]
        X56, W0 = t(2, M, 56), t(2, H, 56)
        yield f"fwd0 56->1024 M{M} x2", [hip.gemm_desc(X56, W0, t(2, M, H), M, H, 56, (56, 1), (1, 56), H, relu=True, batch=2, batch_strides=(M * 56, H * 56, M * H, 0, 0))]
        yield f"dX0 1024->50 M{M} x2", [hip.gemm_desc(h, W0, t(2, M, 52), M, 50, H, (H, 1), (56, 1), 52, batch=2, batch_strides=(M * H, H * 56, M * 52, 0, 0))]


def run(label_path):
    import torch
    from pointcloud_rl_amd import hip
    dev = torch.device("cuda", 0)
    labels = []
    for name, descs in shapes(torch, hip, dev):
        flops = sum(2.0 * d.M * d.N * d.K * d.batch for d in descs)
        for _ in range(REPS):
            hip.gemm_group(descs)
        torch.cuda.synchronize()
        labels.append(f"{name}\t{flops}")
    open(label_path, "w").write("\n".join(labels) + "\n")


def fold(trace_dir, label_path):
    import csv, glob
    f = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "gemm_f" in r["Kernel_Name"]))
    labels = [l.split("\t") for l in open(label_path).read().strip().splitlines()]
    assert len(rows) == REPS * len(labels), (len(rows), len(labels))
    for i, (name, flops) in enumerate(labels):
        d = sorted((e - s) / 1e3 for s, e in rows[i * REPS + 5:(i + 1) * REPS])
        med = d[len(d) // 2]
        print(f"{name:28s} median {med:6.2f} us  min {d[0]:6.2f}  {float(flops) / med / 1e6:6.1f} TF/s")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        fold(sys.argv[2], sys.argv[3])
