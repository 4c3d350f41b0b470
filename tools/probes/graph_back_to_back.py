"""Does a hipGraph launched BEHIND a running one start without a gap?  A 41-node graph (small kernels + one long fill so that a
replay lasts ~300 us) is replayed 300 times (a) with the host waiting for each replay's last node before launching the next
(what `update_parameters` does: it returns the step's metrics), (b) back to back.  python tools/probes/graph_back_to_back.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloud_rl_amd import hip

dev = torch.device("cuda:0")
src = torch.ones(4, device=dev); dst = torch.zeros(4, device=dev)
host = torch.zeros(32).pin_memory(); view = host.numpy().view('uint32')
entries = [(src[i:], dst[i:], False) for i in range(4)]
big = torch.zeros(int(os.environ.get("FILL_MB", "512")) << 18, device=dev)


def body(nodes):
    for _ in range(nodes - 2):
        hip.gather_scalars(entries)
    big.add_(1.0)
    hip.gather_scalars(entries, host_out=host)      # last node publishes to pinned memory


def graph(nodes):
    g = torch.cuda.CUDAGraph()
    body(nodes); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        body(nodes)
    return g


for nodes in (41, 12):
    g = graph(nodes)
    n = 300
    for mode in ("wait", "back-to-back", "one behind"):
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if mode == "wait":
                view[:4] = 0xFFFFFFFF
            g.replay()
            if mode == "wait":
                while (view[:4] == 0xFFFFFFFF).any():
                    pass
            elif mode == "one behind" and i:
                ev_prev.synchronize()
            if mode == "one behind":
                ev_prev = torch.cuda.Event(); ev_prev.record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e6
        print("graph of %2d nodes, %-13s: %.1f us per replay" % (nodes, mode, dt))
