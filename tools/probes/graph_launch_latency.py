"""Host -> device start latency of an eager launch, a 1-node hipGraph and a 41-node hipGraph (first node publishes a flag to
pinned host memory; the host spins on it).  python tools/probes/graph_launch_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloud_rl_amd import hip

dev = torch.device("cuda:0")
src = torch.ones(4, device=dev); dst = torch.zeros(4, device=dev)
host = torch.zeros(32).pin_memory(); view = host.numpy().view('uint32')
entries = [(src[i:], dst[i:], False) for i in range(4)]
fill = torch.zeros(1024, device=dev)


def first():
    hip.gather_scalars(entries, host_out=host)


def measure(fn, n=300):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        view[:4] = 0xFFFFFFFF
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        while (view[:4] == 0xFFFFFFFF).any():
            pass
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts = ts[50:]
    call = sorted(t[0] for t in ts)[len(ts) // 2] * 1e6
    seen = sorted(t[1] for t in ts)[len(ts) // 2] * 1e6
    return call, seen


def graph(nodes):
    g = torch.cuda.CUDAGraph()
    first(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        first()
        for _ in range(nodes - 1):
            hip.gather_scalars(entries)
    return g


print("eager launch        : call %.1f us, flag seen after %.1f us" % measure(first))
for n in (1, 2, 10, 41, 80):
    g = graph(n)
    print("graph of %2d nodes   : call %.1f us, first node's flag seen after %.1f us" % ((n,) + measure(g.replay)))
