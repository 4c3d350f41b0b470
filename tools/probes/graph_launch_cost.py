"""Host time of hipGraphLaunch (torch.cuda.CUDAGraph.replay) against the number of kernel nodes, and the time until the FIRST node has run
(a pinned flag written by node 1): is the launch's cost fixed or per node, and does the device start before the call returns?
    python tools/probes/graph_launch_cost.py"""
import time
import numpy as np, torch

dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
flag = torch.zeros(1, dtype=torch.int32).pin_memory()
flag_dev_view = None
now = time.perf_counter_ns
print("nodes  launch-call us (median)   call start -> chain done us (median)")
for nodes in (1, 2, 4, 8, 17, 29, 58):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(nodes):
                x.add_(1.0)
    torch.cuda.synchronize()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    calls, dones = [], []
    for _ in range(300):
        t0 = now()
        g.replay()
        t1 = now()
        torch.cuda.synchronize()
        t2 = now()
        calls.append(t1 - t0)
        dones.append(t2 - t0)
    print(f"{nodes:5d}  {np.median(calls) / 1e3:10.2f}                {np.median(dones) / 1e3:10.2f}")
