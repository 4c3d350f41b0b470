"""One GEMM group of tools/bench_gemm.py launched N times (for rocprofv3 --pmc passes: every gemm_f32_kernel dispatch is this shape).
    GEMM_M=256 python tools/probes/gemm_one.py "dh1 x2" [launches] [tile64_min]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv, name = [sys.argv[0]] + sys.argv[2:], sys.argv[1]
import torch
from pointcloud_rl_amd import hip
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_gemm.py")).read()
ns = {"__file__": os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_gemm.py")}
exec(compile(src.split("def time_group")[0], "bench_gemm.py", "exec"), ns)       # the shape table only
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2:
    hip.gemm_set_tile64_min(int(sys.argv[2]))
for nm, descs in ns["shapes"]():
    if nm == name:
        for _ in range(n):
            hip.gemm_group(descs)
        torch.cuda.synchronize()
        flops = sum(2.0 * d.M * d.N * d.K * d.batch for d in descs)
        print(name, "launched", n, "x", flops / 1e9, "GFLOP")
        break
else:
    raise SystemExit(f"no shape named {name!r}")
