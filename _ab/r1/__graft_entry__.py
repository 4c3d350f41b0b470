"""Driver hooks: build() compiles every native piece for gfx950; smoke() runs one tiny update on cuda:0."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build():
    """hipcc --offload-arch=gfx950 for every .hip file -> pointcloud_rl_amd/libpcrl_hip.so (in-tree),
    gcc for the CPU oracle (test infrastructure) -> oracle/_build/libpcrl_oracle.so; then import the package."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "pointcloud_rl_amd", "csrc"), "-j4"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    import pointcloud_rl_amd  # noqa: F401
    from pointcloud_rl_amd import _lib
    _lib.lib()     # loads the shared object: every symbol resolves
    import pointcloud_rl_amd.methods  # noqa: F401
    import pointcloud_rl_amd.networks  # noqa: F401


def smoke():
    """One small SAC update (HIP encoder forward + backward through the C ABI) on cuda:0, checked against
    the oracle: the PyTorch-CPU restatement of the reference's update on the same batch and noise."""
    import torch
    from oracle import torch_ref
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    dev = torch.device("cuda:0")
    B, N, A = 8, 200, 6
    cfg = configs.sac_dmc(6, A, B, head_hidden=64)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg)
    params = {n: p.detach().clone() for n, p in agent.named_parameters()}
    ref = torch_ref.RefAgent(params, kind="sac", gamma=agent.gamma, alpha=0.1, target_entropy=agent.target_entropy,
                             update_coeff=agent.update_coeff["default"], mirror_redundancy=False)
    agent = agent.to(dev)
    mem = SyntheticReplay(B, N, A, seed=3, device=dev)
    cpu_batch = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v))
                 for k, v in mem.batch_np.items()}
    g = torch.Generator().manual_seed(1)
    for u in (1, 2):
        eps = [torch.randn(B, A, generator=g) for _ in range(2 if u % 2 == 0 else 1)]
        agent.actor.head.noise_override = [e.to(dev) for e in eps]
        got = agent.update_parameters(mem, u)
        want = ref.update_parameters(cpu_batch, u, eps)
        for k, v in want.items():
            assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (u, k, got[k], v)
    for n, p in agent.named_parameters():
        assert torch.allclose(p.detach().cpu(), ref.P[n].detach(), atol=1e-5, rtol=0), n
    print("smoke ok:", {k: round(v, 5) if isinstance(v, float) else v for k, v in got.items()})


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
