"""Generate tests/golden/ref_replay_tstep.npz by RUNNING THE REFERENCE's ReplayMemory with `TStepTransition` (build container only):
two workers' interleaved episodes pushed into a ring that wraps, then seeded samples with horizon 3 (with and without replacement)
and horizon -1 (whole episodes, padded).  tests/test_aux_aug_acting_gpu.py feeds the same pushes to DeviceReplay and compares every
sampled key and the validity mask.  (numpy >= 1.24 removed `np.int`, which the reference's class still names: restored here, in
the generator process only.)

    python tools/gen_golden_replay_tstep.py
"""
import os
import sys

import numpy as np

if not hasattr(np, "int"):
    np.int = int
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
for m in ("gym.envs", "gym.wrappers", "gym.core", "gym.envs.registration", "h5py", "cv2", "imageio", "matplotlib", "matplotlib.pyplot"):
    ref_stubs._stub(m)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_replay_tstep.npz")


def flat(d, prefix=""):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from flat(v, prefix + k + "/")
        else:
            yield prefix + k, np.asarray(v)


if __name__ == "__main__":
    from pyrl.env.replay_buffer import ReplayMemory
    out = {}
    cap, N, A = 48, 4, 2
    g = np.random.RandomState(1)
    pushes = []
    step = 0
    for i in range(4):                       # 4 x 16 into a ring of 48: the fourth push wraps; workers 0 / 1 alternate, episodes of 7 / 5 steps
        n = 16
        worker = np.array([[(step + k) % 2] for k in range(n)], dtype=np.int32)
        t_in_ep = np.array([(step + k) // 2 for k in range(n)])
        ep_done = np.array([[(t_in_ep[k] + 1) % (7 if worker[k, 0] == 0 else 5) == 0] for k in range(n)])
        step += n
        pushes.append(dict(obs=dict(xyz=g.randn(n, 3, N).astype(np.float32), rgb=g.randint(0, 255, (n, 3, N)).astype(np.uint8)),
                           next_obs=dict(xyz=g.randn(n, 3, N).astype(np.float32), rgb=g.randint(0, 255, (n, 3, N)).astype(np.uint8)),
                           actions=g.randn(n, A).astype(np.float32), rewards=g.randn(n, 1).astype(np.float32),
                           dones=ep_done.copy(), episode_dones=ep_done, worker_indices=worker,
                           is_truncated=np.zeros((n, 1), dtype=np.bool_)))
        for k, v in flat(pushes[-1]):
            out[f"push{i}/{k}"] = v
    for tag, kw in (("h3_with", dict(horizon=3, with_replacement=True)), ("h3_without", dict(horizon=3, with_replacement=False)),
                    ("episode_with", dict(horizon=-1, with_replacement=True))):
        mem = ReplayMemory(capacity=cap, sampling_cfg=dict(type="TStepTransition", seed=11, **kw))
        for items in pushes:
            mem.push_batch({k: (dict(v) if isinstance(v, dict) else v) for k, v in items.items()})
        out[f"{tag}/len_units"] = np.array([len(mem), len(mem.sampling)])
        for s in range(6):
            batch = mem.sample(5)
            for k, v in flat(batch.memory if hasattr(batch, "memory") else batch):
                out[f"{tag}/sample{s}/{k}"] = v
    np.savez_compressed(OUT, **out)
    print(OUT, f"{os.path.getsize(OUT) / 1e3:.1f} KB", len(out), "arrays")
