"""Generate tests/golden/ref_replay.npz by RUNNING THE REFERENCE's ReplayMemory (build container only): pushes with a wrap-around,
then samples with `OneStepTransition` (with and without replacement, seeded).  The fixture holds the pushed items and what the
reference returned; tests/test_aux_aug_acting_gpu.py feeds the same pushes to DeviceReplay and compares every sampled key.

    python tools/gen_golden_replay.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
for m in ("gym.envs", "gym.wrappers", "gym.core", "gym.envs.registration", "h5py", "cv2", "imageio", "matplotlib", "matplotlib.pyplot"):
    ref_stubs._stub(m)        # pyrl.env imports its simulators' and video writers' dependencies at package import time

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_replay.npz")


def flat(d, prefix=""):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from flat(v, prefix + k + "/")
        else:
            yield prefix + k, np.asarray(v)


if __name__ == "__main__":
    from pyrl.env.replay_buffer import ReplayMemory
    out = {}
    cap, N, A = 40, 5, 2
    g = np.random.RandomState(0)
    pushes = []
    for i in range(3):                       # 3 x 16 into a ring of 40: the third push wraps
        n = 16
        pushes.append(dict(obs=dict(xyz=g.randn(n, 3, N).astype(np.float32), rgb=g.randint(0, 255, (n, 3, N)).astype(np.uint8)),
                           next_obs=dict(xyz=g.randn(n, 3, N).astype(np.float32), rgb=g.randint(0, 255, (n, 3, N)).astype(np.uint8)),
                           actions=g.randn(n, A).astype(np.float32), rewards=g.randn(n, 1).astype(np.float32),
                           dones=g.rand(n, 1) < 0.1, episode_dones=g.rand(n, 1) < 0.1))
        for k, v in flat(pushes[-1]):
            out[f"push{i}/{k}"] = v
    for tag, kw in (("with", dict(with_replacement=True)), ("without", dict(with_replacement=False))):
        mem = ReplayMemory(capacity=cap, sampling_cfg=dict(type="OneStepTransition", seed=7, **kw))
        for items in pushes:
            mem.push_batch({k: (dict(v) if isinstance(v, dict) else v) for k, v in items.items()})
        out[f"{tag}/len_position"] = np.array([len(mem), mem.position])
        for s in range(10):
            batch = mem.sample(6)
            for k, v in flat(batch.memory if hasattr(batch, "memory") else batch):
                out[f"{tag}/sample{s}/{k}"] = v
    np.savez_compressed(OUT, **out)
    print(OUT, f"{os.path.getsize(OUT) / 1e3:.1f} KB", len(out), "arrays")
