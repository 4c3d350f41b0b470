"""Generate tests/golden/ref_downsample.npz by RUNNING THE REFERENCE's RandomDownSample (pcd_aug.py:231-268; build container
only): the index it drew (its RNG stream is torch's CPU generator, so the test injects it) and the tensors it returned.

    python tools/gen_golden_downsample.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_downsample.npz")

if __name__ == "__main__":
    from pyrl.utils.augmentations.pcd_aug import RandomDownSample
    g = np.random.RandomState(4)
    B, N = 3, 40
    obs = dict(xyz=torch.from_numpy(g.randn(B, 3, N).astype(np.float32)), rgb=torch.from_numpy(g.randint(0, 255, (B, 3, N)).astype(np.uint8)),
               seg=torch.from_numpy(g.rand(B, 1, N) < 0.4))
    out = {f"in/{k}": v.numpy() for k, v in obs.items()}
    for tag, kw in (("ratio", dict(drop_ratio=0.3, fixed_ratio=True)), ("maxpts", dict(max_num_points=17))):
        torch.manual_seed(3)
        aug = RandomDownSample(main_key="xyz", req_keys=["xyz", "rgb", "seg"], **kw)
        res = aug({k: v.clone() for k, v in obs.items()})
        out[f"{tag}/index"] = np.asarray(aug.infos[1])
        out[f"{tag}/n"] = np.array(aug.infos[0])
        for k in obs:
            out[f"{tag}/out/{k}"] = np.asarray(res[k])
    np.savez_compressed(OUT, **out)
    print(OUT, f"{os.path.getsize(OUT) / 1e3:.1f} KB", {k: v.shape for k, v in out.items() if k.endswith("xyz") or k.endswith("index")})
