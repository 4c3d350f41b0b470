"""Generate tests/golden/ref_rotscaletrans.npz by RUNNING THE REFERENCE's GlobalRotScaleTrans (pcd_aug.py:125-227; build container
only), seeded: the 4x4 matrices it drew and the xyz it returned, for the shipped pn_rot / pn_shift style settings.

    python tools/gen_golden_rotscaletrans.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_rotscaletrans.npz")
CASES = {
    "rot_scale_trans": dict(rot_range=[-0.15, 0.15], rot_axis="z", scale_ratio_range=[0.9, 1.1], translation_range=[0.04, 0.0, 0.04], shift_height=False),
    "rot_y_only": dict(rot_range=0.5, rot_axis="y", scale_ratio_range=None, translation_range=None, shift_height=False),
    "shift_only": dict(rot_range=None, rot_axis="z", scale_ratio_range=None, translation_range=[0.1, 0.2, 0.3], shift_height=True),
}

if __name__ == "__main__":
    from pyrl.utils.augmentations.pcd_aug import GlobalRotScaleTrans
    g = np.random.RandomState(8)
    B, N = 5, 33
    xyz = g.randn(B, 3, N).astype(np.float32)
    out = {"in/xyz": xyz}
    for seed, (tag, kw) in enumerate(CASES.items()):
        torch.manual_seed(100 + seed)
        aug = GlobalRotScaleTrans(main_key="xyz", req_keys=["xyz"], **kw)
        res = aug({"xyz": torch.from_numpy(xyz.copy())})
        out[f"{tag}/seed"] = np.array(100 + seed)
        out[f"{tag}/mat"] = aug.infos.numpy()
        out[f"{tag}/out_xyz"] = np.asarray(res["xyz"])
    np.savez_compressed(OUT, **out)
    print(OUT, f"{os.path.getsize(OUT) / 1e3:.1f} KB", sorted(out))
