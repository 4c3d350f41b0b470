"""INTEGRATION.md Level 2: the ctypes stub shown there (`pointnet_pool`), extracted from the markdown and executed as written."""
import os
import re
import types

import numpy as np
import pytest
import torch

from helpers import make_encoder_weights, make_obs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_level_2_ctypes_stub_of_integration_md_runs_as_written(cuda):
    """The block is exec'ed verbatim (cwd = repo root, as its relative library path assumes) with a stand-in for the reference's
    PointNet module (`self.conv.mlp.{conv0,conv1,norm1,conv2,norm2}` with Conv1d-shaped weights [out, in, 1], pointnet.py:107-109)
    and must return the C oracle's pooled values and first-index argmax bit for bit."""
    from oracle import c_oracle
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes, torch\n.*?def pointnet_pool\(self, inputs\):.*?)```", text, re.S).group(1)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        ns = {}
        exec(compile(block, "INTEGRATION.md:level-2", "exec"), ns)
        w = make_encoder_weights(6, 64, 128, 256, seed=11)
        obs_np = make_obs(5, 333, seed=12)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
        layer = lambda **kw: types.SimpleNamespace(**{k: t(v) for k, v in kw.items()})
        mlp = types.SimpleNamespace(conv0=layer(weight=w["w0"][..., None], bias=w["b0"]), conv1=layer(weight=w["w1"][..., None]),
                                    norm1=layer(weight=w["g1"], bias=w["be1"]), conv2=layer(weight=w["w2"][..., None]),
                                    norm2=layer(weight=w["g2"], bias=w["be2"]))
        module = types.SimpleNamespace(conv=types.SimpleNamespace(mlp=mlp))
        pooled, argmax = ns["pointnet_pool"](module, {k: t(v) for k, v in obs_np.items()})
        torch.cuda.synchronize()
    finally:
        os.chdir(cwd)
    pooled_ref, arg_ref = c_oracle.encoder_fwd(c_oracle.preprocess(obs_np), w)
    assert np.array_equal(argmax.cpu().numpy(), arg_ref)
    assert np.array_equal(pooled.cpu().numpy().view(np.uint32), pooled_ref.view(np.uint32))
