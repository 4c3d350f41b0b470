"""Shared synthetic-input builders for the tests (numpy RandomState, as SURVEY.md section 8d)."""
import numpy as np


def make_encoder_weights(C, c1, c2, c3, seed=0, scale_ln=True):
    g = np.random.RandomState(seed)
    def u(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return g.uniform(-b, b, size=shape).astype(np.float32)
    w = dict(w0=u((c1, C), C), b0=u((c1,), C), w1=u((c2, c1), c1), w2=u((c3, c2), c2))
    if scale_ln:   # non-trivial affine so gamma/beta indexing errors are visible
        w.update(g1=g.uniform(0.5, 1.5, c2).astype(np.float32), be1=g.uniform(-0.5, 0.5, c2).astype(np.float32),
                 g2=g.uniform(0.5, 1.5, c3).astype(np.float32), be2=g.uniform(-0.5, 0.5, c3).astype(np.float32))
    else:
        w.update(g1=np.ones(c2, np.float32), be1=np.zeros(c2, np.float32), g2=np.ones(c3, np.float32), be2=np.zeros(c3, np.float32))
    return w


def make_obs(B, N, seed=1, pos_encoding=0, seg=0, rgb=True):
    g = np.random.RandomState(seed)
    obs = {"xyz": g.randn(B, 3, N).astype(np.float32)}
    if rgb:
        obs["rgb"] = g.randint(0, 256, (B, 3, N)).astype(np.uint8)
    if pos_encoding:
        pe = np.zeros((B, pos_encoding, N), np.uint8)
        per = max(N // pos_encoding, 1)
        for f in range(pos_encoding):
            pe[:, f, f * per:(f + 1) * per] = 1
        obs["pos_encoding"] = pe
    if seg:
        obs["seg"] = g.rand(B, seg, N) < 0.3
    return obs
