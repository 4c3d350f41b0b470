"""ctypes loader for oracle/pcrl_oracle.c (TEST INFRASTRUCTURE, never imported by the product)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpcrl_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "pcrl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def preprocess(obs):
    """PointCloudBase.preprocess (pyrl/networks/backbones/pointnet.py:49-73) in numpy:
    xyz, rgb (uint8 -> /255.0), pos_encoding, seg concatenated on the channel axis -> [B, C, N] f32."""
    feats = [np.asarray(obs["xyz"], dtype=np.float32)]
    if "rgb" in obs:
        rgb = np.asarray(obs["rgb"])
        if rgb.dtype == np.uint8:
            rgb = rgb.astype(np.float32) / np.float32(255.0)
        feats.append(rgb.astype(np.float32))
    for key in ("pos_encoding", "seg"):
        if key in obs:
            feats.append(np.asarray(obs[key]).astype(np.float32))
    return np.ascontiguousarray(np.concatenate(feats, axis=-2), dtype=np.float32)


def encoder_fwd(feat, w, eps=1e-6, want_prepool=False):
    """feat [B,C,N] f32; w: dict with w0,b0,w1,g1,be1,w2,g2,be2 (reference state_dict shapes).
    Returns pooled [B,c3] f32, argmax [B,c3] int32 (, prepool [B,c3,N])."""
    feat = _f32(feat)
    B, C, N = feat.shape
    w0, b0, w1, g1, be1, w2, g2, be2 = [_f32(np.asarray(w[k]).reshape(np.asarray(w[k]).shape[0], -1) if np.asarray(w[k]).ndim > 1 else w[k])
                                        for k in ("w0", "b0", "w1", "g1", "be1", "w2", "g2", "be2")]
    c1, c2, c3 = w0.shape[0], w1.shape[0], w2.shape[0]
    assert w0.shape[1] == C and w1.shape[1] == c1 and w2.shape[1] == c2
    pooled = np.empty((B, c3), np.float32)
    argmax = np.empty((B, c3), np.int32)
    prepool = np.empty((B, c3, N), np.float32) if want_prepool else None
    rc = lib().pcrl_oracle_encoder_fwd_f32(
        _p(feat), B, C, N, c1, c2, c3, _p(w0), _p(b0), _p(w1), _p(g1), _p(be1), _p(w2), _p(g2), _p(be2),
        ctypes.c_float(eps), _p(pooled), _p(argmax), _p(prepool) if want_prepool else None)
    assert rc == 0, rc
    return (pooled, argmax, prepool) if want_prepool else (pooled, argmax)


def point_preacts(x, w, eps=1e-6):
    """x [n, C] f32 points (preprocessed features); returns the values the three per-point ReLUs decide on, in the HIP
    kernels' summation order: (pre0 [n, c1], pre1 [n, c2], pre2 [n, c3])."""
    x = _f32(x)
    n, C = x.shape
    w0, b0, w1, g1, be1, w2, g2, be2 = [_f32(np.asarray(w[k]).reshape(np.asarray(w[k]).shape[0], -1) if np.asarray(w[k]).ndim > 1 else w[k])
                                        for k in ("w0", "b0", "w1", "g1", "be1", "w2", "g2", "be2")]
    c1, c2, c3 = w0.shape[0], w1.shape[0], w2.shape[0]
    assert w0.shape[1] == C
    pre = [np.empty((n, c), np.float32) for c in (c1, c2, c3)]
    rc = lib().pcrl_oracle_point_preacts_f32(_p(x), n, C, c1, c2, c3, _p(w0), _p(b0), _p(w1), _p(g1), _p(be1), _p(w2), _p(g2), _p(be2),
                                             ctypes.c_float(eps), _p(pre[0]), _p(pre[1]), _p(pre[2]))
    assert rc == 0, rc
    return tuple(pre)


def segmax(x):
    x = _f32(x)
    B, c, N = x.shape
    out = np.empty((B, c), np.float32)
    idx = np.empty((B, c), np.int32)
    assert lib().pcrl_oracle_segmax_f32(_p(x), B, c, N, _p(out), _p(idx)) == 0
    return out, idx
