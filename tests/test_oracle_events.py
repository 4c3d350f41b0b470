"""CPU check of the event locator the full-size parity test relies on (tests/test_fullsize_parity_gpu.py::_locate_encoder_events):
a gradient that differs from the restatement's by exactly one flipped near-zero ReLU decision (plus rounding noise) must be
explained by exactly that event, and the patched restatement gradient must then agree everywhere."""
import types

import numpy as np
import torch

from helpers import make_encoder_weights, make_obs
from test_fullsize_parity_gpu import EVENT_TAU, _locate_encoder_events

NAMES = {"conv0.weight": "w0", "conv0.bias": "b0", "conv1.weight": "w1", "norm1.weight": "g1", "norm1.bias": "be1",
         "conv2.weight": "w2", "norm2.weight": "g2", "norm2.bias": "be2"}


def _setup(B=6, N=220, seed=3):
    from oracle import torch_ref as T
    w = make_encoder_weights(6, 64, 128, 256, seed=seed)
    obs = {k: torch.from_numpy(v) for k, v in make_obs(B, N, seed=seed + 1).items()}
    P = {}
    for n, k in NAMES.items():
        t = torch.from_numpy(w[k]).clone()
        P[T.ENC + "conv.mlp." + n] = (t[..., None] if t.ndim == 2 else t).requires_grad_(True)
    f = T.ENC + "final_mlp."
    g = torch.Generator().manual_seed(seed)
    P[f + "0.weight"], P[f + "0.bias"] = torch.randn(50, 256, generator=g) * 0.05, torch.zeros(50)
    P[f + "1.weight"], P[f + "1.bias"] = torch.ones(50), torch.zeros(50)
    return T, P, obs, g


def test_cloud_encoder_reproduces_the_full_backward_and_flips_are_local():
    T, P, obs, g = _setup()
    keep = {}
    feat = T.pointnet_forward(P, obs, keep=keep)
    (feat * torch.randn(feat.shape, generator=g)).sum().backward()
    route = T.pointnet_prepool(P, obs).argmax(-1)
    total = {n: torch.zeros_like(P[T.ENC + "conv.mlp." + n]) for n in NAMES}
    for b in range(route.shape[0]):
        cloud = T.CloudEncoder(P, {k: v[b] for k, v in obs.items()}, route[b], keep["pooled"].grad[b])
        for n, v in cloud.grads().items():
            total[n] += v
    for n in NAMES:
        ref = P[T.ENC + "conv.mlp." + n].grad
        assert float((total[n] - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), n


def test_one_flipped_decision_is_located_and_removed():
    T, P, obs, g = _setup()
    # move a conv0 bias so that some routed point's conv0 pre-activation lands within EVENT_TAU of zero
    route = T.pointnet_prepool(P, obs).argmax(-1)
    b_ev, ch_ev = 2, 11
    pts = torch.unique(route[b_ev])
    with torch.no_grad():
        x = T.preprocess({k: v[b_ev:b_ev + 1, :, pts] for k, v in obs.items()})
        z0 = torch.nn.functional.conv1d(x, P[T.ENC + "conv.mlp.conv0.weight"], P[T.ENC + "conv.mlp.conv0.bias"])[0, ch_ev]
        sl = int(z0.abs().argmin())
        P[T.ENC + "conv.mlp.conv0.bias"][ch_ev] -= z0[sl] - 3e-6          # that unit's pre-activation becomes ~ +3e-6
    route = T.pointnet_prepool(P, obs).argmax(-1)                          # the routing of the modified network
    keep = {}
    feat = T.pointnet_forward(P, obs, route=route, keep=keep)
    (feat * torch.randn(feat.shape, generator=g)).sum().backward()
    cloud = T.CloudEncoder(P, {k: v[b_ev] for k, v in obs.items()}, route[b_ev], keep["pooled"].grad[b_ev])
    cands = [c for c in cloud.candidates(EVENT_TAU) if c[0] == 0 and c[1] == ch_ev]
    assert len(cands) == 1
    base, flipped = cloud.grads(), cloud.grads(flip=cands[0][:3])
    pre = T.ENC + "conv.mlp."
    # "the other implementation": the restatement's gradient with that decision the other way, plus rounding-sized noise
    other = {}
    for n in NAMES:
        gr = P[pre + n].grad
        other[n] = gr + (flipped[n] - base[n]) + 2e-7 * float(gr.abs().max()) * torch.randn(gr.shape, generator=g)
    shift = max(float((flipped[n] - base[n]).abs().max()) / float(P[pre + n].grad.abs().max()) for n in NAMES)
    assert shift > 1e-4                                                   # the event is far above the tolerance it would break
    ref = types.SimpleNamespace(P=P, keep=keep, route=route)
    report = dict(argmax_differs=0, argmax_gap=0.0, encoder_events=0, encoder_candidates=0, event_max_preact=0.0, events=[],
                  candidates_not_confirmed=0, unconfirmed_events=0)
    # (confirm=False: the flip is synthetic -- the HIP summation order does not really decide this unit the other way)
    _locate_encoder_events(ref, other, report, confirm=False)
    assert report["encoder_events"] == 1 and report["argmax_differs"] == 0 and report["unconfirmed_events"] == 1
    ev = report["events"][0]
    assert (ev["cloud"], ev["layer"], ev["channel"]) == (b_ev, 0, ch_ev) and ev["preact"] <= EVENT_TAU
    for n in NAMES:
        gr = P[pre + n].grad
        assert float((other[n] - gr).abs().max()) <= 2e-6 * float(gr.abs().max()), n


def test_an_event_the_hip_summation_order_does_not_confirm_is_not_accepted():
    """Round 4: the projection coefficient alone accepts nothing.  The same synthetic one-flip difference as above, but with the
    confirmation on: oracle/pcrl_oracle.c evaluates the point in the HIP kernels' order, finds the pre-activation (~ +3e-6, far
    above the ~1e-7 the two orders differ by) on the SAME side of zero as ATen -- so a kernel error that merely looks like that
    flip stays in the residual instead of being moved into the restatement."""
    T, P, obs, g = _setup()
    route = T.pointnet_prepool(P, obs).argmax(-1)
    b_ev, ch_ev = 2, 11
    pts = torch.unique(route[b_ev])
    with torch.no_grad():
        x = T.preprocess({k: v[b_ev:b_ev + 1, :, pts] for k, v in obs.items()})
        z0 = torch.nn.functional.conv1d(x, P[T.ENC + "conv.mlp.conv0.weight"], P[T.ENC + "conv.mlp.conv0.bias"])[0, ch_ev]
        P[T.ENC + "conv.mlp.conv0.bias"][ch_ev] -= z0[int(z0.abs().argmin())] - 3e-6
    route = T.pointnet_prepool(P, obs).argmax(-1)
    keep = {}
    feat = T.pointnet_forward(P, obs, route=route, keep=keep)
    (feat * torch.randn(feat.shape, generator=g)).sum().backward()
    cloud = T.CloudEncoder(P, {k: v[b_ev] for k, v in obs.items()}, route[b_ev], keep["pooled"].grad[b_ev])
    cand = [c for c in cloud.candidates(EVENT_TAU) if c[0] == 0 and c[1] == ch_ev][0]
    base, flipped = cloud.grads(), cloud.grads(flip=cand[:3])
    pre = T.ENC + "conv.mlp."
    before = {n: P[pre + n].grad.clone() for n in NAMES}
    other = {n: P[pre + n].grad + (flipped[n] - base[n]) for n in NAMES}
    report = dict(argmax_differs=0, argmax_gap=0.0, encoder_events=0, encoder_candidates=0, event_max_preact=0.0, events=[],
                  candidates_not_confirmed=0, unconfirmed_events=0)
    _locate_encoder_events(types.SimpleNamespace(P=P, keep=keep, route=route), other, report)
    assert report["encoder_events"] == 0 and report["candidates_not_confirmed"] >= 1
    for n in NAMES:
        assert torch.equal(P[pre + n].grad, before[n]), n          # nothing was patched


def test_the_c_oracle_s_relu_inputs_are_the_forward_s():
    """oracle/pcrl_oracle.c::pcrl_oracle_point_preacts_f32 (what confirms an event) against the same file's encoder forward:
    max(pre2, 0) is the pre-pool feature bit for bit, and the three layers agree with ATen to rounding."""
    from oracle import c_oracle
    w = make_encoder_weights(6, 64, 128, 256, seed=9)
    obs = make_obs(3, 50, seed=10)
    feat = c_oracle.preprocess(obs)
    _, _, prepool = c_oracle.encoder_fwd(feat, w, want_prepool=True)
    for b in range(3):
        p0, p1, p2 = c_oracle.point_preacts(np.ascontiguousarray(feat[b].T), w)
        assert np.array_equal(np.maximum(p2, 0).T, prepool[b])
        x = torch.from_numpy(feat[b:b + 1])
        z0 = torch.nn.functional.conv1d(x, torch.from_numpy(w["w0"])[..., None], torch.from_numpy(w["b0"]))[0]
        assert np.abs(z0.numpy().T - p0).max() <= 1e-5


def test_no_event_is_invented_when_the_gradients_agree():
    T, P, obs, g = _setup(seed=5)
    route = T.pointnet_prepool(P, obs).argmax(-1)
    keep = {}
    feat = T.pointnet_forward(P, obs, route=route, keep=keep)
    (feat * torch.randn(feat.shape, generator=g)).sum().backward()
    pre = T.ENC + "conv.mlp."
    other = {n: P[pre + n].grad + 5e-7 * float(P[pre + n].grad.abs().max()) * torch.randn(P[pre + n].grad.shape, generator=g) for n in NAMES}
    report = dict(argmax_differs=0, argmax_gap=0.0, encoder_events=0, encoder_candidates=0, event_max_preact=0.0, events=[],
                  candidates_not_confirmed=0, unconfirmed_events=0)
    _locate_encoder_events(types.SimpleNamespace(P=P, keep=keep, route=route), other, report)
    assert report["encoder_events"] == 0
