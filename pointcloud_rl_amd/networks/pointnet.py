"""PointNet visual backbone on the fused MI355X kernels.

Registry name, constructor keywords and forward contract follow the reference's
pyrl/networks/backbones/pointnet.py:76-157; parameter names follow its state_dict
(`conv.mlp.conv{0,1,2}.weight`, `conv.mlp.norm{1,2}.*`, `final_mlp.{0,1}.*`).
The per-point MLP, its LayerNorms, the max-pool and their backward run in libpcrl_hip.so; there
is no eager or CPU path.
"""
import os

import torch
import torch.nn as nn

from .. import hip
from ..utils.torch_utils import ExtendedModule, ExtendedSequential
from .builder import NETWORK


class LayerNorm1D(nn.LayerNorm):
    """Parameter holder for the channel-first LayerNorm (reference "LN1d", nn_layer.py:222-225)."""


class AugmentedObs(dict):
    """Observation dict that carries a not-yet-applied point-cloud augmentation.  The encoder
    kernel applies `aug` while loading xyz, so the augmented cloud is never materialised.
    repeat > 1: every stored sample stands for `repeat` consecutive rows of the batch (DrQ's
    GDict(obs).repeat(num_aug, 0), drq.py:52-60, without the copies): the encoder reads stored cloud b // repeat for
    cloud b, the augmentation parameters are per row of the repeated batch."""
    aug = None           # dict(jitter_noise=, jitter_range=, seed=, offset=, affine=)
    repeat = 1


def batch_rows(obs):
    """Rows of the batch an observation dict stands for (stored rows x repeat)."""
    return obs["xyz"].shape[0] * int(getattr(obs, "repeat", 1) or 1)


def materialize(obs):
    """Apply a pending augmentation of an AugmentedObs to its xyz tensor (for consumers other than the
    fused encoder, e.g. visualisation); returns a plain dict."""
    aug = dict(getattr(obs, "aug", None) or {})
    out = dict(obs)
    repeat = int(getattr(obs, "repeat", 1) or 1)
    if repeat > 1:
        out = {k: (torch.repeat_interleave(v, repeat, dim=0) if torch.is_tensor(v) else v) for k, v in out.items()}
    color = aug.pop("color", None)
    if color is not None:          # ColorJitterPoints: before any repeat-independent indexing, on the stored uint8 rgb
        rgb = obs["rgb"].contiguous()
        jit = hip.color_jitter_u8(rgb, color)
        out["rgb"] = torch.repeat_interleave(jit, repeat, dim=0) if repeat > 1 else jit
    index = aug.pop("point_index", None)
    count = aug.pop("point_count", None)
    if index is not None and count is not None:
        index = index[:int(count.item())]       # (host read: this path is for consumers outside the hot loop)
    if index is not None:          # RandomDownSample: the same points of every key
        out = {k: (v[..., index.long()] if torch.is_tensor(v) and v.ndim == 3 else v) for k, v in out.items()}
    if aug:
        out["xyz"] = hip.augment_xyz(out["xyz"].contiguous(), **aug)
    return out


class ConvMLP(ExtendedModule):
    """Holds the shared per-point MLP's parameters under the reference's names (mlp.py:43-56, 103-108)."""

    def __init__(self, mlp_spec, eps):
        super().__init__()
        self.mlp = ExtendedSequential()
        for i in range(len(mlp_spec) - 1):
            with_norm = i > 0                       # ignore_first_ln=True drops norm0; bias only when no norm follows
            self.mlp.add_module(f"conv{i}", nn.Conv1d(mlp_spec[i], mlp_spec[i + 1], kernel_size=1, bias=not with_norm))
            if with_norm:
                self.mlp.add_module(f"norm{i}", LayerNorm1D(mlp_spec[i + 1], eps=eps))
            self.mlp.add_module(f"act{i}", nn.ReLU(inplace=True))

    def kernel_params(self):
        m = self.mlp
        return [m.conv0.weight, m.conv0.bias, m.conv1.weight, m.norm1.weight, m.norm1.bias,
                m.conv2.weight, m.norm2.weight, m.norm2.bias]


class _EncoderFn(torch.autograd.Function):
    """pooled = max_n MLP(points); forward and backward are single C-ABI calls."""

    @staticmethod
    def forward(ctx, net, desc, keep, aug, *weights):
        ew, packed = net._weights_desc()
        aug_desc = hip.make_aug_desc(**aug) if aug else None
        pooled, argmax = hip.encoder_fwd(desc, ew, packed, aug=aug_desc, workspace=net._workspace("fwd"), bf16=net.compute_dtype == "bf16",
                                         split=net.compute_dtype == "f32split")
        ctx.net, ctx.desc, ctx.keep, ctx.aug_desc, ctx.aug = net, desc, keep, aug_desc, aug
        ctx.ew, ctx.packed = ew, packed
        ctx.save_for_backward(argmax, pooled)
        ctx.mark_non_differentiable(argmax)
        net.last_argmax = argmax
        return pooled, argmax

    @staticmethod
    def backward(ctx, grad_pooled, _grad_argmax):
        argmax, pooled = ctx.saved_tensors
        net = ctx.net
        flat, n_active = hip.encoder_bwd(ctx.desc, ctx.ew, ctx.packed, argmax, grad_pooled, aug=ctx.aug_desc,
                                         workspace=net._workspace("bwd", ctx.desc.B), want_n_active=True, bf16=net.compute_dtype == "bf16",
                                         split=net.compute_dtype == "f32split",
                                         pooled=pooled)
        net.last_n_active = n_active
        views = hip.encoder_grad_views(flat, ctx.ew)
        return (None, None, None, None) + tuple(views[k] for k in ("conv0.weight", "conv0.bias", "conv1.weight", "norm1.weight",
                                                                     "norm1.bias", "conv2.weight", "norm2.weight", "norm2.bias"))


@NETWORK.register_module()
class PointNet(ExtendedModule):
    def __init__(self, feat_dim, mlp_spec=[64, 128, 1024], out_channels=None, global_feat=True, feature_transform=[1],
                 norm_cfg=dict(type="LN1d", eps=1e-6), act_cfg=dict(type="ReLU"), ignore_first_ln=False, num_patch=1,
                 compute_dtype="f32", **kwargs):
        super().__init__()
        # compute_dtype (not a reference keyword): "bf16" runs conv1 / conv2 of the per-point MLP on the bf16 matrix cores
        # with fp32 accumulation, fp32 master weights and fp32 gradient GEMMs (BASELINE.json config 3); "f32" is exact.
        # "f32split" (experimental): conv1 / conv2 and the backward's data-gradient GEMMs as three-term bf16 splits on the bf16
        # matrix cores (~fp32 accuracy, not bit-comparable with "f32"); weight-gradient GEMMs, LayerNorms, pool stay fp32.
        if compute_dtype not in ("f32", "bf16", "f32split"):
            raise ValueError(f"compute_dtype must be 'f32', 'bf16' or 'f32split', got {compute_dtype!r}")
        self.compute_dtype = compute_dtype
        # The fused kernel implements the configuration every shipped point-cloud SAC/DrQ config uses
        # (configs/mfrl/{sac,drq}/*/pn*.py): no T-Nets, LN1d + ReLU, first LayerNorm dropped.
        if len(feature_transform) > 0:
            raise NotImplementedError("feature_transform (STN/T-Net) is not on the MI355X hot path; use feature_transform=[]")
        if not global_feat:
            raise NotImplementedError("global_feat=False")
        if norm_cfg is None or norm_cfg.get("type") != "LN1d" or (act_cfg or {}).get("type") != "ReLU" or not ignore_first_ln:
            raise NotImplementedError("the fused encoder implements norm_cfg=LN1d, act_cfg=ReLU, ignore_first_ln=True")
        mlp_spec = [int(c) for c in mlp_spec]
        if len(mlp_spec) != 3:
            raise NotImplementedError("the fused encoder implements a 3-layer shared MLP")
        if mlp_spec[-1] > 256 and compute_dtype != "f32":
            raise NotImplementedError("mlp_spec with a last layer wider than 256 channels (the class default [64, 128, 1024]) runs in fp32 only")
        self.feat_dim, self.mlp_spec = int(feat_dim), mlp_spec
        self.global_feat, self.feature_transform = global_feat, feature_transform
        self.eps = float(norm_cfg.get("eps", 1e-5))
        self.conv = ConvMLP([self.feat_dim] + mlp_spec, self.eps)
        self.final_mlp = nn.Sequential(nn.Linear(mlp_spec[-1], out_channels), nn.LayerNorm(out_channels)) if out_channels is not None else None
        self._packed = None
        self._packed_key = None
        self._ws = {}
        self.last_argmax = None
        self.last_n_active = None

    # -- kernel-side state ----------------------------------------------------------------------
    def invalidate_packed(self):
        """Call after the weights were modified outside autograd's version tracking (fused optimizer)."""
        self._packed_key = None

    def _pack_if_stale(self, to_gather=False):
        params = self.conv.kernel_params()
        if not params[0].is_cuda:
            raise RuntimeError("pointcloud_rl_amd.PointNet runs on MI355X only: move the module to a CUDA/HIP device "
                               "(there is no CPU implementation of the encoder)")
        w = [p.detach().reshape(p.shape[0], -1) if p.ndim == 3 else p.detach() for p in params]
        ew, _ = hip.make_encoder_weights(*w, self.eps)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._packed is None or self._packed.device != params[0].device:
            self._packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=params[0].device)
            self._packed_key = None
        if key != self._packed_key:
            if to_gather:
                hip.encoder_pack_attach_to_gather(ew, self._packed)
            else:
                hip.encoder_pack_weights(ew, self._packed)
            self._packed_key = key
            return ew, self._packed, True
        return ew, self._packed, False

    def _weights_desc(self):
        return self._pack_if_stale()[:2]

    def attach_pack_to_gather(self):
        """The re-pack a forward would start with, handed to this thread's NEXT replay sampling launch instead (with the column-gather
        jobs attached so far; `hip.encoder_pack_attach_to_gather`): True if the image was stale and a job is now pending -- the caller
        issues the sampling launch and then `hip.encoder_pack_flush_pending()`; False if the image is current (nothing pending)."""
        return self._pack_if_stale(to_gather=True)[2]

    def _workspace(self, kind, B=None):
        """Scratch is cached per kind and grown on demand (never shrunk)."""
        if kind == "fwd":
            return self._ws.get("fwd")
        import ctypes
        need = ctypes.c_size_t()
        hip.check(hip.lib().pcrl_encoder_bwd_workspace_bytes(B, self.feat_dim, *self.mlp_spec, ctypes.byref(need)))
        ws = self._ws.get("bwd")
        if ws is None or ws.numel() < need.value:
            ws = torch.empty(need.value, dtype=torch.uint8, device=self.conv.mlp.conv0.weight.device)
            self._ws["bwd"] = ws
        return ws

    # -- reference API ----------------------------------------------------------------------------
    def pooled(self, inputs):
        """Max-pooled per-point features [B, mlp_spec[-1]] and their argmax (int32)."""
        aug = getattr(inputs, "aug", None)
        if torch.is_tensor(inputs):
            inputs = {"xyz": inputs.to(dtype=torch.float32)}
        desc, keep = hip.make_cloud_desc(inputs)
        return _EncoderFn.apply(self, desc, keep, aug, *self.conv.kernel_params())

    # -- autograd-free entry points used by the fused update step --------------------------------
    def encode_raw(self, inputs, head=None):
        """(pooled [B,c3], argmax [B,c3] int32, ctx) without building an autograd graph.  head: hip.make_feature_head(...) --
        final_mlp (Linear + LayerNorm) applied to every cloud by the same launch, written where the head's jobs say."""
        aug = getattr(inputs, "aug", None)
        if torch.is_tensor(inputs):
            inputs = {"xyz": inputs.to(dtype=torch.float32)}
        desc, keep = hip.make_cloud_desc(inputs)
        ew, packed = self._weights_desc()
        aug_desc = hip.make_aug_desc(**aug) if aug else None
        pooled, argmax = hip.encoder_fwd(desc, ew, packed, aug=aug_desc, workspace=self._workspace("fwd"), bf16=self.compute_dtype == "bf16",
                                         split=self.compute_dtype == "f32split", head=head)
        return pooled, argmax, (desc, keep, aug, aug_desc, ew, packed, pooled)

    def ctx_for(self, inputs, pooled):
        """The backward context of encode_raw for `inputs` when its rows were encoded as part of a larger launch (`pooled`:
        their rows of that launch's output)."""
        aug = getattr(inputs, "aug", None)
        desc, keep = hip.make_cloud_desc(inputs)
        ew, packed = self._weights_desc()
        return desc, keep, aug, (hip.make_aug_desc(**aug) if aug else None), ew, packed, pooled

    def can_prepare(self, ctx):
        """Whether backward_prepare would launch (exact fp32 arithmetic, the forward's pooled values at hand, at most 2 048 clouds)."""
        desc, keep, aug, aug_desc, ew, packed, pooled = ctx
        return self.compute_dtype == "f32" and pooled is not None and desc.B <= 2048

    def backward_prepare(self, ctx, argmax):
        """The part of backward_raw that needs only the forward's outputs (hip.encoder_bwd_prepare); True when it was launched --
        backward_raw must then be called with prepared=True, behind it.  Exact fp32 arithmetic, Gram form only."""
        desc, keep, aug, aug_desc, ew, packed, pooled = ctx
        if not self.can_prepare(ctx):
            return False
        hip.encoder_bwd_prepare(desc, ew, packed, argmax, pooled, self._workspace("bwd", desc.B), aug=aug_desc)
        return True

    def backward_raw(self, ctx, argmax, grad_pooled, out, prepared=False):
        """Writes the flat gradient of the shared per-point MLP (reference parameter order) into `out`."""
        desc, keep, aug, aug_desc, ew, packed, pooled = ctx
        hip.encoder_bwd(desc, ew, packed, argmax, grad_pooled, aug=aug_desc, workspace=self._workspace("bwd", desc.B), out=out,
                        bf16=self.compute_dtype == "bf16", pooled=pooled, split=self.compute_dtype == "f32split", prepared=prepared)

    def forward(self, inputs, object_feature=True, concat_state=None, **kwargs):
        feature, _ = self.pooled(inputs)
        if self.final_mlp is not None:
            feature = self.final_mlp(feature)
        return feature
