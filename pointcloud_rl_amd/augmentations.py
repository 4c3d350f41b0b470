"""DrQ point-cloud augmentations, fused into the encoder's load.

Registry and class contracts follow the reference's pyrl/utils/augmentations/{builder.py:7-104,
pcd_aug.py:125-227, 306-327}.  Instead of materialising an augmented xyz tensor, an augmentation
returns the observation as an `AugmentedObs` carrying the parameters; the encoder kernel applies
them while loading points (include/pcrl.h, pcrl_aug_desc).  `materialize()` runs the stand-alone
kernel for consumers other than the encoder.
"""
import os
from collections.abc import Sequence

import numpy as np

import torch

from .networks.pointnet import AugmentedObs, batch_rows
from .utils.registry import Registry, build_from_cfg

AUGMENTATIONS = Registry("data augmentation")


class DataAugmentations:
    """Sequence of augmentations applied in order (builder.py:11-46)."""

    def __init__(self, transforms):
        assert isinstance(transforms, Sequence)
        self.transforms = []
        for t in transforms:
            if isinstance(t, dict):
                t = build_from_cfg(t, AUGMENTATIONS)
            elif not callable(t):
                raise TypeError(f"transform must be callable or a dict, but got {type(t)}")
            self.transforms.append(t)

    def __call__(self, data, begin_index=0):
        for t in self.transforms[begin_index:]:
            data = t(data)
            if data is None:
                return None
        return data

    def __getitem__(self, key):
        return self.transforms[key]

    def __repr__(self):
        return type(self).__name__ + "(" + "".join(f"\n    {t}" for t in self.transforms) + "\n)"


def build_data_augmentations(cfg, default_args=None):
    if cfg is None:
        return None
    if not isinstance(cfg, (list, tuple)):
        cfg = [cfg]
    return DataAugmentations(cfg)


def _as_augmented(data):
    out = AugmentedObs(data)
    out.aug = dict(getattr(data, "aug", None) or {})
    out.repeat = int(getattr(data, "repeat", 1) or 1)
    return out


def _batch_shape(data):
    """Shape of the xyz tensor the augmentation acts on: [rows of the (virtually repeated) batch, 3, N]."""
    return (batch_rows(data),) + tuple(data["xyz"].shape[1:])


class _PointAug:
    def __init__(self, main_key, req_keys):
        self.main_key = main_key
        self.req_keys = req_keys or [main_key]
        assert main_key in self.req_keys, f"{main_key}, {self.req_keys} do not satisfy the requirement!"
        if list(self.req_keys) != ["xyz"]:
            raise NotImplementedError("the fused augmentations act on the 'xyz' key of a point-cloud observation "
                                      "(main_key='xyz', req_keys=['xyz'], as in configs/mfrl/drq/*/pn_*.py)")

    def _check(self, data):
        assert self.main_key in data, f"{self.main_key}, {list(data.keys())}"
        assert data["xyz"].shape[-2] == 3


@AUGMENTATIONS.register_module()
class RandomJitterPoints(_PointAug):
    """xyz + U(lo, hi) i.i.d. per coordinate (pcd_aug.py:306-327).  The reference draws the noise on
    the CPU and copies it to the device every step; here it is generated in-kernel (Philox4x32-10,
    keyed by `seed` and a per-call counter) unless `noise_override` is set (parity tests)."""

    def __init__(self, main_key="inputs/xyz", req_keys=None, jitter_range=[-0.1, 0.1], seed=None):
        super().__init__(main_key, req_keys)
        self.jitter_range = [float(jitter_range[0]), float(jitter_range[1])]
        self.seed = int(seed) if seed is not None else int(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
        self.calls = 0
        self._counter = None           # device int64 call counter: a hipGraph replay must draw fresh noise
        self.noise_override = []       # parity tests queue explicit noise tensors here (consumed in call order)
        # Inside an update step whose replay sampling is a device launch the step's own draw counter (DeviceReplay.state[0], advanced
        # by that launch) is the Philox offset and the calls of the step differ by their seed: no per-call `counter += 1` / `clone()`
        # (two 4-5 us ATen launches each; K2 has two calls per step).  Set / cleared by the agent around the step (`begin_step`).
        self._shared, self._slot = None, 0

    def __call__(self, data):
        self._check(data)
        out = _as_augmented(data)
        if self.noise_override:
            noise = self.noise_override.pop(0)
            assert tuple(noise.shape) == _batch_shape(data), f"{tuple(noise.shape)} vs {_batch_shape(data)}"
            out.aug["jitter_noise"] = noise.to(device=data["xyz"].device, dtype=torch.float32).contiguous()
        elif self._shared is not None and self._shared.device == data["xyz"].device:
            self._slot += 1
            out.aug.update(jitter_range=self.jitter_range, seed=(self.seed + 0x9E3779B97F4A7C15 * self._slot) & 0x7FFFFFFFFFFFFFFF,
                           offset=self.calls, offset_tensor=self._shared)
        else:
            dev = data["xyz"].device
            if self._counter is None or self._counter.device != dev:
                self._counter = torch.full((1,), self.calls, dtype=torch.int64, device=dev)
            self._counter += 1
            out.aug.update(jitter_range=self.jitter_range, seed=self.seed, offset=self.calls,
                           offset_tensor=self._counter.clone())   # this call's own slot, filled on the device
        self.calls += 1
        return out

    def begin_step(self, shared_counter):
        """shared_counter: device int64 [1] tensor that changes exactly once per update step before the step's encoder launches
        (None: every call advances its own counter, as outside update steps)."""
        assert shared_counter is None or (shared_counter.dtype == torch.int64 and shared_counter.numel() == 1)
        self._shared, self._slot = shared_counter, 0

    def __repr__(self):
        return f"{type(self).__name__}(jitter_range={self.jitter_range},"


@AUGMENTATIONS.register_module()
class RandomDownSample:
    """Keep a random subset of the points, the same subset for every cloud of the batch and every key
    (pcd_aug.py:231-268: `batch_perm(data[:1, 0, :], 1, max_num_points)[0]` = the first k entries of
    `rand(N).argsort()`, drawn once per call and applied to all `req_keys`).  Nothing is gathered: the index travels
    with the observation and the encoder reads point index[p] for position p.
    With drop_ratio and fixed_ratio=False -- what both shipped `pn_dropout.py` configs use -- the NUMBER of dropped points is itself
    random per call (pcd_aug.py:244-246: `np.random.randint(int(N * ratio))`).  On the GPU that count is drawn on the device
    (uniform on the same range, torch's CUDA generator) and handed to the kernels by pointer next to the full permutation
    (`point_count`, include/pcrl.h n_index_ptr): tensor shapes do not change from call to call, so a captured hipGraph follows the
    count like every other draw.  On CPU tensors (parity tests against the reference class) the count comes from numpy's global
    generator exactly as in the reference and the index is sliced."""

    def __init__(self, main_key="inputs/xyz", req_keys=["input/xyz"], max_num_points=None, drop_ratio=None, fixed_ratio=True):
        assert (drop_ratio is not None) ^ (max_num_points is not None)
        self.main_key, self.req_keys = main_key, list(req_keys)
        self.max_num_points, self.drop_ratio, self.fixed_ratio = max_num_points, drop_ratio, fixed_ratio
        self.index_override = []       # parity tests queue explicit index tensors here
        self.count_override = []       # ... and explicit kept-point counts (random-count mode)
        self.last_count = None

    def __call__(self, data):
        assert self.main_key in data, f"{self.main_key}, {list(data.keys())}"
        point_keys = [k for k, v in data.items() if torch.is_tensor(v) and v.ndim == 3]
        missing = [k for k in point_keys if k not in self.req_keys]
        if missing:
            raise NotImplementedError(f"RandomDownSample: point-cloud keys {missing} are not in req_keys {self.req_keys}; "
                                      "the fused encoder reads every per-point key through the same index")
        x = data[self.main_key]
        N = x.shape[-1]
        out = _as_augmented(data)
        random_count = self.drop_ratio is not None and not self.fixed_ratio
        if random_count and x.is_cuda:
            hi = int(N * self.drop_ratio)                 # n_drop is uniform on [0, hi)
            if hi < 1:
                return out
            if self.index_override:
                index = self.index_override.pop(0).to(device=x.device, dtype=torch.int32).contiguous()
                assert index.numel() == N, "random-count mode takes the whole permutation"
            else:
                index = torch.rand(N, device=x.device).argsort().to(torch.int32)
            if self.count_override:
                count = torch.full((1,), int(self.count_override.pop(0)), dtype=torch.int32, device=x.device)
            else:
                count = (N - torch.randint(0, hi, (1,), device=x.device, dtype=torch.int64)).to(torch.int32)
            out.aug["point_index"], out.aug["point_count"] = index, count
            self.last_count = count          # (a replayed graph rewrites this tensor in place: tests read the count a step used)
            return out
        if self.drop_ratio is not None:
            n_drop = int(N * self.drop_ratio) if self.fixed_ratio else int(np.random.randint(int(N * self.drop_ratio)))
            k = N - n_drop
        else:
            k = min(self.max_num_points, N)
        if self.index_override:
            index = self.index_override.pop(0).to(device=x.device, dtype=torch.int32).contiguous()
        elif k >= N:
            return out
        else:
            index = torch.rand(N, device=x.device).argsort()[:k].to(torch.int32)
        out.aug["point_index"] = index
        return out

    def __repr__(self):
        if self.drop_ratio is not None:
            return f"{type(self).__name__}(drop_ratio={self.drop_ratio}) (fixed_ratio={self.fixed_ratio})"
        return f"{type(self).__name__}(max_num_points={self.max_num_points})"


@AUGMENTATIONS.register_module()
class ColorJitterPoints:
    """torchvision's ColorJitter on the rgb key (pcd_aug.py:269-303): the reference views rgb [B,3,N] (uint8, as the replay
    stores it) as an image batch [B,3,1,N] and applies ONE draw -- a random order of brightness / contrast / saturation / hue
    and one factor each -- to the whole batch, in torchvision's uint8 arithmetic.  Here the draw is made with the same torch
    calls in the same order (`randperm(4)`, then a `uniform_` per enabled factor, from torch's global generator) and
    travels with the observation; the encoder kernel jitters each point's colour while loading it.  The one quantity that
    is not per point -- the contrast step blends with the cloud's own grayscale mean -- comes from a small pre-pass over
    rgb (3 B/point).  Arithmetic restated from torchvision 0.14.1 (oracle/color_jitter_ref.py); torchvision itself is not
    installable here, so parity against it is unpinned."""

    def __init__(self, main_key="inputs/rgb", req_keys="inputs/rgb", brightness=0.5, contrast=0.5, saturation=0.5, hue=0.5):
        self.main_key = main_key
        self.req_keys = [req_keys] if isinstance(req_keys, str) else list(req_keys or [main_key])
        if list(self.req_keys) != ["rgb"] or main_key != "rgb":
            raise NotImplementedError("the fused colour jitter acts on the 'rgb' key of a point-cloud observation "
                                      "(main_key='rgb', req_keys=['rgb'], as in configs/mfrl/drq/*/pn_colorjitter.py)")
        for name, v, hi in (("brightness", brightness, 1), ("contrast", contrast, 1), ("saturation", saturation, 1), ("hue", hue, 0.5)):
            if v < 0 or v > hi:
                raise ValueError(f"{name} shoud be non-negative")
        self.brightness, self.contrast, self.saturation, self.hue = brightness, contrast, saturation, hue

        def rng(value, center, clip=True):          # torchvision ColorJitter._check_input
            lo, hi = center - float(value), center + float(value)
            lo = max(lo, 0.0) if clip else lo
            return None if lo == hi == center else (lo, hi)
        self.ranges = [rng(brightness, 1.0), rng(contrast, 1.0), rng(saturation, 1.0), rng(hue, 0.0, clip=False)]
        self.params_override = []      # parity tests queue (order, factors) draws here
        self.graph_safe = False        # the draw is made on the host (torchvision's stream) and reaches the kernels by value

    def draw(self):
        """torchvision ColorJitter.get_params: (order [4], factors [4], None where a range is empty)."""
        if self.params_override:
            return self.params_override.pop(0)
        order = [int(i) for i in torch.randperm(4)]
        return order, [None if r is None else float(torch.empty(1).uniform_(r[0], r[1])) for r in self.ranges]

    def __call__(self, data):
        from . import hip
        assert "rgb" in data, f"rgb, {list(data.keys())}"
        rgb = data["rgb"]
        assert rgb.shape[-2] == 3
        if rgb.dtype != torch.uint8:
            raise NotImplementedError("ColorJitterPoints: the fused path implements the uint8 rgb the replay stores "
                                      f"(torchvision's float path differs), got {rgb.dtype}")
        order, factors = self.draw()
        color = dict(order=order, factors=factors)
        color["mean"] = hip.color_contrast_mean(rgb, color)      # [stored clouds]: a virtual repeat shares it
        out = _as_augmented(data)
        out.aug["color"] = color
        return out

    def __repr__(self):
        return (f"{type(self).__name__}(brightness={self.brightness},contrast={self.contrast},"
                f"saturation={self.saturation},hue={self.hue})")


def batch_rot_with_axis(angle, rot_axis=2):
    """[.., 1] angles -> [.., 3, 3] rotations about `rot_axis` (reference pyrl/utils/torch/ops.py:171-183)."""
    assert angle.shape[-1] == 1
    c, s = torch.cos(angle)[..., 0], torch.sin(angle)[..., 0]
    j, k = (rot_axis + 1) % 3, (rot_axis + 2) % 3
    rot = torch.zeros(list(angle.shape[:-1]) + [3, 3], dtype=angle.dtype, device=angle.device)
    rot[..., rot_axis, rot_axis] = 1
    rot[..., j, j] = c
    rot[..., k, k] = c
    rot[..., j, k] = -s
    rot[..., k, j] = s
    return rot


@AUGMENTATIONS.register_module()
class GlobalRotScaleTrans(_PointAug):
    """Per-cloud rotation about one axis, per-axis scale and translation (pcd_aug.py:125-227): the
    [B,3,4] matrix is drawn here exactly as the reference builds it (including scale applied to the
    rows of [R|0] before the translation is written, and `delta_xyz[-1] = 0` zeroing the LAST CLOUD's
    translation when shift_height is False); the encoder kernel applies it as R x + t."""

    def __init__(self, main_key=["obs/pointcloud/xyz"], req_keys=None, rot_range=[-0.78539816, 0.78539816], rot_axis="z",
                 scale_ratio_range=[0.95, 1.05], translation_range=[0, 0, 0], shift_height=False, seed=None):
        super().__init__(main_key, req_keys)
        if rot_range is not None and not isinstance(rot_range, (list, tuple, np.ndarray)):
            rot_range = [-rot_range, rot_range]
        self.rot_range = rot_range
        assert rot_axis in ["x", "y", "z", 0, 1, 2]
        self.rot_axis = ord(rot_axis) - ord("x") if isinstance(rot_axis, str) else rot_axis
        self.scale_ratio_range = scale_ratio_range
        if translation_range is not None:
            translation_range = torch.tensor(translation_range, dtype=torch.float)
            assert (translation_range >= 0).all(), "translation_range should be positive"
        self.translation_range = translation_range
        self.shift_height = shift_height
        self.matrix_override = []
        # Inside an update step fed by a device-sampling replay (`begin_step`, as RandomJitterPoints) the matrices are drawn by ONE
        # launch keyed by the step's device draw counter instead of ~15 ATen launches per call from torch's generator.
        self.seed = int(seed) if seed is not None else int(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
        self.calls = 0
        self._shared, self._slot, self._mats = None, 0, {}
        self._predrawn = None                  # (rows, slot) of matrices the previous call's launch has already drawn
        self.pair_draws = True

    def sample_matrix(self, batch_size, device):
        mat = torch.zeros([batch_size, 3, 4], device=device)
        if self.rot_range is not None:
            angle = torch.zeros([batch_size, 1], device=device).uniform_(*self.rot_range)
            mat[..., :3, :3] = batch_rot_with_axis(angle, self.rot_axis)
        if self.scale_ratio_range is not None:
            mat[..., :3, :] *= torch.zeros([batch_size, 3, 1], device=device).uniform_(*self.scale_ratio_range)
        if self.translation_range is not None:
            delta = (torch.rand([batch_size, 3], device=device) - 0.5) * 2 * self.translation_range.to(device)
            if not self.shift_height:
                delta[-1] = 0
            mat[..., :3, 3] = delta
        if self.rot_range is None:
            # apply_rot_trans(with_rot=False) skips the matrix product altogether (pcd_aug.py:205-214)
            mat[..., :3, :3] = torch.eye(3, device=device)
        return mat

    def __call__(self, data):
        self._check(data)
        out = _as_augmented(data)
        if any(k.startswith("jitter") for k in out.aug):
            # the fused point load applies the matrix first and the jitter second (include/pcrl.h, pcrl_aug_desc); a list that
            # jitters first would need M (x + n) -- refuse instead of silently computing M x + n
            raise NotImplementedError("list GlobalRotScaleTrans BEFORE RandomJitterPoints: the fused encoder load applies the affine map "
                                      "first, then the jitter")
        if self.matrix_override:
            mat = self.matrix_override.pop(0)
        elif self._shared is not None and self._shared.device == data["xyz"].device:
            # one launch; its matrices already carry a zero translation when there is no range (pcrl_affine_sample_f32)
            from . import hip
            self._slot += 1
            rows = batch_rows(data)
            buf = lambda slot: self._mats.setdefault((rows, slot), torch.empty(rows, 3, 4, dtype=torch.float32, device=data["xyz"].device))
            seed_of = lambda slot: (self.seed + 0x9E3779B97F4A7C15 * slot) & 0x7FFFFFFFFFFFFFFF
            mat = buf(self._slot)              # persistent output buffers: a captured launch keeps writing the same memory
            if self._predrawn == (rows, self._slot):
                pass                           # drawn by the previous call's launch
            else:
                # the first call of a step also draws for the second (DrQ augments obs and next_obs back to back, drq.py:62-75): one
                # launch instead of two; a second call that never comes costs nothing but the unused matrices
                pair = self.pair_draws and self._slot == 1
                hip.affine_sample(mat, self.rot_axis, self.rot_range, self.scale_ratio_range,
                                  None if self.translation_range is None else self.translation_range.tolist(), self.shift_height,
                                  seed_of(self._slot), offset=self.calls, offset_tensor=self._shared,
                                  second=(buf(2), seed_of(2), self.calls + 1) if pair else None)
                self._predrawn = (rows, 2) if pair else None
            self.calls += 1
            out.aug["affine"] = mat
            return out
        else:
            mat = self.sample_matrix(batch_rows(data), data["xyz"].device)
        if self.translation_range is None:
            mat = mat.clone()
            mat[..., :3, 3] = 0
        out.aug["affine"] = mat.to(torch.float32).contiguous()
        return out

    def begin_step(self, shared_counter):
        """shared_counter: device int64 [1] tensor the step advances once before its encoder launches (DeviceReplay.state[:1]);
        None: every call draws with torch's generator, as the reference does."""
        assert shared_counter is None or (shared_counter.dtype == torch.int64 and shared_counter.numel() == 1)
        self._shared, self._slot, self._predrawn = shared_counter, 0, None     # a pair drawn ahead never outlives its step

    def __repr__(self):
        return (f"{type(self).__name__}(rot_range={self.rot_range}, scale_ratio_range={self.scale_ratio_range}, "
                f"translation_range={self.translation_range}, shift_height={self.shift_height})")
