"""TanhGaussianHead: squashed-Gaussian policy head of SAC/DrQ.

Contract of the reference's pyrl/networks/regression_heads/{gaussian.py:71-87, regression_base.py:11-74}
with ScaledTanhNormal (pyrl/utils/torch/distributions.py:45-119): log-std clamp, reparameterised
sample u = mean + std * eps, action = tanh(u) * scale + bias, and the reference's *epsilon* form of
the log-density correction  log(scale * (1 - tanh(u)^2) + 1e-6).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ..utils.torch_utils import ExtendedModule
from .builder import REGRESSION


@REGRESSION.register_module()
class TanhGaussianHead(ExtendedModule):
    def __init__(self, bound=None, dim_output=None, nn_cfg=None, predict_std=True, init_log_std=-0.5, clip_return=False,
                 num_heads=1, log_std_bound=[-20, 2], epsilon=1e-6):
        super().__init__()
        if num_heads != 1 or not predict_std or nn_cfg is not None:
            raise NotImplementedError("TanhGaussianHead on the hot path: num_heads=1, predict_std=True, nn_cfg=None")
        if bound is not None and bound[0] is not None:
            lo = np.ones(dim_output) * bound[0] if np.isscalar(bound[0]) else np.asarray(bound[0])
            hi = np.ones(dim_output) * bound[1] if np.isscalar(bound[1]) else np.asarray(bound[1])
            assert lo.shape == hi.shape and (dim_output is None or lo.shape[-1] == dim_output)
            dim_output = lo.shape[-1]
            self.lb = nn.Parameter(torch.tensor(lo), requires_grad=False)
            self.ub = nn.Parameter(torch.tensor(hi), requires_grad=False)
            self.scale = nn.Parameter(torch.tensor(hi - lo) / 2, requires_grad=False)
            self.bias = nn.Parameter(torch.tensor(lo + hi) / 2, requires_grad=False)
        else:
            self.lb = self.ub = None
            self.scale, self.bias = 1, 0
        self.bound = bound
        self.dim_output = dim_output
        self.dim_feature = dim_output * 2
        self.log_std_min, self.log_std_max = log_std_bound
        self.epsilon = epsilon
        self.clip_return = False
        self.noise_override = []       # parity tests queue the standard-normal draws here (consumed in call order)

    def _standard_normal(self, like):
        if self.noise_override:
            eps = self.noise_override.pop(0)
            assert eps.shape == like.shape, f"{eps.shape} vs {like.shape}"
            return eps.to(like)
        return torch.randn_like(like)

    def split_feature(self, feature):
        assert feature.shape[-1] == self.dim_feature, f"{feature.shape, self.dim_feature}"
        mean, log_std = feature.chunk(2, dim=-1)
        std = torch.clamp(log_std, min=self.log_std_min, max=self.log_std_max).exp()
        return mean, std

    def log_prob_with_logit(self, logit, mean, std):
        log_prob = -((logit - mean) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))
        return log_prob - torch.log(self.scale * (1 - torch.tanh(logit).pow(2)) + self.epsilon)

    def forward(self, feature, num_samples=1, mode="explore", **kwargs):
        if num_samples > 1:
            feature = feature.repeat_interleave(num_samples, dim=0)
        mean, std = self.split_feature(feature)
        parts = ["rsample-with-neg-logp"] if mode == "max-entropy" else mode.split("_")
        ret = []
        for part in parts:
            if part in ("mean", "eval"):
                ret.append(torch.tanh(mean) * self.scale + self.bias)
            elif part in ("explore", "sample"):
                ret.append(torch.tanh(mean + std * self._standard_normal(mean)) * self.scale + self.bias)
            elif part == "std":
                ret.append(None)        # TransformedNormal.stddev is None in the reference (distributions.py:16-18)
            elif part == "rsample-with-neg-logp":
                logit = mean + std * self._standard_normal(mean)
                log_p = self.log_prob_with_logit(logit, mean, std).sum(-1)
                ret.append([torch.tanh(logit) * self.scale + self.bias, -log_p[..., None]])
            else:
                raise NotImplementedError(f"TanhGaussianHead mode part {part!r}")
        return ret[0] if len(ret) == 1 else ret
