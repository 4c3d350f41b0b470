"""Dense heads: LinearMLP (reference pyrl/networks/backbones/mlp.py:15-100, block_utils.py:137-155).

Parameter names follow the reference (`mlp.linear{i}.weight/bias`) so checkpoints interchange.
"""
import torch
import torch.nn as nn

from ..utils.torch_utils import ExtendedModule, ExtendedSequential
from .builder import NETWORK


@NETWORK.register_module()
class LinearMLP(ExtendedModule):
    """Linear -> ReLU -> ... -> Linear.  norm_cfg must be None (every shipped SAC/DrQ config);
    `inactivated_output=True` leaves the last layer linear; `zero_out_indices` re-initialises the
    selected output rows of the last layer to U(-1e-3, 1e-3) (mlp.py:72-83)."""

    def __init__(self, mlp_spec, norm_cfg=None, act_cfg=dict(type="ReLU"), bias="auto", inactivated_output=True,
                 zero_out_indices=None, dense_init_cfg=None, **kwargs):
        super().__init__()
        if norm_cfg is not None:
            raise NotImplementedError("LinearMLP on the MI355X hot path supports norm_cfg=None only")
        if act_cfg is not None and act_cfg.get("type", "ReLU") != "ReLU":
            raise NotImplementedError("LinearMLP on the MI355X hot path supports ReLU only")
        if dense_init_cfg is not None:
            raise NotImplementedError("dense_init_cfg is not used by the SAC/DrQ point-cloud configs")
        use_bias = True if bias == "auto" else bool(bias)      # need_bias(None) -> True (nn_layer.py:240-242)
        self.mlp = ExtendedSequential()
        n = len(mlp_spec) - 1
        for i in range(n):
            self.mlp.add_module(f"linear{i}", nn.Linear(int(mlp_spec[i]), int(mlp_spec[i + 1]), bias=use_bias))
            last = i == n - 1
            if act_cfg is not None and not (inactivated_output and last):
                self.mlp.add_module(f"act{i}", nn.ReLU(inplace=True))
        if zero_out_indices is not None:
            last_dense = getattr(self.mlp, f"linear{n - 1}")
            with torch.no_grad():
                last_dense.weight[zero_out_indices].uniform_(-1e-3, 1e-3)
                last_dense.bias[zero_out_indices].uniform_(-1e-3, 1e-3)

    @property
    def linears(self):
        return [m for m in self.mlp if isinstance(m, nn.Linear)]

    def forward(self, feature, actions=None, **kwargs):
        if actions is not None:
            feature = torch.cat([feature, actions], dim=-1)
        return self.mlp(feature)
