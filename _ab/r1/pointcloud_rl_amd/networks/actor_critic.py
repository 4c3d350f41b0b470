"""ContinuousActor / ContinuousCritic wrappers (reference pyrl/networks/applications/actor_critic.py:9-133)."""
import torch
import torch.nn as nn

from ..utils.torch_utils import ExtendedModule
from .builder import APPLICATION, build_all


class ActorCriticBase(ExtendedModule):
    def __init__(self, nn_cfg=None, head_cfg=None, mlp_cfg=None, backbone=None):
        super().__init__()
        assert nn_cfg is None or backbone is None
        self.backbone = build_all(nn_cfg) if backbone is None else backbone
        self.is_recurrent = getattr(self.backbone, "is_recurrent", False)
        self.final_mlp = build_all(mlp_cfg)
        self.head = build_all(head_cfg)

    def forward(self, obs, actions=None, *, rnn_mode="base", **kwargs):
        feature = self.backbone(obs, actions, rnn_mode=rnn_mode, **kwargs)
        kwargs.pop("feature", None)
        if self.final_mlp is not None:
            feature = self.final_mlp(feature)
        if self.head is not None:
            feature = self.head(feature, **kwargs)
        if rnn_mode == "base":
            return feature
        # non-recurrent network asked for states: None ("with_states") or [None]*3 ("full_states")
        return feature, ([None] * 3 if rnn_mode == "full_states" else None)


def _is_box(space):
    return space is not None and hasattr(space, "low") and hasattr(space, "high")


@APPLICATION.register_module(name="ContinuousPolicy")
@APPLICATION.register_module()
class ContinuousActor(ActorCriticBase):
    def __init__(self, nn_cfg=None, head_cfg=None, mlp_cfg=None, backbone=None, action_space=None, obs_shape=None,
                 action_shape=None, **kwargs):
        assert _is_box(action_space), "If you are training over discrete action space, you need DiscreteActor"
        if head_cfg is not None and action_space.is_bounded():
            head_cfg = dict(head_cfg)
            head_cfg["bound"] = [action_space.low, action_space.high]
        super().__init__(nn_cfg=nn_cfg, head_cfg=head_cfg, mlp_cfg=mlp_cfg, backbone=backbone)


@APPLICATION.register_module(name="ContinuousValue")
@APPLICATION.register_module()
class ContinuousCritic(ExtendedModule):
    """num_heads independent Q heads; with an injected shared visual_nn every head's Visuomotor holds
    the SAME encoder object (actor_critic.py:85-133, builder.py:61-68)."""

    def __init__(self, nn_cfg=None, head_cfg=None, mlp_cfg=None, backbone=None, share_feature=False, obs_shape=None,
                 action_shape=None, num_heads=1, average_grad=True, **kwargs):
        super().__init__()
        if backbone is not None or share_feature:
            raise NotImplementedError("shared-feature critics are outside the point-cloud SAC/DrQ hot path")
        self.values = nn.ModuleList()
        self.num_heads = num_heads
        self.shared_feature = False
        for _ in range(num_heads):
            self.values.append(ActorCriticBase(nn_cfg=nn_cfg, head_cfg=head_cfg, mlp_cfg=mlp_cfg, backbone=None))
        self.is_recurrent = self.values[0].is_recurrent

    def forward(self, obs, actions=None, **kwargs):
        ret = [value(obs=obs, actions=actions, **kwargs) for value in self.values]
        return torch.cat(ret, dim=-1)
