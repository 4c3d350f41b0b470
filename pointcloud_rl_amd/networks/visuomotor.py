"""Visuomotor backbone: visual encoder -> [detach] -> + robot state -> + action -> dense head.

Contract of the reference's pyrl/networks/backbones/visuomotor.py:16-146 (non-recurrent path):
`visual_nn` may be injected (shared between actor, Q heads and target Q heads), `visual_feature=`
bypasses the encoder, `save_feature` keeps a copy of the visual feature in `saved_visual_feature`.
"""
from copy import copy

import torch

from ..utils.torch_utils import ExtendedModule
from .builder import NETWORK, build_all


@NETWORK.register_module()
class Visuomotor(ExtendedModule):
    def __init__(self, visual_nn_cfg, mlp_cfg, rnn_cfg=None, obs_feat_cfg=None, ac_feat_cfg=None, prev_ac_feat_cfg=None,
                 freeze_visual_nn=False, freeze_mlp=False, **kwargs):
        super().__init__()
        if rnn_cfg is not None or kwargs.get("rnn") is not None:
            raise NotImplementedError("recurrent Visuomotor is outside the point-cloud SAC/DrQ hot path")
        self.visual_nn = kwargs["visual_nn"] if "visual_nn" in kwargs else build_all(visual_nn_cfg)
        self.obs_feat = kwargs["obs_feat"] if "obs_feat" in kwargs else build_all(obs_feat_cfg)
        self.ac_feat = kwargs["ac_feat"] if "ac_feat" in kwargs else build_all(ac_feat_cfg)
        self.rnn = None
        self.final_mlp = build_all(mlp_cfg)
        if freeze_visual_nn:
            for p in self.visual_nn.parameters():
                p.requires_grad = False
        if freeze_mlp:
            for p in self.final_mlp.parameters():
                p.requires_grad = False
        self.saved_feature = None
        self.saved_visual_feature = None

    _DROPPED_SUFFIXES = ("_box", "_seg", "_sem_label")
    _STATE_KEYS = ("state", "agent")

    @classmethod
    def split_obs(cls, obs):
        """(what the visual encoder sees, robot state or None).  Keys naming boxes / segment ids / semantic labels and
        "visual_state" never reach the encoder, "state" / "agent" is the robot state, and an observation without any
        point-cloud / image key must consist of exactly one entry, which is handed over bare (reference
        visuomotor.py:80-96).  The result keeps the class and attributes of `obs` (pending augmentation, virtual repeat)."""
        if not isinstance(obs, dict):
            raise AssertionError(f"obs is not a dict! {type(obs)}")
        state_keys = [k for k in cls._STATE_KEYS if k in obs]
        if len(state_keys) > 1:
            raise AssertionError(f"Please provide only one robot state! Obs Keys: {list(obs.keys())}")
        visual = copy(obs)
        for key in list(visual.keys()):
            if key in state_keys or key == "visual_state" or any(tag in key for tag in cls._DROPPED_SUFFIXES):
                del visual[key]
        if not any(k in visual for k in ("xyz", "rgb", "rgbd")):
            if len(visual) != 1:
                raise AssertionError(f"Observations need to contain only one visual element! Obs Keys: {visual.keys()}!")
            visual = next(iter(visual.values()))
        return visual, (obs[state_keys[0]] if state_keys else None)

    def _encode(self, visual, visual_feature, detach_visual):
        if visual_feature is not None:
            return visual_feature
        encoded = self.visual_nn(visual)
        return encoded.detach() if detach_visual else encoded

    def forward(self, obs, actions=None, feature=None, visual_feature=None, prev_actions=None, save_feature=False,
                detach_visual=False, rnn_mode="base", rnn_states=None, episode_dones=None, is_valid=None,
                with_robot_state=True, **kwargs):
        """encoder -> [detach] -> ++ robot state -> ++ (embedded) action -> dense head; `feature=` skips everything up to the
        action, `visual_feature=` only the encoder.  Whenever a feature is handed in, or save_feature is set, copies of the
        visual feature and of the feature with the robot state are left in saved_visual_feature / saved_feature."""
        if feature is not None and visual_feature is not None:
            raise AssertionError("You cannot provide visual_feature and feature at the same time!")
        self.saved_feature = self.saved_visual_feature = None
        keep = save_feature or feature is not None or visual_feature is not None
        visual, robot_state = self.split_obs(obs)
        x = feature
        if x is None:
            x = self._encode(visual, visual_feature, detach_visual)
            if keep:
                self.saved_visual_feature = x.clone()
            if with_robot_state and robot_state is not None:
                assert x.ndim == robot_state.ndim, "Visual feature and state vector should have the same dimension!"
                x = torch.cat((x, robot_state), dim=-1)
            if keep:
                self.saved_feature = x.clone()
        if actions is not None:
            x = torch.cat((x, actions if self.ac_feat is None else self.ac_feat(actions)), dim=-1)
        return x if self.final_mlp is None else self.final_mlp(x)
