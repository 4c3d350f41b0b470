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

    @staticmethod
    def split_obs(obs):
        """(visual part, robot state): drops *_box/*_seg/*_sem_label/visual_state keys and pops
        "state"/"agent" (visuomotor.py:80-91)."""
        assert isinstance(obs, dict), f"obs is not a dict! {type(obs)}"
        obs = copy(obs)
        for key in list(obs.keys()):
            if "_box" in key or "_seg" in key or "_sem_label" in key or key == "visual_state":
                obs.pop(key)
        robot_state = None
        for key in ("state", "agent"):
            if key in obs:
                assert robot_state is None, f"Please provide only one robot state! Obs Keys: {list(obs.keys())}"
                robot_state = obs.pop(key)
        if not ("xyz" in obs or "rgb" in obs or "rgbd" in obs):
            assert len(obs) == 1, f"Observations need to contain only one visual element! Obs Keys: {obs.keys()}!"
            obs = obs[list(obs.keys())[0]]
        return obs, robot_state

    def forward(self, obs, actions=None, feature=None, visual_feature=None, prev_actions=None, save_feature=False,
                detach_visual=False, rnn_mode="base", rnn_states=None, episode_dones=None, is_valid=None,
                with_robot_state=True, **kwargs):
        assert not (feature is not None and visual_feature is not None), "You cannot provide visual_feature and feature at the same time!"
        self.saved_feature = None
        self.saved_visual_feature = None
        save_feature = save_feature or (feature is not None or visual_feature is not None)
        obs, robot_state = self.split_obs(obs)
        if feature is None:
            if visual_feature is None:
                feat = self.visual_nn(obs)
                if detach_visual:
                    feat = feat.detach()
            else:
                feat = visual_feature
            if save_feature:
                self.saved_visual_feature = feat.clone()
            if robot_state is not None and with_robot_state:
                assert feat.ndim == robot_state.ndim, "Visual feature and state vector should have the same dimension!"
                feat = torch.cat([feat, robot_state], dim=-1)
            if save_feature:
                self.saved_feature = feat.clone()
        else:
            feat = feature
        if actions is not None:
            actions = self.ac_feat(actions) if self.ac_feat is not None else actions
            feat = torch.cat([feat, actions], dim=-1)
        if self.final_mlp is not None:
            feat = self.final_mlp(feat)
        return feat
