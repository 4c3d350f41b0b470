"""NETWORK / REGRESSION / APPLICATION registries and the actor-critic builders.

Same contract as the reference's pyrl/networks/builder.py:6-73: `build_all` searches the three
registries in order; `build_actor_critic(shared_backbone=True)` injects the actor's visual_nn
object into the critic config; `build_target_network` builds fresh Q heads around the SAME shared
visual_nn, hard-copies the online weights and freezes what is not shared.
"""
from copy import deepcopy

from ..utils.registry import Registry, build_from_cfg

NETWORK = Registry("neural_network")
REGRESSION = Registry("regression")
APPLICATION = Registry("application")

SHARED_KEYS = ["visual_nn", "rnn", "obs_feat", "prev_ac_feat", "recent_frame_feat"]


def build_all(cfg, default_args=None):
    if cfg is None:
        return None
    if isinstance(cfg, (list, tuple)):
        return [build_all(c, default_args) for c in cfg]
    for registry in (NETWORK, REGRESSION, APPLICATION):
        if cfg["type"] in registry.module_dict:
            return build_from_cfg(cfg, registry, default_args)
    raise RuntimeError(f"No this model type:{cfg['type']}!")


def _share_backbone_parts(cfg, donor):
    cfg = deepcopy(cfg)
    for name in SHARED_KEYS:
        item = getattr(donor.backbone, name, None)
        if item is not None:
            cfg["nn_cfg"][f"{name}_cfg"] = None
            cfg["nn_cfg"][name] = item
    return cfg


def build_actor_critic(actor_cfg, critic_cfg, shared_backbone=False):
    actor = build_all(actor_cfg)
    if shared_backbone:
        assert "Visuomotor" in actor_cfg["nn_cfg"]["type"], \
            f"Only Visuomotor model could share backbone. Your model has type {actor_cfg['nn_cfg']['type']}!"
        critic_cfg = _share_backbone_parts(critic_cfg, actor)
    return actor, build_all(critic_cfg)


def build_target_network(network_cfg, network, shared_network=None, shared_backbone=False):
    from ..utils.torch_utils import disable_gradients, hard_update
    if shared_network is None:
        shared_network = network
    if shared_backbone:
        target = build_all(_share_backbone_parts(network_cfg, shared_network))
    else:
        target = deepcopy(network)
    hard_update(target, network)
    disable_gradients(target, exclude=[id(p) for p in network.parameters()])
    return target
