"""Agent configs of the shipped point-cloud experiments, with the shape placeholders resolved.

Values restate the reference's config files (configs/mfrl/sac/dm_control/pn.py,
configs/mfrl/sac/maniskill/pn.py, configs/mfrl/drq/{dm_control,maniskill}/{base/pn_base,pn_jitter,
pn_rot}.py); the reference resolves "pcd_all_channel", "action_shape * 2", "128 + agent_shape" ...
from the environment at start-up (pyrl/networks/utils.py:24-119), here they are arguments.
"""
import numpy as np


class Box:
    """Minimal stand-in for gym.spaces.Box (the reference reads .low/.high/.is_bounded())."""

    def __init__(self, low, high, dtype=np.float32):
        self.low, self.high, self.dtype = np.asarray(low, dtype), np.asarray(high, dtype), dtype
        self.shape = self.low.shape

    def is_bounded(self):
        return True


def env_params(obs_shape, action_dim):
    return dict(obs_shape=obs_shape, action_shape=action_dim, action_space=Box(-np.ones(action_dim), np.ones(action_dim)),
                is_discrete=False, message="")


def _agent_cfg(kind, nets, pcd_channels, action_dim, agent_dim, batch_size, gamma, head_hidden, extra):
    mlp_spec, out = nets
    actor_in = out + agent_dim
    cfg = dict(
        type=kind, batch_size=batch_size, gamma=gamma, alpha=0.1, automatic_alpha_tuning=True, ignore_dones=False,
        update_coeff={"default": 0.01, "(.*?)visual_nn(.*?)": 0.05},
        target_update_interval=2, actor_update_interval=2,
        alpha_optim_cfg=dict(type="Adam", lr=1e-3, betas=(0.5, 0.999)),
        shared_backbone=True, detach_actor_feature=True,
        actor_cfg=dict(
            type="ContinuousActor",
            head_cfg=dict(type="TanhGaussianHead", log_std_bound=[-10, 2]),
            nn_cfg=dict(
                type="Visuomotor",
                visual_nn_cfg=dict(type="PointNet", feat_dim=pcd_channels, mlp_spec=list(mlp_spec), out_channels=out,
                                   feature_transform=[], ignore_first_ln=True),
                mlp_cfg=dict(type="LinearMLP", norm_cfg=None, mlp_spec=[actor_in, head_hidden, head_hidden, action_dim * 2],
                             inactivated_output=True),
            ),
            optim_cfg=dict(type="Adam", lr=1e-3, param_cfg={"(.*?)visual_nn(.*?)": None}),
        ),
        critic_cfg=dict(
            type="ContinuousCritic", num_heads=2,
            nn_cfg=dict(type="Visuomotor", visual_nn_cfg=None,
                        mlp_cfg=dict(type="LinearMLP", norm_cfg=None, mlp_spec=[actor_in + action_dim, head_hidden, head_hidden, 1],
                                     inactivated_output=True)),
            optim_cfg=dict(type="Adam", lr=1e-3),
        ),
    )
    cfg.update(extra)
    return cfg


DMC_NETS = ([64, 128, 256], 50)            # configs/mfrl/sac/dm_control/pn.py:25-31
MANISKILL_NETS = ([128, 128, 256], 128)    # configs/mfrl/drq/maniskill/base/pn_base.py:25-31
JITTER = dict(type="RandomJitterPoints", main_key="xyz", req_keys=["xyz"], jitter_range=[-0.01, 0.01])


# "jitter+scale" (BASELINE.json config 3): the reference's scale is a factor on the ROWS of the rotation block (pcd_aug.py:187-189), so
# a scale augmentation is a GlobalRotScaleTrans with a rotation range; values of configs/mfrl/drq/*/pn_rot.py + the class default scale
ROT_SCALE = dict(type="GlobalRotScaleTrans", main_key="xyz", req_keys=["xyz"], rot_range=[-0.15, 0.15], rot_axis="z",
                 scale_ratio_range=[0.95, 1.05], translation_range=None)
MOTIVATING_NETS = ([32, 64, 128], 50)      # configs/mfrl/sac/dm_control/pn_motivating.py:25-31, drq/dm_control/pn_shift_motivating.py


def _aug_cfg(obs_aug):
    """One augmentation dict or a list of them (build_data_augmentations takes either, builder.py:97-104)."""
    if not obs_aug:
        return None
    return [dict(a) for a in obs_aug] if isinstance(obs_aug, (list, tuple)) else dict(obs_aug)


def sac_dmc(pcd_channels=6, action_dim=6, batch_size=256, head_hidden=1024, nets=DMC_NETS, **extra):
    """configs/mfrl/sac/dm_control/pn.py (nets=MOTIVATING_NETS, use_episode_dones=True: pn_motivating.py)"""
    cfg = _agent_cfg("SAC", nets, pcd_channels, action_dim, 0, batch_size, 0.99, head_hidden, dict(extra))
    cfg["critic_cfg"]["nn_cfg"]["mlp_cfg"]["bias"] = True
    return cfg


def sac_maniskill(pcd_channels=7, action_dim=22, agent_dim=68, batch_size=256, head_hidden=1024):
    """configs/mfrl/sac/maniskill/pn.py"""
    cfg = _agent_cfg("SAC", MANISKILL_NETS, pcd_channels, action_dim, agent_dim, batch_size, 0.95, head_hidden, {})
    cfg["actor_cfg"]["nn_cfg"]["mlp_cfg"]["zero_out_indices"] = slice(action_dim, None, None)
    return cfg


def drq_dmc(pcd_channels=6, action_dim=6, batch_size=256, head_hidden=1024, obs_aug=JITTER, num_aug=2, svea=False):
    """configs/mfrl/drq/dm_control/{base/pn_base.py, pn_jitter.py}"""
    cfg = _agent_cfg("DrQ", DMC_NETS, pcd_channels, action_dim, 0, batch_size, 0.95, head_hidden,
                     dict(num_aug=num_aug, svea=svea, obs_aug=_aug_cfg(obs_aug)))
    cfg["critic_cfg"]["nn_cfg"]["mlp_cfg"]["bias"] = True
    return cfg


def drq_maniskill(pcd_channels=7, action_dim=22, agent_dim=68, batch_size=256, head_hidden=1024, obs_aug=JITTER, num_aug=2,
                  encoder_dtype="f32"):
    """configs/mfrl/drq/maniskill/{base/pn_base.py, pn_jitter.py}; encoder_dtype="bf16" = BASELINE.json config 3."""
    cfg = _agent_cfg("DrQ", MANISKILL_NETS, pcd_channels, action_dim, agent_dim, batch_size, 0.95, head_hidden,
                     dict(num_aug=num_aug, svea=False, obs_aug=_aug_cfg(obs_aug)))
    cfg["actor_cfg"]["nn_cfg"]["mlp_cfg"]["zero_out_indices"] = slice(action_dim, None, None)
    cfg["actor_cfg"]["nn_cfg"]["visual_nn_cfg"]["compute_dtype"] = encoder_dtype
    return cfg
