"""DrQ agent (data-regularised Q) on the MI355X hot path.

Contract of the reference's pyrl/methods/mfrl/drq.py:21-165 (svea=False): every sample is
augmented `num_aug` times (repeat_interleave, independent noise for obs and next_obs), the TD
target is averaged over a sample's augmentations, the critic trains on all B*num_aug clouds and
the actor on augmentation #0 of every sample.  The augmentation itself is fused into the encoder
kernel's load (pointcloud_rl_amd/augmentations.py).
"""
import torch

from ..augmentations import build_data_augmentations
from ..networks.pointnet import AugmentedObs
from ..utils.torch_utils import to_torch
from .builder import MFRL
from .sac import SAC


def repeat_obs(obs, n):
    """GDict(obs).repeat(n, 0): repeat_interleave of every leaf (reference array_ops.py:106-121)."""
    return {k: torch.repeat_interleave(v, n, dim=0) for k, v in obs.items()}


def first_augmentation(obs, batch_size, num_aug):
    """GDict(obs).split_axis(0, [B, -1]).slice(0, 1): augmentation #0 of every sample (drq.py:115),
    as strided views; the augmentation rows are remapped so the noise is the one the critic saw."""
    out = AugmentedObs({k: v.reshape(batch_size, num_aug, *v.shape[1:])[:, 0] for k, v in obs.items()})
    aug = getattr(obs, "aug", None)
    if aug:
        out.aug = dict(aug, row_mul=num_aug * aug.get("row_mul", 1), row_add=aug.get("row_add", 0))
    return out


@MFRL.register_module()
class DrQ(SAC):
    metric_prefix = "drq"

    def __init__(self, num_aug=2, obs_aug=None, svea=False, inference_aug=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if svea:
            raise NotImplementedError("SVEA is outside the point-cloud DrQ hot path (svea=False in every pn_* config)")
        self.num_aug, self.svea = num_aug, svea
        self.obs_aug = build_data_augmentations(obs_aug)
        self.inference_aug = self.obs_aug if inference_aug == "same" else build_data_augmentations(inference_aug)

    def _augment(self, obs):
        obs = repeat_obs(obs, self.num_aug)
        return self.obs_aug(obs) if self.obs_aug is not None else obs

    @torch.no_grad()
    def forward(self, obs, **kwargs):
        if self.inference_aug is not None:
            obs = self.inference_aug(to_torch(obs, device=self.device))
        return super().forward(obs, **kwargs)

    def _step_body(self, batch, do_actor, polyak):
        B = batch["actions"].shape[0]
        with torch.no_grad():
            obs = self._augment(batch["obs"])
            actions = torch.repeat_interleave(batch["actions"], self.num_aug, dim=0)
            next_obs = self._augment(batch["next_obs"])
            rewards = torch.repeat_interleave(batch["rewards"], self.num_aug, dim=0)
            dones = torch.repeat_interleave(batch["dones"], self.num_aug, dim=0)
        if self._fused is not None:
            return self._fused.run(obs, next_obs, actions, rewards, dones, do_actor, polyak, group=self.num_aug,
                                   actor_obs=first_augmentation(obs, B, self.num_aug) if do_actor else None)
        stats = {}
        q_target = self._q_target(next_obs, rewards, dones, n_groups=B)
        self._critic_step(obs, actions, q_target, stats, polyak=polyak)
        if do_actor:
            self._actor_step(first_augmentation(obs, B, self.num_aug), stats)
        return stats

    def _fused_args(self, batch, do_actor, polyak):
        B = batch["actions"].shape[0]
        with torch.no_grad():
            obs = self._augment(batch["obs"])
            actions = torch.repeat_interleave(batch["actions"], self.num_aug, dim=0)
            next_obs = self._augment(batch["next_obs"])
            rewards = torch.repeat_interleave(batch["rewards"], self.num_aug, dim=0)
            dones = torch.repeat_interleave(batch["dones"], self.num_aug, dim=0)
        return (obs, next_obs, actions, rewards, dones, do_actor, polyak), dict(
            group=self.num_aug, actor_obs=first_augmentation(obs, B, self.num_aug) if do_actor else None)

    def update_parameters(self, memory, updates):
        if self._flat is None:
            self._prepare()
        sampled_batch = memory.sample(self.batch_size).to_torch(device=self.device, non_blocking=True)
        if self.use_episode_dones:
            sampled_batch["dones"] = sampled_batch["episode_dones"]
        return self._run_step(sampled_batch, updates)
