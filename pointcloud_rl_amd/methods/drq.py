"""DrQ agent (data-regularised Q) on the MI355X hot path.

Contract of the reference's pyrl/methods/mfrl/drq.py:21-165.  svea=False (every shipped pn_* config): every sample is
augmented `num_aug` times (repeat_interleave, independent noise for obs and next_obs), the TD
target is averaged over a sample's augmentations, the critic trains on all B*num_aug clouds and
the actor on augmentation #0 of every sample.  The augmentation itself is fused into the encoder
kernel's load (pointcloud_rl_amd/augmentations.py).  svea=True (drq.py:62-67,87-88,115; num_aug must be 1): the critic
trains on [augmented s_b, plain s_b] pairs against ONE target per sample computed from the plain s'_b, the actor on the
plain observations -- no shipped config turns it on, so it runs on the HIP encoder + autograd-heads path, not on the
fused launch sequence.
"""
import os

import torch

from ..augmentations import build_data_augmentations
from ..networks.pointnet import AugmentedObs, materialize
from ..utils.torch_utils import to_torch
from .builder import MFRL
from .sac import SAC


def repeat_obs(obs, n):
    """GDict(obs).repeat(n, 0): repeat_interleave of every leaf (reference array_ops.py:106-121) -- materialised; only the
    autograd fallback path uses it."""
    return {k: torch.repeat_interleave(v, n, dim=0) for k, v in obs.items()}


def virtual_repeat(obs, n):
    """The same batch without the copies: the stored observation plus the factor.  The encoder kernel reads stored cloud
    b // n for cloud b (pcrl_cloud_desc.row_div) and the fused step reads robot state / actions / rewards / dones of sample
    b // n, so DrQ's eleven repeat_interleave launches per step disappear."""
    out = AugmentedObs(obs)
    out.aug = dict(getattr(obs, "aug", None) or {})
    out.repeat = n * int(getattr(obs, "repeat", 1) or 1)
    return out


def first_augmentation(obs, batch_size, num_aug):
    """GDict(obs).split_axis(0, [B, -1]).slice(0, 1): augmentation #0 of every sample (drq.py:115); the augmentation rows
    are remapped so the noise is the one the critic saw.  A virtually repeated batch simply drops the factor; a
    materialised one is viewed with a stride."""
    repeat = int(getattr(obs, "repeat", 1) or 1)
    if repeat > 1:
        assert repeat == num_aug
        out = AugmentedObs(obs)
    else:
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
            assert num_aug == 1, "SVEA only needs num_aug=1"          # drq.py:24-25 (FusedStep.supported declines such an agent)
        self.num_aug, self.svea = num_aug, svea
        self.obs_aug = build_data_augmentations(obs_aug)
        self.inference_aug = self.obs_aug if inference_aug == "same" else build_data_augmentations(inference_aug)

    def _augment(self, obs, virtual=False):
        obs = virtual_repeat(obs, self.num_aug) if virtual else repeat_obs(obs, self.num_aug)
        return self.obs_aug(obs) if self.obs_aug is not None else obs

    def enable_graphs(self, enabled=True, warmup=2):
        """An augmentation whose random draw is made on the host and handed to the kernels by value (ColorJitterPoints:
        torchvision's parameter draw) would be frozen into a captured graph and replayed with the same draw forever; such
        agents keep launching eagerly."""
        host_drawn = [type(t).__name__ for t in (self.obs_aug.transforms if self.obs_aug is not None else [])
                      if not getattr(t, "graph_safe", True)]
        if enabled and host_drawn:
            import warnings
            warnings.warn(f"hipGraph replay is off: {host_drawn} draw their parameters on the host every step")
            enabled = False
        super().enable_graphs(enabled, warmup)

    @torch.no_grad()
    def forward(self, obs, **kwargs):
        if self.inference_aug is not None:
            obs = self.inference_aug(to_torch(obs, device=self.device))
        return super().forward(obs, **kwargs)

    def _step_body(self, batch, do_actor, polyak):
        B = batch["actions"].shape[0]
        if self._fused is not None:
            args, kwargs = self._fused_args(batch, do_actor, polyak)
            return self._fused.run(*args, **kwargs)
        if self.svea:
            with torch.no_grad():
                # GDict.stack([aug(obs), obs], axis=1).merge_axes([0, 1]) (drq.py:64-65): rows 2b = augmented, 2b + 1 = plain
                aug = materialize(self._augment(batch["obs"]))
                obs = {k: torch.stack([aug[k].to(v.dtype), v], dim=1).flatten(0, 1) for k, v in batch["obs"].items()}
                actions = torch.repeat_interleave(batch["actions"], 2, dim=0)
            stats = {}
            # next_obs, rewards and dones stay un-augmented, one target per sample, repeated over the pair (drq.py:69-88)
            q_target = torch.repeat_interleave(self._q_target(batch["next_obs"], batch["rewards"], batch["dones"]), 2, dim=0)
            self._critic_step(obs, actions, q_target, stats, polyak=polyak)
            if do_actor:
                self._actor_step(batch["obs"], stats)                  # drq.py:115: the plain observations
            return stats
        with torch.no_grad():
            obs = self._augment(batch["obs"])
            actions = torch.repeat_interleave(batch["actions"], self.num_aug, dim=0)
            next_obs = self._augment(batch["next_obs"])
            rewards = torch.repeat_interleave(batch["rewards"], self.num_aug, dim=0)
            dones = torch.repeat_interleave(batch["dones"], self.num_aug, dim=0)
        stats = {}
        q_target = self._q_target(next_obs, rewards, dones, n_groups=B)
        self._critic_step(obs, actions, q_target, stats, polyak=polyak)
        if do_actor:
            self._actor_step(first_augmentation(obs, B, self.num_aug), stats)
        return stats

    def _fused_args(self, batch, do_actor, polyak):
        """Nothing is repeated: observations carry the factor (virtual_repeat), actions / rewards / dones stay one row per
        sample and the kernels index them by row // num_aug (`repeat=`)."""
        B = batch["actions"].shape[0]
        with torch.no_grad():
            obs = self._augment(batch["obs"], virtual=True)
            next_obs = self._augment(batch["next_obs"], virtual=True)
        return (obs, next_obs, batch["actions"], batch["rewards"], batch["dones"], do_actor, polyak), dict(
            group=self.num_aug, repeat=self.num_aug, actor_obs=first_augmentation(obs, B, self.num_aug) if do_actor else None)

    def _entry_shape(self):
        return (self.batch_size * self.num_aug, self.num_aug) if not self.svea else (self.batch_size, 1)

    def update_parameters(self, memory, updates):
        """SAC's, with the jitter augmentations told where the step's device draw counter lives: a replay whose sampling is one
        device launch (`DeviceReplay.graph_sampling`) advances `state[0]` once per sample -- the Philox offset of every jitter
        call of this step (calls differ by seed), instead of a counter increment + clone per call."""
        jitters = [t for t in (self.obs_aug.transforms if self.obs_aug is not None else []) if hasattr(t, "begin_step")]
        shared = memory.state[:1] if (jitters and getattr(memory, "graph_sampling", False) and torch.is_tensor(getattr(memory, "state", None))
                                      ) else None
        # A captured encoder launch holds the counter's ADDRESS.  The shared counter is used only where a captured step would also
        # hold the replay's sampling launch (no obs_processor, PCRL_GRAPH_SAMPLING on: SAC._run_step's rule; the same rule eagerly,
        # so that an eager and a graph-replayed run draw the same noise), and a graph captured with one replay's counter must not
        # be replayed against another replay (or none): a changed address drops the captured graphs.
        if not (self.graph_sampling and self.obs_processor is None):
            shared = None
        if getattr(self, "_use_graphs", False):
            ptr = None if shared is None else shared.data_ptr()
            if self.__dict__.get("_graphs") and self.__dict__.get("_jitter_counter_ptr", ptr) != ptr:
                self._graphs, self._graph_seen, self._fast = {}, {k: self._graph_warmup for k in self._graph_seen}, None
                self._graph_sampler, self._graph_flag = {}, {}
            self._jitter_counter_ptr = ptr
        for t in jitters:
            t.begin_step(shared)
        try:
            return super().update_parameters(memory, updates)
        finally:
            for t in jitters:
                t.begin_step(None)

    _process_sampled_obs = False        # drq.py:46-49 samples without process_obs; update_parameters itself is SAC's
