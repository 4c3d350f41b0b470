"""SAC agent on the MI355X hot path.

Constructor keywords, `update_parameters(memory, updates) -> dict[str, float]` and the returned
metric keys follow the reference's pyrl/methods/mfrl/sac.py:26-214.  What differs is how the step
is executed:
  * the shared PointNet is evaluated once per distinct (weights, input) pair -- next_obs once
    (the reference: 3x), obs once with the gradients of both Q heads summed (2x), and once more for
    the actor after the critic step -- instead of once per head;
  * parameters and gradients of each optimizer live in one flat buffer (one RCCL all-reduce per
    backward when data-parallel);
  * all returned metrics come from one device->host copy at the end of the step.
"""
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..augmentations import build_data_augmentations
from ..networks import build_actor_critic, build_target_network
from ..utils.torch_utils import BaseAgent, build_optimizer, select_optimizer_params, soft_update
from .builder import MFRL


class FlatBuffer:
    """Re-homes a list of parameters into one contiguous buffer (data and grad as views)."""

    def __init__(self, named_params):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.data = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        o = 0
        for p in self.params:
            n = p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            o += n

    def zero_grad(self):
        self.grad.zero_()
        o = 0
        for p in self.params:   # re-attach in case something set .grad to None
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad[o:o + n].data_ptr():
                p.grad = self.grad[o:o + n].view(p.shape)
            o += n

    def grad_norm_sq(self):
        return (self.grad * self.grad).sum()


@MFRL.register_module()
class SAC(BaseAgent):
    metric_prefix = "sac"

    def __init__(self, actor_cfg, critic_cfg, env_params, batch_size=128, gamma=0.99, reward_scale=1, update_coeff=0.005,
                 alpha=0.2, alpha_optim_cfg=None, automatic_alpha_tuning=True, target_entropy=None, ignore_dones=False,
                 use_episode_dones=False, target_update_interval=1, actor_update_interval=1, shared_backbone=False,
                 shared_target_backbone=None, detach_actor_feature=False, target_smooth=0.90, pre_process=None):
        super().__init__()
        self.is_discrete = env_params["is_discrete"]
        if self.is_discrete:
            raise NotImplementedError("discrete SAC is outside the point-cloud hot path")
        self.gamma, self.update_coeff, self.alpha, self.reward_scale = gamma, update_coeff, alpha, reward_scale
        self.ignore_dones, self.batch_size = ignore_dones, batch_size
        self.target_update_interval, self.actor_update_interval = target_update_interval, actor_update_interval
        self.automatic_alpha_tuning, self.shared_backbone = automatic_alpha_tuning, shared_backbone
        self.detach_actor_feature, self.use_episode_dones = detach_actor_feature, use_episode_dones

        self.obs_processor = build_data_augmentations(pre_process)
        actor_cfg, critic_cfg = deepcopy([actor_cfg, critic_cfg])
        actor_optim_cfg, critic_optim_cfg = actor_cfg.pop("optim_cfg"), critic_cfg.pop("optim_cfg")
        actor_cfg.update(env_params)
        critic_cfg.update(env_params)
        self.actor, self.critic = build_actor_critic(actor_cfg, critic_cfg, shared_backbone)
        self._actor_optim_cfg, self._critic_optim_cfg = actor_optim_cfg, critic_optim_cfg
        self.actor_optim = build_optimizer(self.actor, actor_optim_cfg)
        self.critic_optim = build_optimizer(self.critic, critic_optim_cfg)
        shared_target_backbone = shared_backbone if shared_target_backbone is None else shared_target_backbone
        self.target_critic = build_target_network(critic_cfg, self.critic, self.actor, shared_target_backbone)
        self.is_recurrent = False

        self.log_alpha = nn.Parameter(torch.ones(1, requires_grad=True))
        self.log_alpha.data *= float(np.log(np.float32(alpha)))
        self.target_entropy = -float(np.prod(env_params["action_shape"])) if target_entropy is None else target_entropy
        if self.automatic_alpha_tuning:
            self.alpha = self.log_alpha.exp().item()
        self.alpha_optim = build_optimizer(self.log_alpha, alpha_optim_cfg)
        self._flat = None
        self.sync_alpha = True     # data-parallel: all-reduce log_alpha's gradient too (the reference does not)

    # ---------------------------------------------------------------------------------------------
    @property
    def encoder(self):
        return getattr(self.actor.backbone, "visual_nn", None)

    def _encoder_is_shared(self):
        enc = self.encoder
        if enc is None:
            return False
        heads = list(self.critic.values) + list(self.target_critic.values)
        return all(getattr(h.backbone, "visual_nn", None) is enc for h in heads)

    def _prepare(self):
        """Lazily (after .to(device)) move every optimizer's parameters into flat buffers."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("pointcloud_rl_amd agents update on MI355X only (agent.to('cuda') first); there is no CPU path")
        self._flat = {
            "critic": FlatBuffer(select_optimizer_params(self.critic, self._critic_optim_cfg.get("param_cfg"))),
            "actor": FlatBuffer(select_optimizer_params(self.actor, self._actor_optim_cfg.get("param_cfg"))),
        }
        self.log_alpha.grad = torch.zeros_like(self.log_alpha)
        self._alpha_t = self.log_alpha.detach().exp()
        self._dedup = self._encoder_is_shared()
        self._world = torch.distributed.get_world_size() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1

    def _allreduce(self, tensor):
        if self._be_data_parallel and self._world > 1:
            torch.distributed.all_reduce(tensor)
            tensor.div_(self._world)

    def _encode(self, module_for_fallback, obs):
        """Visual feature of `obs` when the encoder is shared (else None: the module encodes itself)."""
        if not self._dedup:
            return None
        visual, _ = type(self.actor.backbone).split_obs(obs)
        return self.encoder(visual)

    # ---------------------------------------------------------------------------------------------
    def _q_target(self, next_obs, rewards, dones, n_groups=None):
        with torch.no_grad():
            vis = self._encode(self.actor, next_obs)
            kw = {} if vis is None else dict(visual_feature=vis)
            next_actions, neg_logp = self.actor(next_obs, mode="max-entropy", **kw)
            q_next = self.target_critic(next_obs, actions=next_actions, **kw)
            min_q = torch.min(q_next, dim=-1, keepdim=True).values + self._alpha_t * neg_logp
            r = rewards * self.reward_scale if self.metric_prefix == "sac" else rewards
            q_target = r + self.gamma * min_q if self.ignore_dones else r + (1 - dones.float()) * self.gamma * min_q
            if n_groups is not None:      # DrQ: average the target over the augmentations of a sample (drq.py:83-86)
                q_target = q_target.reshape(n_groups, -1).mean(1, keepdim=True)
                q_target = torch.repeat_interleave(q_target, q_next.shape[0] // n_groups, dim=0)
            return q_target.repeat(1, q_next.shape[-1])

    def _critic_step(self, obs, actions, q_target, stats):
        fb = self._flat["critic"]
        vis = self._encode(self.critic, obs)
        q = self.critic(obs, actions, **({} if vis is None else dict(visual_feature=vis)))
        critic_loss = F.mse_loss(q, q_target) * q_target.shape[-1]
        fb.zero_grad()
        critic_loss.backward()
        self._allreduce(fb.grad)
        self.critic_optim.step()
        if self.encoder is not None:
            self.encoder.invalidate_packed()
        with torch.no_grad():
            stats["critic_loss"] = critic_loss.detach()
            stats["max_critic_abs_err"] = torch.abs(q - q_target).max()
            stats["q"] = torch.min(q, dim=-1).values.mean()
            stats["q_target"] = q_target.mean()
            # 2-norm over all critic parameters == norm of the per-tensor norms (module_utils.py:40-45)
            stats["critic_grad"] = fb.grad_norm_sq().sqrt()

    def _actor_step(self, obs, stats):
        fb = self._flat["actor"]
        kw = {}
        if self._dedup:
            visual, _ = type(self.actor.backbone).split_obs(obs)
            if self.detach_actor_feature:
                with torch.no_grad():
                    kw["visual_feature"] = self.encoder(visual)
            else:
                kw["visual_feature"] = self.encoder(visual)
        pi, neg_logp = self.actor(obs, mode="max-entropy", save_feature=self.shared_backbone,
                                  detach_visual=self.detach_actor_feature, **kw)[:2]
        entropy_term = neg_logp.mean()
        visual_feature = self.actor.backbone.pop_attr("saved_visual_feature")
        if visual_feature is not None:
            visual_feature = visual_feature.detach()
        # Only d(q)/d(pi) is needed from the Q heads here: the reference also accumulates (and later
        # discards, sac.py:141/148) their weight gradients; skip computing them.
        critic_params = [p for p in self.critic.parameters() if p.requires_grad]
        for p in critic_params:
            p.requires_grad_(False)
        try:
            q_pi = self.critic(obs, actions=pi, visual_feature=visual_feature)
        finally:
            for p in critic_params:
                p.requires_grad_(True)
        q_pi = torch.min(q_pi, dim=-1, keepdim=True).values
        actor_loss = -(q_pi.mean() + self._alpha_t * entropy_term)
        fb.zero_grad()
        actor_loss.backward()
        self._allreduce(fb.grad)
        self.actor_optim.step()
        stats["actor_loss"] = actor_loss.detach()
        stats["entropy"] = entropy_term.detach()
        stats["actor_grad"] = fb.grad_norm_sq().sqrt()
        if self.automatic_alpha_tuning:
            alpha_loss = self.log_alpha.exp() * (entropy_term - self.target_entropy).detach()
            self.log_alpha.grad.zero_()
            alpha_loss.backward()
            if self.sync_alpha:
                self._allreduce(self.log_alpha.grad)
            self.alpha_optim.step()
            self._alpha_t = self.log_alpha.detach().exp()
            stats["alpha_loss"] = alpha_loss.detach().reshape(())
        else:
            stats["alpha_loss"] = torch.zeros((), device=self.device)
        stats["new_alpha"] = self._alpha_t.reshape(())

    def _finish(self, stats, updates):
        """One device->host copy for every metric the reference reads with .item() (sac.py:140-203)."""
        keys = list(stats.keys())
        vals = torch.stack([stats[k].reshape(()).float() for k in keys]).tolist()
        got = dict(zip(keys, vals))
        pre = self.metric_prefix
        ret = {f"{pre}/critic_loss": got["critic_loss"], f"{pre}/max_critic_abs_err": got["max_critic_abs_err"],
               f"{pre}/alpha": self.alpha, f"{pre}/q": got["q"], f"{pre}/q_target": got["q_target"],
               f"{pre}/target_entropy": self.target_entropy, f"{pre}/critic_grad": got["critic_grad"], f"{pre}/grad_steps": 1}
        if "actor_loss" in got:
            ret.update({f"{pre}/actor_loss": got["actor_loss"], f"{pre}/alpha_loss": got["alpha_loss"],
                        f"{pre}/entropy": got["entropy"], f"{pre}/actor_grad": got["actor_grad"]})
            if self.automatic_alpha_tuning:
                self.alpha = got["new_alpha"]
        return ret

    def update_parameters(self, memory, updates):
        if self._flat is None:
            self._prepare()
        sampled_batch = memory.sample(self.batch_size).to_torch(device=self.device, non_blocking=True)
        sampled_batch = self.process_obs(sampled_batch)
        if self.use_episode_dones:
            sampled_batch["dones"] = sampled_batch["episode_dones"]
        stats = {}
        q_target = self._q_target(sampled_batch["next_obs"], sampled_batch["rewards"], sampled_batch["dones"])
        self._critic_step(sampled_batch["obs"], sampled_batch["actions"], q_target, stats)
        if updates % self.actor_update_interval == 0:
            self._actor_step(sampled_batch["obs"], stats)
        if updates % self.target_update_interval == 0:
            soft_update(self.target_critic, self.critic, self.update_coeff)
        return self._finish(stats, updates)
