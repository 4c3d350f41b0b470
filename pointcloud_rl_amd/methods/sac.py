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
import contextlib
import gc
import os
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from ..augmentations import build_data_augmentations
from ..networks import build_actor_critic, build_target_network
from ..utils.dist import Exchange, allreduce_sum_, capture_exchange, exchange_active, quiesce_before_capture, world_size
from ..utils.torch_utils import BaseAgent, build_optimizer, regex_match, select_optimizer_params, soft_update
from .builder import MFRL


@contextlib.contextmanager
def _no_gc():
    """Python's cyclic collector must not run inside a stream capture: collecting a cycle that holds a tensor allocated
    outside the graph's pool frees it mid-capture and the caching allocator aborts the process (seen as a flaky
    'Fatal Python error: Aborted ... Garbage-collecting' under torch.cuda.graph, which only collects once on entry)."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class FlatBuffer:
    """Re-homes a list of parameters into one contiguous buffer (data and grad as views).  Every
    tensor starts on a 16-byte boundary so that the GEMM kernels can use 16-byte operand loads; the
    padding floats stay zero (zero gradient -> Adam leaves them at zero)."""

    ALIGN = 4    # floats

    def __init__(self, named_params, with_grad=True):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        dev = self.params[0].device
        self.offsets, o = [], 0
        for p in self.params:
            self.offsets.append(o)
            o += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = o
        self.data = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev) if with_grad else None
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            if with_grad:
                p.grad = self.grad[o:o + n].view(p.shape)

    def offset_of(self, name):
        return self.offsets[self.names.index(name)]

    def views(self, flat_tensor):
        return [flat_tensor[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]

    def zero_grad(self):
        self.grad.zero_()
        for p, g in zip(self.params, self.views(self.grad)):   # re-attach in case something set .grad to None
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def grad_norm_sq(self):
        return (self.grad * self.grad).sum()


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam-compatible optimizer over the fused flat-buffer kernel (pcrl_adam_step_f32).

    A real `torch.optim.Optimizer`: the reference's checkpoint code only saves / restores attributes that pass
    `isinstance(child, Optimizer)` (checkpoint_utils.py:59-71, 226-229), and train_rl.py:392-405 calls it on the agent after
    the first update.  Keeps what the reference's drivers touch: `param_groups` (one group per tensor, as
    build_optimizer makes them, optimizer_utils.py:43-57), `state_dict()/load_state_dict()` in torch.optim.Adam's format,
    `zero_grad()`, `step()`.  The moments live in two flat buffers next to the flat parameter buffer; `self.state`
    (torch's per-parameter dict) stays empty -- `state_dict()` builds the per-parameter views on demand.
    """

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, **unused):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False)
        super().__init__([dict(params=[p]) for p in flat.params], defaults)
        self.flat = flat
        dev = flat.data.device
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.workspace = torch.empty(hip.adam_workspace_bytes(flat.data.numel()), dtype=torch.uint8, device=dev)

    _OWN_STATE = ("flat", "exp_avg", "exp_avg_sq", "step_counter", "grad_norm", "workspace")

    def __getstate__(self):
        """pickle / copy.deepcopy: torch's Optimizer.__getstate__ keeps defaults / state / param_groups only -- the flat buffers, the
        moments and the device step count are this optimizer's state and travel with it.  (A deep copy owns copies of the flat
        buffers; the copied param_groups still name the ORIGINAL module's tensors, as for any torch optimizer copied without its module:
        copy the agent, not the optimizer.)"""
        state = super().__getstate__()
        state.update({k: getattr(self, k) for k in self._OWN_STATE})
        return state

    def __setstate__(self, state):
        own = {k: state.pop(k) for k in self._OWN_STATE if k in state}
        super().__setstate__(state)
        self.__dict__.update(own)

    def zero_grad(self, set_to_none=False):
        """Gradients are views of the flat gradient buffer the kernels write: zeroed in place, never set to None.  (The class-level
        `step` is wrapped by torch's profile hook like every Optimizer's: a `record_function` per EAGER step; a replayed hipGraph
        does not pass through it.)"""
        self.flat.zero_grad()

    def hyper(self):
        """(lr, beta1, beta2, eps) of the one fused pass.  The reference's build_optimizer gives every tensor its own group
        with the same values (optimizer_utils.py:43-57); groups that differ cannot be served by one flat launch."""
        g = self.param_groups[0]
        sig = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
        for other in self.param_groups[1:]:
            if (float(other["lr"]), float(other["betas"][0]), float(other["betas"][1]), float(other["eps"])) != sig:
                raise NotImplementedError("HipAdam: per-tensor lr / betas / eps differ between param_groups; the fused optimizer "
                                          "applies one set of hyper-parameters to its whole flat buffer")
        return sig

    def rider_args(self, grad_scale=1.0):
        """This (small) optimizer as the rider of another one's launch (hip.adam_step(rider=...))."""
        g = self.param_groups[0]
        self.hyper()
        return dict(param=self.flat.data, grad=self.flat.grad, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lr=g["lr"], beta1=g["betas"][0],
                    beta2=g["betas"][1], eps=g["eps"], grad_scale=grad_scale, step_counter=self.step_counter, grad_norm_out=self.grad_norm,
                    partial=self.workspace)

    def step(self, grad_scale=1.0, target=None, target_range=(0, 0), tau=0.0, defer=False, rider=None):
        """defer=True returns the pending second half (gradient norm, step count) for hip.gather_scalars(pending=...).
        lr / betas / eps are kernel arguments: a launch captured in a hipGraph keeps the values it was captured with, which
        is why SAC._run_step drops its graphs when an optimizer's hyper() changes (scheduler, load_state_dict)."""
        if callable(grad_scale):       # torch's `optimizer.step(closure)` form
            raise NotImplementedError("HipAdam.step takes no closure")
        g = self.param_groups[0]
        self.hyper()
        return hip.adam_step(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0], g["betas"][1],
                             g["eps"], grad_scale, self.step_counter, self.grad_norm, self.workspace,
                             target=target, target_begin=target_range[0], target_end=target_range[1], tau=tau, defer=defer, rider=rider)

    def norm_first(self, grad_scale=1.0, rider=None):
        """First half of a pass that is published before it runs: the gradient norm's partial sums (and the rider's whole pass) now, the
        pending half for hip.gather_scalars; `step_published` launches the pass itself after that gather launch."""
        self.hyper()
        return hip.grad_norm_partials(self.flat.grad, grad_scale, self.step_counter, self.grad_norm, self.workspace, rider=rider)

    def step_published(self, grad_scale=1.0, target=None, target_range=(0, 0), tau=0.0, gather=None):
        g = self.param_groups[0]
        self.hyper()
        hip.adam_step_published(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                grad_scale, self.step_counter, target=target, target_begin=target_range[0], target_end=target_range[1], tau=tau,
                                gather=gather)

    def _views(self, flat_tensor):
        return self.flat.views(flat_tensor)

    def state_dict(self):
        """torch.optim.Adam's layout: {"state": {i: {step, exp_avg, exp_avg_sq}}, "param_groups": [{..., "params": [i]}]}."""
        step = self.step_counter.to(torch.float32).reshape(())
        state = {i: dict(step=step.clone(), exp_avg=m, exp_avg_sq=v)
                 for i, (m, v) in enumerate(zip(self._views(self.exp_avg), self._views(self.exp_avg_sq)))} if int(self.step_counter.item()) > 0 else {}
        groups = [dict({k: v for k, v in g.items() if k != "params"}, params=[i]) for i, g in enumerate(self.param_groups)]
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        saved = sd.get("param_groups", [])
        if saved and len(saved) != len(self.param_groups):
            raise ValueError(f"loaded state dict has {len(saved)} parameter groups, the optimizer has {len(self.param_groups)}")
        for i, st in sd.get("state", {}).items():
            i = int(i)
            self._views(self.exp_avg)[i].copy_(st["exp_avg"])
            self._views(self.exp_avg_sq)[i].copy_(st["exp_avg_sq"])
            self.step_counter.fill_(int(st["step"]))
        for g, src in zip(self.param_groups, saved):
            g.update({k: v for k, v in src.items() if k != "params"})


def _is_capture_error(err):
    """True for what HIP / RCCL / torch raise when the EXCHANGING step cannot be captured whole: an operation refused while a stream is
    being captured (hipErrorStreamCapture*, "operation not permitted when stream is capturing", "... is capturing"), or a collective
    that failed inside the capture (ProcessGroupNCCL's "NCCL error ... unhandled cuda / system error", RCCL's own messages).  Only the
    capture of a step WITH its all-reduces asks this (SAC._run_step), so a collective's failure there is fallback-eligible; out of
    memory, assertions and argument errors of this library ("pcrl") are not."""
    if isinstance(err, (AssertionError, NotImplementedError)):
        return False
    text = str(err).lower()
    if "out of memory" in text or "pcrl_e_" in text:
        return False
    capture = any(k in text for k in ("streamcapture", "stream capture", "is capturing", "capturing stream", "while capturing", "during capture",
                                      "stream is capturing", "captures_underway", "capture_begin", "capture_end"))
    collective = any(k in text for k in ("nccl error", "rccl error", "ncclinternalerror", "ncclunhandledcudaerror", "ncclsystemerror",
                                         "unhandled cuda error", "unhandled hip error", "unhandled system error"))
    return capture or collective


def _plain_adam(optim):
    if type(optim) is not torch.optim.Adam:
        return None
    d = optim.defaults
    if d.get("amsgrad") or d.get("weight_decay") or d.get("maximize"):
        return None
    return dict(lr=d["lr"], betas=d["betas"], eps=d["eps"])


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
        self.graph_sampling = True   # a device replay's sampling launch becomes the first node of the captured step (tests switch it off)
        self.use_fused_step = True   # autograd-free launch sequence (methods/fused.py) when the topology allows
        self.sync_alpha = True     # data-parallel: all-reduce log_alpha's gradient too (the reference does not)

    # -- acting ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, obs, **kwargs):
        """BaseAgent.forward (module_utils.py:147-159).  On the MI355X and with the shipped actor topology the action comes from
        the fused acting path (methods/acting.py: five launches, one hipGraph launch after enable_graphs()); anything else goes
        through the module tree."""
        from ..utils.torch_utils import to_torch
        from .acting import MEAN_MODES, SAMPLE_MODES, FusedActor
        mode = kwargs.get("mode", "explore")
        extra = {k: v for k, v in kwargs.items() if k not in ("mode", "rnn_mode", "rnn_states", "prev_actions", "episode_dones", "is_valid")}
        fast = self.__dict__.get("_fused_actor")
        if fast is None and self.device.type == "cuda" and not extra and mode in SAMPLE_MODES + MEAN_MODES and FusedActor.supported(self.actor) \
                and getattr(self, "use_fused_acting", True):
            fast = self.__dict__["_fused_actor"] = FusedActor(self.actor)
            fast.use_graphs = bool(getattr(self, "_use_graphs", False))
        if fast is None or extra or mode not in SAMPLE_MODES + MEAN_MODES or self.device.type != "cuda" or kwargs.get("num_samples", 1) != 1:
            return super().forward(obs, **kwargs)
        obs = to_torch(obs, device=self.device, non_blocking=True)
        if self.obs_processor is not None:
            obs = self.obs_processor({"obs": obs})["obs"]
        actions = fast(obs, mode=mode)
        rnn_mode = kwargs.get("rnn_mode", "base")
        if rnn_mode == "base":
            return actions
        return actions, ([None] * 3 if rnn_mode == "full_states" else None)

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
        if self.device.type != "cuda":
            raise RuntimeError("pointcloud_rl_amd agents update on MI355X only (agent.to('cuda') first); there is no CPU path")
        self._prepare_buffers()
        self._prepare_schedule()

    def _prepare_buffers(self):
        """Flat parameter / gradient buffers and the fused optimizers that replace torch.optim.Adam (plain tensor bookkeeping:
        runs on any device, which is how the container-side integration probe checks the reference's checkpoint functions
        against the post-update optimizers without a GPU)."""
        dev = self.device
        self._flat = {
            "critic": FlatBuffer(select_optimizer_params(self.critic, self._critic_optim_cfg.get("param_cfg"))),
            "actor": FlatBuffer(select_optimizer_params(self.actor, self._actor_optim_cfg.get("param_cfg"))),
        }
        self._flat["alpha"] = FlatBuffer([("log_alpha", self.log_alpha)])
        # the temperature's gradient lives right behind the actor's: their data-parallel exchange is ONE all-reduce
        fa, fal = self._flat["actor"], self._flat["alpha"]
        self._actor_alpha_grad = torch.zeros(fa.total + fal.total, dtype=torch.float32, device=dev)
        fa.grad, fal.grad = self._actor_alpha_grad[:fa.total], self._actor_alpha_grad[fa.total:]
        fa.zero_grad(), fal.zero_grad()                # re-attaches every p.grad to its view of the joint buffer
        for name in ("critic", "actor", "alpha"):      # torch.optim.Adam (one group per tensor) -> one fused launch
            old = getattr(self, f"{name}_optim")
            hp = _plain_adam(old)
            if hp is not None:
                fused = HipAdam(self._flat[name], **hp)
                if old.state:                          # a checkpoint was loaded before the first update: keep moments and step
                    fused.load_state_dict(old.state_dict())
                setattr(self, f"{name}_optim", fused)

    def _prepare_schedule(self):
        # Polyak: the target's own (non-shared) parameters mirror a tail range of the critic buffer
        online = {id(p) for p in self.critic.parameters()}
        tgt = [(n, p) for n, p in self.target_critic.named_parameters() if id(p) not in online]
        self._target_flat, self._target_range, self._target_tau = None, (0, 0), None
        own = [(n, p) for n, p in zip(self._flat["critic"].names, self._flat["critic"].params)
               if n in {tn for tn, _ in tgt}]
        taus = {self._tau_for(n) for n, _ in own}
        if tgt and len(own) == len(tgt) and len(taus) == 1 and [n for n, _ in own] == [n for n, _ in tgt]:
            fc = self._flat["critic"]
            first = fc.names.index(own[0][0])
            if first + len(own) == len(fc.params):
                target_flat = FlatBuffer(tgt, with_grad=False)
                begin = fc.offsets[first]
                if target_flat.total == fc.total - begin:      # same relative layout (identical padding)
                    self._target_flat, self._target_range, self._target_tau = target_flat, (begin, fc.total), taus.pop()
        self._alpha_t = self.log_alpha.detach().exp()
        self._dedup = self._encoder_is_shared()
        # Encoders whose weights an optimizer writes: the fused Adam kernel updates parameters through raw pointers
        # (no autograd version bump), so each of them must re-pack its MFMA weight image after that optimizer's step --
        # with separate backbones (shared_backbone=False) these are the Q heads' own PointNets, not self.encoder.
        from ..networks.pointnet import PointNet
        self._packed_owners = {}
        for name, module in (("critic", self.critic), ("actor", self.actor)):
            ids = {id(p) for p in self._flat[name].params}
            seen, owners = set(), []
            for m in module.modules():
                if isinstance(m, PointNet) and id(m) not in seen and any(id(p) in ids for p in m.parameters()):
                    seen.add(id(m))
                    owners.append(m)
            self._packed_owners[name] = owners
        from .fused import FusedStep
        self._fused = FusedStep(self) if (self.use_fused_step and FusedStep.supported(self)) else None
        self._world = torch.distributed.get_world_size() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1

    def _tau_for(self, name):
        """soft_update's regex -> tau rule (pyrl/utils/torch/ops.py:66-90)."""
        if not isinstance(self.update_coeff, dict):
            return float(self.update_coeff)
        for pattern, value in self.update_coeff.items():
            if pattern != "default" and regex_match(name, pattern):
                return float(value)
        return float(self.update_coeff["default"])

    def _allreduce(self, tensor):
        """Sum the flat gradient over the ranks (RCCL); the 1/world factor is applied by the optimizer kernel."""
        return allreduce_sum_(tensor, enabled=self._be_data_parallel)

    def _optim_step(self, name, scale, polyak=False, pending=None, rider=None):
        """pending: a list -> the optimizer's second half (gradient norm, step count) is deferred and appended to it.
        rider = (name, scale) of a second (small, fused) optimizer stepped by the same launch -- the temperature next to the actor."""
        opt = getattr(self, f"{name}_optim")
        fb = self._flat[name]
        if rider is not None and not (isinstance(opt, HipAdam) and isinstance(getattr(self, f"{rider[0]}_optim"), HipAdam)
                                      and self._flat[rider[0]].total <= 4096 and not polyak):
            self_norm = self._optim_step(name, scale, polyak=polyak, pending=pending)
            self._optim_step(rider[0], rider[1], pending=pending)
            return self_norm
        if isinstance(opt, HipAdam):
            defer = pending is not None
            if rider is not None:
                pend, r_pend = opt.step(scale, defer=defer, rider=getattr(self, f"{rider[0]}_optim").rider_args(rider[1]))
                if defer:
                    pending.append(r_pend)
            elif polyak and self._target_flat is not None:
                pend = opt.step(scale, target=self._target_flat.data, target_range=self._target_range, tau=self._target_tau, defer=defer)
            else:
                pend = opt.step(scale, defer=defer)
            if defer:
                pending.append(pend)
            for enc in self._packed_owners.get(name, ()):
                enc.invalidate_packed()
            return opt.grad_norm.reshape(())
        if scale != 1.0:
            fb.grad.mul_(scale)
        opt.step()
        for enc in self._packed_owners.get(name, ()):
            enc.invalidate_packed()
        return fb.grad_norm_sq().sqrt()

    def _optim_norm_first(self, name, scale, pending, rider=None):
        """The step's LAST optimizer pass, split so that the metrics leave before it (csrc/optim.hip: gradnorm_kernel): its gradient norm's
        partial sums (+ the temperature's whole pass as a rider) now; returns (norm tensor, finish) -- `finish()` launches the pass itself
        and must be called right after the step's hip.gather_scalars.  None when this optimizer cannot be split (not the fused one)."""
        opt = getattr(self, f"{name}_optim")
        if not isinstance(opt, HipAdam) or pending is None:
            return None
        r_opt = getattr(self, f"{rider[0]}_optim") if rider is not None else None
        if rider is not None and not (isinstance(r_opt, HipAdam) and self._flat[rider[0]].total <= 4096):
            return None
        if rider is not None:
            pend, r_pend = opt.norm_first(scale, rider=r_opt.rider_args(rider[1]))
            pending.append(r_pend)
        else:
            pend = opt.norm_first(scale)
        pending.append(pend)
        polyak = name == "critic"

        def finish(do_polyak, gather=None):
            if do_polyak and polyak and self._target_flat is not None:
                opt.step_published(scale, target=self._target_flat.data, target_range=self._target_range, tau=self._target_tau, gather=gather)
            else:
                opt.step_published(scale, gather=gather)
            for enc in self._packed_owners.get(name, ()):
                enc.invalidate_packed()
        return opt.grad_norm.reshape(()), finish

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

    def _critic_step(self, obs, actions, q_target, stats, polyak=False):
        fb = self._flat["critic"]
        vis = self._encode(self.critic, obs)
        q = self.critic(obs, actions, **({} if vis is None else dict(visual_feature=vis)))
        critic_loss = F.mse_loss(q, q_target) * q_target.shape[-1]
        fb.zero_grad()
        critic_loss.backward()
        scale = self._allreduce(fb.grad)
        grad_norm = self._optim_step("critic", scale, polyak=polyak)      # re-pack of the encoders it owns: _optim_step
        with torch.no_grad():
            stats["critic_loss"] = critic_loss.detach()
            stats["max_critic_abs_err"] = torch.abs(q - q_target).max()
            stats["q"] = torch.min(q, dim=-1).values.mean()
            stats["q_target"] = q_target.mean()
            # 2-norm over all critic parameters == norm of the per-tensor norms (module_utils.py:40-45)
            stats["critic_grad"] = grad_norm

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
        stats["actor_grad"] = self._optim_step("actor", self._allreduce(fb.grad))
        stats["actor_loss"] = actor_loss.detach()
        stats["entropy"] = entropy_term.detach()
        if self.automatic_alpha_tuning:
            alpha_loss = self.log_alpha.exp() * (entropy_term - self.target_entropy).detach()
            self._flat["alpha"].zero_grad()
            alpha_loss.backward()
            self._optim_step("alpha", self._allreduce(self._flat["alpha"].grad) if self.sync_alpha else 1.0)
            self._alpha_t.copy_(self.log_alpha.detach().exp())     # in place: captured graphs read this tensor
            stats["alpha_loss"] = alpha_loss.detach().reshape(())
        else:
            stats["alpha_loss"] = torch.zeros((), device=self.device)
        stats["new_alpha"] = self._alpha_t.reshape(()).clone()

    def _ret_template(self, keys):
        """[(returned key, index into the step's metric vector | None, constant)] in the reference's order (sac.py:150-159,199-204),
        cached per metric-name tuple."""
        cache = self.__dict__.setdefault("_ret_templates", {})
        keys = tuple(keys)
        if keys not in cache:
            pre, at = self.metric_prefix, {k: i for i, k in enumerate(keys)}
            tpl = [(f"{pre}/critic_loss", at["critic_loss"], None), (f"{pre}/max_critic_abs_err", at["max_critic_abs_err"], None),
                   (f"{pre}/alpha", None, "alpha"), (f"{pre}/q", at["q"], None), (f"{pre}/q_target", at["q_target"], None),
                   (f"{pre}/target_entropy", None, self.target_entropy), (f"{pre}/critic_grad", at["critic_grad"], None),
                   (f"{pre}/grad_steps", None, 1)]
            if "actor_loss" in at:
                tpl += [(f"{pre}/actor_loss", at["actor_loss"], None), (f"{pre}/alpha_loss", at["alpha_loss"], None),
                        (f"{pre}/entropy", at["entropy"], None), (f"{pre}/actor_grad", at["actor_grad"], None)]
            new_alpha = at.get("new_alpha") if ("actor_loss" in at and self.automatic_alpha_tuning) else None
            cache[keys] = (tpl, new_alpha)
        return cache[keys]

    def _ret_from_values(self, keys, vals):
        tpl, new_alpha = self._ret_template(keys)
        alpha = self.alpha                        # the value the step ran with (sac.py:153 reads it before the temperature update)
        ret = {k: (vals[i] if i is not None else (alpha if c == "alpha" else c)) for k, i, c in tpl}
        if new_alpha is not None:
            self.alpha = vals[new_alpha]
        return ret

    def _finish(self, stats, updates, host_values=None):
        """One device->host copy for every metric the reference reads with .item() (sac.py:140-203)."""
        keys = list(stats.keys())
        if host_values is not None:
            vals = host_values
        elif getattr(stats, "packed", None) is not None:
            vals = stats.packed.tolist()
        else:
            vals = torch.stack([stats[k].reshape(()).float() for k in keys]).tolist()
        return self._ret_from_values(keys, vals)

    def _polyak_now(self, updates):
        """True when this step's target update is fused into the critic's optimizer pass.  The critic's
        weights do not change between its optimizer step and the reference's soft_update call at the end
        of the step (sac.py:207-208), so updating the target right after the step is equivalent."""
        return updates % self.target_update_interval == 0 and self._target_flat is not None and isinstance(self.critic_optim, HipAdam)

    def _soft_update(self, updates):
        if updates % self.target_update_interval == 0 and not self._polyak_now(updates):
            soft_update(self.target_critic, self.critic, self.update_coeff)

    # -- step execution: eager, or replayed from a hipGraph --------------------------------------
    def _step_body(self, batch, do_actor, polyak):
        """SAC step on a device-resident batch; returns the dict of device scalars for the metrics."""
        if self._fused is not None:
            return self._fused.run(batch["obs"], batch["next_obs"], batch["actions"], batch["rewards"], batch["dones"], do_actor, polyak)
        stats = {}
        q_target = self._q_target(batch["next_obs"], batch["rewards"], batch["dones"])
        self._critic_step(batch["obs"], batch["actions"], q_target, stats, polyak=polyak)
        if do_actor:
            self._actor_step(batch["obs"], stats)
        return stats

    def enable_graphs(self, enabled=True, warmup=2):
        """Capture the whole update step in a hipGraph (one graph per (actor-update?, target-update?)
        combination) and replay it: per-step host work drops to copying the batch into static buffers,
        one graph launch and one device->host copy.  Requires device-resident state only, which is why
        alpha, the Adam step counts and the Philox offsets live in device memory."""
        self._use_graphs, self._graph_warmup = enabled, warmup
        if self.__dict__.get("_fused_actor") is not None:      # the acting path replays its own graphs under the same switch
            self._fused_actor.use_graphs, self._fused_actor.graphs, self._fused_actor._seen = bool(enabled), {}, {}
        self._graphs, self._graph_seen, self._static_batch = {}, {}, None
        self._graph_sampler, self._graph_flag, self._fast = {}, {}, None

    def _to_static(self, batch):
        """Copy `batch` into buffers whose addresses the captured graphs refer to."""
        keys = ("obs", "next_obs", "actions", "rewards", "dones")
        if self._static_batch is None:
            # a device replay hands out long-lived staging tensors: read them in place instead of cloning
            keep = (lambda t: t) if getattr(batch, "persistent", False) else (lambda t: t.clone())
            self._static_batch = {k: ({kk: keep(vv) for kk, vv in batch[k].items()} if isinstance(batch[k], dict) else keep(batch[k]))
                                  for k in keys}
            return self._static_batch
        for k in keys:
            src, dst = batch[k], self._static_batch[k]
            for kk in (src if isinstance(src, dict) else [None]):
                s_, d_ = (src[kk], dst[kk]) if kk is not None else (src, dst)
                if s_.data_ptr() != d_.data_ptr():
                    d_.copy_(s_, non_blocking=True)
        return self._static_batch

    def _aliases_static(self, batch):
        """True when every leaf of `batch` that the step reads IS the matching leaf of the static batch (same address)."""
        for k in ("obs", "next_obs", "actions", "rewards", "dones"):
            src, dst = batch[k], self._static_batch[k]
            if isinstance(src, dict) != isinstance(dst, dict):
                return False
            for kk in (src if isinstance(src, dict) else [None]):
                s_, d_ = (src[kk], dst.get(kk)) if kk is not None else (src, dst)
                if d_ is None or s_.data_ptr() != d_.data_ptr() or s_.shape != d_.shape:
                    return False
        return True

    def _fused_args(self, batch, do_actor, polyak):
        return (batch["obs"], batch["next_obs"], batch["actions"], batch["rewards"], batch["dones"], do_actor, polyak), {}

    def _entry_shape(self):
        """(rows of the critic phase, group) of a step on `batch_size` samples: FusedStep.attach_entry's arguments."""
        return self.batch_size, 1

    def _run_step(self, batch, updates, sampler=None):
        """batch: the sampled batch, or a callable `fetch(launch=True)` returning it -- then `sampler` is the replay it samples
        from, and when that replay's sampling is one host-free launch (`DeviceReplay.graph_sampling`) the launch becomes the
        first node of the captured step: replays call `fetch(launch=False)` (bookkeeping only)."""
        fetch = batch if callable(batch) else (lambda launch=True: batch)
        # a pre-processor produces fresh tensors outside the graph: its output cannot be the captured step's input in place
        if not (callable(batch) and getattr(sampler, "graph_sampling", False) and self.graph_sampling
                and self.obs_processor is None):
            sampler = None
        do_actor = updates % self.actor_update_interval == 0
        polyak = self._polyak_now(updates)
        exchanging = self._be_data_parallel and exchange_active()
        graphable = getattr(self, "_use_graphs", False) and (not (updates % self.target_update_interval == 0) or polyak) \
            and (not exchanging or self._fused is not None)
        if not graphable:
            stats = self._step_body(fetch(), do_actor, polyak)
            self._soft_update(updates)
            return self._finish({k: v for k, v in stats.items()}, updates)
        # launches bake lr / betas / eps in as kernel arguments: a changed hyper-parameter invalidates the captured graphs
        hyper = tuple(opt.hyper() for opt in (self.critic_optim, self.actor_optim, self.alpha_optim) if isinstance(opt, HipAdam))
        if hyper != getattr(self, "_graph_hyper", hyper):
            self._graphs, self._graph_seen, self._fast = {}, {k: self._graph_warmup for k in self._graph_seen}, None
        self._graph_hyper = hyper
        key = (do_actor, polyak, exchanging)       # a step with exchanges is cut into segments, one without is one graph
        if key in self._graphs and self._graph_sampler.get(key) not in (None, sampler):
            # this variant was captured with another replay's sampling launch inside: capture again
            self._graphs, self._graph_seen, self._fast = {}, {k: self._graph_warmup for k in self._graph_seen}, None
        if key not in self._graphs:
            seen = self._graph_seen.get(key, 0)
            self._graph_seen[key] = seen + 1
            if seen < self._graph_warmup:          # eager warm-up: lazy initialisation must not be captured
                return self._finish(self._step_body(self._to_static(fetch()), do_actor, polyak), updates)
            for enc in {id(e): e for owners in self._packed_owners.values() for e in owners}.values():
                enc.invalidate_packed()            # every replay starts by re-packing the (updated) weights
            if sampler is not None:
                # The captured sampling launch writes the replay's staging tensors, the captured step reads _static_batch: they
                # must be the SAME memory.  They are not when the static batch was made from another replay object or from a
                # non-persistent batch (clones) -- the graphs captured so far then read buffers this replay never fills.
                staged = fetch(launch=False)
                if self._static_batch is not None and not self._aliases_static(staged):
                    self._graphs, self._graph_seen, self._fast = {}, {k: self._graph_warmup for k in self._graph_seen}, None
                    self._graph_sampler, self._graph_flag, self._static_batch = {}, {}, None
                if not getattr(staged, "persistent", False):
                    sampler = None                     # would be cloned: copy a freshly sampled batch in on every step instead
            batch = self._to_static(staged if sampler is not None else fetch())
            if sampler is not None:
                assert self._aliases_static(staged), "captured sampling must write the tensors the captured step reads"
            pre = None
            if sampler is not None:
                def pre():
                    # the critic phase's re-pack rides on the sampling launch (it reads nothing that launch writes)
                    if self._fused is not None:
                        self._fused.attach_entry(*self._entry_shape())
                    try:
                        sampler.launch_sample(self.batch_size)
                    except Exception:
                        hip.encoder_pack_drop_pending()
                        self.encoder.invalidate_packed()
                        raise
                    hip.encoder_pack_flush_pending()       # no-op when the sampling launch took the job
            torch.cuda.synchronize()
            quiesce_before_capture()               # RCCL's watchdog must have no eager work left to poll while this thread captures
            self._graph_sampler[key] = sampler
            captured = None
            if not exchanging:
                captured = self._capture_whole(batch, do_actor, polyak, pre)
            elif capture_exchange():
                # data-parallel, RCCL: the all-reduces are nodes of the step's graph (forked onto RCCL's stream by the process
                # group, joined before each optimizer pass) -- one graph launch per step, no host work between the segments
                host_state = self._host_step_state()
                try:
                    captured = self._capture_whole(batch, do_actor, polyak, pre, exchanging=True)
                except RuntimeError as err:
                    # ONLY a stack that refuses collectives under stream capture (every rank fails alike) falls back to segments;
                    # any other error of the step body (a kernel argument check, out of memory, an assertion) is a bug and is raised
                    if not _is_capture_error(err):
                        raise
                    # ProcessGroupNCCL words a STICKY device fault (an illegal instruction, a memory fault) the same way
                    # ("unhandled cuda error"): the fallback is only for a healthy context that refused the capture
                    try:
                        torch.cuda.synchronize()
                    except RuntimeError as dead:
                        raise RuntimeError(f"the device is in error after the failed capture ({str(dead).splitlines()[0]}): not a capture "
                                           "refusal, no fallback") from err
                    import warnings
                    warnings.warn(f"capturing the gradient exchange failed ({str(err).splitlines()[0]}); cutting the step into segments instead")
                    os.environ["PCRL_CAPTURE_EXCHANGE"] = "0"
                    self._host_step_state(restore=host_state)    # the aborted capture ran the step body's host side once
            if captured is None:
                captured = self._capture_segments(batch, do_actor, polyak, pre)
                self._graphs[key] = captured
                self._refresh_fast()
                segments, names, out = captured       # capturing a segmented step also executed it
                return self._finish(dict(zip(names, out.unbind(0))), updates)
            self._graphs[key] = captured
            self._refresh_fast()
        elif self._graph_sampler.get(key) is not None:
            fetch(launch=False)                     # the graph holds the sampling launch; the staging tensors are its inputs
        else:
            self._to_static(fetch())
        segments, names, out = self._graphs[key]
        flag = self._graph_flag.get(key)
        if flag is not None:
            flag[0][:flag[1]] = 0xFFFFFFFF          # sentinel in every slot of the pinned metrics mirror (the step's last kernel fills them)
        ex = Exchange(enabled=exchanging)
        for graph, (kind, pieces) in segments:
            graph.replay()
            for t in pieces:
                ex.start(t)
            if kind == "finish":
                ex.finish()
        self._invalidate_packed_after_replay()
        if flag is not None:
            return self._finish(dict.fromkeys(names), updates, host_values=self._await_flag(*flag))
        if out.device.type == "cpu":        # pinned host copy made by the graph's last node: wait for the graph, read it
            stream = self.__dict__.get("_sync_stream")
            if stream is None or stream.cuda_stream != hip.raw_stream():
                stream = self.__dict__["_sync_stream"] = torch.cuda.current_stream()
            stream.synchronize()
            return self._finish(dict.fromkeys(names), updates, host_values=out.tolist())
        return self._finish(dict(zip(names, out.unbind(0))), updates)

    def _host_step_state(self, restore=None):
        """Host-side state one pass through the step body advances (the jitter augmentations' call counts, the fused step's
        open-branch flag): snapshot, or put a snapshot back after a capture that was aborted half-way."""
        jitters = [t for t in (getattr(getattr(self, "obs_aug", None), "transforms", None) or []) if hasattr(t, "calls")]
        if restore is None:
            return dict(calls=[t.calls for t in jitters], slots=[getattr(t, "_slot", 0) for t in jitters])
        for t, c, sl in zip(jitters, restore["calls"], restore["slots"]):
            t.calls, t._slot = c, sl
            if hasattr(t, "_predrawn"):
                t._predrawn = None          # matrices drawn ahead inside the aborted capture belong to a graph that is thrown away
        if self._fused is not None:
            self._fused._forked = False
            self._fused._entry_cols = None
        # the aborted pass recorded its re-pack (and the job it may have handed to the sampling launch) into a graph that is thrown away
        hip.encoder_pack_drop_pending()
        for enc in {id(e): e for owners in self._packed_owners.values() for e in owners}.values():
            enc.invalidate_packed()

    def _invalidate_packed_after_replay(self):
        """A replayed step updated the encoder weights through raw pointers (no autograd version bump) and re-packs only
        inside the graph: eager users of the encoder (module-tree forward, the acting warm-up, DrQ's inference augmentation)
        must not trust the packed MFMA image they cached."""
        for owners in self._packed_owners.values():
            for enc in owners:
                enc._packed_key = None

    @staticmethod
    def _await_flag(view, n):
        """Spin until the step's last kernel has stored all n metrics into the pinned mirror (slots pre-filled with the sentinel
        0xFFFFFFFF; `view` is the mirror as uint32): no stream synchronisation, no copy node.  If the stream drains without the
        slots changing, the launch failed and the error is raised from the synchronisation."""
        slots, spins, t0 = view[:n], 0, None
        while (slots == 0xFFFFFFFF).any():
            spins += 1
            if spins & 0xFFFF == 0:
                if torch.cuda.current_stream().query():
                    torch.cuda.synchronize()
                    if (slots == 0xFFFFFFFF).any():
                        raise RuntimeError("update step finished without publishing its metrics")
                import time
                t0 = t0 or time.monotonic()
                if time.monotonic() - t0 > float(os.environ.get("PCRL_STEP_TIMEOUT_S", "120")):
                    # a hung kernel or a collective whose peer is gone: stop spinning and say so (no synchronisation here: it would
                    # block on the same wedged stream)
                    raise RuntimeError("update step did not publish its metrics within PCRL_STEP_TIMEOUT_S "
                                       f"({os.environ.get('PCRL_STEP_TIMEOUT_S', '120')} s): a kernel hangs or a peer of a collective is gone")
        return slots.view(np.float32).tolist()

    def _capture_whole(self, batch, do_actor, polyak, pre=None, exchanging=False):
        """exchanging: the step's all-reduces (FusedStep.run -> Exchange) are captured with it.  torch's ProcessGroupNCCL
        launches a collective on its own stream behind an event of the current one and `wait()` joins it back: under capture
        that is a forked branch of the graph, so the Q-head range still travels under the encoder backward."""
        graph = torch.cuda.CUDAGraph()
        host = None
        pinned = torch.empty(16, dtype=torch.float32, pin_memory=True)     # allocated outside the capture
        key = (do_actor, polyak, exchanging)
        self._graph_flag.pop(key, None)
        # thread_local: the process group's watchdog thread may touch the HIP runtime while this thread captures
        with _no_gc(), torch.cuda.graph(graph, **(dict(capture_error_mode="thread_local") if exchanging else {})):
            if pre is not None:
                pre()                       # the replay's sampling launch: first node of the step
            stats = self._step_body(batch, do_actor, polyak)
            names = list(stats.keys())
            packed = getattr(stats, "packed", None)
            mirror = getattr(stats, "host", None)
            out = packed if packed is not None else torch.stack([stats[k].reshape(()).float() for k in names])
            if mirror is not None:          # the step's last kernel stored the metrics to pinned host memory itself
                self._graph_flag[key] = (mirror.numpy().view(np.uint32), len(names))
            elif packed is not None:        # the metrics land in pinned host memory as the graph's last node
                host = pinned[:len(names)]
                host.copy_(out, non_blocking=True)
        return [(graph, ("finish", []))], names, (host if host is not None else out)

    def _capture_segments(self, batch, do_actor, polyak, pre=None):
        """Data-parallel: one hipGraph per stretch between gradient exchanges; the RCCL all-reduces stay
        eager between the graph launches (no collective is ever captured).  Capturing records without
        executing, so each segment is replayed right after its capture to carry the step forward."""
        pool = torch.cuda.graph_pool_handle()
        scale = 1.0 / world_size()
        segments, names, out, gen = [], None, None, None
        ex = Exchange()
        kind = None
        while names is None:
            ex.finish()
            quiesce_before_capture()               # the previous segment's eager all-reduces are still on the watchdog's list
            graph = torch.cuda.CUDAGraph()
            exchange = ("finish", [])
            # thread_local: the RCCL watchdog thread may touch the HIP runtime while this thread captures
            with _no_gc(), torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
                try:
                    if gen is None:     # sampling and batch preparation (DrQ: augmentation draws) belong to the first segment
                        if pre is not None:
                            pre()
                        args, kwargs = self._fused_args(batch, do_actor, polyak)
                        gen = self._fused.steps(*args, **kwargs)
                        exchange = next(gen)
                    else:
                        exchange = gen.send(scale if kind == "finish" else None)
                except StopIteration as done:
                    stats = done.value
                    names = list(stats.keys())
                    packed = getattr(stats, "packed", None)        # the fused step already gathered the metrics into one tensor
                    out = packed if packed is not None else torch.stack([stats[k].reshape(()).float() for k in names])
                    if getattr(stats, "host", None) is not None:
                        self._graph_flag[(do_actor, polyak, True)] = (stats.host.numpy().view(np.uint32), len(names))
            kind, pieces = exchange
            graph.replay()
            for t in pieces:
                ex.start(t)
            if kind == "finish":
                ex.finish()
            segments.append((graph, (kind, list(pieces))))
        return segments, names, out

    _process_sampled_obs = True

    def _fetcher(self, memory):
        """`fetch(launch=True)` -> the sampled batch on the device (sac.py:104-107); launch=False asks a graph-sampling replay
        for its staging batch without launching (the captured step holds the launch)."""
        def fetch(launch=True):
            sample = memory.sample(self.batch_size) if launch else memory.sample(self.batch_size, launch=False)
            sampled_batch = sample.to_torch(device=self.device, non_blocking=True)
            if self._process_sampled_obs:          # sac.py:106 (drq.py's update_parameters does not call it)
                sampled_batch = self.process_obs(sampled_batch)
            if self.use_episode_dones:
                sampled_batch["dones"] = sampled_batch["episode_dones"]
            return sampled_batch
        return fetch

    def update_parameters(self, memory, updates):
        if self._flat is None:
            self._prepare()
        fast = self.__dict__.get("_fast")
        if fast is not None:
            ret = self._replay_fast(fast, memory, updates)
            if ret is not None:
                return ret
        return self._run_step(self._fetcher(memory), updates, sampler=memory)

    # Steady state of a graph-replayed agent fed by a DeviceReplay: everything `_run_step` decides per call has been decided
    # when the variants were captured, so a call is: count the sample, clear the flag, launch the graph(s), spin on the flag,
    # build the dict.  Anything unusual (a variant not captured yet, another replay object, a changed learning rate, every
    # 128th call for the full hyper-parameter check) falls back to `_run_step`.
    def _refresh_fast(self):
        self._fast = None
        if not (getattr(self, "_use_graphs", False) and self._fused is not None and self._target_flat is not None
                and all(isinstance(o, HipAdam) for o in (self.critic_optim, self.actor_optim, self.alpha_optim))):
            return
        entries = {}
        for key, (segments, names, out) in self._graphs.items():
            flag, sampler = self._graph_flag.get(key), self._graph_sampler.get(key)
            if flag is None or sampler is None:
                return
            entries[key] = (segments, tuple(names), flag[0], flag[1], sampler)
        if entries:
            opts = (self.critic_optim, self.actor_optim, self.alpha_optim)
            self._fast = dict(entries=entries, lrs=[(o.param_groups[0], o.param_groups[0]["lr"]) for o in opts], calls=0)

    def _replay_fast(self, fast, memory, updates):
        fast["calls"] += 1
        if fast["calls"] & 127 == 0 or not self._use_graphs:
            return None
        exchanging = self._be_data_parallel and exchange_active()
        entry = fast["entries"].get((updates % self.actor_update_interval == 0, updates % self.target_update_interval == 0, exchanging))
        if entry is None or entry[4] is not memory:
            return None
        for group, lr in fast["lrs"]:
            if group["lr"] != lr:
                return None
        segments, names, view, n, _ = entry
        memory.sample(self.batch_size, launch=False)
        view[:n] = 0xFFFFFFFF
        if exchanging:
            ex = Exchange(enabled=True)
            for graph, (kind, pieces) in segments:
                graph.replay()
                for t in pieces:
                    ex.start(t)
                if kind == "finish":
                    ex.finish()
        else:
            for graph, _ in segments:
                graph.replay()
        self._invalidate_packed_after_replay()
        return self._ret_from_values(names, self._await_flag(view, n))
