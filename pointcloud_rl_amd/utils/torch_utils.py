"""Module / agent base classes and parameter utilities of the hot path's host side.

Same names and call contracts as the reference's pyrl/utils/torch/{module_utils,ops,optimizer_utils}.py
so that run_rl.py / train_rl.py style drivers can use the agent unchanged.
"""
import copy
import inspect
import re
from contextlib import contextmanager

import torch
import torch.nn as nn

from .registry import Registry, build_from_cfg


def regex_match(string, pattern):
    return re.match(pattern, string) is not None


class ExtendedModule(nn.Module):
    """nn.Module + the helpers the reference's drivers call (module_utils.py:11-68)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._in_test = False
        self.is_recurrent = False

    def set_mode(self, mode="train"):
        self._in_test = mode == "test"
        for m in self.children():
            if isinstance(m, ExtendedModule):
                m.set_mode(mode)
        return self

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def trainable_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    @property
    def num_trainable_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    @property
    def size_trainable_parameters(self):
        return sum(p.numel() * p.element_size() for p in self.parameters() if p.requires_grad)

    @property
    @torch.no_grad()
    def grad_norm(self):
        """2-norm of the per-tensor gradient 2-norms (module_utils.py:40-45)."""
        grads = [torch.norm(p.grad.detach(), 2) for p in self.parameters() if p.requires_grad and p.grad is not None]
        return torch.norm(torch.stack(grads), 2).item() if grads else 0.0

    def pop_attr(self, name):
        if hasattr(self, name):
            ret = getattr(self, name)
            setattr(self, name, None)
            return ret
        return None

    @contextmanager
    def no_sync(self):
        yield


class ExtendedModuleList(nn.ModuleList, ExtendedModule):
    pass


class ExtendedSequential(nn.Sequential, ExtendedModule):
    pass


@torch.no_grad()
def soft_update(target, source, tau):
    """Polyak update theta' <- (1 - tau) theta' + tau theta (ops.py:59-90).  `tau` is a number or a
    dict {"default": t, regex: t, ...} matched against the source's parameter names; parameters that
    are the same object in both networks (a shared visual backbone) are left alone."""
    if isinstance(target, nn.Parameter):
        if target is not source:
            target.data.copy_(target.data * (1.0 - tau) + source.data * tau)
        return
    if isinstance(tau, (int, float)):
        for tp, sp in zip(target.parameters(), source.parameters()):
            soft_update(tp, sp, tau)
        return
    assert isinstance(tau, dict), f"tau should be a number or a dict, but the type of tau is {type(tau)}."
    assert "default" in tau, f"The dict needs key default! You dict contains keys: {list(tau.keys())}"
    tau = dict(tau)
    default = tau.pop("default")
    tparams = dict(target.named_parameters())
    for name, sp in source.named_parameters():
        t = default
        for pattern, value in tau.items():
            if regex_match(name, pattern):
                t = value
                break
        soft_update(tparams[name], sp, t)


@torch.no_grad()
def hard_update(target, source):
    if isinstance(target, nn.Parameter):
        if target is not source:
            target.data.copy_(source.data)
        return
    for tp, sp in zip(target.parameters(), source.parameters()):
        hard_update(tp, sp)


def disable_gradients(network, exclude=()):
    for p in network.parameters():
        if id(p) not in exclude:
            p.requires_grad = False


OPTIMIZERS = Registry("optimizer")
for _name in dir(torch.optim):
    _cls = getattr(torch.optim, _name)
    if not _name.startswith("__") and inspect.isclass(_cls) and issubclass(_cls, torch.optim.Optimizer):
        OPTIMIZERS.register_module(module=_cls)


def select_optimizer_params(model, param_cfg=None):
    """Names/parameters an optimizer built by `build_optimizer` owns: trainable, deduplicated, and
    not matched by a `param_cfg` pattern whose value is None (optimizer_utils.py:43-57)."""
    out, seen = [], set()
    for name, p in model.named_parameters():
        if id(p) in seen or not p.requires_grad:
            continue
        seen.add(id(p))
        excluded = False
        for pattern, cfg in (param_cfg or {}).items():
            if regex_match(name, pattern):
                excluded = cfg is None
                break
        if not excluded:
            out.append((name, p))
    return out


def build_optimizer(model, cfg):
    """One parameter group per tensor, as the reference builds them (optimizer_utils.py:31-64)."""
    cfg = copy.deepcopy(dict(cfg))
    if cfg.pop("constructor", "default") != "default":
        raise NotImplementedError
    param_cfg = cfg.pop("param_cfg", None)
    if hasattr(model, "named_parameters"):
        params = [{"params": p} for _, p in select_optimizer_params(model, param_cfg)]
    else:
        params = [model]
    cfg["params"] = params
    return build_from_cfg(cfg, OPTIMIZERS)


def to_torch(x, device=None, non_blocking=False):
    """Nested dict/list of numpy arrays or tensors -> tensors on `device` (GDict.to_torch, dict_array.py:308-318)."""
    import numpy as np
    if isinstance(x, dict):
        out = {k: to_torch(v, device, non_blocking) for k, v in x.items()}
        if type(x) is not dict:            # an observation that carries a pending augmentation / virtual repeat keeps them
            try:
                kept = type(x)(out)
                kept.__dict__.update(getattr(x, "__dict__", {}))
                return kept
            except Exception:
                return out
        return out
    if isinstance(x, (list, tuple)):
        return type(x)(to_torch(v, device, non_blocking) for v in x)
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    if torch.is_tensor(x):
        return x.to(device=device, non_blocking=non_blocking) if device is not None else x
    return x


class BaseAgent(ExtendedModule):
    """Acting path + data-parallel switches (module_utils.py:112-349, the parts SAC/DrQ use)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._device_ids = None
        self._be_data_parallel = False
        self.obs_processor = None
        self.obs_rms = None
        self.rew_rms = None
        self.batch_size = None
        self.recurrent_horizon = -1

    def reset(self, *args, **kwargs):
        pass

    @property
    def has_obs_process(self):
        return self.obs_rms is not None or self.obs_processor is not None

    @torch.no_grad()
    def process_obs(self, data, **kwargs):
        if self.obs_processor is not None:
            for key in ("obs", "next_obs"):
                if key in data:
                    data[key] = self.obs_processor({"obs": data[key]})["obs"]
        return data

    @torch.no_grad()
    def forward(self, obs, **kwargs):
        obs = to_torch(obs, device=self.device, non_blocking=True)
        kwargs = dict(kwargs)
        if "prev_actions" in kwargs:
            kwargs["prev_actions"] = to_torch(kwargs["prev_actions"], device=self.device, non_blocking=True)
        if self.obs_processor is not None:
            obs = self.obs_processor({"obs": obs})["obs"]
        return self.actor(obs, **kwargs)

    # One process per GPU; gradients are exchanged with explicit RCCL all-reduces on flat buffers
    # inside update_parameters, so these switches only record the state the drivers toggle
    # (run_rl.py:329, train_rl.py:396-405).
    def to_ddp(self, device_ids=None):
        """The reference wraps actor / critic / target critic in DistributedDataParallel here, whose constructor
        broadcasts rank 0's parameters and buffers (module_utils.py:322-343) -- drivers seed torch with seed + rank
        before building the agent (run_rl.py:263), so without that broadcast the replicas would start from different
        weights and, with only gradients averaged, never meet.  Same effect here: every parameter of the agent
        (actor, critic, target critic, log_alpha; in place, so flat-buffer views stay valid) is broadcast from rank 0."""
        from .dist import broadcast_parameters_, exchange_active
        self._device_ids = device_ids
        if exchange_active():
            broadcast_parameters_(self)
            for enc in (m for m in self.modules() if hasattr(m, "invalidate_packed")):
                enc.invalidate_packed()
        self.recover_ddp()

    def to_normal(self):
        self._be_data_parallel = False

    def recover_ddp(self):
        if self._device_ids is not None:
            self._be_data_parallel = True

    def is_data_parallel(self):
        return self._be_data_parallel

    def no_sync(self, mode="actor"):
        return getattr(self, mode).no_sync()
