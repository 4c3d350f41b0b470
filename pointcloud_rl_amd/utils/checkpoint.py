"""Checkpoints in the reference's format.

Behaviour of pyrl/utils/torch/checkpoint_utils.py: `get_state_dict` (215-237) is `nn.Module.state_dict` plus the
`state_dict()` of every optimizer found as a module attribute, stored under its attribute path ("actor_optim",
"critic_optim", "alpha_optim" for SAC/DrQ); `save_checkpoint` (240-269) writes `{"meta", "state_dict"[, "optimizer"]}`
with tensors on the CPU; `load_state_dict` (25-96) is non-strict by default, adapts a weight whose shape differs in ONE
dimension (different input channels), loads the optimizers it finds and reports missing / unexpected keys;
`load_checkpoint` (148-179) accepts a bare state_dict or a checkpoint dict, an optional regex `keys_map`, and strips a
DistributedDataParallel "module." prefix.  Parameter and optimizer names of this package's agents are the reference's, so
files written by either side load into the other (the fused optimizer speaks torch.optim.Adam's state_dict format).
Only local files: the reference's URL / torchvision model-zoo loaders are not part of the hot path.
"""
import os
import re
from collections import OrderedDict

import numpy as np
import torch


def _is_optimizer(obj):
    """The reference's filter (checkpoint_utils.py:59, 227): torch optimizers only.  The fused HipAdam is one."""
    return isinstance(obj, torch.optim.Optimizer)


def get_state_dict(module, destination=None, prefix="", keep_vars=False):
    """Parameters, buffers and -- under their attribute names -- optimizer states of `module` and its children."""
    if destination is None:
        destination = OrderedDict()
        destination._metadata = OrderedDict()
    destination._metadata[prefix[:-1]] = dict(version=module._version)
    for name, param in module._parameters.items():
        if param is not None:
            destination[prefix + name] = param if keep_vars else param.detach()
    for name, buf in module._buffers.items():
        if buf is not None:
            destination[prefix + name] = buf if keep_vars else buf.detach()
    seen = set()
    for name, child in module.__dict__.items():
        if child is not None and _is_optimizer(child) and id(child) not in seen:
            seen.add(id(child))
            destination[prefix + name] = child.state_dict()
    for name, child in module._modules.items():
        if child is not None:
            get_state_dict(child, destination, prefix + name + ".", keep_vars=keep_vars)
    return destination


def _to_cpu(x):
    if torch.is_tensor(x):
        return x.detach().cpu()
    if isinstance(x, dict):
        return type(x)((k, _to_cpu(v)) for k, v in x.items()) if not isinstance(x, OrderedDict) else OrderedDict((k, _to_cpu(v)) for k, v in x.items())
    if isinstance(x, (list, tuple)):
        return type(x)(_to_cpu(v) for v in x)
    return x


def load_state_dict(module, state_dict, strict=False, logger=None):
    log = logger.warning if logger is not None else print
    metadata = getattr(state_dict, "_metadata", None)
    state_dict = OrderedDict(state_dict)
    for name, parameter in module.named_parameters():
        src = state_dict.get(name)
        if torch.is_tensor(src) and src.shape != parameter.shape and src.ndim == parameter.ndim:
            diff = np.array(src.shape) != np.array(parameter.shape)
            if diff.sum() == 1:                      # e.g. a different number of input channels: copy the common part
                axis = int(np.nonzero(diff)[0][0])
                log(f"We adapt weight with shape {list(src.shape)} to shape {list(parameter.shape)}.")
                merged = parameter.detach().clone().transpose(0, axis)
                n = min(src.shape[axis], parameter.shape[axis])
                merged[:n] = src.to(merged).transpose(0, axis)[:n]
                state_dict[name] = merged.transpose(0, axis).contiguous()
    missing, unexpected, errors, visited = [], [], [], set()

    def load(mod, prefix=""):
        if id(mod) in visited:
            return
        visited.add(id(mod))
        seen = set()
        for name, child in mod.__dict__.items():
            if child is not None and _is_optimizer(child) and id(child) not in seen:
                seen.add(id(child))
                key = prefix + name
                if key in state_dict:
                    try:
                        child.load_state_dict(state_dict.pop(key))
                    except Exception as e:           # the reference keeps going when an optimizer does not fit
                        log(f"We cannot load optimizer {key}!")
                        log(f"Exception from pytorch is {e}!")
                else:
                    log(f"missing keys in source state_dict for optimizer {key}")
        local_metadata = {} if metadata is None else metadata.get(prefix[:-1], {})
        mod._load_from_state_dict(state_dict, prefix, local_metadata, True, missing, unexpected, errors)
        for name, child in mod._modules.items():
            if child is not None:
                load(child, prefix + name + ".")

    load(module)
    missing = [k for k in missing if "num_batches_tracked" not in k]
    if unexpected:
        errors.append(f'unexpected key in source state_dict: {", ".join(unexpected)}\n')
    if missing:
        errors.append(f'missing keys in source state_dict: {", ".join(missing)}\n')
    if errors:
        msg = "\n".join(["The model and loaded state dict do not match exactly\n"] + errors)
        if strict:
            raise RuntimeError(msg)
        log(msg)


def save_checkpoint(model, filename, optimizer=None, meta=None):
    if meta is None:
        meta = {}
    elif not isinstance(meta, dict):
        raise TypeError(f"meta must be a dict or None, but got {type(meta)}")
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    checkpoint = {"meta": meta, "state_dict": _to_cpu(get_state_dict(model))}
    if optimizer is not None and _is_optimizer(optimizer):
        checkpoint["optimizer"] = _to_cpu(optimizer.state_dict())
    elif isinstance(optimizer, dict):
        checkpoint["optimizer"] = {name: _to_cpu(o.state_dict()) for name, o in optimizer.items()}
    with open(filename, "wb") as f:
        torch.save(checkpoint, f)
        f.flush()


def _map_keys(state_dict, keys_map, log=None):
    out = OrderedDict()
    for key, value in state_dict.items():
        new_key = key
        for pattern, repl in keys_map.items():
            if re.match(pattern, key):
                new_key = None if repl is None else re.sub(pattern, repl, key)
                break
        if new_key is None or new_key == "None":
            if log is not None:
                log(f"Delete {key}!")
            continue
        out[new_key] = value
    return out


def load_checkpoint(model, filename, map_location=None, strict=False, keys_map=None, logger=None):
    filename = str(filename)
    if filename.startswith(("http://", "https://", "torchvision://")):
        raise NotImplementedError("only local checkpoint files (there is no model zoo on this path)")
    if not os.path.isfile(filename):
        raise IOError(f"{filename} is not a checkpoint file")
    checkpoint = torch.load(filename, map_location=map_location, weights_only=False)
    if not isinstance(checkpoint, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")
    state_dict = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    if keys_map is not None:
        state_dict = _map_keys(state_dict, keys_map, logger.info if logger is not None else None)
    if state_dict and next(iter(state_dict)).startswith("module."):
        state_dict = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())
    load_state_dict(model, state_dict, strict, logger)
    return checkpoint
