"""Dev tooling (build container only): make the reference importable without its third-party deps.

The reference (/root/reference, read-only) imports addict, yapf, sorcery, torchvision, torchviz
and gym at module import time; none is installed here and there is no network.  This file
installs minimal stand-ins into sys.modules -- just enough behaviour for `import pyrl.networks,
pyrl.methods` and for building SAC / DrQ agents from the shipped configs.  Nothing here is
reference code and nothing here ships to the GPU box's test run (tests read the .npz fixtures).
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []

    def _getattr(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Anything

    m.__getattr__ = _getattr
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    return m


class AttrDict(dict):
    """Attribute-access recursive dict with the subset of addict.Dict behaviour ConfigDict relies on."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for arg in args:
            if not arg:
                continue
            items = arg.items() if isinstance(arg, dict) else iter(arg)
            for k, v in items:
                self[k] = self._hook(v)
        for k, v in kwargs.items():
            self[k] = self._hook(v)

    @classmethod
    def _hook(cls, item):
        if isinstance(item, dict):
            return cls(item)
        if isinstance(item, (list, tuple)):
            return type(item)(cls._hook(e) for e in item)
        return item

    def __getattr__(self, item):
        return self.__getitem__(item)

    def __setattr__(self, name, value):
        self[name] = value

    def __missing__(self, name):
        raise KeyError(name)

    def copy(self):
        return type(self)(self)

    def __deepcopy__(self, memo):
        import copy
        out = type(self)()
        for k, v in self.items():
            out[copy.deepcopy(k, memo)] = copy.deepcopy(v, memo)
        return out

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, AttrDict) else v) for k, v in self.items()}


class Box:
    """gym.spaces.Box stand-in: isinstance target with .low/.high/.is_bounded()."""

    def __init__(self, low, high, dtype=np.float32):
        self.low, self.high, self.dtype = np.asarray(low, dtype), np.asarray(high, dtype), dtype
        self.shape = self.low.shape

    def is_bounded(self):
        return True

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


class Discrete:
    def __init__(self, n=2):
        self.n = n


def install():
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _stub("addict", Dict=AttrDict)
    for name in ("yapf", "yapf.yapflib", "yapf.yapflib.yapf_api", "sorcery", "torchvision", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchvision.transforms.functional_tensor", "torchviz",
                 "gym", "gym.spaces"):
        _stub(name)
    sys.modules["gym.spaces"].Box = Box
    sys.modules["gym.spaces"].Discrete = Discrete
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]


def build_reference_agent(cfg_file, obs_shape, action_dim, cfg_overrides=None, seed=0):
    """Config.fromfile -> placeholder substitution -> build_agent, as run_rl.py does (run_rl.py:103-111, 298)."""
    import torch
    import pyrl.networks  # noqa: F401
    import pyrl.methods  # noqa: F401
    from pyrl.methods.builder import build_agent
    from pyrl.networks.utils import get_kwargs_from_shape, replace_placeholder_with_args
    from pyrl.utils.meta import Config

    cfg = Config.fromfile(cfg_file)
    if cfg_overrides:
        cfg.merge_from_dict(cfg_overrides)
    space = Box(-np.ones(action_dim), np.ones(action_dim))
    cfg.agent_cfg["env_params"] = dict(obs_shape=obs_shape, action_shape=action_dim, action_space=space, is_discrete=False, message="")
    cfg = replace_placeholder_with_args(cfg, **get_kwargs_from_shape(obs_shape, action_dim))
    torch.manual_seed(seed)
    np.random.seed(seed)
    return build_agent(cfg.agent_cfg), cfg
