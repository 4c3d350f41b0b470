"""String -> class registries and config dicts, the boundary the hot path sits behind.

Mirrors the contract of the reference's pyrl/utils/meta/registry.py:4-136 (Registry,
build_from_cfg) and of ConfigDict (pyrl/utils/meta/config.py:21-34): classes register under their
name with a decorator, duplicate names raise unless force=True, `build_from_cfg` pops "type" and
calls cls(**cfg).
"""
import copy
import inspect


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted on the way in."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(e) for e in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def copy(self):
        return ConfigDict(self)

    def __deepcopy__(self, memo):
        out = ConfigDict()
        for k, v in self.items():
            out[copy.deepcopy(k, memo)] = copy.deepcopy(v, memo)
        return out


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f"{type(self).__name__}(name={self._name}, items={list(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register(self, cls, name=None, force=False):
        if not (inspect.isclass(cls) or inspect.isfunction(cls)):
            raise TypeError(f"module must be a class or a function, but got {type(cls)}")
        name = name or cls.__name__
        if not force and name in self._module_dict:
            raise KeyError(f"{name} is already registered in {self.name}")
        self._module_dict[name] = cls

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f"force must be a boolean, but got {type(force)}")
        if not (name is None or isinstance(name, str)):
            raise TypeError(f"name must be a str, but got {type(name)}")
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco


def build_from_cfg(cfg, registry, default_args=None):
    if cfg is None:
        return None
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if "type" not in cfg and not (default_args and "type" in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f"registry must be a Registry object, but got {type(registry)}")
    if not (isinstance(default_args, dict) or default_args is None):
        raise TypeError(f"default_args must be a dict or None, but got {type(default_args)}")
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        cls = registry.get(obj_type)
        if cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    return cls(**args)
