"""Level-1 binding: make the reference's own `pyrl` registries dispatch to the MI355X classes.

`bind_reference()` is what a maintainer calls at the end of `pyrl/methods/__init__.py`.  It overrides the entries of the
reference's registries (`register_module(..., force=True)`, pyrl/utils/meta/registry.py:41-96) with this package's classes.

The agents need one more thing than a name: the unchanged driver asserts `isinstance(agent, BaseAgent)` against the
REFERENCE's base class (pyrl/apis/run_rl.py:308; the class is pyrl/utils/torch/module_utils.py:112).  So the agent classes that
go into the reference's `MFRL` registry are derived here, at bind time, from both: this package's agent first in the method
resolution order -- every method the drivers call (`forward`, `update_parameters`, `to_ddp`, `to_normal`, `recover_ddp`,
`set_mode`, `process_obs`, ...) resolves to this package's -- and the reference's `BaseAgent` behind it, whose `__init__` runs
inside the cooperative `super().__init__()` chain and whose PPO-only helpers (`compute_gae`, `run_actor`, ...) stay reachable.
Nothing of the reference is imported unless `pyrl` is importable; without it `bind_reference()` raises ImportError and the
maintainer's `try / except ImportError` keeps the reference as shipped.
"""
import sys

_BOUND = {}


def _derive_agent(cls, ref_base):
    """`cls` (a pointcloud_rl_amd agent) re-based so that it also IS a reference BaseAgent; cached per (class, base)."""
    key = (cls, ref_base)
    if key not in _BOUND:
        if issubclass(cls, ref_base):
            _BOUND[key] = cls
        else:
            derived = type(cls.__name__, (cls, ref_base), {"__module__": __name__, "__doc__": cls.__doc__, "__qualname__": cls.__name__})
            setattr(sys.modules[__name__], cls.__name__, derived)     # importable by qualified name (pickle, repr)
            _BOUND[key] = derived
    return _BOUND[key]


def bind_reference(force=True):
    """Register this package's classes in the reference's registries under the reference's names.  Returns
    {registry name: [registered names]}.  Call after the reference's own registrations (end of pyrl/methods/__init__.py)."""
    from pyrl.methods.builder import MFRL                      # pyrl/methods/builder.py:4
    from pyrl.networks.builder import APPLICATION, NETWORK, REGRESSION   # pyrl/networks/builder.py:6
    from pyrl.utils.augmentations.builder import AUGMENTATIONS  # pyrl/utils/augmentations/builder.py:7
    from pyrl.utils.torch import BaseAgent as RefBaseAgent      # pyrl/utils/torch/module_utils.py:112

    from . import augmentations as amd_aug
    from . import methods as amd_methods
    from . import networks as amd_nets

    done = {}
    for reg, src in ((MFRL, amd_methods.MFRL), (NETWORK, amd_nets.NETWORK), (REGRESSION, amd_nets.REGRESSION),
                     (APPLICATION, amd_nets.APPLICATION), (AUGMENTATIONS, amd_aug.AUGMENTATIONS)):
        names = []
        for name, cls in src.module_dict.items():
            if reg is MFRL:
                cls = _derive_agent(cls, RefBaseAgent)
            reg.register_module(name=name, force=force, module=cls)
            names.append(name)
        done[reg.name] = names
    return done
