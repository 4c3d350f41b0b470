from .registry import ConfigDict, Registry, build_from_cfg  # noqa: F401
