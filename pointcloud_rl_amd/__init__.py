"""pointcloud_rl_amd -- MI355X (gfx950) implementation of the point-cloud actor-critic hot path of
lz1oceani/pointcloud_rl: PointNet encoder + SAC/DrQ update step behind the reference's registry API."""
__version__ = "0.1.0"


def bind_reference(force=True):
    """Override the reference's `pyrl` registries with this package's classes (Level 1).  See `bind.py`."""
    from .bind import bind_reference as _bind
    return _bind(force=force)
