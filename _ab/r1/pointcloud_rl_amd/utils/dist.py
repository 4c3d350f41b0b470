"""Data-parallel gradient exchange: one process per GPU, RCCL (`backend="nccl"` on ROCm) over xGMI.

The reference wraps actor / critic / target critic in DistributedDataParallel, which all-reduces
bucketed gradients inside every backward (pyrl/utils/torch/module_utils.py:322-343).
Here every optimizer owns ONE flat gradient buffer, so the exchange is a single sum all-reduce per
backward; the 1/world factor is folded into the fused optimizer kernel's `grad_scale`.
"""
import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def allreduce_sum_(flat_grad, enabled=True):
    """In-place SUM all-reduce of a flat gradient buffer.  Returns the scale (1/world) the caller must
    apply to obtain the mean, 1.0 when nothing was exchanged."""
    w = world_size()
    if not enabled or w == 1:
        return 1.0
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / w


def broadcast_parameters_(module, src=0):
    """Make every rank start from rank `src`'s weights (what DDP's constructor does implicitly)."""
    if world_size() == 1:
        return
    for p in module.parameters():
        dist.broadcast(p.data, src)
    for b in module.buffers():
        dist.broadcast(b.data, src)


def shard_slice(global_batch, rank, world):
    """Contiguous shard of a global batch owned by `rank` (strong scaling: B/world samples each)."""
    assert global_batch % world == 0, "batch must divide over the ranks"
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)
