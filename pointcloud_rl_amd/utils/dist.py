"""Data-parallel gradient exchange: one process per GPU, RCCL (`backend="nccl"` on ROCm) over xGMI.

The reference wraps actor / critic / target critic in DistributedDataParallel, which all-reduces
bucketed gradients inside every backward (pyrl/utils/torch/module_utils.py:322-343).
Here every optimizer owns ONE flat gradient buffer, so the exchange is a single sum all-reduce per
backward; the 1/world factor is folded into the fused optimizer kernel's `grad_scale`.
"""
import os

import torch.distributed as dist


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def exchange_active():
    """True when gradient exchanges have to run: more than one rank -- or one rank with PCRL_EXCHANGE_SINGLE_RANK=1, which
    drives the whole exchange path (segmented graphs, async RCCL all-reduces on the process group's stream, the waits)
    through a one-rank process group; a one-rank SUM leaves the buffer as it is, so the step must equal the plain one bit
    for bit (tests/test_data_parallel_gpu.py runs RCCL itself this way on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PCRL_EXCHANGE_SINGLE_RANK", "0") == "1"


def capture_exchange():
    """True when the gradient all-reduces may be captured into the step's hipGraph: the process group runs on RCCL
    (backend "nccl": device-side collectives on a stream; gloo reduces on host threads and cannot be captured) and
    PCRL_CAPTURE_EXCHANGE is not "0" (which restores the segmented schedule: eager all-reduces between per-segment graphs)."""
    if os.environ.get("PCRL_CAPTURE_EXCHANGE", "1") == "0" or not exchange_active():
        return False
    if str(dist.get_backend()).lower() != "nccl":
        return False
    return _captured_allreduce_works()


_PROBE = None
_PROBE_NOTE = "not run"


def probe_verdict():
    """What the one-time probe of `capture_exchange` found, for logs / bench.py's line: "not run", "works", "failed: <why>", "skipped"."""
    return _PROBE_NOTE


def _captured_allreduce_works():
    """Asked once per process: ONE small all-reduce captured into a hipGraph, replayed twice and checked against the expected sum, the
    verdict agreed between the ranks.  A stack whose RCCL refuses stream capture or breaks under graph replay is found here --
    identically on every rank and before anything of the step is captured -- and the run uses the segmented schedule (eager
    all-reduces between per-segment graphs) instead of ending with exit code 75 at its first replayed step.

    Two agreements, both eager MIN all-reduces that EVERY rank reaches on every path (a rank whose capture raised is no longer
    capturing): (1) did the capture succeed everywhere -- only then does any rank replay its graph, so a replayed 1 024-float SUM is
    never paired with a peer's eager collective; (2) did the replays give the expected sum everywhere.  A replay that HANGS cannot be
    recovered from in-process: it raises after PCRL_EXCHANGE_PROBE_TIMEOUT_S (30) and the rank ends non-zero at once (its peers'
    collectives then fail or time out; bench.py's launcher stops them when it sees the first rank die and starts fresh children for
    its one retry).  PCRL_EXCHANGE_PROBE=0 skips the probe (captured exchange assumed to work)."""
    global _PROBE, _PROBE_NOTE
    if _PROBE is not None:
        return _PROBE
    if os.environ.get("PCRL_EXCHANGE_PROBE", "1") == "0":
        _PROBE, _PROBE_NOTE = True, "skipped"
        return True
    import time
    import warnings
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())

    def agree(flag):
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        quiesce_before_capture()
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)
    ok, why, graph = True, "", None
    buf = torch.ones(1024, device=dev)
    try:
        dist.all_reduce(buf)                          # eager: the communicator's lazy initialisation must not be captured
        quiesce_before_capture()
        graph = torch.cuda.CUDAGraph()
        with warnings.catch_warnings():
            warnings.filterwarnings("ignore", message="The CUDA Graph is empty")      # (a one-rank all-reduce launches nothing)
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                dist.all_reduce(buf, async_op=True).wait()
    except RuntimeError as err:
        ok, why = False, "capture: " + str(err).splitlines()[0]
        try:
            torch.cuda.synchronize()
        except RuntimeError:
            pass
    everyone_captured = agree(ok)
    if everyone_captured:
        try:
            want = float(dist.get_world_size())
            for _ in range(2):
                buf.fill_(1.0)
                graph.replay()
            done = torch.cuda.Event()
            done.record()
            t0, limit = time.monotonic(), float(os.environ.get("PCRL_EXCHANGE_PROBE_TIMEOUT_S", "30"))
            while not done.query():
                if time.monotonic() - t0 > limit:
                    raise TimeoutError(f"a replayed hipGraph holding one RCCL all-reduce did not finish within {limit:.0f} s")
                time.sleep(0.001)
            if not bool((buf == want).all()):
                ok, why = False, f"a replayed all-reduce of ones gave {float(buf[0])} on {int(want)} ranks"
        except TimeoutError as err:
            _PROBE_NOTE = f"failed: {err}"
            raise RuntimeError(f"captured exchange probe: {err} (hipGraph / RCCL: the stream is wedged)") from err
        except RuntimeError as err:
            ok, why = False, "replay: " + str(err).splitlines()[0]
            try:
                torch.cuda.synchronize()
            except RuntimeError:
                pass
    elif ok:
        why = "another rank could not capture the all-reduce"
    del graph
    _PROBE = agree(ok and everyone_captured)          # every rank takes the same schedule
    _PROBE_NOTE = "works" if _PROBE else "failed: " + (why or "another rank's replay failed")
    if not _PROBE:
        warnings.warn("RCCL all-reduce inside a replayed hipGraph does not work on this stack (" + (why or "another rank's probe failed")
                      + "): eager all-reduces between per-segment graphs instead")
        os.environ["PCRL_CAPTURE_EXCHANGE"] = "0"
    quiesce_before_capture()
    return _PROBE


def quiesce_before_capture():
    """Call before a stream capture in a process whose RCCL process group has issued eager collectives.  torch's
    ProcessGroupNCCL watchdog thread polls the completion events of eager work with hipEventQuery, and on ROCm an event query
    from another thread while this thread captures fails with hipErrorStreamCaptureUnsupported -- the watchdog rethrows and the
    process aborts (seen with eager all-reduces between per-segment captures).  Finished work leaves the watchdog's list on its
    next poll (every 100 ms): drain the device, then give it three polls."""
    if not (dist.is_available() and dist.is_initialized()) or str(dist.get_backend()).lower() != "nccl":
        return
    import time
    import torch
    torch.cuda.synchronize()
    time.sleep(0.35)


def allreduce_sum_(flat_grad, enabled=True):
    """In-place SUM all-reduce of a flat gradient buffer.  Returns the scale (1/world) the caller must
    apply to obtain the mean, 1.0 when nothing was exchanged."""
    if not enabled or not exchange_active():
        return 1.0
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / world_size()


class Exchange:
    """Gradient exchanges of one update step.  `start(t)` launches a SUM all-reduce of a piece of a flat gradient buffer that
    is already final while the rest of the backward is still running -- the collective runs on the process group's own
    stream (RCCL) / thread (gloo) -- and `finish()` makes the current stream wait for every piece before the optimizer
    reads the buffer.  The reference's DistributedDataParallel does the same with its gradient buckets
    (pyrl/utils/torch/module_utils.py:322-343); here the pieces are two per backward: the dense heads' range, final before
    the encoder backward starts, and the small encoder + feature-head range after it."""

    def __init__(self, enabled=True):
        self.enabled = enabled and exchange_active()
        self.pending = []
        self.scale = 1.0 / world_size() if self.enabled else 1.0

    def start(self, flat_piece):
        if self.enabled:
            self.pending.append(dist.all_reduce(flat_piece, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        for work in self.pending:
            work.wait()
        self.pending = []
        return self.scale


def broadcast_parameters_(module, src=0):
    """Make every rank start from rank `src`'s weights (what DDP's constructor does implicitly)."""
    if not exchange_active():
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
