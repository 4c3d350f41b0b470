"""Device-resident replay buffer.

Same surface as the part of the reference's ReplayMemory the update loop uses
(pyrl/env/replay_buffer.py:206-322: push_batch, sample, __len__, reset, get_all, tail) with
OneStepTransition's uniform with-replacement sampling (pyrl/env/sampling_strategy.py:26-31,93-101:
`np_random.randint(0, len, batch_size)`), but the ring lives in HBM: `sample()` is ONE HIP launch that
draws the B row numbers (Philox keyed by the buffer's seed and the sample-call count; with
`host_rng=True` they come from numpy's RandomState exactly as in the reference and are shipped as B
int32; `with_replacement=False` walks a shuffled epoch order the same way) and gathers every key into a
persistent staging batch.  The staging tensors keep their addresses from call to
call, so an agent replaying its update step from a hipGraph reads them in place -- no per-step
numpy gather, no pageable host->device copy of 2*B point clouds, no copy into graph inputs.

There is no CPU implementation: constructing the buffer on a CPU device raises.
"""
import numpy as np
import torch

from . import hip


class PersistentBatch(dict):
    """Mapping returned by `DeviceReplay.sample(...).to_torch(...)`; `persistent` tells the agent that the
    tensors are long-lived staging buffers that it may read in place on every step."""
    persistent = True


class _DeviceSample:
    def __init__(self, batch):
        self.batch = batch

    def to_torch(self, device=None, non_blocking=False):
        out = PersistentBatch()
        for k, v in self.batch.items():
            out[k] = dict(v) if isinstance(v, dict) else v
        return out

    def __getitem__(self, key):
        return self.batch[key]


def _flatten(items, prefix=""):
    for k, v in items.items():
        if isinstance(v, dict):
            yield from _flatten(v, prefix + k + "/")
        else:
            yield prefix + k, v


class DeviceReplay:
    MAX_BATCH = 16384         # rows per sample() the launch's ticket words are sized for

    def __init__(self, capacity, device="cuda", seed=None, with_replacement=True, host_rng=False):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("DeviceReplay keeps the ring in MI355X HBM: device must be a CUDA/HIP device")
        self.with_replacement = bool(with_replacement)
        if not with_replacement:
            host_rng = True            # the shuffled epoch order is drawn on the host exactly as the reference draws it
        self.capacity, self.device = int(capacity), device
        self.seed = np.random.randint(0, 2 ** 32 - 1) if seed is None else seed
        self.np_random = np.random.RandomState(self.seed)          # sampling_strategy.py:18-19
        self.storage = None            # flat key -> tensor [capacity, ...]
        self.position = self.running_count = 0
        self._staging = {}             # batch_size -> (flat key -> tensor [B, ...], nested mapping, idx device, pinned idx x2)
        self._flip = 0
        self.host_rng, self.draws = host_rng, 0
        # device copy of {sample-call count, len(self), workgroup ticket}: the sampling launch reads its per-call values from here
        # (and advances the count itself), so a captured update step can contain it (`sample(..., launch=False)`)
        self.state = torch.zeros(3 + self.MAX_BATCH, dtype=torch.int64, device=device)
        self.items, self.item_index, self.need_update = None, 0, False    # without-replacement state (sampling_strategy.py:21-24)

    # -- ring ------------------------------------------------------------------------------------------
    def __len__(self):
        return min(self.running_count, self.capacity)

    def reset(self):
        self.position = self.running_count = 0
        self.items, self.item_index = None, 0
        self.state[1:].fill_(0)          # size, and every ticket word (an aborted sampling launch may have left some non-zero)

    @property
    def graph_sampling(self):
        """True when `sample` is one launch that reads nothing from the host: an agent may capture it in its step graph and
        call `sample(batch_size, launch=False)` on replays (bookkeeping only)."""
        return not self.host_rng

    def _as_tensor(self, v):
        t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
        return t.to(self.device, non_blocking=True)

    def push_batch(self, items):
        """items: nested dict of arrays/tensors with a leading batch axis (replay_buffer.py:206-232)."""
        flat = dict(_flatten(items))
        n = len(next(iter(flat.values())))
        flat.setdefault("worker_indices", np.zeros([n, 1], dtype=np.int32))
        flat.setdefault("is_truncated", np.zeros([n, 1], dtype=np.bool_))
        if n > self.capacity:
            flat, n = {k: v[:self.capacity] for k, v in flat.items()}, self.capacity
        flat = {k: self._as_tensor(v) for k, v in flat.items()}
        if self.storage is None:
            self.storage = {k: torch.zeros((self.capacity,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device) for k, v in flat.items()}
        assert set(flat) == set(self.storage), f"keys changed: {sorted(flat)} vs {sorted(self.storage)}"
        first = min(n, self.capacity - self.position)
        for k, v in flat.items():
            self.storage[k][self.position:self.position + first].copy_(v[:first])
            if first < n:                                   # wrap around (replay_buffer.py:221-226)
                self.storage[k][:n - first].copy_(v[first:])
        self.running_count += n
        self.position = (self.position + n) % self.capacity
        self.state[1:2].fill_(len(self))
        self.need_update = True                              # OneStepTransition.push_batch (sampling_strategy.py:80-83)

    def get_all(self):
        return self._nest({k: v[:len(self)] for k, v in self.storage.items()})

    def tail(self, num):
        assert num <= len(self), f"num={num} is larger than buffer length={len(self)}!"
        if num <= self.position:
            return self._nest({k: v[self.position - num:self.position] for k, v in self.storage.items()})
        return self._nest({k: torch.cat([v[self.capacity - num + self.position:], v[:self.position]]) for k, v in self.storage.items()})

    @staticmethod
    def _nest(flat):
        out = {}
        for k, v in flat.items():
            node = out
            parts = k.split("/")
            for part in parts[:-1]:
                node = node.setdefault(part, {})
            node[parts[-1]] = v
        return out

    # -- sampling ----------------------------------------------------------------------------------------
    def _stage(self, batch_size):
        if batch_size not in self._staging:
            flat = {k: torch.zeros((batch_size,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device) for k, v in self.storage.items()}
            # obs/<key> and next_obs/<key> are the two halves of ONE [2 B, ...] allocation: an encoder that sees them adjacent
            # encodes s and s' in one launch (methods/fused.py)
            for k in list(flat):
                twin = "next_obs/" + k[len("obs/"):]
                if k.startswith("obs/") and twin in flat and flat[twin].shape == flat[k].shape and flat[twin].dtype == flat[k].dtype:
                    pair = torch.zeros((2 * batch_size,) + tuple(flat[k].shape[1:]), dtype=flat[k].dtype, device=self.device)
                    flat[k], flat[twin] = pair[:batch_size], pair[batch_size:]
            segs = hip.gather_segments([(self.storage[k], flat[k]) for k in self.storage])
            flat["is_valid"] = torch.ones(batch_size, 1, dtype=torch.bool, device=self.device)        # sampling_strategy.py:101
            idx = torch.zeros(batch_size, dtype=torch.int32, device=self.device)
            pinned = [torch.zeros(batch_size, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._staging[batch_size] = (flat, _DeviceSample(self._nest(flat)), idx, pinned, segs)
        return self._staging[batch_size]

    def sample_indices(self, batch_size, drop_last=True, auto_restart=True):
        """SamplingStrategy.get_index (sampling_strategy.py:26-48): uniform with replacement, or consecutive slices of a
        shuffled epoch order that is re-drawn after a push or when exhausted (None when exhausted and not auto_restart)."""
        capacity = len(self)
        if self.with_replacement:
            return self.np_random.randint(low=0, high=capacity, size=batch_size)
        if self.items is None or self.need_update:
            self.need_update = False
            self.items = np.arange(capacity)
            self.np_random.shuffle(self.items)
            self.item_index = 0
        min_query_size = batch_size if drop_last else 1
        if self.item_index + min_query_size > capacity:
            if not auto_restart:
                return None
            self.np_random.shuffle(self.items)
            self.item_index = 0
        else:
            batch_size = min(batch_size, capacity - self.item_index)
        index = self.items[self.item_index:self.item_index + batch_size]
        self.item_index += batch_size
        return index

    def launch_sample(self, batch_size):
        """The sampling launch alone (device-drawn rows into the staging batch), not counted as a call: what an agent puts at
        the head of its captured step; every later `sample(batch_size, launch=False)` stands for one replay of it."""
        assert self.graph_sampling, "host-drawn row numbers cannot be captured"
        flat, sample, idx, pinned, segs = self._stage(batch_size)
        assert batch_size <= self.MAX_BATCH
        hip.replay_sample_gather_state(segs, batch_size, self.capacity, self.seed, self.state, idx)
        return sample

    def sample(self, batch_size, auto_restart=True, drop_last=True, launch=True):
        """launch=False (only with `graph_sampling`): the caller replays a hipGraph that contains this call's launch -- the
        staging batch is returned and the call is counted, nothing is launched here."""
        size = len(self)
        if size == 0:
            raise RuntimeError("sampling from an empty replay buffer")
        if self.host_rng:
            index = self.sample_indices(batch_size, drop_last, auto_restart)
            if index is None:
                return None
            batch_size = len(index)                     # a short last batch when drop_last is False
        flat, sample, idx, pinned, segs = self._stage(batch_size)
        if self.host_rng:
            host = pinned[self._flip]
            self._flip ^= 1
            host.numpy()[:] = index
            idx.copy_(host, non_blocking=True)
            hip.replay_gather(segs, idx, self.capacity)
        elif launch:
            assert batch_size <= self.MAX_BATCH
            hip.replay_sample_gather_state(segs, batch_size, self.capacity, self.seed, self.state, idx)
        self.draws += 1
        return sample

    def last_indices(self, batch_size):
        """Row numbers used by the latest sample(batch_size) (device int32 [B])."""
        return self._staging[batch_size][2]
