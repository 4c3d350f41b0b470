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


class TStepIndex:
    """Host-side bookkeeping of the reference's TStepTransition (pyrl/env/sampling_strategy.py:105-246): which runs of `horizon`
    consecutive ring positions of one worker's episode are valid sampling units (horizon = -1: whole episodes), kept up to
    date as transitions are pushed and as the ring overwrites old ones.  Only `episode_dones`, `is_truncated` and
    `worker_indices` of a push are looked at; the transitions themselves stay in HBM."""

    def __init__(self, capacity, horizon):
        self.capacity, self.horizon = int(capacity), int(horizon)
        self.reset()

    def reset(self):
        self.position = self.running_count = 0
        self.worker_indices = np.zeros(self.capacity, dtype=np.int16) - 1
        self.num_procs = 0
        self.current_episode, self.valid_seq = [], []

    def __len__(self):
        return int(np.sum([len(v) for v in self.valid_seq])) if self.valid_seq else 0

    def push(self, worker, done):
        """One transition lands on ring position self.position (sampling_strategy.py:147-209)."""
        if worker + 1 > self.num_procs:
            for _ in range(worker + 1 - self.num_procs):
                self.current_episode.append([])
                self.valid_seq.append([])
            self.num_procs = worker + 1
        if self.worker_indices[self.position] >= 0:           # the ring overwrites an old transition: retire what started there
            last = self.worker_indices[self.position]
            if len(self.current_episode[last]) > 0 and self.position == self.current_episode[last][0]:
                self.current_episode[last].pop(0)
            if self.valid_seq[last] and self.position == self.valid_seq[last][0][0]:
                if self.horizon > 0:
                    self.valid_seq[last].pop(0)
                else:
                    self.valid_seq[last][0].pop(0)
                    if len(self.valid_seq[last][0]) == 0:
                        self.valid_seq[last].pop(0)
        self.current_episode[worker].append(self.position)
        self.worker_indices[self.position] = worker
        if self.horizon > 0:
            if len(self.current_episode[worker]) >= self.horizon:
                self.valid_seq[worker].append(self.current_episode[worker][-self.horizon:])
        elif done:
            self.valid_seq[worker].append(self.current_episode[worker])
        if done:
            self.current_episode[worker] = []
        self.running_count += 1
        self.position = (self.position + 1) % self.capacity

    def blocks(self, index):
        """index: unit numbers drawn by the sampler -> (int32 [B, H] ring positions, bool [B, H, 1] validity), short units
        (whole-episode mode) padded with their first position (sampling_strategy.py:228-246)."""
        query = np.cumsum([len(v) for v in self.valid_seq])
        ret = []
        for i in index:
            j = int(np.searchsorted(query, i, side="right"))
            ret.append(self.valid_seq[j][i - (0 if j == 0 else query[j - 1])])
        width = max(len(r) for r in ret)
        mask = np.zeros([len(ret), width, 1], dtype=np.bool_)
        for i, r in enumerate(ret):
            mask[i, :len(r)] = True
            ret[i] = list(r) + [r[0]] * (width - len(r))
        return np.asarray(ret, dtype=np.int32), mask


class DeviceReplay:
    MAX_BATCH = 16384         # rows per sample() the launch's ticket words are sized for

    def __init__(self, capacity, device="cuda", seed=None, with_replacement=True, host_rng=False, horizon=1, sampling_cfg=None):
        """sampling_cfg: the reference's `ReplayMemory(capacity, sampling_cfg=dict(type="OneStepTransition" | "TStepTransition",
        horizon=, with_replacement=, seed=))` spelling of the same choices (replay_buffer.py:151-170)."""
        if sampling_cfg is not None:
            cfg = dict(sampling_cfg)
            kind = cfg.pop("type", "OneStepTransition")
            assert kind in ("OneStepTransition", "TStepTransition"), f"unknown sampling strategy {kind}"
            horizon = cfg.pop("horizon", 1) if kind == "TStepTransition" else 1
            with_replacement = cfg.pop("with_replacement", with_replacement)
            seed = cfg.pop("seed", seed)
            cfg.pop("capacity", None)
            assert not cfg, f"unsupported sampling options {sorted(cfg)}"
            if kind == "TStepTransition":
                host_rng = True
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("DeviceReplay keeps the ring in MI355X HBM: device must be a CUDA/HIP device")
        self.with_replacement = bool(with_replacement)
        self.horizon = int(horizon)
        # T-step sampling (horizon != 1, or asked for by name): [B, H] index blocks of consecutive transitions of one episode,
        # drawn on the host from the bookkeeping above exactly as the reference does; the gather stays one launch
        self.tstep = TStepIndex(capacity, self.horizon) if (self.horizon != 1 or (sampling_cfg or {}).get("type") == "TStepTransition") else None
        if not with_replacement or self.tstep is not None:
            host_rng = True            # the shuffled epoch order / the unit numbers are drawn on the host exactly as the reference draws them
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
        if self.tstep is not None:
            self.tstep.reset()
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
        if self.tstep is not None:      # episode bookkeeping from the three flag columns (host values), one transition at a time
            host = lambda v: v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            ed, tr, wi = host(flat["episode_dones"]), host(flat["is_truncated"]), host(flat["worker_indices"])
            for i in range(n):
                self.tstep.push(int(wi[i].reshape(-1)[0]), bool(ed[i].reshape(-1)[0]) or bool(tr[i].reshape(-1)[0]))
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

    def sample_indices(self, batch_size, drop_last=True, auto_restart=True, capacity=None):
        """SamplingStrategy.get_index (sampling_strategy.py:26-48): uniform with replacement, or consecutive slices of a
        shuffled epoch order that is re-drawn after a push or when exhausted (None when exhausted and not auto_restart).
        capacity: the number of sampling units (T-step: valid index blocks) when it is not the number of transitions."""
        capacity = len(self) if capacity is None else capacity
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
        if self.tstep is not None:
            return self._sample_tstep(batch_size, auto_restart, drop_last)
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

    def _sample_tstep(self, batch_size, auto_restart, drop_last):
        """TStepTransition.sample + ReplayMemory.sample (sampling_strategy.py:211-246, replay_buffer.py:297-322): every key comes
        back as [B, H, ...], `is_valid` [B, H, 1] marks the padding of short episodes (horizon = -1)."""
        units = len(self.tstep)        # (the reference's "samples will be throwed out" guard compares this number with itself)
        if units == 0:
            raise RuntimeError(f"no run of {self.horizon} consecutive transitions of one episode in the buffer yet")
        index = self.sample_indices(batch_size, drop_last, auto_restart, capacity=units)
        if index is None:
            return None
        blocks, mask = self.tstep.blocks(index)
        B, H = blocks.shape
        flat, sample, idx, pinned, segs = self._stage(B * H)
        host = pinned[self._flip]
        self._flip ^= 1
        host.numpy()[:] = blocks.reshape(-1)
        idx.copy_(host, non_blocking=True)
        hip.replay_gather(segs, idx, self.capacity)
        self.draws += 1
        out = {k: v.view((B, H) + tuple(v.shape[1:])) for k, v in flat.items() if k != "is_valid"}
        out["is_valid"] = torch.from_numpy(mask).to(self.device, non_blocking=True)
        return _DeviceSample(self._nest(out))

    def last_indices(self, batch_size):
        """Row numbers used by the latest sample(batch_size) (device int32 [B])."""
        return self._staging[batch_size][2]
