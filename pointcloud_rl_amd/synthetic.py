"""Synthetic replay batches with the layout the reference's replay produces (SURVEY.md section 8a
row A0/A1, section 8d): obs / next_obs dicts of planar [B, C_key, N] tensors (xyz f32, rgb u8,
optional pos_encoding u8 / seg bool / agent f32), actions, rewards, dones.  Stand-in for
`ReplayMemory.sample(...).to_torch(...)` (pyrl/env/replay_buffer.py:297-322) -- the replay buffer
itself is outside the hot path."""
import numpy as np

from .utils.torch_utils import to_torch


def make_obs_np(g, B, N, pos_encoding=0, seg=0, agent=0):
    obs = {"xyz": g.randn(B, 3, N).astype(np.float32), "rgb": g.randint(0, 256, (B, 3, N)).astype(np.uint8)}
    if pos_encoding:
        pe = np.zeros((B, pos_encoding, N), np.uint8)
        per = max(N // pos_encoding, 1)
        for f in range(pos_encoding):
            pe[:, f, f * per:(f + 1) * per] = 1
        obs["pos_encoding"] = pe
    if seg:
        obs["seg"] = g.rand(B, seg, N) < 0.3
    if agent:
        obs["agent"] = g.randn(B, agent).astype(np.float32)
    return obs


def make_batch_np(B, N, action_dim, seed=1, **obs_kw):
    g = np.random.RandomState(seed)
    return dict(obs=make_obs_np(g, B, N, **obs_kw), next_obs=make_obs_np(g, B, N, **obs_kw),
                actions=g.uniform(-1, 1, (B, action_dim)).astype(np.float32),
                rewards=g.randn(B, 1).astype(np.float32), dones=(g.rand(B, 1) < 0.05),
                episode_dones=(g.rand(B, 1) < 0.05))


class _Sample:
    def __init__(self, batch):
        self.batch = batch

    def to_torch(self, device=None, non_blocking=False):
        return to_torch(self.batch, device=device, non_blocking=non_blocking)


class SyntheticReplay:
    """`memory` argument of update_parameters.  With device != None the batch is resident in HBM and
    .to_torch() is free (the benchmark's timed region starts with inputs on the device)."""

    def __init__(self, B, N, action_dim, seed=1, device=None, **obs_kw):
        self.batch_np = make_batch_np(B, N, action_dim, seed, **obs_kw)
        self.batch = to_torch(self.batch_np, device=device) if device is not None else self.batch_np

    def sample(self, batch_size):
        # shallow copies: update_parameters may rebind keys of the mapping it receives
        return _Sample({k: (dict(v) if isinstance(v, dict) else v) for k, v in self.batch.items()})
