"""Run by tests/test_reference_integration.py in a child process, in the build container only (it imports /root/reference).

Builds agents from the reference's own config files through the reference's own `Config.fromfile` / `build_agent`, once with the
reference's registries as shipped and once after applying, verbatim, the Level-1 override that INTEGRATION.md tells a
maintainer to paste into pyrl/methods/__init__.py; prints one JSON object describing both builds."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_stubs  # noqa: E402

ref_stubs.install()
REF = ref_stubs.REFERENCE_ROOT

DMC, MS = {"xyz": [3, 96], "rgb": [3, 96]}, {"xyz": [3, 96], "rgb": [3, 96], "seg": [1, 96], "agent": 10}
CASES = {
    # name: (config file, obs_shape, action_dim) -- every point-cloud SAC / DrQ config the reference ships (15 files)
    "sac_dmc_pn": ("configs/mfrl/sac/dm_control/pn.py", DMC, 6),
    "sac_dmc_pn_motivating": ("configs/mfrl/sac/dm_control/pn_motivating.py", DMC, 6),
    "sac_maniskill_pn": ("configs/mfrl/sac/maniskill/pn.py", MS, 8),
    "drq_dmc_pn_jitter": ("configs/mfrl/drq/dm_control/pn_jitter.py", DMC, 6),
    "drq_dmc_pn_rot": ("configs/mfrl/drq/dm_control/pn_rot.py", DMC, 6),
    "drq_dmc_pn_shift": ("configs/mfrl/drq/dm_control/pn_shift.py", DMC, 6),
    "drq_dmc_pn_shift_motivating": ("configs/mfrl/drq/dm_control/pn_shift_motivating.py", DMC, 6),
    "drq_dmc_pn_dropout": ("configs/mfrl/drq/dm_control/pn_dropout.py", DMC, 6),
    "drq_dmc_pn_colorjitter": ("configs/mfrl/drq/dm_control/pn_colorjitter.py", DMC, 6),
    "drq_maniskill_pn_jitter": ("configs/mfrl/drq/maniskill/pn_jitter.py", MS, 8),
    "drq_maniskill_pn_rot": ("configs/mfrl/drq/maniskill/pn_rot.py", MS, 8),
    "drq_maniskill_pn_shift": ("configs/mfrl/drq/maniskill/pn_shift.py", MS, 8),
    "drq_maniskill_pn_dropout": ("configs/mfrl/drq/maniskill/pn_dropout.py", MS, 8),
    "drq_maniskill_pn_colorjitter": ("configs/mfrl/drq/maniskill/pn_colorjitter.py", MS, 8),
}
# configs/mfrl/drq/dm_control/pn_sample.py names an augmentation (RandomDownSampleAndFilter) that the reference's own pyrl does not
# define: it cannot be built on either side, and both must say so
UNBUILDABLE = {"drq_dmc_pn_sample": ("configs/mfrl/drq/dm_control/pn_sample.py", DMC, 6)}


def describe(agent):
    enc = agent.actor.backbone.visual_nn
    return {
        "agent_class": f"{type(agent).__module__}.{type(agent).__qualname__}",
        "encoder_class": f"{type(enc).__module__}.{type(enc).__qualname__}",
        "actor_class": f"{type(agent.actor).__module__}.{type(agent.actor).__qualname__}",
        "aug_classes": [f"{type(t).__module__}.{type(t).__qualname__}" for t in getattr(getattr(agent, "obs_aug", None), "transforms", [])],
        "params": [[n, list(p.shape)] for n, p in agent.named_parameters()],
        "encoder_shared": all(v.backbone.visual_nn is enc for v in list(agent.critic.values) + list(agent.target_critic.values)),
        "optim_groups": {k: len(getattr(agent, k).param_groups) for k in ("actor_optim", "critic_optim", "alpha_optim")},
    }


def build_all():
    out = {}
    for name, (cfg, obs_shape, A) in CASES.items():
        agent, _ = ref_stubs.build_reference_agent(os.path.join(REF, cfg), obs_shape, A, seed=0)
        out[name] = describe(agent)
    for name, (cfg, obs_shape, A) in UNBUILDABLE.items():
        try:
            ref_stubs.build_reference_agent(os.path.join(REF, cfg), obs_shape, A, seed=0)
            out[name] = {"error": None}
        except Exception as err:                       # noqa: BLE001 -- whatever the registry raises is the answer
            out[name] = {"error": f"{type(err).__name__}: {str(err)[:200]}"}
    return out


def _pyrl_env_stubs():
    for m in ("gym.envs", "gym.wrappers", "gym.core", "gym.envs.registration", "h5py", "cv2", "imageio", "matplotlib", "matplotlib.pyplot"):
        if m not in sys.modules:
            ref_stubs._stub(m)        # pyrl.env imports its simulators' and video writers' dependencies at package import time


def _transitions(g, n, obs_shape, A):
    import numpy as np

    def obs():
        o = {}
        for k, shp in obs_shape.items():
            if k == "xyz":
                o[k] = g.randn(n, *shp).astype(np.float32)
            elif k == "rgb":
                o[k] = g.randint(0, 256, (n, *shp)).astype(np.uint8)
            elif k == "seg":
                o[k] = g.rand(n, *shp) < 0.3
            else:
                o[k] = g.randn(n, shp).astype(np.float32)
        return o
    return dict(obs=obs(), next_obs=obs(), actions=g.uniform(-1, 1, (n, A)).astype(np.float32), rewards=g.randn(n, 1).astype(np.float32),
                dones=g.rand(n, 1) < 0.1, episode_dones=g.rand(n, 1) < 0.1)


def _structure(batch):
    import torch
    out = {}
    for k in ("obs", "next_obs", "actions", "rewards", "dones"):
        v = batch[k]
        out[k] = ({kk: [str(vv.dtype), list(vv.shape)] for kk, vv in v.items()} if isinstance(v, dict) else [str(v.dtype), list(v.shape)])
        assert all(torch.is_tensor(t) for t in (v.values() if isinstance(v, dict) else [v])), k
    return out


def replay_seam(name, fixture_path=None):
    """The seam the reference drives every step (sac.py:104-108): the REFERENCE's ReplayMemory (replay_buffer.py:206-322) -> its
    `sample()` -> `GDict.to_torch()` (dict_array.py:308-318) -> `process_obs` -> the bound agent's step input, up to the first GPU call:
    `agent._fetcher(memory)()` and `_to_static` (what feeds the captured graphs), on CPU."""
    import numpy as np
    import torch
    _pyrl_env_stubs()
    from pyrl.env.replay_buffer import ReplayMemory
    from pyrl.utils.data import GDict
    cfg_file, obs_shape, A = CASES[name]
    agent, cfg = ref_stubs.build_reference_agent(os.path.join(REF, cfg_file), obs_shape, A, seed=0)
    agent = agent.to("cpu")
    agent.batch_size = 8
    replay_cfg = dict(cfg.get("replay_cfg", None) or {})
    sampling_cfg = dict(replay_cfg.get("sampling_cfg", None) or dict(type="OneStepTransition"))
    sampling_cfg.setdefault("seed", 3)
    mem = ReplayMemory(capacity=48, sampling_cfg=sampling_cfg)
    g = np.random.RandomState(11)
    pushes = [_transitions(g, 20, obs_shape, A) for _ in range(3)]           # 60 into a ring of 48: the last push wraps
    for items in pushes:
        mem.push_batch({k: (dict(v) if isinstance(v, dict) else v) for k, v in items.items()})
    fetch = agent._fetcher(mem)
    batch = fetch()
    out = {"sample_type": f"{type(mem.sample(8)).__module__}.{type(mem.sample(8)).__qualname__}",
           "batch_type": f"{type(batch).__module__}.{type(batch).__qualname__}", "is_gdict": isinstance(batch, GDict),
           "replay_cfg_type": replay_cfg.get("type"), "sampling_type": sampling_cfg.get("type"), "structure": _structure(batch),
           "obs_value_type": type(batch["obs"]).__name__, "persistent": bool(getattr(batch, "persistent", False))}
    # _to_static (what feeds the captured graphs after agent.enable_graphs()): the first call adopts (clones), later calls copy into the
    # same buffers
    agent.enable_graphs(True)
    static = agent._to_static(batch)
    ptrs = {k: ({kk: vv.data_ptr() for kk, vv in v.items()} if isinstance(v, dict) else v.data_ptr()) for k, v in static.items()}
    batch2 = fetch()
    static2 = agent._to_static(batch2)
    same_buffers = all((({kk: vv.data_ptr() for kk, vv in v.items()} if isinstance(v, dict) else v.data_ptr()) == ptrs[k]) for k, v in static2.items())
    equal = all(torch.equal(a, b) for k in static2 for a, b in
                (zip(static2[k].values(), batch2[k].values()) if isinstance(static2[k], dict) else [(static2[k], batch2[k])]))
    out.update(static_structure=_structure(static2), static_buffers_reused=bool(same_buffers), static_holds_the_second_batch=bool(equal))
    # DrQ: the augmentation pipeline takes the GDict's obs (a plain dict of tensors) as it is
    if hasattr(agent, "_augment"):
        aug = agent._augment(batch2["obs"], virtual=True)
        out["augmented_type"] = type(aug).__name__
        out["augmented_keys"] = sorted(k for k in aug.keys())
    if fixture_path is not None:
        arrays = {}
        for k in ("obs", "next_obs", "actions", "rewards", "dones", "episode_dones"):
            v = batch2[k]
            for kk, vv in (v.items() if isinstance(v, dict) else [(None, v)]):
                arrays[k if kk is None else f"{k}/{kk}"] = vv.numpy()
        np.savez_compressed(fixture_path, **arrays)
    return out


def train_rl_cadence(name):
    """train_rl.py:292-296 and 392-405 with the reference's own helpers (EveryNSteps, save_checkpoint) around the bound agent: the update
    loop calls `agent.update_parameters(replay, updates=total_updates)` and reads `training_infos.get("grad_steps", 1)`; the checkpoint
    block calls to_normal / save_checkpoint / recover_ddp.  No GPU here: the agent's own first GPU call is where the loop must stop, with
    this package's explicit error, after everything before it has run."""
    import tempfile
    import numpy as np
    _pyrl_env_stubs()
    from pyrl.env.replay_buffer import ReplayMemory
    from pyrl.utils.math import EveryNSteps                    # train_rl.py:11, 201
    from pyrl.utils.torch import save_checkpoint
    cfg_file, obs_shape, A = CASES[name]
    agent, cfg = ref_stubs.build_reference_agent(os.path.join(REF, cfg_file), obs_shape, A, seed=0)
    agent = agent.to("cpu")
    agent.batch_size = 8
    replay = ReplayMemory(capacity=32, sampling_cfg=dict(type="OneStepTransition", seed=1))
    replay.push_batch(_transitions(np.random.RandomState(2), 32, obs_shape, A))
    out = {}
    agent.train()
    total_updates, grad_steps = 0, 0
    try:
        for i in range(2):                                    # train_rl.py:292-296
            total_updates += 1
            training_infos = agent.update_parameters(replay, updates=total_updates)
            grad_steps += training_infos.get("grad_steps", 1)
        out["update_error"] = None
    except RuntimeError as err:
        out["update_error"] = str(err)[:200]
    out["total_updates_reached"] = total_updates
    check_checkpoint = EveryNSteps(5)
    out["checkpoint_steps"] = [s for s in range(1, 12) if check_checkpoint.check(s)]
    with tempfile.TemporaryDirectory() as tmp:                # train_rl.py:392-405
        model_path = os.path.join(tmp, f"model_{check_checkpoint.standard(5)}.ckpt")
        agent.to_normal()
        save_checkpoint(agent, model_path)
        agent.recover_ddp()
        out["checkpoint_written"] = os.path.getsize(model_path) > 0
    return out


def main_rl_sequence(cfg_file, obs_shape, A):
    """What `main_rl` does to the agent between building it and the first rollout (run_rl.py:283, 298-313), then what
    train_rl.py:392-405 does at a checkpoint -- with the REFERENCE's own functions, on the agent the overridden registries
    build, before and after the first update has swapped the optimizers for the fused ones."""
    import tempfile
    import torch
    from pyrl.utils.torch import BaseAgent, load_checkpoint, save_checkpoint          # run_rl.py:283
    from pyrl.utils.torch.checkpoint_utils import get_state_dict, load_state_dict
    agent, _ = ref_stubs.build_reference_agent(os.path.join(REF, cfg_file), obs_shape, A, seed=0)
    out = {"num_trainable_parameters": int(agent.num_trainable_parameters), "size_trainable_parameters": float(agent.size_trainable_parameters)}
    agent = agent.to("cpu")                                                            # run_rl.py:306-307 (no GPU in this container)
    out["is_base_agent"] = isinstance(agent, BaseAgent)                                # run_rl.py:308
    out["mro"] = [f"{c.__module__}.{c.__qualname__}" for c in type(agent).__mro__]
    out["to_ddp_is_ours"] = type(agent).to_ddp.__module__
    optim_keys = ("actor_optim", "critic_optim", "alpha_optim")
    out["optimizers_before_update"] = {k: type(getattr(agent, k)).__name__ for k in optim_keys}
    out["state_dict_optim_keys_before_update"] = [k for k in get_state_dict(agent) if k in optim_keys]
    # the first update_parameters swaps torch.optim.Adam for the fused optimizer (SAC._prepare_buffers); give it a state
    agent._prepare_buffers()
    out["optimizers_after_update"] = {k: type(getattr(agent, k)).__name__ for k in optim_keys}
    out["optimizers_are_torch_optimizers"] = all(isinstance(getattr(agent, k), torch.optim.Optimizer) for k in optim_keys)
    g = torch.Generator().manual_seed(5)
    for k in optim_keys:
        opt = getattr(agent, k)
        for m, v in zip(opt._views(opt.exp_avg), opt._views(opt.exp_avg_sq)):      # the tensors' own ranges (padding floats stay zero)
            m.copy_(torch.randn(m.shape, generator=g))
            v.copy_(torch.rand(v.shape, generator=g))
        opt.step_counter.fill_(3)
    sd = get_state_dict(agent)                                                          # checkpoint_utils.py:215-237
    out["state_dict_optim_keys_after_update"] = [k for k in sd if k in optim_keys]
    out["moments_in_state_dict"] = {k: [len(sd[k]["state"]), len(sd[k]["param_groups"]), float(sd[k]["state"][0]["step"])] for k in optim_keys if k in sd}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "models", "model_3.ckpt")
        agent.to_normal()                                                               # train_rl.py:396-405
        save_checkpoint(agent, path)
        agent.recover_ddp()
        # a fresh agent (torch.optim.Adam before its first update, fused after): both must take the file back
        for stage in ("fresh", "after_update"):
            other, _ = ref_stubs.build_reference_agent(os.path.join(REF, cfg_file), obs_shape, A, seed=1)
            if stage == "after_update":
                other._prepare_buffers()
            msgs = []

            class _Log:
                info = warning = staticmethod(msgs.append)

                def __call__(self, m):
                    msgs.append(m)
            load_checkpoint(other, path, "cpu", logger=_Log())                          # run_rl.py:313
            if stage == "fresh":
                other._prepare_buffers()                                                # its first update carries the loaded Adam state over
            same = all(torch.equal(a, b) for (_, a), (_, b) in zip(agent.named_parameters(), other.named_parameters()))
            for k in optim_keys:
                a, b = getattr(agent, k), getattr(other, k)
                same = same and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq) and int(b.step_counter) == 3
            out[f"reference_checkpoint_round_trip_{stage}"] = bool(same)
            out[f"load_messages_{stage}"] = [m for m in msgs if "optimizer" in str(m)]
    return out


def main():
    before = build_all()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# pyrl/methods/__init__.py  \(reference side\)\n.*?)```", text, re.S).group(1)
    exec(compile(block, "INTEGRATION.md:level-1", "exec"), {})
    after = build_all()
    import pointcloud_rl_amd.utils.torch_utils as tu
    from pyrl.methods.builder import MFRL
    extra = {"mfrl_sac_is_ours": MFRL.get("SAC").__mro__[1].__module__, "base_agent": f"{tu.BaseAgent.__module__}.{tu.BaseAgent.__qualname__}",
             "main_rl": {name: main_rl_sequence(*CASES[name]) for name in ("sac_dmc_pn", "drq_maniskill_pn_jitter")},
             "replay_seam": {name: replay_seam(name, os.environ.get("PCRL_SEAM_FIXTURE") if name == "drq_maniskill_pn_jitter" else None)
                             for name in ("sac_dmc_pn", "sac_maniskill_pn", "drq_maniskill_pn_jitter", "drq_dmc_pn_dropout")},
             "train_rl": {name: train_rl_cadence(name) for name in ("sac_dmc_pn", "drq_maniskill_pn_jitter")}}
    print("INTEGRATION_JSON " + json.dumps({"before": before, "after": after, "extra": extra, "override": block}))


if __name__ == "__main__":
    main()
