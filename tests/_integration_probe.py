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

CASES = {
    # name: (config file, obs_shape, action_dim)
    "sac_dmc_pn": ("configs/mfrl/sac/dm_control/pn.py", {"xyz": [3, 96], "rgb": [3, 96]}, 6),
    "sac_dmc_pn_motivating": ("configs/mfrl/sac/dm_control/pn_motivating.py", {"xyz": [3, 96], "rgb": [3, 96]}, 6),
    "sac_maniskill_pn": ("configs/mfrl/sac/maniskill/pn.py", {"xyz": [3, 96], "rgb": [3, 96], "seg": [1, 96], "agent": 10}, 8),
    "drq_dmc_pn_jitter": ("configs/mfrl/drq/dm_control/pn_jitter.py", {"xyz": [3, 96], "rgb": [3, 96]}, 6),
    "drq_dmc_pn_rot": ("configs/mfrl/drq/dm_control/pn_rot.py", {"xyz": [3, 96], "rgb": [3, 96]}, 6),
    "drq_maniskill_pn_jitter": ("configs/mfrl/drq/maniskill/pn_jitter.py", {"xyz": [3, 96], "rgb": [3, 96], "seg": [1, 96], "agent": 10}, 8),
}


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
             "main_rl": {name: main_rl_sequence(*CASES[name]) for name in ("sac_dmc_pn", "drq_maniskill_pn_jitter")}}
    print("INTEGRATION_JSON " + json.dumps({"before": before, "after": after, "extra": extra, "override": block}))


if __name__ == "__main__":
    main()
