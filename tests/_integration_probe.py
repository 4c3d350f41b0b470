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
        "is_base_agent": None,
    }


def build_all():
    out = {}
    for name, (cfg, obs_shape, A) in CASES.items():
        agent, _ = ref_stubs.build_reference_agent(os.path.join(REF, cfg), obs_shape, A, seed=0)
        out[name] = describe(agent)
    return out


def main():
    before = build_all()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# pyrl/methods/__init__.py  \(reference side\)\n.*?)```", text, re.S).group(1)
    exec(compile(block, "INTEGRATION.md:level-1", "exec"), {})
    after = build_all()
    import pointcloud_rl_amd.utils.torch_utils as tu
    from pyrl.methods.builder import MFRL
    extra = {"mfrl_sac_is_ours": MFRL.get("SAC").__module__, "base_agent": f"{tu.BaseAgent.__module__}.{tu.BaseAgent.__qualname__}"}
    print("INTEGRATION_JSON " + json.dumps({"before": before, "after": after, "extra": extra, "override": block}))


if __name__ == "__main__":
    main()
