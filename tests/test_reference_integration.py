"""INTEGRATION.md Level 1, executed against the REAL reference registries (build container only).

`/root/reference` never travels to the GPU box, so this test skips there.  Here it runs tests/_integration_probe.py in a child
process (the probe installs stand-ins for the reference's missing third-party imports into sys.modules -- tools/ref_stubs.py --
which must not leak into the pytest process): `Config.fromfile(configs/mfrl/{sac,drq}/.../pn*.py)` -> the reference's own
`build_agent` (pyrl/methods/builder.py:4-11 -> pyrl/utils/meta/registry.py:121-136), before and after the override block that
INTEGRATION.md tells a maintainer to paste (extracted from the markdown and exec'ed verbatim; registry.py:41-48 `force=True`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pyrl")), reason="the reference tree exists only in the build container")


@pytest.fixture(scope="module")
def probe():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_integration_probe.py")], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("INTEGRATION_JSON ")]
    assert len(line) == 1, out.stdout[-2000:]
    return json.loads(line[0].split(" ", 1)[1])


def test_the_override_block_is_the_one_in_integration_md(probe):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert probe["override"] in text and "pointcloud_rl_amd.bind_reference()" in probe["override"]


def test_reference_configs_build_reference_classes_before_the_override(probe):
    for name, d in probe["before"].items():
        assert d["agent_class"].startswith("pyrl.methods.mfrl."), name
        assert d["encoder_class"] == "pyrl.networks.backbones.pointnet.PointNet", name


def test_reference_configs_build_through_the_overridden_registries(probe):
    """Class identity: every registry lookup the shipped pn_* configs make now lands on this package's classes."""
    # the registered agents are bind.py's classes: this package's agent re-based onto the reference's BaseAgent too
    want_agent = {"sac": "pointcloud_rl_amd.bind.SAC", "drq": "pointcloud_rl_amd.bind.DrQ"}
    assert probe["extra"]["mfrl_sac_is_ours"] == "pointcloud_rl_amd.methods.sac"
    assert set(probe["after"]) >= {"sac_dmc_pn", "sac_maniskill_pn", "drq_dmc_pn_jitter", "drq_maniskill_pn_jitter"}
    for name, d in probe["after"].items():
        assert d["agent_class"] == want_agent[name.split("_")[0]], (name, d["agent_class"])
        assert d["encoder_class"] == "pointcloud_rl_amd.networks.pointnet.PointNet", name
        assert d["actor_class"] == "pointcloud_rl_amd.networks.actor_critic.ContinuousActor", name
        assert all(c.startswith("pointcloud_rl_amd.augmentations.") for c in d["aug_classes"]), (name, d["aug_classes"])
        if name.startswith("drq"):
            assert d["aug_classes"], name


def test_parameter_names_shapes_sharing_and_optimizer_groups_are_the_reference_s(probe):
    """Checkpoints address parameters and optimizers by attribute path (checkpoint_utils.py:215-237): the agent built through the
    overridden registries has the reference's named_parameters() -- names, order, shapes --, the same encoder sharing and the
    same one-group-per-tensor optimizers (optimizer_utils.py:43-57)."""
    for name in probe["before"]:
        b, a = probe["before"][name], probe["after"][name]
        assert a["params"] == b["params"], name
        assert a["encoder_shared"] is True and b["encoder_shared"] is True, name
        assert a["optim_groups"] == b["optim_groups"], name


def test_main_rl_sequence_on_the_overridden_registries(probe):
    """run_rl.py:298-313 replayed with the reference's own objects: build_agent, the parameter counts it logs, `.to()`, the
    `isinstance(agent, BaseAgent)` assert against the REFERENCE's base class; every driver-facing method still resolves to this
    package's (its classes come first in the MRO)."""
    for name, d in probe["extra"]["main_rl"].items():
        assert d["is_base_agent"] is True, (name, d["mro"])
        assert d["mro"][1].startswith("pointcloud_rl_amd.methods."), d["mro"]
        assert "pyrl.utils.torch.module_utils.BaseAgent" in d["mro"], d["mro"]
        assert d["mro"].index("pointcloud_rl_amd.utils.torch_utils.BaseAgent") < d["mro"].index("pyrl.utils.torch.module_utils.BaseAgent")
        assert d["to_ddp_is_ours"] == "pointcloud_rl_amd.utils.torch_utils", name
        assert d["num_trainable_parameters"] > 0 and d["size_trainable_parameters"] > 0


def test_the_reference_s_checkpoint_functions_keep_the_optimizers_after_the_first_update(probe):
    """checkpoint_utils.py:59-71, 226-229 only see `isinstance(child, Optimizer)`: the fused optimizers that replace
    torch.optim.Adam at the first update are torch Optimizers, so train_rl.py:392-405's mid-training checkpoint holds all three
    with their moments, and the reference's load_checkpoint restores them into a fresh agent (before or after ITS first update)."""
    keys = ["actor_optim", "critic_optim", "alpha_optim"]
    for name, d in probe["extra"]["main_rl"].items():
        assert d["optimizers_before_update"] == dict.fromkeys(keys, "Adam"), name
        assert d["optimizers_after_update"] == dict.fromkeys(keys, "HipAdam"), name
        assert d["optimizers_are_torch_optimizers"] is True
        assert sorted(d["state_dict_optim_keys_before_update"]) == sorted(keys)
        assert sorted(d["state_dict_optim_keys_after_update"]) == sorted(keys), name
        for k in keys:
            n_state, n_groups, step = d["moments_in_state_dict"][k]
            assert n_state == n_groups > 0 and step == 3.0, (name, k)
        for stage in ("fresh", "after_update"):
            assert d[f"reference_checkpoint_round_trip_{stage}"] is True, (name, stage, d[f"load_messages_{stage}"])
            assert d[f"load_messages_{stage}"] == [], (name, stage)
