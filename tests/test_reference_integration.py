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
        if "error" in d:
            continue
        assert d["agent_class"].startswith("pyrl.methods.mfrl."), name
        assert d["encoder_class"] == "pyrl.networks.backbones.pointnet.PointNet", name


def test_reference_configs_build_through_the_overridden_registries(probe):
    """Class identity: every registry lookup the shipped pn_* configs make now lands on this package's classes."""
    # the registered agents are bind.py's classes: this package's agent re-based onto the reference's BaseAgent too
    want_agent = {"sac": "pointcloud_rl_amd.bind.SAC", "drq": "pointcloud_rl_amd.bind.DrQ"}
    assert probe["extra"]["mfrl_sac_is_ours"] == "pointcloud_rl_amd.methods.sac"
    assert set(probe["after"]) >= {"sac_dmc_pn", "sac_maniskill_pn", "drq_dmc_pn_jitter", "drq_maniskill_pn_jitter"}
    built = {k: v for k, v in probe["after"].items() if "error" not in v}
    assert len(built) == 14, sorted(built)                 # every shipped configs/mfrl/{sac,drq}/*/pn_*.py that the reference itself can build
    for name, d in built.items():
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
        if "error" in b:
            continue
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


def test_the_one_config_the_reference_cannot_build_fails_the_same_way_on_both_sides(probe):
    """configs/mfrl/drq/dm_control/pn_sample.py names RandomDownSampleAndFilter, which pyrl does not define: the registries say so before
    and after the override (no silent fallback)."""
    for side in ("before", "after"):
        err = probe[side]["drq_dmc_pn_sample"]["error"]
        assert err is not None and "RandomDownSampleAndFilter" in err, (side, err)


def test_every_shipped_augmentation_config_lands_on_this_packages_class(probe):
    want = {"jitter": "RandomJitterPoints", "rot": "GlobalRotScaleTrans", "shift": "GlobalRotScaleTrans", "shift_motivating": "GlobalRotScaleTrans",
            "dropout": "RandomDownSample", "colorjitter": "ColorJitterPoints"}
    seen = set()
    for name, d in probe["after"].items():
        if name.startswith("drq") and "error" not in d:
            kind = name.split("_pn_", 1)[1]
            assert d["aug_classes"] == [f"pointcloud_rl_amd.augmentations.{want[kind]}"], (name, d["aug_classes"])
            seen.add(kind)
    assert seen == set(want)


def test_the_reference_replay_container_feeds_the_bound_agents_step(probe):
    """sac.py:104-108 with the REFERENCE's objects: ReplayMemory.sample() returns a DictArray, `.to_torch()` a DictArray (a GDict) whose
    `["obs"]` is a plain dict of tensors -- what `_fetcher` / `_to_static` / DrQ's `_augment` index.  Key set, dtypes and shapes are the ones
    DeviceReplay hands out (tests/test_aux_aug_acting_gpu.py compares those two containers' samples key by key on the GPU)."""
    for name, d in probe["extra"]["replay_seam"].items():
        assert d["sample_type"] == d["batch_type"] == "pyrl.utils.data.dict_array.DictArray" and d["is_gdict"], name
        assert d["replay_cfg_type"] == "ReplayMemory" and d["sampling_type"] == "OneStepTransition", name
        assert d["obs_value_type"] == "dict" and d["persistent"] is False, name
        st = d["structure"]
        assert set(st) == {"obs", "next_obs", "actions", "rewards", "dones"}
        assert st["obs"] == st["next_obs"] and st["obs"]["xyz"] == ["torch.float32", [8, 3, 96]] and st["obs"]["rgb"] == ["torch.uint8", [8, 3, 96]]
        assert st["rewards"] == ["torch.float32", [8, 1]] and st["dones"] == ["torch.bool", [8, 1]] and st["actions"][0] == "torch.float32"
        if "maniskill" in name:
            assert st["obs"]["seg"] == ["torch.bool", [8, 1, 96]] and st["obs"]["agent"] == ["torch.float32", [8, 10]]
        assert d["static_structure"] == st and d["static_buffers_reused"] and d["static_holds_the_second_batch"], name
        if name.startswith("drq"):
            assert d["augmented_type"] == "AugmentedObs" and set(d["augmented_keys"]) == set(st["obs"]), name


def test_the_seam_fixture_is_the_structure_the_probe_saw(probe):
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_seam_batch_drq_maniskill.npz"))
    st = probe["extra"]["replay_seam"]["drq_maniskill_pn_jitter"]["structure"]
    for k, v in st.items():
        for kk, (dt, shape) in (v.items() if isinstance(v, dict) else [(None, v)]):
            a = z[k if kk is None else f"{k}/{kk}"]
            assert f"torch.{a.dtype}" == dt and list(a.shape) == shape, (k, kk)


def test_train_rl_update_and_checkpoint_cadence_reaches_the_first_gpu_call(probe):
    """train_rl.py:292-296 / 392-405 replayed around the bound agent with the reference's own EveryNSteps and save_checkpoint: the update
    loop gets as far as this package's first GPU call and stops THERE, with the explicit no-CPU-path error (the container has no GPU);
    the checkpoint block (to_normal / save_checkpoint / recover_ddp) runs."""
    for name, d in probe["extra"]["train_rl"].items():
        assert d["total_updates_reached"] == 1 and "MI355X only" in d["update_error"], (name, d)
        assert d["checkpoint_steps"] == [5, 10] and d["checkpoint_written"] is True, name
