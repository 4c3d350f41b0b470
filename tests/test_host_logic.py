"""CPU tests of the host side: registries, configs, parameter naming, optimizer construction, Polyak
rules, augmentations' parameter draws, flat buffers.  No kernel is launched here."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from pointcloud_rl_amd import configs
from pointcloud_rl_amd.augmentations import GlobalRotScaleTrans, RandomJitterPoints, build_data_augmentations
from pointcloud_rl_amd.methods import MFRL, build_agent
from pointcloud_rl_amd.methods.drq import first_augmentation, repeat_obs
from pointcloud_rl_amd.methods.sac import FlatBuffer
from pointcloud_rl_amd.networks import NETWORK, build_all
from pointcloud_rl_amd.networks.pointnet import AugmentedObs
from pointcloud_rl_amd.networks.visuomotor import Visuomotor
from pointcloud_rl_amd.utils.registry import ConfigDict, Registry, build_from_cfg
from pointcloud_rl_amd.utils.torch_utils import build_optimizer, hard_update, select_optimizer_params, soft_update

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_agent(kind="sac", hidden=32, A=6):
    cfg = configs.sac_dmc(6, A, 4, hidden) if kind == "sac" else configs.drq_maniskill(7, A, 10, 4, hidden)
    cfg["env_params"] = configs.env_params({"xyz": [3, 16], "rgb": [3, 16]}, A)
    torch.manual_seed(0)
    return build_agent(cfg)


def test_registry_contract():
    reg = Registry("things")

    @reg.register_module()
    class Foo:
        def __init__(self, a, b=2):
            self.a, self.b = a, b
    with pytest.raises(KeyError):
        reg.register_module()(Foo)                       # duplicate name
    reg.register_module(force=True)(Foo)
    assert "Foo" in reg and reg.get("Bar") is None and len(reg) == 1
    obj = build_from_cfg(dict(type="Foo", a=1), reg, default_args=dict(b=5))
    assert (obj.a, obj.b) == (1, 5)
    with pytest.raises(KeyError):
        build_from_cfg(dict(a=1), reg)
    with pytest.raises(KeyError):
        build_from_cfg(dict(type="Nope"), reg)
    with pytest.raises(TypeError):
        build_from_cfg([1], reg)
    assert build_from_cfg(None, reg) is None
    assert {"SAC", "DrQ"} <= set(MFRL.module_dict) and {"PointNet", "Visuomotor", "LinearMLP"} <= set(NETWORK.module_dict)
    with pytest.raises(RuntimeError):
        build_all(dict(type="NoSuchNet"))


def test_config_dict():
    c = ConfigDict(a=1, b=dict(c=[dict(d=2)]))
    assert c.b.c[0].d == 2
    c.e = dict(f=3)
    assert isinstance(c.e, ConfigDict) and c["e"]["f"] == 3
    with pytest.raises(AttributeError):
        c.missing


@pytest.mark.parametrize("fixture,kind", [("sac_dmc_small", "sac"), ("drq_maniskill_jitter_small", "drq")])
def test_parameter_names_match_reference_state_dict(fixture, kind):
    d = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    ref = {k[5:]: tuple(d[k].shape) for k in d.files if k.startswith("init/")}
    A = int(d["meta/dims"][2])
    hidden = d["init/actor.backbone.final_mlp.mlp.linear0.weight"].shape[0]
    agent = make_agent(kind, hidden, A)
    mine = {n: tuple(p.shape) for n, p in agent.named_parameters()}
    assert list(mine.keys()) == list(ref.keys())           # same names, same order
    assert mine == ref


def test_shared_backbone_topology_and_optimizers():
    agent = make_agent()
    enc = agent.actor.backbone.visual_nn
    assert all(v.backbone.visual_nn is enc for v in agent.critic.values)
    assert all(v.backbone.visual_nn is enc for v in agent.target_critic.values)
    assert agent._encoder_is_shared()
    # target Q heads: own parameters, frozen, hard-copied; the shared encoder stays trainable
    for tv, v in zip(agent.target_critic.values, agent.critic.values):
        for (n, tp), (_, p) in zip(tv.backbone.final_mlp.named_parameters(), v.backbone.final_mlp.named_parameters()):
            assert tp is not p and torch.equal(tp, p) and not tp.requires_grad
    assert all(p.requires_grad for p in enc.parameters())
    # one param group per tensor; the actor's optimizer excludes the encoder (param_cfg regex -> None)
    assert len(agent.critic_optim.param_groups) == 24 and len(agent.actor_optim.param_groups) == 6
    names = [n for n, _ in select_optimizer_params(agent.actor, {"(.*?)visual_nn(.*?)": None})]
    assert names and not any("visual_nn" in n for n in names)
    assert agent.target_entropy == -6.0 and abs(agent.alpha - 0.1) < 1e-7


def test_soft_and_hard_update_rules():
    agent = make_agent()
    enc_before = {n: p.clone() for n, p in agent.actor.backbone.visual_nn.named_parameters()}
    with torch.no_grad():
        for p in agent.critic.parameters():
            p.add_(1.0)
    tgt_before = [p.clone() for p in agent.target_critic.values[0].backbone.final_mlp.parameters()]
    soft_update(agent.target_critic, agent.critic, {"default": 0.25, "(.*?)visual_nn(.*?)": 0.9})
    for tb, tp, p in zip(tgt_before, agent.target_critic.values[0].backbone.final_mlp.parameters(),
                         agent.critic.values[0].backbone.final_mlp.parameters()):
        assert torch.allclose(tp, tb * 0.75 + p * 0.25)
    # shared parameters are the same object in both networks -> untouched by the update
    for n, p in agent.actor.backbone.visual_nn.named_parameters():
        assert torch.equal(p, enc_before[n] + 1.0)
    soft_update(agent.target_critic, agent.critic, 1.0)
    hard_update(agent.target_critic, agent.critic)
    for tp, p in zip(agent.target_critic.parameters(), agent.critic.parameters()):
        assert torch.equal(tp, p)
    with pytest.raises(AssertionError):
        soft_update(agent.target_critic, agent.critic, {"x": 0.1})


def test_build_optimizer_variants():
    lin = nn.Linear(3, 2)
    opt = build_optimizer(lin, dict(type="Adam", lr=0.5, betas=(0.5, 0.9)))
    assert isinstance(opt, torch.optim.Adam) and len(opt.param_groups) == 2 and opt.param_groups[0]["lr"] == 0.5
    p = nn.Parameter(torch.ones(1))
    opt = build_optimizer(p, dict(type="Adam", lr=1e-3))
    assert opt.param_groups[0]["params"][0] is p
    with pytest.raises(NotImplementedError):
        build_optimizer(lin, dict(type="Adam", constructor="other"))


def test_flat_buffer_alignment_and_views():
    params = [("a", nn.Parameter(torch.arange(5.0))), ("b", nn.Parameter(torch.arange(6.0).reshape(2, 3))), ("c", nn.Parameter(torch.ones(4)))]
    fb = FlatBuffer(params)
    assert fb.offsets == [0, 8, 16] and fb.total == 20
    assert all(o % 4 == 0 for o in fb.offsets)
    assert torch.equal(params[1][1].data, torch.arange(6.0).reshape(2, 3))
    params[1][1].data.mul_(2)
    assert torch.equal(fb.data[8:14], torch.arange(6.0) * 2)       # the parameter is a view of the buffer
    params[0][1].grad.add_(1)
    assert fb.grad[:5].sum() == 5 and fb.grad[5:8].sum() == 0      # padding stays zero
    fb.zero_grad()
    assert fb.grad.abs().sum() == 0 and params[0][1].grad.data_ptr() == fb.grad.data_ptr()


def test_unsupported_configs_fail_loudly():
    with pytest.raises(NotImplementedError):
        build_all(dict(type="PointNet", feat_dim=6, mlp_spec=[64, 128, 256], feature_transform=[1], ignore_first_ln=True))
    with pytest.raises(NotImplementedError):
        build_all(dict(type="PointNet", feat_dim=6, mlp_spec=[64, 128, 256], feature_transform=[], ignore_first_ln=False))
    net = build_all(dict(type="PointNet", feat_dim=6, mlp_spec=[64, 128, 256], out_channels=50, feature_transform=[], ignore_first_ln=True))
    with pytest.raises(RuntimeError):                      # no CPU implementation of the encoder
        net({"xyz": torch.zeros(1, 3, 8), "rgb": torch.zeros(1, 3, 8, dtype=torch.uint8)})


def test_visuomotor_split_obs():
    obs = {"xyz": 1, "rgb": 2, "seg": 3, "agent": 4, "inst_box": 5, "target_seg": 6, "visual_state": 7}
    visual, state = Visuomotor.split_obs(obs)
    assert set(visual) == {"xyz", "rgb", "seg"} and state == 4 and "agent" in obs     # caller's dict untouched
    with pytest.raises(AssertionError):
        Visuomotor.split_obs({"xyz": 1, "agent": 2, "state": 3})


def test_jitter_augmentation_spec_and_override():
    aug = build_data_augmentations(dict(type="RandomJitterPoints", main_key="xyz", req_keys=["xyz"], jitter_range=[-0.01, 0.01], seed=7))
    obs = {"xyz": torch.zeros(4, 3, 8), "rgb": torch.zeros(4, 3, 8, dtype=torch.uint8)}
    out = aug(obs)
    assert isinstance(out, AugmentedObs) and out["xyz"] is obs["xyz"]            # nothing materialised
    assert out.aug["jitter_range"] == [-0.01, 0.01] and out.aug["seed"] == 7
    out2 = aug(obs)
    assert int(out2.aug["offset_tensor"]) == int(out.aug["offset_tensor"]) + 1   # fresh Philox stream per call
    noise = torch.randn(4, 3, 8)
    aug[0].noise_override.append(noise)
    assert torch.equal(aug(obs).aug["jitter_noise"], noise)
    with pytest.raises(NotImplementedError):
        RandomJitterPoints(main_key="obs/pointcloud/xyz", req_keys=["obs/pointcloud/xyz"])
    # inside an update step: the step's shared draw counter is the offset of every call, the calls differ by seed, no counter of
    # their own is advanced; outside again: one counter step per call
    before = int(aug(obs).aug["offset_tensor"])
    shared = torch.tensor([41], dtype=torch.int64)
    aug[0].begin_step(shared)
    a, b = aug(obs), aug(obs)
    assert a.aug["offset_tensor"] is shared and b.aug["offset_tensor"] is shared
    assert len({a.aug["seed"], b.aug["seed"], 7}) == 3
    aug[0].begin_step(shared)
    assert aug(obs).aug["seed"] == a.aug["seed"]                                 # slots restart with every step (captured descs stay valid)
    aug[0].begin_step(None)
    assert int(aug(obs).aug["offset_tensor"]) == before + 1


def test_global_rot_scale_trans_matrix_semantics():
    torch.manual_seed(0)
    t = GlobalRotScaleTrans(main_key="xyz", req_keys=["xyz"], rot_range=[-0.15, 0.15], rot_axis="z",
                            scale_ratio_range=[0.9, 1.1], translation_range=[0.04, 0.0, 0.04], shift_height=False)
    m = t.sample_matrix(5, torch.device("cpu"))
    assert m.shape == (5, 3, 4)
    assert (m[-1, :, 3] == 0).all() and (m[:-1, 0, 3] != 0).any()       # reference quirk: delta_xyz[-1] = 0 zeroes the LAST cloud
    assert (m[:, 1, 3] == 0).all()                                      # translation_range[1] == 0
    scale = torch.linalg.norm(m[:, :, :3], dim=2)                       # rows of R scaled per axis
    assert ((scale > 0.89) & (scale < 1.11)).all() and (m[:, 2, 0].abs() < 1e-7).all()
    # rot_range=None: the reference skips the matrix product, so scale has no effect either
    t2 = GlobalRotScaleTrans(main_key="xyz", req_keys=["xyz"], rot_range=None, scale_ratio_range=[0.5, 0.6],
                             translation_range=[0.1, 0.1, 0.1], shift_height=True)
    m2 = t2.sample_matrix(3, torch.device("cpu"))
    assert torch.equal(m2[:, :, :3], torch.eye(3).expand(3, 3, 3)) and (m2[:, :, 3] != 0).all()


def test_drq_repeat_and_first_augmentation():
    obs = AugmentedObs({"xyz": torch.arange(2 * 3 * 4.0).reshape(2, 3, 4), "agent": torch.arange(4.0).reshape(2, 2)})
    rep = repeat_obs(obs, 3)
    assert rep["xyz"].shape == (6, 3, 4) and torch.equal(rep["xyz"][0], rep["xyz"][2]) and torch.equal(rep["xyz"][3], obs["xyz"][1])
    aug = AugmentedObs(rep)
    aug.aug = dict(jitter_range=[-1, 1], seed=1, offset=0)
    first = first_augmentation(aug, 2, 3)
    assert first["xyz"].shape == (2, 3, 4) and torch.equal(first["xyz"], obs["xyz"]) and first["xyz"].stride(0) == 3 * 12
    assert first.aug["row_mul"] == 3 and first.aug["row_add"] == 0     # cloud b uses noise row 3b: what the critic saw


def test_tanh_gaussian_head_matches_restatement():
    from oracle import torch_ref
    agent = make_agent()
    head = agent.actor.head
    feat, eps = torch.randn(5, 12), torch.randn(5, 6)
    head.noise_override.append(eps)
    a, nlp = head(feat, mode="max-entropy")
    a_ref, nlp_ref = torch_ref.tanh_gaussian(feat, eps, head.scale, head.bias)
    assert torch.allclose(a, a_ref) and torch.allclose(nlp, nlp_ref)
    assert torch.allclose(head(feat, mode="eval"), torch.tanh(feat[:, :6]) * head.scale + head.bias)
    assert head(feat, mode="explore").shape == (5, 6)


def test_checkpoint_roundtrip_in_the_reference_format(tmp_path):
    """save_checkpoint / load_checkpoint (pyrl/utils/torch/checkpoint_utils.py:25-96,148-179,215-269): optimizers travel under
    their attribute names inside state_dict, tensors are on the CPU, loading is non-strict and adapts a one-dimension mismatch."""
    import torch
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from collections import OrderedDict
    from pointcloud_rl_amd.utils.checkpoint import get_state_dict, load_checkpoint, load_state_dict, save_checkpoint

    def make(seed, C=6):
        cfg = configs.sac_dmc(C, 4, 8, head_hidden=32)
        cfg["env_params"] = configs.env_params({"xyz": [3, 16], "rgb": [3, 16]}, 4)
        torch.manual_seed(seed)
        return build_agent(cfg)

    a = make(0)
    for p in a.actor.parameters():                          # give the optimizers some state
        p.grad = torch.randn_like(p)
    a.actor_optim.step()
    sd = get_state_dict(a)
    assert {"actor_optim", "critic_optim", "alpha_optim", "log_alpha"} <= set(sd)
    assert "actor.backbone.visual_nn.conv.mlp.conv0.weight" in sd and sd["actor_optim"]["state"]
    path = tmp_path / "ckpt" / "model_1.ckpt"
    save_checkpoint(a, str(path), meta=dict(step=1))
    raw = torch.load(str(path), weights_only=False)
    assert set(raw) == {"meta", "state_dict"} and raw["meta"] == dict(step=1)
    assert all(v.device.type == "cpu" for v in raw["state_dict"].values() if torch.is_tensor(v))
    b = make(1)
    load_checkpoint(b, str(path), map_location="cpu", strict=True)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), n
    sa, sb = a.actor_optim.state_dict(), b.actor_optim.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    assert all(torch.equal(sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"]) for k in sa["state"])
    # "module." prefix of a DDP-wrapped save, and a bare state_dict file
    torch.save(OrderedDict(("module." + k, v) for k, v in raw["state_dict"].items() if torch.is_tensor(v)), str(tmp_path / "ddp.ckpt"))
    c = make(2)
    load_checkpoint(c, str(tmp_path / "ddp.ckpt"), map_location="cpu", logger=None)
    assert torch.equal(dict(c.named_parameters())["log_alpha"], dict(a.named_parameters())["log_alpha"])
    # a network with more input channels takes the common part of conv0.weight (the reference's one-dimension adaptation)
    d = make(3, C=9)
    msgs = []
    class L:                                               # noqa: E306
        warning = staticmethod(msgs.append)
        info = staticmethod(msgs.append)
    load_state_dict(d, {k: v for k, v in raw["state_dict"].items() if torch.is_tensor(v)}, strict=False, logger=L)
    w_new = dict(d.named_parameters())["actor.backbone.visual_nn.conv.mlp.conv0.weight"]
    w_old = dict(a.named_parameters())["actor.backbone.visual_nn.conv.mlp.conv0.weight"]
    assert w_new.shape[1] == 9 and torch.equal(w_new[:, :6], w_old) and any("adapt weight" in m for m in msgs)
    with pytest.raises(RuntimeError):
        load_state_dict(make(4), {"nonsense": torch.zeros(1)}, strict=True)


def test_reference_written_checkpoint_loads_and_key_sets_match(tmp_path):
    """tests/golden/ref_sac_dmc_small.ckpt was written by the reference's own save_checkpoint on a reference SAC agent after two
    reference updates (tools/gen_golden_checkpoint.py).  It loads strictly into this package's agent -- every parameter and every
    optimizer's Adam state -- and a checkpoint written here has exactly the reference file's keys."""
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.utils.checkpoint import load_checkpoint, save_checkpoint
    path = os.path.join(os.path.dirname(__file__), "golden", "ref_sac_dmc_small.ckpt")
    cfg = configs.sac_dmc(6, 6, 8, head_hidden=64)
    cfg["env_params"] = configs.env_params({"xyz": [3, 64], "rgb": [3, 64]}, 6)
    torch.manual_seed(9)
    agent = build_agent(cfg)
    ck = load_checkpoint(agent, path, map_location="cpu", strict=True)
    ref = ck["state_dict"]
    assert ck["meta"] == dict(updates=2)
    named = dict(agent.named_parameters(remove_duplicate=False))
    tensors = {k: v for k, v in ref.items() if torch.is_tensor(v)}
    assert set(tensors) == set(named) | set(dict(agent.named_buffers(remove_duplicate=False)))
    for k, v in tensors.items():
        assert torch.equal(named[k].detach(), v), k
    for name in ("actor_optim", "critic_optim", "alpha_optim"):
        mine, theirs = getattr(agent, name).state_dict(), ref[name]
        assert mine["state"].keys() == theirs["state"].keys() and len(mine["param_groups"]) == len(theirs["param_groups"])
        for i, st in theirs["state"].items():
            assert torch.equal(mine["state"][i]["exp_avg"], st["exp_avg"]) and torch.equal(mine["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
            assert float(mine["state"][i]["step"]) == float(st["step"])
    out = str(tmp_path / "mine.ckpt")
    save_checkpoint(agent, out, meta=dict(updates=2))
    mine = torch.load(out, weights_only=False)
    assert set(mine) == set(ck) and list(mine["state_dict"].keys()) == list(ref.keys())


@pytest.mark.parametrize("tag,kw", [
    ("rot_scale_trans", dict(rot_range=[-0.15, 0.15], rot_axis="z", scale_ratio_range=[0.9, 1.1], translation_range=[0.04, 0.0, 0.04], shift_height=False)),
    ("rot_y_only", dict(rot_range=0.5, rot_axis="y", scale_ratio_range=None, translation_range=None, shift_height=False)),
    ("shift_only", dict(rot_range=None, rot_axis="z", scale_ratio_range=None, translation_range=[0.1, 0.2, 0.3], shift_height=True)),
])
def test_global_rot_scale_trans_draws_the_reference_matrices(tag, kw):
    """tests/golden/ref_rotscaletrans.npz: matrices and outputs of the reference's GlobalRotScaleTrans under torch.manual_seed
    (tools/gen_golden_rotscaletrans.py).  Same seed -> the same draws in the same order -> the same [R|t]; applying it as the
    encoder kernel does (R x + t) gives the reference's xyz."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_rotscaletrans.npz"))
    xyz = torch.from_numpy(z["in/xyz"])
    torch.manual_seed(int(z[f"{tag}/seed"]))
    aug = GlobalRotScaleTrans(main_key="xyz", req_keys=["xyz"], **kw)
    out = aug({"xyz": xyz})
    mat = out.aug["affine"]
    ref = torch.from_numpy(z[f"{tag}/mat"])
    if kw["rot_range"] is not None:
        assert torch.equal(mat[:, :, :3], ref[:, :3, :3])
    if kw["translation_range"] is not None:
        assert torch.equal(mat[:, :, 3], ref[:, :3, 3])
    got = torch.einsum("bji,bin->bjn", mat[:, :, :3], xyz) + mat[:, :, 3:]
    np.testing.assert_allclose(got.numpy(), z[f"{tag}/out_xyz"], atol=1e-6, rtol=0)


def test_tstep_index_blocks_reproduce_the_reference_sampler():
    """TStepIndex + the host draw (pointcloud_rl_amd/replay.py) against what the reference's ReplayMemory(sampling_cfg=dict(type=
    "TStepTransition", ...)) returned (tests/golden/ref_replay_tstep.npz, tools/gen_golden_replay_tstep.py): same [B, H] blocks of
    ring positions (checked through the rewards they select from a numpy copy of the ring) and the same validity mask, for
    horizon 3 with / without replacement and for whole episodes (horizon -1, padded)."""
    import os
    from pointcloud_rl_amd.replay import DeviceReplay, TStepIndex
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_replay_tstep.npz"))
    cap = 48
    for tag, horizon, with_repl in (("h3_with", 3, True), ("h3_without", 3, False), ("episode_with", -1, True)):
        mem = DeviceReplay.__new__(DeviceReplay)                 # host-side state only: no device is touched
        mem.capacity, mem.with_replacement, mem.np_random = cap, with_repl, np.random.RandomState(11)
        mem.items, mem.item_index, mem.need_update, mem.running_count = None, 0, False, 0
        mem.tstep = TStepIndex(cap, horizon)
        ring = np.zeros((cap, 1), np.float32)
        pos = 0
        for i in range(4):
            r, ed, wi = d[f"push{i}/rewards"], d[f"push{i}/episode_dones"], d[f"push{i}/worker_indices"]
            for k in range(len(r)):
                ring[pos] = r[k]
                pos = (pos + 1) % cap
                mem.tstep.push(int(wi[k, 0]), bool(ed[k, 0]))
            mem.running_count += len(r)
            mem.need_update = True
        assert [min(mem.running_count, cap), len(mem.tstep)] == d[f"{tag}/len_units"].tolist()
        for s_ in range(6):
            index = mem.sample_indices(5, True, True, capacity=len(mem.tstep))
            blocks, mask = mem.tstep.blocks(index)
            want = d[f"{tag}/sample{s_}/rewards"]
            assert blocks.shape == want.shape[:2] and np.array_equal(mask, d[f"{tag}/sample{s_}/is_valid"]), (tag, s_)
            assert np.array_equal(ring[blocks], want), (tag, s_)



def test_bench_counts_the_backward_s_active_points_and_tiles_from_the_step_s_own_tensors():
    """bench.py::active_points_per_cloud (the `step` object's active_points_per_cloud / backward_tiles_per_rank): distinct argmax positions of
    the LIVE channels (pooled > 0) per cloud, and the 32-point tiles they make -- what the backward's prep launch builds its lists from."""
    import types
    import bench
    am = torch.tensor([[0, 0, 5, 5, 9, 40, 40, 41],          # cloud 0: positions {0, 5, 9, 40, 41}; channel 1 (position 0) dead, channel 4 (9) dead
                       [3, 3, 3, 3, 3, 3, 3, 3],             # cloud 1: one point
                       [0, 1, 2, 3, 4, 5, 6, 7]], dtype=torch.int32)           # cloud 2: eight points, all channels dead but two
    pooled = torch.tensor([[1.0, 0.0, 2.0, 0.5, 0.0, 1.0, 1.0, 3.0],
                           [1.0] * 8,
                           [0.0, 0.0, 0.0, 4.0, 0.0, 0.0, 0.0, 0.1]])
    agent = types.SimpleNamespace(_fused=types.SimpleNamespace(last_argmax=am, last_pooled=pooled))
    mean, tiles = bench.active_points_per_cloud(agent, tiles=True)
    # live positions: cloud 0 -> {0, 5, 40, 41} = 4, cloud 1 -> {3} = 1, cloud 2 -> {3, 7} = 2
    assert abs(mean - (4 + 1 + 2) / 3) < 1e-6 and tiles == 3
    assert abs(bench.active_points_per_cloud(agent) - mean) < 1e-9
    agent._fused.last_pooled = None                           # without the pooled values: every channel counts
    assert abs(bench.active_points_per_cloud(agent) - (5 + 1 + 8) / 3) < 1e-6
    big = torch.arange(40, dtype=torch.int32).repeat(2, 1)    # 40 distinct points per cloud -> two tiles each
    agent = types.SimpleNamespace(_fused=types.SimpleNamespace(last_argmax=big, last_pooled=torch.ones(2, 40)))
    assert bench.active_points_per_cloud(agent, tiles=True) == (40.0, 4)
    assert bench.active_points_per_cloud(types.SimpleNamespace(_fused=None), tiles=True) == (None, None)
    s = bench.step_roofline(bench.WORKLOADS["k1"], 6, [64, 128, 256], 256, 1, 6, 0, 0.81, False, feat=50, active_pts=118.0, bwd_tiles=1075)
    assert 0.45 < s["frac"] < 0.6 and s["frac"] <= 1.0 and 0.78 < s["useful_frac"] < 0.88 and s["backward_tiles_per_rank"] == 1075
