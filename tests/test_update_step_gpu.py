"""GPU parity of the whole update step: pointcloud_rl_amd SAC / DrQ agents (HIP encoder forward and
backward through the C ABI) vs golden vectors captured from the reference's update_parameters."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEP_FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "sac_*.npz")) + glob.glob(os.path.join(GOLDEN, "drq_*.npz")))


class Memory:
    """Stand-in for ReplayMemory: .sample(bs) -> object with .to_torch(device=, non_blocking=) (sac.py:104)."""

    def __init__(self, batch):
        self.batch = batch

    def sample(self, batch_size):
        return self

    def to_torch(self, device=None, non_blocking=False):
        from pointcloud_rl_amd.utils.torch_utils import to_torch
        return to_torch(self.batch, device=device, non_blocking=non_blocking)


def build_from_fixture(d, dev, fused=True):
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    B, N, A, S, n_updates = [int(x) for x in d["meta/dims"]]
    kind = str(d["meta/agent_type"])
    hidden = d["init/actor.backbone.final_mlp.mlp.linear0.weight"].shape[0]
    C = d["init/actor.backbone.visual_nn.conv.mlp.conv0.weight"].shape[1]
    conv = "init/actor.backbone.visual_nn.conv.mlp.conv%d.weight"
    nets = ([int(d[conv % i].shape[0]) for i in range(3)], int(d["init/actor.backbone.visual_nn.final_mlp.0.weight"].shape[0]))
    if kind == "SAC":
        extra = dict(use_episode_dones=True) if ("meta/use_episode_dones" in d.files and bool(d["meta/use_episode_dones"])) else {}
        cfg = configs.sac_dmc(C, A, B, hidden, nets=nets, **extra)
    elif S == 0:
        svea = "meta/svea" in d.files and bool(d["meta/svea"])
        cfg = configs.drq_dmc(C, A, B, hidden, num_aug=int(d["meta/hyper"][6]), svea=svea)
    else:
        cfg = configs.drq_maniskill(C, A, S, B, hidden)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    agent = build_agent(cfg)
    state = {k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("init/")}
    with torch.no_grad():
        for n, p in agent.named_parameters():
            p.copy_(state[n])
    agent.use_fused_step = fused
    return agent.to(dev), n_updates


def batch_of(d, u):
    pre = f"u{u}/batch/"
    batch = {"obs": {}, "next_obs": {}}
    for k in d.files:
        if k.startswith(pre):
            rest = k[len(pre):]
            if "/" in rest:
                side, key = rest.split("/")
                batch[side][key] = d[k]
            else:
                batch[rest] = d[k]
    return batch


def draws(d, u, prefix, dev):
    out, i = [], 0
    while f"u{u}/{prefix}{i}" in d.files:
        out.append(torch.from_numpy(d[f"u{u}/{prefix}{i}"]).to(dev))
        i += 1
    return out


@pytest.mark.parametrize("fused", [True, False], ids=["fused-kernels", "autograd-heads"])
@pytest.mark.parametrize("path", STEP_FIXTURES, ids=os.path.basename)
def test_update_parameters_matches_reference(cuda, path, fused):
    d = np.load(path)
    agent, n_updates = build_from_fixture(d, cuda, fused)
    for u in range(1, n_updates + 1):
        agent.actor.head.noise_override = draws(d, u, "eps", cuda)
        if hasattr(agent, "obs_aug") and agent.obs_aug is not None:
            agent.obs_aug[0].noise_override = draws(d, u, "jitter", cuda)
        ret = agent.update_parameters(Memory(batch_of(d, u)), u)
        # SVEA and the wide last layer (class-default mlp_spec [64, 128, 1024]): HIP encoder kernels + autograd heads only
        wide = agent.encoder.mlp_spec[-1] > 256
        assert (agent._fused is not None) == (fused and not getattr(agent, "svea", False) and not wide)
        assert not agent.actor.head.noise_override
        ref_keys = [k for k in d.files if k.startswith(f"u{u}/ret/")]
        assert {k.split("/", 1)[1] for k in ret} == {k[len(f"u{u}/ret/"):] for k in ref_keys}
        for k, v in ret.items():
            ref = float(d[f"u{u}/ret/{k.split('/', 1)[1]}"])
            assert abs(v - ref) <= 5e-5 * max(1.0, abs(ref)), (u, k, v, ref)
        # encoder argmax of the obs pass (the reference's passes 3/4 are the obs encodes)
        for name, p in agent.named_parameters():
            s = d[f"u{u}/paramsum/{name}"]
            got = np.array([float(p.detach().double().sum()), float(p.detach().double().abs().sum())])
            np.testing.assert_allclose(got, s, rtol=2e-5, atol=2e-5, err_msg=f"u{u} {name}")
        if u == 2:
            for name, p in agent.named_parameters():
                err = np.abs(p.detach().cpu().numpy() - d[f"u2/param/{name}"])
                # 1e-5 everywhere, except that Adam's first updates are lr * g / (|g| + 1e-8): an element whose
                # gradient is ~1e-8 turns a 1e-10 gradient difference into a ~1e-5 parameter difference.
                assert (err <= 1e-5).mean() >= 0.999 and err.max() <= 1e-4, (name, err.max(), (err > 1e-5).sum())


def test_agent_refuses_cpu_update():
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    cfg = configs.sac_dmc(6, 6, 4, 32)
    cfg["env_params"] = configs.env_params({"xyz": [3, 16], "rgb": [3, 16]}, 6)
    agent = build_agent(cfg)
    with pytest.raises(RuntimeError):
        agent.update_parameters(Memory({}), 1)


def test_graph_replay_matches_eager(cuda):
    """The hipGraph-replayed step and the eager step produce the same metrics and parameters when fed the
    same batches and the same device RNG state."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay

    def run(graphs):
        cfg = configs.sac_dmc(6, 6, 16, head_hidden=64)
        cfg["env_params"] = configs.env_params({"xyz": [3, 96], "rgb": [3, 96]}, 6)
        torch.manual_seed(0)
        agent = build_agent(cfg).to(cuda)
        if graphs:
            agent.enable_graphs(warmup=1)
        mem = SyntheticReplay(16, 96, 6, seed=5, device=cuda)
        torch.manual_seed(123)
        rets = [agent.update_parameters(mem, u) for u in range(1, 9)]
        return rets, {n: p.detach().cpu().clone() for n, p in agent.named_parameters()}, agent

    eager, p_eager, _ = run(False)
    graphed, p_graph, agent = run(True)
    assert len(agent._graphs) == 2          # one graph for critic-only steps, one for actor + target steps
    for a, b in zip(eager, graphed):
        assert a.keys() == b.keys()
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(a[k])), (k, a[k], b[k])
    for n in p_eager:
        assert torch.allclose(p_eager[n], p_graph[n], atol=1e-6, rtol=0), n


def test_drq_mixed_precision_step_tracks_the_fp32_step(cuda):
    """BASELINE config 3 (DrQ, jitter, bf16 encoder): same batch, same injected policy / jitter noise -- the bf16 agent's
    losses stay within 5e-2 (relative) of the fp32 agent's over a few updates; parameters: Adam moves an entry by at most lr
    per step whatever the gradient's size, so an entry whose tiny gradient changes sign under bf16 rounding can end up
    2 * lr * steps = 8e-3 away -- the bound on the worst entry -- while the mean distance stays below 3e-4.  The
    graph-replayed step runs."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    B, N, A, S = 8, 160, 5, 7
    agents = {}
    for dt in ("f32", "bf16"):
        cfg = configs.drq_maniskill(7, A, S, B, head_hidden=64, encoder_dtype=dt)
        cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N], "seg": [1, N], "agent": S}, A)
        torch.manual_seed(0)
        agents[dt] = build_agent(cfg).to(cuda)
    assert agents["bf16"].encoder.compute_dtype == "bf16" and agents["f32"].encoder.compute_dtype == "f32"
    mem = SyntheticReplay(B, N, A, seed=4, device=cuda, seg=1, agent=S)
    g = torch.Generator().manual_seed(7)
    rets = {"f32": [], "bf16": []}
    for u in range(1, 5):
        eps = [torch.randn(2 * B, A, generator=g), torch.randn(B, A, generator=g)]
        jit = [torch.empty(2 * B, 3, N).uniform_(-0.01, 0.01, generator=g) for _ in range(2)]
        for dt, agent in agents.items():
            agent.actor.head.noise_override = [e.to(cuda) for e in eps][:2 if u % 2 == 0 else 1]
            agent.obs_aug[0].noise_override = [j.to(cuda) for j in jit]
            rets[dt].append(agent.update_parameters(mem, u))
    for a, b in zip(rets["f32"], rets["bf16"]):
        for k in ("drq/critic_loss", "drq/q", "drq/q_target"):
            assert abs(a[k] - b[k]) <= 5e-2 * max(1.0, abs(a[k])), (k, a[k], b[k])
    pa, pb = dict(agents["f32"].named_parameters()), dict(agents["bf16"].named_parameters())
    diffs = torch.cat([(pa[n].detach() - pb[n].detach()).abs().reshape(-1) for n in pa])
    assert 0 < float(diffs.max()) <= 8.5e-3 and float(diffs.mean()) < 3e-4, (float(diffs.max()), float(diffs.mean()))
    bf = agents["bf16"]
    bf.enable_graphs(warmup=1)
    more = [bf.update_parameters(mem, u) for u in range(5, 11)]
    assert bf._graphs and all(np.isfinite(list(r.values())).all() for r in more)


@pytest.mark.parametrize("path", [p for p in STEP_FIXTURES if os.path.basename(p).startswith("drq_") and "svea" not in p], ids=os.path.basename)
def test_drq_mixed_precision_step_against_the_reference_fixture(cuda, path):
    """BASELINE config 3's arithmetic (bf16 conv1 / conv2 contractions, fp32 accumulation) on the DrQ fixtures captured from the fp32
    REFERENCE, same batches and injected policy / jitter noise: every loss / Q statistic the reference returned is met within
    1e-2 (relative, floor 1) and the gradient norms within 6 % over the fixture's updates -- measured worst: 2.8e-3 and 3.2 %."""
    d = np.load(path)
    agent, n_updates = build_from_fixture(d, cuda, fused=True)
    agent.encoder.compute_dtype = "bf16"
    worst, worst_g = 0.0, 0.0
    for u in range(1, n_updates + 1):
        agent.actor.head.noise_override = draws(d, u, "eps", cuda)
        agent.obs_aug[0].noise_override = draws(d, u, "jitter", cuda)
        ret = agent.update_parameters(Memory(batch_of(d, u)), u)
        for k, v in ret.items():
            ref = float(d[f"u{u}/ret/{k.split('/', 1)[1]}"])
            rel = abs(v - ref) / max(1.0, abs(ref))
            if k.endswith("_grad"):
                worst_g = max(worst_g, abs(v - ref) / max(abs(ref), 1e-6))
            else:
                worst = max(worst, rel)
    print(f"{os.path.basename(path)} bf16 encoder vs reference: worst metric {worst:.2e}, worst gradient norm {worst_g:.2e}")
    assert worst <= 1e-2 and worst_g <= 0.06, (worst, worst_g)


def test_checkpoint_resume_continues_bit_for_bit(cuda, tmp_path):
    """Train 4 steps, save in the reference's checkpoint format, load into a freshly built agent, train 4 more: parameters,
    Adam moments and step counts equal those of 8 uninterrupted steps exactly (same injected policy noise)."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    from pointcloud_rl_amd.utils.checkpoint import load_checkpoint, save_checkpoint
    B, N, A = 8, 96, 4

    def make(seed):
        cfg = configs.sac_dmc(6, A, B, head_hidden=64)
        cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
        torch.manual_seed(seed)
        return build_agent(cfg).to(cuda)

    mem = SyntheticReplay(B, N, A, seed=6, device=cuda)
    g = torch.Generator().manual_seed(11)
    eps = [[torch.randn(B, A, generator=g).to(cuda) for _ in range(2)] for _ in range(8)]

    def run(agent, lo, hi):
        for u in range(lo, hi + 1):
            agent.actor.head.noise_override = list(eps[u - 1][:2 if u % 2 == 0 else 1])
            agent.update_parameters(mem, u)

    whole = make(0)
    run(whole, 1, 8)
    first = make(0)
    run(first, 1, 4)
    path = str(tmp_path / "model_4.ckpt")
    save_checkpoint(first, path)
    resumed = make(123)                                     # different init: everything must come from the file
    load_checkpoint(resumed, path, map_location="cpu", strict=True)
    run(resumed, 5, 8)
    for (n, p), (_, q) in zip(whole.named_parameters(), resumed.named_parameters()):
        assert torch.equal(p, q), n
    for name in ("critic_optim", "actor_optim", "alpha_optim"):
        a, b = getattr(whole, name), getattr(resumed, name)
        assert int(a.step_counter) == int(b.step_counter) > 0
        assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq), name
    # the target network's own (non-shared) parameters travel too
    for (n, p), (_, q) in zip(whole.target_critic.named_parameters(), resumed.target_critic.named_parameters()):
        assert torch.equal(p, q), n


def test_reference_style_state_dict_keeps_the_fused_optimizers(cuda):
    """The reference's get_state_dict / load_state_dict walk `module.__dict__` and keep what passes
    `isinstance(child, torch.optim.Optimizer)` (checkpoint_utils.py:59-71, 226-229); train_rl.py:392-405 calls them on the agent
    AFTER updates, when the three torch.optim.Adam objects have been replaced by the fused optimizer.  The same walk, written out
    here (the reference tree does not travel to the GPU box), must find all three with their moments, in torch.optim.Adam's
    state_dict layout, and a genuine torch.optim.Adam must accept what they wrote."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    B, N, A = 8, 96, 4
    cfg = configs.sac_dmc(6, A, B, head_hidden=64)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    names = ("actor_optim", "critic_optim", "alpha_optim")
    assert all(type(getattr(agent, n)) is torch.optim.Adam for n in names)
    mem = SyntheticReplay(B, N, A, seed=6, device=cuda)
    for u in (1, 2, 3, 4):
        agent.update_parameters(mem, u)

    def reference_walk(module):      # checkpoint_utils.py:226-229
        found, seen = {}, []
        for name, child in module.__dict__.items():
            if child is not None and isinstance(child, torch.optim.Optimizer) and id(child) not in seen:
                seen.append(id(child))
                found[name] = child.state_dict()
        return found

    found = reference_walk(agent)
    assert sorted(found) == sorted(names), sorted(found)
    for n in names:
        opt, sd = getattr(agent, n), found[n]
        assert type(opt).__name__ == "HipAdam"
        assert len(sd["param_groups"]) == len(sd["state"]) == len(opt.param_groups) > 0
        steps = {float(st["step"]) for st in sd["state"].values()}
        assert steps == {4.0 if n == "critic_optim" else 2.0}, (n, steps)
        assert any(float(st["exp_avg"].abs().max()) > 0 for st in sd["state"].values()), n
        # a genuine torch.optim.Adam over the same parameters takes the dict, and hands the same moments back
        twin = torch.optim.Adam([dict(params=g["params"]) for g in opt.param_groups], lr=opt.param_groups[0]["lr"])
        twin.load_state_dict(sd)
        back = twin.state_dict()
        # (torch keeps the tensors it is handed when dtype and device already match: clone before zeroing the originals)
        back = dict(back, state={i: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in back["state"].items()})
        for i, st in sd["state"].items():
            assert torch.equal(back["state"][i]["exp_avg"], st["exp_avg"]) and torch.equal(back["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
        # and the fused optimizer takes a torch.optim.Adam's dict back (checkpoint_utils.py:62: child.load_state_dict(...))
        before = (opt.exp_avg.clone(), opt.exp_avg_sq.clone())
        opt.exp_avg.zero_(), opt.exp_avg_sq.zero_()
        opt.load_state_dict(back)
        assert torch.equal(opt.exp_avg, before[0]) and torch.equal(opt.exp_avg_sq, before[1]), n


@pytest.mark.parametrize("device_replay", [False, True], ids=["fixed-batch", "device-replay"])
def test_graph_replay_honours_a_changed_learning_rate(cuda, device_replay):
    """lr / betas / eps are kernel arguments of the fused Adam launch; graphs captured with the old values must be
    dropped when a scheduler (or load_state_dict) changes param_groups.  With lr = 0 nothing an optimizer owns may move.
    With a DeviceReplay the steady-state path of update_parameters (SAC._replay_fast) is the one that has to notice."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay, make_batch_np
    cfg = configs.sac_dmc(6, 6, 16, head_hidden=64)
    cfg["env_params"] = configs.env_params({"xyz": [3, 96], "rgb": [3, 96]}, 6)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    agent.enable_graphs(warmup=1)
    if device_replay:
        from pointcloud_rl_amd.replay import DeviceReplay
        mem = DeviceReplay(64, device=cuda, seed=5)
        mem.push_batch(make_batch_np(64, 96, 6, seed=5))
    else:
        mem = SyntheticReplay(16, 96, 6, seed=5, device=cuda)
    for u in range(1, 7):
        agent.update_parameters(mem, u)
    assert len(agent._graphs) == 2 and (agent._fast is not None) == device_replay
    for opt in (agent.critic_optim, agent.actor_optim, agent.alpha_optim):
        for g in opt.param_groups:
            g["lr"] = 0.0
    owned = {n: p.detach().clone() for n, p in list(agent.critic.named_parameters()) + list(agent.actor.named_parameters())}
    log_alpha = agent.log_alpha.detach().clone()
    for u in range(7, 13):
        agent.update_parameters(mem, u)
    assert len(agent._graphs) == 2          # re-captured with the new values
    now = dict(list(agent.critic.named_parameters()) + list(agent.actor.named_parameters()))
    for n, p in owned.items():
        assert torch.equal(p, now[n].detach()), n
    assert torch.equal(log_alpha, agent.log_alpha.detach())


def test_separate_backbones_repack_their_own_encoders(cuda):
    """shared_backbone=False (the SAC constructor's default): each Q head owns a PointNet that the critic optimizer
    updates through raw pointers.  After every optimizer step those encoders must run on the updated weights: a
    freshly built PointNet loaded with the current state_dict gives bit-identical features."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.networks import build_all
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    B, N, A = 8, 96, 4
    cfg = configs.sac_dmc(6, A, B, head_hidden=32)
    cfg["shared_backbone"] = False
    cfg["critic_cfg"]["nn_cfg"]["visual_nn_cfg"] = dict(cfg["actor_cfg"]["nn_cfg"]["visual_nn_cfg"])
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    encs = [v.backbone.visual_nn for v in agent.critic.values]
    assert encs[0] is not encs[1] and encs[0] is not agent.actor.backbone.visual_nn
    mem = SyntheticReplay(B, N, A, seed=3, device=cuda)
    w_before = [e.conv.mlp.conv1.weight.detach().clone() for e in encs]
    for u in range(1, 4):
        agent.update_parameters(mem, u)
    obs = {k: v for k, v in mem.batch["obs"].items() if k in ("xyz", "rgb")}
    for e, w0 in zip(encs, w_before):
        assert not torch.equal(e.conv.mlp.conv1.weight.detach(), w0), "the critic optimizer did not train this encoder"
        fresh = build_all(dict(cfg["actor_cfg"]["nn_cfg"]["visual_nn_cfg"])).to(cuda)
        fresh.load_state_dict(e.state_dict())
        with torch.no_grad():
            assert torch.equal(e(obs), fresh(obs))


@pytest.mark.parametrize("kind", ["sac", "drq"])
def test_round4_launch_cuts_equal_the_launches_they_replace(cuda, monkeypatch, kind):
    """Heads 1 024 wide (where the row-split tails, the backward tail, the first-layer fold and the riding column sums / temperature
    apply): four updates with every round-4 fusion switched OFF (the round-3 launch sequence) against the default, same injected
    noise.  The fusions only re-associate sums, so metrics agree to 2e-5 and parameters to 2e-6 (Adam's first steps: 1e-3 per step)."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    B, N, A = 8, 64, 6
    g = torch.Generator().manual_seed(3)
    rows = B * (2 if kind == "drq" else 1)
    eps = [[torch.randn(rows, A, generator=g), torch.randn(B, A, generator=g)] for _ in range(4)]
    jit = [[torch.empty(rows, 3, N).uniform_(-0.01, 0.01, generator=g) for _ in range(2)] for _ in range(4)]

    def run(off):
        cfg = configs.drq_dmc(6, A, B, head_hidden=1024) if kind == "drq" else configs.sac_dmc(6, A, B, head_hidden=1024)
        cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
        torch.manual_seed(0)
        agent = build_agent(cfg).to(cuda)
        agent._prepare()
        agent._fused.tail_bwd = agent._fused.fold_q0 = agent._fused.attach_colsum = not off
        mem = SyntheticReplay(B, N, A, seed=4, device=cuda)
        rets = []
        for u in range(1, 5):
            agent.actor.head.noise_override = [e.to(cuda) for e in eps[u - 1][:2 if u % 2 == 0 else 1]]
            if kind == "drq":
                agent.obs_aug[0].noise_override = [j.to(cuda) for j in jit[u - 1]]
            rets.append(agent.update_parameters(mem, u))
        f = agent._fused
        assert f is not None and f.tail_bwd == (not off) and f.fold_q0 == (not off)
        assert f._fold_fits(rows, 2 if kind == "drq" else 1) == (not off)
        return agent, rets

    base, rets_b = run(off=True)
    new, rets_n = run(off=False)
    for rb, rn in zip(rets_b, rets_n):
        assert rb.keys() == rn.keys()
        for k in rb:
            assert abs(rb[k] - rn[k]) <= 2e-5 * max(1.0, abs(rb[k])), (k, rb[k], rn[k])
    for (n, p), (_, q) in zip(base.named_parameters(), new.named_parameters()):
        err = (p - q).abs()
        assert (err <= 2e-6).float().mean() >= 0.999 and err.max() <= 2.1e-3, (n, float(err.max()))


@pytest.mark.parametrize("switch", ["publish_first", "ln_rider", "gather_rider"])
@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "hipgraph"])
def test_metrics_published_before_the_last_optimizer_pass_change_nothing(cuda, graphs, switch):
    """Round 6, two re-orderings of the step's launches that must change NOTHING.  publish_first: the step's last optimizer pass runs BEHIND
    the launch that publishes the metrics (csrc/optim.hip: gradnorm_kernel takes the pass's norm ahead of it, the temperature's pass rides
    there) -- same partial sums, same reduction tree, same bias corrections.  ln_rider: the feature LayerNorm's backward runs as extra
    workgroups of the encoder backward's prep launch, brought forward in front of the feature GEMMs (pcrl_encoder_bwd_attach_ln_bwd) -- the
    same block function.  gather_rider: the metrics are gathered and published by the first workgroup of the published pass
    (pcrl_adam_step_published_gather_f32) instead of a launch in front of it.  Every metric of six updates (critic-only and actor steps alternate) and every parameter, moment and the target
    network are bit-identical to the order of rounds 1-5, eager and replayed from the step's hipGraph."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    B, N, A = 8, 64, 6
    g = torch.Generator().manual_seed(5)
    eps = [[torch.randn(B, A, generator=g), torch.randn(B, A, generator=g)] for _ in range(6)]

    def run(first):
        cfg = configs.sac_dmc(6, A, B, head_hidden=1024)
        cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
        torch.manual_seed(0)
        agent = build_agent(cfg).to(cuda)
        agent._prepare()
        setattr(agent._fused, switch, first)
        if graphs:
            agent.enable_graphs()
        mem = SyntheticReplay(B, N, A, seed=4, device=cuda)
        rets = []
        for u in range(1, 7):
            if not graphs:
                agent.actor.head.noise_override = [e.to(cuda) for e in eps[u - 1][:2 if u % 2 == 0 else 1]]
            rets.append(agent.update_parameters(mem, u))
        torch.cuda.synchronize()
        return agent, rets

    torch.manual_seed(11)
    old, rets_o = run(False)
    torch.manual_seed(11)
    new, rets_n = run(True)
    assert getattr(new._fused, switch) and not getattr(old._fused, switch)
    for u, (ro, rn) in enumerate(zip(rets_o, rets_n)):
        assert ro.keys() == rn.keys()
        for k in ro:
            assert ro[k] == rn[k] or (ro[k] != ro[k] and rn[k] != rn[k]), (u, k, ro[k], rn[k])
    for (n, p), (_, q) in zip(old.named_parameters(), new.named_parameters()):
        assert torch.equal(p, q), n
    for (n, p), (_, q) in zip(old.target_critic.named_parameters(), new.target_critic.named_parameters()):
        assert torch.equal(p, q), n
    for name in ("critic", "actor", "alpha"):
        a, b = getattr(old, f"{name}_optim"), getattr(new, f"{name}_optim")
        assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq) and torch.equal(a.step_counter, b.step_counter), name
        assert int(a.step_counter.item()) == (6 if name == "critic" else 3)


def test_update_on_the_batch_the_reference_replay_container_produced(cuda):
    """tests/golden/ref_seam_batch_drq_maniskill.npz (tools/gen_golden_seam.py) is what the reference's ReplayMemory -> GDict.to_torch ->
    `agent._fetcher` handed to the step in the build container (configs/mfrl/drq/maniskill/pn_jitter.py: xyz f32, rgb u8, seg bool,
    agent f32, episode_dones next to dones).  The same structure drives update_parameters here -- eagerly and replayed from a hipGraph --
    and a DeviceReplay fed with the same transitions returns a batch of the same keys, dtypes and shapes."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.replay import DeviceReplay
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_seam_batch_drq_maniskill.npz"))
    batch = {}
    for k in z.files:
        node, parts = batch, k.split("/")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = z[k]
    B, N, A, S = 8, 96, 8, 10
    cfg = configs.drq_maniskill(7, A, S, B)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    rets = [agent.update_parameters(Memory(batch), u) for u in (1, 2)]
    assert agent._fused is not None and all(np.isfinite(list(r.values())).all() for r in rets)
    agent.enable_graphs(warmup=1)
    rets += [agent.update_parameters(Memory(batch), u) for u in range(3, 9)]
    assert agent._graphs and all(np.isfinite(list(r.values())).all() for r in rets)
    mem = DeviceReplay(16, device=cuda, seed=1)
    mem.push_batch(batch)
    got = mem.sample(B).to_torch(device=cuda)
    for k in ("obs", "next_obs", "actions", "rewards", "dones", "episode_dones"):
        want = batch[k]
        for kk, vv in (want.items() if isinstance(want, dict) else [(None, want)]):
            t = got[k][kk] if kk is not None else got[k]
            assert str(t.dtype) == f"torch.{vv.dtype}" and tuple(t.shape) == vv.shape, (k, kk, t.dtype, vv.dtype)
