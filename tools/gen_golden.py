"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only).

    python tools/gen_golden.py            # writes tests/golden/{encoder_*,sac_*,drq_*}.npz

The reference Python (/root/reference) is imported unmodified through tools/ref_stubs.py and
executed on CPU (torch 2.10).  Captured: inputs, initial weights, every random draw the
reference makes (actor epsilon, jitter noise), encoder outputs including the argmax indices,
the metrics dict `update_parameters` returns, gradients just before each optimizer step and
the parameters afterwards.  The fixtures are data only; neither this script nor the reference
is needed (or present) when the tests run.
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = ref_stubs.REFERENCE_ROOT


def np_(t):
    return t.detach().cpu().numpy().copy()


def make_obs(g, B, N, pos_encoding=0, seg=0, agent=0):
    obs = {"xyz": g.randn(B, 3, N).astype(np.float32), "rgb": g.randint(0, 256, (B, 3, N)).astype(np.uint8)}
    if pos_encoding:
        pe = np.zeros((B, pos_encoding, N), np.uint8)
        per = N // pos_encoding
        for f in range(pos_encoding):
            pe[:, f, f * per:(f + 1) * per] = 1
        obs["pos_encoding"] = pe
    if seg:
        obs["seg"] = g.rand(B, seg, N) < 0.3
    if agent:
        obs["agent"] = g.randn(B, agent).astype(np.float32)
    return obs


# ----------------------------------------------------------------------------------------------
# A. encoder fixtures: reference PointNet.forward on small clouds
# ----------------------------------------------------------------------------------------------
def gen_encoder(name, C_extra, mlp_spec, out_channels, B, N, seed):
    from pyrl.networks.builder import build_all
    g = np.random.RandomState(seed)
    obs = make_obs(g, B, N, **C_extra)
    C = sum(v.shape[1] for k, v in obs.items())
    torch.manual_seed(seed)
    net = build_all(dict(type="PointNet", feat_dim=C, mlp_spec=list(mlp_spec), out_channels=out_channels,
                         feature_transform=[], ignore_first_ln=True))
    with torch.no_grad():   # non-trivial LayerNorm affine parameters (defaults are 1 / 0)
        for n_, p in net.named_parameters():
            if "norm" in n_ or n_.startswith("final_mlp.1"):
                p.uniform_(0.5, 1.5) if n_.endswith("weight") else p.uniform_(-0.5, 0.5)
    cap = {}
    net.conv.register_forward_hook(lambda m, i, o: cap.__setitem__("prepool", o.detach()))
    tobs = {k: torch.from_numpy(v) for k, v in obs.items()}
    with torch.no_grad():
        feat = net(tobs)
    vals, idx = cap["prepool"].max(-1)
    top2 = cap["prepool"].topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1])
    live = vals > 0
    min_gap = float(gap[live].min()) if live.any() else 0.0
    out = {f"obs/{k}": v for k, v in obs.items()}
    out.update({f"w/{k}": np_(v) for k, v in net.state_dict().items()})
    out.update(pooled=np_(vals), argmax=np_(idx).astype(np.int32), feature=np_(feat),
               min_live_top2_gap=np.float32(min_gap), frac_argmax0=np.float32((idx == 0).float().mean()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: B={B} N={N} C={C} pooled {tuple(vals.shape)} min live top-2 gap {min_gap:.3e} frac idx0 {float((idx==0).float().mean()):.3f}")


# ----------------------------------------------------------------------------------------------
# B/C. update-step fixtures: reference SAC / DrQ update_parameters
# ----------------------------------------------------------------------------------------------
def unique_named_params(agent):
    return {n: p for n, p in agent.named_parameters()}       # nn.Module dedups shared tensors


def tensor_summary(d):
    return {k: np.array([float(v.double().sum()), float(v.double().abs().sum())]) for k, v in d.items()}


def gen_step(name, cfg_file, overrides, obs_kw, B, N, A, n_updates, seed):
    import torch.distributions.normal as tdn
    from pyrl.utils.augmentations.pcd_aug import RandomJitterPoints
    from pyrl.utils.data import DictArray

    obs_shape = {"xyz": [3, N], "rgb": [3, N]}
    if obs_kw.get("seg"):
        obs_shape["seg"] = [obs_kw["seg"], N]
    if obs_kw.get("pos_encoding"):
        obs_shape["pos_encoding"] = [obs_kw["pos_encoding"], N]
    if obs_kw.get("agent"):
        obs_shape["agent"] = obs_kw["agent"]
    agent, cfg = ref_stubs.build_reference_agent(cfg_file, obs_shape, A, overrides, seed=seed)
    agent.batch_size = B
    prefix = "drq" if type(agent).__name__ == "DrQ" else "sac"

    out = {}
    out["meta/agent_type"] = np.array(type(agent).__name__)
    out["meta/dims"] = np.array([B, N, A, obs_kw.get("agent", 0), n_updates])
    out["meta/hyper"] = np.array([agent.gamma, agent.reward_scale, float(np.exp(float(agent.log_alpha.item()))),
                                  agent.target_entropy, agent.actor_update_interval, agent.target_update_interval,
                                  getattr(agent, "num_aug", 1)], dtype=np.float64)
    out["meta/use_episode_dones"] = np.array(bool(getattr(agent, "use_episode_dones", False)))
    out["meta/svea"] = np.array(bool(getattr(agent, "svea", False)))
    out["meta/update_coeff_default"] = np.array(agent.update_coeff["default"] if isinstance(agent.update_coeff, dict) else agent.update_coeff)
    for k, v in unique_named_params(agent).items():
        out[f"init/{k}"] = np_(v)

    cap = {"eps": [], "jit": [], "argmax": [], "pooled": []}
    _orig_sn = tdn._standard_normal

    def _sn(shape, dtype, device):
        e = _orig_sn(shape, dtype=dtype, device=device)
        cap["eps"].append(e.clone())
        return e
    tdn._standard_normal = _sn
    _orig_ps = RandomJitterPoints.process_single

    def _ps(self, xyz, key):
        o = _orig_ps(self, xyz, key)
        cap["jit"].append((o - xyz).clone())
        return o
    RandomJitterPoints.process_single = _ps
    enc = agent.actor.backbone.visual_nn

    def _hook(m, i, o):
        v, ix = o.detach().max(-1)
        cap["pooled"].append(v.clone())
        cap["argmax"].append(ix.clone())
    enc.conv.register_forward_hook(_hook)

    grads = {}

    def wrap_step(optim, tag, module_params):
        orig = optim.step

        def step(*a, **k):
            grads[tag] = {n: p.grad.detach().clone() for n, p in module_params() if p.grad is not None}
            return orig(*a, **k)
        optim.step = step
    wrap_step(agent.critic_optim, "critic", lambda: [(n, p) for n, p in agent.critic.named_parameters() if p.requires_grad])
    wrap_step(agent.actor_optim, "actor", lambda: [(n, p) for n, p in agent.actor.named_parameters() if p.requires_grad])
    wrap_step(agent.alpha_optim, "alpha", lambda: [("log_alpha", agent.log_alpha)])

    g = np.random.RandomState(seed + 100)
    for u in range(1, n_updates + 1):
        batch = dict(obs=make_obs(g, B, N, **obs_kw), next_obs=make_obs(g, B, N, **obs_kw),
                     actions=g.uniform(-1, 1, (B, A)).astype(np.float32), prev_actions=g.uniform(-1, 1, (B, A)).astype(np.float32),
                     rewards=g.randn(B, 1).astype(np.float32), dones=(g.rand(B, 1) < 0.25), episode_dones=(g.rand(B, 1) < 0.25))
        for side in ("obs", "next_obs"):
            for k, v in batch[side].items():
                out[f"u{u}/batch/{side}/{k}"] = v
        for k in ("actions", "rewards", "dones") + (("episode_dones",) if getattr(agent, "use_episode_dones", False) else ()):
            out[f"u{u}/batch/{k}"] = batch[k]

        class Mem:
            def sample(self, bs):
                return DictArray(copy.deepcopy(batch))
        for k in cap:
            cap[k].clear()
        grads.clear()
        ret = agent.update_parameters(Mem(), u)
        for k, v in ret.items():
            out[f"u{u}/ret/{k.split('/', 1)[1]}"] = np.float64(v)
        for i, e in enumerate(cap["eps"]):
            out[f"u{u}/eps{i}"] = np_(e)
        for i, j in enumerate(cap["jit"]):
            out[f"u{u}/jitter{i}"] = np_(j)
        out[f"u{u}/n_encoder_passes"] = np.array(len(cap["argmax"]))
        for i, (a_, p_) in enumerate(zip(cap["argmax"], cap["pooled"])):
            out[f"u{u}/enc{i}/argmax"] = np_(a_).astype(np.int16)
            out[f"u{u}/enc{i}/pooled"] = np_(p_)
        full = u <= 2
        for tag, gd in grads.items():
            if full:
                for n_, v in gd.items():
                    out[f"u{u}/grad_{tag}/{n_}"] = np_(v)
            for n_, s in tensor_summary(gd).items():
                out[f"u{u}/gradsum_{tag}/{n_}"] = s
        params = unique_named_params(agent)
        if u == 2:
            for n_, v in params.items():
                out[f"u{u}/param/{n_}"] = np_(v)
        for n_, s in tensor_summary({k: v.detach() for k, v in params.items()}).items():
            out[f"u{u}/paramsum/{n_}"] = s
        print(f"{name} update {u}: eps {[tuple(e.shape) for e in cap['eps']]} jitter {len(cap['jit'])} "
              f"enc passes {len(cap['argmax'])} ret {{{', '.join(f'{k}: {v:.5g}' for k, v in ret.items())}}}")
    tdn._standard_normal = _orig_sn
    RandomJitterPoints.process_single = _orig_ps
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {os.path.getsize(os.path.join(OUT, name + '.npz')) / 1e6:.2f} MB")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    only = set(sys.argv[1:])          # fixture names to (re)generate; none: all

    def _want(name):
        return not only or name in only
    _gen_encoder, _gen_step = gen_encoder, gen_step
    gen_encoder = lambda name, *a, **k: _gen_encoder(name, *a, **k) if _want(name) else None
    gen_step = lambda name, *a, **k: _gen_step(name, *a, **k) if _want(name) else None
    gen_encoder("encoder_dmc_c6", {}, [64, 128, 256], 50, B=4, N=100, seed=1)
    gen_encoder("encoder_dmc_c9_posenc", dict(pos_encoding=3), [64, 128, 256], 50, B=2, N=96, seed=2)
    gen_encoder("encoder_maniskill_c7", dict(seg=1), [128, 128, 256], 128, B=3, N=75, seed=3)
    small_heads = {
        "agent_cfg.actor_cfg.nn_cfg.mlp_cfg.mlp_spec": [50, 64, 64, "action_shape * 2"],
        "agent_cfg.critic_cfg.nn_cfg.mlp_cfg.mlp_spec": ["50 + action_shape", 64, 64, 1],
    }
    gen_step("sac_dmc_small", f"{REF}/configs/mfrl/sac/dm_control/pn.py", small_heads, {}, B=8, N=64, A=6, n_updates=4, seed=0)
    gen_step("drq_dmc_jitter_small", f"{REF}/configs/mfrl/drq/dm_control/pn_jitter.py", small_heads, {}, B=4, N=64, A=6, n_updates=4, seed=1)
    # SVEA (drq.py:62-67,87-88,115): no shipped config sets it, the constructor argument exists; pn_jitter with svea / num_aug 1.
    # (seed 7: with seed 6 one element of the critic's gradient is a 1.3e-9 cancellation residue, inside Adam's eps band, and the
    # first update of that element differs by 1e-4 between any two summation orders -- DESIGN.md section 2.)
    gen_step("drq_svea_dmc_jitter_small", f"{REF}/configs/mfrl/drq/dm_control/pn_jitter.py",
             dict(small_heads, **{"agent_cfg.svea": True, "agent_cfg.num_aug": 1}), {}, B=4, N=64, A=6, n_updates=4, seed=7)
    # BASELINE config 1 layout (dmc_walker_walk: 3 stacked frames, C = 9 = xyz + rgb + one-hot frame id)
    gen_step("sac_dmc_k0_posenc_small", f"{REF}/configs/mfrl/sac/dm_control/pn.py", small_heads, dict(pos_encoding=3), B=4, N=96, A=6,
             n_updates=2, seed=3)
    ms_heads = {
        "agent_cfg.actor_cfg.nn_cfg.mlp_cfg.mlp_spec": ["128 + agent_shape", 64, 64, "action_shape * 2"],
        "agent_cfg.critic_cfg.nn_cfg.mlp_cfg.mlp_spec": ["128 + agent_shape + action_shape", 64, 64, 1],
    }
    gen_step("drq_maniskill_jitter_small", f"{REF}/configs/mfrl/drq/maniskill/pn_jitter.py", ms_heads, dict(seg=1, agent=10),
             B=4, N=48, A=8, n_updates=2, seed=2)
    # the small encoder of the "motivating" configs: mlp_spec [32, 64, 128], use_episode_dones=True
    gen_encoder("encoder_dmc_motivating_c6", {}, [32, 64, 128], 50, B=3, N=90, seed=4)
    gen_step("sac_dmc_motivating_small", f"{REF}/configs/mfrl/sac/dm_control/pn_motivating.py", small_heads, {}, B=8, N=80, A=4,
             n_updates=4, seed=5)
    # PointNet's class default mlp_spec = [64, 128, 1024] (pointnet.py:81; no shipped SAC / DrQ config uses it): encoder fixture and a
    # whole SAC step of the dm_control config with the encoder's spec overridden
    gen_encoder("encoder_classdefault_c6", {}, [64, 128, 1024], 50, B=3, N=80, seed=5)
    gen_step("sac_dmc_classdefault_small", f"{REF}/configs/mfrl/sac/dm_control/pn.py",
             dict(small_heads, **{"agent_cfg.actor_cfg.nn_cfg.visual_nn_cfg.mlp_spec": [64, 128, 1024]}), {}, B=4, N=72, A=6, n_updates=2, seed=8)

