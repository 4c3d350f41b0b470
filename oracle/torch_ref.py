"""PyTorch-CPU restatement of the reference's point-cloud SAC / DrQ hot path -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
It mirrors the reference op for op -- including what makes the reference slow: one encoder pass
per Q head (3x next_obs, 2x obs, +1 for the actor), the permute->contiguous->layer_norm->permute->
contiguous LayerNorm1D, one Adam parameter group per tensor -- so that it can serve both as the
parity oracle for the update step and as the timed CPU baseline (`cpu_baseline.kind = "port"`).

Parity status: PINNED.  tests/test_oracle_golden.py checks it against tests/golden/*.npz, which
tools/gen_golden.py captured from the reference itself (weights, batches, random draws, metrics,
gradients, post-update parameters).

Parameters live in one flat dict keyed by the reference's own `named_parameters()` names
(e.g. "actor.backbone.visual_nn.conv.mlp.conv0.weight"), see SURVEY.md section 5.
"""
import math

import torch
import torch.nn.functional as F

ENC = "actor.backbone.visual_nn."


# ---------------------------------------------------------------------------------------------
# networks
# ---------------------------------------------------------------------------------------------
def preprocess(obs):
    """PointCloudBase.preprocess (pyrl/networks/backbones/pointnet.py:49-73)."""
    feature = [obs["xyz"]]
    if "rgb" in obs:
        rgb = obs["rgb"]
        if rgb.dtype == torch.uint8:
            rgb = rgb / 255.0
        feature.append(rgb)
    for key in ("pos_encoding", "seg"):
        if key in obs:
            feature.append(obs[key].to(dtype=torch.float32))
    return torch.cat(feature, dim=-2)


def layer_norm_1d(x, weight, bias, eps):
    """LayerNormkD.forward, channels_first (pyrl/networks/modules/nn_layer.py:207-219)."""
    x = x.permute(0, 2, 1).contiguous()
    x = F.layer_norm(x, (x.shape[-1],), weight, bias, eps)
    return x.permute(0, 2, 1).contiguous()


def pointnet_prepool(P, obs, prefix=ENC, ln_eps=1e-6):
    """ConvMLP of PointNet (mlp.py:43-56 via pointnet.py:106-109): conv0+ReLU, conv1+LN+ReLU, conv2+LN+ReLU."""
    x = preprocess(obs)
    c = prefix + "conv.mlp."
    x = F.relu(F.conv1d(x, P[c + "conv0.weight"], P[c + "conv0.bias"]))
    x = F.relu(layer_norm_1d(F.conv1d(x, P[c + "conv1.weight"]), P[c + "norm1.weight"], P[c + "norm1.bias"], ln_eps))
    x = F.relu(layer_norm_1d(F.conv1d(x, P[c + "conv2.weight"]), P[c + "norm2.weight"], P[c + "norm2.bias"], ln_eps))
    return x


def pointnet_forward(P, obs, prefix=ENC, ln_eps=1e-6, route=None, keep=None):
    """PointNet.forward (pointnet.py:148-153): shared MLP, max over points, Linear + LayerNorm(eps=1e-5).

    route (tests only): int64 [B, c3] -- the point each (cloud, channel) sends its value and gradient through, instead of
    torch's own argmax (two points whose fp32 values differ by an ulp are ordered by the summation order; the caller checks
    that the routed point holds the maximum to rounding).  keep (tests only): dict that receives the pooled tensor
    (`retain_grad`: its .grad after backward is dL/dpooled) and the observation, for encoder_cloud_grads below."""
    pre = pointnet_prepool(P, obs, prefix, ln_eps)
    feat = pre.max(-1)[0] if route is None else pre.gather(-1, route[..., None])[..., 0]
    if keep is not None:
        feat.retain_grad()
        keep["pooled"], keep["obs"] = feat, obs
        with torch.no_grad():       # how far the routing is from torch's own: entries that differ, and the value gap there
            top, own = pre.max(-1)
            keep["argmax"] = own
            keep["route_differs"] = int((own != route).sum()) if route is not None else 0
            keep["route_gap"] = float((top - feat).abs().max())
    f = prefix + "final_mlp."
    feat = F.linear(feat, P[f + "0.weight"], P[f + "0.bias"])
    return F.layer_norm(feat, (feat.shape[-1],), P[f + "1.weight"], P[f + "1.bias"], 1e-5)


ENC_TENSORS = ("conv0.weight", "conv0.bias", "conv1.weight", "norm1.weight", "norm1.bias", "conv2.weight", "norm2.weight", "norm2.bias")


class CloudEncoder:
    """One cloud's part of the encoder gradient, restricted to the points that receive gradient (tests only).

    The max-pool hands dL/dpooled[c] to ONE point per channel, so a cloud's contribution to the gradient of the eight
    `visual_nn.conv.mlp.*` tensors is a function of its <= c3 routed points.  This class evaluates that function with the
    ReLU branch decisions as explicit 0/1 masks (h = pre-activation * mask): with the restatement's own decisions it
    reproduces the cloud's share of `backward()`; with ONE decision flipped it gives the gradient another fp32
    implementation produces when its summation order puts that pre-activation -- which must be rounding-sized -- on the
    other side of zero.  tests/test_fullsize_parity_gpu.py uses the difference to locate such events."""

    def __init__(self, P, obs_b, route_b, gpool_b, prefix=ENC, ln_eps=1e-6):
        c = prefix + "conv.mlp."
        self.W = {n: P[c + n].detach().clone().requires_grad_(True) for n in ENC_TENSORS}
        self.eps, self.gpool = ln_eps, gpool_b
        pts, self.slot = torch.unique(route_b, return_inverse=True)            # active points (ascending), slot of every channel
        self.points = pts
        self.x = preprocess({k: v[None, :, pts] for k, v in obs_b.items()})     # [1, C, n_act]
        with torch.no_grad():
            z0, y1, y2 = self._pre(None, None)
        self.z0, self.y1 = z0[0], y1[0]                                       # [c1, n_act], [c2, n_act]
        self.y2 = y2[0].gather(-1, self.slot[:, None])[:, 0]                  # [c3]: the routed entry of every channel
        self.base = (self.z0 > 0, self.y1 > 0, self.y2 > 0)

    def _pre(self, m0, m1):
        W = self.W
        z0 = F.conv1d(self.x, W["conv0.weight"], W["conv0.bias"])
        h0 = F.relu(z0) if m0 is None else z0 * m0[None]
        y1 = layer_norm_1d(F.conv1d(h0, W["conv1.weight"]), W["norm1.weight"], W["norm1.bias"], self.eps)
        h1 = F.relu(y1) if m1 is None else y1 * m1[None]
        y2 = layer_norm_1d(F.conv1d(h1, W["conv2.weight"]), W["norm2.weight"], W["norm2.bias"], self.eps)
        return z0, y1, y2

    def candidates(self, tau):
        """[(layer, channel, active-point slot or -1, |pre-activation|)] of the decisions within tau of zero."""
        out = []
        for layer, z in ((0, self.z0), (1, self.y1)):
            for ch, sl in (z.abs() <= tau).nonzero().tolist():
                out.append((layer, ch, sl, float(z[ch, sl].abs())))
        for (ch,) in (self.y2.abs() <= tau).nonzero().tolist():
            out.append((2, ch, -1, float(self.y2[ch].abs())))
        return out

    def grads(self, flip=None):
        """{tensor name: gradient} of sum_c gpool[c] * pooled[c] for this cloud; flip = (layer, channel, slot) inverts one
        decision."""
        m = [t.clone().float() for t in self.base]
        if flip is not None:
            layer, ch, sl = flip
            if layer == 2:
                m[2][ch] = 1.0 - m[2][ch]
            else:
                m[layer][ch, sl] = 1.0 - m[layer][ch, sl]
        for w in self.W.values():
            w.grad = None
        _, _, y2 = self._pre(m[0], m[1])
        pooled = y2[0].gather(-1, self.slot[:, None])[:, 0] * m[2]
        (pooled * self.gpool).sum().backward()
        return {n: (w.grad.detach().clone() if w.grad is not None else torch.zeros_like(w)) for n, w in self.W.items()}


def linear_mlp(P, prefix, x, masks=None, flips=None):
    """LinearMLP with norm_cfg=None, ReLU between layers, linear output (mlp.py:97-100, 48-49).

    masks (tests only): per hidden layer a 0/1 tensor that REPLACES the ReLU's own branch decision (x * mask instead of
    relu(x)) -- the decisions of another implementation of the same network.  A ReLU input within rounding of zero is
    decided by the summation order, and which way it falls changes the layer's weight gradient by a whole sample's
    contribution; with the decisions injected, two implementations can be compared at tight tolerance, and the decisions
    themselves are compared separately: every disagreement is recorded in `flips` as (count, largest |pre-activation|
    among them), which must be rounding-sized."""
    i = 0
    while f"{prefix}linear{i}.weight" in P:
        x = F.linear(x, P[f"{prefix}linear{i}.weight"], P[f"{prefix}linear{i}.bias"])
        if f"{prefix}linear{i + 1}.weight" in P:
            if masks is None:
                x = F.relu(x)
            else:
                m = masks[i].to(x.dtype)
                if flips is not None:
                    bad = (x.detach() > 0) != (m > 0)
                    flips.append((int(bad.sum()), float(x.detach().abs()[bad].max()) if bad.any() else 0.0))
                x = x * m
        i += 1
    return x


def visuomotor(P, mlp_prefix, obs, actions=None, visual_feature=None, detach_visual=False, count=None, masks=None, flips=None):
    """Visuomotor.forward (pyrl/networks/backbones/visuomotor.py:56-146), non-recurrent path.
    Returns (output of final_mlp, visual feature before the robot state is appended)."""
    obs = dict(obs)
    for key in list(obs.keys()):
        if "_box" in key or "_seg" in key or "_sem_label" in key or key == "visual_state":
            obs.pop(key)
    robot_state = None
    for key in ("state", "agent"):
        if key in obs:
            assert robot_state is None
            robot_state = obs.pop(key)
    if visual_feature is None:
        feat = pointnet_forward(P, obs)
        if count is not None:
            count[0] += 1
        if detach_visual:
            feat = feat.detach()
    else:
        feat = visual_feature
    saved = feat
    if robot_state is not None:
        feat = torch.cat([feat, robot_state], dim=-1)
    if actions is not None:
        feat = torch.cat([feat, actions], dim=-1)
    return linear_mlp(P, mlp_prefix, feat, masks, flips), saved


def tanh_gaussian(feature, eps, scale, bias, log_std_bound=(-10.0, 2.0), epsilon=1e-6):
    """TanhGaussianHead mode="max-entropy" (regression_heads/gaussian.py:23-50, 83-87;
    regression_base.py:50-74) with ScaledTanhNormal.rsample_with_log_prob
    (pyrl/utils/torch/distributions.py:89, 116-119) and the sum over the action axis (122-127).
    Returns (action, neg_logp[..., None])."""
    mean, log_std = feature.chunk(2, dim=-1)
    std = torch.clamp(log_std, min=log_std_bound[0], max=log_std_bound[1]).exp()
    logit = mean + eps * std
    var = std ** 2
    log_prob = -((logit - mean) ** 2) / (2 * var) - std.log() - math.log(math.sqrt(2 * math.pi))
    log_prob = log_prob - torch.log(scale * (1 - torch.tanh(logit).pow(2)) + epsilon)
    sample = torch.tanh(logit) * scale + bias
    return sample, -log_prob.sum(-1)[..., None]


class RefAgent:
    """SAC / DrQ agent state + update_parameters, restated from pyrl/methods/mfrl/sac.py and drq.py.

    params: dict name -> tensor (reference names; shared encoder stored once under
    "actor.backbone.visual_nn.*").  Random draws are injected (eps_list, jitter_list) so that a
    run can be compared with captured reference runs.
    """

    def __init__(self, params, kind="sac", gamma=0.99, reward_scale=1.0, alpha=0.1, target_entropy=None,
                 actor_update_interval=2, target_update_interval=2, update_coeff=0.01, num_aug=2,
                 jitter_range=None, lr=1e-3, alpha_betas=(0.5, 0.999), mirror_redundancy=True, svea=False):
        self.kind, self.gamma, self.reward_scale = kind, gamma, reward_scale
        self.svea = bool(svea)          # drq.py:24-28: num_aug must be 1
        assert not self.svea or (kind == "drq" and num_aug == 1)
        self.actor_update_interval, self.target_update_interval = actor_update_interval, target_update_interval
        self.update_coeff, self.num_aug, self.jitter_range = update_coeff, num_aug, jitter_range
        self.mirror_redundancy = mirror_redundancy
        self.P = {k: v.clone().float() for k, v in params.items()}
        self.action_dim = self.P["actor.head.scale"].shape[0]
        self.target_entropy = -float(self.action_dim) if target_entropy is None else target_entropy
        if "log_alpha" not in self.P:
            self.P["log_alpha"] = torch.ones(1) * float(torch.log(torch.tensor(alpha, dtype=torch.float32)))
        enc = [k for k in self.P if k.startswith(ENC)]
        self.critic_names = enc + [k for k in self.P if k.startswith("critic.")]
        self.actor_names = [k for k in self.P if k.startswith("actor.backbone.final_mlp.")]   # visual_nn excluded (pn.py:41)
        self.actor_module_names = enc + self.actor_names
        for k in self.critic_names + self.actor_names + ["log_alpha"]:
            self.P[k].requires_grad_(True)
        # build_optimizer: one param group per tensor (pyrl/utils/torch/optimizer_utils.py:43-57)
        self.critic_optim = torch.optim.Adam([{"params": self.P[k]} for k in self.critic_names], lr=lr)
        self.actor_optim = torch.optim.Adam([{"params": self.P[k]} for k in self.actor_names], lr=lr)
        self.alpha_optim = torch.optim.Adam([self.P["log_alpha"]], lr=lr, betas=alpha_betas)
        self.alpha = float(self.P["log_alpha"].exp().item())
        self.encoder_passes = [0]
        self.last_grads = {}
        self.flips = []          # (count, max |pre-activation|) per masked hidden layer, see linear_mlp
        self.route = None        # tests only: argmax routing of the gradient-carrying encoder pass (pointnet_forward(route=))
        self.keep = None         # tests only: dict receiving that pass's pooled tensor / observation
        self.critic_grad_hook = None   # tests only: called with self after the critic's backward, before its optimizer step
        self.post_critic_hook = None   # tests only: called with self right after the critic's optimizer step (before the actor phase)

    # -- modules ---------------------------------------------------------------------------
    def actor(self, obs, eps, detach_visual=False, masks=None):
        feat, saved = visuomotor(self.P, "actor.backbone.final_mlp.mlp.", obs, detach_visual=detach_visual, count=self.encoder_passes,
                                 masks=masks, flips=self.flips)
        a, neg_logp = tanh_gaussian(feat, eps, self.P["actor.head.scale"], self.P["actor.head.bias"])
        return a, neg_logp, saved

    def critic(self, obs, actions, which="critic", visual_feature=None, masks=None):
        """ContinuousCritic.forward (applications/actor_critic.py:122-133): one Visuomotor pass per head.
        masks: [head][hidden layer] injected ReLU decisions (see linear_mlp)."""
        outs = []
        shared = None
        for h in (0, 1):
            vf = visual_feature
            if vf is None and not self.mirror_redundancy:
                if shared is None:
                    grad_pass = which == "critic" and torch.is_grad_enabled()
                    shared = pointnet_forward(self.P, {k: v for k, v in obs.items() if k not in ("state", "agent")},
                                              route=self.route if grad_pass else None, keep=self.keep if grad_pass else None)
                    self.encoder_passes[0] += 1
                vf = shared
            q, _ = visuomotor(self.P, f"{which}.values.{h}.backbone.final_mlp.mlp.", obs, actions=actions, visual_feature=vf, count=self.encoder_passes,
                              masks=None if masks is None else masks[h], flips=self.flips)
            outs.append(q)
        return torch.cat(outs, dim=-1)

    @staticmethod
    def grad_norm(tensors):
        """ExtendedModuleBase.grad_norm (pyrl/utils/torch/module_utils.py:40-45)."""
        grads = [torch.norm(t.grad.detach(), 2) for t in tensors if t.requires_grad and t.grad is not None]
        return torch.norm(torch.stack(grads), 2).item() if grads else 0.0

    def soft_update(self):
        """soft_update (pyrl/utils/torch/ops.py:59-90); the shared visual_nn is skipped by identity."""
        tau = self.update_coeff
        with torch.no_grad():
            for k in self.P:
                if k.startswith("critic."):
                    t = self.P["target_" + k]
                    t.copy_(t * (1.0 - tau) + self.P[k] * tau)

    # -- augmentation ----------------------------------------------------------------------
    def _aug(self, obs, noise, affine=None):
        """GDict.repeat(num_aug, 0) = repeat_interleave of every leaf (array_ops.py:106-121), then -- in the order of the
        obs_aug list [GlobalRotScaleTrans, RandomJitterPoints] -- apply_rot_trans on xyz with the injected [rows, 3, 4] matrices
        (pcd_aug.py:84-123: einsum("bin,bji->bjn", x, rot) + xyz[..., None]) and RandomJitterPoints (pcd_aug.py:316-322) with the
        injected noise."""
        rep = {k: torch.repeat_interleave(v, self.num_aug, dim=0) for k, v in obs.items()}
        if affine is not None:
            rep["xyz"] = torch.einsum("bin,bji->bjn", rep["xyz"], affine[:, :, :3]) + affine[:, :, 3:]
        if noise is not None:
            rep["xyz"] = rep["xyz"] + noise
        return rep

    # -- the update step ---------------------------------------------------------------------
    def update_parameters(self, batch, updates, eps_list, jitter_list=None, relu_masks=None, affine_list=None):
        """relu_masks (tests only): {"q": [head][layer], "pi": [layer], "q_pi": [head][layer]} ReLU decisions of the
        gradient-carrying head passes taken from the implementation under test (see linear_mlp)."""
        P = self.P
        relu_masks = relu_masks or {}
        self.flips = []
        pre = self.kind
        eps_list = list(eps_list)
        obs, next_obs = batch["obs"], batch["next_obs"]
        actions, rewards, dones = batch["actions"], batch["rewards"], batch["dones"]
        B = actions.shape[0]
        plain_obs = obs
        if self.kind == "drq" and self.svea:
            # drq.py:62-67: rows [aug(s_b), s_b] interleaved; s', rewards and dones are neither augmented nor repeated
            aug = self._aug(obs, jitter_list[0] if jitter_list else None)
            obs = {k: torch.stack([aug[k], v], dim=1).flatten(0, 1) for k, v in obs.items()}
            actions = torch.repeat_interleave(actions, 2, dim=0)
        elif self.kind == "drq":
            # drq.py:52-63
            obs = self._aug(obs, jitter_list[0] if jitter_list else None, affine_list[0] if affine_list else None)
            next_obs = self._aug(next_obs, jitter_list[1] if jitter_list else None, affine_list[1] if affine_list else None)
            actions = torch.repeat_interleave(actions, self.num_aug, dim=0)
            rewards = torch.repeat_interleave(rewards, self.num_aug, dim=0)
            dones = torch.repeat_interleave(dones, self.num_aug, dim=0)
        with torch.no_grad():
            next_a, neg_logp, _ = self.actor(next_obs, eps_list.pop(0))
            q_next = self.critic(next_obs, next_a, which="target_critic")
            min_q_next = torch.min(q_next, dim=-1, keepdim=True).values
            min_q_next = min_q_next + self.alpha * neg_logp
            if self.kind == "drq":    # drq.py:79-88 (no reward_scale; mean over the aug axis, SVEA: one target per [aug, plain] pair)
                q_target = rewards + (1 - dones.float()) * self.gamma * min_q_next
                if not self.svea:
                    q_target = q_target.reshape(B, self.num_aug).mean(1, keepdim=True)
                q_target = torch.repeat_interleave(q_target, self.num_aug + int(self.svea), dim=0).repeat(1, q_next.shape[-1])
            else:                      # sac.py:131-134
                q_target = rewards * self.reward_scale + (1 - dones.float()) * self.gamma * min_q_next
                q_target = q_target.repeat_interleave(q_next.shape[-1], dim=-1)
        q = self.critic(obs, actions, masks=relu_masks.get("q"))
        critic_loss = F.mse_loss(q, q_target) * q_target.shape[-1]
        with torch.no_grad():
            abs_err = torch.abs(q - q_target).max().item()
        self.critic_optim.zero_grad()
        critic_loss.backward()
        if self.critic_grad_hook is not None:
            self.critic_grad_hook(self)
        self.last_grads = {"critic": {k: P[k].grad.detach().clone() for k in self.critic_names if P[k].grad is not None}}
        self.critic_optim.step()
        if self.post_critic_hook is not None:
            self.post_critic_hook(self)
        critic_grad = self.grad_norm([P[k] for k in self.critic_names])
        self.critic_optim.zero_grad()   # shared_backbone (sac.py:147-148)
        ret = {
            f"{pre}/critic_loss": critic_loss.item(), f"{pre}/max_critic_abs_err": abs_err, f"{pre}/alpha": self.alpha,
            f"{pre}/q": torch.min(q, dim=-1).values.mean().item(), f"{pre}/q_target": torch.mean(q_target).item(),
            f"{pre}/target_entropy": self.target_entropy, f"{pre}/critic_grad": critic_grad, f"{pre}/grad_steps": 1,
        }
        if updates % self.actor_update_interval == 0:
            if self.kind == "drq" and self.svea:     # drq.py:115: the sampled (plain) observations
                a_obs = plain_obs
            elif self.kind == "drq":  # drq.py:115: first augmentation of every sample
                a_obs = {k: v.reshape(B, self.num_aug, *v.shape[1:])[:, 0] for k, v in obs.items()}
            else:
                a_obs = obs
            pi, neg_logp, saved = self.actor(a_obs, eps_list.pop(0), detach_visual=True, masks=relu_masks.get("pi"))
            entropy = neg_logp.mean()
            q_pi = self.critic(a_obs, pi, visual_feature=saved.detach(), masks=relu_masks.get("q_pi"))
            q_pi = torch.min(q_pi, dim=-1, keepdim=True).values
            actor_loss = -(q_pi.mean() + self.alpha * entropy)
            self.actor_optim.zero_grad()
            actor_loss.backward()
            self.last_grads["actor"] = {k: P[k].grad.detach().clone() for k in self.actor_names}
            self.actor_optim.step()
            actor_grad = self.grad_norm([P[k] for k in self.actor_module_names])
            alpha_loss = P["log_alpha"].exp() * (entropy - self.target_entropy).detach()
            self.alpha_optim.zero_grad()
            alpha_loss.backward()
            self.last_grads["alpha"] = {"log_alpha": P["log_alpha"].grad.detach().clone()}
            self.alpha_optim.step()
            self.alpha = P["log_alpha"].exp().item()
            ret.update({f"{pre}/actor_loss": actor_loss.item(), f"{pre}/alpha_loss": alpha_loss.item(),
                        f"{pre}/entropy": entropy.item(), f"{pre}/actor_grad": actor_grad})
        if updates % self.target_update_interval == 0:
            self.soft_update()
        return ret


def params_from_fixture(d, section="init/"):
    return {k[len(section):]: torch.from_numpy(d[k]) for k in d.files if k.startswith(section)}


def batch_from_fixture(d, u):
    pre = f"u{u}/batch/"
    batch = {"obs": {}, "next_obs": {}}
    for k in d.files:
        if not k.startswith(pre):
            continue
        rest = k[len(pre):]
        t = torch.from_numpy(d[k])
        if "/" in rest:
            side, key = rest.split("/")
            batch[side][key] = t
        else:
            batch[rest] = t
    return batch
