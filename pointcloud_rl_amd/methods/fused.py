"""Autograd-free execution of the SAC / DrQ update step on libpcrl_hip.so.

Used by SAC/DrQ when the agent has the topology of every shipped point-cloud config (one PointNet
shared by the actor, both Q heads and both target Q heads; 3-layer LinearMLP heads without norm;
TanhGaussianHead; detach_actor_feature=True; plain Adam).  The step is then a fixed sequence of
~55 launches: fused encoder forward/backward, batched fp32 MFMA GEMMs for the heads (both Q heads in
one launch), fused squashed-Gaussian / TD-target / loss kernels and fused Adam+Polyak -- versus
~800 ATen launches for the same step through autograd.  Semantics (reference sac.py:103-214,
drq.py:46-165) are unchanged and checked against golden vectors in tests/test_update_step_gpu.py.
"""
import torch

from .. import hip
from ..networks.mlp import LinearMLP
from ..networks.heads import TanhGaussianHead
from ..networks.pointnet import PointNet


class PackedStats(dict):
    """name -> device scalar (views into `packed`, one contiguous float32 tensor in the same order); `host`: pinned float32
    tensor that receives the same values from the launch that fills `packed`, one 4-byte store per value (hip.gather_scalars)."""
    packed = None
    host = None


def ceil4(x):
    return (x + 3) & ~3


class MlpView:
    """Three Linear layers of a LinearMLP located inside a flat parameter buffer (data and grad)."""

    def __init__(self, flat, prefix, head_stride=0, nb=1):
        self.flat, self.nb, self.hs = flat, nb, head_stride
        self.off = dict(zip(flat.names, flat.offsets))
        self.dims = []
        for i in range(3):
            w = flat.params[flat.names.index(f"{prefix}linear{i}.weight")]
            self.dims.append((w.shape[1], w.shape[0]))      # (in, out)
        self.w = [self.off[f"{prefix}linear{i}.weight"] for i in range(3)]
        self.b = [self.off[f"{prefix}linear{i}.bias"] for i in range(3)]

    def W(self, i, buf=None, base=0):
        return (buf if buf is not None else self.flat.data)[base + self.w[i]:]

    def Bv(self, i, buf=None, base=0):
        return (buf if buf is not None else self.flat.data)[base + self.b[i]:]


def mlp_forward_descs(mv, params, base, X, ldx, M, outs, ld_out2, out2, out2_bs, x_bs=0):
    """GEMM descriptors, one per layer, of h1 = relu(X W0^T + b0), h2 = relu(h1 W1^T + b1), out = h2 W2^T + b2,
    batched over mv.nb heads.  outs = (h1, h2) buffers [nb][M][H]; out2 written with leading dim ld_out2 and batch
    stride out2_bs."""
    (k0, h), (_, _), (_, n2) = mv.dims
    nb, hs = mv.nb, mv.hs
    h1, h2 = outs
    return [hip.gemm_desc(X, mv.W(0, params, base), h1, M, h, k0, (ldx, 1), (1, k0), h, bias=mv.Bv(0, params, base), relu=True,
                          batch=nb, batch_strides=(x_bs, hs, M * h, hs, 0)),
            hip.gemm_desc(h1, mv.W(1, params, base), h2, M, h, h, (h, 1), (1, h), h, bias=mv.Bv(1, params, base), relu=True,
                          batch=nb, batch_strides=(M * h, hs, M * h, hs, 0)),
            hip.gemm_desc(h2, mv.W(2, params, base), out2, M, n2, h, (h, 1), (1, h), ld_out2, bias=mv.Bv(2, params, base),
                          batch=nb, batch_strides=(M * h, hs, out2_bs, hs, 0))]


def launch_layers(*mlps):
    """Layer l of every given MLP (independent of each other) goes into one grouped launch."""
    for layer in zip(*mlps):
        hip.gemm_group([d for group in layer for d in (group if isinstance(group, (list, tuple)) else [group])])


def mlp_forward(*args, **kwargs):
    launch_layers(mlp_forward_descs(*args, **kwargs))


def mlp_backward_descs(mv, X, ldx, M, h1, h2, dout, dout_strides, dout_bs, dh1, dh2, grad=None, dX=None, dx_cols=None, ld_dx=0, x_bs=0):
    """Backward of mlp_forward as three stages of independent GEMMs [(dW2|db2, dh2), (dW1|db1, dh1), (dW0|db0, dX)].
    dout[z][m][n] at dout + z*dout_bs + m*dout_strides[0] + n*dout_strides[1].
    grad: flat gradient buffer receiving dW|db of every layer (None: data gradients only).
    dX: receives dh1 @ W0[:, c0:c0+nc] for dx_cols = (c0, nc), shape [nb][M][ld_dx]."""
    (k0, h), (_, _), (_, n2) = mv.dims
    nb, hs = mv.nb, mv.hs
    sm, sn = dout_strides
    stages = [[], [], []]
    if grad is not None:
        stages[0].append(hip.gemm_desc(dout, h2, grad[mv.w[2]:], n2, h + 1, M, (sn, sm), (h, 1), h, ones_col=h, c_ones=grad[mv.b[2]:],
                                       c_ones_batch_stride=hs, batch=nb, batch_strides=(dout_bs, M * h, hs, 0, 0)))
    stages[0].append(hip.gemm_desc(dout, mv.W(2), dh2, M, h, n2, (sm, sn), (h, 1), h, mask=h2, ld_mask=h, batch=nb,
                                   batch_strides=(dout_bs, hs, M * h, 0, M * h)))
    if grad is not None:
        stages[1].append(hip.gemm_desc(dh2, h1, grad[mv.w[1]:], h, h + 1, M, (1, h), (h, 1), h, ones_col=h, c_ones=grad[mv.b[1]:],
                                       c_ones_batch_stride=hs, batch=nb, batch_strides=(M * h, M * h, hs, 0, 0)))
    stages[1].append(hip.gemm_desc(dh2, mv.W(1), dh1, M, h, h, (h, 1), (h, 1), h, mask=h1, ld_mask=h, batch=nb,
                                   batch_strides=(M * h, hs, M * h, 0, M * h)))
    if grad is not None:
        stages[2].append(hip.gemm_desc(dh1, X, grad[mv.w[0]:], h, k0 + 1, M, (1, h), (ldx, 1), k0, ones_col=k0, c_ones=grad[mv.b[0]:],
                                       c_ones_batch_stride=hs, batch=nb, batch_strides=(M * h, x_bs, hs, 0, 0)))
    if dX is not None:
        c0, nc = dx_cols
        stages[2].append(hip.gemm_desc(dh1, mv.W(0)[c0:], dX, M, nc, h, (h, 1), (k0, 1), ld_dx, batch=nb,
                                       batch_strides=(M * h, hs, M * ld_dx, 0, 0)))
    return stages


def mlp_backward(*args, **kwargs):
    launch_layers(mlp_backward_descs(*args, **kwargs))


def _adjacent_halves(first, second):
    """{key: [2 B, ...] view} when every tensor of `second` starts where the same key of `first` ends (same storage, shape,
    dtype, contiguous) and neither carries an augmentation or a virtual repeat; else None."""
    if getattr(first, "aug", None) or getattr(second, "aug", None) or getattr(first, "repeat", 1) != 1 or getattr(second, "repeat", 1) != 1:
        return None
    if not isinstance(first, dict) or not isinstance(second, dict) or first.keys() != second.keys():
        return None
    out = {}
    for k, a in first.items():
        b = second[k]
        if not (torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
                and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
                and b.storage_offset() == a.storage_offset() + a.numel()):
            return None
        out[k] = a.as_strided((2 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
    return out


class FusedStep:
    """Buffers + launch sequence of one update step for a fixed (rows, actor rows) geometry."""

    @staticmethod
    def supported(agent):
        from .sac import HipAdam
        enc = agent.encoder
        if getattr(agent, "svea", False):        # DrQ(svea=True): [aug, plain] pairs against one target per pair (drq.py:62-88)
            return False
        if not isinstance(enc, PointNet) or enc.final_mlp is None or not agent._dedup or not agent.detach_actor_feature:
            return False
        if enc.mlp_spec[-1] > 256:          # the wide last layer (class default [64, 128, 1024]): encoder kernels + autograd heads
            return False
        if not all(isinstance(getattr(agent, f"{n}_optim"), HipAdam) for n in ("critic", "actor", "alpha")):
            return False
        if not agent.automatic_alpha_tuning or agent._target_flat is None or len(agent.critic.values) != 2:
            return False
        mlps = [agent.actor.backbone.final_mlp] + [v.backbone.final_mlp for v in agent.critic.values]
        if not all(isinstance(m, LinearMLP) and len(m.linears) == 3 and all(l.bias is not None for l in m.linears) for m in mlps):
            return False
        if not isinstance(agent.actor.head, TanhGaussianHead) or agent.actor.final_mlp is not None:
            return False
        if any(v.final_mlp is not None or v.head is not None for v in agent.critic.values):
            return False
        return agent.actor.backbone.ac_feat is None and agent.actor.backbone.obs_feat is None

    def __init__(self, agent):
        self.a = agent
        enc = agent.encoder
        self.F = enc.final_mlp[0].out_features
        self.c3 = enc.mlp_spec[-1]
        fc, fa = agent._flat["critic"], agent._flat["actor"]
        off = dict(zip(fc.names, fc.offsets))
        self.off = off
        self.n_conv = off["values.0.backbone.visual_nn.final_mlp.0.weight"]
        q0, q1 = "values.0.backbone.final_mlp.mlp.", "values.1.backbone.final_mlp.mlp."
        self.q = MlpView(fc, q0, head_stride=off[q1 + "linear0.weight"] - off[q0 + "linear0.weight"], nb=2)
        self.pi = MlpView(fa, "backbone.final_mlp.mlp.")
        self.q_base = off[q0 + "linear0.weight"]          # start of the Q heads inside the critic buffer
        assert agent._target_range[0] == self.q_base
        self.A = agent.actor.head.dim_output
        self.H = self.q.dims[0][1]
        self.Din_q, self.Din_a = self.q.dims[0][0], self.pi.dims[0][0]
        self.S = self.Din_a - self.F
        assert self.Din_q == self.Din_a + self.A and self.S >= 0
        self.ldq, self.lda = ceil4(self.Din_q), ceil4(self.Din_a)
        self.bufs = {}
        # head tails (csrc/headtail.hip): the last Linear of a head fused with the loss / squashed-Gaussian head that follows
        self.host_stats = None       # pinned mirror of the step's metrics (+ ready flag), allocated on first use
        self.tails = (self.H % 256 == 0 and self.H <= 1024 and 2 * self.A <= 64 and self.q.dims[2][1] == 1)
        dev = fc.data.device
        self.stats_c = torch.zeros(4, device=dev)
        self.stats_a = torch.zeros(3, device=dev)
        self.d_nlp = torch.zeros(1, device=dev)
        # policy noise stream: torch's seed (torch.manual_seed controls it), decorrelated across data-parallel ranks
        from ..utils.dist import rank as _rank
        self.seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (_rank() + 1)) & (2 ** 64 - 1)
        # action scale / bias as the kernels read them: float32 [A] on the device.  The head keeps them as the reference
        # does -- Python ints 1 / 0 for an unbounded action space, a float64 Parameter for scalar bounds
        # (regression_base.py:24-34) -- neither of which may be handed to a kernel as a raw float32 pointer.
        head = agent.actor.head

        def _as_f32(v):
            if torch.is_tensor(v):
                return v.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous().clone()
            return torch.full((self.A,), float(v), dtype=torch.float32, device=dev)
        self.head_scale, self.head_bias = _as_f32(head.scale), _as_f32(head.bias)
        assert self.head_scale.numel() == self.A and self.head_bias.numel() == self.A
        # PCRL_BWD_FORK=1: the encoder backward's first launch (hip.encoder_bwd_prepare needs only the forward's outputs) runs on
        # this stream, forked behind the forward and joined in front of the rest of the backward -- in a captured step a parallel
        # branch of the graph under the head GEMMs.  Measured on MI355X (profiles/r03_graph_branch.md): the two cross-stream edges
        # cost more than the 7.5 us launch they hide (K1 0.877 -> 0.902 ms per step, K3's 128-cloud share 0.701 -> 0.729), so the
        # default keeps the step one chain.
        self._side = None      # (the backward's prepare launch on a forked graph branch measured slower, profiles/r03_graph_branch.md; a test may set a stream)
        self._forked = False
        # rows x outputs up to which the policy's last layer runs inside the head kernels (csrc/headtail.hip) instead of as a GEMM + a
        # separate head launch: 4 096 for the wave-per-row kernels, 16 384 where the row-split kernels apply (H = 1024, <= 512 rows:
        # K3's 128-cloud share 0.669 -> 0.657 ms, K2's actor phase; K3's full batch stays on the GEMMs)
        self.policy_tail_max = 4096
        # forward tail: every row streams the 2A rows of the last layer against ~15.5 us for GEMM + head launch.  With one group of sixteen
        # outputs in flight the limit was 8 192 row-outputs (K2's actor phase, 11 264, took 25.6 us); with every piece requested at once
        # (policy_tail_fwd_split_kernel<G>, A <= 24) all of the row-split domain pays: 512 rows x 48 outputs (K2's 512-row critic phase
        # 12.0 us, its step 0.9873 -> 0.9772 ms, tools/r4_ab15.sh).  The backward tail replaces FOUR launches (~28 us): 16 384.
        self.policy_tail_max_split = 24576
        self.policy_tail_bwd_max_split = 16384
        self.phase_hook = None     # tests only: called between the critic's optimizer pass and the actor phase of an eager step
        # The switches below were the A/B arms of rounds 3-4 (measurements: DESIGN.md appendix); the kept arm is the default, the other one
        # stays reachable as an attribute for the tests that check both give the same step (tests/test_update_step_gpu.py).
        self.tail_split_rows = 512         # the row-split tail kernels' domain (csrc/headtail.hip kTailSplitMaxRows)
        self.fold_q0 = True                # the Q heads' first layer finished inside the policy tail (A <= fold_max_a)
        self.fold_max_a = 8
        self.attach_colsum = True          # the step's leftover column sums ride on the encoder backward's reduce launch
        self.tail_bwd = True               # pcrl_policy_tail_bwd_f32: four launches of the actor's backward in one
        self.entry_pack = True             # the critic phase's re-pack rides on the replay's sampling launch
        self.publish_first = True          # the metrics leave BEFORE the step's last optimizer pass (the host's turn-around overlaps it)
        self.ln_rider = True               # the feature LayerNorm's backward rides on the encoder backward's prep launch (one node fewer)
        self.gather_rider = True           # the metrics are gathered / published by the first workgroup of the published pass (one node fewer)
        self._entry_cols = None            # (M, group) of a column-gather job attach_entry() has already attached for the next critic phase

    def _policy_tail_fits(self, M, bwd=False):
        split = self.H == 1024 and M <= self.tail_split_rows        # the row-split kernels' domain (headtail.hip: PCRL_TAIL_SPLIT_MAX)
        return M * 2 * self.A <= ((self.policy_tail_bwd_max_split if bwd else self.policy_tail_max_split) if split else self.policy_tail_max)

    def _buf(self, name, *shape, dtype=torch.float32):
        key = (name,) + shape
        if key not in self.bufs:
            self.bufs[key] = torch.zeros(*shape, dtype=dtype, device=self.a._flat["critic"].data.device)
        return self.bufs[key]

    def relu_decisions(self, M, Ma=None):
        """ReLU branch decisions (hidden activation > 0) of the gradient-carrying head passes of the last eager step, for
        the parity tests: {"q": [head][layer], "pi": [layer], "q_pi": [head][layer]} of bool [rows, H] tensors."""
        H = self.H
        out = {"q": [[self.bufs[(n, 2, M, H)][h] > 0 for n in ("q_h1", "q_h2")] for h in range(2)]}
        if Ma is not None and ("pi_h1_a", 1, Ma, H) in self.bufs:
            out["pi"] = [self.bufs[(n, 1, Ma, H)][0] > 0 for n in ("pi_h1_a", "pi_h2_a")]
            out["q_pi"] = [[self.bufs[(n, 2, Ma, H)][h] > 0 for n in ("qa_h1", "qa_h2")] for h in range(2)]
        return out

    # -- pieces -------------------------------------------------------------------------------------
    def _feature_jobs(self, jobs):
        """PointNet.final_mlp: Linear(c3, F) + LayerNorm(F) (pointnet.py:152-153) as the epilogue of the encoder launch
        (pcrl_feature_head): jobs = [(first cloud of the launch, M, tag, dsts, save, cats)] -- dsts = the head-input buffers the
        normalised rows go to, cats = [(src [M / div, n], dst buffer, dst column[, div])] the pass-through columns (robot state,
        replay actions: Visuomotor's torch.cat, visuomotor.py:130-141).  Returns (head for encode_raw, [(xhat, rstd)])."""
        fc, off, F = self.a._flat["critic"], self.off, self.F
        pre = "values.0.backbone.visual_nn.final_mlp."
        out, ranges = [], []
        for begin, M, tag, dsts, save, cats in jobs:
            xhat = self._buf(f"feat_xhat_{tag}", M, F) if save else None
            rstd = self._buf(f"feat_rstd_{tag}", M) if save else None
            pending = [(c[0] if c[0].dtype == torch.float32 else c[0].float(), c[1], c[2], c[1].shape[1], c[3] if len(c) > 3 else 1)
                       for c in cats if c[0] is not None]
            while len(pending) > 2:                      # the kernel takes two pass-through blocks per job
                src, dst, col, _, div = pending.pop()
                dst[:, col:col + src.shape[1]].copy_(torch.repeat_interleave(src, div, dim=0) if div > 1 else src)
            ranges.append((begin, dict(M=M, dsts=dsts, xhat=xhat, rstd=rstd, cats=pending)))
            out.append((xhat, rstd))
        head = hip.make_feature_head(fc.data[off[pre + "0.weight"]:], fc.data[off[pre + "0.bias"]:], fc.data[off[pre + "1.weight"]:],
                                     fc.data[off[pre + "1.bias"]:], F, self.a.encoder.final_mlp[1].eps, ranges)
        return head, out

    def _target_cols_job(self, M):
        a, A, H = self.a, self.A, self.H
        w0a_t = self._buf("q_w0_action_cols_target", 2, A, H)
        return w0a_t, (self.q.W(0, a._target_flat.data, -self.q_base), self.q.hs, 2, H, self.Din_q, self.F + self.S, A, w0a_t)

    def attach_entry(self, M, group=1):
        """Called by the agent right BEFORE the replay's sampling launch of a captured step (M rows, DrQ's group): what heads the critic
        phase -- the encoder's re-pack and, with the first-layer fold, the image of the target heads' action columns -- is handed to that
        launch as extra workgroups (pcrl_encoder_pack_attach_to_gather; it depends on nothing the sampling writes).  The caller follows
        the sampling launch with `hip.encoder_pack_flush_pending()`."""
        if not self.entry_pack:
            return
        if self._fold_fits(M, group):
            hip.pack_attach_cols([self._target_cols_job(M)[1]])
            self._entry_cols = (M, group)
        # (an image that is current takes no job: the attached columns then wait for the critic phase's pack_flush_cols)
        self.a.encoder.attach_pack_to_gather()

    def _fold_fits(self, M, group=1):
        """The Q heads' first layer finished inside the policy tail (pcrl_policy_tail_fwd_fold_f32): where the row-split tail runs."""
        # Measured on MI355X (tools/r4_ab6.sh, same box, initial training state): a rank's 32-cloud share of K1 (A = 6) 0.3318 -> 0.3275 ms,
        # K1 itself 0.8440 -> 0.8459 (neutral: one launch and two graph nodes fewer, the tail 2.3 us longer), K3's 128-cloud share
        # (A = 22: 44 KB of action columns per row) 0.6753 -> 0.6807 -- so only for small action spaces.
        return (self.tails and self.fold_q0 and self.H == 1024 and M <= self.tail_split_rows and self.A <= self.fold_max_a and self._policy_tail_fits(M)
                and group in (1, 2, 4))

    def _actor_forward(self, XA, M, tag, act_dst, ld_act, save, extra_l0=(), fold=None):
        """Actor MLP + TanhGaussianHead mode="max-entropy"; the action goes straight into the Q input.  extra_l0: GEMM descriptors
        that share the launch of the actor's first layer; fold: see hip.policy_tail_fwd."""
        a, A, H = self.a, self.A, self.H
        h1, h2 = self._buf(f"pi_h1_{tag}", 1, M, H), self._buf(f"pi_h2_{tag}", 1, M, H)
        feat = self._buf(f"pi_out_{tag}", M, 2 * A)
        head = a.actor.head
        eps = self._buf(f"pi_eps_{tag}", M, A)
        act = self._buf(f"pi_act_{tag}", M, A)
        nlp = self._buf(f"pi_nlp_{tag}", M)
        saved = self._buf(f"pi_saved_{tag}", M, 2 * A) if save else None
        # the head kernel is one wave per row (or four) streaming all 2 A rows of the last layer from L2: a latency chain that
        # beats GEMM + a separate head launch only while rows x outputs is small (K1: 256 x 12 -> 8.6 us; K3: 1 024 x 44 -> 34.5 us
        # against ~13 us for the two launches)
        if self.tails and self._policy_tail_fits(M):   # two layers as GEMMs, the last one inside the head kernel
            descs = mlp_forward_descs(self.pi, None, 0, XA, self.lda, M, (h1, h2), 2 * A, feat, 0)
            hip.gemm_group([descs[0]] + list(extra_l0))
            hip.gemm_group([descs[1]])
            eps_in = head._standard_normal(eps) if head.noise_override else None
            hip.policy_tail_fwd(h2, M, H, self.pi.W(2), self.pi.Bv(2), A, eps_in, self.seed, a.critic_optim.step_counter,
                                0 if tag == "n" else 1, eps, self.head_scale, self.head_bias, head.log_std_min, head.log_std_max,
                                head.epsilon, feat, act, A, nlp, saved, action2_ptr=act_dst, ld_action2=ld_act, fold=fold)
            return feat, (eps_in if eps_in is not None else eps), saved, nlp, h1, h2
        assert not extra_l0 and fold is None
        mlp_forward(self.pi, None, 0, XA, self.lda, M, (h1, h2), 2 * A, feat, 0)
        if head.noise_override:          # parity tests inject the draws
            eps = head._standard_normal(eps)
            hip.tanh_gaussian_fwd(feat, 2 * A, eps, self.head_scale, self.head_bias, M, A, head.log_std_min, head.log_std_max, head.epsilon,
                                  act, A, nlp, saved, action2_ptr=act_dst, ld_action2=ld_act)
        else:                            # drawn in the kernel; the critic optimizer's device step count advances the stream
            hip.tanh_gaussian_sample_fwd(feat, 2 * A, self.seed, a.critic_optim.step_counter, 0 if tag == "n" else 1, eps,
                                         self.head_scale, self.head_bias, M, A, head.log_std_min, head.log_std_max, head.epsilon,
                                         act, A, nlp, saved, action2_ptr=act_dst, ld_action2=ld_act)
        return feat, eps, saved, nlp, h1, h2

    def _fork_prepare(self, enc, ctx, argmax):
        if self._side is None:
            return False
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self._forked = enc.backward_prepare(ctx, argmax)
        if not self._forked:
            main.wait_stream(self._side)
        return self._forked

    def _join_prepare(self):
        if self._forked:
            torch.cuda.current_stream().wait_stream(self._side)
            self._forked = False

    # -- the step -------------------------------------------------------------------------------------
    def run(self, *args, **kwargs):
        """Execute the step in one go; gradient exchanges (data-parallel) happen inline: `steps` yields ("start", pieces) where
        a range of a flat gradient buffer has become final (its all-reduce then overlaps the rest of the backward) and
        ("finish", pieces) right before an optimizer pass."""
        from ..utils.dist import Exchange
        ex = Exchange(enabled=self.a._be_data_parallel)
        gen = self.steps(*args, **kwargs)
        try:
            kind, pieces = next(gen)
            while True:
                for t in pieces:
                    ex.start(t)
                kind, pieces = gen.send(ex.finish() if kind == "finish" else None)
        except StopIteration as done:
            return done.value

    def steps(self, obs, next_obs, actions, rewards, dones, do_actor, polyak, group=1, actor_obs=None, repeat=1, exchanging=True):
        """Generator form of the step: yields ("start", [pieces]) when a range of a flat gradient buffer is final and its
        all-reduce may begin, and ("finish", [pieces]) when the remaining pieces must be reduced and every started one
        waited for -- it then receives the factor 1/world to fold into the optimizer pass.  Between two yields no
        cross-rank communication is issued, so every stretch can be captured as its own hipGraph while the collectives stay
        eager; a caller without peers passes exchanging=False: no yields at all, the step is one graph."""
        a = self.a
        enc, F, S, A, H = a.encoder, self.F, self.S, self.A, self.H
        ldq, lda = self.ldq, self.lda
        fc, fa = a._flat["critic"], a._flat["actor"]
        split = type(a.actor.backbone).split_obs
        # repeat > 1 (DrQ): obs / next_obs are virtually repeated (AugmentedObs.repeat) and actions / rewards / dones / robot
        # state hold one row per SAMPLE: row m of the step's batch reads entry m // repeat
        M = actions.shape[0] * repeat
        stats = {}

        # ---- target y = r + (1-d) gamma (min_h Q'(s', a') + alpha * (-log pi(a'|s')))  (sac.py:110-134) and q = Q(s, a)
        # (sac.py:136).  Both encoder passes first, then the head GEMMs of the two independent branches pairwise in
        # one launch each (feature Linear of s' and s; target Q heads on s' and online Q heads on s).
        vis_n, state_n = split(next_obs)
        vis_o, state_o = split(obs)
        both = _adjacent_halves(vis_o, vis_n)
        XA_n, XQ_n, XQ_o = self._buf("XA_n", M, lda), self._buf("XQ_n", M, ldq), self._buf("XQ_o", M, ldq)
        job_n = lambda begin: (begin, M, "n", [(XA_n, 0, lda), (XQ_n, 0, ldq)], False, [(state_n, XA_n, F, repeat), (state_n, XQ_n, F, repeat)])
        job_o = lambda begin: (begin, M, "o", [(XQ_o, 0, ldq)], True, [(state_o, XQ_o, F, repeat), (actions, XQ_o, F + S, repeat)])
        # The Q heads' first layer on (s', a') is finished by the policy tail (it needs a' only through the action columns): the compact
        # image of the TARGET heads' action columns rides on the re-pack launch that heads this phase
        fold_c = self._fold_fits(M, group)
        tgt = a._target_flat.data
        if fold_c:
            w0a_t, job = self._target_cols_job(M)
            if self._entry_cols != (M, group):       # (else: attach_entry() put it on the replay's sampling launch)
                hip.pack_attach_cols([job])
        self._entry_cols = None
        if both is not None:              # s and s' sit back to back (DeviceReplay's staging): one launch of 2 M clouds
            head, ((xhat, rstd), _) = self._feature_jobs([job_o(0), job_n(M)])
            pooled_all, argmax_all, _ = enc.encode_raw(both, head=head)
            Mo = pooled_all.shape[0] // 2
            pooled_o, pooled_n, argmax_o = pooled_all[:Mo], pooled_all[Mo:], argmax_all[:Mo]
            ctx_o = enc.ctx_for(vis_o, pooled_o)
        else:
            head_n, _ = self._feature_jobs([job_n(0)])
            pooled_n, _, _ = enc.encode_raw(vis_n, head=head_n)
            head_o, ((xhat, rstd),) = self._feature_jobs([job_o(0)])
            pooled_o, argmax_o, ctx_o = enc.encode_raw(vis_o, head=head_o)
        self.last_argmax = argmax_o          # read by the parity tests (first-index argmax of the gradient-carrying pass)
        self.last_pooled = pooled_o          # (with it: which channels are live -- bench.py counts the backward's active points and tiles)
        if fold_c:
            hip.pack_flush_cols()            # (a phase whose weights needed no re-pack: the gather as a launch of its own)
        prepared = self._fork_prepare(enc, ctx_o, argmax_o)
        qn_h1, qn_h2 = self._buf("qn_h1", 2, M, H), self._buf("qn_h2", 2, M, H)
        q_next = self._buf("q_next", M, 2)
        q_h1, q_h2 = self._buf("q_h1", 2, M, H), self._buf("q_h2", 2, M, H)
        q = self._buf("q", M, 2)
        q_tgt_descs = mlp_forward_descs(self.q, tgt, -self.q_base, XQ_n, ldq, M, (qn_h1, qn_h2), 2, q_next, 1)
        q_on_descs = mlp_forward_descs(self.q, None, 0, XQ_o, ldq, M, (q_h1, q_h2), 2, q, 1)
        if fold_c:
            # one launch: the actor's first layer on s', the online heads' first layer on (s, a) -- neither needs the actor's output --
            # and the action-free part of the target heads' first layer, pre = [feature | state] W0[:, :F+S]^T + b0 (no ReLU yet)
            hs, k0 = self.q.hs, self.Din_q
            pre_t = hip.gemm_desc(XQ_n, self.q.W(0, tgt, -self.q_base), qn_h1, M, H, F + S, (ldq, 1), (1, k0), H,
                                  bias=self.q.Bv(0, tgt, -self.q_base), batch=2, batch_strides=(0, hs, M * H, hs, 0))
            _, _, _, nlp_n, _, _ = self._actor_forward(XA_n, M, "n", XQ_n.data_ptr() + 4 * (F + S), ldq, save=False, extra_l0=[pre_t, q_on_descs[0]],
                                                       fold=(qn_h1, M * H, w0a_t, A * H, 2, qn_h1, M * H))
        else:
            _, _, _, nlp_n, _, _ = self._actor_forward(XA_n, M, "n", XQ_n.data_ptr() + 4 * (F + S), ldq, save=False)
        q_target, dq = self._buf("q_target", M), self._buf("dq", M, 2)
        dones_u8 = dones.view(torch.uint8) if dones.dtype == torch.bool else dones.to(torch.uint8)
        reward_scale = a.reward_scale if a.metric_prefix == "sac" else 1.0
        dh1, dh2 = self._buf("q_dh1", 2, M, H), self._buf("q_dh2", 2, M, H)
        dX0 = self._buf("q_dX0", 2, M, ceil4(F))
        off, pre = self.off, "values.0.backbone.visual_nn.final_mlp."
        dy = self._buf("feat_dy", M, F)
        ws = self._buf("ln_ws", ((M + 3) // 4) * 2 * F)
        bwd_stages = mlp_backward_descs(self.q, XQ_o, ldq, M, q_h1, q_h2, dq, (2, 1), 1, dh1, dh2, grad=fc.grad, dX=dX0, dx_cols=(0, F),
                                        ld_dx=ceil4(F))
        assert not fold_c or (self.tails and group in (1, 2, 4))
        if self.tails and group in (1, 2, 4):
            # ---- two layers as GEMMs; the last layer of the four heads, the TD target / critic loss (sac.py:125-157) and the first
            # backward stage in ONE launch; its per-workgroup partials (dW2, db2, the logged statistics) are reduced by the same
            # column-sum launch that finishes the feature LayerNorm's backward ----
            if fold_c:
                launch_layers(q_tgt_descs[1:2], q_on_descs[1:2])
            else:
                launch_layers(q_tgt_descs[:2], q_on_descs[:2])
            n_part, n_stat = hip.q_tail_workspace_floats(M, H)
            part, stat_part = self._buf("q_tail_part", n_part), self._buf("q_tail_stat", n_stat)
            hs = self.q.hs
            hip.q_tail_critic(q_h2, M * H, self.q.W(2), self.q.Bv(2), hs, qn_h2, M * H, self.q.W(2, tgt, -self.q_base),
                              self.q.Bv(2, tgt, -self.q_base), hs, nlp_n, rewards, dones_u8, repeat, a.log_alpha, a.gamma, reward_scale,
                              a.ignore_dones, group, M, H, q, q_target, dq, dh2, part, stat_part)
            launch_layers(*[bwd_stages[1:]])
            # The feature LayerNorm's backward (4.9 us of launch for 256 x 50 values) as extra workgroups of the encoder backward's prep
            # launch, which needs only the forward's outputs and is brought forward to here: prep | LayerNorm -> feature GEMMs -> points ...
            if self.ln_rider and not prepared and enc.can_prepare(ctx_o):
                hip.encoder_bwd_attach_ln_bwd(dX0.data_ptr(), dX0.data_ptr() + 4 * M * ceil4(F), ceil4(F), xhat, rstd,
                                              fc.data[off[pre + "1.weight"]:], M, F, dy, F, ws)
                prepared = enc.backward_prepare(ctx_o, argmax_o)
                assert prepared
            else:
                hip.layernorm_rows_bwd_partials(dX0.data_ptr(), dX0.data_ptr() + 4 * M * ceil4(F), ceil4(F), xhat, rstd,
                                                fc.data[off[pre + "1.weight"]:], M, F, dy, F, ws)
            n_wg, n_ln, Hp = (M + 3) // 4, (M + 3) // 4, H + 4
            g = fc.grad.data_ptr()
            jobs = [(ws.data_ptr(), 2 * F, n_ln, F, g + 4 * off[pre + "1.weight"], 1.0, 0),
                    (ws.data_ptr() + 4 * F, 2 * F, n_ln, F, g + 4 * off[pre + "1.bias"], 1.0, 0)]
            for h in range(2):
                jobs.append((part.data_ptr() + 4 * h * Hp, 2 * Hp, n_wg, H, g + 4 * (self.q.w[2] + h * hs), 1.0, 0))
                jobs.append((part.data_ptr() + 4 * (h * Hp + H), 2 * Hp, n_wg, 1, g + 4 * (self.q.b[2] + h * hs), 1.0, 0))
            sc = self.stats_c.data_ptr()
            for k, (scale_k, op) in enumerate(((1.0 / M, 0), (1.0, 1), (1.0 / M, 0), (1.0 / M, 0))):
                jobs.append((stat_part.data_ptr() + 4 * k, 4, n_wg, 1, sc + 4 * k, scale_k, op))
            # nothing before the optimizer reads these sums: they ride on the encoder backward's reduce launch (one node fewer) --
            # except data-parallel, where the Q heads' range (dW2 | db2 among these sums) starts its all-reduce before that launch
            hip.colsum_jobs(jobs, attach_to_encoder_bwd=self.attach_colsum and not a._be_data_parallel)
        else:
            launch_layers(q_tgt_descs, q_on_descs)
            # ---- critic loss, backward through heads, feature head and encoder (sac.py:137-148) ----
            hip.sac_critic_loss(q_next, 2, nlp_n, rewards, dones_u8, a.log_alpha, a.gamma, reward_scale, a.ignore_dones, group, q, 2, M, 2,
                                q_target, dq, 2, self.stats_c, rd_row_div=repeat)
            launch_layers(*[bwd_stages])
            hip.layernorm_rows_bwd(dX0.data_ptr(), dX0.data_ptr() + 4 * M * ceil4(F), ceil4(F), xhat, rstd, fc.data[off[pre + "1.weight"]:], M, F,
                                   dy, F, fc.grad[off[pre + "1.weight"]:], fc.grad[off[pre + "1.bias"]:], ws)
        # The Q heads' gradients (the tail of the critic's flat buffer: 2 x 1.1 M floats at K1 of 2.27 M) are final: their
        # all-reduce runs under the feature-head and encoder backward, only the small head of the buffer waits for those.
        if exchanging:
            self._join_prepare()             # a segment of the step (one graph each) must not end with an open branch
            yield ("start", [fc.grad[self.q_base:]])
        c3 = self.c3
        dpooled = self._buf("dpooled", M, c3)
        hip.gemm_group([hip.gemm_desc(dy, pooled_o, fc.grad[off[pre + "0.weight"]:], F, c3 + 1, M, (1, F), (c3, 1), c3, ones_col=c3,
                                      c_ones=fc.grad[off[pre + "0.bias"]:]),
                        hip.gemm_desc(dy, fc.data[off[pre + "0.weight"]:], dpooled, M, c3, F, (F, 1), (c3, 1), c3)])
        self._join_prepare()
        enc.backward_raw(ctx_o, argmax_o, dpooled, fc.grad[:self.n_conv], prepared=prepared)
        scale = (yield ("finish", [fc.grad[:self.q_base]])) if exchanging else 1.0
        pending = []          # optimizer passes whose gradient norm / step count are finished by the end-of-step gather launch
        # The step's LAST optimizer pass goes behind the launch that publishes the metrics (a 4 B/parameter norm launch ahead of it instead):
        # the host reads the metrics, returns and queues the next step while that pass still runs -- 12-14 us of idle device per step otherwise
        # (profiles/r05_timeline_k1_b32.txt).  In a critic-only step that pass is the critic's, else the actor's.
        last_pass = None
        if self.publish_first and not do_actor:
            split = a._optim_norm_first("critic", scale, pending)
            if split is not None:
                stats["critic_grad"], finish = split
                last_pass = lambda gather=None: finish(polyak, gather)
        if last_pass is None:
            stats["critic_grad"] = a._optim_step("critic", scale, polyak=polyak, pending=pending)   # also invalidates enc's packed image
        stats.update(critic_loss=self.stats_c[0], max_critic_abs_err=self.stats_c[1], q=self.stats_c[2], q_target=self.stats_c[3])

        # ---- actor + temperature (sac.py:161-205) --------------------------------------------------------
        if do_actor and self.phase_hook is not None:
            # tests only (tests/test_fullsize_parity_gpu.py): lets the caller put another implementation's post-critic-step parameters
            # in place, so that the actor phase is compared from ONE state; eager steps only -- a captured graph would freeze what it does
            assert not torch.cuda.is_current_stream_capturing()
            self.phase_hook()
        if do_actor:
            a_obs = obs if actor_obs is None else actor_obs
            vis_a, state_a = split(a_obs)
            Ma = M if actor_obs is None else vis_a["xyz"].shape[0]
            XA_a, XQ_a = self._buf("XA_a", Ma, lda), self._buf("XQ_a", Ma, ldq)
            head_a, _ = self._feature_jobs([(0, Ma, "a", [(XA_a, 0, lda), (XQ_a, 0, ldq)], False, [(state_a, XA_a, F), (state_a, XQ_a, F)])])
            fold_a = self._fold_fits(Ma)
            w0a = self._buf("q_w0_action_cols", 2, A, H)
            if fold_a:                       # the ONLINE heads' action columns (just updated by the critic's optimizer) ride on this phase's re-pack
                hip.pack_attach_cols([(self.q.W(0), self.q.hs, 2, H, self.Din_q, F + S, A, w0a)])
            pooled_a, _, _ = enc.encode_raw(vis_a, head=head_a)      # updated encoder weights, no gradient
            qa_h1, qa_h2 = self._buf("qa_h1", 2, Ma, H), self._buf("qa_h2", 2, Ma, H)
            if fold_a:
                hip.pack_flush_cols()
                pre_a = hip.gemm_desc(XQ_a, self.q.W(0), qa_h1, Ma, H, F + S, (ldq, 1), (1, self.Din_q), H, bias=self.q.Bv(0), batch=2,
                                      batch_strides=(0, self.q.hs, Ma * H, self.q.hs, 0))
                feat, eps, saved, nlp, p_h1, p_h2 = self._actor_forward(XA_a, Ma, "a", XQ_a.data_ptr() + 4 * (F + S), ldq, save=True, extra_l0=[pre_a],
                                                                        fold=(qa_h1, Ma * H, w0a, A * H, 2, qa_h1, Ma * H))
            else:
                feat, eps, saved, nlp, p_h1, p_h2 = self._actor_forward(XA_a, Ma, "a", XQ_a.data_ptr() + 4 * (F + S), ldq, save=True)
            q_pi = self._buf("q_pi", Ma, 2)
            dq_pi = self._buf("dq_pi", Ma, 2)
            fal = a._flat["alpha"]
            da_h1, da_h2 = self._buf("qa_dh1", 2, Ma, H), self._buf("qa_dh2", 2, Ma, H)
            d_act = self._buf("d_act", 2, Ma, ceil4(A))
            tail_bwd_done = False
            qa_descs = mlp_forward_descs(self.q, None, 0, XQ_a, ldq, Ma, (qa_h1, qa_h2), 2, q_pi, 1)
            qa_bwd = mlp_backward_descs(self.q, XQ_a, ldq, Ma, qa_h1, qa_h2, dq_pi, (2, 1), 1, da_h1, da_h2, grad=None, dX=d_act,
                                        dx_cols=(F + S, A), ld_dx=ceil4(A))
            head = a.actor.head
            dfeat = self._buf("pi_dfeat", Ma, 2 * A)
            dp_h1, dp_h2 = self._buf("pi_dh1", 1, Ma, H), self._buf("pi_dh2", 1, Ma, H)
            if self.tails and self.tail_bwd and self._policy_tail_fits(Ma, bwd=True):
                # ---- the chain q tail -> dh1 GEMM -> ONE launch for [d_act GEMM, TanhGaussianHead backward, the policy's dh2 GEMM,
                # actor_finalize] -> the policy's two remaining backward stages (its last layer's dW2 | db2 rides in the first) ----
                launch_layers(qa_descs[1:2] if fold_a else qa_descs[:2])
                _, n_stat = hip.q_tail_workspace_floats(Ma, H)
                stat_a = self._buf("qa_tail_stat", n_stat)
                if fold_a:                   # the action-column image is already there (this phase's re-pack launch)
                    hip.q_tail_actor(qa_h2, Ma * H, self.q.W(2), self.q.Bv(2), self.q.hs, nlp, a.log_alpha, Ma, H, q_pi, dq_pi, da_h2, self.d_nlp, stat_a)
                else:
                    hip.q_tail_actor_cols(qa_h2, Ma * H, self.q.W(2), self.q.Bv(2), self.q.hs, nlp, a.log_alpha, Ma, H, q_pi, dq_pi, da_h2, self.d_nlp,
                                          stat_a, self.q.W(0), self.q.hs, self.Din_q, F + S, A, w0a)
                launch_layers(*[qa_bwd[1:2]])
                hip.policy_tail_bwd(da_h1, Ma * H, w0a, A * H, Ma, H, A, feat, 2 * A, eps, saved, self.head_scale, head.log_std_min, head.log_std_max,
                                    head.epsilon, self.d_nlp, dfeat, 2 * A, p_h2, self.pi.W(2), dp_h2,
                                    finalize=(stat_a, a.log_alpha, a.target_entropy, fal.grad, self.stats_a))
                pb = mlp_backward_descs(self.pi, XA_a, lda, Ma, p_h1, p_h2, dfeat, (2 * A, 1), 0, dp_h1, dp_h2, grad=fa.grad)
                hip.gemm_group([pb[0][0]] + pb[1])          # dW2 | db2 of the last layer next to [dW1 | db1, dh1]
                hip.gemm_group(pb[2])
                tail_bwd_done = True
            elif self.tails:
                launch_layers(qa_descs[1:2] if fold_a else qa_descs[:2])
                _, n_stat = hip.q_tail_workspace_floats(Ma, H)
                stat_a = self._buf("qa_tail_stat", n_stat)
                hip.q_tail_actor(qa_h2, Ma * H, self.q.W(2), self.q.Bv(2), self.q.hs, nlp, a.log_alpha, Ma, H, q_pi, dq_pi, da_h2, self.d_nlp, stat_a)
                hip.actor_finalize(stat_a, Ma, a.log_alpha, a.target_entropy, fal.grad, self.stats_a)
                launch_layers(*[qa_bwd[1:]])
            else:
                launch_layers(qa_descs)
                hip.sac_actor_loss(q_pi, 2, nlp, a.log_alpha, a.target_entropy, Ma, 2, dq_pi, 2, self.d_nlp, fal.grad, self.stats_a)
                launch_layers(*[qa_bwd])
            if not tail_bwd_done:
                hip.tanh_gaussian_bwd(feat, 2 * A, eps, saved, self.head_scale, Ma, A, head.log_std_min, head.log_std_max, head.epsilon,
                                      d_act.data_ptr(), d_act.data_ptr() + 4 * Ma * ceil4(A), ceil4(A), self.d_nlp, dfeat, 2 * A)
                mlp_backward(self.pi, XA_a, lda, Ma, p_h1, p_h2, dfeat, (2 * A, 1), 0, dp_h1, dp_h2, grad=fa.grad)
            scale = (yield ("finish", [a._actor_alpha_grad] if a.sync_alpha else [fa.grad])) if exchanging else 1.0
            # the temperature (one float, its own betas / moments / step count) rides on the actor's optimizer launch -- or, with the
            # actor's pass behind the metrics, on the norm launch ahead of them (alpha = exp(log_alpha) is one of the metrics)
            split = a._optim_norm_first("actor", scale, pending, rider=("alpha", scale if a.sync_alpha else 1.0)) if self.publish_first else None
            if split is not None:
                stats["actor_grad"], finish_a = split
                last_pass = lambda gather=None: finish_a(False, gather)
            else:
                stats["actor_grad"] = a._optim_step("actor", scale, pending=pending, rider=("alpha", scale if a.sync_alpha else 1.0))
            stats.update(actor_loss=self.stats_a[0], entropy=self.stats_a[1], alpha_loss=self.stats_a[2], new_alpha=None)
        # one launch gathers every reported scalar (and alpha = exp(log_alpha), sac.py:196) into one array
        names = list(stats.keys())
        out = self._buf("stats_out", 16)
        entries = [(a.log_alpha, out[i:], True) if k == "new_alpha" else (stats[k], out[i:], False) for i, k in enumerate(names)]
        if "new_alpha" in stats:
            entries.append((a.log_alpha, a._alpha_t, True))
        if self.host_stats is None:
            self.host_stats = torch.zeros(32, dtype=torch.float32).pin_memory()
        if last_pass is not None and self.gather_rider:
            last_pass((entries, pending, self.host_stats))      # the pass's first workgroup gathers and publishes
        else:
            hip.gather_scalars(entries, pending=pending, host_out=self.host_stats)
            if last_pass is not None:
                last_pass()
        packed = PackedStats((k, out[i]) for i, k in enumerate(names))
        packed.packed = out[:len(names)]
        packed.host = self.host_stats
        return packed
