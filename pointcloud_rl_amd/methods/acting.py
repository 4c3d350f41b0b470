"""Acting path on libpcrl_hip.so: observation -> action in five launches.

The reference's `BaseAgent.forward` (pyrl/utils/torch/module_utils.py:147-159) runs the actor module tree --
`Visuomotor.forward` (visuomotor.py:56-146) -> `PointNet.forward` -> `LinearMLP` -> `TanhGaussianHead.forward`
(gaussian.py:83-87) -- which on a GPU is ~25 small ATen launches per environment step (`Rollout.forward_with_policy`,
rollout.py:78-114, B = 1 ... num_envs).  Here the same function is: [re-pack] -> fused encoder with the feature Linear + LayerNorm
(+ robot state) as its epilogue (pcrl_feature_head) -> two dense layers -> last layer + squashed-Gaussian head.  Used by SAC / DrQ `forward` when the
actor has the topology of the shipped point-cloud configs; every other case falls back to the module tree.
"""
import torch

from .. import hip
from ..networks.heads import TanhGaussianHead
from ..networks.mlp import LinearMLP
from ..networks.pointnet import PointNet, batch_rows

# TanhGaussianHead modes whose result is one action tensor (regression_base.py:50-74): a sample, or the mean action
SAMPLE_MODES = ("explore", "sample")
MEAN_MODES = ("eval", "mean")


def ceil4(x):
    return (x + 3) & ~3


class FusedActor:
    @staticmethod
    def supported(actor):
        bb = getattr(actor, "backbone", None)
        enc, mlp, head = getattr(bb, "visual_nn", None), getattr(bb, "final_mlp", None), getattr(actor, "head", None)
        if not isinstance(enc, PointNet) or enc.final_mlp is None or not isinstance(mlp, LinearMLP) or not isinstance(head, TanhGaussianHead):
            return False
        if actor.final_mlp is not None or bb.ac_feat is not None or bb.obs_feat is not None:
            return False
        if enc.mlp_spec[-1] > 256:          # the feature-head epilogue of the encoder launch is built for c3 <= 256
            return False
        lin = mlp.linears
        if len(lin) != 3 or any(l.bias is None for l in lin):
            return False
        H, A = lin[0].out_features, head.dim_output
        return lin[1].in_features == H and lin[1].out_features == H and H % 256 == 0 and H <= 1024 and 2 * A <= 64 and lin[2].out_features == 2 * A

    def __init__(self, actor):
        self.actor = actor
        self.bufs = {}
        # hipGraph replay of the whole call (`use_graphs`, switched on by agent.enable_graphs()): per (mode, observation
        # signature) the observation is copied into static tensors, one graph launch replaces the five launches and their
        # Python / ctypes cost, the action is returned as a fresh tensor.  The graph starts with the encoder re-pack, so
        # parameters updated in place between two calls are always the ones used.
        self.use_graphs = False
        self.graphs, self._seen = {}, {}

    def _graph_key(self, obs, mode):
        """None when the call cannot be replayed: nested / non-tensor / pending-augmentation observations, host tensors."""
        if not isinstance(obs, dict) or getattr(obs, "aug", None) or getattr(obs, "repeat", 1) != 1 or self.actor.head.noise_override:
            return None
        sig = []
        for k in sorted(obs):
            v = obs[k]
            if not (torch.is_tensor(v) and v.is_cuda):
                return None
            sig.append((k, tuple(v.shape), v.dtype))
        bb = self.actor.backbone
        # storage of the parameters the graph reads: a re-homed parameter (flat buffers of the first update, .to(), a loaded
        # checkpoint that replaced tensors) must not be read through a stale graph
        ptrs = (bb.visual_nn.conv.mlp.conv0.weight.data_ptr(), bb.visual_nn.final_mlp[0].weight.data_ptr(),
                bb.final_mlp.linears[0].weight.data_ptr(), bb.final_mlp.linears[2].bias.data_ptr())
        return (mode in SAMPLE_MODES, tuple(sig), ptrs)

    def _replay(self, key, obs, mode):
        entry = self.graphs.get(key)
        if entry is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n < 2:                                   # eager warm-up: buffers and lazy initialisation outside the capture
                return None
            if len(self.graphs) >= 8:                   # stale parameter storages / many shapes: start over
                self.graphs.clear()
            static = {k: v.clone() for k, v in obs.items()}
            self.actor.backbone.visual_nn.invalidate_packed()        # the capture records the re-pack as the first node
            torch.cuda.synchronize()
            from .sac import _no_gc                     # no Python garbage collection inside a stream capture
            graph = torch.cuda.CUDAGraph()
            with _no_gc(), torch.cuda.graph(graph):
                out = self._run(static, mode)
            entry = self.graphs[key] = (graph, static, out)
        graph, static, out = entry
        for k, v in obs.items():
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        return out.clone()

    @torch.no_grad()
    def __call__(self, obs, mode="explore"):
        """obs: observation dict on the device; returns the action tensor [B, A]."""
        if self.use_graphs:
            key = self._graph_key(obs, mode)
            if key is not None:
                act = self._replay(key, obs, mode)
                if act is not None:
                    return act
        return self._run(obs, mode)

    def _buf(self, name, *shape):
        key = (name,) + shape
        dev = self.actor.head.scale.device if torch.is_tensor(self.actor.head.scale) else next(self.actor.parameters()).device
        if key not in self.bufs or self.bufs[key].device != dev:
            self.bufs[key] = torch.zeros(*shape, dtype=torch.float32, device=dev)
        return self.bufs[key]

    @torch.no_grad()
    def _run(self, obs, mode="explore"):
        actor = self.actor
        bb, head = actor.backbone, actor.head
        enc, lin = bb.visual_nn, bb.final_mlp.linears
        visual, state = type(bb).split_obs(obs)
        M = batch_rows(visual) if isinstance(visual, dict) else visual.shape[0]
        fl, ln = enc.final_mlp[0], enc.final_mlp[1]
        F, c3, S = fl.out_features, fl.in_features, 0 if state is None else state.shape[-1]
        A, H = head.dim_output, lin[0].out_features
        assert lin[0].in_features == F + S, f"actor MLP expects {lin[0].in_features} inputs, got {F} + {S}"
        lda = ceil4(F + S)
        XA = self._buf("XA", M, lda)
        cats = [] if state is None else [(state.float().contiguous(), XA, F, lda)]
        # final_mlp (Linear + LayerNorm) and the robot-state columns are written by the encoder launch itself (pcrl_feature_head)
        fh = hip.make_feature_head(fl.weight, fl.bias, ln.weight, ln.bias, F, ln.eps, [(0, dict(M=M, dsts=[(XA, 0, lda)], cats=cats))])
        enc.encode_raw(visual, head=fh)
        h1, h2 = self._buf("h1", M, H), self._buf("h2", M, H)
        k0 = F + S
        hip.gemm(XA, lin[0].weight, h1, M, H, k0, (lda, 1), (1, k0), H, bias=lin[0].bias, relu=True)
        hip.gemm(h1, lin[1].weight, h2, M, H, H, (H, 1), (1, H), H, bias=lin[1].bias, relu=True)
        # mean action = the sampled action with zero noise: tanh(mean + std * 0) * scale + bias
        eps = head._standard_normal(self._buf("eps", M, A)) if mode in SAMPLE_MODES else self._buf("zeros", M, A)
        scale, bias = self._scale_bias(A)
        feat, act, nlp = self._buf("feat", M, 2 * A), torch.empty(M, A, dtype=torch.float32, device=XA.device), self._buf("nlp", M)
        hip.policy_tail_fwd(h2, M, H, lin[2].weight, lin[2].bias, A, eps.contiguous(), 0, None, 0, self._buf("eps_out", M, A), scale, bias,
                            head.log_std_min, head.log_std_max, head.epsilon, feat, act, A, nlp)
        return act

    def _scale_bias(self, A):
        head = self.actor.head
        key = ("scale_bias", A)
        src = (head.scale.data_ptr(), head.bias.data_ptr()) if torch.is_tensor(head.scale) else (None, None)
        if key not in self.bufs or self.bufs[key][2] != src:
            dev = next(self.actor.parameters()).device

            def f32(v):
                if torch.is_tensor(v):
                    return v.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous().clone()
                return torch.full((A,), float(v), dtype=torch.float32, device=dev)
            self.bufs[key] = (f32(head.scale), f32(head.bias), src)
        return self.bufs[key][0], self.bufs[key][1]
