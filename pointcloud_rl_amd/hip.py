"""Host-side launchers: torch CUDA tensors -> raw pointers -> C ABI (include/pcrl.h).

PyTorch is used for device memory and the current HIP stream only; every computation here is a
call into libpcrl_hip.so.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import AugDesc, CloudDesc, EncoderWeights, FeatSeg, GemmDesc, check, lib

_DT = {torch.float32: _lib.DT_F32, torch.uint8: _lib.DT_U8, torch.bool: _lib.DT_BOOL}


class KernelTimer:
    """Optional HIP-event timing of every C-ABI launch (bench.py's roofline figures).  Events are
    recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.spans = {}
        self.flops = {}            # name -> FLOPs issued under that span name (the GEMM launches count 2 M N K per problem)

    def span(self, name):
        return _Span(self, name)

    def summary(self):
        """name -> (launches, mean ms); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.spans.items() if v}

    def detail(self):
        """name -> dict(min_ms, median_ms, max_ms): a mean far from the median marks spans that held more than
        their kernel (an eager pass the host cannot keep ahead of, a launch queued behind another stream's work)."""
        out = {}
        for k, v in self.spans.items():
            if v:
                ms = sorted(a.elapsed_time(b) for a, b in v)
                out[k] = dict(min_ms=ms[0], median_ms=ms[len(ms) // 2], max_ms=ms[-1])
        return out


class _Span:
    def __init__(self, timer, name):
        self.timer, self.name = timer, name

    def __enter__(self):
        self.start = torch.cuda.Event(enable_timing=True)
        self.start.record()

    def __exit__(self, *exc):
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        self.timer.spans.setdefault(self.name, []).append((self.start, end))


class _NoSpan:
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        pass


TIMER = None          # set to a KernelTimer() to time launches


def _span(name):
    return TIMER.span(name) if TIMER is not None else _NoSpan()


def raw_stream():
    """hipStream_t of torch's current stream as an int (the C calls skip the torch.cuda.Stream object: ~8 us -> < 1 us)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream():
    return ctypes.c_void_p(raw_stream())


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def make_cloud_desc(obs):
    """Describe the observation dict PointCloudBase.preprocess consumes (reference
    pyrl/networks/backbones/pointnet.py:49-73): keys xyz [B,3,N] f32, rgb [B,3,N] u8|f32,
    pos_encoding [B,F,N], seg [B,K,N]; or a bare xyz tensor.  Returns (desc, tensors kept alive)."""
    repeat = int(getattr(obs, "repeat", 1) or 1)      # VirtualRepeat: every stored cloud is seen `repeat` times (DrQ)
    if torch.is_tensor(obs):
        obs = {"xyz": obs}
    keep, segs = [], []
    xyz = obs["xyz"]
    if not xyz.is_cuda:
        raise RuntimeError(f"point clouds must live on the MI355X (got device {xyz.device}); pointcloud_rl_amd has no CPU encoder")
    assert xyz.ndim == 3, f"xyz must be a [B,C,N] tensor, got {tuple(xyz.shape)}"
    B, _, N = xyz.shape
    for key in ("xyz", "rgb", "pos_encoding", "seg"):
        if key not in obs:
            continue
        t = obs[key]
        assert t.is_cuda and t.ndim == 3 and t.shape[0] == B and t.shape[2] == N, f"{key}: bad shape {tuple(t.shape)}"
        if t.dtype not in _DT:
            t = t.to(torch.float32)
        div255 = 1 if (key == "rgb" and t.dtype == torch.uint8) else 0
        s = FeatSeg(ptr=t.data_ptr(), dtype=_DT[t.dtype], channels=t.shape[1], div255=div255,
                    stride_b=t.stride(0), stride_c=t.stride(1), stride_n=t.stride(2))
        segs.append(s)
        keep.append(t)
    desc = CloudDesc(B=B * repeat, N=N, nseg=len(segs), row_div=repeat)
    for i, s in enumerate(segs):
        desc.seg[i] = s
    return desc, keep


def make_interleaved_desc(points):
    """[B, N, C] f32 point tensor (the synthetic benchmark layout of BASELINE.json)."""
    assert points.is_cuda and points.ndim == 3 and points.dtype == torch.float32
    B, N, C = points.shape
    desc = CloudDesc(B=B, N=N, nseg=1)
    desc.seg[0] = FeatSeg(ptr=points.data_ptr(), dtype=_lib.DT_F32, channels=C, div255=0,
                          stride_b=points.stride(0), stride_c=points.stride(2), stride_n=points.stride(1))
    return desc, [points]


def make_encoder_weights(w0, b0, w1, g1, be1, w2, g2, be2, eps):
    ts = [w0, b0, w1, g1, be1, w2, g2, be2]
    for t in ts:
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    c1, c_in = w0.shape[0], w0.shape[1]
    c2, c3 = w1.shape[0], w2.shape[0]
    ew = EncoderWeights(c_in=c_in, c1=c1, c2=c2, c3=c3, eps=eps,
                        w0=w0.data_ptr(), b0=b0.data_ptr(), w1=w1.data_ptr(), g1=g1.data_ptr(), be1=be1.data_ptr(),
                        w2=w2.data_ptr(), g2=g2.data_ptr(), be2=be2.data_ptr())
    return ew, ts


def encoder_packed_bytes(c_in, c1, c2, c3):
    n = ctypes.c_size_t()
    check(lib().pcrl_encoder_packed_bytes(c_in, c1, c2, c3, ctypes.byref(n)))
    return n.value


def encoder_pack_weights(ew, packed):
    check(lib().pcrl_encoder_pack_weights_f32(ctypes.byref(ew), _ptr(packed), ctypes.c_size_t(packed.numel() * packed.element_size()), _stream()))


def encoder_pack_attach_to_gather(ew, packed):
    """The re-pack (with the column-gather jobs attached so far) rides on this thread's NEXT replay sampling launch
    (pcrl_encoder_pack_attach_to_gather); follow the sampling launch with encoder_pack_flush_pending()."""
    check(lib().pcrl_encoder_pack_attach_to_gather(ctypes.byref(ew), _ptr(packed), ctypes.c_size_t(packed.numel() * packed.element_size())))


def encoder_pack_drop_pending():
    """Forgets a pending pack job without running it (an aborted capture): the caller invalidates the image it stood for."""
    check(lib().pcrl_encoder_pack_drop_pending())


def encoder_pack_flush_pending():
    """Launches a pack job no sampling launch has taken; no-op otherwise (pcrl_encoder_pack_flush_pending)."""
    check(lib().pcrl_encoder_pack_flush_pending(_stream()))


def make_aug_desc(jitter_noise=None, jitter_range=None, seed=0, offset=0, affine=None, row_mul=1, row_add=0, offset_tensor=None,
                  point_index=None, color=None, point_count=None):
    """point_count: device int32 [1] tensor -- how many of point_index's positions exist this call (RandomDownSample with a random
    count; read by the kernels at run time, see include/pcrl.h n_index_ptr).
    color: dict(order=[4 step ids in application order], factors=[brightness, contrast, saturation, hue] (None: skip),
    mean=float32 device tensor [stored clouds] or None) -- ColorJitterPoints, see include/pcrl.h PCRL_AUG_COLOR."""
    flags = 0
    aug = AugDesc()
    aug.row_mul, aug.row_add = int(row_mul), int(row_add)
    if jitter_noise is not None or jitter_range is not None:
        flags |= _lib.AUG_JITTER
        aug.jitter_noise = jitter_noise.data_ptr() if jitter_noise is not None else None
        if jitter_range is not None:
            aug.jitter_lo, aug.jitter_hi = float(jitter_range[0]), float(jitter_range[1])
        aug.seed, aug.offset = int(seed), int(offset)
        if offset_tensor is not None:
            assert offset_tensor.dtype == torch.int64 and offset_tensor.is_cuda
            aug.offset_ptr = offset_tensor.data_ptr()
    if affine is not None:
        flags |= _lib.AUG_AFFINE
        aug.affine = affine.data_ptr()
    if point_index is not None:
        assert point_index.dtype == torch.int32 and point_index.is_cuda and point_index.is_contiguous() and point_index.ndim == 1
        flags |= _lib.AUG_SUBSAMPLE
        aug.point_index, aug.n_index = point_index.data_ptr(), point_index.numel()
        if point_count is not None:
            assert point_count.dtype == torch.int32 and point_count.is_cuda and point_count.numel() == 1
            aug.n_index_ptr = point_count.data_ptr()
    if color is not None:
        flags |= _lib.AUG_COLOR
        order = 0
        for k, op in enumerate(color["order"]):
            f = color["factors"][op]
            order |= (15 if f is None else int(op)) << (4 * k)
            if f is not None:
                aug.color_factor[op] = float(f)
                aug.color_one_minus[op] = 1.0 - float(f)       # formed in double, rounded to float by the struct: as torch does
        aug.color_order = order
        mean = color.get("mean")
        if mean is not None:
            assert mean.dtype == torch.float32 and mean.is_cuda and mean.is_contiguous()
            aug.color_mean = mean.data_ptr()
    aug.flags = flags
    return aug


def color_contrast_mean(rgb, color):
    """Per-cloud grayscale mean entering the contrast step of a ColorJitterPoints draw (rgb [B,3,N] uint8); None when the
    draw has no contrast step."""
    assert rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.ndim == 3 and rgb.shape[1] == 3
    if color["factors"][1] is None:
        return None
    desc = make_aug_desc(color=dict(color, mean=None))
    out = torch.empty(rgb.shape[0], dtype=torch.float32, device=rgb.device)
    check(lib().pcrl_color_contrast_mean_u8(_ptr(rgb), ctypes.c_int64(rgb.stride(0)), ctypes.c_int64(rgb.stride(1)), ctypes.c_int64(rgb.stride(2)),
                                            rgb.shape[0], rgb.shape[2], ctypes.byref(desc), _ptr(out), _stream()))
    return out


def color_jitter_u8(rgb, color):
    """Materialised ColorJitterPoints (rgb [B,3,N] uint8 -> new tensor); color["mean"] as returned by color_contrast_mean."""
    assert rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.ndim == 3 and rgb.shape[1] == 3
    out = torch.empty_like(rgb)
    assert out.stride() == rgb.stride()
    desc = make_aug_desc(color=color)
    check(lib().pcrl_color_jitter_u8(_ptr(rgb), _ptr(out), ctypes.c_int64(rgb.stride(0)), ctypes.c_int64(rgb.stride(1)), ctypes.c_int64(rgb.stride(2)),
                                     rgb.shape[0], rgb.shape[2], ctypes.byref(desc), _stream()))
    return out


def encoder_fwd(desc, ew, packed, aug=None, workspace=None, bf16=False, split=False, head=None):
    """Returns pooled [B,c3] f32 and argmax [B,c3] int32 (new tensors on the current device).  bf16=True: conv1 / conv2 on
    the bf16 matrix cores with fp32 accumulation (pcrl_encoder_fwd_bf16).  head: (pcrl_feature_head, keep-alive) of
    make_feature_head -- PointNet.final_mlp applied by the same launch."""
    B, c3 = desc.B, ew.c3
    dev = packed.device
    pooled = torch.empty((B, c3), dtype=torch.float32, device=dev)
    argmax = torch.empty((B, c3), dtype=torch.int32, device=dev)
    need = ctypes.c_size_t()
    if B > 0:
        check(lib().pcrl_encoder_fwd_workspace_bytes(B, desc.N, c3, ctypes.byref(need)))
    if need.value and (workspace is None or workspace.numel() * workspace.element_size() < need.value):
        workspace = torch.empty(need.value, dtype=torch.uint8, device=dev)
    with _span("encoder_fwd"):
        if head is not None:
            fn = lib().pcrl_encoder_fwd_head_f32split if split else lib().pcrl_encoder_fwd_head_bf16 if bf16 else lib().pcrl_encoder_fwd_head_f32
            check(fn(ctypes.byref(desc), ctypes.byref(aug) if aug is not None else None, ctypes.byref(ew), _ptr(packed), _ptr(pooled),
                     _ptr(argmax), ctypes.byref(head[0]), _ptr(workspace), ctypes.c_size_t(need.value), _stream()))
        else:
            fn = lib().pcrl_encoder_fwd_f32split if split else lib().pcrl_encoder_fwd_bf16 if bf16 else lib().pcrl_encoder_fwd_f32
            check(fn(ctypes.byref(desc), ctypes.byref(aug) if aug is not None else None,
                     ctypes.byref(ew), _ptr(packed), _ptr(pooled), _ptr(argmax),
                     _ptr(workspace), ctypes.c_size_t(need.value), _stream()))
    return pooled, argmax


def encoder_num_grads(ew):
    n = ctypes.c_size_t()
    check(lib().pcrl_encoder_num_grads(ew.c_in, ew.c1, ew.c2, ew.c3, ctypes.byref(n)))
    return n.value


def encoder_grad_views(flat, ew):
    """Split the flat gradient into the reference's parameter order/shapes (conv weights as [out,in,1])."""
    C, c1, c2, c3 = ew.c_in, ew.c1, ew.c2, ew.c3
    sizes = [("conv0.weight", (c1, C, 1)), ("conv0.bias", (c1,)), ("conv1.weight", (c2, c1, 1)), ("norm1.weight", (c2,)),
             ("norm1.bias", (c2,)), ("conv2.weight", (c3, c2, 1)), ("norm2.weight", (c3,)), ("norm2.bias", (c3,))]
    out, o = {}, 0
    for name, shape in sizes:
        n = 1
        for d in shape:
            n *= d
        out[name] = flat[o:o + n].view(shape)
        o += n
    assert o == flat.numel()
    return out


def encoder_bwd_prepare(desc, ew, packed, argmax, pooled, workspace, aug=None):
    """First of the two calls of the fp32 backward (pcrl_encoder_bwd_prepare_f32): needs only the forward's outputs, so it may
    run on a side stream while the heads' backward produces grad_pooled; finish with encoder_bwd(..., prepared=True) on the SAME
    workspace, ordered behind this call."""
    assert argmax.dtype == torch.int32 and argmax.is_contiguous() and pooled is not None
    with _span("encoder_bwd_prepare"):
        check(lib().pcrl_encoder_bwd_prepare_f32(ctypes.byref(desc), ctypes.byref(aug) if aug is not None else None, ctypes.byref(ew),
                                                 _ptr(packed), _ptr(argmax), _ptr(pooled), _ptr(workspace),
                                                 ctypes.c_size_t(workspace.numel()), _stream()))


def encoder_bwd(desc, ew, packed, argmax, grad_pooled, aug=None, workspace=None, want_n_active=False, out=None, bf16=False, pooled=None,
                split=False, prepared=False):
    """Flat encoder gradient [pcrl_encoder_num_grads] for d(loss)/d(pooled) = grad_pooled [B,c3]
    (written into `out` when given: a slice of an optimizer's flat gradient buffer)."""
    dev = packed.device
    assert argmax.dtype == torch.int32 and argmax.is_contiguous() and grad_pooled.dtype == torch.float32
    grad_pooled = grad_pooled.contiguous()
    grads = out if out is not None else torch.empty(encoder_num_grads(ew), dtype=torch.float32, device=dev)
    assert grads.numel() == encoder_num_grads(ew) and grads.is_contiguous()
    need = ctypes.c_size_t()
    check(lib().pcrl_encoder_bwd_workspace_bytes(desc.B, ew.c_in, ew.c1, ew.c2, ew.c3, ctypes.byref(need)))
    if workspace is None or workspace.numel() * workspace.element_size() < need.value:
        workspace = torch.empty(max(need.value, 1), dtype=torch.uint8, device=dev)
    n_active = torch.empty(desc.B, dtype=torch.int32, device=dev) if want_n_active else None
    with _span("encoder_bwd"):
        assert not prepared or not (split or bf16)
        fn = lib().pcrl_encoder_bwd_f32split if split else lib().pcrl_encoder_bwd_bf16 if bf16 else \
            lib().pcrl_encoder_bwd_prepared_f32 if prepared else lib().pcrl_encoder_bwd_f32
        check(fn(ctypes.byref(desc), ctypes.byref(aug) if aug is not None else None, ctypes.byref(ew),
                                         _ptr(packed), _ptr(argmax), _ptr(grad_pooled), _ptr(pooled), _ptr(grads), _ptr(n_active),
                                         _ptr(workspace), ctypes.c_size_t(workspace.numel()), _stream()))
    return (grads, n_active) if want_n_active else grads


def adam_workspace_bytes(n):
    need = ctypes.c_size_t()
    check(lib().pcrl_adam_workspace_bytes(ctypes.c_size_t(n), ctypes.byref(need)))
    return need.value


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, grad_scale, step_counter, grad_norm_out, workspace,
              target=None, target_begin=0, target_end=0, tau=0.0, defer=False, rider=None):
    """defer=True: returns an AdamPending that must be passed to gather_scalars(..., pending=[...]) later in the step (it sums
    the gradient norm and advances the step count); otherwise a second launch does that right away and None is returned.
    rider: dict(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, grad_scale, step_counter, grad_norm_out, partial) -- a
    second small optimizer stepped by one extra workgroup of the same launch; returns (pending, rider's pending) then."""
    pending = _lib.AdamPending() if defer else None
    if rider is not None:
        r = _lib.AdamRider(param=rider["param"].data_ptr(), grad=rider["grad"].data_ptr(), exp_avg=rider["exp_avg"].data_ptr(),
                           exp_avg_sq=rider["exp_avg_sq"].data_ptr(), n=rider["param"].numel(), lr=rider["lr"], beta1=rider["beta1"],
                           beta2=rider["beta2"], eps=rider["eps"], grad_scale=rider["grad_scale"], step_counter=rider["step_counter"].data_ptr(),
                           grad_norm_out=rider["grad_norm_out"].data_ptr(), partial=rider["partial"].data_ptr())
        r_pending = _lib.AdamPending() if defer else None
        with _span("adam_step"):
            check(lib().pcrl_adam_step_rider_f32(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), ctypes.c_size_t(param.numel()),
                                                 ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps),
                                                 ctypes.c_float(grad_scale), _ptr(step_counter), _ptr(grad_norm_out),
                                                 _ptr(target), ctypes.c_size_t(target_begin), ctypes.c_size_t(target_end), ctypes.c_float(tau),
                                                 _ptr(workspace), ctypes.c_size_t(workspace.numel() * workspace.element_size()),
                                                 ctypes.byref(pending) if defer else None, ctypes.byref(r), ctypes.byref(r_pending) if defer else None,
                                                 _stream()))
        return pending, r_pending
    with _span("adam_step"):
        check(lib().pcrl_adam_step_f32(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), ctypes.c_size_t(param.numel()),
                                       ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps),
                                       ctypes.c_float(grad_scale), _ptr(step_counter), _ptr(grad_norm_out),
                                       _ptr(target), ctypes.c_size_t(target_begin), ctypes.c_size_t(target_end), ctypes.c_float(tau),
                                       _ptr(workspace), ctypes.c_size_t(workspace.numel() * workspace.element_size()),
                                       ctypes.byref(pending) if defer else None, _stream()))
    return pending


def _rider_struct(rider):
    return _lib.AdamRider(param=rider["param"].data_ptr(), grad=rider["grad"].data_ptr(), exp_avg=rider["exp_avg"].data_ptr(),
                          exp_avg_sq=rider["exp_avg_sq"].data_ptr(), n=rider["param"].numel(), lr=rider["lr"], beta1=rider["beta1"],
                          beta2=rider["beta2"], eps=rider["eps"], grad_scale=rider["grad_scale"], step_counter=rider["step_counter"].data_ptr(),
                          grad_norm_out=rider["grad_norm_out"].data_ptr(), partial=rider["partial"].data_ptr())


def grad_norm_partials(grad, grad_scale, step_counter, grad_norm_out, workspace, rider=None):
    """include/pcrl.h: pcrl_grad_norm_partials_f32 -- an optimizer pass's gradient-norm partial sums WITHOUT the pass (bit for bit the
    pass's own), so that the step can publish its metrics before that pass; returns the AdamPending for gather_scalars (which forms the
    norm and advances the step count), or (pending, rider's pending) when a small optimizer (the temperature) rides on the launch."""
    pending = _lib.AdamPending()
    r = _rider_struct(rider) if rider is not None else None
    r_pending = _lib.AdamPending() if rider is not None else None
    with _span("grad_norm"):
        check(lib().pcrl_grad_norm_partials_f32(_ptr(grad), ctypes.c_size_t(grad.numel()), ctypes.c_float(grad_scale), _ptr(step_counter),
                                                _ptr(grad_norm_out), _ptr(workspace), ctypes.c_size_t(workspace.numel() * workspace.element_size()),
                                                ctypes.byref(pending), ctypes.byref(r) if r is not None else None,
                                                ctypes.byref(r_pending) if r is not None else None, _stream()))
    return (pending, r_pending) if rider is not None else pending


def _scalar_list(entries, pending):
    n = len(entries)
    pend = (_lib.AdamPending * max(len(pending), 1))(*pending)
    src = (ctypes.c_void_p * n)(*[e[0].data_ptr() for e in entries])
    dst = (ctypes.c_void_p * n)(*[e[1].data_ptr() for e in entries])
    flags = (ctypes.c_int32 * n)(*[int(bool(e[2])) for e in entries])
    return src, dst, flags, n, pend


def adam_step_published(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, grad_scale, step_counter, target=None, target_begin=0,
                        target_end=0, tau=0.0, gather=None):
    """include/pcrl.h: pcrl_adam_step_published_f32 -- the optimizer pass of a step whose norm / step count went ahead of it
    (grad_norm_partials): reads the step count as already advanced.  gather = (entries, pending, host_out) as for gather_scalars: the
    pass's first workgroup gathers and publishes the step's metrics (pcrl_adam_step_published_gather_f32) instead of a launch in front."""
    if gather is not None:
        entries, pending, host_out = gather
        assert host_out.dtype == torch.float32 and host_out.is_pinned() and host_out.numel() >= len(entries) and host_out.is_contiguous()
        src, dst, flags, n, pend = _scalar_list(entries, pending)
        with _span("adam_step"):
            check(lib().pcrl_adam_step_published_gather_f32(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), ctypes.c_size_t(param.numel()),
                                                            ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps),
                                                            ctypes.c_float(grad_scale), _ptr(step_counter), _ptr(target), ctypes.c_size_t(target_begin),
                                                            ctypes.c_size_t(target_end), ctypes.c_float(tau), src, dst, flags, n, pend, len(pending),
                                                            ctypes.c_void_p(host_out.data_ptr()), _stream()))
        return
    with _span("adam_step"):
        check(lib().pcrl_adam_step_published_f32(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), ctypes.c_size_t(param.numel()),
                                                 ctypes.c_float(lr), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps),
                                                 ctypes.c_float(grad_scale), _ptr(step_counter), _ptr(target), ctypes.c_size_t(target_begin),
                                                 ctypes.c_size_t(target_end), ctypes.c_float(tau), _stream()))


def polyak(target, src, tau):
    check(lib().pcrl_polyak_f32(_ptr(target), _ptr(src), ctypes.c_size_t(target.numel()), ctypes.c_float(tau), _stream()))


# ---- dense heads ---------------------------------------------------------------------------------
def _f(x):
    return ctypes.c_float(float(x))


_SPAN_SHAPES = False          # debugging: name GEMM spans by their shapes (set from a profiling script)


def gemm_desc(A, B, C, M, N, K, a_strides, b_strides, ldc, bias=None, mask=None, ld_mask=0, relu=False, ones_col=-1, accumulate=False,
              batch=1, batch_strides=(0, 0, 0, 0, 0), c_ones=None, c_ones_batch_stride=0):
    """Descriptor of C[z] = epilogue(A[z] . B[z]); strides in elements; batch_strides = (A, B, C, bias, mask).  See include/pcrl.h."""
    return GemmDesc(A=A.data_ptr(), B=B.data_ptr(), C=C.data_ptr(), bias=bias.data_ptr() if bias is not None else None,
                    mask=mask.data_ptr() if mask is not None else None, M=M, N=N, K=K, batch=batch,
                    a_stride_m=a_strides[0], a_stride_k=a_strides[1], b_stride_k=b_strides[0], b_stride_n=b_strides[1], ldc=ldc, ld_mask=ld_mask,
                    a_batch_stride=batch_strides[0], b_batch_stride=batch_strides[1], c_batch_stride=batch_strides[2],
                    bias_batch_stride=batch_strides[3], mask_batch_stride=batch_strides[4],
                    relu=int(relu), ones_col=ones_col, accumulate=int(accumulate),
                    C_ones=c_ones.data_ptr() if c_ones is not None else None, c_ones_batch_stride=c_ones_batch_stride)


def gemm_group(descs):
    """Launch up to 4 independent GEMMs (gemm_desc results; None entries are skipped) as one kernel."""
    descs = [d for d in descs if d is not None]
    if not descs:
        return
    arr = (GemmDesc * len(descs))(*descs)
    name = "gemm " + " | ".join(f"M{d.M} N{d.N} K{d.K} b{d.batch}" for d in descs) if _SPAN_SHAPES else "gemm"
    if TIMER is not None:
        TIMER.flops["gemm"] = TIMER.flops.get("gemm", 0.0) + sum(2.0 * d.M * (d.ones_col if d.ones_col >= 0 else d.N) * d.K * max(d.batch, 1) for d in descs)
    with _span(name):
        check(lib().pcrl_gemm_group_f32(arr, len(descs), _stream()))


if os.environ.get("PCRL_GEMM_PATHS") == "legacy":      # measurement only (tools/r5_ab*.sh): the rounds-1-4 tile paths in this build
    lib().pcrl_gemm_set_tile64_min(1 << 30)


def encoder_bwd_set_fused(mode):
    """include/pcrl.h: pcrl_encoder_bwd_set_fused -- 0 the points / wgrad / reduce launches, 1 (default) the team kernel for launches of at
    most two tiles per CU, 2 the team kernel wherever it is built."""
    check(lib().pcrl_encoder_bwd_set_fused(int(mode)))


def encoder_bwd_last_schedule():
    """include/pcrl.h: pcrl_encoder_bwd_last_schedule -- 1 round-2 kernels, 2 Gram-form launches, 3 Gram-form team kernel (this thread's last call)."""
    return int(lib().pcrl_encoder_bwd_last_schedule())


if os.environ.get("PCRL_BWD_PATHS") in ("legacy", "team"):       # measurement only (same-box A/B): one backward path for every launch
    lib().pcrl_encoder_bwd_set_fused(0 if os.environ["PCRL_BWD_PATHS"] == "legacy" else 2)


def gemm_plan(descs):
    """[(path, tile shape, workgroups)] pcrl_gemm_group_f32 would use for these problems (include/pcrl.h: pcrl_gemm_group_plan_f32)."""
    descs = [d for d in descs if d is not None]
    arr = (GemmDesc * len(descs))(*descs)
    out = (ctypes.c_int32 * (3 * len(descs)))()
    check(lib().pcrl_gemm_group_plan_f32(arr, len(descs), out))
    return [tuple(out[3 * i:3 * i + 3]) for i in range(len(descs))]


def gemm_set_tile64_min(min_tiles):
    """Tuning knob of pcrl_gemm_group_f32 (include/pcrl.h): minimum number of 64x64 tiles in a launch for the LDS-staged
    path; returns the previous value (negative argument: query only)."""
    return lib().pcrl_gemm_set_tile64_min(int(min_tiles))


def gemm(*args, **kwargs):
    """One GEMM launch; arguments as gemm_desc."""
    gemm_group([gemm_desc(*args, **kwargs)])


def layernorm_rows_fwd(x, ldx, gamma, beta, M, F, eps, dsts, xhat=None, rstd=None):
    """dsts: list of (tensor, column offset, leading dimension)."""
    n = len(dsts)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() + 4 * off for t, off, _ in dsts])
    lds = (ctypes.c_int64 * n)(*[ld for _, _, ld in dsts])
    check(lib().pcrl_layernorm_rows_fwd_f32(_ptr(x), ctypes.c_int64(ldx), _ptr(gamma), _ptr(beta), M, F, _f(eps), ptrs, lds, n,
                                            _ptr(xhat), _ptr(rstd), _stream()))


def make_feature_head(weight, bias, gamma, beta, F, eps, ranges):
    """pcrl_feature_head for encoder_fwd(head=...): ranges = [(first cloud, job dict as in layernorm_rows_fwd_multi without x / ldx)]."""
    h = _lib.FeatureHead()
    h.weight, h.bias, h.gamma, h.beta = weight.data_ptr(), bias.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    h.F, h.eps, h.n_ranges = F, eps, len(ranges)
    keep = [weight, bias, gamma, beta]
    for i, (begin, job) in enumerate(ranges):
        h.begin[i] = begin
        _fill_ln_job(h.job[i], job)
        keep.append(job)
    return h, keep


def _fill_ln_job(j, job):
    if job.get("x") is not None:
        j.x, j.ldx = job["x"].data_ptr(), job["ldx"]
    j.M, j.n_dst = job["M"], len(job["dsts"])
    for i, (t, off, ld) in enumerate(job["dsts"]):
        j.dst[i], j.ld_dst[i] = t.data_ptr() + 4 * off, ld
    j.xhat = job["xhat"].data_ptr() if job.get("xhat") is not None else None
    j.rstd = job["rstd"].data_ptr() if job.get("rstd") is not None else None
    for c, cat in enumerate(job.get("cats", [])):
        src, dst, off, ld = cat[:4]
        assert src.dtype == torch.float32 and src.stride(-1) == 1
        j.cat_src[c], j.cat_dst[c], j.cat_ld_src[c], j.cat_ld_dst[c], j.cat_n[c] = src.data_ptr(), dst.data_ptr() + 4 * off, src.stride(0), ld, src.shape[1]
        j.cat_row_div[c] = int(cat[4]) if len(cat) > 4 else 1        # source stored once per sample, read by each augmentation row


def layernorm_rows_fwd_multi(jobs, gamma, beta, F, eps):
    """jobs: [dict(x=, ldx=, M=, dsts=[(tensor, column offset, leading dim)], xhat=None, rstd=None,
    cats=[(src tensor [M, n] (row stride src.stride(0)), dst tensor, dst column offset, dst leading dim)])], one launch."""
    arr = (_lib.LnJob * len(jobs))()
    for j, job in zip(arr, jobs):
        _fill_ln_job(j, job)
    check(lib().pcrl_layernorm_rows_fwd_multi_f32(arr, len(jobs), _ptr(gamma), _ptr(beta), F, _f(eps), _stream()))


def layernorm_rows_bwd(dy0, dy1, lddy, xhat, rstd, gamma, M, F, dx, lddx, dgamma, dbeta, workspace, accumulate=False):
    check(lib().pcrl_layernorm_rows_bwd_f32(ctypes.c_void_p(dy0), ctypes.c_void_p(dy1) if dy1 else None, ctypes.c_int64(lddy), _ptr(xhat), _ptr(rstd),
                                            _ptr(gamma), M, F, _ptr(dx), ctypes.c_int64(lddx), _ptr(dgamma), _ptr(dbeta), int(accumulate),
                                            _ptr(workspace), ctypes.c_size_t(workspace.numel() * workspace.element_size()), _stream()))


def _check_f32_vec(t, n, what):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= n):
        raise TypeError(f"{what} must be a contiguous float32 device tensor of {n} elements, got "
                        f"{(t.dtype, tuple(t.shape), t.device) if torch.is_tensor(t) else type(t)}")


def tanh_gaussian_fwd(feat, ld_feat, eps, scale, bias, B, A, ls_min, ls_max, epsilon, action, ld_action, neg_logp, saved=None,
                      action2_ptr=None, ld_action2=0):
    _check_f32_vec(scale, A, "scale"), _check_f32_vec(bias, A, "bias")
    check(lib().pcrl_tanh_gaussian_fwd_f32(_ptr(feat), ctypes.c_int64(ld_feat), _ptr(eps), _ptr(scale), _ptr(bias), B, A, _f(ls_min), _f(ls_max),
                                           _f(epsilon), _ptr(action), ctypes.c_int64(ld_action),
                                           ctypes.c_void_p(action2_ptr) if action2_ptr else None, ctypes.c_int64(ld_action2),
                                           _ptr(neg_logp), _ptr(saved), _stream()))


def tanh_gaussian_sample_fwd(feat, ld_feat, seed, step_counter, draw_id, eps_out, scale, bias, B, A, ls_min, ls_max, epsilon, action, ld_action,
                             neg_logp, saved=None, action2_ptr=None, ld_action2=0):
    """tanh_gaussian_fwd with in-kernel Philox draws (written to eps_out); step_counter: device int32 tensor."""
    _check_f32_vec(scale, A, "scale"), _check_f32_vec(bias, A, "bias")
    check(lib().pcrl_tanh_gaussian_sample_fwd_f32(_ptr(feat), ctypes.c_int64(ld_feat), ctypes.c_uint64(seed & (2 ** 64 - 1)), _ptr(step_counter),
                                                  int(draw_id), _ptr(eps_out), _ptr(scale), _ptr(bias), B, A, _f(ls_min), _f(ls_max),
                                                  _f(epsilon), _ptr(action), ctypes.c_int64(ld_action),
                                                  ctypes.c_void_p(action2_ptr) if action2_ptr else None, ctypes.c_int64(ld_action2),
                                                  _ptr(neg_logp), _ptr(saved), _stream()))


def tanh_gaussian_bwd(feat, ld_feat, eps, saved, scale, B, A, ls_min, ls_max, epsilon, da0_ptr, da1_ptr, ld_da, d_neglogp, d_feat, ld_d_feat):
    _check_f32_vec(scale, A, "scale")
    check(lib().pcrl_tanh_gaussian_bwd_f32(_ptr(feat), ctypes.c_int64(ld_feat), _ptr(eps), _ptr(saved), _ptr(scale), B, A, _f(ls_min), _f(ls_max),
                                           _f(epsilon), ctypes.c_void_p(da0_ptr), ctypes.c_void_p(da1_ptr) if da1_ptr else None,
                                           ctypes.c_int64(ld_da), _ptr(d_neglogp), _ptr(d_feat), ctypes.c_int64(ld_d_feat), _stream()))


def sac_critic_loss(q_next, ld_qn, neg_logp_next, rewards, dones_u8, log_alpha, gamma, reward_scale, ignore_dones, group, q, ld_q, B, H,
                    q_target, dq, ld_dq, stats, rd_row_div=1):
    check(lib().pcrl_sac_critic_loss_f32(_ptr(q_next), ctypes.c_int64(ld_qn), _ptr(neg_logp_next), _ptr(rewards), _ptr(dones_u8), int(rd_row_div), _ptr(log_alpha),
                                         _f(gamma), _f(reward_scale), int(ignore_dones), int(group), _ptr(q), ctypes.c_int64(ld_q), B, H,
                                         _ptr(q_target), _ptr(dq), ctypes.c_int64(ld_dq), _ptr(stats), _stream()))


def sac_actor_loss(q_pi, ld_q, neg_logp, log_alpha, target_entropy, B, H, dq, ld_dq, d_neglogp, alpha_grad, stats):
    check(lib().pcrl_sac_actor_loss_f32(_ptr(q_pi), ctypes.c_int64(ld_q), _ptr(neg_logp), _ptr(log_alpha), _f(target_entropy), B, H,
                                        _ptr(dq), ctypes.c_int64(ld_dq), _ptr(d_neglogp), _ptr(alpha_grad), _ptr(stats), _stream()))


def layernorm_rows_bwd_partials(dy0, dy1, lddy, xhat, rstd, gamma, M, F, dx, lddx, workspace):
    """layernorm_rows_bwd without the final column sums: the [ceil(M/4)][2][F] partials stay in `workspace` (float32 tensor)."""
    check(lib().pcrl_layernorm_rows_bwd_partials_f32(ctypes.c_void_p(dy0), ctypes.c_void_p(dy1) if dy1 else None, ctypes.c_int64(lddy), _ptr(xhat),
                                                     _ptr(rstd), _ptr(gamma), M, F, _ptr(dx), ctypes.c_int64(lddx), _ptr(workspace),
                                                     ctypes.c_size_t(workspace.numel() * workspace.element_size()), _stream()))


def encoder_bwd_attach_ln_bwd(dy0, dy1, lddy, xhat, rstd, gamma, M, F, dx, lddx, workspace):
    """include/pcrl.h: pcrl_encoder_bwd_attach_ln_bwd -- the feature head's LayerNorm backward (arguments as layernorm_rows_bwd_partials) rides
    on the NEXT encoder-backward prep launch of this thread (hip.encoder_bwd_prepare / encoder_bwd) instead of being a launch of its own."""
    check(lib().pcrl_encoder_bwd_attach_ln_bwd(ctypes.c_void_p(dy0), ctypes.c_void_p(dy1) if dy1 else None, ctypes.c_int64(lddy), _ptr(xhat), _ptr(rstd),
                                               _ptr(gamma), M, F, _ptr(dx), ctypes.c_int64(lddx), _ptr(workspace),
                                               ctypes.c_size_t(workspace.numel() * workspace.element_size())))


def colsum_jobs(jobs, attach_to_encoder_bwd=False):
    """jobs: [(part ptr (int), blk_stride, nblk, ncols, out ptr (int), scale, op)] -> out[c] = scale * (sum | max)_b part[b * blk_stride + c].
    attach_to_encoder_bwd: no launch -- the jobs ride on the reduce launch of the next encoder_bwd call (pcrl_encoder_bwd_attach_colsum)."""
    arr = (_lib.ColsumJob * len(jobs))()
    for j, (part, stride, nblk, ncols, out, scale, op) in zip(arr, jobs):
        j.part, j.blk_stride, j.nblk, j.ncols, j.out, j.scale, j.op = part, stride, nblk, ncols, out, scale, op
    if attach_to_encoder_bwd:
        check(lib().pcrl_encoder_bwd_attach_colsum(arr, len(jobs)))
    else:
        check(lib().pcrl_colsum_jobs_f32(arr, len(jobs), _stream()))


def q_tail_workspace_floats(M, H):
    a, b = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib().pcrl_q_tail_workspace_floats(M, H, ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def q_tail_critic(h2, h2_hs, w2, b2, w_hs, h2_t, h2_t_hs, w2_t, b2_t, w_t_hs, nlp_next, rewards, dones_u8, rd_row_div, log_alpha, gamma,
                  reward_scale, ignore_dones, group, M, H, q, q_target, dq, dh2, part, stat_part):
    """See include/pcrl.h; q / dq are [M, 2], dh2 [2, M, H]."""
    with _span("q_tail"):
        check(lib().pcrl_q_tail_critic_f32(_ptr(h2), ctypes.c_int64(h2_hs), _ptr(w2), _ptr(b2), ctypes.c_int64(w_hs), _ptr(h2_t),
                                           ctypes.c_int64(h2_t_hs), _ptr(w2_t), _ptr(b2_t), ctypes.c_int64(w_t_hs), _ptr(nlp_next), _ptr(rewards),
                                           _ptr(dones_u8), int(rd_row_div), _ptr(log_alpha), _f(gamma), _f(reward_scale), int(ignore_dones),
                                           int(group), M, H, _ptr(q), ctypes.c_int64(2), _ptr(q_target), _ptr(dq), ctypes.c_int64(2), _ptr(dh2),
                                           ctypes.c_int64(M * H), _ptr(part), _ptr(stat_part), _stream()))


def q_tail_actor(h2, h2_hs, w2, b2, w_hs, neg_logp, log_alpha, M, H, q, dq, dh2, d_neglogp, stat_part):
    with _span("q_tail"):
        check(lib().pcrl_q_tail_actor_f32(_ptr(h2), ctypes.c_int64(h2_hs), _ptr(w2), _ptr(b2), ctypes.c_int64(w_hs), _ptr(neg_logp), _ptr(log_alpha),
                                          M, H, _ptr(q), ctypes.c_int64(2), _ptr(dq), ctypes.c_int64(2), _ptr(dh2), ctypes.c_int64(M * H),
                                          _ptr(d_neglogp), _ptr(stat_part), _stream()))


def q_tail_actor_cols(h2, h2_hs, w2, b2, w_hs, neg_logp, log_alpha, M, H, q, dq, dh2, d_neglogp, stat_part, w0, w0_hs, ld_w0, col0, ncols, cols_out):
    """q_tail_actor + (extra workgroups) columns [col0, col0 + ncols) of the heads' first-layer weight as cols_out [2, ncols, H]."""
    assert cols_out.is_contiguous() and cols_out.numel() == 2 * ncols * H
    with _span("q_tail"):
        check(lib().pcrl_q_tail_actor_cols_f32(_ptr(h2), ctypes.c_int64(h2_hs), _ptr(w2), _ptr(b2), ctypes.c_int64(w_hs), _ptr(neg_logp), _ptr(log_alpha),
                                               M, H, _ptr(q), ctypes.c_int64(2), _ptr(dq), ctypes.c_int64(2), _ptr(dh2), ctypes.c_int64(M * H),
                                               _ptr(d_neglogp), _ptr(stat_part), _ptr(w0), ctypes.c_int64(w0_hs), int(ld_w0), int(col0), int(ncols),
                                               _ptr(cols_out), _stream()))


def policy_tail_bwd(dh1, dh1_hs, w0a, w0a_hs, M, H, A, feat, ld_feat, eps, saved, scale, ls_min, ls_max, epsilon, d_neglogp, d_feat, ld_d_feat,
                    h2, w2, dh2, finalize=None):
    """d_action (Q heads' dh1 x action columns) -> TanhGaussianHead backward -> the policy's dh2, one launch; finalize =
    (stat_part, log_alpha, target_entropy, alpha_grad, stats): one more workgroup does actor_finalize."""
    _check_f32_vec(scale, A, "scale")
    fin = finalize or (None, None, 0.0, None, None)
    with _span("policy_tail_bwd"):
        check(lib().pcrl_policy_tail_bwd_f32(_ptr(dh1), ctypes.c_int64(dh1_hs), _ptr(w0a), ctypes.c_int64(w0a_hs), M, H, A, _ptr(feat),
                                             ctypes.c_int64(ld_feat), _ptr(eps), _ptr(saved), _ptr(scale), _f(ls_min), _f(ls_max), _f(epsilon),
                                             _ptr(d_neglogp), _ptr(d_feat), ctypes.c_int64(ld_d_feat), _ptr(h2), _ptr(w2), _ptr(dh2),
                                             _ptr(fin[0]), _ptr(fin[1]), _f(fin[2]), _ptr(fin[3]), _ptr(fin[4]), _stream()))


def actor_finalize(stat_part, M, log_alpha, target_entropy, alpha_grad, stats):
    check(lib().pcrl_actor_finalize_f32(_ptr(stat_part), M, _ptr(log_alpha), _f(target_entropy), _ptr(alpha_grad), _ptr(stats), _stream()))


def pack_attach_cols(jobs):
    """jobs: [(src tensor view at head 0's matrix, head_stride, heads, rows, ld, col0, ncols, dst [heads, ncols, rows])] -- column-gather
    jobs that ride on the NEXT encoder_pack_weights launch (pcrl_encoder_pack_attach_cols); [] withdraws."""
    arr = (_lib.ColGather * max(len(jobs), 1))()
    for a, (src, hs, heads, rows, ld, col0, ncols, dst) in zip(arr, jobs):
        assert dst.is_contiguous() and dst.numel() == heads * ncols * rows
        a.src, a.head_stride, a.heads, a.rows, a.ld, a.col0, a.ncols, a.dst = src.data_ptr(), hs, heads, rows, ld, col0, ncols, dst.data_ptr()
    check(lib().pcrl_encoder_pack_attach_cols(arr, len(jobs)))


def pack_flush_cols():
    """Runs column-gather jobs that are still attached (no pack launch consumed them) as a launch of their own."""
    check(lib().pcrl_encoder_pack_flush_cols(_stream()))


def policy_tail_fwd(h2, M, H, w2, b2, A, eps, seed, step_counter, draw_id, eps_out, scale, bias, ls_min, ls_max, epsilon, feat, action, ld_action,
                    neg_logp, saved=None, action2_ptr=None, ld_action2=0, fold=None):
    """The policy's last Linear + TanhGaussianHead "max-entropy"; eps None: Philox draws in the kernel (written to eps_out).
    fold = (pre, pre_head_stride, w0_action_cols, w0a_head_stride, n_heads, h1, h1_head_stride): also the Q heads' first layer on the
    action just formed (pcrl_policy_tail_fwd_fold_f32; H = 1024 and M <= 512 only)."""
    _check_f32_vec(scale, A, "scale"), _check_f32_vec(bias, A, "bias")
    if fold is not None:
        pre, pre_hs, w0a, w0a_hs, n_heads, h1, h1_hs = fold
        with _span("policy_tail"):
            check(lib().pcrl_policy_tail_fwd_fold_f32(_ptr(h2), M, H, _ptr(w2), _ptr(b2), A, _ptr(eps), ctypes.c_uint64(seed & (2 ** 64 - 1)),
                                                      _ptr(step_counter), int(draw_id), _ptr(eps_out), _ptr(scale), _ptr(bias), _f(ls_min), _f(ls_max),
                                                      _f(epsilon), _ptr(feat), ctypes.c_int64(2 * A), _ptr(action), ctypes.c_int64(ld_action),
                                                      ctypes.c_void_p(action2_ptr) if action2_ptr else None, ctypes.c_int64(ld_action2),
                                                      _ptr(neg_logp), _ptr(saved), _ptr(pre), ctypes.c_int64(pre_hs), _ptr(w0a),
                                                      ctypes.c_int64(w0a_hs), int(n_heads), _ptr(h1), ctypes.c_int64(h1_hs), _stream()))
        return
    with _span("policy_tail"):
        check(lib().pcrl_policy_tail_fwd_f32(_ptr(h2), M, H, _ptr(w2), _ptr(b2), A, _ptr(eps), ctypes.c_uint64(seed & (2 ** 64 - 1)), _ptr(step_counter),
                                             int(draw_id), _ptr(eps_out), _ptr(scale), _ptr(bias), _f(ls_min), _f(ls_max), _f(epsilon), _ptr(feat),
                                             ctypes.c_int64(2 * A), _ptr(action), ctypes.c_int64(ld_action),
                                             ctypes.c_void_p(action2_ptr) if action2_ptr else None, ctypes.c_int64(ld_action2), _ptr(neg_logp),
                                             _ptr(saved), _stream()))


def gather_segments(pairs):
    """ctypes segment table for replay_gather / replay_sample_gather (build once, reuse every step):
    pairs = [(storage [capacity, ...], staging [B, ...])] contiguous tensors of equal row size."""
    n = len(pairs)
    segs = (_lib.GatherSeg * n)()
    for i, (src, dst) in enumerate(pairs):
        row = src[0].numel() * src.element_size()
        assert src.is_contiguous() and dst.is_contiguous() and dst[0].numel() * dst.element_size() == row
        segs[i].src, segs[i].dst, segs[i].row_bytes = src.data_ptr(), dst.data_ptr(), row
    return segs


def replay_gather(segs, idx, capacity):
    """staging[b] = storage[idx[b]] for every key; idx int32 [B] on the device."""
    with _span("replay_gather"):
        check(lib().pcrl_replay_gather(segs, len(segs), _ptr(idx), idx.numel(), ctypes.c_int64(capacity), _stream()))


def replay_sample_gather(segs, B, size, capacity, seed, draw, idx_out=None):
    """Rows drawn in the kernel (uniform on [0, size), Philox keyed by seed / draw); idx_out int32 [B] receives them."""
    with _span("replay_gather"):
        check(lib().pcrl_replay_sample_gather(segs, len(segs), B, ctypes.c_int64(size), ctypes.c_int64(capacity),
                                              ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.c_uint64(draw), _ptr(idx_out), _stream()))


def replay_sample_gather_state(segs, B, capacity, seed, state, idx_out=None):
    """The same launch with (draw, size) read from the device int64 tensor `state` [3 + B] = {draw, size, ticket, B row tickets};
    the launch advances draw itself, so it can be replayed from a hipGraph."""
    assert state.is_cuda and state.dtype == torch.int64 and state.numel() >= 3 + B and state.is_contiguous()
    with _span("replay_gather"):
        check(lib().pcrl_replay_sample_gather_state(segs, len(segs), B, ctypes.c_int64(capacity), ctypes.c_uint64(seed & (2 ** 64 - 1)),
                                                    _ptr(state), ctypes.c_int64(state.numel()), _ptr(idx_out), _stream()))


def gather_scalars(entries, pending=(), host_out=None):
    """entries: [(src scalar tensor, dst scalar tensor, take_exp)] -> dst = exp?(src), one launch for up to 16 scalars;
    pending: AdamPending objects of deferred adam_step calls, finished by the same launch before the copies;
    host_out: pinned float32 host tensor [>= n] that also receives the n values, one 4-byte store each (a host that pre-filled
    the slots with the bit pattern 0xFFFFFFFF sees a value as soon as its slot changes)."""
    n = len(entries)
    pend = (_lib.AdamPending * max(len(pending), 1))(*pending)
    src = (ctypes.c_void_p * n)(*[e[0].data_ptr() for e in entries])
    dst = (ctypes.c_void_p * n)(*[e[1].data_ptr() for e in entries])
    flags = (ctypes.c_int32 * n)(*[int(bool(e[2])) for e in entries])
    if host_out is None:
        check(lib().pcrl_gather_scalars_f32(src, dst, flags, n, pend, len(pending), _stream()))
        return
    assert host_out.dtype == torch.float32 and host_out.is_pinned() and host_out.numel() >= n and host_out.is_contiguous()
    check(lib().pcrl_gather_scalars_host_f32(src, dst, flags, n, pend, len(pending), ctypes.c_void_p(host_out.data_ptr()), _stream()))


# ---- stand-alone memory-shaped kernels --------------------------------------------------------------
def segmax_fwd(x):
    """x [..., N] f32 contiguous -> (max [...], argmax [...] int32), torch.max(dim=-1) semantics."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    N, rows = x.shape[-1], x.numel() // x.shape[-1]
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    idx = torch.empty(x.shape[:-1], dtype=torch.int32, device=x.device)
    with _span("segmax_fwd"):
        check(lib().pcrl_segmax_fwd_f32(_ptr(x), ctypes.c_int64(rows), N, _ptr(out), _ptr(idx), _stream()))
    return out, idx


def segmax_bwd(grad_out, idx, N):
    grad_out = grad_out.contiguous()
    dx = torch.empty(tuple(idx.shape) + (N,), dtype=torch.float32, device=idx.device)
    with _span("segmax_bwd"):
        check(lib().pcrl_segmax_bwd_f32(_ptr(grad_out), _ptr(idx), ctypes.c_int64(idx.numel()), N, _ptr(dx), _stream()))
    return dx


def affine_sample(out, rot_axis, rot_range, scale_range, translation_range, shift_height, seed, offset=0, offset_tensor=None, second=None):
    """GlobalRotScaleTrans's [B,3,4] matrices drawn by one launch (pcrl_affine_sample_f32) into `out` (float32, contiguous, cuda).
    Ranges are None or sequences of 2 / 2 / 3 floats; offset_tensor: device int64 [1] read at run time (hipGraph replays).
    second = (out2, seed2, offset2): a second, independent draw of the same transform in the same launch (pcrl_affine_sample_pair_f32)."""
    assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape[1:]) == (3, 4)
    arr = lambda v, n: None if v is None else (ctypes.c_float * n)(*[float(x) for x in v])
    u64 = lambda v: ctypes.c_uint64(int(v) & (2 ** 64 - 1))
    if offset_tensor is not None:
        assert offset_tensor.dtype == torch.int64 and offset_tensor.is_cuda
    optr = _ptr(offset_tensor) if offset_tensor is not None else None
    with _span("affine_sample"):
        if second is None:
            check(lib().pcrl_affine_sample_f32(_ptr(out), out.shape[0], int(rot_axis), arr(rot_range, 2), arr(scale_range, 2), arr(translation_range, 3),
                                               int(bool(shift_height)), u64(seed), u64(offset), optr, _stream()))
        else:
            out2, seed2, offset2 = second
            assert out2.is_cuda and out2.dtype == torch.float32 and out2.is_contiguous() and out2.shape == out.shape
            check(lib().pcrl_affine_sample_pair_f32(_ptr(out), _ptr(out2), out.shape[0], int(rot_axis), arr(rot_range, 2), arr(scale_range, 2),
                                                    arr(translation_range, 3), int(bool(shift_height)), u64(seed), u64(offset), u64(seed2), u64(offset2),
                                                    optr, _stream()))
    return out


def augment_xyz(xyz, out=None, **aug):
    """Materialised RandomJitterPoints / GlobalRotScaleTrans on xyz [B,3,N] (same keywords as make_aug_desc)."""
    assert xyz.is_cuda and xyz.dtype == torch.float32 and xyz.is_contiguous() and xyz.shape[1] == 3
    out = torch.empty_like(xyz) if out is None else out
    desc = make_aug_desc(**aug)
    with _span("augment_xyz"):
        check(lib().pcrl_augment_xyz_f32(_ptr(xyz), _ptr(out), xyz.shape[0], xyz.shape[2], ctypes.byref(desc), _stream()))
    return out
