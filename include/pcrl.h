/*
 * pcrl.h -- C ABI of libpcrl_hip.so, the MI355X (gfx950) implementation of the
 * point-cloud actor-critic hot path of lz1oceani/pointcloud_rl.
 *
 * The reference has no FFI on this path: it is pure Python on stock ATen ops
 * (SURVEY.md section 2.1).  This header is therefore the boundary a maintainer
 * would bind with ctypes from the reference's own modules;
 * each entry point names the reference code it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, a negative PCRL_E_* code otherwise;
 *    pcrl_last_error() returns a thread-local message for the last failure;
 *  - all pointers except descriptor structs are DEVICE pointers; nothing is
 *    allocated, freed or retained by the library; scratch is caller-provided;
 *  - `stream` is a hipStream_t passed as void*; all work is asynchronous on it,
 *    no call synchronises;
 *  - tensors are dense row-major unless strides are given (strides in elements).
 */
#ifndef PCRL_H_
#define PCRL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCRL_OK 0
#define PCRL_E_ARG (-1)         /* invalid argument / unsupported shape */
#define PCRL_E_WORKSPACE (-2)   /* workspace too small */
#define PCRL_E_LAUNCH (-3)      /* HIP launch / runtime error */

#define PCRL_MAX_SEG 4
#define PCRL_MAX_CHANNELS 16

enum { PCRL_DT_F32 = 0, PCRL_DT_U8 = 1, PCRL_DT_BOOL = 2 };

/* One observation key ("xyz", "rgb", "pos_encoding", "seg") of the batched
 * observation dict the reference feeds PointCloudBase.preprocess
 * (pyrl/networks/backbones/pointnet.py:49-73).  Planar [B, channels, N] uses
 * stride_b = channels*N, stride_c = N, stride_n = 1; an interleaved [B, N, C]
 * tensor uses stride_b = N*C, stride_c = 1, stride_n = C. */
typedef struct pcrl_feat_seg {
    const void* ptr;
    int32_t dtype;     /* PCRL_DT_* */
    int32_t channels;
    int32_t div255;    /* 1: value / 255.0f (uint8 rgb, pointnet.py:57-58) */
    int32_t _pad;
    int64_t stride_b, stride_c, stride_n;
} pcrl_feat_seg;

/* A batch of B clouds of N points; channels are the concatenation of the
 * segments in order (torch.cat(feature, dim=-2), pointnet.py:63).  Segment 0
 * must be xyz (3 channels, f32) when an augmentation is requested. */
/* row_div > 1: cloud b reads the stored cloud b / row_div of every segment, i.e. the batch is the stored batch with each cloud
 * repeated row_div times in place (DrQ: GDict(obs).repeat(num_aug, 0), drq.py:52-60 -- the copies differ only by their
 * augmentation, which is indexed by b itself).  B counts the clouds the encoder sees (stored clouds * row_div). */
typedef struct pcrl_cloud_desc {
    int32_t B, N, nseg, row_div;
    pcrl_feat_seg seg[PCRL_MAX_SEG];
} pcrl_cloud_desc;

/* DrQ point-cloud augmentations fused into the encoder's load
 * (pyrl/utils/augmentations/pcd_aug.py).  flags is an OR of PCRL_AUG_*.
 *  JITTER  : xyz += noise, noise either explicit (jitter_noise [B,3,N] f32, the
 *            tensor RandomJitterPoints.process_single draws, pcd_aug.py:318) or,
 *            when jitter_noise == NULL, Philox4x32-10 U(lo,hi) keyed by
 *            (seed, offset) -- one draw per (b, axis, n).
 *  AFFINE  : xyz = M[b] (3x4, row-major) applied as R x + t
 *            (GlobalRotScaleTrans.process_single -> apply_rot_trans,
 *            pcd_aug.py:178-215, 84-123); M is built by the host class.
 * Order when both are set: AFFINE first, then JITTER.
 * Cloud b uses row (b * row_mul + row_add) of jitter_noise / affine / the Philox counter
 * (row_mul == 0 means 1), so a strided sub-batch (DrQ's actor step uses augmentation #0 of every
 * sample, drq.py:115) sees exactly the noise the full batch saw. */
/*  SUBSAMPLE : the cloud the encoder sees is points point_index[0..n_index) of the stored cloud, the same index for
 *            every cloud and every key (RandomDownSample.process_single: one shared random permutation prefix,
 *            pcd_aug.py:231-257 + array_ops.py:659-680); N becomes n_index and the returned argmax counts
 *            positions of the subsampled cloud, as it does in the reference where the tensors are sliced.
 *            Jitter noise / Philox counters are indexed by the subsampled position.
 *            n_index_ptr (optional, device): the number of positions that exist is read from there at run time, 1 <= *n_index_ptr <=
 *            n_index -- RandomDownSample(fixed_ratio=False) keeps a different number of points every call (pcd_aug.py:244-246);
 *            a launch replayed from a hipGraph follows the value.  Shapes, strides and counters stay those of n_index. */
/*  COLOR     : ColorJitterPoints (pcd_aug.py:269-303): torchvision's ColorJitter on the uint8 rgb key viewed as a
 *            [B,3,1,N] image batch -- brightness / contrast / saturation / hue applied in the drawn order with ONE set of
 *            factors for the whole batch, uint8 truncation after every step, the contrast step blending with the cloud's
 *            own grayscale mean (color_mean, from pcrl_color_contrast_mean_u8).  Segment 1 must be rgb (3 x uint8). */
enum { PCRL_AUG_JITTER = 1, PCRL_AUG_AFFINE = 2, PCRL_AUG_SUBSAMPLE = 4, PCRL_AUG_COLOR = 8 };
enum { PCRL_COLOR_BRIGHTNESS = 0, PCRL_COLOR_CONTRAST = 1, PCRL_COLOR_SATURATION = 2, PCRL_COLOR_HUE = 3, PCRL_COLOR_SKIP = 15 };
typedef struct pcrl_aug_desc {
    int32_t flags, row_mul, row_add, _pad;
    const float* jitter_noise;
    float jitter_lo, jitter_hi;
    uint64_t seed, offset;
    const float* affine;
    const uint64_t* offset_ptr;   /* device; when non-NULL the Philox offset is read from here at run
                                     time (a launch replayed from a hipGraph then draws fresh noise) */
    const int32_t* point_index;   /* device [n_index], values in [0, N): PCRL_AUG_SUBSAMPLE */
    int32_t n_index;
    int32_t color_order;          /* PCRL_AUG_COLOR: four PCRL_COLOR_* step ids, 4 bits each, lowest nibble first */
    float color_factor[4];        /* brightness, contrast, saturation factors (blend ratios) and the hue shift */
    float color_one_minus[4];     /* (float)(1.0 - (double)factor) for the three blends, as torch forms it; [3] unused */
    const float* color_mean;      /* device [stored clouds]: grayscale mean entering the contrast step (NULL when contrast is skipped) */
    const int32_t* n_index_ptr;   /* device, optional: PCRL_AUG_SUBSAMPLE with a point count decided on the device (see above) */
} pcrl_aug_desc;

/* Weights of the shared per-point MLP in the reference's own state_dict layout
 * (ConvMLP built by PointNet.__init__, pointnet.py:106-109; mlp.py:43-56):
 *   conv0.weight [c1,C,1] conv0.bias [c1] ; conv1.weight [c2,c1,1] norm1.{weight,bias} [c2] ;
 *   conv2.weight [c3,c2,1] norm2.{weight,bias} [c3] ; LN eps (1e-6 in every shipped config).
 * Built (c1,c2,c3): (64,128,256), (128,128,256), (32,64,128) -- every shipped point-cloud SAC / DrQ config -- and the class
 * default (64,128,1024) (pointnet.py:81): fp32 entry points only (pcrl_encoder_fwd_f32 / pcrl_encoder_bwd_f32, the latter
 * needs `pooled`), no feature-head epilogue, not part of pcrl_update_step_*.  Anything else returns PCRL_E_ARG. */
typedef struct pcrl_encoder_weights {
    int32_t c_in, c1, c2, c3;
    const float *w0, *b0, *w1, *g1, *be1, *w2, *g2, *be2;
    float eps;
    int32_t _pad;
} pcrl_encoder_weights;

const char* pcrl_last_error(void);
int pcrl_version(void);

/* Bytes needed for the MFMA-operand-ordered weight image and for the forward
 * scratch (cross-workgroup partial maxima when a cloud is split). */
int pcrl_encoder_packed_bytes(int32_t c_in, int32_t c1, int32_t c2, int32_t c3, size_t* bytes);
int pcrl_encoder_fwd_workspace_bytes(int32_t B, int32_t N, int32_t c3, size_t* bytes);

/* Re-order the weights into the operand order the forward kernel streams.
 * Must be re-run whenever the weights change (after every optimizer step). */
int pcrl_encoder_pack_weights_f32(const pcrl_encoder_weights* w, void* packed, size_t packed_bytes, void* stream);
/* Column-gather jobs riding on the NEXT pcrl_encoder_pack_weights_f32 launch of this host thread (extra workgroups; the step re-packs
 * at the start of each phase anyway): columns [col0, col0 + ncols) of `heads` row-major matrices src + h * head_stride [rows][ld]
 * -> dst [heads][ncols][rows].  The update step uses it for the ACTION columns of the Q heads' first layer (LinearMLP's linear0 over
 * Visuomotor's [feature | state | action] input, mlp.py:97-100, visuomotor.py:130-144), which pcrl_policy_tail_fwd_fold_f32 and
 * pcrl_policy_tail_bwd_f32 contract row-wise.  n = 0 withdraws; pcrl_encoder_pack_flush_cols runs jobs still pending (a phase whose
 * weights needed no re-pack) as a launch of their own. */
typedef struct pcrl_col_gather { const float* src; int64_t head_stride; int32_t heads, rows, ld, col0, ncols, _pad; float* dst; } pcrl_col_gather;
int pcrl_encoder_pack_attach_cols(const pcrl_col_gather* jobs, int32_t n);
int pcrl_encoder_pack_flush_cols(void* stream);
/* The same re-pack (with whatever column-gather jobs are attached at that moment) handed to this host thread's NEXT replay sampling
 * launch (pcrl_replay_sample_gather / _state / pcrl_replay_gather) instead of a launch of its own: it runs as extra workgroups of that
 * launch -- the step's first launch (ReplayMemory.sample, replay_buffer.py:297-322) reads nothing the pack writes, and the encoder
 * forward that needs the image (pointnet.py:148-151) comes after it on the stream.  pcrl_encoder_pack_flush_pending launches a job no
 * sampling launch has taken (the replay had nothing to gather, or was not this library's) and is a no-op otherwise; a second attach
 * while one is pending is an error. */
int pcrl_encoder_pack_attach_to_gather(const pcrl_encoder_weights* w, void* packed, size_t packed_bytes);
int pcrl_encoder_pack_flush_pending(void* stream);
int pcrl_encoder_pack_drop_pending(void);     /* forget a pending job without running it (the caller's capture was aborted) */

/* Fused PointNet encoder forward, fp32:
 *   preprocess (pointnet.py:49-73) -> [augment] -> conv0+ReLU -> conv1+LN1d+ReLU ->
 *   conv2+LN1d+ReLU (mlp.py:43-56, nn_layer.py:207-219) -> max over N with first-index
 *   argmax (pointnet.py:151).
 * pooled [B,c3] f32, argmax [B,c3] int32 (the index torch's autograd keeps as int64). */
int pcrl_encoder_fwd_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug /* may be NULL */,
                         const pcrl_encoder_weights* w, const void* packed,
                         float* pooled, int32_t* argmax,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Mixed-precision forward (BASELINE.json config 3): conv1 and conv2 contract bf16 operands on v_mfma_f32_32x32x16_bf16
 * with fp32 accumulation -- weights rounded (RNE) once by pcrl_encoder_pack_weights_f32 into a second image, activations
 * rounded as they are fed to the next layer; conv0 (raw coordinates), both LayerNorms, ReLU and the max-pool stay fp32.
 * Same arguments and outputs as pcrl_encoder_fwd_f32; results differ from it by bf16 rounding (tests: |diff| <= 3e-2
 * on O(1) outputs against a torch emulation of the same rounding points, argmax agreement >= 95 %). */
int pcrl_encoder_fwd_bf16(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug /* may be NULL */,
                         const pcrl_encoder_weights* w, const void* packed,
                         float* pooled, int32_t* argmax,
                         void* workspace, size_t workspace_bytes, void* stream);

/* EXPERIMENTAL split-precision forward: conv1 / conv2 in ~fp32 accuracy on the bf16 matrix cores -- every fp32 weight and
 * activation is the exact sum of three bf16 terms (8 significand bits each), six of the nine term products are kept (what is
 * dropped is below 3 x 2^-24 of |w| |a| per product), fp32 accumulation.  2.7x less matrix time than the exact fp32 kernel,
 * results within ~1e-6 of it but NOT bit-comparable (a max-pool near-tie of that size can pick another point): offered next
 * to pcrl_encoder_fwd_f32, which stays the default; the backward is pcrl_encoder_bwd_f32.  Same arguments and outputs. */
int pcrl_encoder_fwd_f32split(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug /* may be NULL */,
                              const pcrl_encoder_weights* w, const void* packed,
                              float* pooled, int32_t* argmax,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Backward of pcrl_encoder_fwd_f32split: the recompute uses the same split arithmetic (bit-identical to that forward, so ReLU masks
 * and argmax relations are its own) and so do the two data-gradient GEMMs; the weight-gradient GEMMs are exact fp32.  EXPERIMENTAL. */
int pcrl_encoder_bwd_f32split(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                              const pcrl_encoder_weights* w, const void* packed,
                              const int32_t* argmax, const float* grad_pooled, const float* pooled,
                              float* grads, int32_t* n_active,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Number of floats of the flat encoder gradient, laid out in the reference's parameter order
 * inside visual_nn.conv.mlp: conv0.weight, conv0.bias, conv1.weight, norm1.weight, norm1.bias,
 * conv2.weight, norm2.weight, norm2.bias. */
int pcrl_encoder_num_grads(int32_t c_in, int32_t c1, int32_t c2, int32_t c3, size_t* n);
int pcrl_encoder_bwd_workspace_bytes(int32_t B, int32_t c_in, int32_t c1, int32_t c2, int32_t c3, size_t* bytes);

/* Encoder backward, fp32: gradient of the shared per-point MLP's parameters given d(loss)/d(pooled).
 * Replaces autograd through feature.max(-1), LayerNorm1D, ReLU and Conv1d(k=1)
 * (pointnet.py:148-151, nn_layer.py:207-219, mlp.py:43-56).  Exact: the max-pool routes gradient to
 * at most c3 points per cloud (argmax), so only those points are recomputed and back-propagated.
 * `clouds`/`aug` must describe the same inputs (and the same noise) as the forward call that
 * produced `argmax`.  grads [pcrl_encoder_num_grads] f32 is overwritten; n_active [B] int32 (optional)
 * receives the number of points per cloud that received gradient.  Deterministic (no float atomics).
 * pooled [B, c3] (optional, may be NULL): the forward's output for the same inputs.  With it (and for batches of up to 2 048
 * clouds) the backward runs in Gram form (csrc/encoder_bwd_gram.h): the forward's own values decide each channel's ReLU, channels
 * the forward left at zero are dropped up front (n_active then counts the distinct argmax points of the remaining channels),
 * and the last layer's recompute and W2^T dz2 GEMM are replaced by one C2 -> C2 layer with M = W2^T W2 -- same gradients to
 * ~5e-7 of each tensor's largest entry, 25-35 % less time.  Without it the round-2 kernels run (n_active = all distinct argmax
 * points).  The f32split and bf16 entry points take the same path (see pcrl_encoder_bwd_bf16). */
int pcrl_encoder_bwd_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                         const pcrl_encoder_weights* w, const void* packed,
                         const int32_t* argmax, const float* grad_pooled, const float* pooled,
                         float* grads, int32_t* n_active,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The same backward as two calls, so that the part that does not need d(loss)/d(pooled) can run early (on another stream, or as
 * a forked branch of a captured graph) while the heads' backward is still producing it: `prepare` is the launch that turns the
 * forward's argmax / pooled into the per-cloud lists of gradient-carrying points and builds M = W2^T W2; `prepared` is everything
 * after it and must be ordered behind `prepare` on the same workspace, with the same clouds / aug / w / argmax / pooled and the
 * weights unchanged in between.  Gram form only (PCRL_E_ARG where pcrl_encoder_bwd_f32 would run the round-2 kernels).
 * prepare + prepared == pcrl_encoder_bwd_f32 bit for bit. */
int pcrl_encoder_bwd_prepare_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                 const pcrl_encoder_weights* w, const void* packed,
                                 const int32_t* argmax, const float* pooled,
                                 void* workspace, size_t workspace_bytes, void* stream);
int pcrl_encoder_bwd_prepared_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                  const pcrl_encoder_weights* w, const void* packed,
                                  const int32_t* argmax, const float* grad_pooled, const float* pooled,
                                  float* grads, int32_t* n_active,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* A row-wise LayerNorm backward (the arguments of pcrl_layernorm_rows_bwd_partials_f32 below: PointNet.final_mlp[1], pointnet.py:110) handed
 * over to the NEXT pcrl_encoder_bwd_prepare_f32 / pcrl_encoder_bwd_* call of this host thread: it runs as extra workgroups of that call's prep
 * launch -- which reads nothing the LayerNorm backward writes and writes nothing it reads -- instead of as a launch of its own (one graph node
 * fewer per update step); where the call launches no such prep kernel (round-2 kernels, the wide last layer, B = 0) it is launched on its
 * own in front of everything.  Bit-identical results either way.  (NULL dy0 with M = 0 drops a pending job.) */
int pcrl_encoder_bwd_attach_ln_bwd(const float* dy0, const float* dy1, int64_t lddy, const float* xhat, const float* rstd,
                                   const float* gamma, int32_t M, int32_t F, float* dx, int64_t lddx,
                                   void* workspace, size_t workspace_bytes);
/* Which kernels the Gram-form backward above launches for mlp_spec = [64 | 128, 128, 256] in exact fp32 (and for the bf16 mode's backward,
 * which runs them too).  mode 1 (default): launches of at most two 32-point tiles per CU (up to 64 clouds on 256 CUs: the per-GPU shares of
 * a data-parallel batch) take the team kernel of csrc/encoder_bwd_fused.h -- the per-point chain and the weight-gradient sums in one
 * launch, four waves per tile, no operand pieces in global memory --, larger launches the points / wgrad / reduce launches of
 * csrc/encoder_bwd_gram.h (faster there: four independent tiles per CU).  mode 0: never the team kernel; mode 2: wherever it is built.
 * Same gradients to fp32 summation order; each path bitwise reproducible.  PROCESS-WIDE (one atomic read by every later backward of
 * any host thread; with pcrl_gemm_set_tile64_min the library's only mutable global state): a knob for tests and A/B measurements, not
 * a per-call option.  Returns PCRL_OK, PCRL_E_ARG for another mode. */
int pcrl_encoder_bwd_set_fused(int32_t mode);
/* What the last pcrl_encoder_bwd_* call of this host thread launched: 1 the round-2 kernels (no pooled values given), 2 the Gram form's points /
 * wgrad / reduce launches, 3 the Gram form's team kernel; 0 before the first call.  For tests of the selection above. */
int pcrl_encoder_bwd_last_schedule(void);
/* Backward of pcrl_encoder_fwd_bf16.  With `pooled` (and up to 2 048 clouds; PCRL_BWD_BF16_GRAM, default 1): the fp32 Gram-form
 * backward of pcrl_encoder_bwd_f32 at the bf16 forward's routing (its argmax, its pooled > 0 decisions) -- the exact fp32 gradient
 * along that routing (tests: 2e-4 of each tensor's largest entry against fp32 autograd routed the same way; up to ~6e-2 from autograd
 * through an emulation of the bf16 roundings, which is what bf16 operands change in the two lower layers).  Without `pooled`, or with
 * PCRL_BWD_BF16_GRAM=0, the round-2 kernels: the forward of the active points is recomputed with the same bf16 contractions (so
 * LayerNorm inputs, ReLU masks and argmax relations are the forward's) and the two data-gradient GEMMs contract bf16 too
 * (gradients rounded as they enter, fp32 accumulate); the weight-gradient GEMMs are fp32 on the unrounded operands and the
 * result is the gradient w.r.t. the fp32 master weights (roundings straight-through; within 3e-2 of the emulation's autograd).
 * `argmax` must come from the bf16 forward. */
int pcrl_encoder_bwd_bf16(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                         const pcrl_encoder_weights* w, const void* packed,
                         const int32_t* argmax, const float* grad_pooled, const float* pooled,
                         float* grads, int32_t* n_active,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Fused Adam (+ Polyak + gradient 2-norm) over one flat parameter buffer.
 * Replaces torch.optim.Adam over one parameter group per tensor (build_optimizer,
 * pyrl/utils/torch/optimizer_utils.py:43-57; Adam defaults: no weight decay, no amsgrad), the
 * per-parameter soft_update (pyrl/utils/torch/ops.py:59-90) and grad_norm
 * (pyrl/utils/torch/module_utils.py:40-45).
 *   g = grad * grad_scale (1/world after a sum all-reduce); m, v, param updated in place;
 *   step_counter (device int32) is incremented first and used for the bias corrections, so the call
 *   can be replayed from a hipGraph; grad_norm_out (device float, optional) <- ||g||_2;
 *   target (optional): for target_begin <= i < target_end,
 *   target[i - target_begin] <- (1 - tau) * target[..] + tau * param_new[i]. */
/* A pass whose second half (sum of the per-block partial sums of grad^2 -> grad_norm_out, step_counter += 1) has been
 * deferred: filled by pcrl_adam_step_f32 when defer_finalize != NULL and handed to pcrl_gather_scalars_f32 at the end of
 * the step, which saves one dependent launch per optimizer per step.  Until then the step counter still holds the
 * number of COMPLETED steps, and the workspace must stay untouched. */
typedef struct pcrl_adam_pending { const float* partial; int32_t n_partial, _pad; float* grad_norm_out; int32_t* step_counter; } pcrl_adam_pending;
int pcrl_adam_workspace_bytes(size_t n, size_t* bytes);
int pcrl_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                       float lr, float beta1, float beta2, float eps, float grad_scale,
                       int32_t* step_counter, float* grad_norm_out,
                       float* target, size_t target_begin, size_t target_end, float tau,
                       void* workspace, size_t workspace_bytes, pcrl_adam_pending* defer_finalize, void* stream);
/* The same pass with a second, small optimizer riding on the launch in one extra workgroup: SAC's temperature next to the actor
 * (sac.py:184-195 steps actor_optim and alpha_optim back to back; log_alpha is one float with its own betas, moments and step
 * count).  rider->partial: one float of scratch; rider_defer as defer_finalize (NULL: finished right away by a second launch). */
typedef struct pcrl_adam_rider {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq; size_t n;      /* 1 <= n <= 4096 */
    float lr, beta1, beta2, eps, grad_scale; int32_t _pad;
    int32_t* step_counter; float* grad_norm_out; float* partial;
} pcrl_adam_rider;
int pcrl_adam_step_rider_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                             float lr, float beta1, float beta2, float eps, float grad_scale,
                             int32_t* step_counter, float* grad_norm_out,
                             float* target, size_t target_begin, size_t target_end, float tau,
                             void* workspace, size_t workspace_bytes, pcrl_adam_pending* defer_finalize,
                             const pcrl_adam_rider* rider, pcrl_adam_pending* rider_defer, void* stream);
/* A step that publishes its metrics BEFORE its last optimizer pass (the host's turn-around to the next step then overlaps that pass):
 *   pcrl_grad_norm_partials_f32  -- the pass's gradient-norm partial sums without the pass: same grid, element order and reduction tree as
 *       pcrl_adam_step_f32, so the norm (module_utils.py:40-45) is bit for bit the one the pass would have reported; `pending` is filled as by
 *       defer_finalize above -- except that this launch ADVANCES the pass's step count itself (pending->step_counter comes back NULL) -- and goes to
 *       pcrl_gather_scalars_f32 / pcrl_adam_step_published_gather_f32, which form the norm.  An optional rider (the
 *       temperature: its whole Adam pass, sac.py:192-195, so that alpha = exp(log_alpha) is final before the metrics are gathered) runs in one
 *       extra workgroup, its own `rider_pending` filled alike.
 *   pcrl_adam_step_published_f32 -- the pass itself, launched AFTER that gather launch: reads *step_counter as already advanced (bias
 *       corrections of step *step_counter), writes no partial sums and touches no counter.
 * Parameters, moments and the Polyak target end up bit-identical to pcrl_adam_step_f32's. */
int pcrl_grad_norm_partials_f32(const float* grad, size_t n, float grad_scale, int32_t* step_counter, float* grad_norm_out,
                                void* workspace, size_t workspace_bytes, pcrl_adam_pending* pending,
                                const pcrl_adam_rider* rider, pcrl_adam_pending* rider_pending, void* stream);
int pcrl_adam_step_published_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                 float lr, float beta1, float beta2, float eps, float grad_scale, const int32_t* step_counter,
                                 float* target, size_t target_begin, size_t target_end, float tau, void* stream);
/* The same pass with pcrl_gather_scalars_host_f32 (declared below; same arguments) done by its FIRST workgroup instead of by a launch of its own in
 * front of it: the metrics leave when the pass starts.  None of the pending passes may be this pass itself (its step count was advanced by
 * pcrl_grad_norm_partials_f32: that call's `pending` carries a NULL step counter and only has its norm formed here). */
int pcrl_adam_step_published_gather_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                        float lr, float beta1, float beta2, float eps, float grad_scale, const int32_t* step_counter,
                                        float* target, size_t target_begin, size_t target_end, float tau,
                                        const float* const* src, float* const* dst, const int32_t* take_exp, int32_t n_scalars,
                                        const pcrl_adam_pending* pending, int32_t n_pending, float* host_out, void* stream);
/* target <- (1 - tau) target + tau src  (soft_update / hard_update with tau = 1, ops.py:59-100). */
int pcrl_polyak_f32(float* target, const float* src, size_t n, float tau, void* stream);

/* ---- stand-alone memory-shaped kernels -----------------------------------------------------------
 * Symmetric max-pool with first-index argmax over a materialised [rows, N] f32 tensor (rows = B*c) and
 * its backward: `feature.max(-1)` (pointnet.py:151).  torch CPU rules: first index among equal values;
 * a NaN wins and the first NaN's index is returned.  HBM-bound: 4 B/element read (forward) or written
 * (backward). */
int pcrl_segmax_fwd_f32(const float* x, int64_t rows, int32_t N, float* out, int32_t* idx, void* stream);
int pcrl_segmax_bwd_f32(const float* grad_out, const int32_t* idx, int64_t rows, int32_t N, float* grad_x, void* stream);
/* RandomJitterPoints / GlobalRotScaleTrans applied to a [B,3,N] f32 tensor (in place when xyz_out ==
 * xyz_in): pcd_aug.py:306-327, 84-123.  Same pcrl_aug_desc semantics as the fused encoder load. */
int pcrl_augment_xyz_f32(const float* xyz_in, float* xyz_out, int32_t B, int32_t N, const pcrl_aug_desc* aug, void* stream);
/* GlobalRotScaleTrans.process_single's matrix draw (pcd_aug.py:178-196; batch_rot_with_axis, pyrl/utils/torch/ops.py:171-183) as
 * one launch: mat [B][3][4] row-major = [diag(s) R(angle) | t] with angle ~ U(rot_range), s_i ~ U(scale_range), t_i = (U(0,1) - 0.5)
 * * 2 * translation_range[i] (zero for the LAST cloud unless shift_height) -- each range pointer (host, 2 / 2 / 3 floats) may be
 * NULL: no rotation -> identity block (the scale then has nothing to act on, as in the reference), no translation -> zero.
 * Philox4x32-10 keyed by (seed, offset, cloud); offset_ptr (device, may be NULL) overrides offset at run time, so a launch
 * replayed from a hipGraph draws fresh matrices.  Feeds pcrl_aug_desc.affine. */
int pcrl_affine_sample_f32(float* mat, int32_t B, int32_t rot_axis, const float* rot_range, const float* scale_range,
                           const float* translation_range, int32_t shift_height, uint64_t seed, uint64_t offset,
                           const uint64_t* offset_ptr, void* stream);
/* Two draws of the same transform in ONE launch -- DrQ augments obs and next_obs with independent draws back to back (drq.py:62-75): mat
 * from (seed, offset), mat2 from (seed2, offset2), each exactly what pcrl_affine_sample_f32 writes for those arguments. */
int pcrl_affine_sample_pair_f32(float* mat, float* mat2, int32_t B, int32_t rot_axis, const float* rot_range, const float* scale_range,
                                const float* translation_range, int32_t shift_height, uint64_t seed, uint64_t offset, uint64_t seed2,
                                uint64_t offset2, const uint64_t* offset_ptr, void* stream);
/* ColorJitterPoints on a [B,3,N] uint8 tensor (strides in elements).  pcrl_color_contrast_mean_u8 applies the steps that
 * precede the contrast step and returns each cloud's mean grayscale value at that point (mean_out [B] f32; the
 * reduction over the N points of a cloud is the one thing the fused encoder load cannot do on the fly);
 * pcrl_color_jitter_u8 materialises the jittered tensor (rgb_out may alias rgb_in).  Only flags / color_* of `aug` are read. */
int pcrl_color_contrast_mean_u8(const uint8_t* rgb, int64_t stride_b, int64_t stride_c, int64_t stride_n, int32_t B, int32_t N,
                                const pcrl_aug_desc* aug, float* mean_out, void* stream);
int pcrl_color_jitter_u8(const uint8_t* rgb_in, uint8_t* rgb_out, int64_t stride_b, int64_t stride_c, int64_t stride_n, int32_t B, int32_t N,
                         const pcrl_aug_desc* aug, void* stream);

/* ---- dense heads --------------------------------------------------------------------------------
 * Batched fp32 GEMM  C[z] = epilogue(A[z] . B[z])  with generic operand strides (elements):
 *   A[m][k] at A + z*a_batch_stride + m*a_stride_m + k*a_stride_k,  B[k][n] likewise,  C row-major (ldc).
 * Epilogue: + bias[n]; relu; * (mask[m][n] > 0); accumulate into C.  ones_col >= 0 makes column
 * `ones_col` of B read as 1 for every k (bias gradient as an extra output column).
 * A Linear layer y = x W^T + b of the reference's LinearMLP heads (pyrl/networks/backbones/mlp.py:97-100)
 * maps to:  forward  A = x, B[k][n] = W[n][k] (b_stride_k = 1, b_stride_n = K);
 *           dx = dy W:   A = dy, B = W (b_stride_k = K_in, b_stride_n = 1), mask = the layer input
 *           (its ReLU output);  dW|db = dy^T [x | 1]:  A[m][k] = dy[k][m], B = x, ones_col = K_in. */
typedef struct pcrl_gemm_desc {
    const float* A; const float* B; float* C;
    const float* bias; const float* mask;
    int32_t M, N, K, batch;
    int64_t a_stride_m, a_stride_k, b_stride_k, b_stride_n, ldc, ld_mask;
    int64_t a_batch_stride, b_batch_stride, c_batch_stride, bias_batch_stride, mask_batch_stride;
    int32_t relu, ones_col, accumulate, _pad;
    float* C_ones;                 /* when non-NULL, output column `ones_col` is written to C_ones[m] instead of C */
    int64_t c_ones_batch_stride;
} pcrl_gemm_desc;
int pcrl_gemm_f32(const pcrl_gemm_desc* d, void* stream);
/* Up to 4 INDEPENDENT problems in one launch (no problem may read what another writes): dW and dx of
 * one layer, or the online and target Q heads of one layer.  Same per-problem semantics as above. */
int pcrl_gemm_group_f32(const pcrl_gemm_desc* descs, int32_t n, void* stream);
/* How a launch picks the tile path of each problem (csrc/dense.hip, csrc/dense_wtile.h); the results of all paths agree to fp32
 * summation order.  By default: forward-shaped (A, B k-contiguous) and data-gradient-shaped (A k-contiguous, B contiguous along n)
 * problems with K >= 128, K % 4 == 0 run as wave-private staged tiles of 16 x 16 ... 32 x 64 outputs -- the finest shape that still
 * gives the chip at most one workgroup per CU --, weight-gradient-shaped ones (both operands contiguous along their row index, M and
 * the real column count multiples of 4 and >= 64) as LDS panels of 64 x 64 / 64 x 128 outputs, the largest forward-shaped problems as
 * LDS-staged 64 x 64 tiles when the launch has at least `min_tiles` of them (default 192), everything else as 32 x 32 split-K tiles.
 * Knob for tests and benchmarks: min_tiles <= 1 forces the 64 x 64 staged tiles for every problem they can compute (M, N >= 48,
 * K >= 64, any operand orientation); min_tiles >= 2^30 selects the paths of rounds 1-4 only (32 x 32 split-K tiles, a tile per wave
 * for the weight gradients).  PROCESS-WIDE (an atomic).  Returns the previous value; a negative argument only queries. */
int pcrl_gemm_set_tile64_min(int32_t min_tiles);
/* What pcrl_gemm_group_f32 would launch for these problems, without launching: out[3 i .. 3 i + 2] = {path, tile shape, workgroups}
 * of problem i (path: 0 32 x 32 split-K, 1 64 x 64 staged, 2 / 3 a tile per wave, 4 wave-private staged tiles, 5 weight-gradient
 * panels; -1 for an empty problem).  Tests assert with it that a shape reaches the path they mean to cover. */
int pcrl_gemm_group_plan_f32(const pcrl_gemm_desc* descs, int32_t n, int32_t* out);

/* Row-wise LayerNorm over F <= 256 features (PointNet.final_mlp[1] = nn.LayerNorm(out), pointnet.py:110).
 * The result is written to n_dst <= 4 destinations (dst[i] with leading dimension ld_dst[i]): the
 * concatenated input buffers of the actor / Q heads (Visuomotor's torch.cat, visuomotor.py:130-141).
 * xhat [M,F] and rstd [M] are saved for the backward.  Backward: dy = dy0 (+ dy1), dx, and
 * dgamma/dbeta (optionally accumulated); workspace >= ceil(M/4)*2*F floats. */
int pcrl_layernorm_rows_fwd_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, int32_t M, int32_t F,
                                float eps, float* const* dst, const int64_t* ld_dst, int32_t n_dst,
                                float* xhat, float* rstd, void* stream);
/* Up to 3 row batches through the same LayerNorm in one launch (features of s and s' in the critic phase), each
 * optionally passing up to two blocks of columns through unchanged into a destination (robot state and replay
 * actions of Visuomotor's concatenations, visuomotor.py:130-141): cat_dst[c][m][0..cat_n[c]) = cat_src[c][m][..]. */
typedef struct pcrl_ln_job {
    const float* x; int64_t ldx; int32_t M, n_dst;
    float* dst[4]; int64_t ld_dst[4];
    float* xhat; float* rstd;
    const float* cat_src[2]; float* cat_dst[2]; int64_t cat_ld_src[2], cat_ld_dst[2]; int32_t cat_n[2];
    int32_t cat_row_div[2];        /* > 1: destination row m takes source row m / cat_row_div (a source stored once per sample, used by each of its augmentations) */
} pcrl_ln_job;
int pcrl_layernorm_rows_fwd_multi_f32(const pcrl_ln_job* jobs, int32_t n_jobs, const float* gamma, const float* beta, int32_t F,
                                      float eps, void* stream);
int pcrl_layernorm_rows_bwd_f32(const float* dy0, const float* dy1, int64_t lddy, const float* xhat, const float* rstd,
                                const float* gamma, int32_t M, int32_t F, float* dx, int64_t lddx,
                                float* dgamma, float* dbeta, int32_t accumulate,
                                void* workspace, size_t workspace_bytes, void* stream);

/* The backward kernel alone: the per-block partial sums of dgamma / dbeta stay in `workspace` as [ceil(M/4)][2][F] for a later
 * pcrl_colsum_jobs_f32 launch (which can reduce several such leftovers at once). */
int pcrl_layernorm_rows_bwd_partials_f32(const float* dy0, const float* dy1, int64_t lddy, const float* xhat, const float* rstd,
                                         const float* gamma, int32_t M, int32_t F, float* dx, int64_t lddx,
                                         void* workspace, size_t workspace_bytes, void* stream);
/* PointNet.final_mlp (Linear(c3, F) + LayerNorm(F), pointnet.py:152-153) as the epilogue of the encoder launch: the
 * workgroup that finishes a cloud's max-pool (or the merge launch, for clouds split over workgroups) also forms
 * y = weight . pooled + bias (fixed summation order), normalises it and writes the rows exactly as
 * pcrl_layernorm_rows_fwd_multi_f32 would -- destinations, xhat / rstd saves and pass-through columns of `job[i]` (its x / ldx
 * are ignored), for the clouds [begin[i], begin[i] + job[i].M) of the launch; row = cloud - begin[i].  Saves the feature GEMM
 * launch and the LayerNorm launch of every encoder pass. */
typedef struct pcrl_feature_head {
    const float* weight;           /* [F][c3] row-major (final_mlp.0.weight) */
    const float* bias;             /* [F] */
    const float* gamma; const float* beta;    /* [F] (final_mlp.1) */
    int32_t F;                     /* 1 <= F <= 256 */
    float eps;
    int32_t n_ranges;              /* 1 or 2 */
    int32_t begin[2];
    pcrl_ln_job job[2];
} pcrl_feature_head;
int pcrl_encoder_fwd_head_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug, const pcrl_encoder_weights* w, const void* packed,
                              float* pooled, int32_t* argmax, const pcrl_feature_head* head,
                              void* workspace, size_t workspace_bytes, void* stream);
int pcrl_encoder_fwd_head_bf16(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug, const pcrl_encoder_weights* w, const void* packed,
                               float* pooled, int32_t* argmax, const pcrl_feature_head* head,
                               void* workspace, size_t workspace_bytes, void* stream);
int pcrl_encoder_fwd_head_f32split(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug, const pcrl_encoder_weights* w, const void* packed,
                                   float* pooled, int32_t* argmax, const pcrl_feature_head* head,
                                   void* workspace, size_t workspace_bytes, void* stream);

/* Fixed-order column reductions of per-workgroup partial results, up to 12 jobs in one launch:
 *   out[c] = scale * (op == 0 ? sum : max)_{b < nblk} part[b * blk_stride + c]   for c < ncols. */
typedef struct pcrl_colsum_job { const float* part; int64_t blk_stride; int32_t nblk, ncols; float* out; float scale; int32_t op; } pcrl_colsum_job;
int pcrl_colsum_jobs_f32(const pcrl_colsum_job* jobs, int32_t n, void* stream);
/* The same jobs without a launch of their own: they are copied and run by extra workgroups of the gradient-reduce launch of the NEXT
 * pcrl_encoder_bwd_{f32,bf16,f32split} / pcrl_encoder_bwd_prepared_f32 call of this host thread (same arithmetic, same order; the
 * partials must be complete by then and `out` is valid when that call's launches are).  n = 0 withdraws them.  The update step
 * uses it for the leftovers of pcrl_layernorm_rows_bwd_partials_f32 and pcrl_q_tail_critic_f32, which only the optimizer reads. */
int pcrl_encoder_bwd_attach_colsum(const pcrl_colsum_job* jobs, int32_t n);

/* ---- head tails: the last Linear of a head fused with what follows it (H = hidden width, multiple of 256) ------------------------
 * pcrl_q_tail_critic_f32: for both Q heads h (second head at + *_head_stride floats):
 *   q_next[m][h] = h2_target[h][m] . w2_target[h] + b2_target[h],  q[m][h] = h2[h][m] . w2[h] + b2[h]   (LinearMLP's last
 *   Linear, pyrl/networks/backbones/mlp.py:97-100), then exactly pcrl_sac_critic_loss_f32's target / loss / dq (sac.py:125-157,
 *   drq.py:76-103; group in {1, 2, 4}), dh2[h][m] = dq[m][h] w2[h] (.) [h2[h][m] > 0], and per-workgroup partials:
 *   part [ceil(M/4)][2][H + 4] (columns < H: dW2[h] = sum_m dq[m][h] h2[h][m]; column H: db2[h]) and stat_part [ceil(M/4)][4]
 *   = {sum (q-y)^2, max |q-y|, sum min_h q, sum y} -- reduce with pcrl_colsum_jobs_f32 (sizes: pcrl_q_tail_workspace_floats).
 * pcrl_q_tail_actor_f32: q = Q(s, pi(s)) likewise, dq = d(-mean_m min_h q)/dq (sac.py:177-183), dh2, d_neglogp = -alpha / M,
 *   stat_part [ceil(M/4)][4] = {sum min_h q, sum neg_logp, 0, 0}; pcrl_actor_finalize_f32 turns those into the actor / temperature
 *   losses and d(alpha_loss)/d(log_alpha) (sac.py:183-195), i.e. the two together are pcrl_sac_actor_loss_f32 + the GEMMs around it.
 * pcrl_policy_tail_fwd_f32: feat = h2 w2^T + b2 ([M][2A], the policy's last Linear) followed by pcrl_tanh_gaussian_fwd_f32 /
 *   pcrl_tanh_gaussian_sample_fwd_f32 on it (eps == NULL: in-kernel Philox draws written to eps_out). */
int pcrl_q_tail_workspace_floats(int32_t M, int32_t H, size_t* part_floats, size_t* stat_floats);
int pcrl_q_tail_critic_f32(const float* h2, int64_t h2_head_stride, const float* w2, const float* b2, int64_t w_head_stride,
                           const float* h2_target, int64_t h2_target_head_stride, const float* w2_target, const float* b2_target,
                           int64_t w_target_head_stride, const float* neg_logp_next, const float* rewards, const uint8_t* dones,
                           int32_t rd_row_div, const float* log_alpha, float gamma, float reward_scale, int32_t ignore_dones,
                           int32_t group, int32_t M, int32_t H, float* q, int64_t ld_q, float* q_target, float* dq, int64_t ld_dq,
                           float* dh2, int64_t dh2_head_stride, float* part, float* stat_part, void* stream);
int pcrl_q_tail_actor_f32(const float* h2, int64_t h2_head_stride, const float* w2, const float* b2, int64_t w_head_stride,
                          const float* neg_logp, const float* log_alpha, int32_t M, int32_t H, float* q, int64_t ld_q,
                          float* dq, int64_t ld_dq, float* dh2, int64_t dh2_head_stride, float* d_neglogp, float* stat_part,
                          void* stream);
/* pcrl_q_tail_actor_f32 whose launch also writes, from extra workgroups, columns [col0, col0 + ncols) of the two heads' FIRST
 * layer weight w0 [2][H][ld_w0] as the compact image cols_out [2][ncols][H] -- the action columns pcrl_policy_tail_bwd_f32 reads. */
int pcrl_q_tail_actor_cols_f32(const float* h2, int64_t h2_head_stride, const float* w2, const float* b2, int64_t w_head_stride,
                               const float* neg_logp, const float* log_alpha, int32_t M, int32_t H, float* q, int64_t ld_q,
                               float* dq, int64_t ld_dq, float* dh2, int64_t dh2_head_stride, float* d_neglogp, float* stat_part,
                               const float* w0, int64_t w0_head_stride, int32_t ld_w0, int32_t col0, int32_t ncols, float* cols_out,
                               void* stream);
/* The actor phase's backward between the Q heads' dh1 and the policy's dh2 in ONE launch (autograd of sac.py:177-189 through
 * visuomotor.py:130-144, gaussian.py:83-87, mlp.py:97-100): d_action = sum_h dh1_h W0_h[:, action columns] (w0_action_cols [2][A][H]
 * from pcrl_q_tail_actor_cols_f32), pcrl_tanh_gaussian_bwd_f32's arithmetic -> d_feat [M][2A], and the policy's last layer's data
 * gradient dh2 = (d_feat w2) (.) [h2 > 0] (w2 [2A][H]).  With stat_part != NULL one more workgroup does pcrl_actor_finalize_f32. */
int pcrl_policy_tail_bwd_f32(const float* dh1, int64_t dh1_head_stride, const float* w0_action_cols, int64_t w0a_head_stride,
                             int32_t M, int32_t H, int32_t A, const float* feat, int64_t ld_feat, const float* eps, const float* saved,
                             const float* scale, float log_std_min, float log_std_max, float epsilon, const float* d_neglogp,
                             float* d_feat, int64_t ld_d_feat, const float* h2, const float* w2, float* dh2,
                             const float* stat_part, const float* log_alpha, float target_entropy, float* alpha_grad, float* stats,
                             void* stream);
int pcrl_actor_finalize_f32(const float* stat_part, int32_t M, const float* log_alpha, float target_entropy, float* alpha_grad,
                            float* stats, void* stream);
int pcrl_policy_tail_fwd_f32(const float* h2, int32_t M, int32_t H, const float* w2, const float* b2, int32_t A, const float* eps,
                             uint64_t seed, const int32_t* step_counter, int32_t draw_id, float* eps_out, const float* scale,
                             const float* bias, float log_std_min, float log_std_max, float epsilon, float* feat, int64_t ld_feat,
                             float* action, int64_t ld_action, float* action2, int64_t ld_action2, float* neg_logp, float* saved,
                             void* stream);
/* pcrl_policy_tail_fwd_f32 that also finishes the FIRST layer of n_heads Q heads on (s, a) for the action it just formed:
 *   h1[h][m][:] = relu(pre[h][m][:] + sum_j action[m][j] w0_action_cols[h][j][:])
 * with pre = [feature | state] W0[:, :F+S]^T + b0 from an earlier GEMM (it does not depend on the action) and w0_action_cols
 * [n_heads][A][H] from pcrl_encoder_pack_attach_cols: Visuomotor's torch.cat([feature, state, action]) + LinearMLP's linear0 + ReLU
 * (visuomotor.py:130-144, mlp.py:97-100) without a launch for a K = 50..220 GEMM between the policy and the Q heads' second layer.
 * pre and h1 may alias.  Built for H = 1024 and M <= 512 (PCRL_E_ARG otherwise: the caller keeps the GEMM). */
int pcrl_policy_tail_fwd_fold_f32(const float* h2, int32_t M, int32_t H, const float* w2, const float* b2, int32_t A, const float* eps,
                             uint64_t seed, const int32_t* step_counter, int32_t draw_id, float* eps_out, const float* scale,
                             const float* bias, float log_std_min, float log_std_max, float epsilon, float* feat, int64_t ld_feat,
                             float* action, int64_t ld_action, float* action2, int64_t ld_action2, float* neg_logp, float* saved,
                             const float* pre, int64_t pre_head_stride, const float* w0_action_cols, int64_t w0a_head_stride,
                             int32_t n_heads, float* h1, int64_t h1_head_stride, void* stream);

/* ---- update tail ---------------------------------------------------------------------------------
 * Squashed-Gaussian policy head, mode "max-entropy" (TanhGaussianHead + ScaledTanhNormal,
 * pyrl/networks/regression_heads/gaussian.py:23-50,83-87; pyrl/utils/torch/distributions.py:89,116-127):
 *   std = exp(clamp(log_std)), u = mean + eps*std, action = tanh(u)*scale + bias,
 *   log p = sum_j [ -(u-mean)^2/(2 std^2) - log std - log sqrt(2 pi) - log(scale*(1-tanh(u)^2) + epsilon) ].
 * feat [B, 2A] = mean | log_std.  The action is written to `action` and optionally `action2` (the Q heads'
 * concatenated input).  saved [B, 2A] keeps tanh(u) | std for the backward.
 * Backward: d_action = d_action0 (+ d_action1), d_neglogp = device scalar shared by all rows. */
int pcrl_tanh_gaussian_fwd_f32(const float* feat, int64_t ld_feat, const float* eps, const float* scale, const float* bias,
                               int32_t B, int32_t A, float log_std_min, float log_std_max, float epsilon,
                               float* action, int64_t ld_action, float* action2, int64_t ld_action2,
                               float* neg_logp, float* saved, void* stream);
/* Same forward with the standard-normal draws made in the kernel (Philox4x32-10 keyed by `seed`, counter =
 * (element, draw_id, *step_counter), Box-Muller) and returned in eps_out [B, A] for the backward.  The reference
 * draws from torch's global generator (distributions.py:122-127); any N(0,1) stream is a valid replacement.
 * step_counter is a device int32 that the caller advances between steps (the critic optimizer's step count),
 * so a captured launch draws fresh noise on every hipGraph replay. */
int pcrl_tanh_gaussian_sample_fwd_f32(const float* feat, int64_t ld_feat, uint64_t seed, const int32_t* step_counter, int32_t draw_id,
                                      float* eps_out, const float* scale, const float* bias,
                                      int32_t B, int32_t A, float log_std_min, float log_std_max, float epsilon,
                                      float* action, int64_t ld_action, float* action2, int64_t ld_action2,
                                      float* neg_logp, float* saved, void* stream);
int pcrl_tanh_gaussian_bwd_f32(const float* feat, int64_t ld_feat, const float* eps, const float* saved, const float* scale,
                               int32_t B, int32_t A, float log_std_min, float log_std_max, float epsilon,
                               const float* d_action0, const float* d_action1, int64_t ld_d_action, const float* d_neglogp,
                               float* d_feat, int64_t ld_d_feat, void* stream);

/* Double-Q TD target + critic loss (sac.py:125-157; drq.py:76-103 with group = num_aug):
 *   y = r*reward_scale + (1-done)*gamma*(min_h q_next + exp(log_alpha)*neg_logp_next)  [mean over each
 *   group of `group` consecutive rows], loss = mse_loss(q, y)*H, dq = d loss / d q,
 *   stats = {loss, max|q-y|, mean_b min_h q, mean y}. */
/* rd_row_div > 1: rewards / dones hold one row per SAMPLE and row b reads entry b / rd_row_div (DrQ without materialising
 * the repeat_interleave of drq.py:61-63); 0 or 1: one entry per row. */
int pcrl_sac_critic_loss_f32(const float* q_next, int64_t ld_q_next, const float* neg_logp_next, const float* rewards,
                             const uint8_t* dones, int32_t rd_row_div, const float* log_alpha, float gamma, float reward_scale,
                             int32_t ignore_dones, int32_t group, const float* q, int64_t ld_q, int32_t B, int32_t H,
                             float* q_target, float* dq, int64_t ld_dq, float* stats, void* stream);
/* Actor and temperature losses (sac.py:177-195): actor_loss = -(mean_b min_h q_pi + alpha*mean neg_logp),
 * alpha_loss = exp(log_alpha)*(entropy - target_entropy); outputs dq_pi, d_neglogp (= -alpha/B),
 * alpha_grad (= d alpha_loss / d log_alpha), stats = {actor_loss, entropy, alpha_loss}. */
int pcrl_sac_actor_loss_f32(const float* q_pi, int64_t ld_q, const float* neg_logp, const float* log_alpha, float target_entropy,
                            int32_t B, int32_t H, float* dq, int64_t ld_dq, float* d_neglogp, float* alpha_grad, float* stats,
                            void* stream);

/* ---- device-resident replay ------------------------------------------------------------------------
 * dst_k[b] = src_k[idx[b]] for every stored key k in one launch (rows of row_bytes bytes; idx is a device
 * array of B row numbers, clamped to [0, capacity)).  Replaces ReplayMemory.sample's per-key numpy take and
 * the host->device copy of the batch (replay_buffer.py:297-322, sac.py:104). */
typedef struct pcrl_gather_seg { const void* src; void* dst; int64_t row_bytes; } pcrl_gather_seg;
int pcrl_replay_gather(const pcrl_gather_seg* segs, int32_t n_segs, const int32_t* idx, int32_t B, int64_t capacity, void* stream);
/* Same gather with the B row numbers drawn in the kernel: uniform with replacement on [0, size) as
 * OneStepTransition does with numpy (sampling_strategy.py:30-31), from Philox4x32-10 keyed by `seed` with counter
 * (b, draw); `draw` is the host's sample-call count.  The rows used are written to idx_out (may be NULL). */
int pcrl_replay_sample_gather(const pcrl_gather_seg* segs, int32_t n_segs, int32_t B, int64_t size, int64_t capacity,
                              uint64_t seed, uint64_t draw, int32_t* idx_out, void* stream);
/* The same launch with its two per-call values read from device memory, so that it can be a node of a replayed hipGraph
 * (the sampling becomes part of the captured update step): state[0] = draw (the launch's last workgroup to finish advances
 * it by one), state[1] = size (the host stores len(buffer) there after every push), state[2] and state[3 .. 3 + B) =
 * workgroup tickets (0 between launches; `state` holds 3 + B words).  Same rows as pcrl_replay_sample_gather(..., size = state[1], draw = state[0], ...). */
int pcrl_replay_sample_gather_state(const pcrl_gather_seg* segs, int32_t n_segs, int32_t B, int64_t capacity, uint64_t seed,
                                    uint64_t* state, int64_t state_words, int32_t* idx_out, void* stream);
/* state_words = the number of 8-byte words `state` holds: PCRL_E_ARG when it is below 3 + B (the kernel would write tickets
 * past the end).  PRECONDITION: every ticket word (state[2 .. 3 + B)) is zero on entry; a launch leaves them zero when it
 * completes, an aborted one may not -- zero them (hipMemsetAsync of state + 2) before sampling again, or the draw count
 * never advances and the same rows are drawn forever. */

/* dst[i][0] = take_exp[i] ? exp(src[i][0]) : src[i][0] for up to 16 device scalars in one launch: the metrics
 * update_parameters returns (sac.py:150-159,199-204) and alpha = exp(log_alpha) (sac.py:196).  Up to 4 deferred optimizer
 * passes (pcrl_adam_pending) are finished first, so their gradient norms can be among the gathered scalars. */
int pcrl_gather_scalars_f32(const float* const* src, float* const* dst, const int32_t* take_exp, int32_t n,
                            const pcrl_adam_pending* pending, int32_t n_pending, void* stream);
/* Same, and the n gathered values are also stored to host_out[0..n) -- pinned host memory the device can address
 * (hipHostMalloc / torch pin_memory) -- each as one 4-byte store.  A host that filled the n slots with the bit pattern
 * 0xFFFFFFFF before the launch reads a metric as soon as its slot holds anything else (a value with that pattern is stored
 * as the canonical NaN): no device->host copy node in the captured step and no stream synchronisation for the `.item()`
 * reads of sac.py:150-159,199-204. */
int pcrl_gather_scalars_host_f32(const float* const* src, float* const* dst, const int32_t* take_exp, int32_t n,
                                 const pcrl_adam_pending* pending, int32_t n_pending, float* host_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCRL_H_ */
