// Fused SAC / DrQ update tail for gfx950: squashed-Gaussian policy head (forward + backward), double-Q
// TD target + critic loss, actor + temperature losses.  Each kernel replaces a chain of ~10-30 tiny
// ATen elementwise / reduction launches (and their autograd twins) in the reference:
//   tanh_gaussian_*  : TanhGaussianHead + ScaledTanhNormal.rsample_with_log_prob
//                      (pyrl/networks/regression_heads/gaussian.py:23-50,83-87; pyrl/utils/torch/distributions.py:89,116-127)
//   critic_loss      : min over heads, + alpha * neg_logp, reward/done mix, [DrQ: mean over augmentations],
//                      mse_loss * 2 and the logged statistics (sac.py:125-157, drq.py:76-103)
//   actor_loss       : -(mean min Q + alpha * entropy), alpha loss (sac.py:177-195)
// All are latency-bound (a few KB); reductions run in ONE workgroup in a fixed order (deterministic).
#include "common.h"

namespace pcrl {

constexpr float kHalfLog2Pi = 0.91893853320467274178f;   // log(sqrt(2*pi))

struct TgFwdParams {
    const float* feat; long long ld_feat;   // [B][2A]: mean | log_std
    const float* eps;                       // [B][A] standard-normal draws
    const float* scale; const float* bias;  // [A]
    int B, A; float ls_min, ls_max, epsilon;
    float* act0; long long ld0; float* act1; long long ld1;   // action written to up to two places
    float* neg_logp;                        // [B]
    float* saved;                           // [B][2A]: tanh(u) | std   (backward)
    // eps == NULL: draws come from Philox4x32-10 keyed by `seed`, counter (element, draw_id, *step) and are
    // written to eps_out (the backward needs them)
    float* eps_out; unsigned seed_lo, seed_hi; const int* step; int draw_id;
};

// One N(0,1) draw per (element, draw, step): Box-Muller on two Philox words.
__device__ __forceinline__ float philox_normal(unsigned elem, unsigned draw, unsigned step, unsigned k0, unsigned k1) {
    uint32_t w[4];
    philox4x32_10(elem, draw, step, 0x5AC0FFEEu, k0, k1, w);
    const float u1 = ((float)(w[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
    const float u2 = (float)(w[1] >> 8) * (1.0f / 16777216.0f);               // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__global__ __launch_bounds__(256) void tanh_gaussian_fwd_kernel(const TgFwdParams p) {
    // one wave per sample; lanes stride over the action dims
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= p.B) return;
    float lp = 0.0f;
    for (int j = lane; j < p.A; j += 64) {
        const float mean = p.feat[(long long)b * p.ld_feat + j];
        const float ls = p.feat[(long long)b * p.ld_feat + p.A + j];
        const float std = expf(fminf(fmaxf(ls, p.ls_min), p.ls_max));
        const float e = p.eps ? p.eps[(long long)b * p.A + j]
                              : philox_normal((unsigned)(b * p.A + j), (unsigned)p.draw_id, (unsigned)p.step[0], p.seed_lo, p.seed_hi);
        if (p.eps_out) p.eps_out[(long long)b * p.A + j] = e;
        const float u = mean + e * std;
        const float t = tanhf(u);
        const float s = p.scale[j];
        const float a = t * s + p.bias[j];
        const float diff = u - mean;
        lp += -(diff * diff) / (2.0f * (std * std)) - logf(std) - kHalfLog2Pi - logf(s * (1.0f - t * t) + p.epsilon);
        p.act0[(long long)b * p.ld0 + j] = a;
        if (p.act1) p.act1[(long long)b * p.ld1 + j] = a;
        if (p.saved) { p.saved[(long long)b * 2 * p.A + j] = t; p.saved[(long long)b * 2 * p.A + p.A + j] = std; }
    }
    for (int off = 32; off > 0; off >>= 1) lp += __shfl_xor(lp, off, 64);
    if (lane == 0) p.neg_logp[b] = -lp;
}

struct TgBwdParams {
    const float* feat; long long ld_feat; const float* eps; const float* saved; const float* scale;
    int B, A; float ls_min, ls_max, epsilon;
    const float* da0; const float* da1; long long ld_da;   // dL/d(action), sum of two sources (da1 may be NULL)
    const float* d_neglogp;                                // device scalar: dL/d(neg_logp_b), same for every b
    float* dfeat; long long ld_dfeat;                      // [B][2A]
};

__global__ __launch_bounds__(256) void tanh_gaussian_bwd_kernel(const TgBwdParams p) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)p.B * p.A) return;
    const int b = (int)(idx / p.A), j = (int)(idx - (long long)b * p.A);
    const float t = p.saved[(long long)b * 2 * p.A + j], std = p.saved[(long long)b * 2 * p.A + p.A + j];
    const float e = p.eps[(long long)b * p.A + j], s = p.scale[j];
    const float ls = p.feat[(long long)b * p.ld_feat + p.A + j];
    float ga = p.da0[(long long)b * p.ld_da + j];
    if (p.da1) ga += p.da1[(long long)b * p.ld_da + j];
    const float g_lp = -p.d_neglogp[0];                    // dL/d(log p)
    const float omt2 = 1.0f - t * t;
    // d(log p)/du through the squash term; the Gaussian term's direct u- and mean-dependence cancel
    const float sq = 2.0f * s * t * omt2 / (s * omt2 + p.epsilon);
    const float g_u = ga * s * omt2 + g_lp * sq;           // = dL/d(mean)
    const float g_std = g_u * e + g_lp * (-1.0f / std);
    const bool inside = ls >= p.ls_min && ls <= p.ls_max;  // clamp passes gradient inside the bounds
    p.dfeat[(long long)b * p.ld_dfeat + j] = g_u;
    p.dfeat[(long long)b * p.ld_dfeat + p.A + j] = inside ? g_std * std : 0.0f;
}

// ---- block-wide deterministic reductions (one workgroup of 1024 threads) -------------------------
__device__ __forceinline__ float block_sum(float v, float* s_buf) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.0f;
    if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += s_buf[w];
    return r;    // valid in thread 0
}
__device__ __forceinline__ float block_max(float v, float* s_buf) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = -INFINITY;
    if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, s_buf[w]);
    return r;
}

struct CriticLossParams {
    const float* q_next; long long ld_qn;   // [B][H]
    const float* neg_logp_next;             // [B]
    const float* rewards; const unsigned char* dones;
    int rd_div;                             // row b reads rewards / dones entry b / rd_div
    const float* log_alpha;                 // device scalar
    float gamma, reward_scale; int ignore_dones, group;   // group = num_aug (DrQ target mean), 1 for SAC
    const float* q; long long ld_q;         // [B][H]
    int B, H;
    float* q_target;                        // [B]
    float* dq; long long ld_dq;             // [B][H] = d(loss)/dq
    float* stats;                           // [4]: critic_loss, max |q - y|, mean_b min_h q, mean q_target
};

__global__ __launch_bounds__(1024) void critic_loss_kernel(const CriticLossParams p) {
    __shared__ float s_buf[16];
    const float alpha = expf(p.log_alpha[0]);
    float loss = 0.0f, mx = 0.0f, sq = 0.0f, sy = 0.0f;
    for (int g0 = threadIdx.x; g0 * p.group < p.B; g0 += blockDim.x) {
        float ymean = 0.0f;
        for (int a = 0; a < p.group; ++a) {
            const int b = g0 * p.group + a;
            float mn = p.q_next[(long long)b * p.ld_qn];
            for (int hh = 1; hh < p.H; ++hh) mn = fminf(mn, p.q_next[(long long)b * p.ld_qn + hh]);
            mn = mn + alpha * p.neg_logp_next[b];
            const int e = b / p.rd_div;
            const float r = p.rewards[e] * p.reward_scale;
            const float y = p.ignore_dones ? r + p.gamma * mn : r + (1.0f - (p.dones[e] ? 1.0f : 0.0f)) * p.gamma * mn;
            ymean += y;
        }
        ymean = p.group > 1 ? ymean / (float)p.group : ymean;
        for (int a = 0; a < p.group; ++a) {
            const int b = g0 * p.group + a;
            p.q_target[b] = ymean;
            float qmin = INFINITY;
            for (int hh = 0; hh < p.H; ++hh) {
                const float qv = p.q[(long long)b * p.ld_q + hh], d = qv - ymean;
                loss += d * d;
                mx = fmaxf(mx, fabsf(d));
                qmin = fminf(qmin, qv);
                p.dq[(long long)b * p.ld_dq + hh] = 2.0f * d / (float)p.B;   // d/dq of mean_{b,h}(d^2) * H
            }
            sq += qmin; sy += ymean;
        }
    }
    const float t_loss = block_sum(loss, s_buf), t_q = block_sum(sq, s_buf), t_y = block_sum(sy, s_buf);
    const float t_mx = block_max(mx, s_buf);
    if (threadIdx.x == 0) {
        p.stats[0] = t_loss / (float)p.B;
        p.stats[1] = t_mx;
        p.stats[2] = t_q / (float)p.B;
        p.stats[3] = t_y / (float)p.B;
    }
}

struct ActorLossParams {
    const float* q_pi; long long ld_q;      // [B][H]
    const float* neg_logp;                  // [B]
    const float* log_alpha; float target_entropy;
    int B, H;
    float* dq; long long ld_dq;             // [B][H]
    float* d_neglogp;                       // device scalar = -alpha / B
    float* alpha_grad;                      // device scalar = d(alpha_loss)/d(log_alpha)
    float* stats;                           // [3]: actor_loss, entropy, alpha_loss
};

__global__ __launch_bounds__(1024) void actor_loss_kernel(const ActorLossParams p) {
    __shared__ float s_buf[16];
    const float alpha = expf(p.log_alpha[0]);
    float sq = 0.0f, se = 0.0f;
    for (int b = threadIdx.x; b < p.B; b += blockDim.x) {
        int arg = 0;
        float mn = p.q_pi[(long long)b * p.ld_q];
        for (int hh = 1; hh < p.H; ++hh) {
            const float v = p.q_pi[(long long)b * p.ld_q + hh];
            if (v < mn) { mn = v; arg = hh; }          // torch.min: first index among equal minima
        }
        for (int hh = 0; hh < p.H; ++hh) p.dq[(long long)b * p.ld_dq + hh] = hh == arg ? -1.0f / (float)p.B : 0.0f;
        sq += mn; se += p.neg_logp[b];
    }
    const float t_q = block_sum(sq, s_buf), t_e = block_sum(se, s_buf);
    if (threadIdx.x == 0) {
        const float entropy = t_e / (float)p.B;
        p.stats[0] = -(t_q / (float)p.B + alpha * entropy);
        p.stats[1] = entropy;
        const float al = alpha * (entropy - p.target_entropy);
        p.stats[2] = al;
        p.alpha_grad[0] = al;                          // d/d(log_alpha) of exp(log_alpha) * c  =  exp(log_alpha) * c
        p.d_neglogp[0] = -alpha / (float)p.B;
    }
}

// The scalars a step reports (losses, gradient norms, alpha = exp(log_alpha)) gathered into one array
// with one launch, instead of one tiny copy / exp / stack launch each.  The same launch can finish up to
// four deferred optimizer passes (pcrl_adam_step_f32 with defer_finalize): gradient norm = sqrt of the fixed-order
// sum of the pass's per-block partial sums, step count += 1 -- before the scalars are copied, so a norm can be one
// of them.
__global__ __launch_bounds__(256) void gather_scalars_kernel(const ScalarListParams p) {
    __shared__ float s_part[kMaxFinalize * 4];
    gather_scalars_block(p, (int)threadIdx.x, s_part);
}

}  // namespace pcrl

using namespace pcrl;

extern "C" int pcrl_gather_scalars_f32(const float* const* src, float* const* dst, const int32_t* take_exp, int32_t n,
                                       const pcrl_adam_pending* pending, int32_t n_pending, void* stream) {
    return pcrl_gather_scalars_host_f32(src, dst, take_exp, n, pending, n_pending, nullptr, stream);
}

extern "C" int pcrl_gather_scalars_host_f32(const float* const* src, float* const* dst, const int32_t* take_exp, int32_t n,
                                            const pcrl_adam_pending* pending, int32_t n_pending, float* host_out, void* stream) {
    if (n == 0 && n_pending == 0) return PCRL_OK;
    ScalarListParams p;
    if (int rc = scalar_list_fill(src, dst, take_exp, n, pending, n_pending, host_out, p)) return rc;
    hipLaunchKernelGGL(gather_scalars_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("gather_scalars_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_tanh_gaussian_fwd_f32(const float* feat, int64_t ld_feat, const float* eps, const float* scale, const float* bias,
                                          int32_t B, int32_t A, float log_std_min, float log_std_max, float epsilon,
                                          float* action, int64_t ld_action, float* action2, int64_t ld_action2,
                                          float* neg_logp, float* saved, void* stream) {
    if (!feat || !eps || !scale || !bias || !action || !neg_logp) return fail(PCRL_E_ARG, "NULL argument");
    if (B == 0) return PCRL_OK;
    TgFwdParams p{feat, ld_feat, eps, scale, bias, B, A, log_std_min, log_std_max, epsilon, action, ld_action, action2, ld_action2, neg_logp, saved,
                  nullptr, 0u, 0u, nullptr, 0};
    hipLaunchKernelGGL(tanh_gaussian_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("tanh_gaussian_fwd_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_tanh_gaussian_sample_fwd_f32(const float* feat, int64_t ld_feat, uint64_t seed, const int32_t* step_counter, int32_t draw_id,
                                                 float* eps_out, const float* scale, const float* bias,
                                                 int32_t B, int32_t A, float log_std_min, float log_std_max, float epsilon,
                                                 float* action, int64_t ld_action, float* action2, int64_t ld_action2,
                                                 float* neg_logp, float* saved, void* stream) {
    if (!feat || !step_counter || !eps_out || !scale || !bias || !action || !neg_logp) return fail(PCRL_E_ARG, "NULL argument");
    if (B == 0) return PCRL_OK;
    TgFwdParams p{feat, ld_feat, nullptr, scale, bias, B, A, log_std_min, log_std_max, epsilon, action, ld_action, action2, ld_action2, neg_logp, saved,
                  eps_out, (unsigned)seed, (unsigned)(seed >> 32), step_counter, draw_id};
    hipLaunchKernelGGL(tanh_gaussian_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("tanh_gaussian_fwd_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_tanh_gaussian_bwd_f32(const float* feat, int64_t ld_feat, const float* eps, const float* saved, const float* scale,
                                          int32_t B, int32_t A, float log_std_min, float log_std_max, float epsilon,
                                          const float* d_action0, const float* d_action1, int64_t ld_d_action, const float* d_neglogp,
                                          float* d_feat, int64_t ld_d_feat, void* stream) {
    if (!feat || !eps || !saved || !scale || !d_action0 || !d_neglogp || !d_feat) return fail(PCRL_E_ARG, "NULL argument");
    if (B == 0) return PCRL_OK;
    TgBwdParams p{feat, ld_feat, eps, saved, scale, B, A, log_std_min, log_std_max, epsilon, d_action0, d_action1, ld_d_action, d_neglogp, d_feat, ld_d_feat};
    const long long n = (long long)B * A;
    hipLaunchKernelGGL(tanh_gaussian_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("tanh_gaussian_bwd_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_sac_critic_loss_f32(const float* q_next, int64_t ld_q_next, const float* neg_logp_next, const float* rewards,
                                        const uint8_t* dones, int32_t rd_row_div, const float* log_alpha, float gamma, float reward_scale,
                                        int32_t ignore_dones, int32_t group, const float* q, int64_t ld_q, int32_t B, int32_t H,
                                        float* q_target, float* dq, int64_t ld_dq, float* stats, void* stream) {
    if (!q_next || !neg_logp_next || !rewards || !log_alpha || !q || !q_target || !dq || !stats) return fail(PCRL_E_ARG, "NULL argument");
    if (!ignore_dones && !dones) return fail(PCRL_E_ARG, "dones is NULL");
    if (B < 1 || H < 1 || group < 1 || B % group) return fail(PCRL_E_ARG, "bad shape B=%d H=%d group=%d", B, H, group);
    CriticLossParams p{q_next, ld_q_next, neg_logp_next, rewards, dones, rd_row_div > 1 ? rd_row_div : 1, log_alpha, gamma, reward_scale, ignore_dones, group, q, ld_q, B, H, q_target, dq, ld_dq, stats};
    hipLaunchKernelGGL(critic_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("critic_loss_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_sac_actor_loss_f32(const float* q_pi, int64_t ld_q, const float* neg_logp, const float* log_alpha, float target_entropy,
                                       int32_t B, int32_t H, float* dq, int64_t ld_dq, float* d_neglogp, float* alpha_grad, float* stats,
                                       void* stream) {
    if (!q_pi || !neg_logp || !log_alpha || !dq || !d_neglogp || !alpha_grad || !stats) return fail(PCRL_E_ARG, "NULL argument");
    if (B < 1 || H < 1) return fail(PCRL_E_ARG, "bad shape");
    ActorLossParams p{q_pi, ld_q, neg_logp, log_alpha, target_entropy, B, H, dq, ld_dq, d_neglogp, alpha_grad, stats};
    hipLaunchKernelGGL(actor_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("actor_loss_kernel");
    return PCRL_OK;
}
