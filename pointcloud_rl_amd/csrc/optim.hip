// Fused optimizer pass over a flat parameter buffer for gfx950: Adam + (optional) Polyak target
// update + gradient 2-norm in one sweep.
//
// Replaces, per call,
//   torch.optim.Adam.step over one parameter group PER TENSOR (reference build_optimizer,
//   pyrl/utils/torch/optimizer_utils.py:43-57: 24 groups for the critic, 6 for the actor),
//   soft_update's per-parameter copy_ (pyrl/utils/torch/ops.py:59-90) and
//   ExtendedModuleBase.grad_norm's per-tensor norms (pyrl/utils/torch/module_utils.py:40-45).
// HBM-bound: 16 B read + 12 B written per parameter (+ 8 B per Polyak-tracked parameter).
// The step count lives in device memory so the launch can be replayed from a hipGraph.
#include "common.h"

namespace pcrl {

struct AdamParams {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    long long n;
    float lr, beta1, beta2, eps, grad_scale;
    const int* step;          // device: number of steps BEFORE this one
    float* target;            // Polyak: target[i - t_begin] for t_begin <= i < t_end (may be NULL)
    long long t_begin, t_end;
    float tau;
    float* partial;           // [gridDim.x] sum of (scaled) grad^2 per block (NULL: the norm was taken by gradnorm_kernel before this pass)
    int main_blocks;          // blocks that sweep the buffer above; one more block (if any) does the rider
    int step_add;             // 1: *step counts the steps BEFORE this one; 0: it has already been advanced (a pass published first)
};

// A second, tiny optimizer riding on the launch (SAC's temperature next to the actor: one float, its own betas, moments and
// step count -- sac.py:192-195 steps them back to back): handled by ONE extra block, element by element.
struct AdamRider {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq; long long n;
    float lr, beta1, beta2, eps, grad_scale;
    const int* step; float* partial;          // partial[0] = sum of (scaled) grad^2
};

__device__ __forceinline__ void adam_elem(const AdamParams& p, float g_raw, float& m, float& v, float& w,
                                          float step_size, float bc2_sqrt, float& gsq) {
    const float g = g_raw * p.grad_scale;
    gsq = __builtin_fmaf(g, g, gsq);
    m = p.beta1 * m + (1.0f - p.beta1) * g;
    v = p.beta2 * v + (1.0f - p.beta2) * g * g;
    w = w - step_size * (m / (__builtin_sqrtf(v) / bc2_sqrt + p.eps));
}

__device__ __forceinline__ void adam_rider_block(const AdamRider& r) {
    const float step = (float)(r.step[0] + 1);
    const float bc1 = 1.0f - powf(r.beta1, step);
    const float bc2_sqrt = __builtin_sqrtf(1.0f - powf(r.beta2, step));
    const float step_size = r.lr / bc1;
    float gsq = 0.0f;
    for (long long j = threadIdx.x; j < r.n; j += 256) {
        const float g = r.grad[j] * r.grad_scale;
        gsq = __builtin_fmaf(g, g, gsq);
        const float m = r.beta1 * r.exp_avg[j] + (1.0f - r.beta1) * g;
        const float v = r.beta2 * r.exp_avg_sq[j] + (1.0f - r.beta2) * g * g;
        r.exp_avg[j] = m; r.exp_avg_sq[j] = v;
        r.param[j] = r.param[j] - step_size * (m / (__builtin_sqrtf(v) / bc2_sqrt + r.eps));
    }
    for (int off = 32; off > 0; off >>= 1) gsq += __shfl_down(gsq, off, 64);
    __shared__ float s_rider[4];
    if ((threadIdx.x & 63) == 0) s_rider[threadIdx.x >> 6] = gsq;
    __syncthreads();
    if (threadIdx.x == 0) r.partial[0] = (s_rider[0] + s_rider[1]) + (s_rider[2] + s_rider[3]);
}

__device__ __forceinline__ void adam_main_block(const AdamParams& p, const int block) {
    const float step = (float)(p.step[0] + p.step_add);
    const float bc1 = 1.0f - powf(p.beta1, step);
    const float bc2_sqrt = __builtin_sqrtf(1.0f - powf(p.beta2, step));
    const float step_size = p.lr / bc1;
    float gsq = 0.0f;
    const long long stride = (long long)p.main_blocks * blockDim.x * 4;
    for (long long i = ((long long)block * blockDim.x + threadIdx.x) * 4; i < p.n; i += stride) {
        if (i + 3 < p.n) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.grad + i);
            f32x4 m4 = *reinterpret_cast<const f32x4*>(p.exp_avg + i);
            f32x4 v4 = *reinterpret_cast<const f32x4*>(p.exp_avg_sq + i);
            f32x4 w4 = *reinterpret_cast<const f32x4*>(p.param + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float m = m4[k], v = v4[k], w = w4[k];
                adam_elem(p, g4[k], m, v, w, step_size, bc2_sqrt, gsq);
                m4[k] = m; v4[k] = v; w4[k] = w;
            }
            *reinterpret_cast<f32x4*>(p.exp_avg + i) = m4;
            *reinterpret_cast<f32x4*>(p.exp_avg_sq + i) = v4;
            *reinterpret_cast<f32x4*>(p.param + i) = w4;
            if (p.target && i + 3 >= p.t_begin && i < p.t_end) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i + k >= p.t_begin && i + k < p.t_end) {
                        float* t = p.target + (i + k - p.t_begin);
                        *t = *t * (1.0f - p.tau) + w4[k] * p.tau;
                    }
            }
        } else {
            for (long long j = i; j < p.n; ++j) {
                float m = p.exp_avg[j], v = p.exp_avg_sq[j], w = p.param[j];
                adam_elem(p, p.grad[j], m, v, w, step_size, bc2_sqrt, gsq);
                p.exp_avg[j] = m; p.exp_avg_sq[j] = v; p.param[j] = w;
                if (p.target && j >= p.t_begin && j < p.t_end) {
                    float* t = p.target + (j - p.t_begin);
                    *t = *t * (1.0f - p.tau) + w * p.tau;
                }
            }
        }
    }
    if (!p.partial) return;    // (uniform over the launch)
    // block reduction of grad^2 in a fixed order (wave shuffle tree, then the 4 waves in order)
    for (int off = 32; off > 0; off >>= 1) gsq += __shfl_down(gsq, off, 64);
    __shared__ float s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = gsq;
    __syncthreads();
    if (threadIdx.x == 0) p.partial[block] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamParams p, const AdamRider rider) {
    if ((int)blockIdx.x >= p.main_blocks) { adam_rider_block(rider); return; }
    adam_main_block(p, (int)blockIdx.x);
}

// A published pass (its norm and step count went ahead of it) whose FIRST workgroup gathers and publishes the step's metrics
// (gather_scalars_block, csrc/common.h) instead of a launch of its own in front: the metrics leave when this launch starts.  Nothing that
// workgroup writes -- norms, the step counts of OTHER (deferred) passes, the gathered scalars, the pinned host mirror -- is read by the
// pass's own workgroups.
__global__ __launch_bounds__(256) void adam_gather_kernel(const AdamParams p, const ScalarListParams g) {
    if (blockIdx.x == 0) {
        __shared__ float s_fin[kMaxFinalize * 4];
        gather_scalars_block(g, (int)threadIdx.x, s_fin);
        return;
    }
    adam_main_block(p, (int)blockIdx.x - 1);
}

// The gradient norm of a pass WITHOUT the pass: the same grid, the same per-thread element order and the same reduction tree as
// adam_kernel, so the partial sums (and the norm the end-of-step launch forms from them) are bit for bit what the optimizer pass would
// have produced.  Lets a step publish its metrics BEFORE its last optimizer launch (pcrl_grad_norm_partials_f32): 4 B per parameter read.
struct GradNormParams { const float* grad; long long n; float grad_scale; float* partial; int main_blocks; int* step; };   // step: advanced by ONE thread of this launch (nothing in it reads it)
__global__ __launch_bounds__(256) void gradnorm_kernel(const GradNormParams p, const AdamRider rider) {
    if ((int)blockIdx.x >= p.main_blocks) { adam_rider_block(rider); return; }
    float gsq = 0.0f;
    const long long stride = (long long)p.main_blocks * blockDim.x * 4;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < p.n; i += stride) {
        if (i + 3 < p.n) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.grad + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float g = g4[k] * p.grad_scale; gsq = __builtin_fmaf(g, g, gsq); }
        } else {
            for (long long j = i; j < p.n; ++j) { const float g = p.grad[j] * p.grad_scale; gsq = __builtin_fmaf(g, g, gsq); }
        }
    }
    for (int off = 32; off > 0; off >>= 1) gsq += __shfl_down(gsq, off, 64);
    __shared__ float s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = gsq;
    __syncthreads();
    if (threadIdx.x == 0) {
        p.partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        if (blockIdx.x == 0) p.step[0] += 1;
    }
}

__global__ __launch_bounds__(256) void gradnorm_finalize_kernel(const float* partial, int n, float* out, int* step) {
    // runs after adam_kernel on the same stream: the only writer of the step counter
    if (threadIdx.x == 0) step[0] += 1;
    if (!out) return;
    // one block; thread t sums partial[t], partial[t + 256], ... then a fixed-order tree
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ float s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = __builtin_sqrtf((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
}

// theta' <- (1 - tau) theta' + tau theta without an optimizer step (hard_update with tau = 1).
__global__ void polyak_kernel(float* target, const float* src, long long n, float tau) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) target[i] = target[i] * (1.0f - tau) + src[i] * tau;
}

static int adam_grid(long long n) {
    long long blocks = (n + 1023) / 1024;
    const long long cap = (long long)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace pcrl

using namespace pcrl;

extern "C" int pcrl_adam_workspace_bytes(size_t n, size_t* bytes) {
    if (!bytes) return fail(PCRL_E_ARG, "bytes is NULL");
    *bytes = sizeof(float) * (size_t)adam_grid((long long)n);
    return PCRL_OK;
}

static int adam_launch(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                       float lr, float beta1, float beta2, float eps, float grad_scale,
                       int32_t* step_counter, float* grad_norm_out,
                       float* target, size_t target_begin, size_t target_end, float tau,
                       void* workspace, size_t workspace_bytes, pcrl_adam_pending* defer_finalize,
                       const pcrl_adam_rider* rider, pcrl_adam_pending* rider_defer, void* stream, bool published = false,
                       const ScalarListParams* gather = nullptr) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_counter) return fail(PCRL_E_ARG, "NULL argument");
    if (n == 0) return PCRL_OK;
    if (published) {
        // the pass of a step whose norm and step count pcrl_grad_norm_partials_f32 + the gather launch have already taken care of
        if (target && !(target_begin <= target_end && target_end <= n)) return fail(PCRL_E_ARG, "bad Polyak range");
        const int grid = adam_grid((long long)n);
        AdamParams p{param, grad, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps, grad_scale, step_counter,
                     target, (long long)target_begin, (long long)target_end, tau, nullptr, grid, 0};
        if (gather) {
            hipLaunchKernelGGL(adam_gather_kernel, dim3(grid + 1), dim3(256), 0, (hipStream_t)stream, p, *gather);
            PCRL_CHECK_LAUNCH("adam_gather_kernel");
            return PCRL_OK;
        }
        hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, AdamRider{});
        PCRL_CHECK_LAUNCH("adam_kernel");
        return PCRL_OK;
    }
    if (target && !(target_begin <= target_end && target_end <= n)) return fail(PCRL_E_ARG, "bad Polyak range");
    const int grid = adam_grid((long long)n);
    if (!workspace || workspace_bytes < sizeof(float) * (size_t)grid) return fail(PCRL_E_WORKSPACE, "workspace too small");
    AdamRider r{};
    if (rider) {
        if (!rider->param || !rider->grad || !rider->exp_avg || !rider->exp_avg_sq || !rider->step_counter || !rider->partial || rider->n < 1 || rider->n > 4096)
            return fail(PCRL_E_ARG, "adam rider: NULL argument or n outside [1, 4096]");
        r = AdamRider{rider->param, rider->grad, rider->exp_avg, rider->exp_avg_sq, (long long)rider->n, rider->lr, rider->beta1, rider->beta2,
                      rider->eps, rider->grad_scale, rider->step_counter, rider->partial};
    }
    hipStream_t st = (hipStream_t)stream;
    AdamParams p{param, grad, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps, grad_scale, step_counter,
                 target, (long long)target_begin, (long long)target_end, tau, static_cast<float*>(workspace), grid, 1};
    hipLaunchKernelGGL(adam_kernel, dim3(grid + (rider ? 1 : 0)), dim3(256), 0, st, p, r);
    PCRL_CHECK_LAUNCH("adam_kernel");
    if (rider) {
        if (rider_defer) {
            rider_defer->partial = rider->partial; rider_defer->n_partial = 1;
            rider_defer->grad_norm_out = rider->grad_norm_out; rider_defer->step_counter = rider->step_counter;
        } else {
            hipLaunchKernelGGL(gradnorm_finalize_kernel, dim3(1), dim3(256), 0, st, rider->partial, 1, rider->grad_norm_out, rider->step_counter);
            PCRL_CHECK_LAUNCH("gradnorm_finalize_kernel");
        }
    }
    if (defer_finalize) {       // the caller's end-of-step pcrl_gather_scalars_f32 launch sums the partials and advances the step count
        defer_finalize->partial = p.partial; defer_finalize->n_partial = grid;
        defer_finalize->grad_norm_out = grad_norm_out; defer_finalize->step_counter = step_counter;
        return PCRL_OK;
    }
    hipLaunchKernelGGL(gradnorm_finalize_kernel, dim3(1), dim3(256), 0, st, p.partial, grid, grad_norm_out, step_counter);
    PCRL_CHECK_LAUNCH("gradnorm_finalize_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                  float lr, float beta1, float beta2, float eps, float grad_scale,
                                  int32_t* step_counter, float* grad_norm_out,
                                  float* target, size_t target_begin, size_t target_end, float tau,
                                  void* workspace, size_t workspace_bytes, pcrl_adam_pending* defer_finalize, void* stream) {
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, grad_scale, step_counter, grad_norm_out, target, target_begin,
                       target_end, tau, workspace, workspace_bytes, defer_finalize, nullptr, nullptr, stream);
}

extern "C" int pcrl_adam_step_rider_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                        float lr, float beta1, float beta2, float eps, float grad_scale,
                                        int32_t* step_counter, float* grad_norm_out,
                                        float* target, size_t target_begin, size_t target_end, float tau,
                                        void* workspace, size_t workspace_bytes, pcrl_adam_pending* defer_finalize,
                                        const pcrl_adam_rider* rider, pcrl_adam_pending* rider_defer, void* stream) {
    if (!rider) return fail(PCRL_E_ARG, "rider is NULL");
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, grad_scale, step_counter, grad_norm_out, target, target_begin,
                       target_end, tau, workspace, workspace_bytes, defer_finalize, rider, rider_defer, stream);
}

extern "C" int pcrl_grad_norm_partials_f32(const float* grad, size_t n, float grad_scale, int32_t* step_counter, float* grad_norm_out,
                                           void* workspace, size_t workspace_bytes, pcrl_adam_pending* pending,
                                           const pcrl_adam_rider* rider, pcrl_adam_pending* rider_pending, void* stream) {
    if (!grad || !step_counter || !pending) return fail(PCRL_E_ARG, "NULL argument");
    if (n == 0) return fail(PCRL_E_ARG, "pcrl_grad_norm_partials_f32: empty buffer");
    const int grid = adam_grid((long long)n);
    if (!workspace || workspace_bytes < sizeof(float) * (size_t)grid) return fail(PCRL_E_WORKSPACE, "workspace too small");
    AdamRider r{};
    if (rider) {
        if (!rider->param || !rider->grad || !rider->exp_avg || !rider->exp_avg_sq || !rider->step_counter || !rider->partial || rider->n < 1 || rider->n > 4096 || !rider_pending)
            return fail(PCRL_E_ARG, "adam rider: NULL argument or n outside [1, 4096]");
        r = AdamRider{rider->param, rider->grad, rider->exp_avg, rider->exp_avg_sq, (long long)rider->n, rider->lr, rider->beta1, rider->beta2,
                      rider->eps, rider->grad_scale, rider->step_counter, rider->partial};
    }
    GradNormParams p{grad, (long long)n, grad_scale, static_cast<float*>(workspace), grid, step_counter};
    hipLaunchKernelGGL(gradnorm_kernel, dim3(grid + (rider ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, p, r);
    PCRL_CHECK_LAUNCH("gradnorm_kernel");
    // (the launch has advanced this pass's step count itself: the gather launch only forms the norm)
    pending->partial = p.partial; pending->n_partial = grid; pending->grad_norm_out = grad_norm_out; pending->step_counter = nullptr;
    if (rider) {
        rider_pending->partial = rider->partial; rider_pending->n_partial = 1;
        rider_pending->grad_norm_out = rider->grad_norm_out; rider_pending->step_counter = rider->step_counter;
    }
    return PCRL_OK;
}

extern "C" int pcrl_adam_step_published_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                            float lr, float beta1, float beta2, float eps, float grad_scale, const int32_t* step_counter,
                                            float* target, size_t target_begin, size_t target_end, float tau, void* stream) {
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, grad_scale, const_cast<int32_t*>(step_counter), nullptr, target,
                       target_begin, target_end, tau, nullptr, 0, nullptr, nullptr, nullptr, stream, true);
}

extern "C" int pcrl_adam_step_published_gather_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                                   float lr, float beta1, float beta2, float eps, float grad_scale, const int32_t* step_counter,
                                                   float* target, size_t target_begin, size_t target_end, float tau,
                                                   const float* const* src, float* const* dst, const int32_t* take_exp, int32_t n_scalars,
                                                   const pcrl_adam_pending* pending, int32_t n_pending, float* host_out, void* stream) {
    ScalarListParams g;
    if (int rc = scalar_list_fill(src, dst, take_exp, n_scalars, pending, n_pending, host_out, g)) return rc;
    for (int f = 0; f < n_pending; ++f)
        if (pending[f].step_counter == step_counter) return fail(PCRL_E_ARG, "the published pass's own step count must have been advanced before it (pcrl_grad_norm_partials_f32)");
    if (n == 0) return fail(PCRL_E_ARG, "empty pass");
    return adam_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, grad_scale, const_cast<int32_t*>(step_counter), nullptr, target,
                       target_begin, target_end, tau, nullptr, 0, nullptr, nullptr, nullptr, stream, true, &g);
}

extern "C" int pcrl_polyak_f32(float* target, const float* src, size_t n, float tau, void* stream) {
    if (!target || !src) return fail(PCRL_E_ARG, "NULL argument");
    if (n == 0) return PCRL_OK;
    hipLaunchKernelGGL(polyak_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, target, src, (long long)n, tau);
    PCRL_CHECK_LAUNCH("polyak_kernel");
    return PCRL_OK;
}
