// Dense actor / critic heads for gfx950 (MI355X), fp32 on v_mfma_f32_32x32x2_f32.
//
// Replaces the reference's LinearMLP heads (pyrl/networks/backbones/mlp.py:97-100: Linear + ReLU
// stacks, actor D->1024->1024->2A, each Q head D+A->1024->1024->1) and their autograd backward,
// plus PointNet.final_mlp's Linear (pointnet.py:110).  One batched GEMM kernel with generic operand
// strides serves the three shapes of a Linear layer:
//     forward        Y  = act(X W^T + b)        A = X [M,K]      B[k][n] = W[n][k]
//     data gradient  dX = (dY W) (.) relu-mask  A = dY [M,N']    B[k][n] = W[k][n]
//     weight grad    dW = dY^T X, db = dY^T 1   A[m][k] = dY[k][m]  B = X (+ a virtual ones column)
// The batch is small (M = 256 rows) so a 32x32 output tile is owned by one 4-wave workgroup that
// splits K four ways (every CU gets work at N = 1024) and reduces in LDS in a fixed order.
// Operands are read straight from L2 in MFMA operand order: a k-contiguous operand as one 16-byte
// load per lane per 4 k-steps, an m/n-contiguous one as coalesced 4-byte loads.
#include "common.h"

namespace pcrl {

struct GemmParams {
    const float* A; const float* B; float* C;
    const float* bias;       // [N] added to every row (may be NULL)
    const float* mask;       // [M][ld_mask]: C = acc * (mask > 0) (ReLU backward; may be NULL)
    int M, N, K;
    long long a_sm, a_sk, b_sk, b_sn, ldc, ld_mask;
    long long a_bs, b_bs, c_bs, bias_bs, mask_bs;    // batch strides (elements)
    int relu;                // C = max(acc + bias, 0)
    int ones_col;            // B[k][ones_col] == 1 for every k (bias gradient); -1: none
    int accumulate;          // C += result
    float* C_ones;           // optional separate destination of column `ones_col`
    long long c_ones_bs;
};

constexpr int kGemmWaves = 8;

template <bool A_K4, bool B_K4>
__device__ __forceinline__ void gemm_load_chunk(const GemmParams& p, const float* a_row, const float* b_col, int k,
                                                bool m_ok, bool n_ok, bool n_ones, f32x4& a, f32x4& b) {
    a = f32x4{0.f, 0.f, 0.f, 0.f};
    b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (A_K4 && k + 3 < p.K) {
        if (m_ok) a = *reinterpret_cast<const f32x4*>(a_row + k);
    } else if (m_ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < p.K) a[j] = a_row[(long long)(k + j) * p.a_sk];
    }
    if (n_ones) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = k + j < p.K ? 1.0f : 0.0f;
    } else if (B_K4 && k + 3 < p.K) {
        if (n_ok) b = *reinterpret_cast<const f32x4*>(b_col + k);
    } else if (n_ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < p.K) b[j] = b_col[(long long)(k + j) * p.b_sk];
    }
}

template <bool A_K4, bool B_K4>
__global__ __launch_bounds__(64 * kGemmWaves) void gemm_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float s_red[kGemmWaves][16][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32, bz = blockIdx.z;
    const float* A = p.A + bz * p.a_bs;
    const float* B = p.B + bz * p.b_bs;
    const int m = m0 + i, n = n0 + i;
    const bool m_ok = m < p.M, n_ok = n < p.N, n_ones = n == p.ones_col;
    const int n_chunks = (p.K + 7) / 8;
    const int per_wave = (n_chunks + kGemmWaves - 1) / kGemmWaves;
    const int q_begin = wave * per_wave, q_end = min(q_begin + per_wave, n_chunks);
    const float* a_row = A + (long long)m * p.a_sm;
    const float* b_col = B + (long long)n * p.b_sn;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // two chunks (8 MFMAs) in flight per iteration, next pair's loads issued before this pair's MFMAs
    f32x4 a0, b0, a1, b1;
    int q = q_begin;
    if (q < q_end) gemm_load_chunk<A_K4, B_K4>(p, a_row, b_col, 8 * q + 4 * h, m_ok, n_ok, n_ones, a0, b0);
    if (q + 1 < q_end) gemm_load_chunk<A_K4, B_K4>(p, a_row, b_col, 8 * (q + 1) + 4 * h, m_ok, n_ok, n_ones, a1, b1);
    for (; q < q_end; q += 2) {
        const f32x4 ca0 = a0, cb0 = b0, ca1 = a1, cb1 = b1;
        const bool second = q + 1 < q_end;
        if (q + 2 < q_end) gemm_load_chunk<A_K4, B_K4>(p, a_row, b_col, 8 * (q + 2) + 4 * h, m_ok, n_ok, n_ones, a0, b0);
        if (q + 3 < q_end) gemm_load_chunk<A_K4, B_K4>(p, a_row, b_col, 8 * (q + 3) + 4 * h, m_ok, n_ok, n_ones, a1, b1);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0[j], cb0[j], acc, 0, 0, 0);
        if (second) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1[j], cb1[j], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[r];
    __syncthreads();
    float* C = p.C + bz * p.c_bs;
    const float* bias = p.bias ? p.bias + bz * p.bias_bs : nullptr;
    const float* mask = p.mask ? p.mask + bz * p.mask_bs : nullptr;
#pragma unroll
    for (int e = 0; e < 1024 / (64 * kGemmWaves); ++e) {
        const int idx = tid + 64 * kGemmWaves * e, r = idx >> 6, ln = idx & 63;
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), col = n0 + (ln & 31);
        if (row < p.M && col < p.N) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < kGemmWaves; w += 2) v = v + (s_red[w][r][ln] + s_red[w + 1][r][ln]);
            if (bias) v = v + bias[col];
            if (p.relu) v = v > 0.0f ? v : 0.0f;
            if (mask) v = mask[(long long)row * p.ld_mask + col] > 0.0f ? v : 0.0f;
            float* dst = (p.C_ones && col == p.ones_col) ? p.C_ones + bz * p.c_ones_bs + row : C + (long long)row * p.ldc + col;
            *dst = p.accumulate ? *dst + v : v;
        }
    }
}

// Row-wise LayerNorm over a short feature vector (PointNet.final_mlp[1]: nn.LayerNorm(out), eps 1e-5,
// pointnet.py:110), one wave per row; the output may be scattered into several destination
// buffers (the concatenated inputs of the actor and Q heads).
struct LnParams {
    const float* x; long long ldx;       // [M][F]
    const float* gamma; const float* beta;
    int M, F; float eps;
    float* y[4]; long long ldy[4]; int n_dst;
    float* xhat; float* rstd;            // saved for backward (may be NULL)
};

__global__ __launch_bounds__(256) void layernorm_rows_fwd_kernel(const LnParams p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= p.M) return;
    const float* x = p.x + (long long)row * p.ldx;
    float v[4], s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int f = lane + 64 * j; v[j] = f < p.F ? x[f] : 0.0f; s += v[j]; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)p.F;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int f = lane + 64 * j; const float d = f < p.F ? v[j] - mean : 0.0f; q += d * d; }
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / __builtin_sqrtf(q / (float)p.F + p.eps);
    if (p.rstd && lane == 0) p.rstd[row] = rstd;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = lane + 64 * j;
        if (f < p.F) {
            const float xh = (v[j] - mean) * rstd;
            if (p.xhat) p.xhat[(long long)row * p.F + f] = xh;
            const float y = xh * p.gamma[f] + p.beta[f];
            for (int d = 0; d < p.n_dst; ++d) p.y[d][(long long)row * p.ldy[d] + f] = y;
        }
    }
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)), dxhat = dy * gamma, where dy is the
// sum of up to two upstream gradients (one per Q head); per-block partial dgamma / dbeta.
struct LnBwdParams {
    const float* dy0; const float* dy1; long long lddy;   // dy1 may be NULL
    const float* xhat; const float* rstd; const float* gamma;
    int M, F;
    float* dx; long long lddx;
    float* part;             // [gridDim.x][2][F] partial sums of dy*xhat and dy
};

__global__ __launch_bounds__(256) void layernorm_rows_bwd_kernel(const LnBwdParams p) {
    __shared__ float s_acc[4][2][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < p.M) {
        float dxh[4], xh[4], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = lane + 64 * j;
            dxh[j] = xh[j] = 0.0f;
            if (f < p.F) {
                float dy = p.dy0[(long long)row * p.lddy + f];
                if (p.dy1) dy += p.dy1[(long long)row * p.lddy + f];
                xh[j] = p.xhat[(long long)row * p.F + f];
                dg[j] = dy * xh[j]; db[j] = dy;
                dxh[j] = dy * p.gamma[f];
                s1 += dxh[j]; s2 += dxh[j] * xh[j];
            }
        }
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
        const float m1 = s1 / (float)p.F, m2 = s2 / (float)p.F, rstd = p.rstd[row];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = lane + 64 * j;
            if (f < p.F) p.dx[(long long)row * p.lddx + f] = rstd * ((dxh[j] - m1) - xh[j] * m2);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_acc[wave][0][lane + 64 * j] = dg[j]; s_acc[wave][1][lane + 64 * j] = db[j]; }
    __syncthreads();
    const int f = threadIdx.x;
    if (f < p.F) {
        p.part[((long long)blockIdx.x * 2 + 0) * p.F + f] = (s_acc[0][0][f] + s_acc[1][0][f]) + (s_acc[2][0][f] + s_acc[3][0][f]);
        p.part[((long long)blockIdx.x * 2 + 1) * p.F + f] = (s_acc[0][1][f] + s_acc[1][1][f]) + (s_acc[2][1][f] + s_acc[3][1][f]);
    }
}

__global__ void colsum_partials_kernel(const float* part, int nblk, int n, float* out0, float* out1, int half, int accumulate) {
    // out0[f] = sum_b part[b][0][f], out1[f] = sum_b part[b][1][f]  (fixed order), n = F
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    float a = 0.0f, c = 0.0f;
    for (int b = 0; b < nblk; ++b) { a += part[((long long)b * 2 + 0) * n + f]; c += part[((long long)b * 2 + 1) * n + f]; }
    out0[f] = accumulate ? out0[f] + a : a;
    out1[f] = accumulate ? out1[f] + c : c;
}

}  // namespace pcrl

using namespace pcrl;

extern "C" int pcrl_gemm_f32(const pcrl_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C) return fail(PCRL_E_ARG, "NULL argument");
    if (d->M < 0 || d->N < 0 || d->K < 0 || d->batch < 1) return fail(PCRL_E_ARG, "bad GEMM shape");
    if (d->M == 0 || d->N == 0) return PCRL_OK;
    GemmParams p{};
    p.A = d->A; p.B = d->B; p.C = d->C; p.bias = d->bias; p.mask = d->mask;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.a_sm = d->a_stride_m; p.a_sk = d->a_stride_k; p.b_sk = d->b_stride_k; p.b_sn = d->b_stride_n;
    p.ldc = d->ldc; p.ld_mask = d->ld_mask;
    p.a_bs = d->a_batch_stride; p.b_bs = d->b_batch_stride; p.c_bs = d->c_batch_stride;
    p.bias_bs = d->bias_batch_stride; p.mask_bs = d->mask_batch_stride;
    p.relu = d->relu; p.ones_col = d->ones_col; p.accumulate = d->accumulate;
    p.C_ones = d->C_ones; p.c_ones_bs = d->c_ones_batch_stride;
    auto aligned = [](const float* ptr, long long sm, long long bs) {
        return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && sm % 4 == 0 && bs % 4 == 0;
    };
    const bool a4 = p.a_sk == 1 && aligned(p.A, p.a_sm, p.a_bs);
    const bool b4 = p.b_sk == 1 && aligned(p.B, p.b_sn, p.b_bs);
    const dim3 grid((p.N + 31) / 32, (p.M + 31) / 32, d->batch);
    hipStream_t st = (hipStream_t)stream;
    if (a4 && b4) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(64 * kGemmWaves), 0, st, p);
    else if (a4) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(64 * kGemmWaves), 0, st, p);
    else if (b4) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(64 * kGemmWaves), 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(64 * kGemmWaves), 0, st, p);
    PCRL_CHECK_LAUNCH("gemm_f32_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_layernorm_rows_fwd_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, int32_t M, int32_t F,
                                           float eps, float* const* dst, const int64_t* ld_dst, int32_t n_dst,
                                           float* xhat, float* rstd, void* stream) {
    if (!x || !gamma || !beta || !dst || !ld_dst) return fail(PCRL_E_ARG, "NULL argument");
    if (F < 1 || F > 256 || n_dst < 1 || n_dst > 4) return fail(PCRL_E_ARG, "LayerNorm rows: 1 <= F <= 256, 1 <= n_dst <= 4");
    if (M == 0) return PCRL_OK;
    LnParams p{};
    p.x = x; p.ldx = ldx; p.gamma = gamma; p.beta = beta; p.M = M; p.F = F; p.eps = eps; p.n_dst = n_dst;
    for (int i = 0; i < n_dst; ++i) { p.y[i] = dst[i]; p.ldy[i] = ld_dst[i]; }
    p.xhat = xhat; p.rstd = rstd;
    hipLaunchKernelGGL(layernorm_rows_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("layernorm_rows_fwd_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_layernorm_rows_bwd_f32(const float* dy0, const float* dy1, int64_t lddy, const float* xhat, const float* rstd,
                                           const float* gamma, int32_t M, int32_t F, float* dx, int64_t lddx,
                                           float* dgamma, float* dbeta, int32_t accumulate,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy0 || !xhat || !rstd || !gamma || !dx || !dgamma || !dbeta) return fail(PCRL_E_ARG, "NULL argument");
    if (F < 1 || F > 256) return fail(PCRL_E_ARG, "LayerNorm rows: 1 <= F <= 256");
    if (M == 0) return PCRL_OK;
    const int nblk = (M + 3) / 4;
    const size_t need = sizeof(float) * (size_t)nblk * 2 * F;
    if (!workspace || workspace_bytes < need) return fail(PCRL_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
    LnBwdParams p{dy0, dy1, lddy, xhat, rstd, gamma, M, F, dx, lddx, static_cast<float*>(workspace)};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(layernorm_rows_bwd_kernel, dim3(nblk), dim3(256), 0, st, p);
    PCRL_CHECK_LAUNCH("layernorm_rows_bwd_kernel");
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((F + 255) / 256), dim3(256), 0, st, p.part, nblk, F, dgamma, dbeta, 0, accumulate);
    PCRL_CHECK_LAUNCH("colsum_partials_kernel");
    return PCRL_OK;
}
