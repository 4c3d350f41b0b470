// Stand-alone memory-shaped kernels of the hot path for gfx950 (HBM-bound; the fused encoder does
// these in registers, these entry points exist for callers that hold materialised tensors and for
// the HBM-roofline report of SURVEY.md section 8d):
//   segmax_fwd / segmax_bwd : `feature.max(-1)` over a materialised [B, c, N] tensor and its backward
//                             (reference pyrl/networks/backbones/pointnet.py:151), torch CPU tie/NaN rules
//   augment_xyz             : RandomJitterPoints / GlobalRotScaleTrans on a [B, 3, N] tensor
//                             (reference pyrl/utils/augmentations/pcd_aug.py:306-327, 84-123)
#include "encoder_common.h"

namespace pcrl {

// order-preserving map float -> uint32; every NaN maps to the top so that a NaN wins the max
__device__ __forceinline__ unsigned ord_key(float v) {
    const unsigned b = f2u(v);
    if (v != v) return 0xFFFFFFFFu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// one wave per row of N floats; 16-byte loads; (key, index) reduction with first-index tie-break
__global__ __launch_bounds__(256) void segmax_fwd_kernel(const float* __restrict__ x, long long rows, int N,
                                                          float* __restrict__ out, int* __restrict__ idx) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* r = x + row * N;
    unsigned best = 0u; int bi = 0x7FFFFFFF; float bv = 0.0f;
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(r) & 15) == 0);
    if (vec) {
        for (int n = 4 * lane; n < N; n += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(r + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned k = ord_key(v[j]);
                if (bi == 0x7FFFFFFF || k > best) { best = k; bi = n + j; bv = v[j]; }
            }
        }
    } else {
        for (int n = lane; n < N; n += 64) {
            const float v = r[n];
            const unsigned k = ord_key(v);
            if (bi == 0x7FFFFFFF || k > best) { best = k; bi = n; bv = v; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned ok = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        const float ov = __shfl_xor(bv, off, 64);
        if (oi != 0x7FFFFFFF && (bi == 0x7FFFFFFF || ok > best || (ok == best && oi < bi))) { best = ok; bi = oi; bv = ov; }
    }
    if (lane == 0) { out[row] = bv; idx[row] = bi; }
}

__global__ __launch_bounds__(256) void segmax_bwd_kernel(const float* __restrict__ g, const int* __restrict__ idx,
                                                          long long rows, int N, float* __restrict__ dx) {
    // dx[row][n] = g[row] if n == idx[row] else 0; one wave per row, 16-byte stores
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int hit = idx[row];
    const float gv = g[row];
    float* d = dx + row * N;
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(d) & 15) == 0);
    if (vec) {
        for (int n = 4 * lane; n < N; n += 256) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (hit >= n && hit < n + 4) v[hit - n] = gv;
            *reinterpret_cast<f32x4*>(d + n) = v;
        }
    } else {
        for (int n = lane; n < N; n += 64) d[n] = n == hit ? gv : 0.0f;
    }
}

struct AugParams {
    const float* in; float* out; int B, N;
    int flags, row_mul, row_add; float lo, hi;
    const float* noise; const float* affine; unsigned long long seed, offset; const unsigned long long* offset_ptr;
};

__global__ __launch_bounds__(256) void augment_xyz_kernel(const AugParams p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)p.B * p.N) return;
    const int b = (int)(i / p.N), n = (int)(i - (long long)b * p.N);
    const float* src = p.in + (long long)b * 3 * p.N + n;
    float x0 = src[0], x1 = src[p.N], x2 = src[2 * (long long)p.N];
    const long long row = (long long)b * p.row_mul + p.row_add;
    if (p.flags & PCRL_AUG_AFFINE) {
        const float* M = p.affine + row * 12;
        const float y0 = ((M[0] * x0 + M[1] * x1) + M[2] * x2) + M[3];
        const float y1 = ((M[4] * x0 + M[5] * x1) + M[6] * x2) + M[7];
        const float y2 = ((M[8] * x0 + M[9] * x1) + M[10] * x2) + M[11];
        x0 = y0; x1 = y1; x2 = y2;
    }
    if (p.flags & PCRL_AUG_JITTER) {
        if (p.noise) {
            x0 += p.noise[(row * 3 + 0) * p.N + n]; x1 += p.noise[(row * 3 + 1) * p.N + n]; x2 += p.noise[(row * 3 + 2) * p.N + n];
        } else {
            const unsigned long long e = (unsigned long long)row * p.N + n;
            const unsigned long long off = p.offset_ptr ? *p.offset_ptr : p.offset;
            uint32_t w[4];
            philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), (uint32_t)off, (uint32_t)(off >> 32), (uint32_t)p.seed, (uint32_t)(p.seed >> 32), w);
            x0 += u01_to_range(w[0], p.lo, p.hi); x1 += u01_to_range(w[1], p.lo, p.hi); x2 += u01_to_range(w[2], p.lo, p.hi);
        }
    }
    float* dst = p.out + (long long)b * 3 * p.N + n;
    dst[0] = x0; dst[p.N] = x1; dst[2 * (long long)p.N] = x2;
}

}  // namespace pcrl

using namespace pcrl;

extern "C" int pcrl_segmax_fwd_f32(const float* x, int64_t rows, int32_t N, float* out, int32_t* idx, void* stream) {
    if (!x || !out || !idx) return fail(PCRL_E_ARG, "NULL argument");
    if (rows < 0 || N < 1) return fail(PCRL_E_ARG, "bad shape rows=%lld N=%d", (long long)rows, N);
    if (rows == 0) return PCRL_OK;
    hipLaunchKernelGGL(segmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, N, out, idx);
    PCRL_CHECK_LAUNCH("segmax_fwd_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_segmax_bwd_f32(const float* grad_out, const int32_t* idx, int64_t rows, int32_t N, float* grad_x, void* stream) {
    if (!grad_out || !idx || !grad_x) return fail(PCRL_E_ARG, "NULL argument");
    if (rows < 0 || N < 1) return fail(PCRL_E_ARG, "bad shape");
    if (rows == 0) return PCRL_OK;
    hipLaunchKernelGGL(segmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, grad_out, idx, (long long)rows, N, grad_x);
    PCRL_CHECK_LAUNCH("segmax_bwd_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_augment_xyz_f32(const float* xyz_in, float* xyz_out, int32_t B, int32_t N, const pcrl_aug_desc* aug, void* stream) {
    if (!xyz_in || !xyz_out || !aug) return fail(PCRL_E_ARG, "NULL argument");
    if (B < 0 || N < 1) return fail(PCRL_E_ARG, "bad shape");
    if ((aug->flags & PCRL_AUG_AFFINE) && !aug->affine) return fail(PCRL_E_ARG, "AFFINE without matrix");
    if (aug->flags & PCRL_AUG_SUBSAMPLE) return fail(PCRL_E_ARG, "SUBSAMPLE is an index on the encoder's point load; slice the tensor for stand-alone use");
    if (B == 0) return PCRL_OK;
    AugParams p{xyz_in, xyz_out, B, N, aug->flags, aug->row_mul ? aug->row_mul : 1, aug->row_add, aug->jitter_lo, aug->jitter_hi,
                aug->jitter_noise, aug->affine, aug->seed, aug->offset, reinterpret_cast<const unsigned long long*>(aug->offset_ptr)};
    const long long n = (long long)B * N;
    hipLaunchKernelGGL(augment_xyz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("augment_xyz_kernel");
    return PCRL_OK;
}
