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

struct ColorParams {
    const unsigned char* in; unsigned char* out; long long sb, sc, sn; int B, N;
    int order; float fac[4], omf[4]; const float* mean; float* mean_out; int n_steps;
};

// One workgroup per cloud: the steps before the contrast step, then the mean of the grayscale values (integers: the fp32 sum
// is exact in any order up to 65 793 points, as is torch's).
__global__ __launch_bounds__(256) void color_contrast_mean_kernel(const ColorParams p) {
    __shared__ float s_part[256];
    const int b = blockIdx.x;
    float acc = 0.0f;
    for (int n = threadIdx.x; n < p.N; n += 256) {
        const unsigned char* src = p.in + (long long)b * p.sb + (long long)n * p.sn;
        float r = (float)src[0], g = (float)src[p.sc], bl = (float)src[2 * p.sc];
        cj_apply(r, g, bl, p.order, p.fac, p.omf, 0.0f, p.n_steps);
        acc = acc + cj_gray(r, g, bl);
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) s_part[threadIdx.x] = s_part[threadIdx.x] + s_part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.mean_out[b] = s_part[0] / (float)p.N;
}

__global__ __launch_bounds__(256) void color_jitter_kernel(const ColorParams p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)p.B * p.N) return;
    const int b = (int)(i / p.N), n = (int)(i - (long long)b * p.N);
    const long long off = (long long)b * p.sb + (long long)n * p.sn;
    float r = (float)p.in[off], g = (float)p.in[off + p.sc], bl = (float)p.in[off + 2 * p.sc];
    cj_apply(r, g, bl, p.order, p.fac, p.omf, p.mean ? p.mean[b] : 0.0f, 4);
    p.out[off] = (unsigned char)r; p.out[off + p.sc] = (unsigned char)g; p.out[off + 2 * p.sc] = (unsigned char)bl;
}

// ---- GlobalRotScaleTrans: the per-cloud [3 x 4] matrix drawn on the device -------------------------------------------------
// pcd_aug.py:178-196: mat = 0; R(angle ~ U(rot_range)) about rot_axis into [:3,:3]; rows of [:3,:] scaled by s_i ~ U(scale range)
// (so the scale acts only through a rotation block, and before the translation is written); translation (U(0,1) - 0.5) * 2 * range,
// zero for the LAST cloud unless shift_height; apply_rot_trans skips the product when there is no rotation (identity here) and the
// shift when there is no translation range (zero here).  The reference draws with ~15 ATen launches per call from torch's
// generator; this is one launch, Philox4x32-10 keyed by (seed, offset | *offset_ptr, cloud) -- a launch replayed from a hipGraph
// draws fresh matrices when offset_ptr names a counter the step advances.
struct AffineSampleParams {
    float* mat; int B, rot_axis, has_rot, has_scale, has_trans, shift_height;
    float rot_lo, rot_hi, scale_lo, scale_hi, trans[3];
    unsigned long long seed, offset; const unsigned long long* offset_ptr;
    // a SECOND draw of the same transform in the same launch (blocks past the first draw's: pcrl_affine_sample_pair_f32), or NULL
    float* mat2; unsigned long long seed2, offset2;
};
__global__ __launch_bounds__(256) void affine_sample_kernel(const AffineSampleParams p) {
    const int per_draw = (p.B + 255) / 256;
    const bool second = (int)blockIdx.x >= per_draw;
    const int b = ((int)blockIdx.x - (second ? per_draw : 0)) * 256 + threadIdx.x;
    if (b >= p.B) return;
    const unsigned long long off = p.offset_ptr ? *p.offset_ptr : (second ? p.offset2 : p.offset);
    const unsigned long long seed = second ? p.seed2 : p.seed;
    uint32_t w0[4], w1[4];
    philox4x32_10((uint32_t)b, 0u, (uint32_t)off, (uint32_t)(off >> 32), (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0xA0F1E2D3u, w0);
    philox4x32_10((uint32_t)b, 1u, (uint32_t)off, (uint32_t)(off >> 32), (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0xA0F1E2D3u, w1);
    float m[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = 0.0f;
    if (p.has_rot) {
        const float ang = u01_to_range(w0[0], p.rot_lo, p.rot_hi);
        const float c = cosf(ang), s = sinf(ang);
        const int a = p.rot_axis, j = (a + 1) % 3, k = (a + 2) % 3;          // batch_rot_with_axis, ops.py:171-183
        m[a][a] = 1.0f; m[j][j] = c; m[k][k] = c; m[j][k] = -s; m[k][j] = s;
    }
    if (p.has_scale) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float sc = u01_to_range(w0[1 + i], p.scale_lo, p.scale_hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] *= sc;
        }
    }
    if (p.has_trans) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float u = (float)(w1[i] >> 8) * (1.0f / 16777216.0f);
            m[i][3] = (!p.shift_height && b == p.B - 1) ? 0.0f : (u - 0.5f) * 2.0f * p.trans[i];
        }
    }
    if (!p.has_rot) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) m[i][j] = i == j ? 1.0f : 0.0f;
    }
    float* dst = (second ? p.mat2 : p.mat) + (long long)b * 12;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[4 * i + j] = m[i][j];
}

}  // namespace pcrl

using namespace pcrl;

static int color_fill(const pcrl_aug_desc* aug, ColorParams& p, int* contrast_pos) {
    if (!aug || !(aug->flags & PCRL_AUG_COLOR)) return fail(PCRL_E_ARG, "aug without PCRL_AUG_COLOR");
    p.order = aug->color_order; p.mean = aug->color_mean;
    *contrast_pos = -1;
    for (int k = 0; k < 4; ++k) {
        p.fac[k] = aug->color_factor[k]; p.omf[k] = aug->color_one_minus[k];
        const int op = (aug->color_order >> (4 * k)) & 15;
        if (op == PCRL_COLOR_CONTRAST) *contrast_pos = k;
        else if (op != PCRL_COLOR_BRIGHTNESS && op != PCRL_COLOR_SATURATION && op != PCRL_COLOR_HUE && op != PCRL_COLOR_SKIP)
            return fail(PCRL_E_ARG, "bad colour step id %d", op);
    }
    return PCRL_OK;
}

extern "C" int pcrl_color_contrast_mean_u8(const uint8_t* rgb, int64_t stride_b, int64_t stride_c, int64_t stride_n, int32_t B, int32_t N,
                                           const pcrl_aug_desc* aug, float* mean_out, void* stream) {
    if (!rgb || !mean_out) return fail(PCRL_E_ARG, "NULL argument");
    if (B < 0 || N < 1) return fail(PCRL_E_ARG, "bad shape");
    ColorParams p{};
    int cpos;
    if (int rc = color_fill(aug, p, &cpos)) return rc;
    if (B == 0) return PCRL_OK;
    p.in = rgb; p.sb = stride_b; p.sc = stride_c; p.sn = stride_n; p.B = B; p.N = N; p.mean_out = mean_out;
    p.n_steps = cpos < 0 ? 0 : cpos;
    hipLaunchKernelGGL(color_contrast_mean_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("color_contrast_mean_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_color_jitter_u8(const uint8_t* rgb_in, uint8_t* rgb_out, int64_t stride_b, int64_t stride_c, int64_t stride_n, int32_t B, int32_t N,
                                    const pcrl_aug_desc* aug, void* stream) {
    if (!rgb_in || !rgb_out) return fail(PCRL_E_ARG, "NULL argument");
    if (B < 0 || N < 1) return fail(PCRL_E_ARG, "bad shape");
    ColorParams p{};
    int cpos;
    if (int rc = color_fill(aug, p, &cpos)) return rc;
    if (cpos >= 0 && !aug->color_mean) return fail(PCRL_E_ARG, "contrast step without color_mean");
    if (B == 0) return PCRL_OK;
    p.in = rgb_in; p.out = rgb_out; p.sb = stride_b; p.sc = stride_c; p.sn = stride_n; p.B = B; p.N = N;
    const long long n = (long long)B * N;
    hipLaunchKernelGGL(color_jitter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("color_jitter_kernel");
    return PCRL_OK;
}

static int affine_sample_launch(float* mat, float* mat2, int32_t B, int32_t rot_axis, const float* rot_range, const float* scale_range,
                                const float* translation_range, int32_t shift_height, uint64_t seed, uint64_t offset, uint64_t seed2,
                                uint64_t offset2, const uint64_t* offset_ptr, void* stream);

extern "C" int pcrl_affine_sample_f32(float* mat, int32_t B, int32_t rot_axis, const float* rot_range, const float* scale_range,
                                      const float* translation_range, int32_t shift_height, uint64_t seed, uint64_t offset,
                                      const uint64_t* offset_ptr, void* stream) {
    return affine_sample_launch(mat, nullptr, B, rot_axis, rot_range, scale_range, translation_range, shift_height, seed, offset, 0, 0, offset_ptr, stream);
}

extern "C" int pcrl_affine_sample_pair_f32(float* mat, float* mat2, int32_t B, int32_t rot_axis, const float* rot_range, const float* scale_range,
                                           const float* translation_range, int32_t shift_height, uint64_t seed, uint64_t offset, uint64_t seed2,
                                           uint64_t offset2, const uint64_t* offset_ptr, void* stream) {
    if (!mat2) return fail(PCRL_E_ARG, "affine sample pair: mat2 is NULL");
    return affine_sample_launch(mat, mat2, B, rot_axis, rot_range, scale_range, translation_range, shift_height, seed, offset, seed2, offset2, offset_ptr, stream);
}

static int affine_sample_launch(float* mat, float* mat2, int32_t B, int32_t rot_axis, const float* rot_range, const float* scale_range,
                                const float* translation_range, int32_t shift_height, uint64_t seed, uint64_t offset, uint64_t seed2,
                                uint64_t offset2, const uint64_t* offset_ptr, void* stream) {
    if (!mat || B < 1 || rot_axis < 0 || rot_axis > 2) return fail(PCRL_E_ARG, "affine sample: bad arguments");
    AffineSampleParams p{};
    p.mat = mat; p.B = B; p.rot_axis = rot_axis; p.shift_height = shift_height;
    p.has_rot = rot_range != nullptr; p.has_scale = scale_range != nullptr; p.has_trans = translation_range != nullptr;
    if (rot_range) { p.rot_lo = rot_range[0]; p.rot_hi = rot_range[1]; }
    if (scale_range) { p.scale_lo = scale_range[0]; p.scale_hi = scale_range[1]; }
    if (translation_range) for (int i = 0; i < 3; ++i) p.trans[i] = translation_range[i];
    p.seed = seed; p.offset = offset; p.offset_ptr = reinterpret_cast<const unsigned long long*>(offset_ptr);
    p.mat2 = mat2; p.seed2 = seed2; p.offset2 = offset2;
    hipLaunchKernelGGL(affine_sample_kernel, dim3((B + 255) / 256 * (mat2 ? 2 : 1)), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("affine_sample_kernel");
    return PCRL_OK;
}

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
