// Device code shared by the encoder forward and backward kernels: observation loading
// (PointCloudBase.preprocess, reference pyrl/networks/backbones/pointnet.py:49-73), the fused
// DrQ augmentations (pyrl/utils/augmentations/pcd_aug.py) and the per-point LayerNorm.
#pragma once
#include "common.h"

namespace pcrl {

struct ChanSrc {
    const void* base;     // already offset to this channel
    long long stride_b;   // elements
    long long stride_n;   // elements
    int dtype;            // PCRL_DT_*
    int div255;
};

// Where the points come from and how they are augmented (shared by forward and backward so that
// the backward recomputes exactly the forward's inputs).
struct CloudParams {
    int B, N, C;
    int aug_flags;
    int row_mul, row_add;     // augmentation row of cloud b = b * row_mul + row_add
    int row_div;              // cloud b reads the stored cloud b / row_div (>= 1)
    float jitter_lo, jitter_hi;
    const float* jitter_noise;
    const float* affine;
    unsigned long long seed, offset;
    const unsigned long long* offset_ptr;
    const int* point_index;   // SUBSAMPLE: stored point of position n (N is then the subsampled count)
    const int* n_ptr;         // SUBSAMPLE, optional: how many of the N positions exist, read at run time (1 <= *n_ptr <= N)
    int color_order;          // COLOR: four step ids, 4 bits each
    float color_fac[4], color_omf[4];
    const float* color_mean;  // [stored clouds]
    ChanSrc ch[PCRL_MAX_CHANNELS];
};

// ---- ColorJitterPoints = torchvision 0.14 ColorJitter on uint8 rgb (reference pyrl/utils/augmentations/pcd_aug.py:269-303;
// algorithm restated in oracle/color_jitter_ref.py).  r, g, b hold uint8 VALUES (0..255) as floats; every step ends with
// torch's `.to(uint8)` (truncation after the clamp).  Every fp32 operation below is the one torch performs, in its order.
__device__ __forceinline__ float cj_u8(float x) { return __builtin_floorf(__builtin_fminf(__builtin_fmaxf(x, 0.0f), 255.0f)); }
__device__ __forceinline__ float cj_gray(float r, float g, float b) { return __builtin_floorf((0.2989f * r + 0.587f * g) + 0.114f * b); }
__device__ __forceinline__ float cj_clamp01(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }

__device__ __forceinline__ void cj_hue(float& r8, float& g8, float& b8, float shift) {
    const float r = r8 / 255.0f, g = g8 / 255.0f, b = b8 / 255.0f;
    const float maxc = __builtin_fmaxf(__builtin_fmaxf(r, g), b), minc = __builtin_fminf(__builtin_fminf(r, g), b);
    const bool eqc = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eqc ? 1.0f : maxc);
    const float crd = eqc ? 1.0f : cr;
    const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
    const float hr = (maxc == r) ? (bc - gc) : 0.0f;
    const float hg = (maxc == g && maxc != r) ? ((2.0f + rc) - bc) : 0.0f;
    const float hb = (maxc != g && maxc != r) ? ((4.0f + gc) - rc) : 0.0f;
    float h = __builtin_fmodf(((hr + hg) + hb) / 6.0f + 1.0f, 1.0f);
    h = h + shift;                                  // torch: (h + f) % 1.0 == remainder(h + f, 1.0)
    float m = __builtin_fmodf(h, 1.0f);
    if (m != 0.0f && m < 0.0f) m = m + 1.0f;
    h = m;
    const float v = maxc;
    const float h6 = h * 6.0f;
    const float fi = __builtin_floorf(h6);
    const float f = h6 - fi;
    int i = (int)fi;
    const float pp = cj_clamp01(v * (1.0f - s));
    const float q = cj_clamp01(v * (1.0f - s * f));
    const float t = cj_clamp01(v * (1.0f - (s * (1.0f - f))));
    i = i % 6;
    if (i < 0) i += 6;
    const float ro = i == 0 ? v : i == 1 ? q : i == 2 ? pp : i == 3 ? pp : i == 4 ? t : v;
    const float go = i == 0 ? t : i == 1 ? v : i == 2 ? v : i == 3 ? q : i == 4 ? pp : pp;
    const float bo = i == 0 ? pp : i == 1 ? pp : i == 2 ? t : i == 3 ? v : i == 4 ? v : q;
    r8 = (float)(unsigned char)(int)(ro * 255.0f); g8 = (float)(unsigned char)(int)(go * 255.0f); b8 = (float)(unsigned char)(int)(bo * 255.0f);
}

// Steps [0, n_steps) of the drawn order (n_steps = 4: everything; the contrast-mean pre-pass stops before the contrast step).
__device__ __forceinline__ void cj_apply(float& r, float& g, float& b, int order, const float* fac, const float* omf, float mean, int n_steps) {
    for (int k = 0; k < n_steps; ++k) {
        const int op = (order >> (4 * k)) & 15;
        if (op == PCRL_COLOR_BRIGHTNESS) {
            r = cj_u8(fac[0] * r + omf[0] * 0.0f); g = cj_u8(fac[0] * g + omf[0] * 0.0f); b = cj_u8(fac[0] * b + omf[0] * 0.0f);
        } else if (op == PCRL_COLOR_CONTRAST) {
            const float m = omf[1] * mean;
            r = cj_u8(fac[1] * r + m); g = cj_u8(fac[1] * g + m); b = cj_u8(fac[1] * b + m);
        } else if (op == PCRL_COLOR_SATURATION) {
            const float m = omf[2] * cj_gray(r, g, b);
            r = cj_u8(fac[2] * r + m); g = cj_u8(fac[2] * g + m); b = cj_u8(fac[2] * b + m);
        } else if (op == PCRL_COLOR_HUE) {
            cj_hue(r, g, b, fac[3]);
        }
    }
}

// Channel descriptors are staged in LDS (not SGPRs: 16 x 32 B of kernel arguments would stay
// live across the whole tile body).  All lanes read the same descriptor; the dtype flags are
// made scalar again so the branches stay wave-uniform.
// raw: skip the /255 of a uint8 rgb channel (the colour jitter works on the uint8 values and divides afterwards)
__device__ __forceinline__ float load_chan(const ChanSrc* s_desc, int c, int b, int n, bool raw = false) {
    const ChanSrc d = s_desc[c];
    const long long off = (long long)b * d.stride_b + (long long)n * d.stride_n;
    const int dtype = __builtin_amdgcn_readfirstlane(d.dtype);
    const int div255 = __builtin_amdgcn_readfirstlane(d.div255);
    float v;
    if (dtype == PCRL_DT_F32) {
        v = static_cast<const float*>(d.base)[off];
    } else {
        v = (float)static_cast<const unsigned char*>(d.base)[off];
        if (dtype == PCRL_DT_BOOL) v = v != 0.0f ? 1.0f : 0.0f;
    }
    if (div255 && !raw) v = v / 255.0f;
    return v;
}

// Features of point n of cloud b, channels 0..2*T0-1 (zero beyond C), augmentation applied to xyz.
template <int T0>
__device__ __forceinline__ f32x16 load_point(const CloudParams& p, const ChanSrc* s_desc, int b, int n) {
    f32x16 x;
    const int n_src = p.point_index ? p.point_index[n] : n;
    const int b_src = p.row_div > 1 ? b / p.row_div : b;      // b is wave-uniform: one scalar division per tile
    const bool color = (p.aug_flags & PCRL_AUG_COLOR) != 0;       // channels 3..5 are the uint8 rgb key (checked on the host)
#pragma unroll
    for (int c = 0; c < 2 * T0; ++c) x[c] = c < p.C ? load_chan(s_desc, c, b_src, n_src, color && c >= 3 && c < 6) : 0.0f;
    if (color) {
        if (2 * T0 >= 6) {
            float r = x[3], g = x[4], bl = x[5];
            cj_apply(r, g, bl, p.color_order, p.color_fac, p.color_omf, p.color_mean ? p.color_mean[b_src] : 0.0f, 4);
            x[3] = r / 255.0f; x[4] = g / 255.0f; x[5] = bl / 255.0f;
        }
    }
    const long long row = (long long)b * p.row_mul + p.row_add;
    if (p.aug_flags & PCRL_AUG_AFFINE) {
        const float* M = p.affine + row * 12;
        const float x0 = x[0], x1 = x[1], x2 = x[2];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            x[j] = ((M[4 * j + 0] * x0 + M[4 * j + 1] * x1) + M[4 * j + 2] * x2) + M[4 * j + 3];
    }
    if (p.aug_flags & PCRL_AUG_JITTER) {
        if (p.jitter_noise) {
#pragma unroll
            for (int j = 0; j < 3; ++j) x[j] = x[j] + p.jitter_noise[(row * 3 + j) * p.N + n];
        } else {
            const unsigned long long e = (unsigned long long)row * p.N + n;
            const unsigned long long off = p.offset_ptr ? *p.offset_ptr : p.offset;
            uint32_t w[4];
            philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), (uint32_t)off, (uint32_t)(off >> 32),
                          (uint32_t)p.seed, (uint32_t)(p.seed >> 32), w);
#pragma unroll
            for (int j = 0; j < 3; ++j) x[j] = x[j] + u01_to_range(w[j], p.jitter_lo, p.jitter_hi);
        }
    }
    return x;
}

// Two accumulator registers at a time: gfx950's v_pk_{add,mul,fma}_f32 do two fp32 operations per VALU issue, and a
// VALU issue costs the same FP32 ALU cycles as the MFMAs (tools/probes/mfma_valu_overlap.hip).  The element order of
// the sums is unchanged (partial sums over registers r % 4, as in oracle/pcrl_oracle.c): results are bit-identical.
#define PCRL_PAIR(v, r) (f32x2{(v)[(r)], (v)[(r) + 1]})

template <int C>
__device__ __forceinline__ float ln_center_rstd(f32x16 (&a)[C / 32], float eps, bool* var_is_nan) {
    // mean and variance in the canonical order (oracle/pcrl_oracle.c); `a` is replaced by a - mean.
    constexpr int MB = C / 32;
    f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            p01 = p01 + PCRL_PAIR(a[mb], r);
            p23 = p23 + PCRL_PAIR(a[mb], r + 2);
        }
    }
    float lo, hi;
    both_halves((p01[0] + p01[1]) + (p23[0] + p23[1]), lo, hi);
    const float mean = (lo + hi) / (float)C;
    const f32x2 mean2 = {mean, mean};
    p01 = f32x2{0.f, 0.f}; p23 = f32x2{0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            const f32x2 c01 = PCRL_PAIR(a[mb], r) - mean2, c23 = PCRL_PAIR(a[mb], r + 2) - mean2;
            a[mb][r + 0] = c01[0]; a[mb][r + 1] = c01[1]; a[mb][r + 2] = c23[0]; a[mb][r + 3] = c23[1];
            p01 = __builtin_elementwise_fma(c01, c01, p01);
            p23 = __builtin_elementwise_fma(c23, c23, p23);
        }
    }
    both_halves((p01[0] + p01[1]) + (p23[0] + p23[1]), lo, hi);
    const float var = (lo + hi) / (float)C;
    *var_is_nan = var != var;
    return 1.0f / __builtin_sqrtf(var + eps);
}

// Per-point LayerNorm (biased variance, eps inside the sqrt, affine) + ReLU on an accumulator
// set (LayerNormkD.forward, reference pyrl/networks/modules/nn_layer.py:207-219).
// s_ln holds, per channel pair (2j, 2j + 1), {gamma_2j, gamma_2j+1, beta_2j, beta_2j+1} (ln_pair_table).
// Returns true for a point whose variance is NaN (all outputs NaN).
// NO_RELU: the affine output is left as it is (the max-pool that follows orders the raw bits as SIGNED integers, which is
// the ReLU for free: every non-positive value sorts below the smallest positive one).
template <int C, bool INT_RELU, bool NO_RELU = false>
__device__ __forceinline__ bool ln_relu_acc(f32x16 (&a)[C / 32], const float* __restrict__ s_ln, int half, float eps) {
    constexpr int MB = C / 32;
    bool nan_pt;
    const float rstd = ln_center_rstd<C>(a, eps, &nan_pt);
    const f32x2 rstd2 = {rstd, rstd};
    // gamma/beta of the 16 channels of a row block are fetched with 8 back-to-back 16-byte LDS reads
    // (one wait per block, the next block's reads already in flight) instead of a read + wait per pair.
    const f32x4* s_gb = reinterpret_cast<const f32x4*>(s_ln);
    f32x4 gb[2][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) gb[0][q] = s_gb[(acc_chan(2 * (q & 1) + 4 * (q >> 1), 0) + 4 * half) >> 1];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        if (mb + 1 < MB) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                gb[(mb + 1) & 1][q] = s_gb[(acc_chan((mb + 1) * 16 + 2 * (q & 1) + 4 * (q >> 1), 0) + 4 * half) >> 1];
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            // registers r, r + 1 of the block: channel pair (r & 3) / 2 of quad r >> 2 -> gb slot q = (r >> 2) * 2 + ((r & 3) >> 1)
            const f32x4 g4 = gb[mb & 1][(r >> 2) * 2 + ((r & 3) >> 1)];
            const f32x2 y = __builtin_elementwise_fma(PCRL_PAIR(a[mb], r) * rstd2, __builtin_shufflevector(g4, g4, 0, 1),
                                                      __builtin_shufflevector(g4, g4, 2, 3));
            const float y0 = y[0], y1 = y[1];
            if (NO_RELU) {
                a[mb][r] = y0;
                a[mb][r + 1] = y1;
            } else if (INT_RELU) {
                const int i0 = __builtin_bit_cast(int, y0), i1 = __builtin_bit_cast(int, y1);
                a[mb][r] = __builtin_bit_cast(float, i0 > 0 ? i0 : 0);
                a[mb][r + 1] = __builtin_bit_cast(float, i1 > 0 ? i1 : 0);
            } else {
                a[mb][r] = relu_nan(y0);
                a[mb][r + 1] = relu_nan(y1);
            }
        }
    }
    return nan_pt;
}

// Copy `n16` 16-byte pieces from global memory to LDS with all of a thread's loads in flight before the first LDS
// write.  (Written as a plain strided loop over blockDim.x the compiler emits load -> wait -> write per piece: 16
// dependent L2 round trips = 11 us for the 128 KB conv2 image.)
template <int THREADS, int N16>
__device__ __forceinline__ void stage_to_lds(f32x4* __restrict__ dst, const f32x4* __restrict__ src, int tid) {
    constexpr int PER = (N16 + THREADS - 1) / THREADS;
    f32x4 tmp[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (tid + THREADS * k < N16) tmp[k] = src[tid + THREADS * k];
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (tid + THREADS * k < N16) dst[tid + THREADS * k] = tmp[k];
}

// LDS image of a LayerNorm's affine parameters for ln_relu_acc from the packed image's [C][2] = (gamma, beta) rows.
__device__ __forceinline__ void ln_pair_table(float* s_ln, const float* packed_ln, int C, int tid, int nthreads) {
    for (int i = tid; i < 2 * C; i += nthreads) {
        const int c = i >> 1, which = i & 1;             // packed_ln[i] = which ? beta_c : gamma_c
        s_ln[4 * (c >> 1) + 2 * which + (c & 1)] = packed_ln[i];
    }
}

// Streams the A operands of one dense layer through a register ring DEPTH groups deep: group g
// (4 k-steps of one 32-row block) is consumed while groups g+1 .. g+DEPTH are in flight, so the
// L2 / LDS latency of an operand load is hidden behind 4*DEPTH MFMAs instead of being paid per group.
// `load(g)` returns the f32x4 of group g, `body(g, w)` issues its 4 MFMAs.  Fully unrolled.
template <int NG, int DEPTH, class LoadFn, class BodyFn>
__device__ __forceinline__ void stream_operands(LoadFn load, BodyFn body) {
    f32x4 ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < NG) ring[d] = load(d);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const f32x4 w = ring[g % DEPTH];
        if (g + DEPTH < NG) ring[g % DEPTH] = load(g + DEPTH);
        body(g, w);
    }
}

// One dense layer on the MFMA: acc[mb] += W[32mb.., :] . act, all MB row blocks, K = 8 * TQ channels.
// Two row blocks are advanced together so that consecutive MFMAs never touch the same accumulator
// (a dependent v_mfma_f32_32x32x2_f32 must otherwise issue in the exact cycle its predecessor retires;
// any instruction slipped in between -- an operand load, a wait -- idles the matrix pipe), and the A
// operands of pair-group g + DEPTH are in flight while group g computes.
// Backward kernels: one 32-row block of an operand array (h0, h1, dz2, dz1, dz0) for one tile.  The lane that holds point p (l31, wave half h) owns rows
// acc_chan(r, h), r = 0..15, of that point; the wgrad kernel reads pieces [octet p >> 3][k-lane (p >> 2) & 1][row][slot p & 3].
// Stored straight from the registers that is sixteen 4-byte stores per block whose 64 lanes hit sixteen different 16-byte
// pieces -- 199-263 such stores per tile, ~100 cycles of the vector-memory pipe each, and every later load of the wave queues behind
// them (vmcnt is one in-order counter).  Instead the block is transposed through 4.1 KB of the wave's LDS (pitch 33 float4 per
// (octet, k-lane) pair: both directions conflict-free) and leaves as four 1 KB stores, one octet each, 16 bytes per lane.
constexpr int kTrFloats = 8 * 33 * 4;            // per wave
__device__ __forceinline__ void store_block_pieces(const __amdgpu_buffer_rsrc_t& rs, float* s_tr, unsigned blk_bytes, unsigned tile_bytes,
                                                   const f32x16& v, int l31, int half, int lane) {
    float* w = s_tr + ((l31 >> 2) * 33 + 4 * half) * 4 + (l31 & 3);
#pragma unroll
    for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2)) * 4] = v[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const f32x4* rd = reinterpret_cast<const f32x4*>(s_tr) + (lane >> 5) * 33 + (lane & 31);
    f32x4 t[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) t[n] = rd[2 * n * 33];
    // soffset stays the literal 0 and the block / octet offsets travel in the vector offset and the instruction's immediate: with
    // an SGPR soffset the compiler's hazard recogniser assumes a 128-bit buffer store has read its data registers when it issues
    // and lets the next VALU instruction overwrite them -- on gfx950 it has not (measured: lanes 12-15 of every 16 of the second
    // register arrived holding the following v_pk_add's result).  With a literal soffset the required wait states are inserted.
    const unsigned voff = 16u * (unsigned)lane + tile_bytes + blk_bytes;
#pragma unroll
    for (int n = 0; n < 4; ++n) buf_store_f4(rs, voff + 1024u * (unsigned)n, 0u, t[n]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// load(mb, tq) -> the f32x4 holding A operands of k-steps 4tq..4tq+3 of row block mb;
// act(t) -> the B operand (activation register) of k-step t.
template <int MB, int TQ, int DEPTH, bool ZERO_START = true, class LoadFn, class ActFn>
__device__ __forceinline__ void dense_layer_mfma(f32x16 (&acc)[MB], LoadFn load, ActFn act) {
    if constexpr (MB == 1) {
        // a single 32-row block (c = 32 channels): one accumulator chain, operands DEPTH groups ahead
        f32x4 ring[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < TQ) ring[d] = load(0, d);
#pragma unroll
        for (int g = 0; g < TQ; ++g) {
            const f32x4 w = ring[g % DEPTH];
            if (g + DEPTH < TQ) ring[g % DEPTH] = load(0, g + DEPTH);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool first = ZERO_START && g == 0 && j == 0;
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], act(4 * g + j), first ? zero : acc[0], 0, 0, 0);
            }
        }
        return;
    } else {
    static_assert(MB % 2 == 0, "row blocks are processed in pairs");
    constexpr int NG = (MB / 2) * TQ;
    f32x4 ra[DEPTH], rb[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < NG) { ra[d] = load(2 * (d / TQ), d % TQ); rb[d] = load(2 * (d / TQ) + 1, d % TQ); }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const f32x4 wa = ra[g % DEPTH], wb = rb[g % DEPTH];
        if (g + DEPTH < NG) {
            ra[g % DEPTH] = load(2 * ((g + DEPTH) / TQ), (g + DEPTH) % TQ);
            rb[g % DEPTH] = load(2 * ((g + DEPTH) / TQ) + 1, (g + DEPTH) % TQ);
        }
        const int mb = 2 * (g / TQ), tq = g % TQ;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = act(4 * tq + j);
            // the first k-step takes the literal 0 as its C operand: the accumulators need no zero fill (on gfx950 a
            // v_mov costs the same FP32 ALU cycles the MFMAs need -- tools/probes/mfma_valu_overlap.hip)
            const bool first = ZERO_START && tq == 0 && j == 0;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], b, first ? zero : acc[mb], 0, 0, 0);
            acc[mb + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[j], b, first ? zero : acc[mb + 1], 0, 0, 0);
        }
    }
    }
}

// dense_layer_mfma for operands that stream from L2 (MB even): the same pair loop, with the scheduler forbidden to move anything across
// the points where the ring is refilled.  Left free, the machine scheduler sinks every load to just in front of its first use (it
// minimises live registers even where 512 are available): the ISA of the backward's conv1 showed `buffer_load x2; s_waitcnt vmcnt(1);
// mfma; s_waitcnt vmcnt(0); mfma ...` -- a ring of depth one, an L2 round trip exposed per 8 MFMAs -- whatever DEPTH said.
template <int MB, int TQ, int DEPTH, class LoadFn, class ActFn>
__device__ __forceinline__ void dense_layer_mfma_stream(f32x16 (&acc)[MB], LoadFn load, ActFn act) {
    static_assert(MB % 2 == 0, "row blocks are processed in pairs");
    constexpr int NG = (MB / 2) * TQ;
    f32x4 ra[DEPTH], rb[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < NG) { ra[d] = load(2 * (d / TQ), d % TQ); rb[d] = load(2 * (d / TQ) + 1, d % TQ); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const f32x4 wa = ra[g % DEPTH], wb = rb[g % DEPTH];
        const int mb = 2 * (g / TQ), tq = g % TQ;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = act(4 * tq + j);
            const bool first = tq == 0 && j == 0;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], b, first ? zero : acc[mb], 0, 0, 0);
            acc[mb + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[j], b, first ? zero : acc[mb + 1], 0, 0, 0);
        }
        if (g + DEPTH < NG) {       // refill the slot just consumed, DEPTH groups ahead of its use
            ra[g % DEPTH] = load(2 * ((g + DEPTH) / TQ), (g + DEPTH) % TQ);
            rb[g % DEPTH] = load(2 * ((g + DEPTH) / TQ) + 1, (g + DEPTH) % TQ);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Mixed-precision dense layer: acc[mb] (+)= W[32mb.., :] . act with bf16 operands on v_mfma_f32_32x32x16_bf16 and fp32
// accumulation.  One MFMA contracts 16 input channels: the 8 accumulator registers 8g..8g+7 of BOTH wave halves, i.e.
// channels acc_chan(8g + r, h) -- the weight image (PackedLayout::w1b/w2b) is packed in exactly that order, so the
// activations still never leave registers: 8 fp32 registers -> 4 packed bf16 registers (v_cvt_pk_bf16_f32, RNE).
// load(mb, g) -> 16 bytes = the 8 bf16 A operands of lane (i, h) for row block mb, channel group g; act(t) -> register t.
template <int MB, int G, class LoadFn, class ActFn>
__device__ __forceinline__ void dense_layer_bf16(f32x16 (&acc)[MB], LoadFn load, ActFn act) {
    f32x4 w[2][MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) w[0][mb] = load(mb, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g + 1 < G) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) w[(g + 1) & 1][mb] = load(mb, g + 1);
        }
        bf16x8 b;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            const bf16x2 pr = __builtin_convertvector(f32x2{act(8 * g + r), act(8 * g + r + 1)}, bf16x2);
            b[r] = pr[0]; b[r + 1] = pr[1];
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[g & 1][mb]), b, g == 0 ? zero : acc[mb], 0, 0, 0);
    }
}

// Split-precision dense layer (experimental): acc[mb] = W[32mb.., :] . act in ~fp32 accuracy on the bf16 matrix cores.  Every
// fp32 operand is the exact sum of three bf16 terms (hi + mid + lo, 8 significand bits each, by truncation); of the nine
// term products the six largest are kept -- hi hi, hi mid, mid hi, hi lo, lo hi, mid mid -- so what is dropped is below
// 3 x 2^-24 of |w| |a| per product, the size of fp32's own rounding.  Six v_mfma_f32_32x32x16_bf16 (32 cycles each) replace
// eight v_mfma_f32_32x32x2_f32 (64 cycles each) per 16 input channels: 2.7x less matrix time, paid with ~6 vector
// instructions per activation for the split.  Not bit-comparable with the fp32 chain (the products are exact, the
// accumulation order differs): offered next to the exact kernel, never instead of it.
// load(k, mb, g) -> 16 bytes = the 8 bf16 A operands of term k; act(t) -> activation register t.
template <int MB, int G, class LoadFn, class ActFn>
__device__ __forceinline__ void dense_layer_split(f32x16 (&acc)[MB], LoadFn load, ActFn act) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
        u32x4 bw[3];                       // the three B operands: 8 bf16 = four 32-bit words each
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            float x0 = act(8 * g + r), x1 = act(8 * g + r + 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // the upper halves of the two fp32 words = their truncated bf16 terms, element r in the low half: one v_perm
                bw[k][r >> 1] = __builtin_amdgcn_perm(f2u(x1), f2u(x0), 0x07060302u);
                if (k < 2) {               // exact remainders: at most 16 (then 8) significand bits are left
                    x0 = x0 - u2f(f2u(x0) & 0xFFFF0000u);
                    x1 = x1 - u2f(f2u(x1) & 0xFFFF0000u);
                }
            }
        }
        bf16x8 b[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) b[k] = __builtin_bit_cast(bf16x8, bw[k]);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // Row blocks in pairs: the twelve MFMAs of a pair alternate between two accumulators (no MFMA waits for its
        // predecessor's result), and the six operand pieces of the next pair are in flight meanwhile.
        constexpr int NP = (MB + 1) / 2;
        f32x4 w[2][2][3];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (e < MB) w[0][e][k] = load(k, e, g);
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            if (pr + 1 < NP) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (2 * (pr + 1) + e < MB) w[(pr + 1) & 1][e][k] = load(k, 2 * (pr + 1) + e, g);
            }
            const int m0 = 2 * pr, m1 = 2 * pr + 1 < MB ? 2 * pr + 1 : 2 * pr;
            const bool two = 2 * pr + 1 < MB;
            f32x16 a0 = g == 0 ? zero : acc[m0], a1 = g == 0 ? zero : acc[m1];
            // term pairs from the smallest product to the largest: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
            constexpr int WK[6] = {2, 0, 1, 1, 0, 0}, BK[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[pr & 1][0][WK[t]]), b[BK[t]], a0, 0, 0, 0);
                if (two) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[pr & 1][1][WK[t]]), b[BK[t]], a1, 0, 0, 0);
            }
            acc[m0] = a0;
            if (two) acc[m1] = a1;
        }
    }
}

bool encoder_dims_supported(int c1, int c2, int c3);
// Host side: validate the descriptors of the C ABI and flatten them into CloudParams.
int fill_cloud_params(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug, int expect_channels, CloudParams* out);

}  // namespace pcrl
