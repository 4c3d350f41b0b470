// Internal helpers shared by the HIP translation units of libpcrl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pcrl.h"

namespace pcrl {

// Records a thread-local error message and returns `code` (see pcrl_last_error()).
int fail(int code, const char* fmt, ...);
// Number of compute units of the current device (cached per device).
int num_cus();
// hipFuncAttributeMaxDynamicSharedMemorySize >= bytes for `kernel` on the current device (done once per device and kernel).
int ensure_dynamic_lds(const void* kernel, size_t bytes);

#define PCRL_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return ::pcrl::fail(PCRL_E_LAUNCH, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define PCRL_CHECK_LAUNCH(name)                                                               \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess) return ::pcrl::fail(PCRL_E_LAUNCH, "launch %s: %s", name, hipGetErrorString(_e)); \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- 32x32 MFMA accumulator geometry -------------------------------------------------
// v_mfma_f32_32x32x2_f32: lane l holds column (l & 31); register r of the tile holds row
// (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).  With output channels on rows and points on
// columns, accumulator slot R = 16 * block + r of lane-half h holds channel acc_chan(R, h)
// of point (l & 31) -- which is exactly the B-operand layout of k-step R of the next
// layer (B[k = l >> 5][j = l & 31]), so activations never leave registers between layers.
__host__ __device__ constexpr int acc_chan(int R, int h) {
    return 32 * (R >> 4) + ((R & 15) & 3) + 8 * ((R & 15) >> 2) + 4 * h;
}

// Offsets (in floats) inside the packed weight image, shared by pack kernel, forward and backward.
struct PackedLayout {
    int T0, C1, C2, C3;
    __host__ __device__ constexpr int w0() const { return 0; }                       // [C1/32][T0][64]
    __host__ __device__ constexpr int b0() const { return w0() + (C1 / 32) * T0 * 64; }  // [C1]
    __host__ __device__ constexpr int w1() const { return align4(b0() + C1); }       // [C2/32][C1/8][64][4]
    __host__ __device__ constexpr int ln1() const { return w1() + C2 * C1; }         // [C2][2] (gamma, beta)
    __host__ __device__ constexpr int w2() const { return align4(ln1() + 2 * C2); }  // [C3/32][C2/8][64][4]
    __host__ __device__ constexpr int ln2() const { return w2() + C3 * C2; }         // [C3][2]
    __host__ __device__ constexpr int w2t() const { return align4(ln2() + 2 * C3); } // [C2/32][C3/8][64][4]
    __host__ __device__ constexpr int w1t() const { return w2t() + C2 * C3; }        // [C1/32][C2/8][64][4]
    // bf16 images for the mixed-precision forward: [row block][16-channel group][64 lanes][8 bf16], two per float slot
    __host__ __device__ constexpr int w1b() const { return align4(w1t() + C1 * C2); }    // conv1: C2 x C1 bf16
    __host__ __device__ constexpr int w2b() const { return align4(w1b() + C1 * C2 / 2); } // conv2: C3 x C2 bf16
    __host__ __device__ constexpr int w2tb() const { return align4(w2b() + C2 * C3 / 2); } // conv2 transposed: C2 rows x C3 k, bf16
    __host__ __device__ constexpr int w1tb() const { return align4(w2tb() + C2 * C3 / 2); } // conv1 transposed: C1 rows x C2 k, bf16
    // split-precision forward (experimental): three bf16 images per layer, hi / mid / lo with w = hi + mid + lo exactly, same
    // [row block][16-channel group][64 lanes][8 bf16] order as w1b / w2b
    __host__ __device__ constexpr int w1s(int k) const { return align4(w1tb() + C1 * C2 / 2) + k * (C1 * C2 / 2); }
    __host__ __device__ constexpr int w2s(int k) const { return align4(w1s(3)) + k * (C2 * C3 / 2); }
    // ... and of the transposed matrices for the backward's data-gradient GEMMs (rows = c2 / c1, contraction over c3 / c2)
    __host__ __device__ constexpr int w2ts(int k) const { return align4(w2s(3)) + k * (C2 * C3 / 2); }
    __host__ __device__ constexpr int w1ts(int k) const { return align4(w2ts(3)) + k * (C1 * C2 / 2); }
    __host__ __device__ constexpr int total() const { return align4(w1ts(3)); }
    __host__ __device__ static constexpr int align4(int x) { return (x + 3) & ~3; }
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// fp32 -> bf16 bits, round to nearest even, NaN stays NaN (same rule as v_cvt_pk_bf16_f32 / torch.Tensor.bfloat16()).
__host__ __device__ inline unsigned bf16_rne_bits(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// w = hi + mid + lo with three bf16 terms, exactly (8 + 8 + 8 significand bits by truncation): term k of x.
__host__ __device__ inline unsigned bf16_split_bits(float x, int k) {
    float r = x;
    for (int i = 0; i < k; ++i) r = r - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r) & 0xFFFF0000u);
    return __builtin_bit_cast(unsigned, r) >> 16;
}

// Buffer addressing: SGPR resource + one 32-bit VGPR byte offset + scalar/immediate offset.  Used
// wherever a wave walks many constant-stride pieces of one array, so that no 64-bit VGPR address
// is materialised per piece.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load_f4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ f32x2 buf_load_f2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ float buf_load_f1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store_f1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// NOTE: pass soff as the literal 0 when VALU code may follow closely (see store_block_pieces in encoder_bwd_gram.h: the store-data
// hazard of stores wider than 64 bits is only guarded by the compiler for a non-register soffset).
__device__ __forceinline__ void buf_store_f4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, voff, soff, 0);
}

__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// lo for the lower wave half, hi for the upper one, as ONE bit-select (written as `half ? hi : lo` on two elements of a
// vector the compiler turns it into a 15-deep compare/select chain over a run-time element index).
__device__ __forceinline__ float half_select(float lo, float hi, unsigned half_mask) {
    return u2f((f2u(hi) & half_mask) | (f2u(lo) & ~half_mask));
}

// ReLU that propagates NaN and maps -0 to +0 (torch.relu semantics).
__device__ __forceinline__ float relu_nan(float x) { return !(x <= 0.0f) ? x : 0.0f; }

// Values of lane (l & 31) in the low half and in the high half of the wave, in every lane:
// v_permlane32_swap exchanges vdst[32..63] with vsrc[0..31].
__device__ __forceinline__ void both_halves(float x, float& lo, float& hi) {
    auto r = __builtin_amdgcn_permlane32_swap(f2u(x), f2u(x), false, false);
    lo = u2f(r[0]);
    hi = u2f(r[1]);
}

// Max over the 32 lanes of each wave half, result in every lane, unsigned compare.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned allreduce_umax32(unsigned v) {
    v = umax_(v, dpp_u<0xB1>(v));    // quad_perm [1,0,3,2]
    v = umax_(v, dpp_u<0x4E>(v));    // quad_perm [2,3,0,1]
    v = umax_(v, dpp_u<0x141>(v));   // row_half_mirror
    v = umax_(v, dpp_u<0x140>(v));   // row_mirror
    // lane ^ 16 without the LDS pipe: v_permlane16_swap exchanges row 1 of vdst with row 0 of vsrc and
    // row 3 of vdst with row 2 of vsrc (rows = 16 lanes), so with both operands = v the two results
    // hold {row0,row0,row2,row2} and {row1,row1,row3,row3}.
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return umax_(r[0], r[1]);
}

__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned allreduce_umin32(unsigned v) {
    v = umin_(v, dpp_u<0xB1>(v));
    v = umin_(v, dpp_u<0x4E>(v));
    v = umin_(v, dpp_u<0x141>(v));
    v = umin_(v, dpp_u<0x140>(v));
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return umin_(r[0], r[1]);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return u2f((unsigned)__builtin_amdgcn_update_dpp(0, (int)f2u(v), CTRL, 0xF, 0xF, false));
}
// Sum over the 32 lanes of each wave half, same fixed butterfly order in every lane.
__device__ __forceinline__ float allreduce_add32(float v) {
    v = v + dpp_f<0xB1>(v);
    v = v + dpp_f<0x4E>(v);
    v = v + dpp_f<0x141>(v);
    v = v + dpp_f<0x140>(v);
    v = v + u2f((unsigned)__builtin_amdgcn_ds_swizzle((int)f2u(v), 0x401F));
    return v;
}

// Sixteen independent 32-lane sums advanced together (the dependent DPP chains of one sum at a time cost ~100
// cycles each); lane ^ 16 through v_permlane16_swap instead of ds_swizzle keeps the LDS pipe out of it.
__device__ __forceinline__ void allreduce_add32_x16(float (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = v[r] + dpp_f<0xB1>(v[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = v[r] + dpp_f<0x4E>(v[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = v[r] + dpp_f<0x141>(v[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = v[r] + dpp_f<0x140>(v[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        auto sw = __builtin_amdgcn_permlane16_swap(f2u(v[r]), f2u(v[r]), false, false);
        v[r] = u2f(sw[0]) + u2f(sw[1]);
    }
}

// Philox4x32-10 (Salmon et al. 2011); one call yields four 32-bit words.
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// U[lo, hi) from a 32-bit word with 24 bits of mantissa.
__host__ __device__ inline float u01_to_range(uint32_t w, float lo, float hi) {
    return lo + (float)(w >> 8) * (1.0f / 16777216.0f) * (hi - lo);
}

// ---- the scalars a step reports, gathered by ONE workgroup of 256 threads (csrc/tail.hip: gather_scalars_kernel; csrc/optim.hip: the first
// workgroup of a published optimizer pass, pcrl_adam_step_published_gather_f32): up to four deferred optimizer passes are finished first --
// gradient norm = sqrt of the fixed-order sum of the pass's per-block partial sums, step count += 1 where the pass has not advanced it itself
// (step == NULL) -- so a norm can be one of the gathered scalars. ------------------------------------------------------------------------
constexpr int kMaxScalars = 16;
constexpr int kMaxFinalize = 4;
struct ScalarListParams {
    const float* src[kMaxScalars]; float* dst[kMaxScalars]; unsigned exp_mask; int n;
    const float* partial[kMaxFinalize]; int n_partial[kMaxFinalize]; float* norm[kMaxFinalize]; int* step[kMaxFinalize]; int n_fin;
    float* host_out;      // optional pinned host mirror of the n values (slots pre-filled with 0xFFFFFFFF by the host)
};
// s_part: kMaxFinalize * 4 floats of LDS; tid in [0, 256)
__device__ __forceinline__ void gather_scalars_block(const ScalarListParams& p, int tid, float* s_part) {
    // The partials were written by the previous launch on every XCD: each load is a miss.  All of them (every optimizer's, eight per
    // thread at a time) are requested before the first sum; thread t still adds partial[t], partial[t + 256], ... in that order, then the
    // fixed-order tree gradnorm_finalize_kernel uses.
    float s[kMaxFinalize];
#pragma unroll
    for (int f = 0; f < kMaxFinalize; ++f) {
        s[f] = 0.0f;
        if (f < p.n_fin) {
            const int n = p.n_partial[f];
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = tid + 256 * k; v[k] = i < n ? p.partial[f][i] : 0.0f; }
#pragma unroll
            for (int k = 0; k < 8; ++k) if (tid + 256 * k < n) s[f] += v[k];
            for (int i = tid + 2048; i < n; i += 256) s[f] += p.partial[f][i];
        }
    }
#pragma unroll
    for (int f = 0; f < kMaxFinalize; ++f) {
        for (int off = 32; off > 0; off >>= 1) s[f] += __shfl_down(s[f], off, 64);
        if ((tid & 63) == 0) s_part[f * 4 + (tid >> 6)] = s[f];
    }
    __syncthreads();
    if (tid < p.n_fin) {
        const int f = tid;
        if (p.norm[f]) p.norm[f][0] = __builtin_sqrtf((s_part[f * 4 + 0] + s_part[f * 4 + 1]) + (s_part[f * 4 + 2] + s_part[f * 4 + 3]));
        if (p.step[f]) p.step[f][0] += 1;
    }
    __threadfence_block();
    __syncthreads();
    const int i = tid;
    float out = 0.0f;
    if (i < p.n) {
        const float v = p.src[i][0];
        out = ((p.exp_mask >> i) & 1u) ? expf(v) : v;
        p.dst[i][0] = out;
    }
    // Pinned host mirror: every value is ONE 4-byte store, so no ordering between them is needed -- the host pre-fills the
    // slots with a sentinel bit pattern (0xFFFFFFFF, a NaN no arithmetic here produces) and reads a value as soon as its
    // slot differs.  (A system-scope fence + flag would first write the whole dirty L2 back; that happens at the end of the
    // kernel anyway, now under the host's work between two steps.)
    if (p.host_out && i < p.n) {
        unsigned bits = __builtin_bit_cast(unsigned, out);
        if (bits == 0xFFFFFFFFu) bits = 0x7FC00000u;
        __builtin_nontemporal_store(bits, reinterpret_cast<unsigned*>(p.host_out) + i);
    }
}

// host side: the C arguments of pcrl_gather_scalars_*_f32 -> ScalarListParams (validated)
inline int scalar_list_fill(const float* const* src, float* const* dst, const int32_t* take_exp, int32_t n,
                            const pcrl_adam_pending* pending, int32_t n_pending, float* host_out, ScalarListParams& p) {
    if (n < 0 || n > kMaxScalars || (n > 0 && (!src || !dst))) return fail(PCRL_E_ARG, "pcrl_gather_scalars_f32: 0 <= n <= %d", kMaxScalars);
    if (n_pending < 0 || n_pending > kMaxFinalize || (n_pending > 0 && !pending))
        return fail(PCRL_E_ARG, "pcrl_gather_scalars_f32: 0 <= n_pending <= %d", kMaxFinalize);
    p = ScalarListParams{};
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i]) return fail(PCRL_E_ARG, "NULL scalar pointer");
        p.src[i] = src[i]; p.dst[i] = dst[i];
        if (take_exp && take_exp[i]) p.exp_mask |= 1u << i;
    }
    p.n = n;
    for (int f = 0; f < n_pending; ++f) {
        // (step_counter NULL: the pass has advanced its step count itself -- pcrl_grad_norm_partials_f32)
        if (!pending[f].partial || pending[f].n_partial < 1) return fail(PCRL_E_ARG, "bad pending optimizer pass %d", f);
        p.partial[f] = pending[f].partial; p.n_partial[f] = pending[f].n_partial;
        p.norm[f] = pending[f].grad_norm_out; p.step[f] = pending[f].step_counter;
    }
    p.n_fin = n_pending;
    p.host_out = host_out;
    return PCRL_OK;
}

// ---- backward of a row-wise LayerNorm over F <= 256 features (PointNet.final_mlp[1]), four rows per 256-thread block: dx and the per-block
// partial sums of dgamma / dbeta.  One block function, two hosts: layernorm_rows_bwd_kernel (csrc/dense.hip) and -- riding on the encoder
// backward's prep launch, which needs nothing it writes (pcrl_encoder_bwd_attach_ln_bwd) -- encoder_bwdg_prep_kernel. ------------------------
struct LnBwdParams {
    const float* dy0; const float* dy1; long long lddy;   // dy1 may be NULL
    const float* xhat; const float* rstd; const float* gamma;
    int M, F;
    float* dx; long long lddx;
    float* part;             // [blocks][2][F] partial sums of dy*xhat and dy
};
// s_acc: 4 * 2 * 256 floats of LDS; blk = which group of four rows; tid in [0, 256)
__device__ __forceinline__ void layernorm_rows_bwd_block(const LnBwdParams& p, int blk, int tid, float* s_acc) {
    const int wave = tid >> 6, lane = tid & 63;
    const int row = blk * 4 + wave;
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
    for (int j = 0; j < 4; ++j) { s_acc[(wave * 2 + 0) * 256 + lane + 64 * j] = dg[j]; s_acc[(wave * 2 + 1) * 256 + lane + 64 * j] = db[j]; }
    __syncthreads();
    const int f = tid;
    if (f < p.F) {
        p.part[((long long)blk * 2 + 0) * p.F + f] = (s_acc[(0 * 2 + 0) * 256 + f] + s_acc[(1 * 2 + 0) * 256 + f]) + (s_acc[(2 * 2 + 0) * 256 + f] + s_acc[(3 * 2 + 0) * 256 + f]);
        p.part[((long long)blk * 2 + 1) * p.F + f] = (s_acc[(0 * 2 + 1) * 256 + f] + s_acc[(1 * 2 + 1) * 256 + f]) + (s_acc[(2 * 2 + 1) * 256 + f] + s_acc[(3 * 2 + 1) * 256 + f]);
    }
}

// ---- fixed-order column reductions of per-workgroup partials (pcrl_colsum_jobs_f32; also riding on the encoder backward's reduce
// launch, pcrl_encoder_bwd_attach_colsum) ----------------------------------------------------------------------------------------
constexpr int kColsumJobs = 12;
struct ColsumJob { const float* part; long long blk_stride; int nblk, ncols; float* out; float scale; int op, blk_begin; };   // op 0 sum, 1 max
struct ColsumParams { ColsumJob job[kColsumJobs]; int n; };

// Block `blk` (of 256 columns) of the job list; tid in [0, 256).
__device__ __forceinline__ void colsum_block(const ColsumParams& p, int blk, int tid) {
    int ji = 0;
#pragma unroll
    for (int j = 1; j < kColsumJobs; ++j)
        if (j < p.n && blk >= p.job[j].blk_begin) ji = j;
    const ColsumJob& jb = p.job[ji];
    const int col = (blk - jb.blk_begin) * 256 + tid;
    if (col >= jb.ncols) return;
    const float* src = jb.part + col;
    float acc = jb.op ? -INFINITY : 0.0f;
    int b = 0;
    for (; b + 16 <= jb.nblk; b += 16) {           // sixteen loads in flight, combined in block order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(long long)(b + u) * jb.blk_stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = jb.op ? fmaxf(acc, v[u]) : acc + v[u];
    }
    for (; b < jb.nblk; ++b) { const float v = src[(long long)b * jb.blk_stride]; acc = jb.op ? fmaxf(acc, v) : acc + v; }
    jb.out[col] = acc * jb.scale;
}

// Host: the C-ABI job list -> kernel parameters; returns the number of 256-column blocks (0: nothing to do), -1 on a bad job.
inline int colsum_fill(const pcrl_colsum_job* jobs, int n, ColsumParams& p) {
    p = ColsumParams{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const pcrl_colsum_job& s = jobs[i];
        if (!s.part || !s.out || s.nblk < 0 || s.ncols < 0) return -1;
        if (s.ncols == 0) continue;
        p.job[p.n++] = ColsumJob{s.part, s.blk_stride, s.nblk, s.ncols, s.out, s.scale, s.op, blocks};
        blocks += (s.ncols + 255) / 256;
    }
    return blocks;
}

}  // namespace pcrl
