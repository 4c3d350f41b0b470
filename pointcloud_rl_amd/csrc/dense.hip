// Dense actor / critic heads for gfx950 (MI355X), fp32 on v_mfma_f32_32x32x2_f32.
//
// Replaces the reference's LinearMLP heads (pyrl/networks/backbones/mlp.py:97-100: Linear + ReLU
// stacks, actor D->1024->1024->2A, each Q head D+A->1024->1024->1) and their autograd backward,
// plus PointNet.final_mlp's Linear (pointnet.py:110).  One batched GEMM kernel with generic operand
// strides serves the three shapes of a Linear layer:
//     forward        Y  = act(X W^T + b)        A = X [M,K]      B[k][n] = W[n][k]
//     data gradient  dX = (dY W) (.) relu-mask  A = dY [M,N']    B[k][n] = W[k][n]
//     weight grad    dW = dY^T X, db = dY^T 1   A[m][k] = dY[k][m]  B = X (+ a virtual ones column)
// and up to four independent problems (dW and dX of one layer, the online and the target Q heads)
// share one launch, so that small ones overlap instead of paying a dispatch each.
//
// The batch is small (M = 256 rows): a 32x32 output tile is owned by one 8-wave workgroup that splits
// K eight ways and reduces through LDS in a fixed order.  With only 64 MFMAs per wave at K = 1024 the
// kernel lives or dies by its scalar overhead (measured: an earlier, more general version spent more
// wave-cycles on addressing and guards than on MFMAs), hence:
//   * operands are addressed as SGPR buffer resource + one 32-bit lane offset + scalar offset, rows
//     clamped into range instead of guarded (out-of-range tile rows compute garbage that is never stored);
//   * a k-contiguous operand is four 16-byte loads per lane per 32 k-steps (lane (i,h) holds
//     k = k0 + 16h + 4t + j of row i -- any k order works as long as A and B agree), an m/n-contiguous
//     one is sixteen coalesced 4-byte loads; two 32-k chunks are in flight per wave;
//   * only the ragged K tail (K % 32) takes a guarded path, where out-of-range k reads as zero through
//     the buffer bounds check;
//   * the epilogue issues its bias / mask loads before the split-K barrier and stores through the
//     buffer path with out-of-range rows / columns dropped by the bounds check.
#include "common.h"
#include <atomic>
#include <cstdlib>

namespace pcrl {

struct GemmParams {
    const float* A; const float* B; float* C;
    const float* bias;       // [N] added to every row (may be NULL)
    const float* mask;       // [M][ld_mask]: C = acc * (mask > 0) (ReLU backward; may be NULL)
    float* C_ones;           // optional separate destination of column `ones_col`
    long long a_bs, b_bs, c_bs, bias_bs, mask_bs, c_ones_bs;    // batch strides (elements)
    unsigned a_sm4, a_sk4, b_sn4, b_sk4, ldc4, ld_mask4;        // strides in bytes
    int M, N, K;
    int relu;                // C = max(acc + bias, 0)
    int ones_col;            // B[k][ones_col] == 1 for every k (bias gradient); -1: none
    int accumulate;          // C += result
    int a_k4, b_k4;          // operand is k-contiguous and 16-byte aligned
    int cfg;                 // 0: 32x32 tile per workgroup, K split over the 8 waves; 1: 64x64 tile staged through LDS; 2: a 32x32 tile per WAVE; 3: 64x32 per wave;
                             // 4: wave-private staged tiles (dense_wtile.h: gemm_wtile); 5: weight-gradient panels (gemm_wgrad_panel)
    int shape;               // cfg 4 / 5: which tile shape (kWTile* / kWPanel* below)
    int a_rc, b_rc;          // cfg 1: operand is contiguous along its row index (m resp. n) instead of along k
    int wg_n, wg_nm;         // n tiles, n tiles * m tiles
    float inv_wg_n, inv_wg_nm, inv_wg_m;
    int wg_begin;            // first workgroup of this problem inside the launch
    int tiles;               // workgroups of this problem (all batch elements)
};

constexpr int kGemmWaves = 8;
constexpr int kGemmGroup = 4;
constexpr unsigned kGemmOob = 0x80000000u;          // beyond every resource's num_records: loads give 0, stores are dropped
constexpr unsigned kGemmRecords = 0x7FFFFFFFu;
// wg_begin[] first: one scalar load finds the problem, a second clause fetches its whole descriptor
struct GemmGroup { int wg_begin[kGemmGroup]; int n; int _pad[3]; GemmParams p[kGemmGroup]; };

// 32 k-steps of one operand for lane (i,h): v[t][j] = X[row i][32 sc + 16 h + 4 t + j].  `off` is the lane's byte
// offset of (row i, k = 16 h).
__device__ __forceinline__ void gemm_load_full(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned sk4, bool k4, int sc, f32x4 (&v)[4]) {
    if (k4) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = buf_load_f4(rs, off + 16 * t, sc * 128);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[t][j] = buf_load_f1(rs, off, (unsigned)(32 * sc + 4 * t + j) * sk4);
    }
}
// The ragged last chunk: k >= K reads as zero (offset forced out of the resource's range).
__device__ __forceinline__ void gemm_load_tail(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned sk4, int kbase, int K, f32x4 (&v)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kbase + 4 * t + j;
            v[t][j] = buf_load_f1(rs, k < K ? off + (unsigned)(k - (kbase & 16)) * sk4 : kGemmOob, 0);
        }
}

// A wave-uniform pointer as the compiler can prove it (keeps buffer resources in SGPRs).
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* ptr) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

}  // namespace pcrl
#include "dense_wtile.h"
namespace pcrl {

__device__ __forceinline__ void gemm_mfma16(const f32x4 (&a)[4], const f32x4 (&b)[4], f32x16& acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[t][j], acc, 0, 0, 0);
}

// cfg 0: one 32x32 output tile per workgroup, K split over the 8 waves, operands straight from L2.
__device__ __forceinline__ void gemm_tile32_splitk(const GemmParams& p, const int wg, float* smem) {
    float (*s_red)[16][64] = reinterpret_cast<float (*)[16][64]>(smem);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, h = lane >> 5;
    // every field is needed within the next microsecond: one clause of scalar loads instead of dependent waits
    asm volatile("" ::"s"(p.A), "s"(p.B), "s"(p.C), "s"(p.bias), "s"(p.mask), "s"(p.C_ones), "s"(p.a_bs), "s"(p.b_bs), "s"(p.c_bs),
                 "s"(p.bias_bs), "s"(p.mask_bs), "s"(p.c_ones_bs));
    asm volatile("" ::"s"(p.a_sm4), "s"(p.a_sk4), "s"(p.b_sn4), "s"(p.b_sk4), "s"(p.ldc4), "s"(p.ld_mask4), "s"(p.M), "s"(p.N), "s"(p.K),
                 "s"(p.relu), "s"(p.ones_col), "s"(p.accumulate), "s"(p.a_k4), "s"(p.b_k4), "s"(p.wg_n), "s"(p.wg_nm), "s"(p.inv_wg_n),
                 "s"(p.inv_wg_nm), "s"(p.wg_begin));
    // tile decode without integer division (exact for < 2^20 workgroups; checked on the host)
    const int local = wg - p.wg_begin;
    const int bz = __builtin_amdgcn_readfirstlane((int)(((float)local + 0.5f) * p.inv_wg_nm));
    const int rem = local - bz * p.wg_nm;
    const int my = __builtin_amdgcn_readfirstlane((int)(((float)rem + 0.5f) * p.inv_wg_n));
    const int nx = rem - my * p.wg_n;
    const int m0 = my * 32, n0 = nx * 32;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(uniform_ptr(p.A + bz * p.a_bs), kGemmRecords);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(uniform_ptr(p.B + bz * p.b_bs), kGemmRecords);
    const int n = n0 + i;
    int n_read = min(n, p.N - 1);
    if (n_read == p.ones_col) n_read = 0;                       // that column is never read from memory
    const unsigned a_off = (unsigned)min(m0 + i, p.M - 1) * p.a_sm4 + 16u * h * p.a_sk4;
    const unsigned b_off = (unsigned)n_read * p.b_sn4 + 16u * h * p.b_sk4;
    const bool a4 = p.a_k4 != 0, b4 = p.b_k4 != 0, ones = p.ones_col >= 0, n_ones = n == p.ones_col;
    const int n_sc = (p.K + 31) >> 5, per = (n_sc + kGemmWaves - 1) / kGemmWaves;
    const int sc_begin = wave * per, sc_end = min(sc_begin + per, n_sc), full_end = min(sc_end, p.K >> 5);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    f32x4 a0[4], b0[4], a1[4], b1[4];
    auto load = [&](int sc, f32x4 (&a)[4], f32x4 (&b)[4]) {
        gemm_load_full(rs_a, a_off, p.a_sk4, a4, sc, a);
        gemm_load_full(rs_b, b_off, p.b_sk4, b4, sc, b);
        if (ones) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) b[t][j] = n_ones ? 1.0f : b[t][j];
        }
    };
    int sc = sc_begin;
    if (sc < full_end) load(sc, a0, b0);
    if (sc + 1 < full_end) load(sc + 1, a1, b1);
    while (sc < full_end) {             // two 32-k chunks in flight
        gemm_mfma16(a0, b0, acc);
        if (sc + 2 < full_end) load(sc + 2, a0, b0);
        if (++sc >= full_end) break;
        gemm_mfma16(a1, b1, acc);
        if (sc + 2 < full_end) load(sc + 2, a1, b1);
        ++sc;
    }
    if (sc < sc_end) {                  // the ragged last chunk (K % 32 != 0), owned by one k-slice
        const int kbase = 32 * sc + 16 * h;
        gemm_load_tail(rs_a, a_off, p.a_sk4, kbase, p.K, a0);
        gemm_load_tail(rs_b, b_off, p.b_sk4, kbase, p.K, b0);
        if (ones) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) b0[t][j] = n_ones ? (kbase + 4 * t + j < p.K ? 1.0f : 0.0f) : b0[t][j];
        }
        gemm_mfma16(a0, b0, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[r];

    // Epilogue: thread (wave, lane) owns column n0 + (lane & 31) of accumulator rows r = wave and wave + 8.
    const int col = n0 + (lane & 31);
    const bool col_ok = col < p.N;
    float bv = 0.0f, mv[2] = {1.0f, 1.0f}, old[2] = {0.0f, 0.0f};
    unsigned c_off[2];
    int row[2];
    const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(uniform_ptr(p.C + bz * p.c_bs), kGemmRecords);
    if (p.bias) bv = (p.bias + bz * p.bias_bs)[min(col, p.N - 1)];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = wave + 8 * e;
        row[e] = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        c_off[e] = (row[e] < p.M && col_ok) ? (unsigned)row[e] * p.ldc4 + 4u * col : kGemmOob;
    }
    if (p.mask) {
        const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(uniform_ptr(p.mask + bz * p.mask_bs), kGemmRecords);
#pragma unroll
        for (int e = 0; e < 2; ++e)
            mv[e] = buf_load_f1(rs_m, c_off[e] == kGemmOob ? kGemmOob : (unsigned)row[e] * p.ld_mask4 + 4u * col, 0);
    }
    if (p.accumulate) {
#pragma unroll
        for (int e = 0; e < 2; ++e) old[e] = buf_load_f1(rs_c, c_off[e], 0);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = wave + 8 * e;
        float v = ((s_red[0][r][lane] + s_red[1][r][lane]) + (s_red[2][r][lane] + s_red[3][r][lane])) +
                  ((s_red[4][r][lane] + s_red[5][r][lane]) + (s_red[6][r][lane] + s_red[7][r][lane]));
        v = v + bv;
        if (p.relu) v = v > 0.0f ? v : 0.0f;
        v = mv[e] > 0.0f ? v : 0.0f;
        v = old[e] + v;
        if (p.C_ones && col == p.ones_col) {
            if (c_off[e] != kGemmOob) (p.C_ones + bz * p.c_ones_bs)[row[e]] = v;
        } else {
            buf_store_f1(rs_c, c_off[e], 0, v);
        }
    }
}


// cfg 2: EIGHT 32x32 output tiles per workgroup, one per wave with the whole K loop -- no split-K, no LDS, no barrier.  For problems with
// thousands of tiles and a short contraction, i.e. the heads' weight gradients (1 024 x 1 025 outputs, K = batch): split eight ways a wave of
// cfg 0 had 16 MFMAs (K = 256) between its prologue and the LDS reduce, and the launch ran at 40-48 TFLOP/s.  The workgroup's waves form a
// 2 (m) x 4 (n) block of tiles: an A row block is shared by four waves, a B column block by two (L1).
__device__ __forceinline__ void gemm_tile32_wave(const GemmParams& p, const int wg) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, h = lane >> 5;
    // every field is needed within the next microsecond: one clause of scalar loads instead of dependent waits
    asm volatile("" ::"s"(p.A), "s"(p.B), "s"(p.C), "s"(p.bias), "s"(p.mask), "s"(p.C_ones), "s"(p.a_bs), "s"(p.b_bs), "s"(p.c_bs),
                 "s"(p.bias_bs), "s"(p.mask_bs), "s"(p.c_ones_bs));
    asm volatile("" ::"s"(p.a_sm4), "s"(p.a_sk4), "s"(p.b_sn4), "s"(p.b_sk4), "s"(p.ldc4), "s"(p.ld_mask4), "s"(p.M), "s"(p.N), "s"(p.K),
                 "s"(p.relu), "s"(p.ones_col), "s"(p.accumulate), "s"(p.a_k4), "s"(p.b_k4), "s"(p.wg_n), "s"(p.wg_nm), "s"(p.inv_wg_n),
                 "s"(p.inv_wg_nm), "s"(p.wg_begin));
    // tile decode without integer division (exact for < 2^20 workgroups; checked on the host)
    const int local = wg - p.wg_begin;
    const int bz = __builtin_amdgcn_readfirstlane((int)(((float)local + 0.5f) * p.inv_wg_nm));
    const int rem = local - bz * p.wg_nm;
    const int my = __builtin_amdgcn_readfirstlane((int)(((float)rem + 0.5f) * p.inv_wg_n));
    const int nx = rem - my * p.wg_n;
    const int m0 = (2 * my + (wave >> 2)) * 32, n0 = (4 * nx + (wave & 3)) * 32;
    if (m0 >= p.M || n0 >= p.N) return;              // (the whole wave: this path has no barrier)
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(uniform_ptr(p.A + bz * p.a_bs), kGemmRecords);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(uniform_ptr(p.B + bz * p.b_bs), kGemmRecords);
    const int n = n0 + i;
    int n_read = min(n, p.N - 1);
    if (n_read == p.ones_col) n_read = 0;                       // that column is never read from memory
    const unsigned a_off = (unsigned)min(m0 + i, p.M - 1) * p.a_sm4 + 16u * h * p.a_sk4;
    const unsigned b_off = (unsigned)n_read * p.b_sn4 + 16u * h * p.b_sk4;
    const bool a4 = p.a_k4 != 0, b4 = p.b_k4 != 0, ones = p.ones_col >= 0, n_ones = n == p.ones_col;
    const int n_sc = (p.K + 31) >> 5;
    const int sc_begin = 0, sc_end = n_sc, full_end = p.K >> 5;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    f32x4 a0[4], b0[4], a1[4], b1[4];
    auto load = [&](int sc, f32x4 (&a)[4], f32x4 (&b)[4]) {
        gemm_load_full(rs_a, a_off, p.a_sk4, a4, sc, a);
        gemm_load_full(rs_b, b_off, p.b_sk4, b4, sc, b);
        if (ones) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) b[t][j] = n_ones ? 1.0f : b[t][j];
        }
    };
    int sc = sc_begin;
    if (sc < full_end) load(sc, a0, b0);
    if (sc + 1 < full_end) load(sc + 1, a1, b1);
    while (sc < full_end) {             // two 32-k chunks in flight
        gemm_mfma16(a0, b0, acc);
        if (sc + 2 < full_end) load(sc + 2, a0, b0);
        if (++sc >= full_end) break;
        gemm_mfma16(a1, b1, acc);
        if (sc + 2 < full_end) load(sc + 2, a1, b1);
        ++sc;
    }
    if (sc < sc_end) {                  // the ragged last chunk (K % 32 != 0), owned by one k-slice
        const int kbase = 32 * sc + 16 * h;
        gemm_load_tail(rs_a, a_off, p.a_sk4, kbase, p.K, a0);
        gemm_load_tail(rs_b, b_off, p.b_sk4, kbase, p.K, b0);
        if (ones) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) b0[t][j] = n_ones ? (kbase + 4 * t + j < p.K ? 1.0f : 0.0f) : b0[t][j];
        }
        gemm_mfma16(a0, b0, acc);
    }
    // Epilogue: the lane owns column n0 + (lane & 31) of its sixteen accumulator rows.
    const int col = n0 + (lane & 31);
    const bool col_ok = col < p.N;
    const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(uniform_ptr(p.C + bz * p.c_bs), kGemmRecords);
    const float bv = p.bias ? (p.bias + bz * p.bias_bs)[min(col, p.N - 1)] : 0.0f;
    float mv[16], old[16];
    unsigned c_off[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        c_off[r] = (row < p.M && col_ok) ? (unsigned)row * p.ldc4 + 4u * col : kGemmOob;
        mv[r] = 1.0f; old[r] = 0.0f;
    }
    if (p.mask) {
        const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(uniform_ptr(p.mask + bz * p.mask_bs), kGemmRecords);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            mv[r] = buf_load_f1(rs_m, c_off[r] == kGemmOob ? kGemmOob : (unsigned)row * p.ld_mask4 + 4u * col, 0);
        }
    }
    if (p.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = buf_load_f1(rs_c, c_off[r], 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[r] + bv;
        if (p.relu) v = v > 0.0f ? v : 0.0f;
        v = mv[r] > 0.0f ? v : 0.0f;
        v = old[r] + v;
        if (p.C_ones && col == p.ones_col) {
            if (c_off[r] != kGemmOob) (p.C_ones + bz * p.c_ones_bs)[row] = v;
        } else {
            buf_store_f1(rs_c, c_off[r], 0, v);
        }
    }
}


// cfg 3: cfg 2 with TWO row blocks per wave when the A operand is contiguous along its rows (the weight gradient's dY^T): lane i loads rows
// 2 i and 2 i + 1 of a k-step with ONE 8-byte load, and the two values feed two MFMAs (one per accumulator; an accumulator's MFMA rows are the
// even resp. odd rows of the wave's 64 x 32 tile) against the same B value -- three load instructions per four MFMAs instead of eight for two
// row-contiguous operands, which is what bounds cfg 0 / cfg 2 on these problems.  k runs in half chunks of 16 (lane half h takes 8 h .. 8 h + 7),
// three of them in flight.  The workgroup's waves form a 2 (m) x 4 (n) block: 128 x 128 outputs.
__device__ __forceinline__ void gemm_tile64x32_wave(const GemmParams& p, const int wg) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, h = lane >> 5;
    asm volatile("" ::"s"(p.A), "s"(p.B), "s"(p.C), "s"(p.bias), "s"(p.mask), "s"(p.C_ones), "s"(p.a_bs), "s"(p.b_bs), "s"(p.c_bs),
                 "s"(p.bias_bs), "s"(p.mask_bs), "s"(p.c_ones_bs));
    asm volatile("" ::"s"(p.a_sm4), "s"(p.a_sk4), "s"(p.b_sn4), "s"(p.b_sk4), "s"(p.ldc4), "s"(p.ld_mask4), "s"(p.M), "s"(p.N), "s"(p.K),
                 "s"(p.relu), "s"(p.ones_col), "s"(p.accumulate), "s"(p.a_k4), "s"(p.b_k4), "s"(p.wg_n), "s"(p.wg_nm), "s"(p.inv_wg_n),
                 "s"(p.inv_wg_nm), "s"(p.wg_begin));
    const int local = wg - p.wg_begin;
    const int bz = __builtin_amdgcn_readfirstlane((int)(((float)local + 0.5f) * p.inv_wg_nm));
    const int rem = local - bz * p.wg_nm;
    const int my = __builtin_amdgcn_readfirstlane((int)(((float)rem + 0.5f) * p.inv_wg_n));
    const int nx = rem - my * p.wg_n;
    const int m0 = (2 * my + (wave >> 2)) * 64, n0 = (4 * nx + (wave & 3)) * 32;
    if (m0 >= p.M || n0 >= p.N) return;              // (the whole wave: this path has no barrier)
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(uniform_ptr(p.A + bz * p.a_bs), kGemmRecords);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(uniform_ptr(p.B + bz * p.b_bs), kGemmRecords);
    const int n = n0 + i;
    int n_read = min(n, p.N - 1);
    if (n_read == p.ones_col) n_read = 0;                       // that column is never read from memory
    // rows 2 i, 2 i + 1 of the tile (M is even on this path; a pair hanging over the end is clamped and never stored)
    const unsigned a_off = 4u * (unsigned)min(m0 + 2 * i, p.M - 2) + 8u * h * p.a_sk4;
    const unsigned b_off = (unsigned)n_read * p.b_sn4 + 8u * h * p.b_sk4;
    const bool b4 = p.b_k4 != 0, ones = p.ones_col >= 0, n_ones = n == p.ones_col;
    const int n_hc = (p.K + 15) >> 4, full_end = p.K >> 4;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    f32x2 a[3][8];
    float b[3][8];
    auto load = [&](int hc, f32x2 (&av)[8], float (&bv)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = buf_load_f2(rs_a, a_off, (unsigned)(16 * hc + u) * p.a_sk4);
        if (b4) {
            const f32x4 lo = buf_load_f4(rs_b, b_off, 64u * (unsigned)hc), hi = buf_load_f4(rs_b, b_off + 16u, 64u * (unsigned)hc);
#pragma unroll
            for (int u = 0; u < 4; ++u) { bv[u] = lo[u]; bv[4 + u] = hi[u]; }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) bv[u] = buf_load_f1(rs_b, b_off, (unsigned)(16 * hc + u) * p.b_sk4);
        }
        if (ones) {
#pragma unroll
            for (int u = 0; u < 8; ++u) bv[u] = n_ones ? 1.0f : bv[u];
        }
    };
    auto mfma8 = [&](const f32x2 (&av)[8], const float (&bv)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][0], bv[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][1], bv[u], acc1, 0, 0, 0);
        }
    };
    if (0 < full_end) load(0, a[0], b[0]);
    if (1 < full_end) load(1, a[1], b[1]);
    if (2 < full_end) load(2, a[2], b[2]);
    int hc = 0;
    while (hc < full_end) {             // three half chunks in flight
        mfma8(a[0], b[0]);
        if (hc + 3 < full_end) load(hc + 3, a[0], b[0]);
        if (++hc >= full_end) break;
        mfma8(a[1], b[1]);
        if (hc + 3 < full_end) load(hc + 3, a[1], b[1]);
        if (++hc >= full_end) break;
        mfma8(a[2], b[2]);
        if (hc + 3 < full_end) load(hc + 3, a[2], b[2]);
        ++hc;
    }
    if (hc < n_hc) {                    // the ragged last half chunk (K % 16 != 0): k >= K reads as zero
        const int kbase = 16 * hc + 8 * h;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = kbase + u < p.K;
            a[0][u] = buf_load_f2(rs_a, in ? a_off + (unsigned)(16 * hc + u) * p.a_sk4 : kGemmOob, 0);
            b[0][u] = buf_load_f1(rs_b, in ? b_off + (unsigned)(16 * hc + u) * p.b_sk4 : kGemmOob, 0);
            if (ones) b[0][u] = n_ones ? (in ? 1.0f : 0.0f) : b[0][u];
        }
        mfma8(a[0], b[0]);
    }
    // Epilogue: the lane owns column n0 + (lane & 31); accumulator e holds tile rows 2 rho + e, rho = the MFMA row of register r.
    const int col = n0 + (lane & 31);
    const bool col_ok = col < p.N;
    const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(uniform_ptr(p.C + bz * p.c_bs), kGemmRecords);
    const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(uniform_ptr((p.mask ? p.mask : p.C) + bz * p.mask_bs), kGemmRecords);
    const float bv = p.bias ? (p.bias + bz * p.bias_bs)[min(col, p.N - 1)] : 0.0f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float mv[16], old[16];
        unsigned c_off[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + e;
            c_off[r] = (row < p.M && col_ok) ? (unsigned)row * p.ldc4 + 4u * col : kGemmOob;
            mv[r] = p.mask ? buf_load_f1(rs_m, c_off[r] == kGemmOob ? kGemmOob : (unsigned)row * p.ld_mask4 + 4u * col, 0) : 1.0f;
            old[r] = p.accumulate ? buf_load_f1(rs_c, c_off[r], 0) : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + e;
            float v = (e ? acc1[r] : acc0[r]) + bv;
            if (p.relu) v = v > 0.0f ? v : 0.0f;
            v = mv[r] > 0.0f ? v : 0.0f;
            v = old[r] + v;
            if (p.C_ones && col == p.ones_col) {
                if (c_off[r] != kGemmOob) (p.C_ones + bz * p.c_ones_bs)[row] = v;
            } else {
                buf_store_f1(rs_c, c_off[r], 0, v);
            }
        }
    }
}

// cfg 1: one 64x64 output tile per workgroup, for the layers where all three dimensions are large (the 1024 x 1024 layers
// of the heads and their gradients).  Both operands are staged through LDS in 64-k chunks (two stages: the global loads of
// chunk c + 1 are in flight while chunk c is multiplied), so every operand element is fetched from L2 once per workgroup
// and used by two waves -- 16 FLOP per byte of L2 traffic instead of 8 -- with coalesced 16-byte loads whatever the
// operand's orientation.  Waves: (k half, 2 x 2 blocks of 32 x 32); the two k halves are added in LDS in a fixed order.
// An operand is "k-contiguous" (LDS image [row][k], 16-byte operand reads) or "row-contiguous" ([k][row], 4-byte reads).
constexpr int kT64Kc = 64;                       // k per stage: one barrier per 32 MFMAs of a wave
constexpr int kT64LdK = kT64Kc + 4;              // LDS pitch (floats) of a k-contiguous image [64 rows][k]: + 4 keeps 16-byte accesses conflict-free
constexpr int kT64LdR = 68;                      // ... of a row-contiguous image [k][64 rows]
constexpr int kT64Stage = kT64Kc * kT64LdR > 64 * kT64LdK ? kT64Kc * kT64LdR : 64 * kT64LdK;   // floats per operand per stage
constexpr int kT64Pieces = kT64Kc / 32;          // 16-byte pieces per thread, operand and chunk
constexpr size_t kGemmLdsBytes32 = sizeof(float) * kGemmWaves * 16 * 64;       // cfg 0: the split-K partial sums
constexpr size_t kGemmLdsBytes = sizeof(float) * 4 * kT64Stage;   // 2 operands x 2 stages = 139 264 B (>= the 32 KB of cfg 0)

struct T64Operand {
    __amdgpu_buffer_rsrc_t rs;    // batch element (buffer addressing: loads do not touch the LDS counter, out-of-range offsets read 0)
    unsigned voff[kT64Pieces];    // this thread's byte offsets of its 16-byte pieces inside chunk 0
    int lofs[kT64Pieces];         // where they go in the LDS image (floats)
    unsigned step;                // bytes from one chunk to the next
    int r0, rows_mem, rc, ones_row;
    bool edge;                    // the tile hangs over the operand's last row: row-contiguous pieces are read element-wise
    bool plain;                   // every piece of a chunk inside K is one unconditional 16-byte load
};

// Piece idx = tid + 512 u of a 64 x kT64Kc operand chunk: k-contiguous operand -> row idx / (Kc/4), k = 4 (idx % (Kc/4)) .. + 3;
// row-contiguous operand -> k = idx >> 4, rows 4 (idx & 15) .. + 3.
__device__ __forceinline__ void t64_piece(int rc, int idx, int& hi, int& lo4) {
    if (rc) { hi = idx >> 4; lo4 = 4 * (idx & 15); }
    else { hi = idx / (kT64Kc / 4); lo4 = 4 * (idx % (kT64Kc / 4)); }
}

__device__ __forceinline__ T64Operand t64_operand_setup(const float* base, unsigned s_row4, unsigned s_k4, int r0, int rows_mem, int rc,
                                                        int ones_row, int tid) {
    T64Operand o;
    o.rs = make_rsrc(base, kGemmRecords);
    o.r0 = r0; o.rows_mem = rows_mem; o.rc = rc; o.ones_row = ones_row;
    o.edge = r0 + 64 > rows_mem;
    o.plain = !(rc && o.edge) && !(ones_row >= r0 && ones_row < r0 + 64);
    o.step = rc ? (unsigned)kT64Kc * s_k4 : 4u * kT64Kc;
#pragma unroll
    for (int u = 0; u < kT64Pieces; ++u) {
        int hi, lo4;
        t64_piece(rc, tid + 512 * u, hi, lo4);
        // k-contiguous: rows past the end are clamped (they compute garbage that is never stored)
        o.voff[u] = rc ? (unsigned)hi * s_k4 + 4u * (unsigned)(r0 + lo4) : (unsigned)min(r0 + hi, rows_mem - 1) * s_row4 + 4u * (unsigned)lo4;
        o.lofs[u] = hi * (rc ? kT64LdR : kT64LdK) + lo4;
    }
    return o;
}

// Branches are workgroup-uniform; per-lane conditions only select offsets (kGemmOob reads 0).
// o.plain (no virtual ones row in this tile, no element-wise edge reads) and a chunk inside K: straight-line 16-byte loads.
__device__ __forceinline__ void t64_gload(const T64Operand& o, int c, int K, int tid, f32x4 (&v)[kT64Pieces]) {
    const int k0 = kT64Kc * c;
    const bool full = k0 + kT64Kc <= K;                                      // the chunk lies inside K: no k predicate
    const unsigned soff = (unsigned)c * o.step;
    if (full && o.plain) {
#pragma unroll
        for (int u = 0; u < kT64Pieces; ++u) v[u] = buf_load_f4(o.rs, o.voff[u], soff);
        return;
    }
#pragma unroll
    for (int u = 0; u < kT64Pieces; ++u) {
        int hi, lo4;
        t64_piece(o.rc, tid + 512 * u, hi, lo4);
        f32x4 x;
        if (!o.rc) {
            const int k = k0 + lo4;
            x = buf_load_f4(o.rs, (full || k < K) ? o.voff[u] : kGemmOob, soff);
            if (!full) {                                                    // ragged K: both operands read zero beyond it
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = k + e < K ? x[e] : 0.0f;
            }
            if (o.ones_row >= 0 && o.r0 + hi == o.ones_row) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = k + e < K ? 1.0f : 0.0f;
            }
        } else {
            const bool k_ok = full || k0 + hi < K;
            if (!o.edge) {
                x = buf_load_f4(o.rs, k_ok ? o.voff[u] : kGemmOob, soff);
            } else {
                const int row = o.r0 + lo4;
                const unsigned kbase = o.voff[u] - 4u * (unsigned)row;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    x[e] = buf_load_f1(o.rs, k_ok ? kbase + 4u * (unsigned)min(row + e, o.rows_mem - 1) : kGemmOob, soff);
                if (o.ones_row >= 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (row + e == o.ones_row) x[e] = k_ok ? 1.0f : 0.0f;
                }
            }
        }
        v[u] = x;
    }
}

__device__ __forceinline__ void t64_lds_store(float* s, const T64Operand& o, const f32x4 (&v)[kT64Pieces]) {
#pragma unroll
    for (int u = 0; u < kT64Pieces; ++u) *reinterpret_cast<f32x4*>(s + o.lofs[u]) = v[u];
}

// Operand values of the 4 MFMAs of a group for lane (i, h): k = kq + j, j = 0..3, of row `row` (kq already holds + 4 h).
template <bool RC>
__device__ __forceinline__ f32x4 t64_operand(const float* s, int row, int kq) {
    if (!RC) return *reinterpret_cast<const f32x4*>(s + row * kT64LdK + kq);
    f32x4 x;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = s[(kq + j) * kT64LdR + row];
    return x;
}

// Main loop of a 64x64 tile, one instantiation per pair of operand orientations (straight-line LDS reads and MFMAs).
template <bool ARC, bool BRC>
__device__ __forceinline__ void t64_mainloop(const T64Operand& oa, const T64Operand& ob, int K, float* smem, int tid, int arow, int i, int kb,
                                             f32x16& acc0, f32x16& acc1) {
    float* sA = smem;
    float* sB = smem + 2 * kT64Stage;
    const int n_chunks = (K + kT64Kc - 1) / kT64Kc;
    // Register ring two chunks deep: chunk c + 2 is requested before chunk c is multiplied and lands in LDS one iteration
    // later, so a load has two chunk times to come back from L2 / MALL.
    f32x4 ra[2][kT64Pieces], rb[2][kT64Pieces];
    auto compute = [&](int st) {
        const float* a_s = sA + st * kT64Stage;
        const float* b_s = sB + st * kT64Stage;
        constexpr int NQ = kT64Kc / 32;                            // groups of 8 k for this wave's quarter of the chunk
        f32x4 a[NQ], b0[NQ], b1[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            a[q] = t64_operand<ARC>(a_s, arow, kb + 8 * q);
            b0[q] = t64_operand<BRC>(b_s, i, kb + 8 * q);
            b1[q] = t64_operand<BRC>(b_s, 32 + i, kb + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], b0[q][j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], b1[q][j], acc1, 0, 0, 0);
            }
    };
    t64_gload(oa, 0, K, tid, ra[0]);
    t64_gload(ob, 0, K, tid, rb[0]);
    if (n_chunks > 1) {
        t64_gload(oa, 1, K, tid, ra[1]);
        t64_gload(ob, 1, K, tid, rb[1]);
    }
    t64_lds_store(sA, oa, ra[0]);
    t64_lds_store(sB, ob, rb[0]);
    __syncthreads();
    for (int c = 0; c < n_chunks; c += 2) {
        // even chunk c in stage 0; ring slot 1 holds chunk c + 1, slot 0 is free for chunk c + 2
        if (c + 2 < n_chunks) {
            t64_gload(oa, c + 2, K, tid, ra[0]);
            t64_gload(ob, c + 2, K, tid, rb[0]);
        }
        compute(0);
        if (c + 1 < n_chunks) {
            t64_lds_store(sA + kT64Stage, oa, ra[1]);
            t64_lds_store(sB + kT64Stage, ob, rb[1]);
        }
        __syncthreads();
        if (c + 1 >= n_chunks) break;
        // odd chunk c + 1 in stage 1; slot 0 holds chunk c + 2, slot 1 is free for chunk c + 3
        if (c + 3 < n_chunks) {
            t64_gload(oa, c + 3, K, tid, ra[1]);
            t64_gload(ob, c + 3, K, tid, rb[1]);
        }
        compute(1);
        if (c + 2 < n_chunks) {
            t64_lds_store(sA, oa, ra[0]);
            t64_lds_store(sB, ob, rb[0]);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void gemm_tile64(const GemmParams& p, const int wg, float* smem) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, h = lane >> 5;
    // wave = (k quarter kg, row half bm): a 32 x 64 strip = two accumulators that share the A operand.  Two independent
    // accumulators matter: an instruction slipped between two MFMAs on the SAME accumulator costs ~43 idle matrix cycles,
    // between MFMAs on different ones ~6 (MI355X_MICROARCH.md), and the A operand is read from LDS once for both.
    const int kg = wave >> 1, bm = wave & 1;
    int local = wg - p.wg_begin;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: XCD x takes a contiguous eighth of the tile
    // list (batch, N tile, M tile with M fastest), i.e. a few N tiles with all their M tiles, so a slice of the large B
    // operand is fetched into one L2 only and shared there by the M tiles that use it.
    if ((p.tiles & 7) == 0) local = (local & 7) * (p.tiles >> 3) + (local >> 3);
    const int bz = __builtin_amdgcn_readfirstlane((int)(((float)local + 0.5f) * p.inv_wg_nm));
    const int rem = local - bz * p.wg_nm;
    const int wg_m = p.wg_nm / p.wg_n;
    const int nx = __builtin_amdgcn_readfirstlane((int)(((float)rem + 0.5f) * p.inv_wg_m));
    const int my = rem - nx * wg_m;
    const int m0 = my * 64, n0 = nx * 64;
    const T64Operand oa = t64_operand_setup(uniform_ptr(p.A + bz * p.a_bs), p.a_sm4, p.a_sk4, m0, p.M, p.a_rc, -1, tid);
    const T64Operand ob = t64_operand_setup(uniform_ptr(p.B + bz * p.b_bs), p.b_sn4, p.b_sk4, n0, p.ones_col >= 0 ? p.ones_col : p.N, p.b_rc,
                                            p.ones_col, tid);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
    const int arow = 32 * bm + i, kb = (kT64Kc / 4) * kg + 4 * h;
    if (!p.a_rc && !p.b_rc) t64_mainloop<false, false>(oa, ob, p.K, smem, tid, arow, i, kb, acc0, acc1);        // forward
    else if (!p.a_rc && p.b_rc) t64_mainloop<false, true>(oa, ob, p.K, smem, tid, arow, i, kb, acc0, acc1);     // data gradient
    else if (p.a_rc && p.b_rc) t64_mainloop<true, true>(oa, ob, p.K, smem, tid, arow, i, kb, acc0, acc1);       // weight gradient
    else t64_mainloop<true, false>(oa, ob, p.K, smem, tid, arow, i, kb, acc0, acc1);
    // ---- the four k quarters meet in LDS (the stages are free: every wave is past the last barrier) -----------------
    float (*s_red)[32][64] = reinterpret_cast<float (*)[32][64]>(smem);         // [wave][accumulator register][lane], 64 KB
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_red[wave][r][lane] = acc0[r]; s_red[wave][16 + r][lane] = acc1[r]; }
    // wave (kg, bm) finishes accumulator registers 8 kg .. 8 kg + 7 (of 32) of row half bm: block column kg >> 1
    const int bn = kg >> 1;
    const int col = n0 + 32 * bn + i;
    const bool col_ok = col < p.N;
    float bv = 0.0f, mv[8], old[8];
    unsigned c_off[8];
    int row[8];
    const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(uniform_ptr(p.C + bz * p.c_bs), kGemmRecords);
    if (p.bias) bv = (p.bias + bz * p.bias_bs)[min(col, p.N - 1)];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = (8 * kg + e) & 15;
        row[e] = m0 + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * h;
        c_off[e] = (row[e] < p.M && col_ok) ? (unsigned)row[e] * p.ldc4 + 4u * col : kGemmOob;
        mv[e] = 1.0f; old[e] = 0.0f;
    }
    if (p.mask) {
        const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(uniform_ptr(p.mask + bz * p.mask_bs), kGemmRecords);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            mv[e] = buf_load_f1(rs_m, c_off[e] == kGemmOob ? kGemmOob : (unsigned)row[e] * p.ld_mask4 + 4u * col, 0);
    }
    if (p.accumulate) {
#pragma unroll
        for (int e = 0; e < 8; ++e) old[e] = buf_load_f1(rs_c, c_off[e], 0);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int R = 8 * kg + e;
        float v = (s_red[bm][R][lane] + s_red[2 + bm][R][lane]) + (s_red[4 + bm][R][lane] + s_red[6 + bm][R][lane]);
        v = v + bv;
        if (p.relu) v = v > 0.0f ? v : 0.0f;
        v = mv[e] > 0.0f ? v : 0.0f;
        v = old[e] + v;
        if (p.C_ones && col == p.ones_col) {
            if (c_off[e] != kGemmOob) (p.C_ones + bz * p.c_ones_bs)[row[e]] = v;
        } else {
            buf_store_f1(rs_c, c_off[e], 0, v);
        }
    }
}

// cfg 4 tile shapes: E x (BM, BN) blocks; a row-contiguous B (data gradient) takes the transposed 2-block shape and has no 32 x 64 one
__host__ __device__ constexpr int wtile_shapes(int brc) { return brc ? 3 : 4; }
__host__ __device__ constexpr int wtile_tm(int shape, int brc) { return shape == 0 ? 16 : shape == 1 ? (brc ? 32 : 16) : 32; }
__host__ __device__ constexpr int wtile_tn(int shape, int brc) { return shape == 0 ? 16 : shape == 1 ? (brc ? 16 : 32) : shape == 2 ? 32 : 64; }
// cfg 5 panel shapes: 0 = 64 x 128 (no split-K), 1 = 64 x 64 (two k halves)
__host__ __device__ constexpr int wpanel_tn(int shape) { return shape == 0 ? 128 : 64; }

// One kernel per set of tile-path FAMILIES (round 6).  Rounds 1-5 ran every launch through one of two kernels that held several paths
// behind a run-time switch; a launch now takes the kernel instantiated for exactly the families its problems use (the table below), so a
// family's code is compiled, register-allocated and fetched on its own.  -Rpass-analysis=kernel-resource-usage on this file:
//   legacy : cfg 0 (32 x 32 split-K), cfg 2 / 3 (a tile per wave)            114 registers, 0 spilled
//   wt     : cfg 4, B k-contiguous (forward): 16 x 16 ... 32 x 64            124, 0 spilled
//   wtx    : cfg 4, B contiguous along n (data gradient)                      96, 0 spilled
//   panel  : cfg 5 (weight-gradient panels)                                   68, 0 spilled
//   t64    : cfg 1 (64 x 64 tiles staged through 68 KB of LDS)               128, 51 spilled -- kept ON PURPOSE: 68 KB of LDS lets TWO
//            workgroups share a CU, which needs <= 128 registers; the spill-free 162-register build runs one workgroup per CU and measured
//            13 % slower where the path matters (four heads at 1 024 rows: 81.3 -> 92.2 us, tools/r6_gemm_exp.sh, same box).  The scratch
//            traffic sits in the data- / weight-gradient orientations (55 / 26 scratch instructions per 32 MFMAs of their k loops); the
//            forward orientation's loop has 3.
// Every family is compiled for 4 waves per SIMD (128 registers) -- see amdgpu_waves_per_eu below.
constexpr unsigned kFamLegacy = 1u, kFamT64 = 2u, kFamWt = 4u, kFamWtx = 8u, kFamPanel = 16u, kFamAll = 31u;
__host__ __device__ constexpr unsigned gemm_family(int cfg, int b_rc) {
    return cfg == 1 ? kFamT64 : cfg == 4 ? (b_rc ? kFamWtx : kFamWt) : cfg == 5 ? kFamPanel : kFamLegacy;
}

__device__ __forceinline__ int gemm_find_problem(const GemmGroup& g, int wg) {
    int gi = 0;
#pragma unroll
    for (int j = 1; j < kGemmGroup; ++j)
        if (j < g.n && wg >= g.wg_begin[j]) gi = j;
    return gi;
}

// W = waves per SIMD the register allocation is made for (4: 128 registers, 2: 256; every instantiated variant uses 4).  A 32 x 64
// data-gradient shape (wtx shape 3: 166 registers) exists only where W == 2 and is not planned for.
// amdgpu_waves_per_eu(W, W) -- minimum AND maximum: with only the minimum (`__launch_bounds__(512, 4)`) the scheduler of a kernel that needs
// 96 registers aims at the NEXT occupancy step and shortens its load rings for it (measured: the grouped dW | dX launch 27.4 -> 30.2 us);
// the union kernels of round 5 were pinned at 128 registers by their hungriest path and never showed it.
template <unsigned FAM, int W>
__global__ __launch_bounds__(64 * kGemmWaves) __attribute__((amdgpu_waves_per_eu(W, W))) void gemm_fam_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
    const int wg = blockIdx.x;
    const GemmParams p = g.p[gemm_find_problem(g, wg)];
    constexpr bool single = (FAM & (FAM - 1)) == 0;          // one family: no dispatch on the problem's cfg
    if constexpr ((FAM & kFamWt) != 0) {
        if (single || (p.cfg == 4 && !p.b_rc)) {
            if (p.shape == 0) gemm_wtile<16, 1, 1, 32, false, 2>(p, wg, gemm_smem);
            else if (p.shape == 1) gemm_wtile<16, 1, 2, 32, false, 2>(p, wg, gemm_smem);
            else if (p.shape == 2) gemm_wtile<32, 1, 1, 16, false, 2>(p, wg, gemm_smem);
            else gemm_wtile<32, 1, 2, 16, false, W == 2 ? 2 : 1>(p, wg, gemm_smem);
            return;
        }
    }
    if constexpr ((FAM & kFamWtx) != 0) {
        if (single || (p.cfg == 4 && p.b_rc)) {
            if (p.shape == 0) gemm_wtile<16, 1, 1, 32, true, 2>(p, wg, gemm_smem);
            else if (p.shape == 1) gemm_wtile<16, 2, 1, 32, true, 2>(p, wg, gemm_smem);
            else if (W != 2 || p.shape == 2) gemm_wtile<32, 1, 1, 16, true, 2>(p, wg, gemm_smem);
            else if constexpr (W == 2) gemm_wtile<32, 1, 2, 16, true, 2>(p, wg, gemm_smem);
            return;
        }
    }
    if constexpr ((FAM & kFamPanel) != 0) {
        if (single || p.cfg == 5) {
            if (p.shape == 0) gemm_wgrad_panel<2, 4, 32>(p, wg, gemm_smem);
            else gemm_wgrad_panel<2, 2, 64>(p, wg, gemm_smem);
            return;
        }
    }
    if constexpr ((FAM & kFamT64) != 0) {
        if (single || p.cfg == 1) { gemm_tile64(p, wg, gemm_smem); return; }
    }
    if constexpr ((FAM & kFamLegacy) != 0) {
        if (p.cfg == 2) gemm_tile32_wave(p, wg);
        else if (p.cfg == 3) gemm_tile64x32_wave(p, wg);
        else gemm_tile32_splitk(p, wg, gemm_smem);
    }
}

// The instantiated variants, most specific first: a launch takes the first one whose family set covers the launch's.
struct GemmVariant { unsigned fam; int waves; const void* fn; const char* name; };
#define PCRL_GEMM_VARIANT(F, W) GemmVariant{F, W, reinterpret_cast<const void*>(&gemm_fam_kernel<F, W>), "gemm_fam_kernel<" #F "," #W ">"}
static const GemmVariant kGemmVariants[] = {
    PCRL_GEMM_VARIANT(kFamLegacy, 4),
    PCRL_GEMM_VARIANT(kFamT64, 4),
    PCRL_GEMM_VARIANT(kFamWt, 4),
    PCRL_GEMM_VARIANT(kFamWtx, 4),
    PCRL_GEMM_VARIANT(kFamPanel, 4),
    PCRL_GEMM_VARIANT(kFamPanel | kFamWtx, 4),                  // dW | dX of a 1 024-wide layer
    PCRL_GEMM_VARIANT(kFamLegacy | kFamWtx, 4),                 // dW0 | dX0 of the first layer
    PCRL_GEMM_VARIANT(kFamLegacy | kFamPanel | kFamWtx, 4),     // the policy's dW2 next to [dW1, dh1]
    PCRL_GEMM_VARIANT(kFamLegacy | kFamWt, 4),                  // a first layer sharing the launch of a 1 024-wide one
    PCRL_GEMM_VARIANT(kFamPanel | kFamT64, 4),                  // dW | dX at 1 024 rows and more
    PCRL_GEMM_VARIANT(kFamLegacy | kFamT64, 4),
    PCRL_GEMM_VARIANT(kFamAll, 4),                              // anything else
};
#undef PCRL_GEMM_VARIANT
static const GemmVariant& gemm_variant(unsigned need) {
    for (const GemmVariant& v : kGemmVariants)
        if ((v.fam & need) == need) return v;
    return kGemmVariants[sizeof(kGemmVariants) / sizeof(kGemmVariants[0]) - 1];
}

static size_t gemm_lds_bytes(const GemmParams& p) {
    if (p.cfg == 4) {
        if (!p.b_rc) return p.shape == 0 ? WTile<16, 1, 1, 32, false>::lds_bytes() : p.shape == 1 ? WTile<16, 1, 2, 32, false>::lds_bytes()
                          : p.shape == 2 ? WTile<32, 1, 1, 16, false>::lds_bytes() : WTile<32, 1, 2, 16, false>::lds_bytes();
        return p.shape == 0 ? WTile<16, 1, 1, 32, true>::lds_bytes() : p.shape == 1 ? WTile<16, 2, 1, 32, true>::lds_bytes()
                            : p.shape == 2 ? WTile<32, 1, 1, 16, true>::lds_bytes() : WTile<32, 1, 2, 16, true>::lds_bytes();
    }
    if (p.cfg == 5) return p.shape == 0 ? WPanel<2, 4, 32>::lds_bytes() : WPanel<2, 2, 64>::lds_bytes();
    if (p.cfg == 1) return kGemmLdsBytes;
    return p.cfg == 0 ? kGemmLdsBytes32 : 0;
}

// Row-wise LayerNorm over a short feature vector (PointNet.final_mlp[1]: nn.LayerNorm(out), eps 1e-5,
// pointnet.py:110), one wave per row; the output may be scattered into several destination
// buffers (the concatenated inputs of the actor and Q heads).
struct LnJob {
    const float* x; long long ldx;       // [M][F]
    int M, n_dst, blk_begin;
    float* y[4]; long long ldy[4];
    float* xhat; float* rstd;            // saved for backward (may be NULL)
    // optional pass-through columns (robot state / replay actions of Visuomotor's torch.cat, visuomotor.py:130-141):
    // cat_dst[m][0..cat_n) = cat_src[m][0..cat_n)
    const float* cat_src[2]; float* cat_dst[2]; long long cat_lds[2], cat_ldd[2]; int cat_n[2], cat_div[2];
};
constexpr int kLnJobs = 3;
struct LnParams {
    const float* gamma; const float* beta; int F; float eps; int n_jobs;
    LnJob job[kLnJobs];
};

__global__ __launch_bounds__(256) void layernorm_rows_fwd_kernel(const LnParams p) {
    int ji = 0;
#pragma unroll
    for (int j = 1; j < kLnJobs; ++j)
        if (j < p.n_jobs && (int)blockIdx.x >= p.job[j].blk_begin) ji = j;
    const LnJob& jb = p.job[ji];
    const int row = ((int)blockIdx.x - jb.blk_begin) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= jb.M) return;
    const float* x = jb.x + (long long)row * jb.ldx;
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
    if (jb.rstd && lane == 0) jb.rstd[row] = rstd;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = lane + 64 * j;
        if (f < p.F) {
            const float xh = (v[j] - mean) * rstd;
            if (jb.xhat) jb.xhat[(long long)row * p.F + f] = xh;
            const float y = xh * p.gamma[f] + p.beta[f];
            for (int d = 0; d < jb.n_dst; ++d) jb.y[d][(long long)row * jb.ldy[d] + f] = y;
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
        if (jb.cat_src[c])
            for (int f = lane; f < jb.cat_n[c]; f += 64)
                jb.cat_dst[c][(long long)row * jb.cat_ldd[c] + f] = jb.cat_src[c][(long long)(row / jb.cat_div[c]) * jb.cat_lds[c] + f];
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)), dxhat = dy * gamma, where dy is the
// sum of up to two upstream gradients (one per Q head); per-block partial dgamma / dbeta.
__global__ __launch_bounds__(256) void layernorm_rows_bwd_kernel(const LnBwdParams p) {
    __shared__ float s_acc[4 * 2 * 256];
    layernorm_rows_bwd_block(p, (int)blockIdx.x, (int)threadIdx.x, s_acc);
}

__global__ void colsum_partials_kernel(const float* part, int nblk, int n, float* out0, float* out1, int half, int accumulate) {
    // out0[f] = sum_b part[b][0][f], out1[f] = sum_b part[b][1][f]  (fixed order), n = F
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    float a = 0.0f, c = 0.0f;
    int b = 0;
    for (; b + 16 <= nblk; b += 16) {          // 32 independent loads in flight, summed in block order
        float va[16], vc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { va[u] = part[((long long)(b + u) * 2 + 0) * n + f]; vc[u] = part[((long long)(b + u) * 2 + 1) * n + f]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { a += va[u]; c += vc[u]; }
    }
    for (; b < nblk; ++b) { a += part[((long long)b * 2 + 0) * n + f]; c += part[((long long)b * 2 + 1) * n + f]; }
    out0[f] = accumulate ? out0[f] + a : a;
    out1[f] = accumulate ? out1[f] + c : c;
}

}  // namespace pcrl

using namespace pcrl;

// a tile per wave (cfg 2 / 3, the legacy weight-gradient path): from this many 32 x 32 tiles, up to this K
constexpr long long kWaveTileMinTiles = 1536;
constexpr int kWaveTileMaxK = 1024;

// How a launch picks its tile paths (pcrl_gemm_set_tile64_min):
//   auto (default)   the wave-private staged tiles / weight-gradient panels of dense_wtile.h wherever an operand layout allows them, the
//                    LDS-staged 64 x 64 tiles for the largest problems, the 32 x 32 split-K tiles for the rest (short K, odd strides);
//   min_tiles <= 1   the 64 x 64 tiles forced for every problem they can compute (tests: every orientation);
//   min_tiles >= 2^30  the paths of rounds 1-4 only: 32 x 32 split-K tiles and a tile per wave (tests keep them covered: they serve every
//                    shape the newer paths decline).
static std::atomic<int> g_tile64_min{192};
static int tile64_min_tiles() { return g_tile64_min.load(std::memory_order_relaxed); }
static bool legacy_only() { return tile64_min_tiles() >= (1 << 30); }

extern "C" int pcrl_gemm_set_tile64_min(int32_t min_tiles) {
    const int prev = tile64_min_tiles();
    if (min_tiles >= 0) g_tile64_min.store(min_tiles, std::memory_order_relaxed);
    return prev;
}

struct GemmPlan {
    bool kind_f, kind_x, kind_w;     // forward-shaped / data-gradient-shaped / weight-gradient-shaped for the dense_wtile.h paths
    bool t64_ok;                     // computable by the 64 x 64 staged tiles
    bool t64_pays;                   // ... and the rounds-2/3 measurements say they pay (given enough tiles in the launch)
    bool a_kc, b_kc;
};

static int gemm_fill_common(const pcrl_gemm_desc* d, GemmParams& p, GemmPlan& pl) {
    if (!d->A || !d->B || !d->C) return fail(PCRL_E_ARG, "NULL argument");
    if (d->M < 0 || d->N < 0 || d->K < 0 || d->batch < 1) return fail(PCRL_E_ARG, "bad GEMM shape");
    p = GemmParams{};
    p.A = d->A; p.B = d->B; p.C = d->C; p.bias = d->bias; p.mask = d->mask; p.C_ones = d->C_ones;
    p.a_bs = d->a_batch_stride; p.b_bs = d->b_batch_stride; p.c_bs = d->c_batch_stride;
    p.bias_bs = d->bias_batch_stride; p.mask_bs = d->mask_batch_stride; p.c_ones_bs = d->c_ones_batch_stride;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.relu = d->relu; p.ones_col = d->ones_col; p.accumulate = d->accumulate;
    // 32-bit byte offsets inside one batch element (buffer addressing)
    const int64_t lim = 0x7FFFFFFF / 4;
    auto span = [](int64_t rows, int64_t rs, int64_t cols, int64_t cs) {
        return (rows > 0 ? (rows - 1) * rs : 0) + (cols > 0 ? (cols - 1) * cs : 0);
    };
    if (d->a_stride_m < 0 || d->a_stride_k < 0 || d->b_stride_k < 0 || d->b_stride_n < 0 || d->ldc < 0 || d->ld_mask < 0 ||
        span(d->M, d->a_stride_m, d->K, d->a_stride_k) >= lim || span(d->N, d->b_stride_n, d->K, d->b_stride_k) >= lim ||
        span(d->M, d->ldc, d->N, 1) >= lim || span(d->M, d->ld_mask, d->N, 1) >= lim)
        return fail(PCRL_E_ARG, "GEMM operand strides must be non-negative and span < 2 GiB per batch element");
    p.a_sm4 = 4u * (unsigned)d->a_stride_m; p.a_sk4 = 4u * (unsigned)d->a_stride_k;
    p.b_sn4 = 4u * (unsigned)d->b_stride_n; p.b_sk4 = 4u * (unsigned)d->b_stride_k;
    p.ldc4 = 4u * (unsigned)d->ldc; p.ld_mask4 = 4u * (unsigned)d->ld_mask;
    auto aligned = [](const float* ptr, long long sm, long long bs) {
        return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && sm % 4 == 0 && bs % 4 == 0;
    };
    p.a_k4 = d->a_stride_k == 1 && aligned(p.A, d->a_stride_m, p.a_bs);
    p.b_k4 = d->b_stride_k == 1 && aligned(p.B, d->b_stride_n, p.b_bs);
    // 64x64 LDS-staged tiles: all three dimensions large, both operands readable in 16-byte pieces in one of the two orientations
    pl.a_kc = p.a_k4 && d->a_stride_m >= d->K;
    pl.b_kc = p.b_k4 && d->b_stride_n >= d->K;
    const bool a_rc = d->a_stride_m == 1 && aligned(p.A, d->a_stride_k, p.a_bs);
    const bool b_rc = d->b_stride_n == 1 && aligned(p.B, d->b_stride_k, p.b_bs);
    const bool big = d->M >= 48 && d->N >= 48 && d->K >= 64;
    pl.t64_ok = big && (pl.a_kc || a_rc) && (pl.b_kc || b_rc);
    // Measured (tools/bench_gemm.py, rounds 2-3): the staged 64 x 64 tiles win where K is long and both operands are k-contiguous, and for the
    // data gradient (dY k-contiguous, W row-contiguous) once the batch gives it several hundred tiles of its own.
    const long long own_tiles64 = (long long)((d->M + 63) / 64) * ((d->N + 63) / 64) * d->batch;
    pl.t64_pays = (pl.a_kc && pl.b_kc && d->K >= 512) || (pl.a_kc && b_rc && d->K >= 512 && own_tiles64 >= 384);
    p.a_rc = !pl.a_kc; p.b_rc = !pl.b_kc;
    // dense_wtile.h: A staged in 16-byte pieces along k (K % 4 == 0: a piece lies inside K or outside), a contraction long enough to give
    // the eight k slices something to do, no column of ones
    // (K >= 512: the 1 024-wide layers.  A short contraction -- K3 / K2's first layers, K = 196 -- pays the staging prologue for two chunks
    // and measured slower than the split-K tiles inside the step, tools/r5_ab2.sh)
    const bool a_stage = pl.a_kc && d->K % 4 == 0 && d->K >= 512 && d->ones_col < 0 && d->M > 0 && d->N > 0;
    pl.kind_f = a_stage && pl.b_kc;
    pl.kind_x = a_stage && !pl.b_kc && d->b_stride_n == 1;
    // ... weight gradient: both operands contiguous along their row index and readable in 16-byte pieces of four rows, the ones column last
    const int n_real = d->ones_col >= 0 ? d->ones_col : d->N;
    // ... and an output large enough to give the chip a few dozen 64 x 64 panels (the heads' 1 024-wide layers; the feature head's
    // 128 x 256 weight gradient would be 8 workgroups)
    pl.kind_w = a_rc && b_rc && d->M % 4 == 0 && n_real % 4 == 0 && d->M >= 256 && n_real >= 256 && d->K >= 1 &&
                (d->ones_col < 0 || d->ones_col == d->N - 1);
    return PCRL_OK;
}

static void gemm_set_tiles(GemmParams& p, int tm, int tn, int n_cols, int batch, int& wg_total) {
    p.wg_n = (n_cols + tn - 1) / tn;
    const int wg_m = (p.M + tm - 1) / tm;
    p.wg_nm = p.wg_n * wg_m;
    p.inv_wg_n = 1.0f / (float)(p.wg_n > 0 ? p.wg_n : 1);
    p.inv_wg_nm = 1.0f / (float)(p.wg_nm > 0 ? p.wg_nm : 1);
    p.inv_wg_m = 1.0f / (float)(wg_m > 0 ? wg_m : 1);
    p.wg_begin = wg_total;
    p.tiles = p.wg_nm * batch;
    wg_total += p.tiles;
}

// The legacy choice between cfg 0 / 2 / 3 for a problem that is not a 64 x 64 one
static void gemm_pick_legacy(const pcrl_gemm_desc* d, GemmParams& p, int& tm, int& tn) {
    p.cfg = 0;
    const long long tiles32 = (long long)((d->M + 31) / 32) * ((d->N + 31) / 32) * d->batch;
    if (tiles32 >= kWaveTileMinTiles && d->K <= kWaveTileMaxK) {
        // two row blocks per wave from 8-byte loads where A is contiguous along its rows and pairs of rows never straddle anything
        const bool pairs = d->a_stride_m == 1 && d->M % 2 == 0 && d->M >= 2 && reinterpret_cast<uintptr_t>(p.A) % 8 == 0 &&
                           d->a_stride_k % 2 == 0 && p.a_bs % 2 == 0;
        p.cfg = pairs ? 3 : 2;
    }
    tn = p.cfg >= 2 ? 128 : 32;
    tm = p.cfg == 3 ? 128 : p.cfg == 2 ? 64 : 32;               // a workgroup's block of the output
}

// Tile paths, workgroup ranges and the LDS size of one launch; src[i] = the caller's index of the launch's i-th problem.
static int gemm_plan(const pcrl_gemm_desc* descs, int32_t n, GemmGroup& g, size_t& lds, int& wg_total_out, int (&src)[kGemmGroup], unsigned& families) {
    if (!descs || n < 1 || n > kGemmGroup) return fail(PCRL_E_ARG, "pcrl_gemm_group_f32: 1 <= n <= %d problems", kGemmGroup);
    // Workgroups are dispatched in index order: a problem with a long K loop and few tiles (the data gradient of a small batch:
    // K = 1 024, <= 256 tiles) goes first, so that its few long workgroups start at once and the many short ones of its
    // partner (the weight gradient, K = batch) fill in around them (round 3, tools/bench_gemm.py).  The problems of a launch are
    // independent, so the order changes no result.
    int order[kGemmGroup], n_first = 0;
    bool first[kGemmGroup];
    for (int i = 0; i < n; ++i) {
        const pcrl_gemm_desc& d = descs[i];
        const long long tiles32 = (long long)((d.M + 31) / 32) * ((d.N + 31) / 32) * (d.batch > 0 ? d.batch : 1);
        first[i] = d.K >= 512 && tiles32 > 0 && tiles32 <= 256;
        if (first[i]) order[n_first++] = i;
    }
    for (int i = 0, k = n_first; i < n; ++i)
        if (!first[i]) order[k++] = i;

    g = GemmGroup{};
    GemmPlan plan[kGemmGroup];
    const pcrl_gemm_desc* dd[kGemmGroup];
    for (int oi = 0; oi < n; ++oi) {
        const pcrl_gemm_desc* d = &descs[order[oi]];
        if (int rc = gemm_fill_common(d, g.p[g.n], plan[g.n])) return rc;
        dd[g.n] = d; src[g.n] = order[oi];
        if (d->M > 0 && d->N > 0) ++g.n;        // empty problems contribute no workgroups
    }
    wg_total_out = 0; lds = 0; families = 0;
    if (g.n == 0) return PCRL_OK;
    const bool legacy = legacy_only(), force64 = tile64_min_tiles() <= 1;
    const int cus = num_cus();
    // 64 x 64 staged tiles: everything eligible (forced), or what pays -- if the launch then has enough of them to fill the chip
    bool use64[kGemmGroup];
    int tiles64 = 0;
    for (int i = 0; i < g.n; ++i) {
        use64[i] = !legacy && plan[i].t64_ok && (force64 || plan[i].t64_pays);
        if (use64[i]) tiles64 += ((dd[i]->M + 63) / 64) * ((dd[i]->N + 63) / 64) * dd[i]->batch;
    }
    if (!force64 && tiles64 < tile64_min_tiles())
        for (int i = 0; i < g.n; ++i) use64[i] = false;
    // Launch-level choice for the forward-shaped problems (round 6): the per-problem rule below picks, for each problem on its own, the
    // finest wave-tile shape with at most one workgroup per CU -- two such problems in one launch (the target and the online Q heads'
    // second layer at 256 rows: 2 x 256 workgroups of 32 x 64 outputs, 120 KB of LDS each) then run as two rounds.  When the launch's
    // forward-shaped problems together exceed one round of wave tiles AND form at least 3/4 of a round of 64 x 64 staged tiles, they take
    // those instead: one round, half the prologues / split-K reductions per FLOP (same box: 29.3 -> 25.2 us for that launch, K1's step
    // -3.5 us; at 128 rows the 64 x 64 tiles would leave half the chip idle and the wave tiles stay).
    bool launch64[kGemmGroup] = {false, false, false, false};
    if (!legacy && !force64) {
        long long wt_wgs = 0, t64_tiles = 0;
        bool all_ok = true;
        for (int i = 0; i < g.n; ++i) {
            if (!plan[i].kind_f) continue;
            const pcrl_gemm_desc* d = dd[i];
            long long wgs = 0;
            for (int sh = 0; sh < wtile_shapes(0); ++sh) {
                wgs = (long long)((d->M + wtile_tm(sh, 0) - 1) / wtile_tm(sh, 0)) * ((d->N + wtile_tn(sh, 0) - 1) / wtile_tn(sh, 0)) * d->batch;
                if (wgs <= cus) break;
            }
            wt_wgs += wgs;
            t64_tiles += (long long)((d->M + 63) / 64) * ((d->N + 63) / 64) * d->batch;
            all_ok = all_ok && plan[i].t64_ok;
        }
        if (all_ok && wt_wgs > cus && 4 * t64_tiles >= 3 * (long long)cus)
            for (int i = 0; i < g.n; ++i)       // (a small neighbour -- fewer than a quarter round of 64 x 64 tiles -- keeps its wave tiles)
                launch64[i] = plan[i].kind_f && 4LL * ((dd[i]->M + 63) / 64) * ((dd[i]->N + 63) / 64) * dd[i]->batch >= cus;
    }
    int wg_total = 0;
    unsigned fams = 0;
    for (int i = 0; i < g.n; ++i) {
        const pcrl_gemm_desc* d = dd[i];
        GemmParams& p = g.p[i];
        int tm = 32, tn = 32, n_cols = d->N;
        const bool wt = !legacy && !force64 && (plan[i].kind_f || plan[i].kind_x);
        if (launch64[i]) {
            p.cfg = 1; tm = tn = 64;
        } else if (wt) {
            // the finest tile shape that still gives the chip at most one workgroup per CU (measured, tools/probes/gemm_staged.hip: a launch
            // is fastest with the largest number of workgroups <= #CUs); beyond the coarsest shape, the 64 x 64 staged tiles where they pay
            const int brc = plan[i].kind_x ? 1 : 0;
            const int n_shapes = wtile_shapes(brc);
            int shape = n_shapes - 1;
            long long wgs_coarsest = 0;
            for (int sh = 0; sh < n_shapes; ++sh) {
                const long long wgs = (long long)((d->M + wtile_tm(sh, brc) - 1) / wtile_tm(sh, brc)) * ((d->N + wtile_tn(sh, brc) - 1) / wtile_tn(sh, brc)) * d->batch;
                wgs_coarsest = wgs;
                if (wgs <= cus) { shape = sh; break; }
            }
            if (shape == n_shapes - 1 && wgs_coarsest > cus && use64[i]) {
                p.cfg = 1; tm = tn = 64;
            } else {
                p.cfg = 4; p.shape = shape; p.b_rc = brc; tm = wtile_tm(shape, brc); tn = wtile_tn(shape, brc);
            }
        } else if (!legacy && !force64 && plan[i].kind_w) {
            const int n_real = d->ones_col >= 0 ? d->ones_col : d->N;
            const long long wgs64 = (long long)((d->M + 63) / 64) * ((n_real + 63) / 64) * d->batch;
            p.cfg = 5; p.shape = wgs64 <= cus ? 1 : 0;
            tm = 64; tn = wpanel_tn(p.shape); n_cols = n_real;
        } else if (use64[i]) {
            p.cfg = 1; tm = tn = 64;
        } else {
            gemm_pick_legacy(d, p, tm, tn);
        }
        fams |= gemm_family(p.cfg, p.b_rc);
        gemm_set_tiles(p, tm, tn, n_cols, d->batch, wg_total);
        g.wg_begin[i] = p.wg_begin;
        const size_t need = gemm_lds_bytes(p);
        lds = need > lds ? need : lds;
    }
    if (wg_total >= (1 << 20)) return fail(PCRL_E_ARG, "GEMM group too large (%d tiles)", wg_total);
    wg_total_out = wg_total; families = fams;
    return PCRL_OK;
}

extern "C" int pcrl_gemm_group_f32(const pcrl_gemm_desc* descs, int32_t n, void* stream) {
    GemmGroup g;
    size_t lds;
    int wg_total, src[kGemmGroup];
    unsigned fams;
    if (int rc = gemm_plan(descs, n, g, lds, wg_total, src, fams)) return rc;
    if (g.n == 0 || wg_total == 0) return PCRL_OK;
    const GemmVariant& v = gemm_variant(fams);
    if (int rc = ensure_dynamic_lds(v.fn, 160 * 1024)) return rc;
    void* args[] = {&g};
    PCRL_CHECK_HIP(hipLaunchKernel(v.fn, dim3(wg_total), dim3(64 * kGemmWaves), args, lds, (hipStream_t)stream));
    PCRL_CHECK_LAUNCH(v.name);
    return PCRL_OK;
}

extern "C" int pcrl_gemm_group_plan_f32(const pcrl_gemm_desc* descs, int32_t n, int32_t* out) {
    if (!out) return fail(PCRL_E_ARG, "NULL argument");
    GemmGroup g;
    size_t lds;
    int wg_total, src[kGemmGroup];
    unsigned wt;
    if (int rc = gemm_plan(descs, n, g, lds, wg_total, src, wt)) return rc;
    for (int i = 0; i < n; ++i) { out[3 * i] = -1; out[3 * i + 1] = 0; out[3 * i + 2] = 0; }
    for (int i = 0; i < g.n; ++i) { out[3 * src[i]] = g.p[i].cfg; out[3 * src[i] + 1] = g.p[i].shape; out[3 * src[i] + 2] = g.p[i].tiles; }
    return PCRL_OK;
}

extern "C" int pcrl_gemm_f32(const pcrl_gemm_desc* d, void* stream) {
    if (!d) return fail(PCRL_E_ARG, "NULL argument");
    return pcrl_gemm_group_f32(d, 1, stream);
}

extern "C" int pcrl_layernorm_rows_fwd_multi_f32(const pcrl_ln_job* jobs, int32_t n_jobs, const float* gamma, const float* beta, int32_t F,
                                                 float eps, void* stream) {
    if (!jobs || !gamma || !beta) return fail(PCRL_E_ARG, "NULL argument");
    if (F < 1 || F > 256 || n_jobs < 1 || n_jobs > kLnJobs) return fail(PCRL_E_ARG, "LayerNorm rows: 1 <= F <= 256, 1 <= n_jobs <= %d", kLnJobs);
    LnParams p{};
    p.gamma = gamma; p.beta = beta; p.F = F; p.eps = eps;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const pcrl_ln_job& s = jobs[j];
        if (s.M == 0) continue;
        if (!s.x || s.M < 0 || s.n_dst < 0 || s.n_dst > 4) return fail(PCRL_E_ARG, "bad LayerNorm job %d", j);
        LnJob& d = p.job[p.n_jobs++];
        d.x = s.x; d.ldx = s.ldx; d.M = s.M; d.n_dst = s.n_dst; d.blk_begin = blocks; d.xhat = s.xhat; d.rstd = s.rstd;
        for (int i = 0; i < s.n_dst; ++i) { if (!s.dst[i]) return fail(PCRL_E_ARG, "NULL destination"); d.y[i] = s.dst[i]; d.ldy[i] = s.ld_dst[i]; }
        for (int c = 0; c < 2; ++c) {
            if (s.cat_n[c] > 0 && (!s.cat_src[c] || !s.cat_dst[c])) return fail(PCRL_E_ARG, "NULL pass-through columns");
            d.cat_src[c] = s.cat_n[c] > 0 ? s.cat_src[c] : nullptr; d.cat_dst[c] = s.cat_dst[c];
            d.cat_lds[c] = s.cat_ld_src[c]; d.cat_ldd[c] = s.cat_ld_dst[c]; d.cat_n[c] = s.cat_n[c];
            d.cat_div[c] = s.cat_row_div[c] > 1 ? s.cat_row_div[c] : 1;
        }
        blocks += (s.M + 3) / 4;
    }
    if (blocks == 0) return PCRL_OK;
    hipLaunchKernelGGL(layernorm_rows_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("layernorm_rows_fwd_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_layernorm_rows_fwd_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, int32_t M, int32_t F,
                                           float eps, float* const* dst, const int64_t* ld_dst, int32_t n_dst,
                                           float* xhat, float* rstd, void* stream) {
    if (!x || !dst || !ld_dst) return fail(PCRL_E_ARG, "NULL argument");
    if (n_dst < 1 || n_dst > 4) return fail(PCRL_E_ARG, "LayerNorm rows: 1 <= F <= 256, 1 <= n_dst <= 4");
    pcrl_ln_job job{};
    job.x = x; job.ldx = ldx; job.M = M; job.n_dst = n_dst; job.xhat = xhat; job.rstd = rstd;
    for (int i = 0; i < n_dst; ++i) { job.dst[i] = dst[i]; job.ld_dst[i] = ld_dst[i]; }
    return pcrl_layernorm_rows_fwd_multi_f32(&job, 1, gamma, beta, F, eps, stream);
}

extern "C" int pcrl_layernorm_rows_bwd_partials_f32(const float* dy0, const float* dy1, int64_t lddy, const float* xhat, const float* rstd,
                                                    const float* gamma, int32_t M, int32_t F, float* dx, int64_t lddx,
                                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy0 || !xhat || !rstd || !gamma || !dx) return fail(PCRL_E_ARG, "NULL argument");
    if (F < 1 || F > 256) return fail(PCRL_E_ARG, "LayerNorm rows: 1 <= F <= 256");
    if (M == 0) return PCRL_OK;
    const int nblk = (M + 3) / 4;
    const size_t need = sizeof(float) * (size_t)nblk * 2 * F;
    if (!workspace || workspace_bytes < need) return fail(PCRL_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
    LnBwdParams p{dy0, dy1, lddy, xhat, rstd, gamma, M, F, dx, lddx, static_cast<float*>(workspace)};
    hipLaunchKernelGGL(layernorm_rows_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("layernorm_rows_bwd_kernel");
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
