// PointNet encoder backward for gfx950 (MI355X), fp32.
//
// Replaces autograd through the reference's ConvMLP + max-pool
//   feature.max(-1) backward, 2x LayerNorm1D backward (+4 permute copies each), 3x ReLU backward,
//   3x Conv1d(k=1) backward  (pyrl/networks/backbones/pointnet.py:148-151, mlp.py:43-56,
//   nn_layer.py:207-219)
// which the reference runs over all B*N points although the max-pool passes gradient to at most
// c3 = 256 points per cloud.  This implementation is exact, not approximate: it visits only those
// points.
//
//   prep (one 256-thread workgroup per cloud)
//     the cloud's active point list = the distinct argmax points in ascending order (a bitmap in LDS and a prefix
//     popcount: no sort), each channel's slot in it, and -- when the forward's pooled values are given -- the per-channel
//     and per-point quantities of the LayerNorm-2 / max-pool backward; norm2's own gradients.
//   kernel A (one wave per tile of 32 active points; the tiles of all clouds are dealt over the chip's waves, so a small
//   batch still uses every CU and a SIMD holds at most ceil(tiles / SIMDs) of them)
//     recompute the forward chain for the
//     active points with the same MFMA chains as the forward kernel; LayerNorm/ReLU backward in
//     the accumulator layout; input-gradient GEMMs chained in registers against transposed
//     weight images; every weight-gradient operand (dz2, h1, dz1, h0, dz0, x|1) is written to a
//     per-cloud workspace already in MFMA operand order ("pieces": 64 lanes x 4 k-slots).
//   kernel B (one workgroup per cloud)
//     the three weight-gradient GEMMs, contraction over the cloud's active points, operands
//     read as fully coalesced 1 KB pieces -> per-cloud partial gradients.
//   reduce
//     fixed-order sum of the per-cloud partials -> one flat gradient buffer.  No float atomics:
//     results are bit-reproducible run to run.
#include <algorithm>
#include <atomic>
#include "encoder_common.h"

// This file is compiled four times (encoder_bwd_{f32,bf16,split,f32_nw4}.hip define PCRL_BWD_MODE 0 / 1 / 2 / 3): each translation
// unit instantiates the kernels of ONE arithmetic mode (3: only the four-wave fp32 tile kernel) for the twelve supported shapes, so they compile in parallel; the
// C entry points, the reduce kernel and the host-side helpers live in the mode-0 unit.
#ifndef PCRL_BWD_MODE
#error "include through encoder_bwd_{f32,bf16,split}.hip"
#endif

namespace pcrl {

constexpr int kPiece = 256;   // floats per piece (64 lanes x 4)

// Flat gradient layout = the reference's parameter order inside visual_nn.conv.mlp:
// conv0.weight, conv0.bias, conv1.weight, norm1.weight, norm1.bias, conv2.weight, norm2.weight, norm2.bias
struct GradLayout {
    int C, C1, C2, C3;
    __host__ __device__ int w0() const { return 0; }
    __host__ __device__ int b0() const { return C1 * C; }
    __host__ __device__ int w1() const { return b0() + C1; }
    __host__ __device__ int g1() const { return w1() + C2 * C1; }
    __host__ __device__ int be1() const { return g1() + C2; }
    __host__ __device__ int w2() const { return be1() + C2; }
    __host__ __device__ int g2() const { return w2() + C3 * C2; }
    __host__ __device__ int be2() const { return g2() + C3; }
    __host__ __device__ int total() const { return be2() + C3; }
};

// Per-cloud operand workspace (floats): block arrays [nblk][32 octets][piece]; a cloud has at most C3 active points = C3 / 32 tiles
struct OpsLayout {
    int MB1, MB2, MB3;
    __host__ __device__ constexpr int blk() const { return 32 * kPiece; }      // floats per 32-channel block: room for 256 active points whatever c3
    __host__ __device__ constexpr int dz2() const { return 0; }
    __host__ __device__ constexpr int h1() const { return dz2() + MB3 * blk(); }
    __host__ __device__ constexpr int dz1() const { return h1() + MB2 * blk(); }
    __host__ __device__ constexpr int h0() const { return dz1() + MB2 * blk(); }
    __host__ __device__ constexpr int dz0() const { return h0() + MB1 * blk(); }
    __host__ __device__ constexpr int xb() const { return dz0() + MB1 * blk(); }
    __host__ __device__ constexpr int total() const { return xb() + blk(); }
};
constexpr int kXsFloats = 8 * 64 * 64;   // xhat1 spill: [tile][R][lane], sized for c2 = 128, c3 = 256

struct BwdParams {
    CloudParams cl;
    float eps;
    const float* packed;
    const int* argmax;       // [B][C3]
    const float* gpool;      // [B][C3]
    const float* pooled;     // [B][C3] forward output (optional): lets the per-point LayerNorm-2 backward sums be formed per channel
    float* ops;              // [B][OpsLayout.total()]
    float* xs;               // [B][kXsFloats]
    float* pw;               // [B][GradLayout.total()]
    int* n_act;              // [B]
    float* grads;            // [GradLayout.total()]
    // written by the prep kernel, read by kernel A
    int* flag;               // [B] 1: per-channel shortcut valid for this cloud (pooled given, no lossy channel)
    int* act;                // [B][kC3] active point indices, ascending
    unsigned char* slot;     // [B][kC3] slot of the channel's argmax point
    float* dx;               // [B][kC3] dL/d(xhat2) of the channel at its argmax point
    float2* pt;              // [B][kC3] per active point: (sum dx, sum dx * xhat) over the channels it owns
    float* n1part;           // [B][8][kC2][2] norm1 (dgamma, dbeta) partial sums per tile
    int tile_mode;           // 1: B < #CUs, work items are single tiles found through a prefix sum of the clouds' tile counts
    int parts;               // kernel B: workgroups per cloud (1, 2, 4 or 8; > 1 only for small batches)
    // Gram form of the backward (encoder_bwd_gram.h)
    const float* w2;         // conv2.weight as the optimizer holds it: [C3][C2] row-major
    unsigned* own;           // [B][kC3] per active slot: first entry of its channels in own_chan (low 16 bits) | their number (high 16)
    unsigned char* own_chan; // [B][kC3] the channels grouped by the slot of their argmax point, ascending within a slot
    float4* ptc;             // [B][kC3] per active slot: (a = rstd2^2 m2, rstd2 m1, a mu2, 0) -- coefficients of G, v, u
    float* chc;              // [B][kC3] per channel: rstd2 dL/dxhat2 at its argmax point (the sparse rows of dW2)
    float* gvu;              // [C2*C2 + 2*C2] sums over the clouds of G, v, u
    float* mimg;             // [C2*C2 + C2] M = W2^T W2 in A-operand order, then s = W2^T 1 (built by the prep launch)
    int pw_stride;           // floats per cloud in pw (GradLayout.total() [+ C2*C2 + 2*C2 in the Gram form])
    int phase;               // Gram form, host side: 0 whole backward, 1 the prep launch only, 2 everything after it (pcrl_encoder_bwd_prepare_f32)
    // team kernel (encoder_bwd_fused.h): one row of partial sums per workgroup, the sparse rows of dW2 per (cloud, channel)
    int fused;               // host side: 0 never, 1 the team kernel for small launches (at most two tiles per CU), 2 wherever it is built
    int fused_rows;          // rows the workspace has room for (= the largest grid)
    float* wgrows;           // [fused_rows][FusedRow.total()]
    float* srows;            // [B][C3][C2]
    int* n_items;            // [1] tiles of the launch (written by the team kernel, read by the reduce launch)
    unsigned long long* own_pack;  // [B][kC3] per active slot: its first eight channels, one byte each, 0xFF = none (c3 <= 256 prep only)
    int* schedule_out;       // host: which schedule the launch code took (pcrl_encoder_bwd_last_schedule), or NULL
    // column-sum jobs riding on the reduce launch (pcrl_encoder_bwd_attach_colsum): cs_blocks extra workgroups behind its own
    ColsumParams cs;
    int cs_blocks;
    // a row-wise LayerNorm backward riding on the PREP launch (pcrl_encoder_bwd_attach_ln_bwd): ln_blocks extra workgroups behind the clouds'
    // and the Gram image's (the feature head's LayerNorm: the prep launch needs nothing it writes, and nothing it reads is written by prep)
    LnBwdParams ln;
    int ln_blocks;
};

// LayerNorm statistics in the forward's canonical order; `a` becomes xhat = (a - mean) * rstd.
template <int C>
__device__ __forceinline__ float ln_to_xhat(f32x16 (&a)[C / 32], float eps) {
    bool unused;
    const float rstd = ln_center_rstd<C>(a, eps, &unused);
#pragma unroll
    for (int mb = 0; mb < C / 32; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[mb][r] = a[mb][r] * rstd;
    return rstd;
}

// Byte offset of accumulator slot (mb, r) inside an operand block array, lane part excluded.
__host__ __device__ constexpr unsigned op_off(int arr_floats, int mb, int r) {
    return 4u * (unsigned)(arr_floats + mb * 32 * kPiece + ((r & 3) + 8 * (r >> 2)) * 4);
}

// ---- prep: active list, slots, per-channel / per-point sums of the pool + LayerNorm-2 backward --------------------------
constexpr int kBitmapMaxWords = 8192;      // N <= 262 144 points per cloud
// Called by EVERY thread of the block (256 threads in the stand-alone kernel, 512 in kernel A's prologue: threads >= 256 only
// take part in the barriers).  s_words: [2 * nW] bitmap + prefix, s_scan: [256] ints (s_scan[255] = n_act on return),
// s_slot: [kC3] bytes, s_dx / s_xh: [kC3] floats, s_flag: 1 int (on return 1 when the per-channel shortcut must not be used).
template <int T0, int C1, int kC2, int kC3>
__device__ __forceinline__ void bwd_prep_cloud(const BwdParams& p, int b, int tid_all, unsigned* s_words, int* s_scan, unsigned char* s_slot,
                                               float* s_dx, float* s_xh, int* s_flag_p) {
    constexpr PackedLayout L{T0, C1, kC2, kC3};
    const GradLayout GL{p.cl.C, C1, kC2, kC3};
    const int nW = (p.cl.N + 31) >> 5;
    unsigned* s_pre = s_words + nW;
    const bool on = tid_all < 256;                      // the 256 threads that scan the bitmap
    const int tid = on ? tid_all : 255;
    const bool chan = on && tid < kC3;                  // thread = channel for the per-channel work (c3 <= 256)
    int& s_flag = *s_flag_p;
    // every global load of the cloud is issued up front (they are independent of the bitmap phase)
    int pc = 0;
    float y_in = 0.0f, g_in = 0.0f, gam = 1.0f, bet = 0.0f;
    const bool have_pooled = p.pooled != nullptr;
    if (chan) {
        pc = p.argmax[(long long)b * kC3 + tid];
        if (have_pooled) {
            y_in = p.pooled[(long long)b * kC3 + tid];
            g_in = p.gpool[(long long)b * kC3 + tid];
            gam = p.packed[L.ln2() + 2 * tid]; bet = p.packed[L.ln2() + 2 * tid + 1];
        }
    }
    __syncthreads();
    if (on) {
        for (int w = tid; w < nW; w += 256) s_words[w] = 0u;
        if (tid == 0) s_flag = 0;
    }
    __syncthreads();
    pc = pc < 0 ? 0 : (pc >= p.cl.N ? p.cl.N - 1 : pc);
    if (chan) atomicOr(&s_words[pc >> 5], 1u << (pc & 31));
    __syncthreads();
    // exclusive prefix popcount over the words: each thread owns `per` consecutive words; the 256 per-thread counts are
    // scanned inside each wave with DPP-free shuffles and the four wave totals are added through LDS (one barrier)
    const int per = (nW + 255) >> 8, w0 = tid * per;
    int local = 0;
    for (int k = 0; k < per; ++k)
        if (w0 + k < nW) local += __popc(s_words[w0 + k]);
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d) incl += v;
    }
    if (on && (tid & 63) == 63) s_scan[tid >> 6] = incl;
    __syncthreads();
    int wave_base = 0;
#pragma unroll
    for (int w = 0; w < 3; ++w)
        if (w < (tid >> 6)) wave_base += s_scan[w];
    if (tid_all == 255) s_scan[255] = wave_base + incl;
    int run = wave_base + incl - local;
    if (on)
        for (int k = 0; k < per; ++k)
            if (w0 + k < nW) { s_pre[w0 + k] = (unsigned)run; run += __popc(s_words[w0 + k]); }
    __syncthreads();
    const int n_act = s_scan[255];
    const int slot = (int)s_pre[pc >> 5] + __popc(s_words[pc >> 5] & ((1u << (pc & 31)) - 1u));
    if (chan) {
        s_slot[tid] = (unsigned char)slot;
        p.slot[(long long)b * kC3 + tid] = (unsigned char)slot;
        p.act[(long long)b * kC3 + slot] = pc;                   // every channel of the point writes the same value
    }
    // With the forward's pooled values the LayerNorm-2 / max-pool backward needs no search for "which of my 128 registers
    // hold a channel I own": channel c contributes only at its argmax point, where y = pooled[c] (the recompute is
    // bit-identical to the forward), so dL/dxhat_c = [y > 0] g_c gamma_c and xhat_c = (y - beta_c) / gamma_c are per-CHANNEL
    // quantities, and a point's two sums are sums over the channels that name it.
    float dyl = 0.0f, xh = 0.0f, dxc = 0.0f;
    if (have_pooled && chan) {
        const float y = y_in, g = g_in;
        const bool live = y > 0.0f;
        dyl = live ? g : 0.0f;
        // xhat = (y - beta) / gamma loses |beta| / |gamma| ulps by cancellation and is not recoverable at all through a
        // zero gamma; a non-finite y carries no xhat either.  In those cases the whole cloud takes the dense path, which
        // recomputes xhat at the point itself (the default affine, gamma ~ 1 / beta ~ 0, never gets here).
        const bool lossy = __builtin_fabsf(gam) < 1e-3f || __builtin_fabsf(bet) > 8.0f * __builtin_fabsf(gam);
        if ((live && lossy) || !(__builtin_fabsf(y) <= 3.0e38f)) s_flag = 1;
        xh = (live && gam != 0.0f) ? (y - bet) / gam : 0.0f;
        dxc = dyl * gam;
    }
    if (chan) {
        s_dx[tid] = dxc;
        s_xh[tid] = xh;
        p.dx[(long long)b * kC3 + tid] = dxc;
    }
    __syncthreads();
    const bool use_pooled = have_pooled && s_flag == 0;
    if (tid_all == 0) { p.n_act[b] = n_act; p.flag[b] = use_pooled ? 1 : 0; }
    if (use_pooled && chan) {
        float* pw = p.pw + (long long)b * p.pw_stride;
        pw[GL.g2() + tid] = dyl * xh;                               // norm2.weight / norm2.bias gradients of this cloud
        pw[GL.be2() + tid] = dyl;
        if (tid < n_act) {
            // The channels that name this point, in ascending order.  First a branch-free sweep over the slot table (four
            // channels per word, sixteen words in flight) that only collects a 256-bit membership mask; then the few set
            // bits (a point owns ~1.5 channels) are walked in order.  (Adding inside the sweep put ~250 divergent, dependent
            // LDS round trips on every wave: 12 us.)
            float t1 = 0.0f, t2 = 0.0f;
            const unsigned* slot_w = reinterpret_cast<const unsigned*>(s_slot);
            const unsigned me = (unsigned)tid * 0x01010101u;
            unsigned mine[kC3 / 32];
#pragma unroll
            for (int j0 = 0; j0 < kC3 / 4; j0 += 16) {
                unsigned w[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = slot_w[j0 + j] ^ me;        // a zero byte marks a channel of this point
                unsigned lo = 0u, hi = 0u;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    unsigned m4 = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) m4 |= (((w[j] >> (8 * e)) & 0xFFu) == 0u ? 1u : 0u) << e;
                    if (j < 8) lo |= m4 << (4 * j); else hi |= m4 << (4 * (j - 8));
                }
                mine[j0 / 8] = lo; mine[j0 / 8 + 1] = hi;
            }
#pragma unroll
            for (int k = 0; k < kC3 / 32; ++k) {
                unsigned m = mine[k];
                while (m) {
                    const int c = 32 * k + __builtin_ctz(m);
                    m &= m - 1u;
                    t1 = t1 + s_dx[c];
                    t2 = __builtin_fmaf(s_dx[c], s_xh[c], t2);
                }
            }
            p.pt[(long long)b * kC3 + tid] = float2{t1, t2};
        }
    }
}

template <int T0, int C1, int kC2, int kC3>
__global__ __launch_bounds__(256) void encoder_bwd_prep_kernel(const BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned s_words[];   // [nW] bitmap, then [nW] exclusive prefix popcounts
    __shared__ int s_scan[256];
    __shared__ __attribute__((aligned(4))) unsigned char s_slot[kC3];
    __shared__ float s_dx[kC3], s_xh[kC3];
    __shared__ int s_flag;
    for (int b = blockIdx.x; b < p.cl.B; b += gridDim.x)
        bwd_prep_cloud<T0, C1, kC2, kC3>(p, b, threadIdx.x, s_words, s_scan, s_slot, s_dx, s_xh, &s_flag);
}

// ---- kernel A, cloud mode (B >= #CUs): one workgroup per cloud, wave w takes the cloud's tile w --------------------------------
// The cloud's tables (active list by a bitonic sort of the argmax keys, slots, per-channel / per-point sums) live in LDS and
// are built by the workgroup itself.  Same per-tile chain as the tile-mode kernel below; kept as its own kernel because
// every value the tile-mode kernel carries per wave (cloud index, table pointers, resources) is a spilled register here.
template <int T0, int C1, int kC2, int kC3, bool BF16, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void encoder_bwd_points_cloud_kernel(const BwdParams p) {
    constexpr PackedLayout L{T0, C1, kC2, kC3};
    constexpr int MB1 = C1 / 32, MB2 = kC2 / 32, MB3 = kC3 / 32;
    constexpr OpsLayout OL{C1 / 32, kC2 / 32, kC3 / 32};
    const GradLayout GL{p.cl.C, C1, kC2, kC3};

    extern __shared__ __attribute__((aligned(16))) char smem[];
    ChanSrc* s_desc = reinterpret_cast<ChanSrc*>(smem);
    unsigned* s_key = reinterpret_cast<unsigned*>(s_desc + PCRL_MAX_CHANNELS);   // [256] sort keys
    int* s_scan = reinterpret_cast<int*>(s_key + kC3);                            // [256]
    int* s_act = s_scan + kC3;                                                    // [256] active point indices
    int* s_misc = s_act + kC3;                                                 // [8]   n_act, flags (all LDS is dynamic: G17)
    unsigned char* s_slot = reinterpret_cast<unsigned char*>(s_misc + 8);         // [256] slot of the channel's argmax point
    float* s_g = reinterpret_cast<float*>(s_slot + kC3);                          // [256] grad_pooled row
    float* s_ln1 = s_g + kC3;
    float* s_ln2 = s_ln1 + 2 * kC2;
    float* s_b0 = s_ln2 + 2 * kC3;
    float* s_w0 = s_b0 + C1;
    float* s_red = s_w0 + MB1 * T0 * 64;                                          // [8][kC2][2]
    float2* s_dgb = reinterpret_cast<float2*>(s_red + 8 * kC2 * 2);               // [256] norm2 (dgamma, dbeta), left by the owning lane
    float* s_dx = reinterpret_cast<float*>(s_dgb + kC3);                          // [256] dL/d(xhat2) of the channel at its argmax point
    float* s_xh = s_dx + kC3;                                                     // [256] xhat2 of the channel at its argmax point
    float2* s_pt = reinterpret_cast<float2*>(s_xh + kC3);                         // [256] per active point: (sum dx, sum dx * xhat)
    int* s_first = reinterpret_cast<int*>(s_pt + kC3);                         // [256] sorted position where the point's run of keys starts
    float* s_w2 = reinterpret_cast<float*>(s_first + kC3);
    // whole-piece stores (store_block_pieces) stay off here: this 256-register build already spills and measured 386 -> 503 us with
    // them at 512 x 1200 clouds (the tile-dealing kernel below gains: 162 -> 152 us at 256 x 1024)
    float* s_tr = nullptr;
    constexpr bool WS = false;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    {
        const f32x4* g = reinterpret_cast<const f32x4*>(p.packed + (SPLIT ? L.w2s(0) : BF16 ? L.w2b() : L.w2()));
        f32x4* s = reinterpret_cast<f32x4*>(s_w2);
        stage_to_lds<512, BF16 ? kC3 * kC2 / 8 : kC3 * kC2 / 4>(s, g, tid);
        for (int i = tid; i < MB1 * T0 * 64; i += 512) s_w0[i] = p.packed[L.w0() + i];
        for (int i = tid; i < C1; i += 512) s_b0[i] = p.packed[L.b0() + i];
        for (int i = tid; i < 2 * kC2; i += 512) s_ln1[i] = p.packed[L.ln1() + i];
        for (int i = tid; i < 2 * kC3; i += 512) s_ln2[i] = p.packed[L.ln2() + i];
        if (tid < PCRL_MAX_CHANNELS) s_desc[tid] = p.cl.ch[tid];
    }
    const __amdgpu_buffer_rsrc_t r_packed = make_rsrc(p.packed, 4u * (unsigned)L.total());
    const unsigned lane16 = 16u * (unsigned)lane;
    const f32x4* s_w2v = reinterpret_cast<const f32x4*>(s_w2);

    for (int b = blockIdx.x; b < p.cl.B; b += gridDim.x) {
        __syncthreads();
        // ---- phase 0: active point list = sorted unique argmax -----------------------------------
        if (tid < kC3) {
            s_key[tid] = ((unsigned)p.argmax[(long long)b * kC3 + tid] << 8) | (unsigned)tid;
            s_g[tid] = p.gpool[(long long)b * kC3 + tid];
        }
        if (tid == 0) s_misc[5] = 0;
        for (int k = 2; k <= kC3; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                __syncthreads();
                const int ixj = tid ^ j;
                if (tid < kC3 && ixj > tid) {
                    const unsigned a = s_key[tid], c = s_key[ixj];
                    if ((a > c) == ((tid & k) == 0)) { s_key[tid] = c; s_key[ixj] = a; }
                }
            }
        __syncthreads();
        bool head = false;
        if (tid < kC3) {
            head = tid == 0 || (s_key[tid] >> 8) != (s_key[tid - 1] >> 8);
            s_scan[tid] = head ? 1 : 0;
        }
        for (int d = 1; d < kC3; d <<= 1) {
            __syncthreads();
            int v = 0;
            if (tid < kC3 && tid >= d) v = s_scan[tid - d];
            __syncthreads();
            if (tid < kC3) s_scan[tid] += v;
        }
        __syncthreads();
        if (tid < kC3) {
            const int slot = s_scan[tid] - 1;
            if (head) s_act[slot] = (int)(s_key[tid] >> 8);
            s_slot[s_key[tid] & 255u] = (unsigned char)slot;
            if (tid == kC3 - 1) { s_misc[0] = slot + 1; p.n_act[b] = slot + 1; }
        }
        for (int i = tid; i < 8 * kC2 * 2; i += 512) s_red[i] = 0.0f;
        // With the forward's pooled values the LayerNorm-2 / max-pool backward needs no search for "which of my 128 registers
        // hold a channel I own": channel c contributes only at its argmax point, where y = pooled[c] (the recompute is
        // bit-identical to the forward), so dL/dxhat_c = [y > 0] g_c gamma_c and xhat_c = (y - beta_c) / gamma_c are per-CHANNEL
        // quantities, and a point's two sums are sums over the run of sorted keys that name it.
        bool use_pooled = p.pooled != nullptr;
        if (use_pooled && tid < kC3) {
            const float y = p.pooled[(long long)b * kC3 + tid];
            const float2 gb = reinterpret_cast<const float2*>(s_ln2)[tid];
            const bool live = y > 0.0f;
            const float dyl = live ? s_g[tid] : 0.0f;
            // xhat = (y - beta) / gamma loses |beta| / |gamma| ulps by cancellation and is not recoverable at all through a
            // zero gamma; a non-finite y carries no xhat either.  In those cases the whole cloud takes the dense path, which
            // recomputes xhat at the point itself (the default affine, gamma ~ 1 / beta ~ 0, never gets here).
            const bool lossy = __builtin_fabsf(gb.x) < 1e-3f || __builtin_fabsf(gb.y) > 8.0f * __builtin_fabsf(gb.x);
            if ((live && lossy) || !(__builtin_fabsf(y) <= 3.0e38f)) s_misc[5] = 1;
            const float xh = (live && gb.x != 0.0f) ? (y - gb.y) / gb.x : 0.0f;
            s_dx[tid] = dyl * gb.x;
            s_xh[tid] = xh;
            s_dgb[tid] = float2{dyl * xh, dyl};                 // norm2.weight / norm2.bias gradients of this cloud
            if (head) s_first[s_scan[tid] - 1] = tid;           // here tid is also a sorted position: where this point's run starts
        }
        __syncthreads();
        const int n_act = s_misc[0];
        if (use_pooled && s_misc[5] != 0) use_pooled = false;
        if (use_pooled && tid < n_act) {
            const int first = s_first[tid], last = tid + 1 < n_act ? s_first[tid + 1] : kC3;
            float t1 = 0.0f, t2 = 0.0f;
            for (int q = first; q < last; ++q) {                // fixed order: ascending channel within the run
                const int c = (int)(s_key[q] & 255u);
                t1 = t1 + s_dx[c];
                t2 = __builtin_fmaf(s_dx[c], s_xh[c], t2);
            }
            s_pt[tid] = float2{t1, t2};
        }
        __syncthreads();

        float* pw = p.pw + (long long)b * p.pw_stride;
        const __amdgpu_buffer_rsrc_t r_ops = make_rsrc(p.ops + (long long)b * OL.total(), 4u * (unsigned)OL.total());
        const __amdgpu_buffer_rsrc_t r_xs = make_rsrc(p.xs + (long long)b * kXsFloats, 4u * (unsigned)kXsFloats);
        if (32 * wave < n_act) {
            const int s = 32 * wave + l31;
            const bool valid = s < n_act;
            const int pidx = s_act[valid ? s : n_act - 1];
            const unsigned s_match = valid ? (unsigned)s : 0xFFFFu;   // never equals a slot byte when invalid
            // lane-dependent byte offset of an operand element: octet q = s >> 3, k-lane (s >> 2) & 1, k-slot s & 3
            const unsigned lane_off = 4u * (unsigned)(((s >> 3) * 64 + ((s >> 2) & 1) * 32 + 4 * half) * 4 + (s & 3));
            const unsigned xs_off = 4u * (unsigned)(wave * 64 * 64 + lane);
            const unsigned tile_bytes = 4096u * (unsigned)wave;       // tile = wave; whole-piece stores of the bf16 build (store_block_pieces)

            const f32x16 x = load_point<T0>(p.cl, s_desc, b, pidx);
            if (half == 0) {   // B operand of the conv0 weight gradient: rows = input channels, row C = 1 (bias)
#pragma unroll
                for (int c = 0; c < 2 * T0; ++c)
                    if (c < p.cl.C) buf_store_f1(r_ops, lane_off, 4u * (unsigned)(OL.xb() + c * 4), x[c]);
                buf_store_f1(r_ops, lane_off + 16u * (unsigned)p.cl.C, 4u * (unsigned)OL.xb(), 1.0f);
            }
            const unsigned half_mask = half ? 0xFFFFFFFFu : 0u;
            // ---- forward recompute: conv0 + ReLU -------------------------------------------------
            f32x16 a0[MB1];
            unsigned mask0[MB1];
#pragma unroll
            for (int mb = 0; mb < MB1; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a0[mb][r] = s_b0[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
                for (int t = 0; t < T0; ++t) {
                    const float bop = half_select(x[2 * t], x[2 * t + 1], half_mask);
                    a0[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_w0[(mb * T0 + t) * 64 + lane], bop, a0[mb], 0, 0, 0);
                }
                mask0[mb] = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    a0[mb][r] = relu_nan(a0[mb][r]);
                    mask0[mb] |= (a0[mb][r] > 0.0f ? 1u : 0u) << r;
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.h0(), mb, r), a0[mb][r]);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.h0(), mb, 0), tile_bytes, a0[mb], l31, half, lane);
            }
            // ---- conv1 + LN: xhat1 is spilled to the workspace, h1 stays ---------------------------
            f32x16 a1[MB2];
            if (SPLIT)      // the split-precision forward's arithmetic: the recompute is bit-identical to that forward
                dense_layer_split<MB2, C1 / 16>(
                    a1, [&](int k, int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1s(k) + (mb * (C1 / 16) + g) * 256)); },
                    [&](int t) { return a0[t >> 4][t & 15]; });
            else if (BF16)
                dense_layer_bf16<MB2, C1 / 16>(
                    a1, [&](int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1b() + (mb * (C1 / 16) + g) * 256)); },
                    [&](int t) { return a0[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB2, C1 / 8, 3>(
                    a1, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1() + (mb * (C1 / 8) + tq) * 256)); },
                    [&](int t) { return a0[t >> 4][t & 15]; });
            const float rstd1 = ln_to_xhat<kC2>(a1, p.eps);
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb) {
                float2 gbv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) gbv[r] = reinterpret_cast<const float2*>(s_ln1)[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    buf_store_f1(r_xs, xs_off, 4u * (unsigned)((mb * 16 + r) * 64), a1[mb][r]);
                    a1[mb][r] = relu_nan(__builtin_fmaf(a1[mb][r], gbv[r].x, gbv[r].y));
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.h1(), mb, r), a1[mb][r]);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.h1(), mb, 0), tile_bytes, a1[mb], l31, half, lane);
            }
            // ---- conv2 + LN -> xhat2 -------------------------------------------------------------
            f32x16 a2[MB3];
            if (SPLIT)
                dense_layer_split<MB3, kC2 / 16>(
                    a2, [&](int k, int mb, int g) {
                        return k < 2 ? s_w2v[k * (kC3 * kC2 / 8) + (mb * (kC2 / 16) + g) * 64 + lane]
                                     : buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w2s(2) + (mb * (kC2 / 16) + g) * 256)); },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            else if (BF16)
                dense_layer_bf16<MB3, kC2 / 16>(
                    a2, [&](int mb, int g) { return s_w2v[(mb * (kC2 / 16) + g) * 64 + lane]; },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB3, kC2 / 8, 2>(
                    a2, [&](int mb, int tq) { return s_w2v[(mb * (kC2 / 8) + tq) * 64 + lane]; },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            const float rstd2 = ln_to_xhat<kC3>(a2, p.eps);

            // ---- max-pool + ReLU + LN2 backward ----------------------------------------------------
            // dY2[point][c] = grad_pooled[c] if this point is channel c's argmax, else 0; a point owns
            // ~c3/n_act channels.  own[] marks the channels whose argmax lies in THIS tile (wave-uniform
            // masks), so the ownership arithmetic (branch-free inside) runs for ~1/3 of the registers.
            unsigned long long own[kC3 / 64];
#pragma unroll
            for (int k = 0; k < kC3 / 64; ++k) own[k] = __ballot(((unsigned)s_slot[64 * k + lane] >> 5) == (unsigned)wave);
            float m1, m2;
            if (use_pooled) {                       // the point's sums were formed per channel before the tiles
                const float2 t = s_pt[valid ? s : 0];
                m1 = valid ? t.x / (float)kC3 : 0.0f;      // padding lanes of the last tile own nothing: their dz must stay 0
                m2 = valid ? t.y / (float)kC3 : 0.0f;
            } else {
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int mb = 0; mb < MB3; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch0 = acc_chan(mb * 16 + r, 0);
                        if ((own[ch0 >> 6] >> (ch0 & 63)) & 0x11ull) {      // channel ch0 or ch0 + 4 owned in this tile
                            const int ch = ch0 + 4 * half;
                            const float2 gb = reinterpret_cast<const float2*>(s_ln2)[ch];
                            const float y = __builtin_fmaf(a2[mb][r], gb.x, gb.y);
                            const bool mine = (unsigned)s_slot[ch] == s_match;
                            const float dyl = (mine && y > 0.0f) ? s_g[ch] : 0.0f;
                            // exactly one point per channel contributes; the pair goes to LDS (a global store here needs the
                            // spilled base address back and with it a wait for every store in flight) and out after the tiles
                            if (mine) s_dgb[ch] = float2{dyl * a2[mb][r], dyl};
                            const float dx = dyl * gb.x;
                            s1 = s1 + dx;
                            s2 = __builtin_fmaf(dx, a2[mb][r], s2);
                        }
                    }
                float lo, hi;
                both_halves(s1, lo, hi);
                m1 = (lo + hi) / (float)kC3;
                both_halves(s2, lo, hi);
                m2 = (lo + hi) / (float)kC3;
            }
            const float cA = -(rstd2 * m2), cB = -(rstd2 * m1);     // dz = rstd*dx - rstd*m1 - xhat*rstd*m2
#pragma unroll
            for (int mb = 0; mb < MB3; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch0 = acc_chan(mb * 16 + r, 0);
                    float dz = __builtin_fmaf(a2[mb][r], cA, cB);
                    if ((own[ch0 >> 6] >> (ch0 & 63)) & 0x11ull) {
                        const int ch = ch0 + 4 * half;
                        const bool mine = (unsigned)s_slot[ch] == s_match;
                        float dx;
                        if (use_pooled) {                       // the channel's dL/dxhat is already in the table
                            dx = mine ? s_dx[ch] : 0.0f;
                        } else {
                            const float2 gb = reinterpret_cast<const float2*>(s_ln2)[ch];
                            const float y = __builtin_fmaf(a2[mb][r], gb.x, gb.y);
                            dx = ((mine && y > 0.0f) ? s_g[ch] : 0.0f) * gb.x;
                        }
                        dz = __builtin_fmaf(rstd2, dx, dz);
                    }
                    a2[mb][r] = dz;
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.dz2(), mb, r), dz);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.dz2(), mb, 0), tile_bytes, a2[mb], l31, half, lane);
            }
            // ---- dH1 = W2^T dz2 ; ReLU + LN1 backward ----------------------------------------------
            f32x16 d1[MB2];
            if (SPLIT)
                dense_layer_split<MB2, kC3 / 16>(
                    d1, [&](int k, int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w2ts(k) + (mb * (kC3 / 16) + g) * 256)); },
                    [&](int t) { return a2[t >> 4][t & 15]; });
            else if (BF16)     // the gradient is rounded to bf16 as it enters the contraction, like the activations of the forward
                dense_layer_bf16<MB2, kC3 / 16>(
                    d1, [&](int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w2tb() + (mb * (kC3 / 16) + g) * 256)); },
                    [&](int t) { return a2[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB2, kC3 / 8, 3>(
                    d1, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w2t() + (mb * (kC3 / 8) + tq) * 256)); },
                    [&](int t) { return a2[t >> 4][t & 15]; });
            f32x16 xh1[MB2];
            float s1 = 0.0f, s2 = 0.0f, lo, hi;
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb)           // all 64 reloads of xhat1 in flight at once
#pragma unroll
                for (int r = 0; r < 16; ++r) xh1[mb][r] = buf_load_f1(r_xs, xs_off, 4u * (unsigned)((mb * 16 + r) * 64));
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb) {
                float2 gbv[16]; float tg[16], tb[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) gbv[r] = reinterpret_cast<const float2*>(s_ln1)[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float y = __builtin_fmaf(xh1[mb][r], gbv[r].x, gbv[r].y);
                    const float dyl = y > 0.0f ? d1[mb][r] : 0.0f;
                    tg[r] = dyl * xh1[mb][r];          // norm1.weight / norm1.bias gradients: summed over this tile's 32 points below
                    tb[r] = dyl;
                    const float dx = dyl * gbv[r].x;
                    d1[mb][r] = dx;
                    s1 = s1 + dx;
                    s2 = __builtin_fmaf(dx, xh1[mb][r], s2);
                }
                allreduce_add32_x16(tg);
                allreduce_add32_x16(tb);
                if (l31 == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = acc_chan(mb * 16 + r, 0) + 4 * half;
                        reinterpret_cast<float2*>(s_red)[wave * kC2 + ch] = float2{tg[r], tb[r]};
                    }
                }
            }
            both_halves(s1, lo, hi);
            const float n1 = (lo + hi) / (float)kC2;
            both_halves(s2, lo, hi);
            const float n2 = (lo + hi) / (float)kC2;
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    d1[mb][r] = rstd1 * ((d1[mb][r] - n1) - xh1[mb][r] * n2);
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.dz1(), mb, r), d1[mb][r]);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.dz1(), mb, 0), tile_bytes, d1[mb], l31, half, lane);
            }
            // ---- dH0 = W1^T dz1 ; ReLU backward ----------------------------------------------------
            f32x16 d0[MB1];
            if (SPLIT)
                dense_layer_split<MB1, kC2 / 16>(
                    d0, [&](int k, int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1ts(k) + (mb * (kC2 / 16) + g) * 256)); },
                    [&](int t) { return d1[t >> 4][t & 15]; });
            else if (BF16)
                dense_layer_bf16<MB1, kC2 / 16>(
                    d0, [&](int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1tb() + (mb * (kC2 / 16) + g) * 256)); },
                    [&](int t) { return d1[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB1, kC2 / 8, 3>(
                    d0, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1t() + (mb * (kC2 / 8) + tq) * 256)); },
                    [&](int t) { return d1[t >> 4][t & 15]; });
#pragma unroll
            for (int mb = 0; mb < MB1; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    d0[mb][r] = ((mask0[mb] >> r) & 1u) ? d0[mb][r] : 0.0f;
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.dz0(), mb, r), d0[mb][r]);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.dz0(), mb, 0), tile_bytes, d0[mb], l31, half, lane);
            }
        }
        __syncthreads();
        if (tid < 2 * kC2) {   // norm1 gradients: fixed-order sum over the waves of this cloud
            const int ch = tid >> 1, which = tid & 1;
            float acc = 0.0f;
            for (int w = 0; w < 8; ++w) acc = acc + s_red[(w * kC2 + ch) * 2 + which];
            pw[(which ? GL.be1() : GL.g1()) + ch] = acc;
        }
        if (tid < kC3) {       // norm2 gradients: every channel was written by the lane holding its argmax point
            const float2 d = s_dgb[tid];
            pw[GL.g2() + tid] = d.x;
            pw[GL.be2() + tid] = d.y;
        }
    }
}

// ---- kernel A, tile mode (B < #CUs) ----------------------------------------------------------------------------------
// BF16: the forward recompute contracts bf16 operands exactly as encoder_fwd_kernel<.., true> does (so that the LayerNorm
// inputs, ReLU masks and argmax relations are those of the forward that produced `argmax`) and so do the two data-gradient
// GEMMs (bf16 transposed weight images, gradients rounded to bf16 as they enter, fp32 accumulation; roundings straight-
// through).  The weight-gradient GEMMs of kernel B stay fp32 on the unrounded operands.
#ifdef PCRL_BWD_STAMPS
// Development build only (-DPCRL_BWD_STAMPS on the fp32 unit): shader-clock stamps at the phase boundaries of a tile's chain,
// read back with pcrl_debug_bwd_stamps (tools/bwd_stamps.py prints the per-phase medians).
__device__ unsigned long long g_bwd_stamps[16384][8];
#define PCRL_STAMP(k) do { if (lane == 0 && item < 16384) g_bwd_stamps[item][k] = __builtin_readcyclecounter(); } while (0)
#else
#define PCRL_STAMP(k) do { } while (0)
#endif
constexpr int kTileTabBytes = 256 + 4 * 256;       // per wave: slot bytes + dx floats of the tile's cloud (c3 <= 256)
constexpr int kMaxTileModeClouds = 2048;          // tile mode keeps the clouds' tile prefix in LDS
// NW: waves per workgroup.  8 (two per SIMD, 256 registers each, ~110 of them spilled) is the throughput build; 4 (one per SIMD,
// 512 registers: nothing spills) serves batches whose tiles fit one per SIMD anyway -- the per-GPU shares of a multi-GPU run.
template <int T0, int C1, int kC2, int kC3, bool BF16, bool SPLIT = false, int NW = 8>
__global__ __launch_bounds__(64 * NW, 1) void encoder_bwd_points_kernel(const BwdParams p) {
    constexpr PackedLayout L{T0, C1, kC2, kC3};
    constexpr int MB1 = C1 / 32, MB2 = kC2 / 32, MB3 = kC3 / 32;
    constexpr OpsLayout OL{C1 / 32, kC2 / 32, kC3 / 32};
    const GradLayout GL{p.cl.C, C1, kC2, kC3};

    extern __shared__ __attribute__((aligned(16))) char smem[];
    ChanSrc* s_desc = reinterpret_cast<ChanSrc*>(smem);
    char* s_tab = reinterpret_cast<char*>(s_desc + PCRL_MAX_CHANNELS);             // [8 waves][kTileTabBytes]
    int* s_tstart = reinterpret_cast<int*>(s_tab + 8 * kTileTabBytes);            // [kMaxTileModeClouds + 8] tile prefix (tile mode)
    float* s_ln1 = reinterpret_cast<float*>(s_tstart + kMaxTileModeClouds + 8);
    float* s_ln2 = s_ln1 + 2 * kC2;
    float* s_b0 = s_ln2 + 2 * kC3;
    float* s_w0 = s_b0 + C1;
    float* s_w2 = s_w0 + MB1 * T0 * 64;
    // bf16 build: the conv2 image takes half of its room; the rest holds the waves' transposition scratch (store_block_pieces)
    float* s_tr = s_w2 + kC3 * kC2 / 2 + (threadIdx.x >> 6) * kTrFloats;
    constexpr bool WS = BF16 && 8 * kTrFloats <= kC3 * kC2 / 2;      // whole-piece stores: the scratch of 8 waves fits behind the bf16 image

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    {
        const f32x4* g = reinterpret_cast<const f32x4*>(p.packed + (SPLIT ? L.w2s(0) : BF16 ? L.w2b() : L.w2()));
        f32x4* s = reinterpret_cast<f32x4*>(s_w2);
        stage_to_lds<64 * NW, BF16 ? kC3 * kC2 / 8 : kC3 * kC2 / 4>(s, g, tid);
        for (int i = tid; i < MB1 * T0 * 64; i += 64 * NW) s_w0[i] = p.packed[L.w0() + i];
        for (int i = tid; i < C1; i += 64 * NW) s_b0[i] = p.packed[L.b0() + i];
        for (int i = tid; i < 2 * kC2; i += 64 * NW) s_ln1[i] = p.packed[L.ln1() + i];
        for (int i = tid; i < 2 * kC3; i += 64 * NW) s_ln2[i] = p.packed[L.ln2() + i];
        if (tid < PCRL_MAX_CHANNELS) s_desc[tid] = p.cl.ch[tid];
        if (wave == 0) {          // exclusive prefix of the clouds' tile counts: 64 clouds per wave-wide scan step
            int run = 0;
            for (int b0 = 0; b0 < p.cl.B; b0 += 64) {
                const int b = b0 + lane;
                const int nt = b < p.cl.B ? (p.n_act[b] + 31) >> 5 : 0;
                int inc = nt;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(inc, off, 64); if (lane >= off) inc += up; }
                if (b < p.cl.B) s_tstart[b] = run + inc - nt;
                run += __shfl(inc, 63, 64);
            }
            if (lane == 0) s_tstart[p.cl.B] = run;
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r_packed = make_rsrc(p.packed, 4u * (unsigned)L.total());
    const unsigned lane16 = 16u * (unsigned)lane;
    const f32x4* s_w2v = reinterpret_cast<const f32x4*>(s_w2);
    unsigned char* s_slot = reinterpret_cast<unsigned char*>(s_tab + wave * kTileTabBytes);
    float* s_dx = reinterpret_cast<float*>(s_slot + kC3);

    // Work items: the tiles of all clouds form one list and wave w of workgroup g takes item w * grid + g, so the first
    // `grid` items go to wave 0 of every workgroup (one tile per SIMD before any SIMD gets a second one).  With B < #CUs
    // there are at most 8 * B <= 8 * grid items: one per wave at most; a larger batch makes further rounds of the same deal.
    const int n_items = s_tstart[p.cl.B];
    for (int item = wave * (int)gridDim.x + (int)blockIdx.x; item < n_items; item += NW * (int)gridDim.x) {
        int b, tile, n_act;
        {
            int lo = 0, hi = p.cl.B;             // largest b with s_tstart[b] <= item
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_tstart[mid] <= item) lo = mid; else hi = mid; }
            b = lo; tile = item - s_tstart[lo];
        }
        n_act = p.n_act[b];
        const bool use_pooled = p.flag[b] != 0;
        // this wave's copy of the cloud's tables
        if (lane < kC3 / 4) {
            reinterpret_cast<unsigned*>(s_slot)[lane] = reinterpret_cast<const unsigned*>(p.slot + (long long)b * kC3)[lane];
            reinterpret_cast<f32x4*>(s_dx)[lane] = reinterpret_cast<const f32x4*>(p.dx + (long long)b * kC3)[lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        b = __builtin_amdgcn_readfirstlane(b);
        tile = __builtin_amdgcn_readfirstlane(tile);
        n_act = __builtin_amdgcn_readfirstlane(n_act);
        const float* g_row = p.gpool + (long long)b * kC3;

        float* pw = p.pw + (long long)b * p.pw_stride;
        const __amdgpu_buffer_rsrc_t r_ops = make_rsrc(p.ops + (long long)b * OL.total(), 4u * (unsigned)OL.total());
        const __amdgpu_buffer_rsrc_t r_xs = make_rsrc(p.xs + (long long)b * kXsFloats, 4u * (unsigned)kXsFloats);
        {
            const int s = 32 * tile + l31;
            const bool valid = s < n_act;
            const int pidx = p.act[(long long)b * kC3 + (valid ? s : n_act - 1)];
            const unsigned s_match = valid ? (unsigned)s : 0xFFFFu;   // never equals a slot byte when invalid
            // lane-dependent byte offset of an operand element: octet q = s >> 3, k-lane (s >> 2) & 1, k-slot s & 3
            const unsigned lane_off = 4u * (unsigned)(((s >> 3) * 64 + ((s >> 2) & 1) * 32 + 4 * half) * 4 + (s & 3));
            const unsigned xs_off = 4u * (unsigned)(tile * 64 * 64 + lane);
            const unsigned tile_bytes = 4096u * (unsigned)tile;       // whole-piece stores of the bf16 build (store_block_pieces)

            PCRL_STAMP(0);
            const f32x16 x = load_point<T0>(p.cl, s_desc, b, pidx);
            if (half == 0) {   // B operand of the conv0 weight gradient: rows = input channels, row C = 1 (bias)
#pragma unroll
                for (int c = 0; c < 2 * T0; ++c)
                    if (c < p.cl.C) buf_store_f1(r_ops, lane_off, 4u * (unsigned)(OL.xb() + c * 4), x[c]);
                buf_store_f1(r_ops, lane_off + 16u * (unsigned)p.cl.C, 4u * (unsigned)OL.xb(), 1.0f);
            }
            const unsigned half_mask = half ? 0xFFFFFFFFu : 0u;
            // ---- forward recompute: conv0 + ReLU -------------------------------------------------
            f32x16 a0[MB1];
            unsigned mask0[MB1];
#pragma unroll
            for (int mb = 0; mb < MB1; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a0[mb][r] = s_b0[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
                for (int t = 0; t < T0; ++t) {
                    const float bop = half_select(x[2 * t], x[2 * t + 1], half_mask);
                    a0[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_w0[(mb * T0 + t) * 64 + lane], bop, a0[mb], 0, 0, 0);
                }
                mask0[mb] = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    a0[mb][r] = relu_nan(a0[mb][r]);
                    mask0[mb] |= (a0[mb][r] > 0.0f ? 1u : 0u) << r;
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.h0(), mb, r), a0[mb][r]);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.h0(), mb, 0), tile_bytes, a0[mb], l31, half, lane);
            }
            PCRL_STAMP(1);
            // ---- conv1 + LN: xhat1 is spilled to the workspace, h1 stays ---------------------------
            f32x16 a1[MB2];
            if (SPLIT)      // the split-precision forward's arithmetic: the recompute is bit-identical to that forward
                dense_layer_split<MB2, C1 / 16>(
                    a1, [&](int k, int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1s(k) + (mb * (C1 / 16) + g) * 256)); },
                    [&](int t) { return a0[t >> 4][t & 15]; });
            else if (BF16)
                dense_layer_bf16<MB2, C1 / 16>(
                    a1, [&](int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1b() + (mb * (C1 / 16) + g) * 256)); },
                    [&](int t) { return a0[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB2, C1 / 8, 3>(
                    a1, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1() + (mb * (C1 / 8) + tq) * 256)); },
                    [&](int t) { return a0[t >> 4][t & 15]; });
            const float rstd1 = ln_to_xhat<kC2>(a1, p.eps);
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb) {
                float2 gbv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) gbv[r] = reinterpret_cast<const float2*>(s_ln1)[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    buf_store_f1(r_xs, xs_off, 4u * (unsigned)((mb * 16 + r) * 64), a1[mb][r]);
                    a1[mb][r] = relu_nan(__builtin_fmaf(a1[mb][r], gbv[r].x, gbv[r].y));
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.h1(), mb, r), a1[mb][r]);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.h1(), mb, 0), tile_bytes, a1[mb], l31, half, lane);
            }
            PCRL_STAMP(2);
            // ---- conv2 + LN -> xhat2 -------------------------------------------------------------
            f32x16 a2[MB3];
            if (SPLIT)
                dense_layer_split<MB3, kC2 / 16>(
                    a2, [&](int k, int mb, int g) {
                        return k < 2 ? s_w2v[k * (kC3 * kC2 / 8) + (mb * (kC2 / 16) + g) * 64 + lane]
                                     : buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w2s(2) + (mb * (kC2 / 16) + g) * 256)); },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            else if (BF16)
                dense_layer_bf16<MB3, kC2 / 16>(
                    a2, [&](int mb, int g) { return s_w2v[(mb * (kC2 / 16) + g) * 64 + lane]; },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB3, kC2 / 8, 2>(
                    a2, [&](int mb, int tq) { return s_w2v[(mb * (kC2 / 8) + tq) * 64 + lane]; },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            const float rstd2 = ln_to_xhat<kC3>(a2, p.eps);

            PCRL_STAMP(3);
            // ---- max-pool + ReLU + LN2 backward ----------------------------------------------------
            // dY2[point][c] = grad_pooled[c] if this point is channel c's argmax, else 0; a point owns
            // ~c3/n_act channels.  own[] marks the channels whose argmax lies in THIS tile (wave-uniform
            // masks), so the ownership arithmetic (branch-free inside) runs for ~1/3 of the registers.
            unsigned long long own[kC3 / 64];
#pragma unroll
            for (int k = 0; k < kC3 / 64; ++k) own[k] = __ballot(((unsigned)s_slot[64 * k + lane] >> 5) == (unsigned)tile);
            float m1, m2;
            if (use_pooled) {                       // the point's sums were formed per channel before the tiles
                const float2 t = p.pt[(long long)b * kC3 + (valid ? s : 0)];
                m1 = valid ? t.x / (float)kC3 : 0.0f;      // padding lanes of the last tile own nothing: their dz must stay 0
                m2 = valid ? t.y / (float)kC3 : 0.0f;
            } else {
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int mb = 0; mb < MB3; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch0 = acc_chan(mb * 16 + r, 0);
                        if ((own[ch0 >> 6] >> (ch0 & 63)) & 0x11ull) {      // channel ch0 or ch0 + 4 owned in this tile
                            const int ch = ch0 + 4 * half;
                            const float2 gb = reinterpret_cast<const float2*>(s_ln2)[ch];
                            const float y = __builtin_fmaf(a2[mb][r], gb.x, gb.y);
                            const bool mine = (unsigned)s_slot[ch] == s_match;
                            const float dyl = (mine && y > 0.0f) ? g_row[ch] : 0.0f;
                            // exactly one point per channel contributes: norm2's gradients of this cloud (dense path only: with the
                            // per-channel shortcut the prep kernel has written them)
                            if (mine) { pw[GL.g2() + ch] = dyl * a2[mb][r]; pw[GL.be2() + ch] = dyl; }
                            const float dx = dyl * gb.x;
                            s1 = s1 + dx;
                            s2 = __builtin_fmaf(dx, a2[mb][r], s2);
                        }
                    }
                float lo, hi;
                both_halves(s1, lo, hi);
                m1 = (lo + hi) / (float)kC3;
                both_halves(s2, lo, hi);
                m2 = (lo + hi) / (float)kC3;
            }
            const float cA = -(rstd2 * m2), cB = -(rstd2 * m1);     // dz = rstd*dx - rstd*m1 - xhat*rstd*m2
#pragma unroll
            for (int mb = 0; mb < MB3; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch0 = acc_chan(mb * 16 + r, 0);
                    float dz = __builtin_fmaf(a2[mb][r], cA, cB);
                    if ((own[ch0 >> 6] >> (ch0 & 63)) & 0x11ull) {
                        const int ch = ch0 + 4 * half;
                        const bool mine = (unsigned)s_slot[ch] == s_match;
                        float dx;
                        if (use_pooled) {                       // the channel's dL/dxhat is already in the table
                            dx = mine ? s_dx[ch] : 0.0f;
                        } else {
                            const float2 gb = reinterpret_cast<const float2*>(s_ln2)[ch];
                            const float y = __builtin_fmaf(a2[mb][r], gb.x, gb.y);
                            dx = ((mine && y > 0.0f) ? g_row[ch] : 0.0f) * gb.x;
                        }
                        dz = __builtin_fmaf(rstd2, dx, dz);
                    }
                    a2[mb][r] = dz;
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.dz2(), mb, r), dz);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.dz2(), mb, 0), tile_bytes, a2[mb], l31, half, lane);
            }
            PCRL_STAMP(4);
            // ---- dH1 = W2^T dz2 ; ReLU + LN1 backward ----------------------------------------------
            f32x16 d1[MB2];
            if (SPLIT)
                dense_layer_split<MB2, kC3 / 16>(
                    d1, [&](int k, int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w2ts(k) + (mb * (kC3 / 16) + g) * 256)); },
                    [&](int t) { return a2[t >> 4][t & 15]; });
            else if (BF16)     // the gradient is rounded to bf16 as it enters the contraction, like the activations of the forward
                dense_layer_bf16<MB2, kC3 / 16>(
                    d1, [&](int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w2tb() + (mb * (kC3 / 16) + g) * 256)); },
                    [&](int t) { return a2[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB2, kC3 / 8, 3>(
                    d1, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w2t() + (mb * (kC3 / 8) + tq) * 256)); },
                    [&](int t) { return a2[t >> 4][t & 15]; });
            PCRL_STAMP(5);
            f32x16 xh1[MB2];
            float s1 = 0.0f, s2 = 0.0f, lo, hi;
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb)           // all 64 reloads of xhat1 in flight at once
#pragma unroll
                for (int r = 0; r < 16; ++r) xh1[mb][r] = buf_load_f1(r_xs, xs_off, 4u * (unsigned)((mb * 16 + r) * 64));
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb) {
                float2 gbv[16]; float tg[16], tb[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) gbv[r] = reinterpret_cast<const float2*>(s_ln1)[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float y = __builtin_fmaf(xh1[mb][r], gbv[r].x, gbv[r].y);
                    const float dyl = y > 0.0f ? d1[mb][r] : 0.0f;
                    tg[r] = dyl * xh1[mb][r];          // norm1.weight / norm1.bias gradients: summed over this tile's 32 points below
                    tb[r] = dyl;
                    const float dx = dyl * gbv[r].x;
                    d1[mb][r] = dx;
                    s1 = s1 + dx;
                    s2 = __builtin_fmaf(dx, xh1[mb][r], s2);
                }
                allreduce_add32_x16(tg);
                allreduce_add32_x16(tb);
                if (l31 == 0) {        // this tile's partial sums; kernel B adds a cloud's tiles in tile order
                    float2* n1 = reinterpret_cast<float2*>(p.n1part) + ((long long)b * 8 + tile) * kC2;
#pragma unroll
                    for (int r = 0; r < 16; ++r) n1[acc_chan(mb * 16 + r, 0) + 4 * half] = float2{tg[r], tb[r]};
                }
            }
            both_halves(s1, lo, hi);
            const float n1 = (lo + hi) / (float)kC2;
            both_halves(s2, lo, hi);
            const float n2 = (lo + hi) / (float)kC2;
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    d1[mb][r] = rstd1 * ((d1[mb][r] - n1) - xh1[mb][r] * n2);
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.dz1(), mb, r), d1[mb][r]);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.dz1(), mb, 0), tile_bytes, d1[mb], l31, half, lane);
            }
            PCRL_STAMP(6);
            // ---- dH0 = W1^T dz1 ; ReLU backward ----------------------------------------------------
            f32x16 d0[MB1];
            if (SPLIT)
                dense_layer_split<MB1, kC2 / 16>(
                    d0, [&](int k, int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1ts(k) + (mb * (kC2 / 16) + g) * 256)); },
                    [&](int t) { return d1[t >> 4][t & 15]; });
            else if (BF16)
                dense_layer_bf16<MB1, kC2 / 16>(
                    d0, [&](int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1tb() + (mb * (kC2 / 16) + g) * 256)); },
                    [&](int t) { return d1[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB1, kC2 / 8, 3>(
                    d0, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1t() + (mb * (kC2 / 8) + tq) * 256)); },
                    [&](int t) { return d1[t >> 4][t & 15]; });
#pragma unroll
            for (int mb = 0; mb < MB1; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    d0[mb][r] = ((mask0[mb] >> r) & 1u) ? d0[mb][r] : 0.0f;
                    if constexpr (!WS) buf_store_f1(r_ops, lane_off, op_off(OL.dz0(), mb, r), d0[mb][r]);
                }
                if constexpr (WS) store_block_pieces(r_ops, s_tr, op_off(OL.dz0(), mb, 0), tile_bytes, d0[mb], l31, half, lane);
            }
            PCRL_STAMP(7);
        }
    }
}

// ---- kernel B: per-cloud weight-gradient GEMMs ------------------------------------------------
// out[32 x 32 block (mb, nb)] = sum over slots of A[32mb + i][slot] * Bm[32nb + j][slot]
// b_block_stride: distance between two column blocks of the B operand in 16-byte units (32 * 64 in the global workspace,
// n_oct * 64 in the compact LDS copy).
template <int NB, class BPtr, int D = 3>
__device__ __forceinline__ void wgrad_blocks(const float* __restrict__ A, BPtr b4_base, int b_block_stride, int mb, int nb0,
                                             int n_oct, int lane, f32x16 (&acc)[NB]) {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(A) + (long long)mb * 32 * 64 + lane;
    const auto b4 = b4_base + lane;
    // the operands of octets q + 1 .. q + D are in flight while the 4 * NB MFMAs of octet q issue: an L2 round trip is longer
    // than one octet's MFMAs (16 x 64 cycles), so a single octet of look-ahead left the matrix pipe waiting every iteration
    f32x4 ar[D], br[D][NB];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < n_oct) {
            ar[d] = a4[d * 64];
#pragma unroll
            for (int n = 0; n < NB; ++n) br[d][n] = b4[(nb0 + n) * b_block_stride + d * 64];
        }
    for (int q0 = 0; q0 < n_oct; q0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int q = q0 + d;
            if (q < n_oct) {
                const f32x4 a = ar[d];
                f32x4 bv[NB];
#pragma unroll
                for (int n = 0; n < NB; ++n) bv[n] = br[d][n];
                if (q + D < n_oct) {
                    ar[d] = a4[(q + D) * 64];
#pragma unroll
                    for (int n = 0; n < NB; ++n) br[d][n] = b4[(nb0 + n) * b_block_stride + (q + D) * 64];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[n][j], acc[n], 0, 0, 0);
            }
        }
    }
}

// D tile -> row-major [rows][ld] matrix; rows are MFMA rows (A's channels), columns B's channels.
__device__ __forceinline__ void store_tile(float* out, int ld, int mb, int nb, int ncols, const f32x16& acc, int lane) {
    const int col = 32 * nb + (lane & 31);
    if (col >= ncols) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(long long)row * ld + col] = acc[r];
    }
}

// conv0.weight [C1][C] and conv0.bias (column C of the x|1 operand), row block mb.
__device__ __forceinline__ void wgrad_conv0(const BwdParams& p, const float* ops, float* pw, const GradLayout& GL, int dz0_off, int xb_off,
                                            int mb, int n_oct, int lane) {
    f32x16 acc;
    // rows of the x|1 block beyond C are never written by kernel A: mask them out of the B operand
    const f32x4* a4 = reinterpret_cast<const f32x4*>(ops + dz0_off) + (long long)mb * 32 * 64 + lane;
    const f32x4* b4 = reinterpret_cast<const f32x4*>(ops + xb_off) + lane;
    const bool live = (lane & 31) <= p.cl.C;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int q = 0; q < n_oct; ++q) {
        const f32x4 a = a4[q * 64];
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (live) bv = b4[q * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[j], acc, 0, 0, 0);
    }
    const int col = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (col < p.cl.C) pw[GL.w0() + row * p.cl.C + col] = acc[r];
        else if (col == p.cl.C) pw[GL.b0() + row] = acc[r];
    }
}

template <int C1, int kC2, int kC3>
__global__ __launch_bounds__(512, 2) void encoder_bwd_wgrad_kernel(const BwdParams p) {
    constexpr int MB1 = C1 / 32;
    constexpr OpsLayout OL{C1 / 32, kC2 / 32, kC3 / 32};
    const GradLayout GL{p.cl.C, C1, kC2, kC3};
    extern __shared__ __attribute__((aligned(16))) f32x4 s_h1[];      // [4][n_oct][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = p.parts;
    for (int item = blockIdx.x; item < p.cl.B * P; item += gridDim.x) {
        const int b = item / P, part = item - b * P;
        const float* ops = p.ops + (long long)b * OL.total();
        float* pw = p.pw + (long long)b * p.pw_stride;
        const int n_tiles = (p.n_act[b] + 31) / 32, n_oct = n_tiles * 4;
        if (p.tile_mode && part == 0 && tid < 2 * kC2) {   // norm1 gradients: fixed-order sum over the cloud's tiles (one partial per tile)
            const float* n1 = p.n1part + (long long)b * 8 * kC2 * 2;
            float acc = 0.0f;
            for (int t = 0; t < n_tiles; ++t) acc = acc + n1[t * kC2 * 2 + tid];
            pw[((tid & 1) ? GL.be1() : GL.g1()) + (tid >> 1)] = acc;
        }
        constexpr bool kStandard = kC2 == 128 && kC3 == 256 && (C1 == 64 || C1 == 128);   // the 8-wave mapping below assumes these
        if (P > 1 || !kStandard) {
            // Small batch (or a shape without a hand-laid wave mapping): the cloud's MB3 MB2 + MB2 MB1 + MB1 output blocks are dealt
            // one at a time over the 8 P waves of its P workgroups (operands straight from L2; every block is one chain of
            // n_oct * 4 MFMAs).
            constexpr int MB2 = kC2 / 32, MB3 = kC3 / 32;
            const int n_tasks = MB3 * MB2 + MB2 * MB1 + MB1;
            for (int t = part * 8 + wave; t < n_tasks; t += 8 * P) {
                f32x16 acc[1];
                if (t < MB3 * MB2) {
                    wgrad_blocks<1>(ops + OL.dz2(), reinterpret_cast<const f32x4*>(ops + OL.h1()), 32 * 64, t / MB2, t % MB2, n_oct, lane, acc);
                    store_tile(pw + GL.w2(), kC2, t / MB2, t % MB2, kC2, acc[0], lane);
                } else if (t < MB3 * MB2 + MB2 * MB1) {
                    const int u = t - MB3 * MB2;
                    wgrad_blocks<1>(ops + OL.dz1(), reinterpret_cast<const f32x4*>(ops + OL.h0()), 32 * 64, u / MB1, u % MB1, n_oct, lane, acc);
                    store_tile(pw + GL.w1(), C1, u / MB1, u % MB1, C1, acc[0], lane);
                } else {
                    wgrad_conv0(p, ops, pw, GL, OL.dz0(), OL.xb(), t - MB3 * MB2 - MB2 * MB1, n_oct, lane);
                }
            }
            continue;
        }
        if constexpr (kStandard) {
        {   // conv2.weight [256][128]: wave w owns row block w, all 4 column blocks.  Every wave contracts against the whole
            // of h1, so the cloud's h1 operand (4 blocks x n_oct KB) is staged in LDS once instead of being fetched from L2
            // by each of the 8 waves.
            __syncthreads();
            const f32x4* h1g = reinterpret_cast<const f32x4*>(ops + OL.h1());
            for (int i = tid; i < 4 * n_oct * 64; i += 512) {
                const int nb = i / (n_oct * 64), r = i - nb * (n_oct * 64);
                s_h1[i] = h1g[nb * 32 * 64 + r];
            }
            __syncthreads();
            f32x16 acc[4];
            wgrad_blocks<4>(ops + OL.dz2(), (const f32x4*)s_h1, n_oct * 64, wave, 0, n_oct, lane, acc);
#pragma unroll
            for (int n = 0; n < 4; ++n) store_tile(pw + GL.w2(), kC2, wave, n, kC2, acc[n], lane);
        }
        {   // conv1.weight [128][C1]: 4 x MB1 blocks over 8 waves
            constexpr int NB = MB1 / 2;     // column blocks per wave (1 or 2)
            f32x16 acc[NB];
            const int mb = wave >> 1, nb0 = (wave & 1) * NB;
            wgrad_blocks<NB>(ops + OL.dz1(), reinterpret_cast<const f32x4*>(ops + OL.h0()), 32 * 64, mb, nb0, n_oct, lane, acc);
#pragma unroll
            for (int n = 0; n < NB; ++n) store_tile(pw + GL.w1(), C1, mb, nb0 + n, C1, acc[n], lane);
        }
        if (wave < MB1) wgrad_conv0(p, ops, pw, GL, OL.dz0(), OL.xb(), wave, n_oct, lane);
        }
    }
}

#if PCRL_BWD_MODE == 0
// ---- reduce: grads[i] = sum_b pw[b][i], fixed order ---------------------------------------------
// HBM/L2-bound (B x n floats read once).  A 1024-thread block owns 64 consecutive elements; thread (g, c) sums the
// clouds b = g, g + 16, ... of element c with four loads in flight, then the 16 partials are added in g order.
__global__ __launch_bounds__(1024) void encoder_bwd_reduce_kernel(const float* __restrict__ pw, int B, int n, float* __restrict__ grads) {
    __shared__ float s_part[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + c;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    if (i < n) {
        int b = g;
        for (; b + 48 < B; b += 64) {
            p0 = p0 + pw[(long long)(b + 0) * n + i]; p1 = p1 + pw[(long long)(b + 16) * n + i];
            p2 = p2 + pw[(long long)(b + 32) * n + i]; p3 = p3 + pw[(long long)(b + 48) * n + i];
        }
        for (; b < B; b += 16) p0 = p0 + pw[(long long)b * n + i];
    }
    s_part[g][c] = (p0 + p1) + (p2 + p3);
    __syncthreads();
    if (g == 0 && i < n) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = acc + s_part[k][c];
        grads[i] = acc;
    }
}

#endif

static size_t bwd_lds_bytes_cloud(int T0, int C1, int kC2, int kC3) {
    return sizeof(ChanSrc) * PCRL_MAX_CHANNELS + 4 * (size_t)kC3 * 2 + 4 * (size_t)kC3 + 32 + kC3 + 4 * (size_t)kC3 +
           sizeof(float) * (2 * kC2 + 2 * kC3 + C1 + (size_t)(C1 / 32) * T0 * 64 + 8 * kC2 * 2 + 2 * (size_t)kC3 + 2 * (size_t)kC3 + 3 * (size_t)kC3 + (size_t)kC3 * kC2);
}

static size_t bwd_lds_bytes_tile(int T0, int C1, int kC2, int kC3) {
    return sizeof(ChanSrc) * PCRL_MAX_CHANNELS + 8 * (size_t)kTileTabBytes + 4 * (size_t)(kMaxTileModeClouds + 8) +
           sizeof(float) * (2 * kC2 + 2 * kC3 + C1 + (size_t)(C1 / 32) * T0 * 64 + (size_t)kC3 * kC2);
}

int encoder_bwd_points_nw4_f32(int T0, int c1, int c2, int c3, const BwdParams& p, size_t lds, hipStream_t stream);

template <int T0, int C1, int C2, int C3, bool BF16, bool SPLIT = false>
static int launch_bwd(const BwdParams& p, int grid, hipStream_t stream) {
    if (p.tile_mode) {
        // small batch: prep (tables) -> one wave per tile, dealt over every CU
        const int nW = (p.cl.N + 31) / 32;
        auto prep = encoder_bwd_prep_kernel<T0, C1, C2, C3>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(prep), 2 * sizeof(unsigned) * (size_t)kBitmapMaxWords)) return rc;
        hipLaunchKernelGGL(prep, dim3(p.cl.B), dim3(256), 2 * sizeof(unsigned) * (size_t)nW, stream, p);
        PCRL_CHECK_LAUNCH("encoder_bwd_prep_kernel");
        const size_t lds = bwd_lds_bytes_tile(T0, C1, C2, C3);
        // at most C3 / 32 tiles per cloud: when even that many fit one per SIMD, take the four-wave build (no register spills)
        const bool four = !BF16 && !SPLIT && (long long)p.cl.B * (C3 / 32) <= 4ll * num_cus();
        if (four) {            // instantiated in its own translation unit (encoder_bwd_f32_nw4.hip) so that the two builds compile in parallel
            if (int rc = encoder_bwd_points_nw4_f32(T0, C1, C2, C3, p, lds, stream)) return rc;
        } else {
            auto kern = encoder_bwd_points_kernel<T0, C1, C2, C3, BF16, SPLIT>;
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
            hipLaunchKernelGGL(kern, dim3(num_cus()), dim3(512), lds, stream, p);
        }
        PCRL_CHECK_LAUNCH("encoder_bwd_points_kernel");
    } else {
        const size_t lds = bwd_lds_bytes_cloud(T0, C1, C2, C3);
        auto kern = encoder_bwd_points_cloud_kernel<T0, C1, C2, C3, BF16, SPLIT>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
        PCRL_CHECK_LAUNCH("encoder_bwd_points_cloud_kernel");
    }
    constexpr size_t wgrad_lds = 4 * 32 * 64 * sizeof(f32x4);        // h1 operand of one cloud: 128 KB at 256 active points
    auto wgrad = encoder_bwd_wgrad_kernel<C1, C2, C3>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(wgrad), wgrad_lds)) return rc;
    hipLaunchKernelGGL(wgrad, dim3(grid * p.parts), dim3(512), wgrad_lds, stream, p);
    PCRL_CHECK_LAUNCH("encoder_bwd_wgrad_kernel");
    return PCRL_OK;
}

struct BwdWorkspace {
    size_t ops, xs, pw, nact, flag, act, slot, dx, pt, n1part, total;
};
static BwdWorkspace bwd_workspace(int B, int C, int C1, int kC2, int kC3) {
    const OpsLayout OL{C1 / 32, kC2 / 32, kC3 / 32};
    const GradLayout GL{C, C1, kC2, kC3};
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    BwdWorkspace w;
    w.ops = 0;
    w.xs = al(w.ops + sizeof(float) * (size_t)B * OL.total());
    w.pw = al(w.xs + sizeof(float) * (size_t)B * kXsFloats);
    w.nact = al(w.pw + sizeof(float) * (size_t)B * GL.total());
    w.flag = al(w.nact + sizeof(int) * (size_t)B);
    w.act = al(w.flag + sizeof(int) * (size_t)B);
    w.slot = al(w.act + sizeof(int) * (size_t)B * kC3);
    w.dx = al(w.slot + (size_t)B * kC3);
    w.pt = al(w.dx + sizeof(float) * (size_t)B * kC3);
    w.n1part = al(w.pt + sizeof(float2) * (size_t)B * kC3);
    w.total = al(w.n1part + sizeof(float) * (size_t)B * 8 * kC2 * 2);
    return w;
}

#if PCRL_BWD_MODE == 3
// Mode 3 = this unit only: the four-wave (spill-free) fp32 tile kernel for every supported shape.
int encoder_bwd_points_nw4_f32(int T0, int c1, int c2, int c3, const BwdParams& p, size_t lds, hipStream_t stream) {
    int rc = PCRL_E_ARG;
#define PCRL_BWD_CASE(T0_, C1_, C2_, C3_)                                                                     \
    if (T0 == T0_ && c1 == C1_ && c2 == C2_ && c3 == C3_) {                                                  \
        auto kern = encoder_bwd_points_kernel<T0_, C1_, C2_, C3_, false, false, 4>;                          \
        rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);                                   \
        if (rc == PCRL_OK) hipLaunchKernelGGL(kern, dim3(num_cus()), dim3(256), lds, stream, p);             \
    }
    PCRL_BWD_CASE(2, 64, 128, 256) PCRL_BWD_CASE(3, 64, 128, 256) PCRL_BWD_CASE(4, 64, 128, 256) PCRL_BWD_CASE(5, 64, 128, 256)
    PCRL_BWD_CASE(2, 128, 128, 256) PCRL_BWD_CASE(3, 128, 128, 256) PCRL_BWD_CASE(4, 128, 128, 256) PCRL_BWD_CASE(5, 128, 128, 256)
    PCRL_BWD_CASE(2, 32, 64, 128) PCRL_BWD_CASE(3, 32, 64, 128) PCRL_BWD_CASE(4, 32, 64, 128) PCRL_BWD_CASE(5, 32, 64, 128)
#undef PCRL_BWD_CASE
    return rc;
}
#elif PCRL_BWD_MODE == 4
// Mode 4 = encoder_bwd_gram_f32.hip: only the declarations above; the Gram-form kernels follow in encoder_bwd_gram.h.
#else
// The kernels of this unit's mode for every supported shape.
int PCRL_BWD_LAUNCH_NAME(int T0, int c1, int c2, int c3, const BwdParams& p, int grid, hipStream_t st) {
    constexpr bool kBf16 = PCRL_BWD_MODE == 1, kSplit = PCRL_BWD_MODE == 2;
    int rc = PCRL_E_ARG;
#define PCRL_BWD_CASE(T0_, C1_, C2_, C3_) \
    if (T0 == T0_ && c1 == C1_ && c2 == C2_ && c3 == C3_) rc = launch_bwd<T0_, C1_, C2_, C3_, kBf16, kSplit>(p, grid, st);
    PCRL_BWD_CASE(2, 64, 128, 256) PCRL_BWD_CASE(3, 64, 128, 256) PCRL_BWD_CASE(4, 64, 128, 256) PCRL_BWD_CASE(5, 64, 128, 256)
    PCRL_BWD_CASE(2, 128, 128, 256) PCRL_BWD_CASE(3, 128, 128, 256) PCRL_BWD_CASE(4, 128, 128, 256) PCRL_BWD_CASE(5, 128, 128, 256)
    PCRL_BWD_CASE(2, 32, 64, 128) PCRL_BWD_CASE(3, 32, 64, 128) PCRL_BWD_CASE(4, 32, 64, 128) PCRL_BWD_CASE(5, 32, 64, 128)
#undef PCRL_BWD_CASE
    return rc;
}
#endif

int encoder_bwd_launch_f32(int T0, int c1, int c2, int c3, const BwdParams& p, int grid, hipStream_t st);
int encoder_bwd_launch_bf16(int T0, int c1, int c2, int c3, const BwdParams& p, int grid, hipStream_t st);
int encoder_bwd_launch_split(int T0, int c1, int c2, int c3, const BwdParams& p, int grid, hipStream_t st);

}  // namespace pcrl

#if PCRL_BWD_MODE == 0
#include "encoder_bwd_gram.h"
using namespace pcrl;

extern "C" int pcrl_encoder_num_grads(int32_t c_in, int32_t c1, int32_t c2, int32_t c3, size_t* n) {
    size_t dummy;
    if (int rc = pcrl_encoder_packed_bytes(c_in, c1, c2, c3, &dummy)) return rc;
    if (!n) return fail(PCRL_E_ARG, "n is NULL");
    *n = (size_t)GradLayout{c_in, c1, c2, c3}.total();
    return PCRL_OK;
}

extern "C" int pcrl_encoder_bwd_workspace_bytes(int32_t B, int32_t c_in, int32_t c1, int32_t c2, int32_t c3, size_t* bytes) {
    size_t dummy;
    if (int rc = pcrl_encoder_packed_bytes(c_in, c1, c2, c3, &dummy)) return rc;
    if (!bytes || B < 0) return fail(PCRL_E_ARG, "bad arguments");
    *bytes = std::max(bwd_workspace(B, c_in, c1, c2, c3).total, bwdg_workspace(B, c_in, c1, c2, c3).total);
    return PCRL_OK;
}

// pcrl_encoder_bwd_set_fused: 0 = the points / wgrad / reduce launches always, 1 (default) = the team kernel of encoder_bwd_fused.h for
// launches of at most two tiles per CU (where it is faster: up to 64 clouds on 256 CUs), 2 = the team kernel wherever it is built
// (process-wide, like pcrl_gemm_set_tile64_min: an atomic that every later launch of any host thread reads once)
static std::atomic<int> g_bwd_fused{1};
extern "C" int pcrl_encoder_bwd_set_fused(int32_t mode) {
    if (mode < 0 || mode > 2) return fail(PCRL_E_ARG, "pcrl_encoder_bwd_set_fused: mode 0, 1 or 2");
    g_bwd_fused.store(mode, std::memory_order_relaxed);
    return PCRL_OK;
}

// pcrl_encoder_bwd_last_schedule: what the last backward of this host thread launched
static thread_local int t_bwd_schedule = 0;
extern "C" int pcrl_encoder_bwd_last_schedule(void) { return t_bwd_schedule; }

// pcrl_encoder_bwd_attach_colsum: jobs handed over for the NEXT backward of this host thread
static thread_local pcrl_colsum_job t_colsum_jobs[kColsumJobs];
static thread_local int t_colsum_n = 0;

extern "C" int pcrl_encoder_bwd_attach_colsum(const pcrl_colsum_job* jobs, int32_t n) {
    if (n == 0) { t_colsum_n = 0; return PCRL_OK; }
    if (!jobs || n < 0 || n > kColsumJobs) return fail(PCRL_E_ARG, "colsum jobs: 0 <= n <= %d", kColsumJobs);
    ColsumParams probe;
    if (colsum_fill(jobs, n, probe) < 0) return fail(PCRL_E_ARG, "bad colsum job");
    for (int i = 0; i < n; ++i) t_colsum_jobs[i] = jobs[i];
    t_colsum_n = n;
    return PCRL_OK;
}

// pcrl_encoder_bwd_attach_ln_bwd: a LayerNorm backward handed over for the NEXT prep launch of this host thread
static thread_local LnBwdParams t_ln_job;
static thread_local size_t t_ln_ws_bytes = 0;
static thread_local bool t_ln_on = false;

extern "C" int pcrl_encoder_bwd_attach_ln_bwd(const float* dy0, const float* dy1, int64_t lddy, const float* xhat, const float* rstd,
                                              const float* gamma, int32_t M, int32_t F, float* dx, int64_t lddx,
                                              void* workspace, size_t workspace_bytes) {
    if (!dy0 && M == 0) { t_ln_on = false; return PCRL_OK; }
    if (!dy0 || !xhat || !rstd || !gamma || !dx) return fail(PCRL_E_ARG, "NULL argument");
    if (F < 1 || F > 256 || M < 1) return fail(PCRL_E_ARG, "LayerNorm rows: 1 <= F <= 256, M >= 1");
    const size_t need = sizeof(float) * (size_t)((M + 3) / 4) * 2 * F;
    if (!workspace || workspace_bytes < need) return fail(PCRL_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
    t_ln_job = LnBwdParams{dy0, dy1, lddy, xhat, rstd, gamma, M, F, dx, lddx, static_cast<float*>(workspace)};
    t_ln_ws_bytes = workspace_bytes;
    t_ln_on = true;
    return PCRL_OK;
}

extern "C" int pcrl_layernorm_rows_bwd_partials_f32(const float* dy0, const float* dy1, int64_t lddy, const float* xhat, const float* rstd,
                                                    const float* gamma, int32_t M, int32_t F, float* dx, int64_t lddx,
                                                    void* workspace, size_t workspace_bytes, void* stream);

static int encoder_bwd_impl(int mode /* 0 fp32, 1 bf16, 2 split */, const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                            const pcrl_encoder_weights* w, const void* packed,
                            const int32_t* argmax, const float* grad_pooled, const float* pooled,
                            float* grads, int32_t* n_active,
                            void* workspace, size_t workspace_bytes, void* stream, int phase = 0) {
    // the attached column-sum jobs belong to the call that launches the reduce kernel (not to the prepare half); whatever path this
    // call takes, they are consumed by it
    pcrl_colsum_job cs_jobs[kColsumJobs];
    int cs_n = 0;
    if (phase != 1) {
        cs_n = t_colsum_n;
        for (int i = 0; i < cs_n; ++i) cs_jobs[i] = t_colsum_jobs[i];
        t_colsum_n = 0;
    }
    // an attached LayerNorm backward belongs to the call that launches prep (phase 0 or 1): it rides there when that is the Gram form's prep
    // launch for c3 <= 256, and is a launch of its own in front of everything otherwise -- whatever path this call takes, it is consumed
    LnBwdParams ln_job{};
    bool ln_on = false;
    if (phase != 2 && t_ln_on) { ln_job = t_ln_job; ln_on = true; t_ln_on = false; }
    auto ln_alone = [&]() -> int {
        return pcrl_layernorm_rows_bwd_partials_f32(ln_job.dy0, ln_job.dy1, ln_job.lddy, ln_job.xhat, ln_job.rstd, ln_job.gamma, ln_job.M, ln_job.F,
                                                    ln_job.dx, ln_job.lddx, ln_job.part, t_ln_ws_bytes, stream);
    };
    if (!clouds || !w || !packed || !argmax || (phase != 1 && (!grad_pooled || !grads))) return fail(PCRL_E_ARG, "NULL argument");
    size_t need;
    if (int rc = pcrl_encoder_packed_bytes(w->c_in, w->c1, w->c2, w->c3, &need)) return rc;
    BwdParams p{};
    p.cs_blocks = cs_n ? colsum_fill(cs_jobs, cs_n, p.cs) : 0;
    if (int rc = fill_cloud_params(clouds, aug, w->c_in, &p.cl)) return rc;
    const GradLayout GL{w->c_in, w->c1, w->c2, w->c3};
    hipStream_t st = (hipStream_t)stream;
    if (p.cl.B == 0) {
        if (ln_on) { if (int rc = ln_alone()) return rc; }
        if (phase != 1) PCRL_CHECK_HIP(hipMemsetAsync(grads, 0, sizeof(float) * GL.total(), st));
        return cs_n ? pcrl_colsum_jobs_f32(cs_jobs, cs_n, stream) : PCRL_OK;
    }
    const BwdWorkspace ws = bwd_workspace(p.cl.B, w->c_in, w->c1, w->c2, w->c3);
    const BwdgWorkspace wg = bwdg_workspace(p.cl.B, w->c_in, w->c1, w->c2, w->c3);
    const size_t ws_need = std::max(ws.total, wg.total);
    if (!workspace || workspace_bytes < ws_need) return fail(PCRL_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws_need);
    char* base = static_cast<char*>(workspace);
    p.eps = w->eps; p.packed = static_cast<const float*>(packed); p.argmax = argmax; p.gpool = grad_pooled; p.pooled = pooled;
    p.grads = grads;
    p.pw_stride = GL.total();
    if (p.cl.N > 32 * kBitmapMaxWords) return fail(PCRL_E_ARG, "encoder backward: N = %d > %d points per cloud", p.cl.N, 32 * kBitmapMaxWords);
    {
        // Gram form (encoder_bwd_gram.h): fp32 and split-precision arithmetic, every batch the tile list covers, when the forward's
        // pooled values are given (the agents always pass them).  A call WITHOUT pooled values keeps the round-2 kernels below (a dense
        // search for the owned channels; every mode): tests/test_encoder_bwd_gpu.py is their main caller.
        // bf16 mode: the same fp32 Gram-form backward at the bf16 forward's argmax / pooled values.  The
        // recompute of the two lower layers is then fp32 instead of the forward's bf16 contractions: the gradient of the fp32
        // function at the forward's routing, as close to the bf16 function's straight-through gradient as that one's own bf16
        // data-gradient GEMMs were (tests: the same 3e-2 bounds), and 12 % faster at K2's 512 x 1 200 clouds (380 -> 334 us).
        if (p.cl.B <= kMaxTileModeClouds && w->w2 && pooled) {
            if (ln_on && w->c3 <= 256) { p.ln = ln_job; p.ln_blocks = (ln_job.M + 3) / 4; }
            else if (ln_on) { if (int rc = ln_alone()) return rc; }
            ln_on = false;
            p.ops = reinterpret_cast<float*>(base + wg.ops); p.pw = reinterpret_cast<float*>(base + wg.pw);
            p.n_act = reinterpret_cast<int*>(base + wg.nact); p.act = reinterpret_cast<int*>(base + wg.act);
            p.slot = reinterpret_cast<unsigned char*>(base + wg.slot); p.own = reinterpret_cast<unsigned*>(base + wg.own);
            p.own_chan = reinterpret_cast<unsigned char*>(base + wg.own_chan); p.ptc = reinterpret_cast<float4*>(base + wg.ptc);
            p.chc = reinterpret_cast<float*>(base + wg.chc); p.n1part = reinterpret_cast<float*>(base + wg.n1part);
            p.gvu = reinterpret_cast<float*>(base + wg.gvu); p.mimg = reinterpret_cast<float*>(base + wg.mimg);
            p.wgrows = reinterpret_cast<float*>(base + wg.wgrows); p.n_items = reinterpret_cast<int*>(base + wg.nitems);
            p.own_pack = reinterpret_cast<unsigned long long*>(base + wg.own_pack);
            p.srows = p.ops;          // the team kernel writes no operand pieces: their region holds the sparse rows of dW2
            p.fused = g_bwd_fused.load(std::memory_order_relaxed); p.fused_rows = wg.fused_rows;
            p.schedule_out = &t_bwd_schedule; t_bwd_schedule = 2;
            p.w2 = w->w2;
            p.pw_stride = GL.total() + GramExtra{w->c2}.total();
            p.tile_mode = 1;
            p.phase = phase;
            p.parts = 1;
            while (p.parts < 8 && 2 * p.parts * p.cl.B <= num_cus()) p.parts *= 2;
            const int T0g = (p.cl.C + 1) / 2;
            const int rcg = mode == 2 ? encoder_bwdg_launch_split(T0g, w->c1, w->c2, w->c3, p, st)
                                      : encoder_bwdg_launch_f32(T0g, w->c1, w->c2, w->c3, p, st);
            if (rcg == PCRL_E_ARG) return fail(PCRL_E_ARG, "no fused kernel for C=%d (supported: 3..10 channels)", p.cl.C);
            if (rcg) return rcg;
            if (n_active && phase != 1) PCRL_CHECK_HIP(hipMemcpyAsync(n_active, p.n_act, sizeof(int) * p.cl.B, hipMemcpyDeviceToDevice, st));
            return PCRL_OK;
        }
        if (phase != 0)
            return fail(PCRL_E_ARG, "the two-call backward exists in Gram form only: fp32, the forward's pooled values given, at most %d clouds",
                        kMaxTileModeClouds);
        if (w->c3 > 256)
            return fail(PCRL_E_ARG, "mlp_spec=[%d,%d,%d]: the wide last layer is built in Gram form only: fp32, the forward's pooled values "
                                    "given, at most %d clouds", w->c1, w->c2, w->c3, kMaxTileModeClouds);
    }
    t_bwd_schedule = 1;
    if (ln_on) { if (int rc = ln_alone()) return rc; }
    p.ops = reinterpret_cast<float*>(base + ws.ops); p.xs = reinterpret_cast<float*>(base + ws.xs);
    p.pw = reinterpret_cast<float*>(base + ws.pw); p.n_act = reinterpret_cast<int*>(base + ws.nact);
    p.flag = reinterpret_cast<int*>(base + ws.flag); p.act = reinterpret_cast<int*>(base + ws.act);
    p.slot = reinterpret_cast<unsigned char*>(base + ws.slot); p.dx = reinterpret_cast<float*>(base + ws.dx);
    p.pt = reinterpret_cast<float2*>(base + ws.pt); p.n1part = reinterpret_cast<float*>(base + ws.n1part);
    p.grads = grads;
    if (p.cl.N > 32 * kBitmapMaxWords) return fail(PCRL_E_ARG, "encoder backward: N = %d > %d points per cloud", p.cl.N, 32 * kBitmapMaxWords);
    {
        // measured (tools/bench_encoder.py, round 2): dealing single tiles over all SIMDs also wins for large
        // batches of the c1 = 64 shapes (B 256: 212 -> 207 us, B 1024: 822 -> 739 us); with c1 = 128 the two schedules tie
        // (946 vs 960 us at B 1024, N 1200) and the one-workgroup-per-cloud kernel stays
        // (the bf16 build's tile kernel writes whole operand pieces: 512 x 1200 clouds with c1 = 128, 384 us by clouds, 366 us by tiles)
        const bool automatic = p.cl.B < num_cus() || w->c1 <= 64 || mode == 1;
        p.tile_mode = automatic && p.cl.B <= kMaxTileModeClouds ? 1 : 0;
    }
    p.parts = 1;
    while (p.parts < 8 && 2 * p.parts * p.cl.B <= num_cus()) p.parts *= 2;

    const int grid = min(p.cl.B, num_cus());
    const int T0 = (p.cl.C + 1) / 2;
    int rc = PCRL_E_ARG;
    rc = mode == 1 ? encoder_bwd_launch_bf16(T0, w->c1, w->c2, w->c3, p, grid, st)
       : mode == 2 ? encoder_bwd_launch_split(T0, w->c1, w->c2, w->c3, p, grid, st)
                   : encoder_bwd_launch_f32(T0, w->c1, w->c2, w->c3, p, grid, st);
    if (rc == PCRL_E_ARG) return fail(PCRL_E_ARG, "no fused kernel for C=%d (supported: 3..10 channels)", p.cl.C);
    if (rc) return rc;
    const int n = GL.total();
    hipLaunchKernelGGL(encoder_bwd_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, p.pw, p.cl.B, n, grads);
    PCRL_CHECK_LAUNCH("encoder_bwd_reduce_kernel");
    if (cs_n) { if (int rc2 = pcrl_colsum_jobs_f32(cs_jobs, cs_n, stream)) return rc2; }      // the round-2 kernels: a launch of their own
    if (n_active) PCRL_CHECK_HIP(hipMemcpyAsync(n_active, p.n_act, sizeof(int) * p.cl.B, hipMemcpyDeviceToDevice, st));
    return PCRL_OK;
}

extern "C" int pcrl_encoder_bwd_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                    const pcrl_encoder_weights* w, const void* packed,
                                    const int32_t* argmax, const float* grad_pooled, const float* pooled,
                                    float* grads, int32_t* n_active,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_bwd_impl(0, clouds, aug, w, packed, argmax, grad_pooled, pooled, grads, n_active, workspace, workspace_bytes, stream);
}

extern "C" int pcrl_encoder_bwd_prepare_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                            const pcrl_encoder_weights* w, const void* packed,
                                            const int32_t* argmax, const float* pooled,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!pooled) return fail(PCRL_E_ARG, "pcrl_encoder_bwd_prepare_f32 needs the forward's pooled values");
    return encoder_bwd_impl(0, clouds, aug, w, packed, argmax, nullptr, pooled, nullptr, nullptr, workspace, workspace_bytes, stream, 1);
}

extern "C" int pcrl_encoder_bwd_prepared_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                             const pcrl_encoder_weights* w, const void* packed,
                                             const int32_t* argmax, const float* grad_pooled, const float* pooled,
                                             float* grads, int32_t* n_active,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_bwd_impl(0, clouds, aug, w, packed, argmax, grad_pooled, pooled, grads, n_active, workspace, workspace_bytes, stream, 2);
}

extern "C" int pcrl_encoder_bwd_bf16(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                     const pcrl_encoder_weights* w, const void* packed,
                                     const int32_t* argmax, const float* grad_pooled, const float* pooled,
                                     float* grads, int32_t* n_active,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_bwd_impl(1, clouds, aug, w, packed, argmax, grad_pooled, pooled, grads, n_active, workspace, workspace_bytes, stream);
}

extern "C" int pcrl_encoder_bwd_f32split(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                         const pcrl_encoder_weights* w, const void* packed,
                                         const int32_t* argmax, const float* grad_pooled, const float* pooled,
                                         float* grads, int32_t* n_active,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_bwd_impl(2, clouds, aug, w, packed, argmax, grad_pooled, pooled, grads, n_active, workspace, workspace_bytes, stream);
}
#ifdef PCRL_BWD_STAMPS
extern "C" int pcrl_debug_bwd_stamps(unsigned long long* host_out, int n_items) {
    PCRL_CHECK_HIP(hipDeviceSynchronize());
    PCRL_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pcrl::g_bwd_stamps), sizeof(unsigned long long) * 8 * (size_t)n_items));
    return PCRL_OK;
}
#endif
#endif  // PCRL_BWD_MODE == 0
