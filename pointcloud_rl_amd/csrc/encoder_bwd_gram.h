// PointNet encoder backward, Gram form (fp32) -- round 3.
//
// Same function as encoder_bwd_impl.h (autograd through pyrl/networks/backbones/pointnet.py:148-151 restricted to the <= c3 points
// per cloud the max-pool routes gradient to), different algebra for the last layer.  The max-pool hands dL/dy2 to ONE point
// per channel, so at a point p only the few channels own(p) it owns carry an upstream gradient dx_c = [y_c > 0] g_c gamma_c,
// and LayerNorm-2's backward is
//     dz2_p = rstd_p (dx_p - m1_p 1 - m2_p xhat2_p),   m1_p = sum_own dx_c / C3,   m2_p = sum_own dx_c xhat2_c / C3,
//     xhat2_p = rstd_p (W2 h1_p - mu_p 1).
// With  M = W2^T W2 [C2 x C2]  and  s = W2^T 1  (both part of the packed weight image, rebuilt with it):
//     mu_p = s.h1_p / C3,   var_p = h1_p.M h1_p / C3 - mu_p^2,
//     dH1_p = W2^T dz2_p = rstd_p [ sum_own dx_c W2[c,:] - m1_p s - m2_p rstd_p (M h1_p - mu_p s) ],
//     dW2  = sum_p dz2_p h1_p^T = S - 1 v^T + 1 u^T - W2 G,
//         S[c,:] = rstd_p(c) dx_c h1_p(c)   (one scaled row of h1 per channel),
//         v = sum_p rstd_p m1_p h1_p,   u = sum_p a_p mu_p h1_p,   G = sum_p a_p h1_p h1_p^T,   a_p = rstd_p^2 m2_p.
// Per 32-point tile that is ONE C2 -> C2 layer (M h1: 256 MFMAs at c2 = 128) instead of the conv2 recompute and the
// W2^T dz2 GEMM (512 + 512), the dz2 operand (128 of a tile's 360 scattered stores) is never written, and the per-cloud
// weight-gradient GEMM of the last layer shrinks from [C3 x C2] to the [C2 x C2] Gram matrix; W2 G is one small GEMM per launch.
// xhat2_c at a channel's own point comes from the 128-term dot product W2[c,:].h1_p -- no (y - beta) / gamma reconstruction, so no
// "lossy channel" fallback; the forward's pooled values (when given) decide y_c > 0 exactly as the forward saw it.
// Measured against fp32 autograd: 3e-7 ... 6e-7 of each tensor's largest entry (tests/test_encoder_bwd_gpu.py, same tolerance as
// before: 2e-5).
//
//   prep   (one 256-thread workgroup per cloud): active point list and slots as before (bitmap + prefix popcount), then the
//          channels grouped by slot (CSR: own / own_chan).
//   points (one wave per 32-point tile, tiles of all clouds dealt over the chip): x -> conv0 -> conv1 + LN1 (recompute, same
//          MFMA chains as the forward) -> q = M h1 -> mu, rstd2 -> loop over the owned channels -> dH1 -> LN1 backward -> dz1 ->
//          W1^T dz1 -> dz0.  Operand pieces written: x|1, h0, h1, dz1, dz0; per slot (a, rstd2 m1, a mu); per channel rstd2 dx.
//   wgrad  (per cloud): h1 staged in LDS; G, (v, u), S from it; dW1, dW0 from the pieces in L2.
//   reduce (fixed order over the clouds), finalize (dW2 = S - 1 v^T + 1 u^T - W2 G).
#pragma once

#ifndef PCRL_BWDG_DH0SPLIT
#define PCRL_BWDG_DH0SPLIT 1  // dH0 = W1^T dz1 as a three-term bf16 split in the exact-fp32 mode too (K1 stand-alone 138.5 -> 135.3 us)
#endif
#ifndef PCRL_BWDG_MSPLIT
#define PCRL_BWDG_MSPLIT 1    // q = Mc h1 on the bf16 matrix cores as a three-term split (round 4); 0: the fp32 MFMA chain of round 3
#endif

namespace pcrl {

// floats of the Gram image the points kernel stages in LDS: the three bf16 term images (3 x C2^2 / 2) or the fp32 image (C2^2)
__host__ __device__ constexpr int bwdg_m_lds_floats(int c2) { return PCRL_BWDG_MSPLIT ? 3 * c2 * c2 / 2 : c2 * c2; }

struct GramExtra {      // what a cloud's pw row holds behind the reference-ordered gradients
    int C2;
    __host__ __device__ constexpr int G() const { return 0; }
    __host__ __device__ constexpr int v() const { return C2 * C2; }
    __host__ __device__ constexpr int u() const { return C2 * C2 + C2; }
    __host__ __device__ constexpr int total() const { return C2 * C2 + 2 * C2; }
};

// Octets (8 active points) a block array has room for, tiles per cloud, and the integer type that holds a slot / a channel number:
// c3 <= 256 keeps round 2's sizes (32 octets, 8 tiles, one byte); the wide last layer (c3 = 1024) has up to c3 active points.
__host__ __device__ constexpr int bwdg_np(int c3) { return c3 > 256 ? c3 / 8 : 32; }
__host__ __device__ constexpr int bwdg_tpc(int c3) { return c3 > 256 ? c3 / 32 : 8; }
template <int kC3> struct BwdgIdx { typedef unsigned char type; };
template <> struct BwdgIdx<512> { typedef unsigned short type; };
template <> struct BwdgIdx<768> { typedef unsigned short type; };
template <> struct BwdgIdx<1024> { typedef unsigned short type; };

// Operand workspace of the Gram form: no dz2 array.
struct OpsLayoutG {
    int MB1, MB2, NP;
    __host__ __device__ constexpr int blk() const { return NP * kPiece; }
    __host__ __device__ constexpr int h1() const { return 0; }
    __host__ __device__ constexpr int dz1() const { return h1() + MB2 * blk(); }
    __host__ __device__ constexpr int h0() const { return dz1() + MB2 * blk(); }
    __host__ __device__ constexpr int dz0() const { return h0() + MB1 * blk(); }
    __host__ __device__ constexpr int xb() const { return dz0() + MB1 * blk(); }
    __host__ __device__ constexpr int total() const { return xb() + blk(); }
};

// Byte offset of accumulator slot (mb, r) inside an operand block array of NP octets per block, lane part excluded.
__host__ __device__ constexpr unsigned gop_off(int arr_floats, int mb, int r, int NP) {
    return 4u * (unsigned)(arr_floats + mb * NP * kPiece + ((r & 3) + 8 * (r >> 2)) * 4);
}

// ---- prep: active list, slots, channels grouped by slot; M = W2^T W2 and s = W2^T 1 --------------------------------------------
// Index of M[row][col] inside the operand-ordered image [C2/32][C2/8][64][4] a C2 -> C2 dense_layer_mfma streams (the contraction
// index runs through acc_chan, like the packed conv weights).
__host__ __device__ constexpr int gram_image_index(int C2, int row, int col) {
    const int a = col >> 5, bb = col & 31, h = (bb >> 2) & 1, R = 16 * a + (bb & 3) + 4 * (bb >> 3);
    return (((row >> 5) * (C2 / 8) + (R >> 2)) * 64 + h * 32 + (row & 31)) * 4 + (R & 3);
}

// One 16 x 16 tile of M per 256-thread block (the blocks behind the clouds of the prep launch): the two 16-column strips of W2
// it contracts are staged in LDS with every load in flight, then each thread forms one entry.  Tile row 0 also leaves s.
constexpr int kGramTileLds = 2 * 16 * 256 * 4;      // bytes: two [C3 <= 256][16] strips
template <int kC2, int kC3>
__device__ __forceinline__ void gram_tile(const float* __restrict__ w2, float* __restrict__ mimg, int tile, int tid, float* s_strip) {
    constexpr int TPR = kC2 / 16;                       // tiles per row of M
    const int ti = tile / TPR, tj = tile % TPR;
    float* s_a = s_strip;                               // [kC3][16]: W2[c][16 ti + i]
    float* s_b = s_strip + kC3 * 16;                    // [kC3][16]: W2[c][16 tj + j]
    {
        constexpr int N4 = kC3 * 16 / 4;                // float4 pieces per strip
        constexpr int PER = (N4 + 255) / 256;
        f32x4 ta[PER], tb[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + 256 * k, c = e >> 2, q4 = e & 3;
            if (e < N4) {
                ta[k] = *reinterpret_cast<const f32x4*>(w2 + c * kC2 + 16 * ti + 4 * q4);
                tb[k] = *reinterpret_cast<const f32x4*>(w2 + c * kC2 + 16 * tj + 4 * q4);
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + 256 * k;
            if (e < N4) {
                reinterpret_cast<f32x4*>(s_a)[e] = ta[k]; reinterpret_cast<f32x4*>(s_b)[e] = tb[k];
            }
        }
    }
    __syncthreads();
    // The CENTRED Gram matrix Mc = M - s s^T / C3 = sum_c (W2[c,i] - s_i / C3) (W2[c,j] - s_j / C3): LayerNorm-2's variance is then
    // var_p = h1_p . Mc h1_p / C3 with no "E z^2 - mu^2" subtraction (W2 columns with a large common component made that difference
    // cancel down to rounding noise, and the clamp at zero hid it), and Mc h1_p IS the (M h1_p - mu_p s) the data gradient needs.
    // The column sums first (32 threads, the order the uncentred build used for s), then every thread centres its share of the strips.
    __shared__ float s_mean[32];
    if (tid < 32) {
        const float* strip = tid < 16 ? s_a : s_b;
        float acc = 0.0f;
        for (int c = 0; c < kC3; ++c) acc = acc + strip[c * 16 + (tid & 15)];
        if (ti == 0 && tid >= 16) mimg[kC2 * kC2 + 16 * tj + (tid & 15)] = acc;      // s[16 tj + .] = column sum of W2
        s_mean[tid] = acc / (float)kC3;
    }
    __syncthreads();
    for (int e = tid; e < 2 * kC3 * 16; e += 256) s_strip[e] = s_strip[e] - s_mean[(e >= kC3 * 16 ? 16 : 0) + (e & 15)];
    __syncthreads();
    const int i = tid >> 4, j = tid & 15;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll 4
    for (int c = 0; c < kC3; c += 4) {
        a0 = __builtin_fmaf(s_a[(c + 0) * 16 + i], s_b[(c + 0) * 16 + j], a0);
        a1 = __builtin_fmaf(s_a[(c + 1) * 16 + i], s_b[(c + 1) * 16 + j], a1);
        a2 = __builtin_fmaf(s_a[(c + 2) * 16 + i], s_b[(c + 2) * 16 + j], a2);
        a3 = __builtin_fmaf(s_a[(c + 3) * 16 + i], s_b[(c + 3) * 16 + j], a3);
    }
    const float v = (a0 + a1) + (a2 + a3);
    mimg[gram_image_index(kC2, 16 * ti + i, 16 * tj + j)] = v;
#if PCRL_BWDG_MSPLIT
    // The same entry as three bf16 terms (truncation: hi + mid + lo = v exactly), in the A-operand order of v_mfma_f32_32x32x16_bf16 that
    // dense_layer_split streams (the order of the pack kernel's split images: [row block][16-channel group][lane][8]): q = Mc h1 then
    // costs 192 bf16 MFMAs of 32 cycles instead of 256 fp32 ones of 64, to ~3 x 2^-24 of |Mc||h1| per product (encoder_common.h).
    {
        const int row = 16 * ti + i, col = 16 * tj + j;
        const int a = col >> 5, bb = col & 31, h = (bb >> 2) & 1, R = 16 * a + (bb & 3) + 4 * (bb >> 3);     // col = acc_chan(R, h)
        const int e = ((((row >> 5) * (kC2 / 16) + (R >> 3)) * 64 + h * 32 + (row & 31)) << 3) + (R & 7);
        unsigned short* img = reinterpret_cast<unsigned short*>(mimg + kC2 * kC2 + kC2);
        float x = v;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned top = f2u(x) & 0xFFFF0000u;
            img[k * kC2 * kC2 + e] = (unsigned short)(top >> 16);
            x = x - u2f(top);
        }
    }
#endif
}

template <int C1, int kC2, int kC3>
__global__ __launch_bounds__(256) void encoder_bwdg_prep_kernel(const BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned s_words[];   // [nW] bitmap, then [nW] exclusive prefix popcounts
    __shared__ int s_scan[256];
    __shared__ __attribute__((aligned(4))) unsigned char s_slot[kC3];
    __shared__ int s_lastpc;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= p.cl.B + (kC2 / 16) * (kC2 / 16)) {      // a LayerNorm backward riding on this launch (pcrl_encoder_bwd_attach_ln_bwd)
        layernorm_rows_bwd_block(p.ln, (int)blockIdx.x - p.cl.B - (kC2 / 16) * (kC2 / 16), tid, reinterpret_cast<float*>(s_words));
        return;
    }
    if ((int)blockIdx.x >= p.cl.B) {          // the blocks behind the clouds build the Gram image
        gram_tile<kC2, kC3>(p.w2, p.mimg, (int)blockIdx.x - p.cl.B, tid, reinterpret_cast<float*>(s_words));
        return;
    }
    const GradLayout GL{p.cl.C, C1, kC2, kC3};
    const int nW = (p.cl.N + 31) >> 5;
    unsigned* s_pre = s_words + nW;
    {
        const int b = blockIdx.x;
        const bool chan = tid < kC3;
        int pc = chan ? p.argmax[(long long)b * kC3 + tid] : 0;
        // A channel the forward left at zero (ReLU-dead or non-positive maximum) passes no gradient: with the forward's pooled
        // values such channels are dropped here -- they would otherwise all sit on point 0 (torch's first-index rule) and make
        // that point's loop over its channels the longest of the launch.
        const bool live = chan && (!p.pooled || p.pooled[(long long)b * kC3 + tid] > 0.0f);
        __syncthreads();
        for (int w = tid; w < nW; w += 256) s_words[w] = 0u;
        __syncthreads();
        pc = pc < 0 ? 0 : (pc >= p.cl.N ? p.cl.N - 1 : pc);
        if (live) atomicOr(&s_words[pc >> 5], 1u << (pc & 31));
        __syncthreads();
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
        if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
        __syncthreads();
        int wave_base = 0;
#pragma unroll
        for (int w = 0; w < 3; ++w)
            if (w < (tid >> 6)) wave_base += s_scan[w];
        const int n_act = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
        int run = wave_base + incl - local;
        for (int k = 0; k < per; ++k)
            if (w0 + k < nW) { s_pre[w0 + k] = (unsigned)run; run += __popc(s_words[w0 + k]); }
        __syncthreads();
        const int slot = (int)s_pre[pc >> 5] + __popc(s_words[pc >> 5] & ((1u << (pc & 31)) - 1u));
        if (chan) {
            // 0xFF never names a slot that has channels: 256 active points would need 256 live channels
            s_slot[tid] = live ? (unsigned char)slot : (unsigned char)0xFF;
            p.slot[(long long)b * kC3 + tid] = live ? (unsigned char)slot : (unsigned char)0;
            if (live) {
                p.act[(long long)b * kC3 + slot] = pc;               // every channel of the point writes the same value
                if (slot == n_act - 1) s_lastpc = pc;
            } else {
                float* pw = p.pw + (long long)b * p.pw_stride;
                pw[GL.g2() + tid] = 0.0f;
                pw[GL.be2() + tid] = 0.0f;
                p.chc[(long long)b * kC3 + tid] = 0.0f;
            }
        }
        if (tid == 0) p.n_act[b] = n_act;
        __syncthreads();
        // thread = slot: the channels that name it (256-bit membership mask from a branch-free sweep over the slot bytes), their
        // number, an exclusive scan of the numbers over the slots, and the channels written out in ascending order
        unsigned mine[kC3 / 32];
        int cnt = 0;
        {
            const unsigned* slot_w = reinterpret_cast<const unsigned*>(s_slot);
            const unsigned me = (unsigned)tid * 0x01010101u;
#pragma unroll
            for (int j0 = 0; j0 < kC3 / 4; j0 += 16) {
                unsigned w[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = slot_w[j0 + j] ^ me;        // a zero byte marks a channel of this slot
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
            if (tid >= n_act) {
#pragma unroll
                for (int k = 0; k < kC3 / 32; ++k) mine[k] = 0u;
            }
#pragma unroll
            for (int k = 0; k < kC3 / 32; ++k) cnt += __popc(mine[k]);
        }
        int inc2 = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(inc2, d, 64);
            if ((tid & 63) >= d) inc2 += v;
        }
        if ((tid & 63) == 63) s_scan[4 + (tid >> 6)] = inc2;
        __syncthreads();
        int base2 = 0;
#pragma unroll
        for (int w = 0; w < 3; ++w)
            if (w < (tid >> 6)) base2 += s_scan[4 + w];
        int pos = base2 + inc2 - cnt;
        if (tid < kC3) p.own[(long long)b * kC3 + tid] = (unsigned)pos | ((unsigned)cnt << 16);
        unsigned long long first8 = ~0ull;      // the slot's first eight channels, one byte each (0xFF: none) -- the team kernel's prefetch
        int nth = 0;
#pragma unroll
        for (int k = 0; k < kC3 / 32; ++k) {
            unsigned m = mine[k];
            while (m) {
                const unsigned c = (unsigned)(32 * k + __builtin_ctz(m));
                p.own_chan[(long long)b * kC3 + pos] = (unsigned char)c;
                if (nth < 8) first8 = (first8 & ~(0xFFull << (8 * nth))) | ((unsigned long long)c << (8 * nth));
                m &= m - 1u;
                ++pos; ++nth;
            }
        }
        if (tid < kC3) p.own_pack[(long long)b * kC3 + tid] = first8;
        // the slots between n_act and the end of its tile repeat the last active point and own nothing: a tile's loads need no n_act
        if (tid >= n_act && tid < ((n_act + 31) & ~31) && tid < kC3) {
            p.act[(long long)b * kC3 + tid] = n_act > 0 ? s_lastpc : 0;
            p.own[(long long)b * kC3 + tid] = 0u;
        }
    }
}

#ifdef PCRL_BWDG_STAMPS
// Development build only (-DPCRL_BWDG_STAMPS on encoder_bwd_gram_f32.hip): shader-clock stamps at the phase boundaries of a tile's
// chain (points kernel) and of a wave's task list (wgrad kernel), read back with pcrl_debug_bwdg_stamps (tools/bwdg_stamps.py).
__device__ unsigned long long g_bwdg_stamps[16384][12];
__device__ unsigned long long g_bwdg_wstamps[4096][8];
#define PCRL_GSTAMP(k) do { if (lane == 0 && item < 16384) g_bwdg_stamps[item][k] = __builtin_readcyclecounter(); } while (0)
#define PCRL_WSTAMP(k) do { if (lane == 0 && item * 8 + wave < 4096) g_bwdg_wstamps[item * 8 + wave][k] = __builtin_readcyclecounter(); } while (0)
#else
#define PCRL_GSTAMP(k) do { } while (0)
#define PCRL_WSTAMP(k) do { } while (0)
#endif

// ---- points: one wave per tile ----------------------------------------------------------------------------------------
// SPLIT: the arithmetic of the forward that produced `argmax` (encoder_fwd_kernel<.., false, SPLIT>).  The conv1 recompute and the
// W1^T dz1 GEMM use that mode's MFMA chains exactly as encoder_bwd_impl.h does; conv2 is fp32-accurate in either mode, so the
// algebra above applies unchanged.  (The mixed-precision bf16 mode keeps the round-2 kernels: with conv2 and W2^T dz2 on the bf16
// matrix cores they cost less than this form's fp32 M h1 -- measured 157 vs 163 us at B = 256.)
// Four waves per workgroup = one per SIMD (512 registers each: nothing spills and xhat1 stays in registers).
constexpr int kBwdgWaves = 4;
#ifndef PCRL_BWDG_RING
#define PCRL_BWDG_RING 6      // 16-byte operand loads in flight per row block of a pair, in groups of 8 MFMAs (dense_layer_mfma_stream)
#endif
#ifndef PCRL_BWDG_WRING
#define PCRL_BWDG_WRING 4     // octets of dW1 operands in flight per wave of the wgrad kernel (wgrad_blocks_stream)
#endif
#ifndef PCRL_BWDG_W1NB
#define PCRL_BWDG_W1NB(MB1) ((MB1) >= 2 ? 2 : 1)
#endif

template <int T0, int C1, int kC2, int kC3, bool SPLIT>
__global__ __launch_bounds__(64 * kBwdgWaves, 1) void encoder_bwdg_points_kernel(const BwdParams p) {
    constexpr int NW = kBwdgWaves;
    constexpr PackedLayout L{T0, C1, kC2, kC3};
    constexpr int MB1 = C1 / 32, MB2 = kC2 / 32;
    constexpr OpsLayoutG OL{C1 / 32, kC2 / 32, bwdg_np(kC3)};
    const GradLayout GL{p.cl.C, C1, kC2, kC3};
    typedef typename BwdgIdx<kC3>::type idx_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    ChanSrc* s_desc = reinterpret_cast<ChanSrc*>(smem);
    int* s_tstart = reinterpret_cast<int*>(s_desc + PCRL_MAX_CHANNELS);           // [kMaxTileModeClouds + 8] tile prefix
    float* s_ln1 = reinterpret_cast<float*>(s_tstart + kMaxTileModeClouds + 8);
    float* s_ln2 = s_ln1 + 2 * kC2;
    float* s_b0 = s_ln2 + 2 * kC3;
    float* s_w0 = s_b0 + C1;
    float* s_sv = s_w0 + MB1 * T0 * 64;                                            // [kC2] column sums of W2
    float* s_m = s_sv + kC2;                                                       // [kC2 * kC2] M image
    float* s_tr = s_m + bwdg_m_lds_floats(kC2) + (threadIdx.x >> 6) * kTrFloats;   // this wave's transposition scratch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    {
#if PCRL_BWDG_MSPLIT
        stage_to_lds<64 * NW, 3 * kC2 * kC2 / 8>(reinterpret_cast<f32x4*>(s_m), reinterpret_cast<const f32x4*>(p.mimg + kC2 * kC2 + kC2), tid);
#else
        stage_to_lds<64 * NW, kC2 * kC2 / 4>(reinterpret_cast<f32x4*>(s_m), reinterpret_cast<const f32x4*>(p.mimg), tid);
#endif
        for (int i = tid; i < kC2; i += 64 * NW) s_sv[i] = p.mimg[kC2 * kC2 + i];
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
    const f32x4* s_mv = reinterpret_cast<const f32x4*>(s_m);
    const float2* s_ln2v = reinterpret_cast<const float2*>(s_ln2);

    const int n_items = s_tstart[p.cl.B];
    for (int item = wave * (int)gridDim.x + (int)blockIdx.x; item < n_items; item += NW * (int)gridDim.x) {
        int b, tile, n_act;
        {
            int lo = 0, hi = p.cl.B;             // largest b with s_tstart[b] <= item
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_tstart[mid] <= item) lo = mid; else hi = mid; }
            b = lo; tile = item - s_tstart[lo];
        }
        b = __builtin_amdgcn_readfirstlane(b);
        tile = __builtin_amdgcn_readfirstlane(tile);
        n_act = __builtin_amdgcn_readfirstlane(p.n_act[b]);
        const float* g_row = p.gpool + (long long)b * kC3;
        float* pw = p.pw + (long long)b * p.pw_stride;
        const __amdgpu_buffer_rsrc_t r_ops = make_rsrc(p.ops + (long long)b * OL.total(), 4u * (unsigned)OL.total());

        const int s = 32 * tile + l31;
        const bool valid = s < n_act;
        // c3 <= 256: the prep launch pads a cloud's last tile with copies of its last active point that own nothing and packs a slot's first
        // eight channels into one word -- the three loads below need no n_act, and the loop over the owned channels no index load in front of
        // its W2 row loads (points kernel, same box: K1 68.5 -> 67.4 us, K3's share 49.4 -> 48.5)
        int pidx;
        unsigned own_w;
        unsigned long long own_pk = 0ull;
        if constexpr (kC3 <= 256) {
            pidx = p.act[(long long)b * kC3 + s];
            own_w = p.own[(long long)b * kC3 + s];
            own_pk = p.own_pack[(long long)b * kC3 + s];
        } else {
            pidx = p.act[(long long)b * kC3 + (valid ? s : n_act - 1)];
            own_w = valid ? p.own[(long long)b * kC3 + s] : 0u;
        }
        // lane-dependent byte offset of an operand element: octet q = s >> 3, k-lane (s >> 2) & 1, k-slot s & 3
        const unsigned lane_off = 4u * (unsigned)(((s >> 3) * 64 + ((s >> 2) & 1) * 32 + 4 * half) * 4 + (s & 3));
        const unsigned tile_bytes = 4096u * (unsigned)tile;       // four octets of 1 KB per tile and block

        PCRL_GSTAMP(0);
        const f32x16 x = load_point<T0>(p.cl, s_desc, b, pidx);
        // The operand pieces the wgrad kernel reads (x | 1, h0, h1, dz1, dz0) leave BEHIND the loads of the phase that follows
        // their values (vmcnt counts loads and stores in one order: a weight load issued after a batch of stores waits for them),
        // and as whole 1 KB pieces (store_block_pieces).  Points kernel, K1: 91.5 -> 84.8 us; 512 x 8192: 331 -> 315 us whole backward.
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
            }
        }
        PCRL_GSTAMP(1);
        // ---- conv1 + LN: xhat1 and h1 stay in registers ---------------------------------------
        f32x16 a1[MB2];
        if (SPLIT)      // the split-precision forward's arithmetic: the recompute is bit-identical to that forward
            dense_layer_split<MB2, C1 / 16>(
                a1, [&](int k, int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1s(k) + (mb * (C1 / 16) + g) * 256)); },
                [&](int t) { return a0[t >> 4][t & 15]; });
        else
            dense_layer_mfma_stream<MB2, C1 / 8, PCRL_BWDG_RING>(
                a1, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1() + (mb * (C1 / 8) + tq) * 256)); },
                [&](int t) { return a0[t >> 4][t & 15]; });
        if (half == 0) {   // B operand of the conv0 weight gradient: rows = input channels, row C = 1 (bias)
#pragma unroll
            for (int c = 0; c < 2 * T0; ++c)
                if (c < p.cl.C) buf_store_f1(r_ops, lane_off, 4u * (unsigned)(OL.xb() + c * 4), x[c]);
            buf_store_f1(r_ops, lane_off + 16u * (unsigned)p.cl.C, 4u * (unsigned)OL.xb(), 1.0f);
        }
#pragma unroll
        for (int mb = 0; mb < MB1; ++mb) store_block_pieces(r_ops, s_tr, gop_off(OL.h0(), mb, 0, OL.NP), tile_bytes, a0[mb], l31, half, lane);
        const float rstd1 = ln_to_xhat<kC2>(a1, p.eps);
        f32x16 xh1[MB2];
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb) {
            float2 gbv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) gbv[r] = reinterpret_cast<const float2*>(s_ln1)[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xh1[mb][r] = a1[mb][r];                       // xhat1 stays in registers for LayerNorm-1's backward
                a1[mb][r] = relu_nan(__builtin_fmaf(a1[mb][r], gbv[r].x, gbv[r].y));
            }
        }
        PCRL_GSTAMP(2);
        // ---- mu = s.h1 / C3 (the mean of z2 = W2 h1 over its C3 channels, without forming z2) ------------------------------------
        float mu;
        {
            float ps = 0.0f, lo, hi;
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) ps = __builtin_fmaf(a1[mb][r], s_sv[acc_chan(mb * 16 + r, 0) + 4 * half], ps);
            both_halves(ps, lo, hi);
            mu = (lo + hi) / (float)kC3;
        }
        PCRL_GSTAMP(3);
        // ---- the channels this point owns (all live: the prep dropped the others).  z2_c - mu from the dot product W2[c,:].h1; the
        // sparse part of W2^T dz2.  rstd2 is not known yet (it needs M h1, which is formed after the loop so that its 64
        // accumulators are not live here): everything that carries a factor rstd2 is left without it and finished below / by
        // the wgrad kernel (norm2.weight's gradient, the channel's row scale).
        f32x16 gacc[MB2];
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) gacc[mb][r] = 0.0f;
        float t1 = 0.0f, t2r = 0.0f;
        {
            const int start = (int)(own_w & 0xFFFFu), cnt = (int)(own_w >> 16);
            const idx_t* oc = reinterpret_cast<const idx_t*>(p.own_chan) + (long long)b * kC3 + start;
            for (int i = 0; __any(i < cnt); ++i) {
                const bool has = i < cnt;
                const int c = has ? ((kC3 <= 256 && i < 8) ? (int)((own_pk >> (8 * i)) & 0xFFull) : (int)oc[i]) : 0;
                const f32x4* wrow = reinterpret_cast<const f32x4*>(p.w2 + (long long)c * kC2 + 4 * half);   // floats 32 mb + 8 g4 + 4 half + (0..3)
                const float2 gb = s_ln2v[c];
                const float dyl = has ? g_row[c] : 0.0f;
                const float dx = dyl * gb.x;
                float d = 0.0f, lo, hi;
#pragma unroll
                for (int mb = 0; mb < MB2; ++mb) {
                    f32x4 w4[4];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) w4[g4] = wrow[8 * mb + 2 * g4];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        d = __builtin_fmaf(w4[r >> 2][r & 3], a1[mb][r], d);
                        gacc[mb][r] = __builtin_fmaf(dx, w4[r >> 2][r & 3], gacc[mb][r]);
                    }
                }
                both_halves(d, lo, hi);
                const float zc = (lo + hi) - mu;
                if (has && half == 0) {           // exactly one point per channel; the factor rstd2 of this point is applied by the wgrad kernel
                    pw[GL.g2() + c] = dyl * zc;
                    pw[GL.be2() + c] = dyl;
                    p.chc[(long long)b * kC3 + c] = dx;
                }
                t1 = t1 + dx;
                t2r = __builtin_fmaf(dx, zc, t2r);
            }
        }
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb)       // h1 pieces: behind the loop's loads; M h1 and LayerNorm-1's backward issue no loads
            store_block_pieces(r_ops, s_tr, gop_off(OL.h1(), mb, 0, OL.NP), tile_bytes, a1[mb], l31, half, lane);
        PCRL_GSTAMP(4);
        // ---- q = Mc h1 (Mc = M - s s^T / C3, the centred Gram image: q = M h1 - mu s); var = h1.q / C3 -------------------------------
        f32x16 q[MB2];
#if PCRL_BWDG_MSPLIT
        dense_layer_split<MB2, kC2 / 16>(
            q, [&](int k, int mb, int g) { return s_mv[k * (kC2 * kC2 / 8) + (mb * (kC2 / 16) + g) * 64 + lane]; },
            [&](int t) { return a1[t >> 4][t & 15]; });
#else
        dense_layer_mfma<MB2, kC2 / 8, 2>(
            q, [&](int mb, int tq) { return s_mv[(mb * (kC2 / 8) + tq) * 64 + lane]; },
            [&](int t) { return a1[t >> 4][t & 15]; });
#endif
        float rstd2;
        {
            float pe = 0.0f, lo, hi;
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) pe = __builtin_fmaf(a1[mb][r], q[mb][r], pe);
            both_halves(pe, lo, hi);
            const float var = __builtin_fmaxf((lo + hi) / (float)kC3, 0.0f);      // a sum of squares up to rounding: no cancellation
            rstd2 = 1.0f / __builtin_sqrtf(var + p.eps);
        }
        PCRL_GSTAMP(5);
        // ---- dH1 = rstd2 (gacc - m1 s) - a q,  q = M h1 - mu s,  a = rstd2^2 m2,  m2 = rstd2 sum_own dx (z_c - mu) / C3 -----------------
        {
            const float m1 = t1 / (float)kC3, m2 = (rstd2 * t2r) / (float)kC3;
            const float a = (rstd2 * rstd2) * m2, vco = rstd2 * m1, uco = a * mu;
            if (half == 0) p.ptc[(long long)b * kC3 + s] = valid ? float4{a, vco, uco, rstd2} : float4{0.0f, 0.0f, 0.0f, 0.0f};
            const float cs = -vco;                 // coefficient of s (the a mu s term is inside the centred q)
#pragma unroll
            for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sj = s_sv[acc_chan(mb * 16 + r, 0) + 4 * half];
                    q[mb][r] = __builtin_fmaf(rstd2, gacc[mb][r], __builtin_fmaf(cs, sj, -(a * q[mb][r])));
                }
        }
        PCRL_GSTAMP(6);
        // ---- ReLU + LN1 backward (as in encoder_bwd_impl.h) ------------------------------------------------------------------
        float s1 = 0.0f, s2 = 0.0f, lo, hi;
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb) {
            float2 gbv[16]; float tg[16], tb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) gbv[r] = reinterpret_cast<const float2*>(s_ln1)[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = __builtin_fmaf(xh1[mb][r], gbv[r].x, gbv[r].y);
                const float dyl = y > 0.0f ? q[mb][r] : 0.0f;
                tg[r] = dyl * xh1[mb][r];          // norm1.weight / norm1.bias gradients: summed over this tile's 32 points below
                tb[r] = dyl;
                const float dx = dyl * gbv[r].x;
                q[mb][r] = dx;
                s1 = s1 + dx;
                s2 = __builtin_fmaf(dx, xh1[mb][r], s2);
            }
            allreduce_add32_x16(tg);
            allreduce_add32_x16(tb);
            if (l31 == 0) {        // this tile's partial sums; the wgrad kernel adds a cloud's tiles in tile order
                float2* n1 = reinterpret_cast<float2*>(p.n1part) + ((long long)b * bwdg_tpc(kC3) + tile) * kC2;
#pragma unroll
                for (int r = 0; r < 16; ++r) n1[acc_chan(mb * 16 + r, 0) + 4 * half] = float2{tg[r], tb[r]};
            }
        }
        both_halves(s1, lo, hi);
        const float n1 = (lo + hi) / (float)kC2;
        both_halves(s2, lo, hi);
        const float n2 = (lo + hi) / (float)kC2;
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                q[mb][r] = rstd1 * ((q[mb][r] - n1) - xh1[mb][r] * n2);
            }
        PCRL_GSTAMP(7);
        // ---- dH0 = W1^T dz1 ; ReLU backward ----------------------------------------------------
        f32x16 d0[MB1];
        if (SPLIT || PCRL_BWDG_DH0SPLIT)      // a purely linear layer (no decision depends on it): three-term bf16 split in either mode
            dense_layer_split<MB1, kC2 / 16>(
                d0, [&](int k, int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1ts(k) + (mb * (kC2 / 16) + g) * 256)); },
                [&](int t) { return q[t >> 4][t & 15]; });
        else
            if constexpr (MB1 % 2 == 0)
                dense_layer_mfma_stream<MB1, kC2 / 8, PCRL_BWDG_RING>(
                    d0, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1t() + (mb * (kC2 / 8) + tq) * 256)); },
                    [&](int t) { return q[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB1, kC2 / 8, 6>(
                    d0, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1t() + (mb * (kC2 / 8) + tq) * 256)); },
                    [&](int t) { return q[t >> 4][t & 15]; });
        PCRL_GSTAMP(9);
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb) store_block_pieces(r_ops, s_tr, gop_off(OL.dz1(), mb, 0, OL.NP), tile_bytes, q[mb], l31, half, lane);
#pragma unroll
        for (int mb = 0; mb < MB1; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) d0[mb][r] = ((mask0[mb] >> r) & 1u) ? d0[mb][r] : 0.0f;
            store_block_pieces(r_ops, s_tr, gop_off(OL.dz0(), mb, 0, OL.NP), tile_bytes, d0[mb], l31, half, lane);
        }
        PCRL_GSTAMP(8);
    }
}

// wgrad_blocks (encoder_bwd_impl.h) for two operands that stream from L2, through one buffer resource over the cloud's operand
// region: out[32 x 32 block (mb, nb0 + n)] = sum over slots of A[32 mb + i][slot] * Bm[32 (nb0 + n) + j][slot].  D octets of operands
// are in flight; the refills are UNCONDITIONAL (an octet past the end reads through an out-of-range offset: zeros, no memory request)
// and pinned behind the MFMAs that freed their registers.  With conditional refills the compiler's wait bookkeeping assumed the
// shortest queue and drained it after every octet (`7 mfma; vmcnt(1); 1 mfma; vmcnt(0)`): a ring of depth one whatever D said.
template <int NB, int D>
__device__ __forceinline__ void wgrad_blocks_stream(const __amdgpu_buffer_rsrc_t& rs, unsigned a_bytes, unsigned b_bytes, unsigned b_block_bytes,
                                                    int n_oct, int lane, f32x16 (&acc)[NB], unsigned b_dead = 0u) {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    const unsigned lane16 = 16u * (unsigned)lane;
    f32x4 ar[D], br[D][NB];
    auto fetch = [&](int q, f32x4& a, f32x4 (&b)[NB]) {
        const unsigned off = q < n_oct ? lane16 + 1024u * (unsigned)q : 0x80000000u;      // one octet = 64 lanes x 16 bytes
        a = buf_load_f4(rs, off, a_bytes);
#pragma unroll
        for (int n = 0; n < NB; ++n) b[n] = buf_load_f4(rs, off | b_dead, b_bytes + (unsigned)n * b_block_bytes);   // b_dead: lanes whose B rows read as zero
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, ar[d], br[d]);
    __builtin_amdgcn_sched_barrier(0);
    for (int q0 = 0; q0 < n_oct; q0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const f32x4 a = ar[d];
            f32x4 bv[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) bv[n] = br[d][n];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[n][j], acc[n], 0, 0, 0);
            fetch(q0 + d + D, ar[d], br[d]);       // past the end: zeros, which the MFMAs above add harmlessly
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- wgrad: per cloud --------------------------------------------------------------------------------------------------
// One or two 32 x 32 blocks of G = sum over slots of (a_slot h1[32 mb + i][slot]) h1[32 nb + j][slot], every operand from the LDS copy of
// the cloud's h1 pieces; the reads of the next octet are issued before this octet's MFMAs.  G is symmetric: only blocks with mb <= nb are
// formed (10 of 16 at c2 = 128; the others are stored as their mirror images), two per task -- (mbA, nbA) and (mbB, nbB), mbB < 0: one block.
__device__ __forceinline__ void gram_blocks(const f32x4* s_h1, const f32x4* s_a4, int mbA, int nbA, int mbB, int nbB, int n_oct, int lane,
                                            f32x16 (&acc)[2]) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    if (n_oct == 0) return;
    const bool two = mbB >= 0;
    const f32x4* a40 = s_h1 + mbA * n_oct * 64 + lane;
    const f32x4* b40 = s_h1 + nbA * n_oct * 64 + lane;
    const f32x4* a41 = two ? s_h1 + mbB * n_oct * 64 + lane : a40;
    const f32x4* b41 = two ? s_h1 + nbB * n_oct * 64 + lane : b40;
    const f32x4* sc = s_a4 + (lane >> 5);
    f32x4 a0 = a40[0], a1 = a41[0], s4 = sc[0], b0 = b40[0], b1 = b41[0];
    for (int qo = 0; qo < n_oct; ++qo) {
        const f32x4 as0 = a0 * s4, as1 = a1 * s4, c0 = b0, c1 = b1;
        if (qo + 1 < n_oct) {
            a0 = a40[(qo + 1) * 64]; a1 = a41[(qo + 1) * 64]; s4 = sc[2 * (qo + 1)]; b0 = b40[(qo + 1) * 64]; b1 = b41[(qo + 1) * 64];
        }
        if (two) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(as0[j], c0[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(as1[j], c1[j], acc[1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(as0[j], c0[j], acc[0], 0, 0, 0);
        }
    }
}

// The mirror image of an off-diagonal block of the symmetric G: D tile (mb, nb) -> rows 32 nb .., columns 32 mb .. of the row-major matrix.
// A lane holds column j of the tile, i.e. row j of the mirror image, as four runs of four consecutive columns: four 16-byte stores.
__device__ __forceinline__ void store_tile_mirrored(float* out, int ld, int mb, int nb, const f32x16& acc, int lane) {
    float* row = out + (long long)(32 * nb + (lane & 31)) * ld + 32 * mb + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(row + 8 * g) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
}

// The wgrad kernel's task lists: task[8 part + wave] = up to kWgradMaxTasks codes (kind << 5 | index; kind 1: dW1 task, 2: pair of G blocks,
// 3: single G block, 4: dW0 block; 0 ends the list).  Built on the host (make_wgrad_sched), a kernel argument.
constexpr int kWgradMaxTasks = 8;
struct alignas(8) WgradSched { unsigned char task[64][kWgradMaxTasks]; };

// The upper-triangle blocks of G as tasks: pairs that share their row block first, then the left-over diagonal blocks one by one (they
// are the short tasks the schedule below uses to level the waves).  code = mbA | nbA << 4 | (mbB + 1) << 8 | nbB << 12.
template <int MB2>
struct GramTasks {
    static_assert(MB2 == 2 || MB2 == 4, "G task tables are written for c2 = 64 and c2 = 128");
    static constexpr int n_pairs = MB2 == 4 ? 4 : 1, n_single = MB2 == 4 ? 2 : 1;
    __host__ __device__ static constexpr int pair(int t) {
        return MB2 == 4 ? (t == 0 ? (0 | 0 << 4 | 1 << 8 | 1 << 12) : t == 1 ? (0 | 2 << 4 | 1 << 8 | 3 << 12)
                           : t == 2 ? (1 | 2 << 4 | 2 << 8 | 3 << 12) : (2 | 2 << 4 | 3 << 8 | 3 << 12))
                        : (0 | 0 << 4 | 1 << 8 | 1 << 12);
    }
    __host__ __device__ static constexpr int single(int t) { return MB2 == 4 ? (t == 0 ? (1 | 1 << 4) : (3 | 3 << 4)) : (1 | 1 << 4); }
};

// Longest task first, each to the least loaded of the cloud's 8 P waves (ties: the lowest wave): dW1 tasks (W1NB blocks, both operands
// from L2), G pairs (LDS-fed), dW0 blocks (L2-fed), single G blocks.  Which wave forms a block changes no result.
template <int MB1, int MB2, int W1NB>
static void make_wgrad_sched(int P, WgradSched* sched) {
    typedef GramTasks<MB2> GT;
    const int G8 = 8 * P;
    float load[64];
    int n[64];
    for (int w = 0; w < 64; ++w) { load[w] = 0.0f; n[w] = 0; for (int i = 0; i < kWgradMaxTasks; ++i) sched->task[w][i] = 0; }
    auto give = [&](int kind, int idx, float cost) {
        int best = 0;
        for (int w = 1; w < G8; ++w) if (load[w] < load[best]) best = w;
        load[best] += cost;
        if (n[best] < kWgradMaxTasks) sched->task[best][n[best]++] = (unsigned char)(kind << 5 | idx);
    };
    for (int u = 0; u < MB2 * MB1 / W1NB; ++u) give(1, u, 1.3f * W1NB);
    for (int t = 0; t < GT::n_pairs; ++t) give(2, t, 2.0f);
    for (int t = 0; t < MB1; ++t) give(4, t, 1.3f);
    for (int t = 0; t < GT::n_single; ++t) give(3, t, 1.0f);
}

// conv0.weight [C1][C] and conv0.bias (column C of the x|1 operand), row block mb: wgrad_conv0 of encoder_bwd_impl.h with the
// operands of the next three octets in flight (its load -> wait -> 4 MFMAs loop paid an L2 round trip per octet).
__device__ __forceinline__ void wgrad_conv0_pipelined(const BwdParams& p, const __amdgpu_buffer_rsrc_t& rs, int blk_floats, float* pw, const GradLayout& GL,
                                                      int dz0_off, int xb_off, int mb, int n_oct, int lane) {
    // rows of the x|1 block beyond C are never written: those lanes read their B operand through an out-of-range offset (zeros)
    const bool live = (lane & 31) <= p.cl.C;
    f32x16 accs[1];
    wgrad_blocks_stream<1, PCRL_BWDG_WRING>(rs, 4u * (unsigned)(dz0_off + mb * blk_floats), 4u * (unsigned)xb_off, 0u, n_oct, lane, accs, live ? 0u : 0x80000000u);
    const f32x16 acc = accs[0];
    const int col = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (col < p.cl.C) pw[GL.w0() + row * p.cl.C + col] = acc[r];
        else if (col == p.cl.C) pw[GL.b0() + row] = acc[r];
    }
}

template <int C1, int kC2, int kC3>
__global__ __launch_bounds__(512, 1) void encoder_bwdg_wgrad_kernel(const BwdParams p, const WgradSched sched) {
    constexpr int MB1 = C1 / 32, MB2 = kC2 / 32;
    constexpr OpsLayoutG OL{C1 / 32, kC2 / 32, bwdg_np(kC3)};
    constexpr GramExtra GX{kC2};
    static_assert(kC3 <= 256, "the LDS-staged wgrad kernel holds a cloud's whole h1: c3 <= 256 (wide: encoder_bwdg_wgrad_wide_kernel)");
    const GradLayout GL{p.cl.C, C1, kC2, kC3};
    extern __shared__ __attribute__((aligned(16))) f32x4 s_h1[];      // [MB2][n_oct][64] pieces, then the per-slot coefficients
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = p.parts;
    for (int item = blockIdx.x; item < p.cl.B * P; item += gridDim.x) {
        const int b = item / P, part = item - b * P;
        const float* ops = p.ops + (long long)b * OL.total();
        const __amdgpu_buffer_rsrc_t r_ops = make_rsrc(ops, 4u * (unsigned)OL.total());
        float* pw = p.pw + (long long)b * p.pw_stride;
        float* px = pw + GL.total();
        const int n_tiles = (p.n_act[b] + 31) / 32, n_oct = n_tiles * 4;
        PCRL_WSTAMP(0);
        if (part == 0 && tid < 2 * kC2) {   // norm1 gradients: fixed-order sum over the cloud's tiles (one partial per tile)
            const float* n1 = p.n1part + (long long)b * 8 * kC2 * 2;
            float part_t[8], acc = 0.0f;                      // at most C3 / 32 <= 8 tiles: every load in flight, then the adds in tile order
#pragma unroll
            for (int t = 0; t < 8; ++t) part_t[t] = t < n_tiles ? n1[t * kC2 * 2 + tid] : 0.0f;
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = t < n_tiles ? acc + part_t[t] : acc;
            pw[((tid & 1) ? GL.be1() : GL.g1()) + (tid >> 1)] = acc;
        }
        // stage the cloud's h1 pieces (every load of a thread in flight before its first LDS write) and the per-slot
        // coefficients, as float4 per (octet, k-lane) = the four k-slots of a piece
        __syncthreads();
        f32x4* s_a4 = s_h1 + MB2 * n_oct * 64;        // [n_oct * 2]
        f32x4* s_v4 = s_a4 + 2 * n_oct;
        f32x4* s_u4 = s_v4 + 2 * n_oct;
        float* s_r2 = reinterpret_cast<float*>(s_u4 + 2 * n_oct);      // [8 n_oct] rstd2 of every slot
        float* s_co = s_r2 + 8 * n_oct;                                // [kC3] the channel's row scale (without rstd2), then with it
        int* s_sl = reinterpret_cast<int*>(s_co + kC3);                // [kC3] the channel's slot
        int* s_cnt = s_sl + kC3;                                       // ticket counter of the LDS-only chunks (v / u, S rows)
        {
            const f32x4* h1g = reinterpret_cast<const f32x4*>(ops + OL.h1());
            const int per_blk = n_oct * 64;            // <= 2048 = 4 x 512
            f32x4 tmp[MB2][4];
#pragma unroll
            for (int nb = 0; nb < MB2; ++nb)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tid + 512 * k < per_blk) tmp[nb][k] = h1g[nb * 32 * 64 + tid + 512 * k];
#pragma unroll
            for (int nb = 0; nb < MB2; ++nb)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tid + 512 * k < per_blk) s_h1[nb * per_blk + tid + 512 * k] = tmp[nb][k];
            const float4* ptc = p.ptc + (long long)b * kC3;
            if (tid < 2 * n_oct) {                     // slots 4 tid .. 4 tid + 3
                const float4 c0 = ptc[4 * tid], c1 = ptc[4 * tid + 1], c2 = ptc[4 * tid + 2], c3 = ptc[4 * tid + 3];
                s_a4[tid] = f32x4{c0.x, c1.x, c2.x, c3.x};
                s_v4[tid] = f32x4{c0.y, c1.y, c2.y, c3.y};
                s_u4[tid] = f32x4{c0.z, c1.z, c2.z, c3.z};
                reinterpret_cast<f32x4*>(s_r2)[tid] = f32x4{c0.w, c1.w, c2.w, c3.w};
            }
            if (tid < kC3) {
                s_co[tid] = p.chc[(long long)b * kC3 + tid];
                s_sl[tid] = (int)p.slot[(long long)b * kC3 + tid];
            }
            if (tid == 0) *s_cnt = 0;
        }
        __syncthreads();
        if (tid < kC3) {       // the factor rstd2 of the channel's point, left out by the points kernel: row scale and norm2.weight
            // (a dropped channel has slot 0 and zeros everywhere: its r2 may be anything, the products below are guarded)
            const float co = s_co[tid];
            const float r2 = s_r2[s_sl[tid]];
            s_co[tid] = co != 0.0f ? co * r2 : 0.0f;
            if (part == 0) {
                const float raw = pw[GL.g2() + tid];
                pw[GL.g2() + tid] = raw != 0.0f ? raw * r2 : 0.0f;
            }
        }
        __syncthreads();       // the scaled row coefficients are read by whichever wave takes the channel's S chunk
        PCRL_WSTAMP(1);
        // MFMA tasks of the cloud, spread over the 8 P waves of its P workgroups.
        // dW1: two column blocks per task share the dz1 operand (three loads per eight MFMAs instead of four): wgrad kernel 47.4 -> 43.8 us
        // at K1, 296 -> 254 us at 1 024 x 1 200 clouds with c1 = 128
        constexpr int W1NB = PCRL_BWDG_W1NB(MB1);
        typedef GramTasks<MB2> GT;
        // this wave's tasks: make_wgrad_sched (host) -- longest task first, each to the least loaded of the cloud's 8 P waves
        unsigned long long mine;                      // the wave's whole list in one load (a kernel-argument read per task would be a round trip each)
        __builtin_memcpy(&mine, sched.task[part * 8 + wave], 8);
        static_assert(kWgradMaxTasks == 8, "a wave's list is read as one 64-bit word");
        for (int i = 0; i < kWgradMaxTasks; ++i) {
            const int task = (int)((mine >> (8 * i)) & 0xFFull), kind = task >> 5, code = task & 31;
            if (kind == 0) break;
            if (kind == 1) {
                const int u = code;
                f32x16 acc[W1NB];
                constexpr int per_row = MB1 / W1NB;
                // both operands stream from L2: six octets in flight (three left the matrix pipe waiting two thirds of the time)
                wgrad_blocks_stream<W1NB, PCRL_BWDG_WRING>(r_ops, 4u * (unsigned)(OL.dz1() + (u / per_row) * OL.blk()),
                                                           4u * (unsigned)(OL.h0() + W1NB * (u % per_row) * OL.blk()), 4u * (unsigned)OL.blk(), n_oct, lane, acc);
#pragma unroll
                for (int n = 0; n < W1NB; ++n) store_tile(pw + GL.w1(), C1, u / per_row, W1NB * (u % per_row) + n, C1, acc[n], lane);
            } else if (kind == 2 || kind == 3) {
                f32x16 acc[2];
                const int gc = kind == 2 ? GT::pair(code) : GT::single(code);
                const int mbA = gc & 15, nbA = (gc >> 4) & 15, mbB = ((gc >> 8) & 15) - 1, nbB = (gc >> 12) & 15;
                gram_blocks(s_h1, s_a4, mbA, nbA, mbB, nbB, n_oct, lane, acc);
                store_tile(px + GX.G(), kC2, mbA, nbA, kC2, acc[0], lane);
                if (mbA != nbA) store_tile_mirrored(px + GX.G(), kC2, mbA, nbA, acc[0], lane);
                if (mbB >= 0) {
                    store_tile(px + GX.G(), kC2, mbB, nbB, kC2, acc[1], lane);
                    if (mbB != nbB) store_tile_mirrored(px + GX.G(), kC2, mbB, nbB, acc[1], lane);
                }
            } else {
                wgrad_conv0_pipelined(p, r_ops, OL.blk(), pw, GL, OL.dz0(), OL.xb(), code, n_oct, lane);
            }
        }
        PCRL_WSTAMP(2);
        PCRL_WSTAMP(3);
        PCRL_WSTAMP(4);
        // What needs only the LDS copy -- v, u (2 C2 dot products over the active slots; as MFMA blocks they cost C2 / 32 blocks for two
        // useful rows) and the sparse rows of dW2, S[c][j] = (rstd2 dx)_c h1[slot(c)][j] -- is cut into 1 024-element chunks that the
        // waves draw from a ticket counter as they finish their MFMA blocks: the waves with the lighter block lists (no dW0 block,
        // LDS-fed blocks only) absorb them instead of waiting at a barrier for the others.  A chunk's result does not depend on
        // who computes it.
        PCRL_WSTAMP(5);
        {
            constexpr int CH_ROWS = 1024 / kC2;                              // channels per S chunk
            const int n_vu = part == P - 1 ? (2 * kC2 + 63) / 64 : 0, n_s = (kC3 / P) / CH_ROWS;
            const float* s_h1f = reinterpret_cast<const float*>(s_h1);
            for (;;) {
                int ticket = 0;
                if (lane == 0) ticket = atomicAdd(s_cnt, 1);
                ticket = __builtin_amdgcn_readfirstlane(ticket);
                if (ticket >= n_vu + n_s) break;
                if (ticket < n_vu) {
                    const int d = ticket * 64 + lane;                         // dot product d: j = d % C2 of v (d < C2) or u
                    if (d < 2 * kC2) {
                        const int j = d % kC2, which = d / kC2;
                        const f32x4* hj = s_h1 + (j >> 5) * n_oct * 64 + (j & 31);
                        const f32x4* co = which ? s_u4 : s_v4;
                        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
                        for (int q2 = 0; q2 < 2 * n_oct; ++q2)         // (octet, k-lane) pairs in slot order
                            acc4 = __builtin_elementwise_fma(hj[(q2 >> 1) * 64 + 32 * (q2 & 1)], co[q2], acc4);
                        px[(which ? GX.u() : GX.v()) + j] = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
                    }
                } else {
                    const int c_base = part * (kC3 / P) + (ticket - n_vu) * CH_ROWS;
#pragma unroll 4
                    for (int e = 0; e < 16; ++e) {
                        const int flat = e * 64 + lane, c = c_base + flat / kC2, j = flat % kC2;
                        const float co = s_co[c];
                        const int sl = s_sl[c];
                        // piece element of (channel j, slot sl): block j >> 5, octet sl >> 3, lane (j & 31) + 32 ((sl >> 2) & 1), k-slot sl & 3
                        const float h = s_h1f[(((j >> 5) * n_oct + (sl >> 3)) * 64 + (j & 31) + 32 * ((sl >> 2) & 1)) * 4 + (sl & 3)];
                        pw[GL.w2() + c * kC2 + j] = co != 0.0f ? co * h : 0.0f;
                    }
                }
            }
        }
        PCRL_WSTAMP(6);
    }
}

// ---- wide last layer (c3 = 1024, the class default mlp_spec): simple kernels, no shipped config runs them ------------------------------
// The Gram form needs only M = W2^T W2 [C2 x C2] whatever c3 is, so the points kernel above is the same template (two-byte slot /
// channel numbers, up to c3 / 32 tiles per cloud).  What is specific to c3 <= 256 are the prep kernel (thread = channel = slot with
// 256 threads) and the wgrad kernel (a cloud's whole h1 in LDS): the versions below trade speed for plainness -- 1 024 threads per
// cloud in the prep, every operand of the wgrad blocks streamed from L2.
template <int kC2, int kC3>
__global__ __launch_bounds__(256) void encoder_bwdg_mtile_kernel(const BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float s_strip[];
    gram_tile<kC2, kC3>(p.w2, p.mimg, (int)blockIdx.x, threadIdx.x, s_strip);
}

template <int C1, int kC2, int kC3>
__global__ __launch_bounds__(1024) void encoder_bwdg_prep_wide_kernel(const BwdParams p) {
    static_assert(kC3 == 1024, "one thread per channel");
    extern __shared__ __attribute__((aligned(16))) unsigned s_words[];   // [nW] bitmap, then [nW] exclusive prefix popcounts
    __shared__ int s_scan[32];
    __shared__ unsigned short s_slot[kC3];
    typedef typename BwdgIdx<kC3>::type idx_t;
    const GradLayout GL{p.cl.C, C1, kC2, kC3};
    const int tid = threadIdx.x, b = blockIdx.x, wv = tid >> 6;
    const int nW = (p.cl.N + 31) >> 5;
    unsigned* s_pre = s_words + nW;
    int pc = p.argmax[(long long)b * kC3 + tid];
    const bool live = p.pooled[(long long)b * kC3 + tid] > 0.0f;
    for (int w = tid; w < nW; w += 1024) s_words[w] = 0u;
    __syncthreads();
    pc = pc < 0 ? 0 : (pc >= p.cl.N ? p.cl.N - 1 : pc);
    if (live) atomicOr(&s_words[pc >> 5], 1u << (pc & 31));
    __syncthreads();
    // block-wide exclusive scan helper: per-thread count -> exclusive prefix, total in s_scan[16]
    auto block_scan = [&](int local, int* total) {
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if ((tid & 63) >= d) incl += v; }
        __syncthreads();
        if ((tid & 63) == 63) s_scan[wv] = incl;
        __syncthreads();
        int base = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int v = s_scan[w]; if (w < wv) base += v; tot += v; }
        *total = tot;
        return base + incl - local;
    };
    const int per = (nW + 1023) >> 10, w0 = tid * per;
    int local = 0;
    for (int k = 0; k < per; ++k)
        if (w0 + k < nW) local += __popc(s_words[w0 + k]);
    int n_act;
    int run = block_scan(local, &n_act);
    for (int k = 0; k < per; ++k)
        if (w0 + k < nW) { s_pre[w0 + k] = (unsigned)run; run += __popc(s_words[w0 + k]); }
    __syncthreads();
    const int slot = (int)s_pre[pc >> 5] + __popc(s_words[pc >> 5] & ((1u << (pc & 31)) - 1u));
    s_slot[tid] = live ? (unsigned short)slot : (unsigned short)0xFFFF;      // 0xFFFF never names a slot (n_act <= 1024)
    reinterpret_cast<idx_t*>(p.slot)[(long long)b * kC3 + tid] = live ? (idx_t)slot : (idx_t)0;
    if (live) {
        p.act[(long long)b * kC3 + slot] = pc;
    } else {
        float* pw = p.pw + (long long)b * p.pw_stride;
        pw[GL.g2() + tid] = 0.0f;
        pw[GL.be2() + tid] = 0.0f;
        p.chc[(long long)b * kC3 + tid] = 0.0f;
    }
    if (tid == 0) p.n_act[b] = n_act;
    __syncthreads();
    // thread = slot: its channels (ascending), counted and then written behind the slots before it
    int cnt = 0;
    if (tid < n_act)
        for (int c = 0; c < kC3; ++c) cnt += s_slot[c] == (unsigned short)tid ? 1 : 0;
    int total_own;
    int pos = block_scan(cnt, &total_own);
    p.own[(long long)b * kC3 + tid] = (unsigned)pos | ((unsigned)cnt << 16);
    if (cnt > 0)
        for (int c = 0; c < kC3; ++c)
            if (s_slot[c] == (unsigned short)tid) reinterpret_cast<idx_t*>(p.own_chan)[(long long)b * kC3 + pos++] = (idx_t)c;
}

// One 32 x 32 block = sum over the octets of A-piece x B-piece, both streamed from L2 (four octets in flight); `scale(q, kl)` gives the
// four per-slot factors of the A piece (octet q, k-lane kl) or ones.
template <class ScaleFn>
__device__ __forceinline__ void l2_block(const f32x4* a4, const f32x4* b4, int n_oct, ScaleFn scale, bool b_live, f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    constexpr int D = 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ar[D], br[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < n_oct) { ar[d] = a4[d * 64]; br[d] = b_live ? b4[d * 64] : zero4; }
    for (int q0 = 0; q0 < n_oct; q0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int q = q0 + d;
            if (q < n_oct) {
                const f32x4 a = ar[d] * scale(q), bv = br[d];
                if (q + D < n_oct) { ar[d] = a4[(q + D) * 64]; br[d] = b_live ? b4[(q + D) * 64] : zero4; }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[j], acc, 0, 0, 0);
            }
        }
    }
}

template <int C1, int kC2, int kC3>
__global__ __launch_bounds__(512, 1) void encoder_bwdg_wgrad_wide_kernel(const BwdParams p) {
    constexpr int MB1 = C1 / 32, MB2 = kC2 / 32, NP = bwdg_np(kC3), TPC = bwdg_tpc(kC3);
    constexpr OpsLayoutG OL{C1 / 32, kC2 / 32, NP};
    constexpr GramExtra GX{kC2};
    typedef typename BwdgIdx<kC3>::type idx_t;
    const GradLayout GL{p.cl.C, C1, kC2, kC3};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kl = lane >> 5;
    for (int b = blockIdx.x; b < p.cl.B; b += gridDim.x) {
        const float* ops = p.ops + (long long)b * OL.total();
        float* pw = p.pw + (long long)b * p.pw_stride;
        float* px = pw + GL.total();
        const float4* ptc = p.ptc + (long long)b * kC3;
        const int n_tiles = (p.n_act[b] + 31) / 32, n_oct = n_tiles * 4;
        if (tid < 2 * kC2) {   // norm1 gradients: fixed-order sum over the cloud's tiles
            const float* n1 = p.n1part + (long long)b * TPC * kC2 * 2;
            float acc = 0.0f;
            for (int t = 0; t < n_tiles; ++t) acc = acc + n1[t * kC2 * 2 + tid];
            pw[((tid & 1) ? GL.be1() : GL.g1()) + (tid >> 1)] = acc;
        }
        const f32x4* h1p = reinterpret_cast<const f32x4*>(ops + OL.h1());
        const f32x4* dz1p = reinterpret_cast<const f32x4*>(ops + OL.dz1());
        const f32x4* h0p = reinterpret_cast<const f32x4*>(ops + OL.h0());
        const f32x4* dz0p = reinterpret_cast<const f32x4*>(ops + OL.dz0());
        const f32x4* xbp = reinterpret_cast<const f32x4*>(ops + OL.xb());
        const f32x4 ones = {1.f, 1.f, 1.f, 1.f};
        auto a_of = [&](int q) { const float4* c = ptc + 8 * q + 4 * kl; return f32x4{c[0].x, c[1].x, c[2].x, c[3].x}; };
        auto one = [&](int) { return ones; };
        constexpr int nG = MB2 * MB2, nW1 = MB2 * MB1, nW0 = MB1;
        for (int t = wave; t < nG + nW1 + nW0; t += 8) {
            f32x16 acc;
            if (t < nG) {               // G[32 mb .., 32 nb ..] = sum_slots a h1 h1^T
                const int mb = t / MB2, nb = t % MB2;
                l2_block(h1p + mb * NP * 64 + lane, h1p + nb * NP * 64 + lane, n_oct, a_of, true, acc);
                store_tile(px + GX.G(), kC2, mb, nb, kC2, acc, lane);
            } else if (t < nG + nW1) {
                const int u = t - nG, mb = u / MB1, nb = u % MB1;
                l2_block(dz1p + mb * NP * 64 + lane, h0p + nb * NP * 64 + lane, n_oct, one, true, acc);
                store_tile(pw + GL.w1(), C1, mb, nb, C1, acc, lane);
            } else {                    // conv0.weight | conv0.bias: rows of the x|1 block beyond C were never written
                const int mb = t - nG - nW1;
                l2_block(dz0p + mb * NP * 64 + lane, xbp + lane, n_oct, one, (lane & 31) <= p.cl.C, acc);
                const int col = lane & 31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (col < p.cl.C) pw[GL.w0() + row * p.cl.C + col] = acc[r];
                    else if (col == p.cl.C) pw[GL.b0() + row] = acc[r];
                }
            }
        }
        if (tid < 2 * kC2) {   // v[j] = sum_slots (rstd2 m1) h1[j][slot], u[j] = sum_slots (a mu) h1[j][slot], slot order
            const int j = tid % kC2, which = tid / kC2;
            const f32x4* hj = h1p + (j >> 5) * NP * 64 + (j & 31);
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
            for (int q2 = 0; q2 < 2 * n_oct; ++q2) {
                const float4* c = ptc + 4 * q2;
                const f32x4 co = which ? f32x4{c[0].z, c[1].z, c[2].z, c[3].z} : f32x4{c[0].y, c[1].y, c[2].y, c[3].y};
                acc4 = __builtin_elementwise_fma(hj[(q2 >> 1) * 64 + 32 * (q2 & 1)], co, acc4);
            }
            px[(which ? GX.u() : GX.v()) + j] = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        }
        {   // S rows and norm2.weight's missing factor rstd2 of the channel's point
            const idx_t* slot = reinterpret_cast<const idx_t*>(p.slot) + (long long)b * kC3;
            const float* chc = p.chc + (long long)b * kC3;
            const float* h1f = ops + OL.h1();
            const int j = tid % kC2, cofs = tid / kC2;
            for (int c0 = 0; c0 < kC3; c0 += 512 / kC2) {
                const int c = c0 + cofs, sl = (int)slot[c];
                const float raw = chc[c];
                const float r2 = raw != 0.0f ? ptc[sl].w : 0.0f;
                const float h = raw != 0.0f ? h1f[(((j >> 5) * NP + (sl >> 3)) * 64 + (j & 31) + 32 * ((sl >> 2) & 1)) * 4 + (sl & 3)] : 0.0f;
                pw[GL.w2() + c * kC2 + j] = raw != 0.0f ? (raw * r2) * h : 0.0f;
                if (j == 0) { const float g = pw[GL.g2() + c]; pw[GL.g2() + c] = g != 0.0f ? g * ptc[sl].w : 0.0f; }
            }
        }
    }
}

#if PCRL_BWD_MODE == 4
// ---- reduce over the clouds (fixed order) and the finish of dW2 -----------------------------------------------------------
template <int ARITH>      // (one instance per translation unit)
__global__ __launch_bounds__(1024) void encoder_bwdg_reduce_kernel(const float* __restrict__ pw, int B, int stride, int n_main,
                                                                    float* __restrict__ grads, float* __restrict__ extra,
                                                                    const ColsumParams cs, int main_blocks) {
    if ((int)blockIdx.x >= main_blocks) {      // column-sum jobs riding on this launch (pcrl_encoder_bwd_attach_colsum)
        if (threadIdx.x < 256) colsum_block(cs, (int)blockIdx.x - main_blocks, (int)threadIdx.x);
        return;
    }
    __shared__ float s_part[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + c;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    if (i < stride) {
        int b = g;
        for (; b + 48 < B; b += 64) {
            p0 = p0 + pw[(long long)(b + 0) * stride + i]; p1 = p1 + pw[(long long)(b + 16) * stride + i];
            p2 = p2 + pw[(long long)(b + 32) * stride + i]; p3 = p3 + pw[(long long)(b + 48) * stride + i];
        }
        for (; b < B; b += 16) p0 = p0 + pw[(long long)b * stride + i];
    }
    s_part[g][c] = (p0 + p1) + (p2 + p3);
    __syncthreads();
    if (g == 0 && i < stride) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = acc + s_part[k][c];
        if (i < n_main) grads[i] = acc; else extra[i - n_main] = acc;
    }
}

// grads[w2][c][j] = S[c][j] - v[j] + u[j] - sum_k W2[c][k] G[k][j]: a block stages G in LDS (every load in flight) and finishes
// 512 / C2 x 2 rows of dW2
template <int kC2>
__global__ __launch_bounds__(256) void encoder_bwdg_finish_kernel(const float* __restrict__ w2, const float* __restrict__ extra, int C3,
                                                                   float* __restrict__ gw2) {
    extern __shared__ __attribute__((aligned(16))) float s_G[];       // [kC2][kC2], then [rows][kC2] of W2
    constexpr GramExtra GX{kC2};
    constexpr int ROWS = (256 / kC2) * 2;                               // rows of dW2 per block
    const int tid = threadIdx.x, c0 = blockIdx.x * ROWS;
    stage_to_lds<256, kC2 * kC2 / 4>(reinterpret_cast<f32x4*>(s_G), reinterpret_cast<const f32x4*>(extra + GX.G()), tid);
    float* s_w = s_G + kC2 * kC2;
    for (int i = tid; i < ROWS * kC2; i += 256) s_w[i] = (c0 + i / kC2 < C3) ? w2[(long long)c0 * kC2 + i] : 0.0f;
    __syncthreads();
    const int j = tid % kC2, rg = tid / kC2;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = rg * 2 + rr, c = c0 + row;
        if (c >= C3) continue;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        const float* wr = s_w + row * kC2;
#pragma unroll 4
        for (int k = 0; k < kC2; k += 4) {
            a0 = __builtin_fmaf(wr[k], s_G[(k + 0) * kC2 + j], a0);
            a1 = __builtin_fmaf(wr[k + 1], s_G[(k + 1) * kC2 + j], a1);
            a2 = __builtin_fmaf(wr[k + 2], s_G[(k + 2) * kC2 + j], a2);
            a3 = __builtin_fmaf(wr[k + 3], s_G[(k + 3) * kC2 + j], a3);
        }
        const long long e = (long long)c * kC2 + j;
        gw2[e] = ((gw2[e] - extra[GX.v() + j]) + extra[GX.u() + j]) - ((a0 + a1) + (a2 + a3));
    }
}
#endif  // PCRL_BWD_MODE == 4

// floats of one workgroup row of the team kernel (FusedRow::total(), encoder_bwd_fused.h)
__host__ __device__ constexpr int fused_row_floats(int C, int C1, int C2, int C3) {
    return (C1 * C + C1 + C2 * C1 + 2 * C2 + 2 * C3 + (C2 / 32) * (C2 / 32 + 1) / 2 * 1024 + 2 * C2 + 63) & ~63;
}

static size_t bwdg_lds_bytes_points(int T0, int C1, int kC2, int kC3) {
    return sizeof(ChanSrc) * PCRL_MAX_CHANNELS + 4 * (size_t)(kMaxTileModeClouds + 8) +
           sizeof(float) * (2 * kC2 + 2 * kC3 + C1 + (size_t)(C1 / 32) * T0 * 64 + kC2 + (size_t)bwdg_m_lds_floats(kC2) + 4 * 8 * 33 * 4);
}

constexpr int kFusedMaxRows = 1024;     // workgroups of the team kernel (encoder_bwd_fused.h): two per CU
struct BwdgWorkspace {
    size_t ops, pw, nact, act, slot, own, own_chan, ptc, chc, n1part, gvu, mimg, wgrows, nitems, own_pack, total;
    int fused_rows;
};
static BwdgWorkspace bwdg_workspace(int B, int C, int C1, int kC2, int kC3) {
    const OpsLayoutG OL{C1 / 32, kC2 / 32, bwdg_np(kC3)};
    const GradLayout GL{C, C1, kC2, kC3};
    const GramExtra GX{kC2};
    const size_t isz = kC3 > 256 ? 2 : 1;          // bytes per slot / channel number
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    BwdgWorkspace w;
    w.ops = 0;
    w.pw = al(w.ops + sizeof(float) * (size_t)B * OL.total());
    w.nact = al(w.pw + sizeof(float) * (size_t)B * (GL.total() + GX.total()));
    w.act = al(w.nact + sizeof(int) * (size_t)B);
    w.slot = al(w.act + sizeof(int) * (size_t)B * kC3);
    w.own = al(w.slot + isz * (size_t)B * kC3);
    w.own_chan = al(w.own + sizeof(unsigned) * (size_t)B * kC3);
    w.ptc = al(w.own_chan + isz * (size_t)B * kC3);
    w.chc = al(w.ptc + sizeof(float4) * (size_t)B * kC3);
    w.n1part = al(w.chc + sizeof(float) * (size_t)B * kC3);
    w.gvu = al(w.n1part + sizeof(float) * (size_t)B * bwdg_tpc(kC3) * kC2 * 2);
    w.mimg = al(w.gvu + sizeof(float) * (size_t)GX.total());
    w.wgrows = al(w.mimg + sizeof(float) * ((size_t)kC2 * kC2 + kC2 + 3 * (size_t)kC2 * kC2 / 2));     // fp32 image, s, three bf16 term images
    w.fused_rows = (int)std::min<size_t>(kFusedMaxRows, (size_t)B * bwdg_tpc(kC3));
    w.nitems = al(w.wgrows + sizeof(float) * (size_t)w.fused_rows * fused_row_floats(C, C1, kC2, kC3));
    w.own_pack = al(w.nitems + 64);
    w.total = al(w.own_pack + sizeof(unsigned long long) * (size_t)B * kC3);
    return w;
}

int encoder_bwdg_launch_f32(int T0, int c1, int c2, int c3, const BwdParams& p, hipStream_t st);
int encoder_bwdg_launch_split(int T0, int c1, int c2, int c3, const BwdParams& p, hipStream_t st);

#if PCRL_BWD_MODE == 4
#if PCRL_BWDG_ARITH == 0
#include "encoder_bwd_fused.h"
#endif
// PCRL_BWDG_SYNC=1 (development): synchronise after every launch so that a faulting kernel is named.
#define PCRL_BWDG_AFTER(name)                                                                                     \
    do {                                                                                                           \
        PCRL_CHECK_LAUNCH(name);                                                                                   \
    } while (0)

template <int T0, int C1, int C2, int C3>
static int launch_bwdg_wide(const BwdParams& p, hipStream_t stream) {
    const int nW = (p.cl.N + 31) / 32;
    if (p.phase != 2) {
        auto mt = encoder_bwdg_mtile_kernel<C2, C3>;
        constexpr size_t mt_lds = 2 * 16 * (size_t)C3 * sizeof(float);
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(mt), mt_lds)) return rc;
        hipLaunchKernelGGL(mt, dim3((C2 / 16) * (C2 / 16)), dim3(256), mt_lds, stream, p);
        PCRL_BWDG_AFTER("encoder_bwdg_mtile_kernel");
        auto prep = encoder_bwdg_prep_wide_kernel<C1, C2, C3>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(prep), 2 * sizeof(unsigned) * (size_t)kBitmapMaxWords)) return rc;
        hipLaunchKernelGGL(prep, dim3(p.cl.B), dim3(1024), 2 * sizeof(unsigned) * (size_t)nW, stream, p);
        PCRL_BWDG_AFTER("encoder_bwdg_prep_wide_kernel");
    }
    if (p.phase == 1) return PCRL_OK;
    const size_t lds = bwdg_lds_bytes_points(T0, C1, C2, C3);
    auto kern = encoder_bwdg_points_kernel<T0, C1, C2, C3, false>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(num_cus()), dim3(64 * kBwdgWaves), lds, stream, p);
    PCRL_BWDG_AFTER("encoder_bwdg_points_kernel");
    hipLaunchKernelGGL((encoder_bwdg_wgrad_wide_kernel<C1, C2, C3>), dim3(min(p.cl.B, num_cus())), dim3(512), 0, stream, p);
    PCRL_BWDG_AFTER("encoder_bwdg_wgrad_wide_kernel");
    const GradLayout GL{p.cl.C, C1, C2, C3};
    const GramExtra GX{C2};
    const int stride = GL.total() + GX.total();
    hipLaunchKernelGGL(encoder_bwdg_reduce_kernel<PCRL_BWDG_ARITH>, dim3((stride + 63) / 64 + p.cs_blocks), dim3(1024), 0, stream, p.pw, p.cl.B, stride, GL.total(), p.grads, p.gvu,
                       p.cs, (stride + 63) / 64);
    PCRL_BWDG_AFTER("encoder_bwdg_reduce_kernel");
    constexpr int rows = (256 / C2) * 2;
    constexpr size_t fin_lds = sizeof(float) * ((size_t)C2 * C2 + (size_t)rows * C2);
    auto fin = encoder_bwdg_finish_kernel<C2>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(fin), fin_lds)) return rc;
    hipLaunchKernelGGL(fin, dim3((C3 + rows - 1) / rows), dim3(256), fin_lds, stream, p.w2, p.gvu, C3, p.grads + GL.w2());
    PCRL_BWDG_AFTER("encoder_bwdg_finish_kernel");
    return PCRL_OK;
}

template <int T0, int C1, int C2, int C3>
static int launch_bwdg(const BwdParams& p, hipStream_t stream) {
    constexpr bool kSplit = PCRL_BWDG_ARITH == 2;
    const int nW = (p.cl.N + 31) / 32;
    if (p.phase != 2) {
        auto prep = encoder_bwdg_prep_kernel<C1, C2, C3>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(prep), 2 * sizeof(unsigned) * (size_t)kBitmapMaxWords)) return rc;
        const size_t prep_lds = std::max(2 * sizeof(unsigned) * (size_t)nW, (size_t)kGramTileLds);
        // behind the clouds: the Gram image's 16 x 16 tiles, then the workgroups of an attached LayerNorm backward (8 KB of the same LDS)
        hipLaunchKernelGGL(prep, dim3(p.cl.B + (C2 / 16) * (C2 / 16) + p.ln_blocks), dim3(256), prep_lds, stream, p);
        PCRL_BWDG_AFTER("encoder_bwdg_prep_kernel");
    }
    if (p.phase == 1) return PCRL_OK;
#if PCRL_BWDG_ARITH == 0
    if constexpr (C2 == 128 && C3 == 256 && (C1 == 64 || C1 == 128)) {
        if (p.fused == 2 || (p.fused == 1 && p.cl.B * bwdg_tpc(C3) <= 2 * num_cus())) {
            // the team kernel (encoder_bwd_fused.h): chain + weight-gradient sums in one launch, then the reduce over the workgroup rows
            if (p.schedule_out) *p.schedule_out = 3;
            auto kern = encoder_bwdg_fused_kernel<T0, C1, C2, C3>;
            const size_t flds = fused_lds_bytes(T0, C1, C2, C3);
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), flds)) return rc;
            // one workgroup per CU is resident (four waves of up to 512 registers: __launch_bounds__(256, 1)); tiles are dealt round-robin
            const int grid = std::min(std::min(num_cus(), p.fused_rows), p.cl.B * bwdg_tpc(C3));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kFusedWaves), flds, stream, p);
            PCRL_BWDG_AFTER("encoder_bwdg_fused_kernel");
            const GradLayout GL{p.cl.C, C1, C2, C3};
            const FusedRow FR{p.cl.C, C1, C2, C3};
            const int row_blocks = (FR.total() + 63) / 64, main_blocks = row_blocks + C3 * C2 / 64;
            hipLaunchKernelGGL(encoder_bwdg_reduce_fused_kernel<C2>, dim3(main_blocks + p.cs_blocks), dim3(1024), 0, stream, p.wgrows, grid, p.n_items, FR, GL,
                               p.srows, p.chc, p.cl.B, p.grads, p.gvu, p.cs, row_blocks, main_blocks);
            PCRL_BWDG_AFTER("encoder_bwdg_reduce_fused_kernel");
            constexpr int rows = (256 / C2) * 2;
            constexpr size_t fin_lds = sizeof(float) * ((size_t)C2 * C2 + (size_t)rows * C2);
            auto fin = encoder_bwdg_finish_kernel<C2>;
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(fin), fin_lds)) return rc;
            hipLaunchKernelGGL(fin, dim3((C3 + rows - 1) / rows), dim3(256), fin_lds, stream, p.w2, p.gvu, C3, p.grads + GL.w2());
            PCRL_BWDG_AFTER("encoder_bwdg_finish_kernel");
            return PCRL_OK;
        }
    }
#endif
    const size_t lds = bwdg_lds_bytes_points(T0, C1, C2, C3);
    {   // (an eight-wave build -- two tiles in flight per SIMD, 256 registers each -- spilled ~150 registers and measured slower at
        // every batch size: B 256 175 vs 164 us, K3's 1024 clouds 864 vs 785 us, 512 x 8192 388 vs 339 us)
        auto kern = encoder_bwdg_points_kernel<T0, C1, C2, C3, kSplit>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
        hipLaunchKernelGGL(kern, dim3(num_cus()), dim3(64 * kBwdgWaves), lds, stream, p);
    }
    PCRL_BWDG_AFTER("encoder_bwdg_points_kernel");
    // h1 of one cloud (C2 / 32 blocks x up to 32 KB) + the per-slot coefficients + the per-channel tables
    constexpr size_t wgrad_lds = (size_t)(C2 / 32) * 32 * 64 * sizeof(f32x4) + 3 * 64 * sizeof(f32x4) + 256 * 4 + (size_t)C3 * 8 + 16;
    auto wgrad = encoder_bwdg_wgrad_kernel<C1, C2, C3>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(wgrad), wgrad_lds)) return rc;
    WgradSched sched;
    make_wgrad_sched<C1 / 32, C2 / 32, PCRL_BWDG_W1NB(C1 / 32)>(p.parts, &sched);
    hipLaunchKernelGGL(wgrad, dim3(min(p.cl.B, num_cus()) * p.parts), dim3(512), wgrad_lds, stream, p, sched);
    PCRL_BWDG_AFTER("encoder_bwdg_wgrad_kernel");
    const GradLayout GL{p.cl.C, C1, C2, C3};
    const GramExtra GX{C2};
    const int stride = GL.total() + GX.total();
    hipLaunchKernelGGL(encoder_bwdg_reduce_kernel<PCRL_BWDG_ARITH>, dim3((stride + 63) / 64 + p.cs_blocks), dim3(1024), 0, stream, p.pw, p.cl.B, stride, GL.total(), p.grads, p.gvu,
                       p.cs, (stride + 63) / 64);
    PCRL_BWDG_AFTER("encoder_bwdg_reduce_kernel");
    {
        constexpr int rows = (256 / C2) * 2;
        constexpr size_t fin_lds = sizeof(float) * ((size_t)C2 * C2 + (size_t)rows * C2);
        auto fin = encoder_bwdg_finish_kernel<C2>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(fin), fin_lds)) return rc;
        hipLaunchKernelGGL(fin, dim3((C3 + rows - 1) / rows), dim3(256), fin_lds, stream, p.w2, p.gvu, C3, p.grads + GL.w2());
    }
    PCRL_BWDG_AFTER("encoder_bwdg_finish_kernel");
    return PCRL_OK;
}

int PCRL_BWDG_LAUNCH_NAME(int T0, int c1, int c2, int c3, const BwdParams& p, hipStream_t st) {
    int rc = PCRL_E_ARG;
#define PCRL_BWDG_CASE(T0_, C1_, C2_, C3_) \
    if (T0 == T0_ && c1 == C1_ && c2 == C2_ && c3 == C3_) rc = launch_bwdg<T0_, C1_, C2_, C3_>(p, st);
    PCRL_BWDG_CASE(3, 64, 128, 256) PCRL_BWDG_CASE(4, 128, 128, 256)
#ifndef PCRL_BWDG_FEWER
    PCRL_BWDG_CASE(2, 64, 128, 256) PCRL_BWDG_CASE(4, 64, 128, 256) PCRL_BWDG_CASE(5, 64, 128, 256)
    PCRL_BWDG_CASE(2, 128, 128, 256) PCRL_BWDG_CASE(3, 128, 128, 256) PCRL_BWDG_CASE(5, 128, 128, 256)
    PCRL_BWDG_CASE(2, 32, 64, 128) PCRL_BWDG_CASE(3, 32, 64, 128) PCRL_BWDG_CASE(4, 32, 64, 128) PCRL_BWDG_CASE(5, 32, 64, 128)
#endif
#undef PCRL_BWDG_CASE
#if PCRL_BWDG_ARITH == 0 && !defined(PCRL_BWDG_FEWER)
    // the class default mlp_spec = [64, 128, 1024] (fp32 only)
    if (c1 == 64 && c2 == 128 && c3 == 1024) {
        if (T0 == 2) rc = launch_bwdg_wide<2, 64, 128, 1024>(p, st);
        if (T0 == 3) rc = launch_bwdg_wide<3, 64, 128, 1024>(p, st);
        if (T0 == 4) rc = launch_bwdg_wide<4, 64, 128, 1024>(p, st);
        if (T0 == 5) rc = launch_bwdg_wide<5, 64, 128, 1024>(p, st);
    }
#endif
    return rc;
}
#endif  // PCRL_BWD_MODE == 4

}  // namespace pcrl

#if defined(PCRL_BWDG_STAMPS) && PCRL_BWD_MODE == 4 && PCRL_BWDG_ARITH == 0
extern "C" int pcrl_debug_bwdg_stamps(unsigned long long* tiles_out, int n_items, unsigned long long* waves_out, int n_waves) {
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    if (hipMemcpyFromSymbol(tiles_out, HIP_SYMBOL(pcrl::g_bwdg_stamps), sizeof(unsigned long long) * 12 * (size_t)n_items) != hipSuccess) return -3;
    if (hipMemcpyFromSymbol(waves_out, HIP_SYMBOL(pcrl::g_bwdg_wstamps), sizeof(unsigned long long) * 8 * (size_t)n_waves) != hipSuccess) return -3;
    return 0;
}
extern "C" int pcrl_debug_fused_stamps(unsigned long long* out, int n_items) {
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pcrl::g_fused_stamps), sizeof(unsigned long long) * 96 * (size_t)n_items) != hipSuccess) return -3;
    return 0;
}
#endif

