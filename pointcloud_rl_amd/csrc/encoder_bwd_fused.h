// PointNet encoder backward, Gram form, ONE kernel for the per-point chain AND the weight-gradient sums (fp32) -- round 5.
//
// Same algebra as encoder_bwd_gram.h (read its header first).  What changes is who does what:
//
//   * a 32-point tile is worked by a TEAM of four waves (one workgroup = one team = one CU: the waves use up to 512 registers), wave w owning
//     row block w of every C2-wide quantity (h1, q = Mc h1, dH1, dz1).  Per-point scalars that need all C2 channels (LayerNorm-1's mean /
//     variance and its backward sums, mu = s.h1 / C3, var = h1.q / C3, the owned channels' dot products W2[c,:].h1) are added
//     over the four waves through small LDS arrays, six workgroup barriers per tile (LayerNorm-1's statistics in one exchange);
//   * each wave's row block is transposed ONCE into LDS (pitch 33 float4: both directions conflict-free, store_block_pieces' layout)
//     and serves twice from there: as the B operand of the next layer for the other three waves (q = Mc h1, dH0 = W1^T dz1), and as
//     the A / B operand pieces of the weight-gradient blocks -- which are therefore accumulated where their operands are produced:
//     G (the ten upper 32 x 32 blocks), dW1, dW0 | b0 live in registers across all the tiles a workgroup takes, v / u in one register
//     per thread, norm1 sums per lane, norm2 sums in LDS, and leave once per workgroup.  No operand pieces in global memory (84 MB per K1 launch
//     before), no per-cloud wgrad launch; the reduce launch adds <= #CUs workgroup rows instead of B cloud rows;
//   * the sparse rows of dW2, S[c,:] = rstd2 dx_c h1_p(c), still leave per (cloud, channel) -- one point owns the row, there is nothing to
//     accumulate -- and the reduce launch adds them over the clouds, skipping the channels the forward left dead.
//
// Sums over the four waves are formed in wave order, the tiles of a workgroup in tile order, the workgroup rows in row order: bitwise
// reproducible for a given launch geometry; the split of a 128-term sum into 4 x 32 differs from encoder_bwd_gram.h's in the last bits.
// Exact-fp32 arithmetic only (mode 0; the bf16 mode's backward runs these kernels too, see encoder_bwd_impl.h).
// Used for launches of at most two tiles per CU (pcrl_encoder_bwd_set_fused, include/pcrl.h): one tile at a time per CU is what limits it --
// measurements, per-phase stamps and the builds that tried to lift that limit: profiles/r05_bwd_team.md, DESIGN.md section 4.2.
#pragma once
// (included by encoder_bwd_gram.h inside namespace pcrl, mode 4 / arithmetic 0 only)

constexpr int kFusedWaves = 4;
#ifdef PCRL_BWDG_STAMPS
// development build: shader-clock stamps of every wave at the phase boundaries of a tile (tools/fused_stamps.py)
__device__ unsigned long long g_fused_stamps[8192][4][24];
#define PCRL_FSTAMP(k) do { if (lane == 0 && item < 8192) g_fused_stamps[item][wave][k] = __builtin_readcyclecounter(); } while (0)
#else
#define PCRL_FSTAMP(k) do { } while (0)
#endif
constexpr int kTrBlk = 8 * 33 * 4;                // floats of one transposed 32-channel x 32-point block (= kTrFloats)

// One workgroup's partial sums.  [0, main()) follows GradLayout (conv0.weight, conv0.bias, conv1.weight, norm1.weight, norm1.bias); behind
// it norm2.weight / norm2.bias, the upper blocks of G packed as [block][32][32], v, u.
struct FusedRow {
    int C, C1, C2, C3;
    __host__ __device__ constexpr int main() const { return C1 * C + C1 + C2 * C1 + 2 * C2; }
    __host__ __device__ constexpr int g2() const { return main(); }
    __host__ __device__ constexpr int be2() const { return g2() + C3; }
    __host__ __device__ constexpr int G() const { return be2() + C3; }
    __host__ __device__ constexpr int nG() const { return (C2 / 32) * (C2 / 32 + 1) / 2; }
    __host__ __device__ constexpr int v() const { return G() + nG() * 1024; }
    __host__ __device__ constexpr int u() const { return v() + C2; }
    __host__ __device__ constexpr int total() const { return (u() + C2 + 63) & ~63; }
};
static_assert(FusedRow{6, 64, 128, 256}.total() == fused_row_floats(6, 64, 128, 256) && FusedRow{7, 128, 128, 256}.total() == fused_row_floats(7, 128, 128, 256),
              "the workspace is sized with fused_row_floats (encoder_bwd_gram.h)");
__host__ __device__ constexpr int fused_g_index(int MB2, int i, int j) { return i * MB2 - i * (i - 1) / 2 + (j - i); }   // i <= j

// The weight-gradient blocks a wave accumulates: code = kind << 8 | i << 4 | j, kind 1: G(i, j) (i <= j), 2: dW1 (dz1 block i, h0 block j),
// 3: dW0 | b0 (dz0 block i); 0: none.  G blocks are formed before LayerNorm-1's backward, the others at the end of the tile, where at c1 = 64
// waves 0 and 1 first form dH0 = W1^T dz1 (64 MFMAs each): the dW1 blocks go to waves 2 and 3.
constexpr int FBc(int kind, int i, int j) { return kind << 8 | i << 4 | j; }
template <int MB1> struct FusedBlocks;
template <> struct FusedBlocks<2> {
    static constexpr int NB = 7;
    __host__ __device__ static constexpr int code(int w, int k) {
        // the last phase: waves 0 and 1 form dH0 (64 MFMAs), wave 3 loads the next tile's point features -- wave 2 takes five of the eight dW1 blocks
        constexpr int t[4][7] = {{FBc(1, 0, 0), FBc(1, 0, 1), FBc(1, 0, 2), FBc(3, 0, 0), 0, 0, 0},
                                 {FBc(1, 1, 1), FBc(1, 1, 2), FBc(1, 1, 3), FBc(3, 1, 0), 0, 0, 0},
                                 {FBc(1, 2, 2), FBc(1, 2, 3), FBc(2, 0, 0), FBc(2, 0, 1), FBc(2, 1, 0), FBc(2, 1, 1), FBc(2, 2, 0)},
                                 {FBc(1, 3, 3), FBc(1, 0, 3), FBc(2, 2, 1), FBc(2, 3, 0), FBc(2, 3, 1), 0, 0}};
        return t[w][k];
    }
};
template <> struct FusedBlocks<4> {
    static constexpr int NB = 8;
    __host__ __device__ static constexpr int code(int w, int k) {
        constexpr int t[4][8] = {{FBc(1, 0, 0), FBc(1, 0, 1), FBc(1, 0, 2), FBc(2, 0, 0), FBc(2, 0, 1), FBc(2, 0, 2), FBc(2, 0, 3), FBc(3, 0, 0)},
                                 {FBc(1, 1, 1), FBc(1, 1, 2), FBc(1, 1, 3), FBc(2, 1, 0), FBc(2, 1, 1), FBc(2, 1, 2), FBc(2, 1, 3), FBc(3, 1, 0)},
                                 {FBc(1, 2, 2), FBc(1, 2, 3), 0, FBc(2, 2, 0), FBc(2, 2, 1), FBc(2, 2, 2), FBc(2, 2, 3), FBc(3, 2, 0)},
                                 {FBc(1, 3, 3), FBc(1, 0, 3), 0, FBc(2, 3, 0), FBc(2, 3, 1), FBc(2, 3, 2), FBc(2, 3, 3), FBc(3, 3, 0)}};
        return t[w][k];
    }
};

// A wave's row block in the accumulator layout (lane = point l31 of half h, register r = channel (r & 3) + 8 (r >> 2) + 4 h) to / from its
// transposed LDS block: float4 (2 octet + k-lane) * 33 + channel holds the channel's values at points 4 (2 octet + k-lane) + 0..3.
__device__ __forceinline__ void tr_write(float* blk, const f32x16& v, int l31, int half) {
    float* w = blk + ((l31 >> 2) * 33 + 4 * half) * 4 + (l31 & 3);
#pragma unroll
    for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2)) * 4] = v[r];
}
__device__ __forceinline__ void tr_read_acc(const float* blk, float (&out)[16], int l31, int half) {
    const float* w = blk + ((l31 >> 2) * 33 + 4 * half) * 4 + (l31 & 3);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = w[((r & 3) + 8 * (r >> 2)) * 4];
}
// MFMA operand piece of octet q (8 points): lane (channel = lane & 31, k-lane = lane >> 5) holds the channel at points 8 q + 4 k-lane + 0..3
__device__ __forceinline__ f32x4 tr_piece(const float* blk, int q, int lane) {
    return reinterpret_cast<const f32x4*>(blk)[(2 * q + (lane >> 5)) * 33 + (lane & 31)];
}

// out = A[32 rows of block `blk`][:] . act over K = 128 channels: A pieces stream from L2 (image [block][K / 8][64 lanes][4], six in
// flight, refills pinned), the B operand is the four transposed blocks in LDS (sixteen values per lane and block).
template <int DEPTH>
__device__ __forceinline__ void fused_k128_preload(f32x4 (&ring)[DEPTH], const __amdgpu_buffer_rsrc_t& rs, unsigned img_bytes, int blk, unsigned lane16) {
    const unsigned base = img_bytes + 4u * (unsigned)(blk * 16 * 256);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = buf_load_f4(rs, lane16 + base, 1024u * (unsigned)d);
}
// (ring: the first DEPTH pieces, requested by fused_k128_preload as early as the caller could -- ahead of the phase in front of this layer)
template <int DEPTH>
__device__ __forceinline__ f32x16 fused_layer_k128(f32x4 (&ring)[DEPTH], const __amdgpu_buffer_rsrc_t& rs, unsigned img_bytes, int blk, const float* tr4, int l31, int half,
                                                   unsigned lane16) {
    constexpr int TQ = 16;
    const unsigned base = img_bytes + 4u * (unsigned)(blk * TQ * 256);
    float hb[2][16];
    tr_read_acc(tr4, hb[0], l31, half);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        if (kb + 1 < 4) tr_read_acc(tr4 + (kb + 1) * kTrBlk, hb[(kb + 1) & 1], l31, half);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int tq = 4 * kb + t4;
            const f32x4 w = ring[tq % DEPTH];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], hb[kb & 1][4 * t4 + j], (tq == 0 && j == 0) ? zero : acc, 0, 0, 0);
            }
            if (tq + DEPTH < TQ) ring[tq % DEPTH] = buf_load_f4(rs, lane16 + base, 1024u * (unsigned)(tq + DEPTH));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return acc;
}

template <int MB1, int W, int KIND>
__device__ __forceinline__ void fused_accumulate(f32x16 (&acc)[FusedBlocks<MB1>::NB], const float* R1, const float* R2, const float* R3, const float* R4,
                                                 const f32x4* s_a4, int lane) {
    typedef FusedBlocks<MB1> FB;
    // the operand pieces of octet q + 1 are read while octet q's MFMAs run; blocks that share a piece share its read (identical loads fold)
    f32x4 pa[2][FB::NB], pb[2][FB::NB];
    auto fetch = [&](int q, f32x4 (&A)[FB::NB], f32x4 (&Bv)[FB::NB]) {
        f32x4 a4 = {1.f, 1.f, 1.f, 1.f};
        if (KIND == 1) a4 = s_a4[2 * q + (lane >> 5)];
#pragma unroll
        for (int k = 0; k < FB::NB; ++k) {
            const int code = FB::code(W, k), kind = code >> 8, i = (code >> 4) & 15, j = code & 15;
            if (kind != KIND) continue;
            if (KIND == 1) { A[k] = tr_piece(R1 + i * kTrBlk, q, lane) * a4; Bv[k] = tr_piece(R1 + j * kTrBlk, q, lane); }
            else if (KIND == 2) { A[k] = tr_piece(R1 + i * kTrBlk, q, lane); Bv[k] = tr_piece(R2 + j * kTrBlk, q, lane); }
            else { A[k] = tr_piece(R4 + i * kTrBlk, q, lane); Bv[k] = tr_piece(R3, q, lane); }
        }
    };
    fetch(0, pa[0], pb[0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q + 1 < 4) fetch(q + 1, pa[(q + 1) & 1], pb[(q + 1) & 1]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < FB::NB; ++k) {
                const int kind = FB::code(W, k) >> 8;
                if (kind != KIND) continue;
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[q & 1][k][e], pb[q & 1][k][e], acc[k], 0, 0, 0);
            }
    }
}

template <int MB1, int W>
__device__ __forceinline__ void fused_flush(const f32x16 (&acc)[FusedBlocks<MB1>::NB], float* row, const FusedRow& FR, const GradLayout& GL, int C, int lane) {
    typedef FusedBlocks<MB1> FB;
#pragma unroll
    for (int k = 0; k < FB::NB; ++k) {
        const int code = FB::code(W, k), kind = code >> 8, i = (code >> 4) & 15, j = code & 15;
        if (kind == 1) store_tile(row + FR.G() + fused_g_index(4, i, j) * 1024, 32, 0, 0, 32, acc[k], lane);
        else if (kind == 2) store_tile(row + GL.w1(), 32 * MB1, i, j, 32 * MB1, acc[k], lane);
        else if (kind == 3) {
            const int col = lane & 31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rw = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (col < C) row[GL.w0() + rw * C + col] = acc[k][r];
                else if (col == C) row[GL.b0() + rw] = acc[k][r];
            }
        }
    }
}

constexpr int kFusedRound = 16;                 // owned channels of a point per round of the dot-product exchange (a point owns ~1.5)
__host__ __device__ constexpr int fused_lds_floats(int T0, int C1, int kC2, int kC3) {
    return (4 + 2 * (C1 / 32) + 1) * kTrBlk                                     // R1 (h1 / dz1), R2 (h0), R4 (dz0), R3 (x | 1)
           + 6 * 4 * 32 + kFusedRound * 4 * 32 + 2 * kFusedRound * 32 + 4 * 3 * 32  // s_red, s_dot, s_dx / s_dy, s_coef
           + 2 * kC3 + 16 * 32                                                 // s_g2, s_be2, the next tile's point features
           + 2 * kC2 + kC3 + C1 + (C1 / 32) * T0 * 64 + kC2                    // ln1, gamma2, b0, w0, s
           + kMaxTileModeClouds + 8;                                           // tile prefix
}
static size_t fused_lds_bytes(int T0, int C1, int kC2, int kC3) {
    return sizeof(float) * (size_t)fused_lds_floats(T0, C1, kC2, kC3) + sizeof(ChanSrc) * PCRL_MAX_CHANNELS;
}

template <int T0, int C1, int kC2, int kC3>
__global__ __launch_bounds__(64 * kFusedWaves, 1) void encoder_bwdg_fused_kernel(const BwdParams p) {
    static_assert(kC2 == 128 && kC3 <= 256 && (C1 == 64 || C1 == 128), "team kernel: c2 = 128 (four row blocks = four waves), c3 <= 256");
    constexpr int NW = kFusedWaves;
    constexpr PackedLayout L{T0, C1, kC2, kC3};
    constexpr int MB1 = C1 / 32;
    typedef FusedBlocks<MB1> FB;
    const GradLayout GL{p.cl.C, C1, kC2, kC3};
    const FusedRow FR{p.cl.C, C1, kC2, kC3};
    typedef unsigned char idx_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* R1 = reinterpret_cast<float*>(smem);                    // [4][kTrBlk]   h1, later dz1
    float* R2 = R1 + 4 * kTrBlk;                                   // [MB1][kTrBlk] h0
    float* R4 = R2 + MB1 * kTrBlk;                                 // [MB1][kTrBlk] dz0
    float* R3 = R4 + MB1 * kTrBlk;                                 // [kTrBlk]      x | 1
    float* s_red = R3 + kTrBlk;                                    // [6][4][32]
    float* s_dot = s_red + 6 * 4 * 32;                             // [kFusedRound][4][32]
    float* s_dx = s_dot + kFusedRound * 4 * 32;                    // [kFusedRound][32] dx_c of the round's channels (wave 0 writes)
    float* s_dy = s_dx + kFusedRound * 32;                         // [kFusedRound][32] grad_pooled[c]
    float* s_coef = s_dy + kFusedRound * 32;                       // [4 waves][3][32]: a, rstd2 m1, a mu of the tile's points
    float* s_g2 = s_coef + 4 * 3 * 32;                             // [kC3] norm2.weight gradient of this workgroup's tiles
    float* s_be2 = s_g2 + kC3;
    float* s_x = s_be2 + kC3;                                      // [16][32] the next tile's point features (wave 3 loads them)
    float* s_ln1 = s_x + 16 * 32;                                  // [kC2][2] (gamma, beta)
    float* s_gam2 = s_ln1 + 2 * kC2;                               // [kC3]
    float* s_b0 = s_gam2 + kC3;
    float* s_w0 = s_b0 + C1;
    float* s_sv = s_w0 + MB1 * T0 * 64;
    int* s_tstart = reinterpret_cast<int*>(s_sv + kC2);            // [kMaxTileModeClouds + 8]
    ChanSrc* s_desc = reinterpret_cast<ChanSrc*>(s_tstart + kMaxTileModeClouds + 8);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31;
    {
        // every global load of the prologue in flight before the first LDS write (written as one loop per array each loop paid its own round
        // trip): the clouds' tile counts, the per-channel tables, conv0's operands
        constexpr int PER = kMaxTileModeClouds / (64 * NW), W0N = (MB1 * T0 * 64 + 64 * NW - 1) / (64 * NW);
        int nt[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) nt[k] = tid + 64 * NW * k < p.cl.B ? p.n_act[tid + 64 * NW * k] : 0;
        static_assert(2 * kC2 <= 64 * NW && kC3 <= 64 * NW && C1 <= 64 * NW, "one prologue element per thread");
        const float v_sv = tid < kC2 ? p.mimg[kC2 * kC2 + tid] : 0.0f;
        const float v_b0 = tid < C1 ? p.packed[L.b0() + tid] : 0.0f;
        const float v_ln1 = tid < 2 * kC2 ? p.packed[L.ln1() + tid] : 0.0f;
        const float v_g2 = tid < kC3 ? p.packed[L.ln2() + 2 * tid] : 0.0f;
        float v_w0[W0N];
#pragma unroll
        for (int k = 0; k < W0N; ++k) v_w0[k] = tid + 64 * NW * k < MB1 * T0 * 64 ? p.packed[L.w0() + tid + 64 * NW * k] : 0.0f;
        ChanSrc v_desc{};
        if (tid < PCRL_MAX_CHANNELS) v_desc = p.cl.ch[tid];
        if (tid < kC2) s_sv[tid] = v_sv;
        if (tid < C1) s_b0[tid] = v_b0;
        if (tid < 2 * kC2) s_ln1[tid] = v_ln1;
        if (tid < kC3) { s_gam2[tid] = v_g2; s_g2[tid] = 0.0f; s_be2[tid] = 0.0f; }
#pragma unroll
        for (int k = 0; k < W0N; ++k)
            if (tid + 64 * NW * k < MB1 * T0 * 64) s_w0[tid + 64 * NW * k] = v_w0[k];
        for (int i = tid; i < kTrBlk; i += 64 * NW) R3[i] = 0.0f;
        if (tid < PCRL_MAX_CHANNELS) s_desc[tid] = v_desc;
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (tid + 64 * NW * k < p.cl.B) s_tstart[tid + 64 * NW * k + 1] = (nt[k] + 31) >> 5;
        __syncthreads();
        if (wave == 0) {          // exclusive prefix of the counts (in place: entry b + 1 holds cloud b's count), 64 clouds per scan step
            int run = 0;
            for (int b0 = 0; b0 < p.cl.B; b0 += 64) {
                const int b = b0 + lane;
                const int ntb = b < p.cl.B ? s_tstart[b + 1] : 0;
                int inc = ntb;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(inc, off, 64); if (lane >= off) inc += up; }
                if (b < p.cl.B) s_tstart[b + 1] = run + inc;          // inclusive prefix = exclusive prefix of cloud b + 1
                run += __shfl(inc, 63, 64);
            }
            if (lane == 0) s_tstart[0] = 0;
        }
    }
    __syncthreads();
    const int n_items = s_tstart[p.cl.B];
    if (blockIdx.x == 0 && tid == 0) *p.n_items = n_items;
    if ((int)blockIdx.x >= n_items) return;            // this workgroup has no tile: no row (the reduce launch reads min(grid, n_items) rows)

    const __amdgpu_buffer_rsrc_t r_packed = make_rsrc(p.packed, 4u * (unsigned)L.total());
    const __amdgpu_buffer_rsrc_t r_mimg = make_rsrc(p.mimg, 4u * (unsigned)(kC2 * kC2));
    const unsigned lane16 = 16u * (unsigned)lane;
    const unsigned half_mask = half ? 0xFFFFFFFFu : 0u;

    f32x16 acc[FB::NB];
#pragma unroll
    for (int k = 0; k < FB::NB; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    float vu_acc = 0.0f;
    float n1g[16], n1b[16];        // norm1.weight / norm1.bias gradients of this lane's point over the workgroup's tiles (32-lane sums at the end)
#pragma unroll
    for (int r = 0; r < 16; ++r) { n1g[r] = 0.0f; n1b[r] = 0.0f; }

    // cloud and tile of an item: the largest b with s_tstart[b] <= item
    auto decode = [&](int item, int& b, int& tile) {
        int lo = 0, hi = p.cl.B;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_tstart[mid] <= item) lo = mid; else hi = mid; }
        b = __builtin_amdgcn_readfirstlane(lo);
        tile = __builtin_amdgcn_readfirstlane(item - s_tstart[lo]);
    };
    // Software pipeline over the workgroup's tiles: the slot tables of the NEXT tile are requested at the top of a tile and its point
    // features behind q = Mc h1 (the prep launch pads a cloud's last tile with copies of its last active point that own nothing, and packs
    // the first eight channels of a slot into one word: none of these addresses depends on a loaded value but the point index).
    int b, tile;
    decode((int)blockIdx.x, b, tile);
    int pidx = p.act[(long long)b * kC3 + 32 * tile + l31];
    unsigned own_w = p.own[(long long)b * kC3 + 32 * tile + l31];
    unsigned long long pack = p.own_pack[(long long)b * kC3 + 32 * tile + l31];
    f32x16 x = load_point<T0>(p.cl, s_desc, b, pidx);

#pragma unroll 1
    for (int item = blockIdx.x; item < n_items; item += (int)gridDim.x) {
        const float* g_row = p.gpool + (long long)b * kC3;
        const int n_act = __builtin_amdgcn_readfirstlane(p.n_act[b]);
        const bool valid = 32 * tile + l31 < n_act;
        // next tile (the last tile of the workgroup requests its own again: harmless)
        const int nitem = item + (int)gridDim.x < n_items ? item + (int)gridDim.x : item;
        int nb, ntile;
        decode(nitem, nb, ntile);
        const int npidx = p.act[(long long)nb * kC3 + 32 * ntile + l31];
        const unsigned nown = p.own[(long long)nb * kC3 + 32 * ntile + l31];
        const unsigned long long npack = p.own_pack[(long long)nb * kC3 + 32 * ntile + l31];
        PCRL_FSTAMP(0);

        // conv1's first operand pieces are requested ahead of conv0 (their round trip hid nothing at the head of conv1)
        constexpr int TQ1 = C1 / 8, DEPTH1 = 6;
        const unsigned base1 = 4u * (unsigned)L.w1() + 4u * (unsigned)(wave * TQ1 * 256);
        f32x4 ring1[DEPTH1];
#pragma unroll
        for (int d = 0; d < DEPTH1; ++d) ring1[d] = buf_load_f4(r_packed, lane16 + base1, 1024u * (unsigned)d);
        __builtin_amdgcn_sched_barrier(0);
        // ---- conv0 + ReLU: every wave forms all of h0 (the B operand of its conv1 block) ---------------------------------------------
        f32x16 a0[MB1];
        unsigned mask_own = 0u;
#pragma unroll
        for (int mb = 0; mb < MB1; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a0[mb][r] = s_b0[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
            for (int t = 0; t < T0; ++t) {
                const float bop = half_select(x[2 * t], x[2 * t + 1], half_mask);
                a0[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_w0[(mb * T0 + t) * 64 + lane], bop, a0[mb], 0, 0, 0);
            }
            unsigned m = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                a0[mb][r] = relu_nan(a0[mb][r]);
                m |= (a0[mb][r] > 0.0f ? 1u : 0u) << r;
            }
            mask_own = wave == mb ? m : mask_own;
        }
        PCRL_FSTAMP(1);
        // ---- conv1, row block `wave`: the forward's chain for that block (same order: the recompute of the block is bit-identical) -------
        f32x16 a1;
        {
            constexpr int TQ = TQ1, DEPTH = DEPTH1;
            const unsigned base = base1;
            f32x4 (&ring)[DEPTH1] = ring1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq) {
                const f32x4 w = ring[tq % DEPTH];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = 4 * tq + j;
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], a0[t >> 4][t & 15], t == 0 ? zero : a1, 0, 0, 0);
                }
                if (tq + DEPTH < TQ) ring[tq % DEPTH] = buf_load_f4(r_packed, lane16 + base, 1024u * (unsigned)(tq + DEPTH));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PCRL_FSTAMP(2);
        // ---- LayerNorm-1 over the four waves' blocks, one exchange: every wave centres its 32 channels on their own mean, the four
        // (mean, centred sum of squares) pairs combine exactly (mean = sum m_w / 4, M2 = sum M2_w + 32 sum (m_w - mean)^2) -------------------
        float lo, hi, mean_w;
        {
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ps = ps + a1[r];
            both_halves(ps, lo, hi);
            mean_w = (lo + hi) / 32.0f;
            float pq = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a1[r] = a1[r] - mean_w; pq = __builtin_fmaf(a1[r], a1[r], pq); }
            both_halves(pq, lo, hi);
            if (half == 0) { s_red[(0 * 4 + wave) * 32 + l31] = mean_w; s_red[(1 * 4 + wave) * 32 + l31] = lo + hi; }
        }
        __syncthreads();                                                                                  // B1 (also: the previous tile is done everywhere)
        PCRL_FSTAMP(3);
        // h0 and x | 1 of this tile for the weight-gradient blocks (behind B1: nobody still reads the previous tile's)
#pragma unroll
        for (int mb = 0; mb < MB1; ++mb)
            if (wave == mb) tr_write(R2 + mb * kTrBlk, a0[mb], l31, half);
        if (wave == NW - 1 && half == 0) {
            float* w = R3 + ((l31 >> 2) * 33) * 4 + (l31 & 3);
#pragma unroll
            for (int c = 0; c < 2 * T0; ++c)
                if (c < p.cl.C) w[c * 4] = x[c];
            w[p.cl.C * 4] = 1.0f;
        }
        float rstd1;
        {
            const float m0 = s_red[(0 * 4 + 0) * 32 + l31], m1 = s_red[(0 * 4 + 1) * 32 + l31], m2 = s_red[(0 * 4 + 2) * 32 + l31], m3 = s_red[(0 * 4 + 3) * 32 + l31];
            const float mean1 = ((m0 + m1) + (m2 + m3)) / 4.0f;
            const float d0 = m0 - mean1, d1 = m1 - mean1, d2 = m2 - mean1, d3 = m3 - mean1;
            const float between = 32.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            const float within = (s_red[(1 * 4 + 0) * 32 + l31] + s_red[(1 * 4 + 1) * 32 + l31]) + (s_red[(1 * 4 + 2) * 32 + l31] + s_red[(1 * 4 + 3) * 32 + l31]);
            rstd1 = 1.0f / __builtin_sqrtf((within + between) / (float)kC2 + p.eps);
            mean_w = mean_w - mean1;                       // what this wave's centred values still miss
        }
        PCRL_FSTAMP(4);
        f32x16 xh1;
        {
            float ps = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = acc_chan(wave * 16 + r, 0) + 4 * half;
                const float2 gb = reinterpret_cast<const float2*>(s_ln1)[ch];
                xh1[r] = (a1[r] + mean_w) * rstd1;
                a1[r] = relu_nan(__builtin_fmaf(xh1[r], gb.x, gb.y));          // h1
                ps = __builtin_fmaf(a1[r], s_sv[ch], ps);
            }
            both_halves(ps, lo, hi);
            if (half == 0) s_red[(2 * 4 + wave) * 32 + l31] = lo + hi;       // this block's share of s.h1
        }
        tr_write(R1 + wave * kTrBlk, a1, l31, half);
        __syncthreads();                                                                                  // B3
        PCRL_FSTAMP(5);
        // ---- the channels this point owns, first pass: the block's share of W2[c,:].h1 (to LDS) and of sum dx_c W2[c,:]; the W2 rows
        // of channel i + 1 are in flight while channel i is worked --------------------------------------------------------------------------
        f32x4 ringq[6];              // q = Mc h1: its first operand pieces travel under the first pass over the owned channels
        fused_k128_preload<6>(ringq, r_mimg, 0u, wave, lane16);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 gacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) gacc[r] = 0.0f;
        const int cnt = (int)(own_w >> 16);
        const idx_t* oc = reinterpret_cast<const idx_t*>(p.own_chan) + (long long)b * kC3 + (int)(own_w & 0xFFFFu);
        const int cmax = __builtin_amdgcn_readfirstlane((int)allreduce_umax32((unsigned)cnt));
        float t1 = 0.0f, t2r = 0.0f, mu = 0.0f, rstd2 = 0.0f;
        f32x16 q;
        float* srow = p.srows + (long long)b * kC3 * kC2 + 32 * wave + 4 * half;
        auto chan_of = [&](int i) -> int {              // channel i of this lane's point (i < cnt), else 0
            int c = i < 8 ? (int)((pack >> (8 * i)) & 0xFFull) : (i < cnt ? (int)oc[i] : 0);
            return i < cnt ? c : 0;
        };
        for (int r0 = 0; r0 == 0 || r0 < cmax; r0 += kFusedRound) {
            const int nr = cmax - r0 < kFusedRound ? cmax - r0 : kFusedRound;
            {
                f32x4 wA[4], wB[4];
                float dyA, dyB;
                int cA, cB;
                auto fetch = [&](int i, f32x4 (&w4)[4], float& dy, int& c) {
                    c = chan_of(i);
                    const f32x4* wrow = reinterpret_cast<const f32x4*>(p.w2 + (long long)c * kC2 + 32 * wave + 4 * half);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) w4[g4] = wrow[2 * g4];
                    dy = i < cnt ? g_row[c] : 0.0f;
                };
                auto work = [&](int i, const f32x4 (&w4)[4], float dy, int c) {
                    const float dx = dy * s_gam2[c];
                    float d = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        d = __builtin_fmaf(w4[r >> 2][r & 3], a1[r], d);
                        gacc[r] = __builtin_fmaf(dx, w4[r >> 2][r & 3], gacc[r]);
                    }
                    both_halves(d, lo, hi);
                    if (half == 0) {
                        s_dot[((i - r0) * 4 + wave) * 32 + l31] = lo + hi;
                        if (wave == 0) { s_dx[(i - r0) * 32 + l31] = dx; s_dy[(i - r0) * 32 + l31] = dy; }
                    }
                    t1 = t1 + dx;
                };
                if (nr > 0) fetch(r0, wA, dyA, cA);
                for (int i = r0; i < r0 + nr; i += 2) {
                    fetch(i + 1, wB, dyB, cB);
                    work(i, wA, dyA, cA);
                    if (i + 1 < r0 + nr) {
                        fetch(i + 2, wA, dyA, cA);
                        work(i + 1, wB, dyB, cB);
                    }
                }
            }
            if (r0 == 0) {
                PCRL_FSTAMP(6);
                // ---- q = Mc h1, row block `wave`; its share of h1.q --------------------------------------------------------------------
                q = fused_layer_k128<6>(ringq, r_mimg, 0u, wave, R1, l31, half, lane16);
                float pe = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) pe = __builtin_fmaf(a1[r], q[r], pe);
                both_halves(pe, lo, hi);
                if (half == 0) s_red[(3 * 4 + wave) * 32 + l31] = lo + hi;
                PCRL_FSTAMP(15);
            }
            __syncthreads();                                                                              // B4 (+ two per further round)
            if (r0 == 0) PCRL_FSTAMP(16);
            if (r0 == 0) {
                mu = ((s_red[(2 * 4 + 0) * 32 + l31] + s_red[(2 * 4 + 1) * 32 + l31]) + (s_red[(2 * 4 + 2) * 32 + l31] + s_red[(2 * 4 + 3) * 32 + l31])) / (float)kC3;
                const float e = ((s_red[(3 * 4 + 0) * 32 + l31] + s_red[(3 * 4 + 1) * 32 + l31]) + (s_red[(3 * 4 + 2) * 32 + l31] + s_red[(3 * 4 + 3) * 32 + l31])) / (float)kC3;
                rstd2 = 1.0f / __builtin_sqrtf(__builtin_fmaxf(e, 0.0f) + p.eps);
            }
            // second pass: z_c - mu from the four shares, the two LayerNorm-2 sums, the sparse row of dW2, (wave 0) norm2's own gradients
            for (int i = r0; i < r0 + nr; ++i) {
                const float* dk = s_dot + ((i - r0) * 4) * 32 + l31;
                const float zc = ((dk[0] + dk[32]) + (dk[64] + dk[96])) - mu;
                const float dx = s_dx[(i - r0) * 32 + l31];
                t2r = __builtin_fmaf(dx, zc, t2r);
                if (i < cnt) {
                    const int c = chan_of(i);
                    const float co = dx != 0.0f ? dx * rstd2 : 0.0f;
                    f32x4* dst = reinterpret_cast<f32x4*>(srow + (long long)c * kC2);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = co != 0.0f ? co * a1[4 * g4 + e] : 0.0f;
                        dst[2 * g4] = v;
                    }
                    if (wave == 0 && half == 0) {          // exactly one point per (cloud, channel): plain read-modify-write, tile order
                        const float dy = s_dy[(i - r0) * 32 + l31];
                        const float raw = dy * zc;
                        s_g2[c] = s_g2[c] + (raw != 0.0f ? raw * rstd2 : 0.0f);
                        s_be2[c] = s_be2[c] + dy;
                        p.chc[(long long)b * kC3 + c] = dx;
                    }
                }
            }
            if (r0 + kFusedRound < cmax) __syncthreads();       // the next round overwrites the shares
        }
        PCRL_FSTAMP(17);
        PCRL_FSTAMP(7);
        // ---- dH1 block; LayerNorm-1 backward sums; G, v, u of this tile --------------------------------------------------------------------
        {
            const float m1 = t1 / (float)kC3, m2 = (rstd2 * t2r) / (float)kC3;
            const float a = valid ? (rstd2 * rstd2) * m2 : 0.0f, vco = valid ? rstd2 * m1 : 0.0f, uco = a * mu;
            if (half == 0) {
                float* cw = s_coef + wave * 96 + l31;
                cw[0] = a; cw[32] = vco; cw[64] = uco;
            }
            const float cs = -vco;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sj = s_sv[acc_chan(wave * 16 + r, 0) + 4 * half];
                q[r] = __builtin_fmaf(rstd2, gacc[r], __builtin_fmaf(cs, sj, -(a * q[r])));
            }
        }
        {
            float s1 = 0.0f, s2 = 0.0f, tg[16], tb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 gb = reinterpret_cast<const float2*>(s_ln1)[acc_chan(wave * 16 + r, 0) + 4 * half];
                const float y = __builtin_fmaf(xh1[r], gb.x, gb.y);
                const float dyl = y > 0.0f ? q[r] : 0.0f;
                tg[r] = dyl * xh1[r];
                tb[r] = dyl;
                const float dx = dyl * gb.x;
                q[r] = dx;
                s1 = s1 + dx;
                s2 = __builtin_fmaf(dx, xh1[r], s2);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { n1g[r] = n1g[r] + tg[r]; n1b[r] = n1b[r] + tb[r]; }
            both_halves(s1, lo, hi);
            if (half == 0) s_red[(4 * 4 + wave) * 32 + l31] = lo + hi;
            both_halves(s2, lo, hi);
            if (half == 0) s_red[(5 * 4 + wave) * 32 + l31] = lo + hi;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        PCRL_FSTAMP(8);
        {
            const f32x4* s_a4 = reinterpret_cast<const f32x4*>(s_coef + wave * 96);
            if (wave == 0) fused_accumulate<MB1, 0, 1>(acc, R1, R2, R3, R4, s_a4, lane);
            else if (wave == 1) fused_accumulate<MB1, 1, 1>(acc, R1, R2, R3, R4, s_a4, lane);
            else if (wave == 2) fused_accumulate<MB1, 2, 1>(acc, R1, R2, R3, R4, s_a4, lane);
            else fused_accumulate<MB1, 3, 1>(acc, R1, R2, R3, R4, s_a4, lane);
            // v (threads 0..127) and u: one dot product over the tile's 32 points per thread, coefficients from this wave's copy
            const int j = tid & (kC2 - 1), which = tid >> 7;
            const f32x4* hj = reinterpret_cast<const f32x4*>(R1 + (j >> 5) * kTrBlk) + (j & 31);
            const f32x4* co = reinterpret_cast<const f32x4*>(s_coef + wave * 96 + 32 + 32 * which);
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 8; ++g) a4 = __builtin_elementwise_fma(hj[g * 33], co[g], a4);
            vu_acc = vu_acc + ((a4[0] + a4[1]) + (a4[2] + a4[3]));
        }
        PCRL_FSTAMP(9);
        __syncthreads();                                                                                  // B5: h1's blocks are free
        PCRL_FSTAMP(10);
        {
            const float n1 = ((s_red[(4 * 4 + 0) * 32 + l31] + s_red[(4 * 4 + 1) * 32 + l31]) + (s_red[(4 * 4 + 2) * 32 + l31] + s_red[(4 * 4 + 3) * 32 + l31])) / (float)kC2;
            const float n2 = ((s_red[(5 * 4 + 0) * 32 + l31] + s_red[(5 * 4 + 1) * 32 + l31]) + (s_red[(5 * 4 + 2) * 32 + l31] + s_red[(5 * 4 + 3) * 32 + l31])) / (float)kC2;
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r] = rstd1 * ((q[r] - n1) - xh1[r] * n2);                      // dz1
        }
        f32x4 ringd[6];              // dH0 = W1^T dz1 (waves < MB1): its first operand pieces travel under the transposition and B6
        if (wave < MB1) fused_k128_preload<6>(ringd, r_packed, 4u * (unsigned)L.w1t(), wave, lane16);
        __builtin_amdgcn_sched_barrier(0);
        tr_write(R1 + wave * kTrBlk, q, l31, half);
        __syncthreads();                                                                                  // B6
        PCRL_FSTAMP(11);
        // ---- dH0 = W1^T dz1 (waves < MB1, one row block each) -> dz0; the dW1 blocks ---------------------------------------------------------
        if (wave < MB1) {
            f32x16 d0 = fused_layer_k128<6>(ringd, r_packed, 4u * (unsigned)L.w1t(), wave, R1, l31, half, lane16);
#pragma unroll
            for (int r = 0; r < 16; ++r) d0[r] = ((mask_own >> r) & 1u) ? d0[r] : 0.0f;
            tr_write(R4 + wave * kTrBlk, d0, l31, half);
        }
        if (wave == 0) fused_accumulate<MB1, 0, 2>(acc, R1, R2, R3, R4, nullptr, lane);
        else if (wave == 1) fused_accumulate<MB1, 1, 2>(acc, R1, R2, R3, R4, nullptr, lane);
        else if (wave == 2) fused_accumulate<MB1, 2, 2>(acc, R1, R2, R3, R4, nullptr, lane);
        else fused_accumulate<MB1, 3, 2>(acc, R1, R2, R3, R4, nullptr, lane);
        // The next tile's point features: ONE wave loads them, in the slack it has in front of B7 (waves 2 and 3 wait there for dH0), and
        // hands them over through LDS.  (load_point consumes its loads at once -- dtype conversion, augmentation -- so wherever it stands a
        // wave stalls for a global round trip: ~4 k cycles per tile when every wave did it behind the owned channels.)
        if (wave == NW - 1) {
            const f32x16 nx = load_point<T0>(p.cl, s_desc, nb, npidx);
            if (half == 0) {
#pragma unroll
                for (int c = 0; c < 2 * T0; ++c) s_x[c * 32 + l31] = nx[c];
            }
        }
        PCRL_FSTAMP(12);
        __syncthreads();                                                                                  // B7
        PCRL_FSTAMP(13);
        if (wave == 0) fused_accumulate<MB1, 0, 3>(acc, R1, R2, R3, R4, nullptr, lane);
        else if (wave == 1) fused_accumulate<MB1, 1, 3>(acc, R1, R2, R3, R4, nullptr, lane);
        else if (wave == 2) fused_accumulate<MB1, 2, 3>(acc, R1, R2, R3, R4, nullptr, lane);
        else fused_accumulate<MB1, 3, 3>(acc, R1, R2, R3, R4, nullptr, lane);
        PCRL_FSTAMP(14);
        b = nb; tile = ntile; pidx = npidx; own_w = nown; pack = npack;
#pragma unroll
        for (int c = 0; c < 2 * T0; ++c) x[c] = s_x[c * 32 + l31];
    }
    // ---- this workgroup's row ----------------------------------------------------------------------------------------------------------------
    float* row = p.wgrows + (long long)blockIdx.x * FR.total();
    if (wave == 0) fused_flush<MB1, 0>(acc, row, FR, GL, p.cl.C, lane);
    else if (wave == 1) fused_flush<MB1, 1>(acc, row, FR, GL, p.cl.C, lane);
    else if (wave == 2) fused_flush<MB1, 2>(acc, row, FR, GL, p.cl.C, lane);
    else fused_flush<MB1, 3>(acc, row, FR, GL, p.cl.C, lane);
    row[FR.v() + tid] = vu_acc;                                    // threads 0..127: v, 128..255: u
    __syncthreads();
    for (int i = tid; i < kC3; i += 64 * NW) { row[FR.g2() + i] = s_g2[i]; row[FR.be2() + i] = s_be2[i]; }
    allreduce_add32_x16(n1g);
    allreduce_add32_x16(n1b);
    if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = acc_chan(wave * 16 + r, 0) + 4 * half;
            row[GL.g1() + ch] = n1g[r];
            row[GL.be1() + ch] = n1b[r];
        }
    }
}

// ---- reduce: the workgroup rows in row order, the sparse rows of dW2 in cloud order -------------------------------------------------------
template <int kC2>
__global__ __launch_bounds__(1024) void encoder_bwdg_reduce_fused_kernel(const float* __restrict__ rows, int grid_rows, const int* __restrict__ n_items,
                                                                          const FusedRow FR, const GradLayout GL, const float* __restrict__ srows,
                                                                          const float* __restrict__ chc, int B, float* __restrict__ grads,
                                                                          float* __restrict__ extra, const ColsumParams cs, int row_blocks, int main_blocks) {
    if ((int)blockIdx.x >= main_blocks) {      // column-sum jobs riding on this launch (pcrl_encoder_bwd_attach_colsum)
        if (threadIdx.x < 256) colsum_block(cs, (int)blockIdx.x - main_blocks, (int)threadIdx.x);
        return;
    }
    constexpr GramExtra GX{kC2};
    __shared__ float s_part[16][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    if ((int)blockIdx.x < row_blocks) {
        const int i = blockIdx.x * 64 + c;
        const int R = min(grid_rows, *n_items), stride = FR.total();
        const bool on = i < FR.u() + FR.C2;
        if (on) {
            int r = g;
            for (; r + 48 < R; r += 64) {
                p0 = p0 + rows[(long long)(r + 0) * stride + i]; p1 = p1 + rows[(long long)(r + 16) * stride + i];
                p2 = p2 + rows[(long long)(r + 32) * stride + i]; p3 = p3 + rows[(long long)(r + 48) * stride + i];
            }
            for (; r < R; r += 16) p0 = p0 + rows[(long long)r * stride + i];
        }
        s_part[g][c] = (p0 + p1) + (p2 + p3);
        __syncthreads();
        if (g == 0 && on) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = acc + s_part[k][c];
            if (i < FR.main()) grads[i] = acc;
            else if (i < FR.G()) grads[GL.g2() + (i - FR.g2())] = acc;
            else if (i < FR.v()) {
                const int e = i - FR.G(), blk = e >> 10, rr = (e >> 5) & 31, cc = e & 31;
                int bi = 0, rem = blk;                  // packed upper block -> (bi, bj)
                constexpr int MB2 = kC2 / 32;
                while (rem >= MB2 - bi) { rem -= MB2 - bi; ++bi; }
                const int bj = bi + rem;
                extra[GX.G() + (32 * bi + rr) * kC2 + 32 * bj + cc] = acc;
                if (bi != bj) extra[GX.G() + (32 * bj + cc) * kC2 + 32 * bi + rr] = acc;
            } else extra[GX.v() + (i - FR.v())] = acc;
        }
        return;
    }
    {
        const int e = ((int)blockIdx.x - row_blocks) * 64 + c;          // element of S = [C3][C2]; 64 | C2: one channel per block
        const int ch = e / kC2;
        // (the row of a dead channel holds whatever an earlier launch left there: it is loaded WITH the flag, not behind it, and dropped)
        int b = g;
        for (; b + 48 < B; b += 64) {
            const float c0 = chc[(long long)(b + 0) * FR.C3 + ch], c1 = chc[(long long)(b + 16) * FR.C3 + ch];
            const float c2 = chc[(long long)(b + 32) * FR.C3 + ch], c3 = chc[(long long)(b + 48) * FR.C3 + ch];
            const float r0 = srows[(long long)(b + 0) * FR.C3 * kC2 + e], r1 = srows[(long long)(b + 16) * FR.C3 * kC2 + e];
            const float r2 = srows[(long long)(b + 32) * FR.C3 * kC2 + e], r3 = srows[(long long)(b + 48) * FR.C3 * kC2 + e];
            p0 = p0 + (c0 != 0.0f ? r0 : 0.0f); p1 = p1 + (c1 != 0.0f ? r1 : 0.0f); p2 = p2 + (c2 != 0.0f ? r2 : 0.0f); p3 = p3 + (c3 != 0.0f ? r3 : 0.0f);
        }
        for (; b < B; b += 16) {
            const float c0 = chc[(long long)b * FR.C3 + ch], r0 = srows[(long long)b * FR.C3 * kC2 + e];
            p0 = p0 + (c0 != 0.0f ? r0 : 0.0f);
        }
        s_part[g][c] = (p0 + p1) + (p2 + p3);
        __syncthreads();
        if (g == 0) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = acc + s_part[k][c];
            grads[GL.w2() + e] = acc;
        }
    }
}

