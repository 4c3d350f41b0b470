// Round-5 tile paths of gemm_f32_kernel (included by dense.hip inside namespace pcrl, after GemmParams).
//
// Measured on MI355X (tools/probes/cu_load_rate.hip, tools/probes/gemm_staged.hip): the 32 x 32 split-K tiles of cfg 0 read a k-contiguous
// operand with lane = row, i.e. every wave instruction touches 32 rows 4 KB apart and takes 32 bytes of each -- ONE workgroup pulls its
// 256 KB of operands at ~50 GB/s that way, against ~480 GB/s when 64 lanes read 1 KB of one row.  The k loop of a 1 024 x 1 024 layer cost
// 7 us whatever M was (32 ... 256 rows), 5 of them that access pattern.  Here:
//
//   gemm_wtile<E, BM, BN, KW, BRC>   forward (A, B k-contiguous) and data gradient (A k-contiguous, B contiguous along n: BRC).  K of every
//       chunk is split over the 8 waves as before, but each wave fetches ITS k-slice of the tile's rows in row-coalesced 16-byte pieces,
//       stages it in its OWN LDS region (pitch KW + pad: conflict-free both ways) and reads it back in MFMA operand order.  Nothing is
//       shared between waves, so the main loop has no workgroup barrier: the two to four waves of a SIMD drift apart and cover each
//       other's fetch / LDS phases.  A row-contiguous B needs no LDS: its operand of one MFMA is one coalesced 4-byte load per lane.
//       E = 32: v_mfma_f32_32x32x2_f32 blocks; E = 16: v_mfma_f32_16x16x4_f32 blocks, so that a launch with few rows still has a
//       workgroup for every CU (M = 32, one head: 128 tiles of 16 x 16 instead of 32 of 32 x 32: 9.8 -> 3.9 us).
//   gemm_wgrad_panel<BM, BN, KC>     weight gradient (both operands contiguous along their row index, K = batch): a (32 BM) x (32 BN)
//       tile whose two panels are staged in LDS as [k][rows] and shared by all waves (32 instead of 8 FLOP per byte of L2 traffic); a
//       wave owns one 32 x 32 block with 1 / KS of every chunk's k (KS = 8 / blocks; KS = 1: no split-K reduce at all).  The column of
//       ones (bias gradient) is the running sum of the A operand values the block column 0 waves read anyway.
#pragma once

namespace pcrl {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---- one output value through the epilogue (bias, ReLU, mask, accumulate, the ones column's separate destination) ------------------
struct GemmEpi {
    __amdgpu_buffer_rsrc_t rs_c, rs_m;
    const float* bias; float* c_ones;
    unsigned ldc4, ld_mask4;
    int M, N, relu, accumulate, ones_col;
    bool has_mask;
};
__device__ __forceinline__ GemmEpi gemm_epi_setup(const GemmParams& p, int bz) {
    GemmEpi e;
    e.rs_c = make_rsrc(uniform_ptr(p.C + bz * p.c_bs), kGemmRecords);
    e.rs_m = make_rsrc(uniform_ptr((p.mask ? p.mask : p.C) + bz * p.mask_bs), kGemmRecords);
    e.bias = p.bias ? p.bias + bz * p.bias_bs : nullptr;
    e.c_ones = p.C_ones ? p.C_ones + bz * p.c_ones_bs : nullptr;
    e.ldc4 = p.ldc4; e.ld_mask4 = p.ld_mask4; e.M = p.M; e.N = p.N; e.relu = p.relu; e.accumulate = p.accumulate; e.ones_col = p.ones_col;
    e.has_mask = p.mask != nullptr;
    return e;
}
// what the epilogue of (row, col) reads besides the accumulator: issued before the split-K barrier
struct GemmEpiIn { unsigned c_off; float bv, mv, old; };
__device__ __forceinline__ GemmEpiIn gemm_epi_fetch(const GemmEpi& e, int row, int col) {
    GemmEpiIn x;
    const bool ok = row < e.M && col < e.N;
    x.c_off = ok ? (unsigned)row * e.ldc4 + 4u * (unsigned)col : kGemmOob;
    x.bv = e.bias ? e.bias[min(col, e.N - 1)] : 0.0f;
    x.mv = e.has_mask ? buf_load_f1(e.rs_m, ok ? (unsigned)row * e.ld_mask4 + 4u * (unsigned)col : kGemmOob, 0) : 1.0f;
    x.old = e.accumulate ? buf_load_f1(e.rs_c, x.c_off, 0) : 0.0f;
    return x;
}
__device__ __forceinline__ void gemm_epi_store(const GemmEpi& e, const GemmEpiIn& x, int row, int col, float v) {
    v = v + x.bv;
    if (e.relu) v = v > 0.0f ? v : 0.0f;
    v = x.mv > 0.0f ? v : 0.0f;
    v = x.old + v;
    if (e.c_ones && col == e.ones_col) {
        if (x.c_off != kGemmOob) e.c_ones[row] = v;
    } else {
        buf_store_f1(e.rs_c, x.c_off, 0, v);
    }
}

// tile decode shared by the paths below: workgroup -> (batch element, m tile, n tile), n fastest (workgroups b, b + 8, ... share an XCD and
// an n tile whenever the n tile count is a multiple of 8: a slice of the weights lives in one L2)
__device__ __forceinline__ void gemm_tile_decode(const GemmParams& p, int wg, int& bz, int& mt, int& nt) {
    const int local = wg - p.wg_begin;
    bz = __builtin_amdgcn_readfirstlane((int)(((float)local + 0.5f) * p.inv_wg_nm));
    const int rem = local - bz * p.wg_nm;
    mt = __builtin_amdgcn_readfirstlane((int)(((float)rem + 0.5f) * p.inv_wg_n));
    nt = rem - mt * p.wg_n;
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int E, int BM, int BN, int KW, bool BRC>
struct WTile {
    static constexpr int RA = E * BM, RB = BRC ? 0 : E * BN, R = RA + RB;
    static constexpr int PAD = E == 32 ? 4 : 8;                  // conflict-free ds_read_b128 by (row, k group) lanes: see tools/probes/gemm_staged.hip
    static constexpr int PITCH = KW + PAD, STAGE = R * PITCH, WAVE_LDS = 2 * STAGE;
    static constexpr int PPR = KW / 4, NP = (R * PPR + 63) / 64, NACC = BM * BN, REGS = E == 32 ? 16 : 4;
    static constexpr int KQ = E == 32 ? 8 : 16, NQ = KW / KQ, NBV = BRC ? BN * NQ * 4 : 1, KC = 8 * KW;
    static constexpr int VAL = NACC * REGS * 64;
    static constexpr size_t lds_bytes() {
        const size_t stage = sizeof(float) * 8 * WAVE_LDS, red = sizeof(float) * 8 * VAL;
        return stage > red ? stage : red;
    }
    static_assert(64 % PPR == 0 && (R * PPR) % 64 == 0 && KW % KQ == 0 && (RA * PPR) % 64 == 0, "shape");
};

template <int E, int BM, int BN, int KW, bool BRC, int RING>
__device__ __forceinline__ void gemm_wtile(const GemmParams& p, const int wg, float* smem) {
    using S = WTile<E, BM, BN, KW, BRC>;
    constexpr int RA = S::RA, PITCH = S::PITCH, STAGE = S::STAGE, PPR = S::PPR, NP = S::NP, NPA = RA * PPR / 64, NACC = S::NACC, REGS = S::REGS;
    constexpr int KQ = S::KQ, NQ = S::NQ, NBV = S::NBV, KC = S::KC;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & (E - 1), g = lane / E;                  // operand row inside a block; k group (2 resp. 4 per wave)
    int bz, mt, nt;
    gemm_tile_decode(p, wg, bz, mt, nt);
    const int m0 = E * BM * mt, n0 = E * BN * nt;
    // The staged operands' resources end with the operand's last row: a tile row past it reads zeros (never stored), so the pieces of one
    // lane are ONE offset + a wave-uniform row step instead of a clamped offset each.
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(uniform_ptr(p.A + bz * p.a_bs), (unsigned)(p.M - 1) * p.a_sm4 + 4u * (unsigned)p.K);
    const __amdgpu_buffer_rsrc_t rs_b = BRC ? make_rsrc(uniform_ptr(p.B + bz * p.b_bs), kGemmRecords)
                                            : make_rsrc(uniform_ptr(p.B + bz * p.b_bs), (unsigned)(p.N - 1) * p.b_sn4 + 4u * (unsigned)p.K);
    float* my = smem + wave * S::WAVE_LDS;
    // staged pieces: piece pc = lane + 64 u of this wave's [R rows][KW] slice -> row pc / PPR, 16-byte piece pc % PPR (the same for every u)
    constexpr int RSTEP = 64 / PPR;                              // rows between a lane's consecutive pieces
    const int kp = lane % PPR, prow = lane / PPR;
    const int kpos = KW * wave + 4 * kp;                         // first k of this lane's pieces inside a chunk
    const unsigned voff_a = (unsigned)(m0 + prow) * p.a_sm4 + 4u * (unsigned)kpos;
    const unsigned voff_b = (unsigned)(n0 + prow) * p.b_sn4 + 4u * (unsigned)kpos;
    const unsigned step_a = (unsigned)RSTEP * p.a_sm4, step_b = (unsigned)RSTEP * p.b_sn4;
    float* lst = my + prow * PITCH + 4 * kp;                     // piece u goes to lst + u RSTEP PITCH
    // a row-contiguous B: element (k, n) at k b_sk4 + 4 n; this lane's column of every block column, k of its lane group
    unsigned boff[BN];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) boff[bn] = (unsigned)(KW * wave + 4 * g) * p.b_sk4 + 4u * (unsigned)min(n0 + E * bn + i, p.N - 1);
    typedef float accv __attribute__((ext_vector_type(REGS)));
    accv acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < REGS; ++r) acc[a][r] = 0.0f;
    const int n_chunks = (p.K + KC - 1) / KC;
    f32x4 ring[RING][NP];
    float bring[2][NBV];
    // Refills are unconditional: a chunk (or a piece of the ragged last chunk) past K reads an out-of-range offset -- zeros, no memory
    // request.  A conditional refill makes the compiler's vmcnt bookkeeping assume the shorter queue and drain the ring before every LDS store.
    auto gload_staged = [&](int c, f32x4 (&v)[NP]) {
        const unsigned oobv = c * KC + kpos < p.K ? 0u : kGemmOob;          // K % 4 == 0 on this path: a piece lies inside K or outside
        const unsigned soff = (unsigned)c * (KC * 4u);
        const unsigned va = voff_a | oobv, vb = voff_b | oobv;
#pragma unroll
        for (int u = 0; u < NP; ++u)
            v[u] = u < NPA ? buf_load_f4(rs_a, va + (unsigned)u * step_a, soff) : buf_load_f4(rs_b, vb + (unsigned)(u - NPA) * step_b, soff);
    };
    auto gload_brc = [&](int c, float (&bv)[NBV]) {
        if constexpr (BRC) {
            const int kleft = p.K - (c * KC + KW * wave + 4 * g);           // rows of B this lane group may read: k_rel < kleft
            const unsigned srow = (unsigned)c * (unsigned)KC * p.b_sk4;
            if (__builtin_amdgcn_readfirstlane(c * KC + KC <= p.K)) {         // (wave-uniform: the chunk lies inside K)
#pragma unroll
                for (int bn = 0; bn < BN; ++bn)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j) bv[(bn * NQ + q) * 4 + j] = buf_load_f1(rs_b, boff[bn], srow + (unsigned)(KQ * q + j) * p.b_sk4);
            } else {
#pragma unroll
                for (int bn = 0; bn < BN; ++bn)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            bv[(bn * NQ + q) * 4 + j] = buf_load_f1(rs_b, KQ * q + j < kleft ? boff[bn] : kGemmOob, srow + (unsigned)(KQ * q + j) * p.b_sk4);
            }
        }
    };
    auto lstore = [&](int st, const f32x4 (&v)[NP]) {
#pragma unroll
        for (int u = 0; u < NP; ++u) *reinterpret_cast<f32x4*>(lst + st * STAGE + u * (RSTEP * PITCH)) = v[u];
        wave_lds_fence();
    };
    auto compute = [&](int st, const float (&bv)[NBV]) {
        const float* s = my + st * STAGE + 4 * g;
        f32x4 a[BM][NQ], b[BRC ? 1 : BN][NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int bm = 0; bm < BM; ++bm) a[bm][q] = *reinterpret_cast<const f32x4*>(s + (E * bm + i) * PITCH + KQ * q);
            if constexpr (!BRC) {
#pragma unroll
                for (int bn = 0; bn < BN; ++bn) b[bn][q] = *reinterpret_cast<const f32x4*>(s + (RA + E * bn + i) * PITCH + KQ * q);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int bm = 0; bm < BM; ++bm)
#pragma unroll
                    for (int bn = 0; bn < BN; ++bn) {
                        float bval;
                        if constexpr (BRC) bval = bv[(bn * NQ + q) * 4 + j]; else bval = b[bn][q][j];
                        if constexpr (E == 32)
                            acc[bm * BN + bn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bm][q][j], bval, acc[bm * BN + bn], 0, 0, 0);
                        else
                            acc[bm * BN + bn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[bm][q][j], bval, acc[bm * BN + bn], 0, 0, 0);
                    }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // chunk c lives in LDS stage c & 1; the next RING chunks wait in the register ring
    constexpr int S1 = RING > 1 ? 1 : 0;
    gload_staged(0, ring[0]);
    if (RING > 1) gload_staged(1, ring[S1]);
    gload_brc(0, bring[0]); gload_brc(1, bring[1]);
    lstore(0, ring[0]);
    for (int c = 0; c < n_chunks; c += 2) {
        gload_staged(c + RING, ring[0]);         // chunk c is in LDS: its register slot is free
        __builtin_amdgcn_sched_barrier(0);
        compute(0, bring[0]);
        lstore(1, ring[S1]);
        gload_brc(c + 2, bring[0]);              // ... and now its B operands' slot
        if (c + 1 >= n_chunks) break;
        gload_staged(c + 1 + RING, ring[S1]);
        __builtin_amdgcn_sched_barrier(0);
        compute(1, bring[1]);
        lstore(0, ring[0]);
        gload_brc(c + 3, bring[1]);
    }
    // ---- the eight k slices meet in LDS, fixed order ---------------------------------------------------------------------------------
    const GemmEpi epi = gemm_epi_setup(p, bz);
    constexpr int VAL = S::VAL, NE = (VAL + 511) / 512;
    GemmEpiIn ein[NE];
    int row[NE], col[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int v = tid + 512 * e, a = v / (REGS * 64), r = (v / 64) % REGS, ln = v & 63, bm = a / BN, bn = a % BN;
        if constexpr (E == 32) { row[e] = m0 + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5); col[e] = n0 + 32 * bn + (ln & 31); }
        else { row[e] = m0 + 16 * bm + 4 * (ln >> 4) + r; col[e] = n0 + 16 * bn + (ln & 15); }
        if (v >= VAL) row[e] = p.M;              // (only the 16 x 16 single-block tile: half of the threads have nothing to finish)
        ein[e] = gemm_epi_fetch(epi, row[e], col[e]);
    }
    __syncthreads();                             // every wave is done with its staging region
    float (*red)[NACC][REGS][64] = reinterpret_cast<float (*)[NACC][REGS][64]>(smem);
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < REGS; ++r) red[wave][a][r][lane] = acc[a][r];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int v = tid + 512 * e, a = (v / (REGS * 64)) % NACC, r = (v / 64) % REGS, ln = v & 63;
        const float x = ((red[0][a][r][ln] + red[1][a][r][ln]) + (red[2][a][r][ln] + red[3][a][r][ln])) +
                        ((red[4][a][r][ln] + red[5][a][r][ln]) + (red[6][a][r][ln] + red[7][a][r][ln]));
        gemm_epi_store(epi, ein[e], row[e], col[e], x);
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int KC>
struct WPanel {
    static constexpr int NB = BM * BN, KS = 8 / NB, TM = 32 * BM, TN = 32 * BN, PA = TM + 4, PB = TN + 4, STAGE = KC * (PA + PB);
    static constexpr int NPA = KC * (TM / 4) / 512, NPB = KC * (TN / 4) / 512, KWV = KC / KS;
    static constexpr size_t lds_bytes() {
        const size_t stage = sizeof(float) * 2 * STAGE, red = sizeof(float) * 8 * 17 * 64;
        return stage > red ? stage : red;
    }
    static_assert(NB <= 8 && 8 % NB == 0 && (KC * (TM / 4)) % 512 == 0 && (KC * (TN / 4)) % 512 == 0 && KWV % 2 == 0, "shape");
};

template <int BM, int BN, int KC>
__device__ __forceinline__ void gemm_wgrad_panel(const GemmParams& p, const int wg, float* smem) {
    using S = WPanel<BM, BN, KC>;
    constexpr int NB = S::NB, KS = S::KS, TM = S::TM, TN = S::TN, PA = S::PA, PB = S::PB, STAGE = S::STAGE, NPA = S::NPA, NPB = S::NPB, KWV = S::KWV;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, g = lane >> 5;
    const int blk = wave % NB, ks = wave / NB, bm = blk / BN, bn = blk % BN;
    int bz, mt, nt;
    gemm_tile_decode(p, wg, bz, mt, nt);
    const int m0 = TM * mt, n0 = TN * nt;
    const int n_real = p.ones_col >= 0 ? p.ones_col : p.N;       // the column of ones is never read from memory
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(uniform_ptr(p.A + bz * p.a_bs), kGemmRecords);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(uniform_ptr(p.B + bz * p.b_bs), kGemmRecords);
    // panel pieces: 16 bytes = four consecutive rows of one k; rows past the operand's end read zeros (M % 4 == 0 and n_real % 4 == 0 here)
    unsigned va[NPA], vb[NPB];
    int la[NPA], lb[NPB], ka[NPA], kb[NPB];
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
        const int pc = tid + 512 * u, k = pc / (TM / 4), r4 = pc % (TM / 4);
        va[u] = m0 + 4 * r4 < p.M ? (unsigned)k * p.a_sk4 + 4u * (unsigned)(m0 + 4 * r4) : kGemmOob;
        la[u] = k * PA + 4 * r4; ka[u] = k;
    }
#pragma unroll
    for (int u = 0; u < NPB; ++u) {
        const int pc = tid + 512 * u, k = pc / (TN / 4), r4 = pc % (TN / 4);
        vb[u] = n0 + 4 * r4 < n_real ? (unsigned)k * p.b_sk4 + 4u * (unsigned)(n0 + 4 * r4) : kGemmOob;
        lb[u] = KC * PA + k * PB + 4 * r4; kb[u] = k;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float ones_sum = 0.0f;                       // sum over this wave's k of A[k][m0 + 32 bm + i]: the bias gradient (block column 0 of n tile 0)
    const bool ones_wave = p.ones_col >= 0 && nt == 0 && bn == 0;
    const int n_chunks = (p.K + KC - 1) / KC;
    f32x4 rga[NPA], rgb[NPB];
    auto gload = [&](int c) {
        const unsigned sa = (unsigned)c * (unsigned)KC * p.a_sk4, sb = (unsigned)c * (unsigned)KC * p.b_sk4;
        if (__builtin_amdgcn_readfirstlane(c * KC + KC <= p.K)) {
#pragma unroll
            for (int u = 0; u < NPA; ++u) rga[u] = buf_load_f4(rs_a, va[u], sa);
#pragma unroll
            for (int u = 0; u < NPB; ++u) rgb[u] = buf_load_f4(rs_b, vb[u], sb);
        } else {                                  // the ragged last chunk, and the refill past the end: k >= K reads zeros
#pragma unroll
            for (int u = 0; u < NPA; ++u) rga[u] = buf_load_f4(rs_a, c * KC + ka[u] < p.K ? va[u] : kGemmOob, sa);
#pragma unroll
            for (int u = 0; u < NPB; ++u) rgb[u] = buf_load_f4(rs_b, c * KC + kb[u] < p.K ? vb[u] : kGemmOob, sb);
        }
    };
    auto lstore = [&](int st) {
#pragma unroll
        for (int u = 0; u < NPA; ++u) *reinterpret_cast<f32x4*>(smem + st * STAGE + la[u]) = rga[u];
#pragma unroll
        for (int u = 0; u < NPB; ++u) *reinterpret_cast<f32x4*>(smem + st * STAGE + lb[u]) = rgb[u];
    };
    auto compute = [&](int st) {
        const float* sa = smem + st * STAGE + (ks * KWV + g) * PA + 32 * bm + i;
        const float* sb = smem + st * STAGE + KC * PA + (ks * KWV + g) * PB + 32 * bn + i;
        float a[KWV / 2], b[KWV / 2];
#pragma unroll
        for (int s2 = 0; s2 < KWV / 2; ++s2) { a[s2] = sa[2 * s2 * PA]; b[s2] = sb[2 * s2 * PB]; }
#pragma unroll
        for (int s2 = 0; s2 < KWV / 2; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], b[s2], acc, 0, 0, 0);
        if (ones_wave) {
#pragma unroll
            for (int s2 = 0; s2 < KWV / 2; ++s2) ones_sum += a[s2];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        gload(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(c & 1);
        lstore((c & 1) ^ 1);
        __syncthreads();
    }
    const GemmEpi epi = gemm_epi_setup(p, bz);
    const int col = n0 + 32 * bn + i;
    const bool col_real = col < n_real;
    if constexpr (KS == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * g;
            const GemmEpiIn x = gemm_epi_fetch(epi, row, col_real ? col : p.N);
            gemm_epi_store(epi, x, row, col, acc[r]);
        }
        if (ones_wave) {                          // both lane halves hold one parity of k: their sum is the column of ones for row m0 + 32 bm + i
            float lo, hi;
            both_halves(ones_sum, lo, hi);
            const int row = m0 + 32 * bm + i;
            if (g == 0) { const GemmEpiIn x = gemm_epi_fetch(epi, row, p.ones_col); gemm_epi_store(epi, x, row, p.ones_col, lo + hi); }
        }
    } else {
        float (*red)[17][64] = reinterpret_cast<float (*)[17][64]>(smem);      // [wave][register | ones sum][lane]
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
        red[wave][16][lane] = ones_sum;
        __syncthreads();
        // the KS waves of a block share its 16 registers; partial sums are added in k-slice order
#pragma unroll
        for (int e = 0; e < 16 / KS; ++e) {
            const int r = ks * (16 / KS) + e;
            float v = red[blk][r][lane];
#pragma unroll
            for (int q = 1; q < KS; ++q) v = v + red[blk + NB * q][r][lane];
            const int row = m0 + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * g;
            const GemmEpiIn x = gemm_epi_fetch(epi, row, col_real ? col : p.N);
            gemm_epi_store(epi, x, row, col, v);
        }
        if (ones_wave && ks == 0 && g == 0) {
            float v = red[blk][16][i] + red[blk][16][32 + i];
#pragma unroll
            for (int q = 1; q < KS; ++q) v = v + (red[blk + NB * q][16][i] + red[blk + NB * q][16][32 + i]);
            const int row = m0 + 32 * bm + i;
            const GemmEpiIn x = gemm_epi_fetch(epi, row, p.ones_col);
            gemm_epi_store(epi, x, row, p.ones_col, v);
        }
    }
}

}  // namespace pcrl
