// Probe (round 5): the heads' 1 024 x 1 024 forward layer, C = relu(A W^T + b), as
//   direct      : today's cfg 0 -- a 32 x 32 tile per 8-wave workgroup, K split over the waves, operands straight from L2 (rows 4 KB apart per lane)
//   staged<BM,BN>: (32 BM) x (32 BN) tile, operands fetched ROW-COALESCED (64 lanes = 1 KB of one row), staged in LDS, K of every chunk split
//                  over the 8 waves, split-K reduce through LDS
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off gemm_staged.hip -o gemm_staged && ./gemm_staged
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* ptr) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

struct P { const float* A; const float* W; const float* bias; float* C; int M, N, K, mtiles, ntiles; };

__global__ __launch_bounds__(512, 4) void direct(const P p) {
    __shared__ float s_red[8][16][64];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int tph = p.mtiles * p.ntiles, head = blockIdx.x / tph, rem = blockIdx.x % tph, mt = rem / p.ntiles, nt = rem % p.ntiles;
    const int m0 = 32 * mt, n0 = 32 * nt;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(uniform_ptr(p.A + (size_t)head * p.M * p.K)), rb = make_rsrc(uniform_ptr(p.W + (size_t)head * p.N * p.K));
    const unsigned a_off = (unsigned)min(m0 + i, p.M - 1) * p.K * 4u + 64u * h, b_off = (unsigned)min(n0 + i, p.N - 1) * p.K * 4u + 64u * h;
    const int n_sc = p.K >> 5, per = n_sc / 8, sc0 = wave * per, sc1 = sc0 + per;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    f32x4 a0[4], b0[4], a1[4], b1[4];
    auto load = [&](int sc, f32x4 (&a)[4], f32x4 (&b)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { a[t] = ld4(ra, a_off + 16 * t, sc * 128); b[t] = ld4(rb, b_off + 16 * t, sc * 128); }
    };
    auto mm = [&](const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[t][j], acc, 0, 0, 0);
    };
    int sc = sc0;
    if (sc < sc1) load(sc, a0, b0);
    if (sc + 1 < sc1) load(sc + 1, a1, b1);
    while (sc < sc1) {
        mm(a0, b0);
        if (sc + 2 < sc1) load(sc + 2, a0, b0);
        if (++sc >= sc1) break;
        mm(a1, b1);
        if (sc + 2 < sc1) load(sc + 2, a1, b1);
        ++sc;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[r];
    const int col = n0 + i;
    const float bv = p.bias[(size_t)head * p.N + min(col, p.N - 1)];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = wave + 8 * e, row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = ((s_red[0][r][lane] + s_red[1][r][lane]) + (s_red[2][r][lane] + s_red[3][r][lane])) +
                  ((s_red[4][r][lane] + s_red[5][r][lane]) + (s_red[6][r][lane] + s_red[7][r][lane]));
        v = v + bv;
        v = v > 0.0f ? v : 0.0f;
        if (row < p.M && col < p.N) p.C[((size_t)head * p.M + row) * p.N + col] = v;
    }
}

template <int BM, int BN, int RING>
__global__ __launch_bounds__(512) void staged(const P p) {
    constexpr int KW = 32 / (BM * BN), KC = 8 * KW, R = 32 * (BM + BN), PITCH = KC + 4, STAGE = R * PITCH;
    constexpr int PPR = KC / 4, NP = R * PPR / 512, NPA = 32 * BM * PPR / 512, NQ = KW / 8, NACC = BM * BN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int tph = p.mtiles * p.ntiles, head = blockIdx.x / tph, rem = blockIdx.x % tph, mt = rem / p.ntiles, nt = rem % p.ntiles;
    const int m0 = 32 * BM * mt, n0 = 32 * BN * nt;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(uniform_ptr(p.A + (size_t)head * p.M * p.K)), rb = make_rsrc(uniform_ptr(p.W + (size_t)head * p.N * p.K));
    unsigned voff[NP];
    int lofs[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int pc = tid + 512 * u, row = pc / PPR, kp = pc % PPR;
        const int grow = u < NPA ? min(m0 + row, p.M - 1) : min(n0 + row - 32 * BM, p.N - 1);
        voff[u] = (unsigned)grow * p.K * 4u + 16u * kp;
        lofs[u] = row * PITCH + 4 * kp;
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    const int n_chunks = p.K / KC;
    f32x4 ring[RING][NP];
    auto gload = [&](int c, f32x4 (&v)[NP]) {
#pragma unroll
        for (int u = 0; u < NP; ++u) v[u] = ld4(u < NPA ? ra : rb, voff[u], (unsigned)c * KC * 4u);
    };
    auto lstore = [&](int st, const f32x4 (&v)[NP]) {
#pragma unroll
        for (int u = 0; u < NP; ++u) *reinterpret_cast<f32x4*>(smem + st * STAGE + lofs[u]) = v[u];
    };
    auto compute = [&](int st) {
        const float* s = smem + st * STAGE + KW * wave + 4 * h;
        f32x4 a[BM][NQ], b[BN][NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int bm = 0; bm < BM; ++bm) a[bm][q] = *reinterpret_cast<const f32x4*>(s + (32 * bm + i) * PITCH + 8 * q);
#pragma unroll
            for (int bn = 0; bn < BN; ++bn) b[bn][q] = *reinterpret_cast<const f32x4*>(s + (32 * BM + 32 * bn + i) * PITCH + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int bm = 0; bm < BM; ++bm)
#pragma unroll
                    for (int bn = 0; bn < BN; ++bn)
                        acc[bm * BN + bn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bm][q][j], b[bn][q][j], acc[bm * BN + bn], 0, 0, 0);
    };
    // ring of RING chunks in registers ahead of the one in LDS
    gload(0, ring[0]);
    if (RING > 1 && n_chunks > 1) gload(1, ring[1]);
    lstore(0, ring[0]);
    __syncthreads();
    auto body = [&](int c, f32x4 (&ld)[NP], const f32x4 (&stv)[NP]) {
        const int st = c & 1;
        if (c + RING < n_chunks) gload(c + RING, ld);       // the slot of chunk c, which is in LDS already
        compute(st);
        if (c + 1 < n_chunks) lstore(st ^ 1, stv);
        __syncthreads();
    };
    for (int c = 0; c < n_chunks; c += 2) {
        body(c, ring[0], ring[RING > 1 ? 1 : 0]);
        if (c + 1 >= n_chunks) break;
        body(c + 1, ring[RING > 1 ? 1 : 0], ring[0]);
    }
    // split-K reduce through LDS, NR accumulators per round
    constexpr int LDS_FLOATS = 2 * STAGE;
    constexpr int NR = (LDS_FLOATS / (8 * 16 * 64)) >= NACC ? NACC : (LDS_FLOATS / (8 * 16 * 64));
    static_assert(NR >= 1 && NACC % NR == 0, "reduce rounds");
    float (*red)[NR][16][64] = reinterpret_cast<float (*)[NR][16][64]>(smem);
    const float* bias = p.bias + (size_t)head * p.N;
    float* C = p.C + (size_t)head * p.M * p.N;
#pragma unroll
    for (int round = 0; round < NACC / NR; ++round) {
        if (round) __syncthreads();
#pragma unroll
        for (int a = 0; a < NR; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave][a][r][lane] = acc[round * NR + a][r];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            const int ai = round * NR + a, bm = ai / BN, bn = ai % BN;
            const int col = n0 + 32 * bn + i;
            const float bv = bias[min(col, p.N - 1)];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = wave + 8 * e, row = m0 + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = ((red[0][a][r][lane] + red[1][a][r][lane]) + (red[2][a][r][lane] + red[3][a][r][lane])) +
                          ((red[4][a][r][lane] + red[5][a][r][lane]) + (red[6][a][r][lane] + red[7][a][r][lane]));
                v = v + bv;
                v = v > 0.0f ? v : 0.0f;
                if (row < p.M && col < p.N) C[(size_t)row * p.N + col] = v;
            }
        }
    }
}


// Wave-private staging: K of a chunk is split over the 8 waves anyway, so every wave fetches ITS k-slice of all tile rows (row-coalesced
// pieces of KW * 4 bytes), stages it in its own LDS region and reads it back in MFMA operand order -- no workgroup barrier in the main loop.
// E = 32: v_mfma_f32_32x32x2_f32 blocks, E = 16: v_mfma_f32_16x16x4_f32 blocks; tile (E BM) x (E BN).
template <int E, int BM, int BN, int KW, int RING>
__global__ __launch_bounds__(512) void wstaged(const P p) {
    constexpr int R = E * (BM + BN), PAD = E == 32 ? 4 : 8, PITCH = KW + PAD, STAGE = R * PITCH, WAVE_LDS = 2 * STAGE;
    constexpr int PPR = KW / 4, NP = R * PPR / 64, NACC = BM * BN, REGS = E == 32 ? 16 : 4;
    constexpr int KQ = E == 32 ? 8 : 16, NQ = KW / KQ;               // k per b128 operand read of the whole wave
    static_assert(R * PPR % 64 == 0 && KW % KQ == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & (E - 1), g = lane / E;                       // operand row inside a block, k group (2 resp. 4 of them)
    const int tph = p.mtiles * p.ntiles, head = blockIdx.x / tph, rem = blockIdx.x % tph, mt = rem / p.ntiles, nt = rem % p.ntiles;
    const int m0 = E * BM * mt, n0 = E * BN * nt;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(uniform_ptr(p.A + (size_t)head * p.M * p.K)), rb = make_rsrc(uniform_ptr(p.W + (size_t)head * p.N * p.K));
    float* my = smem + wave * WAVE_LDS;
    unsigned voff[NP];
    int lofs[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int pc = lane + 64 * u, row = pc / PPR, kp = pc % PPR;
        const bool is_a = row < E * BM;
        const int grow = is_a ? min(m0 + row, p.M - 1) : min(n0 + row - E * BM, p.N - 1);
        voff[u] = (unsigned)grow * p.K * 4u + 16u * kp + (unsigned)(KW * wave) * 4u;
        lofs[u] = row * PITCH + 4 * kp;
    }
    typedef float accv __attribute__((ext_vector_type(REGS)));
    accv acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < REGS; ++r) acc[a][r] = 0.0f;
    constexpr int KC = 8 * KW;
    const int n_chunks = p.K / KC;
    f32x4 ring[RING][NP];
    // refills are unconditional (a chunk past the end reads an out-of-range offset: zeros, no memory request): a conditional refill makes
    // the compiler's vmcnt bookkeeping assume the shorter queue and drain the ring in front of every LDS store
    auto gload = [&](int c, f32x4 (&v)[NP]) {
        const unsigned oob = c < n_chunks ? 0u : 0x80000000u;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const bool is_a = (lane + 64 * u) / PPR < E * BM;         // compile-time per u when 64 / PPR divides E BM
            v[u] = is_a ? ld4(ra, voff[u] | oob, (unsigned)c * KC * 4u) : ld4(rb, voff[u] | oob, (unsigned)c * KC * 4u);
        }
    };
    auto lstore = [&](int st, const f32x4 (&v)[NP]) {
#pragma unroll
        for (int u = 0; u < NP; ++u) *reinterpret_cast<f32x4*>(my + st * STAGE + lofs[u]) = v[u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto compute = [&](int st) {
        const float* s = my + st * STAGE + 4 * g;
        f32x4 a[BM][NQ], b[BN][NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int bm = 0; bm < BM; ++bm) a[bm][q] = *reinterpret_cast<const f32x4*>(s + (E * bm + i) * PITCH + KQ * q);
#pragma unroll
            for (int bn = 0; bn < BN; ++bn) b[bn][q] = *reinterpret_cast<const f32x4*>(s + (E * BM + E * bn + i) * PITCH + KQ * q);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int bm = 0; bm < BM; ++bm)
#pragma unroll
                    for (int bn = 0; bn < BN; ++bn) {
                        if constexpr (E == 32)
                            acc[bm * BN + bn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bm][q][j], b[bn][q][j], acc[bm * BN + bn], 0, 0, 0);
                        else
                            acc[bm * BN + bn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[bm][q][j], b[bn][q][j], acc[bm * BN + bn], 0, 0, 0);
                    }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    gload(0, ring[0]);
    if (RING > 1) gload(1, ring[1]);
    lstore(0, ring[0]);
    auto body = [&](int c, f32x4 (&ld)[NP], const f32x4 (&stv)[NP]) {
        const int st = c & 1;
        gload(c + RING, ld);
        __builtin_amdgcn_sched_barrier(0);
        compute(st);
        lstore(st ^ 1, stv);
    };
    for (int c = 0; c < n_chunks; c += 2) {                         // (n_chunks is even in this probe)
        body(c, ring[0], ring[RING > 1 ? 1 : 0]);
        body(c + 1, ring[RING > 1 ? 1 : 0], ring[0]);
    }
    __syncthreads();                                                 // every wave is done with its staging region
    float (*red)[NACC][REGS][64] = reinterpret_cast<float (*)[NACC][REGS][64]>(smem);
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < REGS; ++r) red[wave][a][r][lane] = acc[a][r];
    __syncthreads();
    const float* bias = p.bias + (size_t)head * p.N;
    float* C = p.C + (size_t)head * p.M * p.N;
    constexpr int VAL = NACC * REGS * 64;
#pragma unroll
    for (int e = 0; e < (VAL + 511) / 512; ++e) {
        const int v = tid + 512 * e;
        if (v >= VAL) break;
        const int a = v / (REGS * 64), r = (v / 64) % REGS, ln = v & 63, bm = a / BN, bn = a % BN;
        int row, col;
        if constexpr (E == 32) { row = m0 + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5); col = n0 + 32 * bn + (ln & 31); }
        else { row = m0 + 16 * bm + 4 * (ln >> 4) + r; col = n0 + 16 * bn + (ln & 15); }
        float x = ((red[0][a][r][ln] + red[1][a][r][ln]) + (red[2][a][r][ln] + red[3][a][r][ln])) +
                  ((red[4][a][r][ln] + red[5][a][r][ln]) + (red[6][a][r][ln] + red[7][a][r][ln]));
        x = x + bias[min(col, p.N - 1)];
        x = x > 0.0f ? x : 0.0f;
        if (row < p.M && col < p.N) C[(size_t)row * p.N + col] = x;
    }
}

// Data gradient: C = (A W) (.) [mask > 0], A = dY [M][K] k-contiguous (wave-private staging as above), W [K][N] row-contiguous: the B operand
// of k-step (q, j) is ONE coalesced 4-byte load per lane (lane = column), straight into the MFMA operand register -- no LDS for B.
template <int E, int BM, int BN, int KW>
__global__ __launch_bounds__(512) void wdx(const P p) {
    constexpr int R = E * BM, PAD = E == 32 ? 4 : 8, PITCH = KW + PAD, STAGE = R * PITCH, WAVE_LDS = 2 * STAGE;
    constexpr int PPR = KW / 4, NP = (R * PPR + 63) / 64, NACC = BM * BN, REGS = E == 32 ? 16 : 4;
    constexpr int KQ = E == 32 ? 8 : 16, NQ = KW / KQ, NB = BN * NQ * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & (E - 1), g = lane / E;
    const int tph = p.mtiles * p.ntiles, head = blockIdx.x / tph, rem = blockIdx.x % tph, mt = rem / p.ntiles, nt = rem % p.ntiles;
    const int m0 = E * BM * mt, n0 = E * BN * nt;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(uniform_ptr(p.A + (size_t)head * p.M * p.K)), rb = make_rsrc(uniform_ptr(p.W + (size_t)head * p.N * p.K));
    float* my = smem + wave * WAVE_LDS;
    unsigned voff[NP];
    int lofs[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int pc = lane + 64 * u, row = pc / PPR, kp = pc % PPR;
        voff[u] = row < R ? (unsigned)min(m0 + row, p.M - 1) * p.K * 4u + 16u * kp + (unsigned)(KW * wave) * 4u : 0x80000000u;
        lofs[u] = min(row, R - 1) * PITCH + 4 * kp;
    }
    // B: element (k, n) at (k N + n) * 4; this lane's column(s) and the k offset of its lane group
    unsigned boff[BN];
#pragma unroll
    for (int bn = 0; bn < BN; ++bn) boff[bn] = ((unsigned)(KW * wave + 4 * g) * p.N + (unsigned)min(n0 + E * bn + i, p.N - 1)) * 4u;
    typedef float accv __attribute__((ext_vector_type(REGS)));
    accv acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < REGS; ++r) acc[a][r] = 0.0f;
    constexpr int KC = 8 * KW;
    const int n_chunks = p.K / KC;
    f32x4 ring[2][NP];
    float bring[2][NB];
    auto gload_a = [&](int c, f32x4 (&v)[NP]) {
        const unsigned oob = c < n_chunks ? 0u : 0x80000000u;
#pragma unroll
        for (int u = 0; u < NP; ++u) v[u] = ld4(ra, voff[u] | oob, (unsigned)c * KC * 4u);
    };
    auto gload_b = [&](int c, float (&bv)[NB]) {
        const unsigned oob = c < n_chunks ? 0u : 0x80000000u;
        const unsigned brow = (unsigned)c * KC * p.N * 4u;
#pragma unroll
        for (int bn = 0; bn < BN; ++bn)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bv[(bn * NQ + q) * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (boff[bn] + (unsigned)(KQ * q + j) * p.N * 4u) | oob, brow, 0));
    };
    auto lstore = [&](int st, const f32x4 (&v)[NP]) {
#pragma unroll
        for (int u = 0; u < NP; ++u)
            if (R * PPR % 64 == 0 || lane + 64 * u < R * PPR) *reinterpret_cast<f32x4*>(my + st * STAGE + lofs[u]) = v[u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto compute = [&](int st, const float (&bv)[NB]) {
        const float* s = my + st * STAGE + 4 * g;
        f32x4 a[BM][NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int bm = 0; bm < BM; ++bm) a[bm][q] = *reinterpret_cast<const f32x4*>(s + (E * bm + i) * PITCH + KQ * q);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int bm = 0; bm < BM; ++bm)
#pragma unroll
                    for (int bn = 0; bn < BN; ++bn) {
                        if constexpr (E == 32)
                            acc[bm * BN + bn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bm][q][j], bv[(bn * NQ + q) * 4 + j], acc[bm * BN + bn], 0, 0, 0);
                        else
                            acc[bm * BN + bn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[bm][q][j], bv[(bn * NQ + q) * 4 + j], acc[bm * BN + bn], 0, 0, 0);
                    }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    gload_a(0, ring[0]); gload_a(1, ring[1]); gload_b(0, bring[0]); gload_b(1, bring[1]);
    lstore(0, ring[0]);
    for (int c = 0; c < n_chunks; c += 2) {
        gload_a(c + 2, ring[0]);                 // chunk c is in LDS: its register slot is free
        __builtin_amdgcn_sched_barrier(0);
        compute(0, bring[0]);
        lstore(1, ring[1]);
        gload_b(c + 2, bring[0]);                // ... and so is its B slot, now
        gload_a(c + 3, ring[1]);
        __builtin_amdgcn_sched_barrier(0);
        compute(1, bring[1]);
        lstore(0, ring[0]);
        gload_b(c + 3, bring[1]);
    }
    __syncthreads();
    float (*red)[NACC][REGS][64] = reinterpret_cast<float (*)[NACC][REGS][64]>(smem);
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < REGS; ++r) red[wave][a][r][lane] = acc[a][r];
    __syncthreads();
    const float* mask = p.bias + (size_t)head * p.M * p.N;             // (the probe passes the mask through the bias pointer)
    float* C = p.C + (size_t)head * p.M * p.N;
    constexpr int VAL = NACC * REGS * 64;
#pragma unroll
    for (int e = 0; e < (VAL + 511) / 512; ++e) {
        const int v = tid + 512 * e;
        if (v >= VAL) break;
        const int a = v / (REGS * 64), r = (v / 64) % REGS, ln = v & 63, bm = a / BN, bn = a % BN;
        int row, col;
        if constexpr (E == 32) { row = m0 + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5); col = n0 + 32 * bn + (ln & 31); }
        else { row = m0 + 16 * bm + 4 * (ln >> 4) + r; col = n0 + 16 * bn + (ln & 15); }
        float x = ((red[0][a][r][ln] + red[1][a][r][ln]) + (red[2][a][r][ln] + red[3][a][r][ln])) +
                  ((red[4][a][r][ln] + red[5][a][r][ln]) + (red[6][a][r][ln] + red[7][a][r][ln]));
        if (row < p.M && col < p.N) {
            x = mask[(size_t)row * p.N + col] > 0.0f ? x : 0.0f;
            C[(size_t)row * p.N + col] = x;
        }
    }
}

// Weight gradient: C[m][n] = sum_b A[b][m] B[b][n]  (A = dY [Kb][M], B = X [Kb][N], both contiguous along their row index; Kb = batch).
// Workgroup tile (32 BM) x (32 BN); both panels staged in LDS as [k][rows] (16-byte coalesced fetches, shared by all waves); a wave owns one
// 32 x 32 block and 1 / KS of every chunk's k (KS = 8 / (BM BN) wave groups; KS > 1 meets in LDS at the end).
template <int BM, int BN, int KC>
__global__ __launch_bounds__(512) void wdw(const P p) {
    constexpr int NB = BM * BN, KS = 8 / NB, TM = 32 * BM, TN = 32 * BN, PA = TM + 4, PB = TN + 4, STAGE = KC * (PA + PB);
    constexpr int NPA = KC * (TM / 4) / 512, NPB = KC * (TN / 4) / 512, KWV = KC / KS;      // k per wave and chunk
    static_assert(NB <= 8 && 8 % NB == 0 && KC * (TM / 4) % 512 == 0 && KC * (TN / 4) % 512 == 0 && KWV % 2 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, g = lane >> 5;
    const int blk = wave % NB, ks = wave / NB, bm = blk / BN, bn = blk % BN;
    const int tph = p.mtiles * p.ntiles, head = blockIdx.x / tph, rem = blockIdx.x % tph, mt = rem / p.ntiles, nt = rem % p.ntiles;
    const int m0 = TM * mt, n0 = TN * nt;
    // p.K = batch (contraction), A [K][M], B [K][N]
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(uniform_ptr(p.A + (size_t)head * p.K * p.M)), rb = make_rsrc(uniform_ptr(p.W + (size_t)head * p.K * p.N));
    unsigned va[NPA], vb[NPB];
    int la[NPA], lb[NPB];
#pragma unroll
    for (int u = 0; u < NPA; ++u) { const int pc = tid + 512 * u, k = pc / (TM / 4), r4 = pc % (TM / 4); va[u] = ((unsigned)k * p.M + m0 + 4 * r4) * 4u; la[u] = k * PA + 4 * r4; }
#pragma unroll
    for (int u = 0; u < NPB; ++u) { const int pc = tid + 512 * u, k = pc / (TN / 4), r4 = pc % (TN / 4); vb[u] = ((unsigned)k * p.N + n0 + 4 * r4) * 4u; lb[u] = KC * PA + k * PB + 4 * r4; }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int n_chunks = p.K / KC;
    f32x4 rga[NPA], rgb[NPB];
    auto gload = [&](int c) {
        const unsigned oob = c < n_chunks ? 0u : 0x80000000u;
#pragma unroll
        for (int u = 0; u < NPA; ++u) rga[u] = ld4(ra, va[u] | oob, (unsigned)c * KC * p.M * 4u);
#pragma unroll
        for (int u = 0; u < NPB; ++u) rgb[u] = ld4(rb, vb[u] | oob, (unsigned)c * KC * p.N * 4u);
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
    float* C = p.C + (size_t)head * p.M * p.N;
    const int col = n0 + 32 * bn + i;
    if (KS == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (row < p.M && col < p.N) C[(size_t)row * p.N + col] = acc[r];
        }
    } else {
        float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(smem);      // [wave][reg][lane]
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
        __syncthreads();
        // block blk: KS partial sums; the KS waves of a block share its 16 registers
#pragma unroll
        for (int e = 0; e < 16 / KS; ++e) {
            const int r = ks * (16 / KS) + e;
            float x = red[blk][r][lane];
#pragma unroll
            for (int q = 1; q < KS; ++q) x = x + red[blk + NB * q][r][lane];
            const int row = m0 + 32 * bm + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (row < p.M && col < p.N) C[(size_t)row * p.N + col] = x;
        }
    }
}

__global__ void empty_k(const P p) { if (p.M < 0) p.C[0] = 1.0f; }

static std::vector<float> hA, hW, hB;
static bool check(const float* dC, int heads, int M, int N, int K, const char* name) {
    std::vector<float> hC((size_t)heads * M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, big = 0;
    srand(7);
    for (int s = 0; s < 4000; ++s) {
        const int hd = rand() % heads, m = rand() % M, n = rand() % N;
        double acc = hB[(size_t)hd * N + n];
        for (int k = 0; k < K; ++k) acc += (double)hA[((size_t)hd * M + m) * K + k] * (double)hW[((size_t)hd * N + n) * K + k];
        if (acc < 0) acc = 0;
        worst = fmax(worst, fabs(acc - hC[((size_t)hd * M + m) * N + n])); big = fmax(big, fabs(acc));
    }
    if (!(worst <= 2e-5 * big) || big < 0.1) { printf("  !! %s M%d x%d: max |err| %.3e\n", name, M, heads, worst); return false; }
    return true;
}

template <typename F>
static float time_it(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 10; ++it) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 200; ++it) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3f / 200;
}

template <int BM, int BN, int RING>
static float run_staged(P p, int heads, const char* name) {
    constexpr int KW = 32 / (BM * BN), KC = 8 * KW, R = 32 * (BM + BN), PITCH = KC + 4, STAGE = R * PITCH;
    const size_t lds = 2 * STAGE * sizeof(float);
    static bool once = false;
    if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(staged<BM, BN, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
    p.mtiles = (p.M + 32 * BM - 1) / (32 * BM); p.ntiles = (p.N + 32 * BN - 1) / (32 * BN);
    const int wgs = heads * p.mtiles * p.ntiles;
    hipMemset(p.C, 0, (size_t)heads * p.M * p.N * 4);
    hipLaunchKernelGGL((staged<BM, BN, RING>), dim3(wgs), dim3(512), lds, 0, p);
    hipDeviceSynchronize();
    if (!check(p.C, heads, p.M, p.N, p.K, name)) return -1.0f;
    return time_it([&] { hipLaunchKernelGGL((staged<BM, BN, RING>), dim3(wgs), dim3(512), lds, 0, p); });
}


template <int E, int BM, int BN, int KW, int RING>
static float run_w(P p, int heads, const char* name) {
    constexpr int R = E * (BM + BN), PAD = E == 32 ? 4 : 8, PITCH = KW + PAD, STAGE = R * PITCH;
    const size_t lds = 8 * 2 * STAGE * sizeof(float);
    static bool once = false;
    if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(wstaged<E, BM, BN, KW, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
    p.mtiles = (p.M + E * BM - 1) / (E * BM); p.ntiles = (p.N + E * BN - 1) / (E * BN);
    const int wgs = heads * p.mtiles * p.ntiles;
    hipMemset(p.C, 0, (size_t)heads * p.M * p.N * 4);
    hipLaunchKernelGGL((wstaged<E, BM, BN, KW, RING>), dim3(wgs), dim3(512), lds, 0, p);
    hipDeviceSynchronize();
    if (!check(p.C, heads, p.M, p.N, p.K, name)) return -1.0f;
    return time_it([&] { hipLaunchKernelGGL((wstaged<E, BM, BN, KW, RING>), dim3(wgs), dim3(512), lds, 0, p); });
}

static std::vector<float> hMask;
static bool check_dx(const float* dC, int heads, int M, int N, int K, const char* name) {
    std::vector<float> hC((size_t)heads * M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, big = 0;
    srand(11);
    for (int s = 0; s < 4000; ++s) {
        const int hd = rand() % heads, m = rand() % M, n = rand() % N;
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)hA[((size_t)hd * M + m) * K + k] * (double)hW[((size_t)hd * K + k) * N + n];
        if (!(hMask[((size_t)hd * M + m) * N + n] > 0)) acc = 0;
        worst = fmax(worst, fabs(acc - hC[((size_t)hd * M + m) * N + n])); big = fmax(big, fabs(acc));
    }
    if (!(worst <= 2e-5 * big) || big < 0.1) { printf("  !! %s M%d x%d: max |err| %.3e\n", name, M, heads, worst); return false; }
    return true;
}
template <int E, int BM, int BN, int KW>
static float run_dx(P p, int heads, const char* name) {
    constexpr int R = E * BM, PAD = E == 32 ? 4 : 8, PITCH = KW + PAD, STAGE = R * PITCH;
    constexpr int REGS = E == 32 ? 16 : 4;
    size_t lds = 8 * 2 * STAGE * sizeof(float);
    const size_t red = 8 * BM * BN * REGS * 64 * sizeof(float);
    if (red > lds) lds = red;
    static bool once = false;
    if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(wdx<E, BM, BN, KW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
    p.mtiles = (p.M + E * BM - 1) / (E * BM); p.ntiles = (p.N + E * BN - 1) / (E * BN);
    const int wgs = heads * p.mtiles * p.ntiles;
    hipMemset(p.C, 0, (size_t)heads * p.M * p.N * 4);
    hipLaunchKernelGGL((wdx<E, BM, BN, KW>), dim3(wgs), dim3(512), lds, 0, p);
    hipDeviceSynchronize();
    if (!check_dx(p.C, heads, p.M, p.N, p.K, name)) return -1.0f;
    return time_it([&] { hipLaunchKernelGGL((wdx<E, BM, BN, KW>), dim3(wgs), dim3(512), lds, 0, p); });
}

static bool check_dw(const float* dC, int heads, int M, int N, int K, const char* name) {
    std::vector<float> hC((size_t)heads * M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, big = 0;
    srand(13);
    for (int s = 0; s < 4000; ++s) {
        const int hd = rand() % heads, m = rand() % M, n = rand() % N;
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)hA[((size_t)hd * K + k) * M + m] * (double)hW[((size_t)hd * K + k) * N + n];
        worst = fmax(worst, fabs(acc - hC[((size_t)hd * M + m) * N + n])); big = fmax(big, fabs(acc));
    }
    if (!(worst <= 2e-5 * big) || big < 0.1) { printf("  !! %s K%d x%d: max |err| %.3e\n", name, K, heads, worst); return false; }
    return true;
}
template <int BM, int BN, int KC>
static float run_dw(P p, int heads, const char* name) {
    constexpr int TM = 32 * BM, TN = 32 * BN, STAGE = KC * (TM + 4 + TN + 4);
    size_t lds = 2 * STAGE * sizeof(float);
    if (lds < 8 * 16 * 64 * 4) lds = 8 * 16 * 64 * 4;
    static bool once = false;
    if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(wdw<BM, BN, KC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
    p.mtiles = p.M / TM; p.ntiles = p.N / TN;
    const int wgs = heads * p.mtiles * p.ntiles;
    hipMemset(p.C, 0, (size_t)heads * p.M * p.N * 4);
    hipLaunchKernelGGL((wdw<BM, BN, KC>), dim3(wgs), dim3(512), lds, 0, p);
    hipDeviceSynchronize();
    if (!check_dw(p.C, heads, p.M, p.N, p.K, name)) return -1.0f;
    return time_it([&] { hipLaunchKernelGGL((wdw<BM, BN, KC>), dim3(wgs), dim3(512), lds, 0, p); });
}

int main() {
    const int N = 1024, K = 1024, HEADS = 4, MMAX = 256;
    hA.resize((size_t)HEADS * MMAX * K); hW.resize((size_t)HEADS * N * K); hB.resize((size_t)HEADS * N);
    srand(1);
    for (auto& x : hA) x = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& x : hW) x = (rand() % 2001 - 1000) * 3e-5f * 1000.0f;
    for (auto& x : hB) x = (rand() % 2001 - 1000) * 1e-4f;
    float *A, *W, *B, *C;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)HEADS * MMAX * N * 4);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    printf("per-launch us over 200 back-to-back launches (N = K = 1024); s = workgroup-staged BMxBN of 32 (ring), w = wave-private E:BMxBN:KW (ring)\n");
    printf("%-9s %6s %7s %7s %7s %7s | %8s %8s %8s %8s %8s %8s %8s %8s %8s\n", "shape", "empty", "direct", "s1x1(1)", "s2x1(1)", "s2x2(1)",
           "w32:1x1:32", "w32:1x1:16", "w32:2x1:16", "w32:1x2:16", "w16:1x1:64", "w16:1x1:32", "w16:2x1:32", "w16:1x2:32", "w16:2x2:32");
    for (int M : {32, 64, 128, 256})
        for (int heads : {1, 2, 4}) {
            // heads are stacked with THIS M as their stride
            std::vector<float> a((size_t)heads * M * K);
            for (int hd = 0; hd < heads; ++hd)
                for (size_t x = 0; x < (size_t)M * K; ++x) a[(size_t)hd * M * K + x] = hA[(size_t)hd * MMAX * K + x];
            hipMemcpy(A, a.data(), a.size() * 4, hipMemcpyHostToDevice);
            std::vector<float> keep = hA;
            hA = a;
            P p{A, W, B, C, M, N, K, (M + 31) / 32, N / 32};
            const int wgs = heads * p.mtiles * p.ntiles;
            const float t_e = time_it([&] { hipLaunchKernelGGL(empty_k, dim3(wgs), dim3(512), 0, 0, p); });
            hipMemset(C, 0, (size_t)heads * M * N * 4);
            hipLaunchKernelGGL(direct, dim3(wgs), dim3(512), 0, 0, p);
            hipDeviceSynchronize();
            check(C, heads, M, N, K, "direct");
            const float t_d = time_it([&] { hipLaunchKernelGGL(direct, dim3(wgs), dim3(512), 0, 0, p); });
            const float t11a = run_staged<1, 1, 1>(p, heads, "s1x1"), t21a = run_staged<2, 1, 1>(p, heads, "s2x1"), t22a = run_staged<2, 2, 1>(p, heads, "s2x2");
            const float w1 = run_w<32, 1, 1, 32, 2>(p, heads, "w32:1x1:32"), w2 = run_w<32, 1, 1, 16, 2>(p, heads, "w32:1x1:16");
            const float w3 = run_w<32, 2, 1, 16, 2>(p, heads, "w32:2x1:16"), w4 = run_w<32, 1, 2, 16, 2>(p, heads, "w32:1x2:16");
            const float w5 = run_w<16, 1, 1, 64, 2>(p, heads, "w16:1x1:64"), w6 = run_w<16, 1, 1, 32, 2>(p, heads, "w16:1x1:32");
            const float w7 = run_w<16, 2, 1, 32, 2>(p, heads, "w16:2x1:32"), w8 = run_w<16, 1, 2, 32, 2>(p, heads, "w16:1x2:32");
            const float w9 = run_w<16, 2, 2, 32, 2>(p, heads, "w16:2x2:32");
            char nm[32]; snprintf(nm, sizeof nm, "M%d x%d", M, heads);
            printf("%-9s %6.2f %7.2f %7.2f %7.2f %7.2f | %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f\n", nm, t_e, t_d, t11a, t21a, t22a, w1, w2, w3, w4, w5, w6, w7, w8, w9);
            hA = keep;
        }
    // ---- data gradient: C = (dY W) (.) [mask > 0], W [K][N] row-contiguous
    float* Mk; hipMalloc(&Mk, (size_t)HEADS * MMAX * N * 4);
    printf("\ndata gradient (A k-contiguous via wave-private LDS, B row-contiguous direct)\n%-9s %10s %10s %10s %10s %10s %10s\n", "shape", "x16:1x1:32", "x16:1x2:32", "x16:2x1:32", "x32:1x1:16", "x32:1x2:16", "x32:1x1:32");
    for (int M : {32, 64, 128, 256})
        for (int heads : {1, 2}) {
            std::vector<float> a((size_t)heads * M * K);
            for (int hd = 0; hd < heads; ++hd)
                for (size_t x = 0; x < (size_t)M * K; ++x) a[(size_t)hd * M * K + x] = hA[(size_t)hd * MMAX * K + x];
            hipMemcpy(A, a.data(), a.size() * 4, hipMemcpyHostToDevice);
            std::vector<float> keep = hA;
            hA = a;
            hMask.resize((size_t)heads * M * N);
            for (auto& x : hMask) x = (rand() % 3) ? 1.0f : 0.0f;
            hipMemcpy(Mk, hMask.data(), hMask.size() * 4, hipMemcpyHostToDevice);
            P p{A, W, Mk, C, M, N, K, 0, 0};
            const float x1 = run_dx<16, 1, 1, 32>(p, heads, "x16:1x1:32"), x2 = run_dx<16, 1, 2, 32>(p, heads, "x16:1x2:32"), x3 = run_dx<16, 2, 1, 32>(p, heads, "x16:2x1:32");
            const float x4 = run_dx<32, 1, 1, 16>(p, heads, "x32:1x1:16"), x5 = run_dx<32, 1, 2, 16>(p, heads, "x32:1x2:16"), x6 = run_dx<32, 1, 1, 32>(p, heads, "x32:1x1:32");
            char nm[32]; snprintf(nm, sizeof nm, "M%d x%d", M, heads);
            printf("%-9s %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", nm, x1, x2, x3, x4, x5, x6);
            hA = keep;
        }
    // ---- weight gradient: C [1024][1024] per head = A^T B, A [Kb][1024], B [Kb][1024]
    printf("\nweight gradient, 1024 x 1024 outputs per head, K = batch (both operands row-contiguous, LDS panels)\n%-9s %10s %10s %10s %10s %10s %10s\n", "shape",
           "2x4:32", "2x4:64", "2x2:64", "2x2:128", "1x2:128", "1x1:256");
    float* Cw; hipMalloc(&Cw, (size_t)2 * 1024 * 1024 * 4);
    for (int Kb : {32, 64, 128, 256})
        for (int heads : {1, 2}) {
            std::vector<float> keepA = hA, keepW = hW;
            hA.assign(hA.begin(), hA.begin() + (size_t)heads * Kb * 1024);     // A [heads][Kb][1024]
            hW.assign(hW.begin(), hW.begin() + (size_t)heads * Kb * 1024);
            hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
            P p{A, W, nullptr, Cw, 1024, 1024, Kb, 0, 0};
            const float d1 = Kb % 32 ? -1 : run_dw<2, 4, 32>(p, heads, "2x4:32"), d2 = Kb % 64 ? -1 : run_dw<2, 4, 64>(p, heads, "2x4:64");
            const float d3 = Kb % 64 ? -1 : run_dw<2, 2, 64>(p, heads, "2x2:64"), d4 = Kb % 128 ? -1 : run_dw<2, 2, 128>(p, heads, "2x2:128");
            const float d5 = Kb % 128 ? -1 : run_dw<1, 2, 128>(p, heads, "1x2:128"), d6 = Kb % 256 ? -1 : run_dw<1, 1, 256>(p, heads, "1x1:256");
            char nm[32]; snprintf(nm, sizeof nm, "Kb%d x%d", Kb, heads);
            printf("%-9s %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", nm, d1, d2, d3, d4, d5, d6);
            hA = keepA; hW = keepW;
        }
    return 0;
}
