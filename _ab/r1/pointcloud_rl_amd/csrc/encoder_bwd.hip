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
//   kernel A (one workgroup per cloud, one wave per 32 active points)
//     sort/unique the cloud's argmax -> active point list; recompute the forward chain for the
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
#include "encoder_common.h"

namespace pcrl {

constexpr int kC2 = 128, kC3 = 256, kSlots = 256, kPiece = 256;   // floats per piece (64 lanes x 4)

// Flat gradient layout = the reference's parameter order inside visual_nn.conv.mlp:
// conv0.weight, conv0.bias, conv1.weight, norm1.weight, norm1.bias, conv2.weight, norm2.weight, norm2.bias
struct GradLayout {
    int C, C1;
    __host__ __device__ int w0() const { return 0; }
    __host__ __device__ int b0() const { return C1 * C; }
    __host__ __device__ int w1() const { return b0() + C1; }
    __host__ __device__ int g1() const { return w1() + kC2 * C1; }
    __host__ __device__ int be1() const { return g1() + kC2; }
    __host__ __device__ int w2() const { return be1() + kC2; }
    __host__ __device__ int g2() const { return w2() + kC3 * kC2; }
    __host__ __device__ int be2() const { return g2() + kC3; }
    __host__ __device__ int total() const { return be2() + kC3; }
};

// Per-cloud operand workspace (floats): block arrays [nblk][32 octets][piece]
struct OpsLayout {
    int MB1;
    __host__ __device__ int dz2() const { return 0; }
    __host__ __device__ int h1() const { return dz2() + 8 * 32 * kPiece; }
    __host__ __device__ int dz1() const { return h1() + 4 * 32 * kPiece; }
    __host__ __device__ int h0() const { return dz1() + 4 * 32 * kPiece; }
    __host__ __device__ int dz0() const { return h0() + MB1 * 32 * kPiece; }
    __host__ __device__ int xb() const { return dz0() + MB1 * 32 * kPiece; }
    __host__ __device__ int total() const { return xb() + 32 * kPiece; }
};
constexpr int kXsFloats = 8 * 64 * 64;   // xhat1 spill: [wave][R][lane]

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
};

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

// BF16: the forward recompute contracts bf16 operands exactly as encoder_fwd_kernel<.., true> does (so that the LayerNorm
// inputs, ReLU masks and argmax relations are those of the forward that produced `argmax`) and so do the two data-gradient
// GEMMs (bf16 transposed weight images, gradients rounded to bf16 as they enter, fp32 accumulation; roundings straight-
// through).  The weight-gradient GEMMs of kernel B stay fp32 on the unrounded operands.
template <int T0, int C1, bool BF16>
__global__ __launch_bounds__(512, 2) void encoder_bwd_points_kernel(const BwdParams p) {
    constexpr PackedLayout L{T0, C1, kC2, kC3};
    constexpr int MB1 = C1 / 32, MB2 = kC2 / 32, MB3 = kC3 / 32;
    constexpr OpsLayout OL{MB1};
    const GradLayout GL{p.cl.C, C1};

    extern __shared__ __attribute__((aligned(16))) char smem[];
    ChanSrc* s_desc = reinterpret_cast<ChanSrc*>(smem);
    unsigned* s_key = reinterpret_cast<unsigned*>(s_desc + PCRL_MAX_CHANNELS);   // [256] sort keys
    int* s_scan = reinterpret_cast<int*>(s_key + kC3);                            // [256]
    int* s_act = s_scan + kC3;                                                    // [256] active point indices
    int* s_misc = s_act + kSlots;                                                 // [8]   n_act, flags (all LDS is dynamic: G17)
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
    int* s_first = reinterpret_cast<int*>(s_pt + kSlots);                         // [256] sorted position where the point's run of keys starts
    float* s_w2 = reinterpret_cast<float*>(s_first + kSlots);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    {
        const f32x4* g = reinterpret_cast<const f32x4*>(p.packed + (BF16 ? L.w2b() : L.w2()));
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
            if (live && gb.x == 0.0f) s_misc[5] = 1;            // xhat not recoverable through a zero gamma: dense path for this cloud
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

        float* pw = p.pw + (long long)b * GL.total();
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
                    buf_store_f1(r_ops, lane_off, op_off(OL.h0(), mb, r), a0[mb][r]);
                }
            }
            // ---- conv1 + LN: xhat1 is spilled to the workspace, h1 stays ---------------------------
            f32x16 a1[MB2];
            if (BF16)
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
                    buf_store_f1(r_ops, lane_off, op_off(OL.h1(), mb, r), a1[mb][r]);
                }
            }
            // ---- conv2 + LN -> xhat2 -------------------------------------------------------------
            f32x16 a2[MB3];
            if (BF16)
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
            unsigned long long own[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) own[k] = __ballot(((unsigned)s_slot[64 * k + lane] >> 5) == (unsigned)wave);
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
            for (int mb = 0; mb < MB3; ++mb)
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
                    buf_store_f1(r_ops, lane_off, op_off(OL.dz2(), mb, r), dz);
                }
            // ---- dH1 = W2^T dz2 ; ReLU + LN1 backward ----------------------------------------------
            f32x16 d1[MB2];
            if (BF16)     // the gradient is rounded to bf16 as it enters the contraction, like the activations of the forward
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
            for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    d1[mb][r] = rstd1 * ((d1[mb][r] - n1) - xh1[mb][r] * n2);
                    buf_store_f1(r_ops, lane_off, op_off(OL.dz1(), mb, r), d1[mb][r]);
                }
            // ---- dH0 = W1^T dz1 ; ReLU backward ----------------------------------------------------
            f32x16 d0[MB1];
            if (BF16)
                dense_layer_bf16<MB1, kC2 / 16>(
                    d0, [&](int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1tb() + (mb * (kC2 / 16) + g) * 256)); },
                    [&](int t) { return d1[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB1, kC2 / 8, 3>(
                    d0, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1t() + (mb * (kC2 / 8) + tq) * 256)); },
                    [&](int t) { return d1[t >> 4][t & 15]; });
#pragma unroll
            for (int mb = 0; mb < MB1; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_f1(r_ops, lane_off, op_off(OL.dz0(), mb, r), ((mask0[mb] >> r) & 1u) ? d0[mb][r] : 0.0f);
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

// ---- kernel B: per-cloud weight-gradient GEMMs ------------------------------------------------
// out[32 x 32 block (mb, nb)] = sum over slots of A[32mb + i][slot] * Bm[32nb + j][slot]
// b_block_stride: distance between two column blocks of the B operand in 16-byte units (32 * 64 in the global workspace,
// n_oct * 64 in the compact LDS copy).
template <int NB, class BPtr>
__device__ __forceinline__ void wgrad_blocks(const float* __restrict__ A, BPtr b4_base, int b_block_stride, int mb, int nb0,
                                             int n_oct, int lane, f32x16 (&acc)[NB]) {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(A) + (long long)mb * 32 * 64 + lane;
    const auto b4 = b4_base + lane;
    // the operands of octet q + 1 are in flight while the 4 * NB MFMAs of octet q issue
    f32x4 a_nxt = {0.f, 0.f, 0.f, 0.f}, b_nxt[NB];
    if (n_oct > 0) {
        a_nxt = a4[0];
#pragma unroll
        for (int n = 0; n < NB; ++n) b_nxt[n] = b4[(nb0 + n) * b_block_stride];
    }
    for (int q = 0; q < n_oct; ++q) {
        const f32x4 a = a_nxt;
        f32x4 bv[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) bv[n] = b_nxt[n];
        if (q + 1 < n_oct) {
            a_nxt = a4[(q + 1) * 64];
#pragma unroll
            for (int n = 0; n < NB; ++n) b_nxt[n] = b4[(nb0 + n) * b_block_stride + (q + 1) * 64];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[n][j], acc[n], 0, 0, 0);
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

template <int C1>
__global__ __launch_bounds__(512, 2) void encoder_bwd_wgrad_kernel(const BwdParams p) {
    constexpr int MB1 = C1 / 32;
    constexpr OpsLayout OL{MB1};
    const GradLayout GL{p.cl.C, C1};
    extern __shared__ __attribute__((aligned(16))) f32x4 s_h1[];      // [4][n_oct][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = blockIdx.x; b < p.cl.B; b += gridDim.x) {
        const float* ops = p.ops + (long long)b * OL.total();
        float* pw = p.pw + (long long)b * GL.total();
        const int n_oct = ((p.n_act[b] + 31) / 32) * 4;
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
        if (wave < MB1) {   // conv0.weight [C1][C] and conv0.bias (column C of the x|1 operand)
            f32x16 acc[1];
            // rows of the x|1 block beyond C are never written by kernel A: mask them out of the B operand
            const float* xb = ops + OL.xb();
            const f32x4* a4 = reinterpret_cast<const f32x4*>(ops + OL.dz0()) + (long long)wave * 32 * 64 + lane;
            const f32x4* b4 = reinterpret_cast<const f32x4*>(xb) + lane;
            const bool live = (lane & 31) <= p.cl.C;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
            for (int q = 0; q < n_oct; ++q) {
                const f32x4 a = a4[q * 64];
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (live) bv = b4[q * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[j], acc[0], 0, 0, 0);
            }
            const int col = lane & 31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (col < p.cl.C) pw[GL.w0() + row * p.cl.C + col] = acc[0][r];
                else if (col == p.cl.C) pw[GL.b0() + row] = acc[0][r];
            }
        }
    }
}

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

static size_t bwd_lds_bytes(int T0, int C1) {
    return sizeof(ChanSrc) * PCRL_MAX_CHANNELS + 4 * (size_t)kC3 * 2 + 4 * (size_t)kSlots + 32 + kC3 + 4 * (size_t)kC3 +
           sizeof(float) * (2 * kC2 + 2 * kC3 + C1 + (size_t)(C1 / 32) * T0 * 64 + 8 * kC2 * 2 + 2 * (size_t)kC3 + 2 * (size_t)kC3 + 3 * (size_t)kSlots + (size_t)kC3 * kC2);
}

template <int T0, int C1, bool BF16>
static int launch_bwd(const BwdParams& p, int grid, hipStream_t stream) {
    static bool attr_set = false;
    const size_t lds = bwd_lds_bytes(T0, C1);
    auto kern = encoder_bwd_points_kernel<T0, C1, BF16>;
    if (!attr_set) {
        PCRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
    PCRL_CHECK_LAUNCH("encoder_bwd_points_kernel");
    constexpr size_t wgrad_lds = 4 * 32 * 64 * sizeof(f32x4);        // h1 operand of one cloud: 128 KB at 256 active points
    static bool wgrad_attr_set = false;
    if (!wgrad_attr_set) {
        PCRL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(encoder_bwd_wgrad_kernel<C1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)wgrad_lds));
        wgrad_attr_set = true;
    }
    hipLaunchKernelGGL(encoder_bwd_wgrad_kernel<C1>, dim3(grid), dim3(512), wgrad_lds, stream, p);
    PCRL_CHECK_LAUNCH("encoder_bwd_wgrad_kernel");
    return PCRL_OK;
}

struct BwdWorkspace {
    size_t ops, xs, pw, nact, total;
};
static BwdWorkspace bwd_workspace(int B, int C, int C1) {
    const OpsLayout OL{C1 / 32};
    const GradLayout GL{C, C1};
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    BwdWorkspace w;
    w.ops = 0;
    w.xs = al(w.ops + sizeof(float) * (size_t)B * OL.total());
    w.pw = al(w.xs + sizeof(float) * (size_t)B * kXsFloats);
    w.nact = al(w.pw + sizeof(float) * (size_t)B * GL.total());
    w.total = al(w.nact + sizeof(int) * (size_t)B);
    return w;
}

}  // namespace pcrl

using namespace pcrl;

extern "C" int pcrl_encoder_num_grads(int32_t c_in, int32_t c1, int32_t c2, int32_t c3, size_t* n) {
    size_t dummy;
    if (int rc = pcrl_encoder_packed_bytes(c_in, c1, c2, c3, &dummy)) return rc;
    if (!n) return fail(PCRL_E_ARG, "n is NULL");
    *n = (size_t)GradLayout{c_in, c1}.total();
    return PCRL_OK;
}

extern "C" int pcrl_encoder_bwd_workspace_bytes(int32_t B, int32_t c_in, int32_t c1, int32_t c2, int32_t c3, size_t* bytes) {
    size_t dummy;
    if (int rc = pcrl_encoder_packed_bytes(c_in, c1, c2, c3, &dummy)) return rc;
    if (!bytes || B < 0) return fail(PCRL_E_ARG, "bad arguments");
    *bytes = bwd_workspace(B, c_in, c1).total;
    return PCRL_OK;
}

static int encoder_bwd_impl(bool bf16, const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                            const pcrl_encoder_weights* w, const void* packed,
                            const int32_t* argmax, const float* grad_pooled, const float* pooled,
                            float* grads, int32_t* n_active,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!clouds || !w || !packed || !argmax || !grad_pooled || !grads) return fail(PCRL_E_ARG, "NULL argument");
    size_t need;
    if (int rc = pcrl_encoder_packed_bytes(w->c_in, w->c1, w->c2, w->c3, &need)) return rc;
    BwdParams p{};
    if (int rc = fill_cloud_params(clouds, aug, w->c_in, &p.cl)) return rc;
    const GradLayout GL{w->c_in, w->c1};
    hipStream_t st = (hipStream_t)stream;
    if (p.cl.B == 0) {
        PCRL_CHECK_HIP(hipMemsetAsync(grads, 0, sizeof(float) * GL.total(), st));
        return PCRL_OK;
    }
    const BwdWorkspace ws = bwd_workspace(p.cl.B, w->c_in, w->c1);
    if (!workspace || workspace_bytes < ws.total) return fail(PCRL_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.total);
    char* base = static_cast<char*>(workspace);
    p.eps = w->eps; p.packed = static_cast<const float*>(packed); p.argmax = argmax; p.gpool = grad_pooled; p.pooled = pooled;
    p.ops = reinterpret_cast<float*>(base + ws.ops); p.xs = reinterpret_cast<float*>(base + ws.xs);
    p.pw = reinterpret_cast<float*>(base + ws.pw); p.n_act = reinterpret_cast<int*>(base + ws.nact);
    p.grads = grads;

    const int grid = min(p.cl.B, num_cus());
    const int T0 = (p.cl.C + 1) / 2;
    int rc = PCRL_E_ARG;
#define PCRL_BWD_CASE(T0_, C1_) \
    if (T0 == T0_ && w->c1 == C1_) rc = bf16 ? launch_bwd<T0_, C1_, true>(p, grid, st) : launch_bwd<T0_, C1_, false>(p, grid, st);
    PCRL_BWD_CASE(2, 64) PCRL_BWD_CASE(3, 64) PCRL_BWD_CASE(4, 64) PCRL_BWD_CASE(5, 64)
    PCRL_BWD_CASE(2, 128) PCRL_BWD_CASE(3, 128) PCRL_BWD_CASE(4, 128) PCRL_BWD_CASE(5, 128)
#undef PCRL_BWD_CASE
    if (rc == PCRL_E_ARG) return fail(PCRL_E_ARG, "no fused kernel for C=%d (supported: 3..10 channels)", p.cl.C);
    if (rc) return rc;
    const int n = GL.total();
    hipLaunchKernelGGL(encoder_bwd_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, p.pw, p.cl.B, n, grads);
    PCRL_CHECK_LAUNCH("encoder_bwd_reduce_kernel");
    if (n_active) PCRL_CHECK_HIP(hipMemcpyAsync(n_active, p.n_act, sizeof(int) * p.cl.B, hipMemcpyDeviceToDevice, st));
    return PCRL_OK;
}

extern "C" int pcrl_encoder_bwd_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                    const pcrl_encoder_weights* w, const void* packed,
                                    const int32_t* argmax, const float* grad_pooled, const float* pooled,
                                    float* grads, int32_t* n_active,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_bwd_impl(false, clouds, aug, w, packed, argmax, grad_pooled, pooled, grads, n_active, workspace, workspace_bytes, stream);
}

extern "C" int pcrl_encoder_bwd_bf16(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                     const pcrl_encoder_weights* w, const void* packed,
                                     const int32_t* argmax, const float* grad_pooled, const float* pooled,
                                     float* grads, int32_t* n_active,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_bwd_impl(true, clouds, aug, w, packed, argmax, grad_pooled, pooled, grads, n_active, workspace, workspace_bytes, stream);
}
