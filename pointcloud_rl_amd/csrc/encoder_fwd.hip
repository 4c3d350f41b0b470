// Fused PointNet encoder forward for gfx950 (MI355X), fp32.
//
// Replaces, per call, the reference's op sequence
//   PointCloudBase.preprocess (pyrl/networks/backbones/pointnet.py:49-73)
//   [RandomJitterPoints / GlobalRotScaleTrans (pyrl/utils/augmentations/pcd_aug.py:306-327, 125-227)]
//   ConvMLP: Conv1d(k=1)+ReLU, Conv1d+LN1d+ReLU, Conv1d+LN1d+ReLU (mlp.py:43-56, nn_layer.py:207-219)
//   feature.max(-1) (pointnet.py:151)
// with one kernel that never materialises a [B, c, N] activation.
//
// Mapping to CDNA4.  One wave owns a tile of 32 points.  Output channels are MFMA rows,
// points are MFMA columns (v_mfma_f32_32x32x2_f32, exact f32 fma chains).  The accumulator
// layout of a layer (lane = point, register = channel) is already the B-operand layout of
// the next layer's k-steps, so the three layers chain in registers; only the weights
// stream, as A operands: conv2 from LDS (128 KB image, loaded once per workgroup), conv1
// from L1/L2 in 1 KB lane-linear pieces.  Per-point LayerNorm is an in-lane sum plus one
// v_permlane32_swap.  The symmetric max-pool is a DPP max over the 32 lanes of a half-wave
// followed by one ds_max_u64 on a {value bits, ~point index} key per winning lane, which
// gives torch's first-index tie rule for free and merges the 8 waves of the workgroup.
#include "encoder_common.h"
#include "encoder_pack.h"

namespace pcrl {

// PointNet.final_mlp as the epilogue of the launch (pcrl_feature_head)
struct HeadRange {
    int begin, count, n_dst;
    float* y[4]; long long ldy[4];
    float* xhat; float* rstd;
    const float* cat_src[2]; float* cat_dst[2]; long long cat_lds[2], cat_ldd[2]; int cat_n[2], cat_div[2];
};
struct HeadParams {
    const float* weight; const float* bias; const float* gamma; const float* beta;
    int F; float eps; int n_ranges;
    HeadRange range[2];
};

// y = weight . pooled + bias, LayerNorm, outputs: run by the `nwaves` waves of a workgroup that holds the cloud's pooled values
// as floats at s_val[2 c] (the low words of the max-pool keys, which the caller has just read out) and may use s_val[2 f + 1]
// (the high words) for y.  Fixed order: lane l sums channels l, l + 64, ... then a DPP/swizzle butterfly; same arithmetic as
// layernorm_rows_fwd_kernel afterwards.  Must be entered by all threads; ends with every output written.
constexpr int kHeadFC = 8;
__device__ __forceinline__ void feature_head_load_weights(const HeadParams& h, int C3, int f0, int lane, float (&wv)[kHeadFC][4]) {
#pragma unroll
    for (int k = 0; k < kHeadFC; ++k) {
        const int f = f0 + k < h.F ? f0 + k : h.F - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[k][j] = lane + 64 * j < C3 ? h.weight[(long long)f * C3 + lane + 64 * j] : 0.0f;
    }
}

// `first`: the weights of the wave's FIRST chunk of features (f0 = 8 wave), fetched by the caller before it had the pooled values
// (encoder_merge_head_kernel: under the partial keys' round trip), or NULL.
template <bool HAS_FIRST>
__device__ __forceinline__ void feature_head_epilogue_t(const HeadParams& h, int cloud, int C3, float* s_val, int tid, int nthreads,
                                                        const float (&first)[kHeadFC][4]) {
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    // wave 0's LayerNorm parameters are fetched now, under the dot products' loads
    float gam[4] = {0.f, 0.f, 0.f, 0.f}, bet[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + 64 * j < h.F) { gam[j] = h.gamma[lane + 64 * j]; bet[j] = h.beta[lane + 64 * j]; }
    }
    __syncthreads();
    {
        // a wave takes eight consecutive features at a time: their 32 weight loads per lane are all in flight together (one
        // feature after the other paid an L2 round trip each: +10 us per cloud), then eight butterflies advance side by side
        constexpr int FC = kHeadFC;
        float pv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = lane + 64 * j < C3 ? s_val[2 * (lane + 64 * j)] : 0.0f;
        for (int f0 = wave * FC; f0 < h.F; f0 += nwaves * FC) {
            float wv[FC][4];
            const float bias_l = (lane < FC && f0 + lane < h.F) ? h.bias[f0 + lane] : 0.0f;
            if (HAS_FIRST && f0 == wave * FC) {
#pragma unroll
                for (int k = 0; k < FC; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) wv[k][j] = first[k][j];
            } else {
                feature_head_load_weights(h, C3, f0, lane, wv);
            }
            float acc[FC];
#pragma unroll
            for (int k = 0; k < FC; ++k) {
                acc[k] = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[k] = __builtin_fmaf(wv[k][j], pv[j], acc[k]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int k = 0; k < FC; ++k) acc[k] += __shfl_xor(acc[k], off, 64);
            if (lane < FC && f0 + lane < h.F) {
                float mine = acc[0];
#pragma unroll
                for (int k = 1; k < FC; ++k) mine = lane == k ? acc[k] : mine;
                s_val[2 * (f0 + lane) + 1] = mine + bias_l;
            }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    int ri = 0;
    if (h.n_ranges > 1 && cloud >= h.range[1].begin) ri = 1;
    const HeadRange& rg = h.range[ri];
    const int row = cloud - rg.begin;
    if (row < 0 || row >= rg.count) return;
    float v[4], s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int f = lane + 64 * j; v[j] = f < h.F ? s_val[2 * f + 1] : 0.0f; s += v[j]; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)h.F;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int f = lane + 64 * j; const float d = f < h.F ? v[j] - mean : 0.0f; q += d * d; }
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / __builtin_sqrtf(q / (float)h.F + h.eps);
    if (rg.rstd && lane == 0) rg.rstd[row] = rstd;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = lane + 64 * j;
        if (f < h.F) {
            const float xh = (v[j] - mean) * rstd;
            if (rg.xhat) rg.xhat[(long long)row * h.F + f] = xh;
            const float y = xh * gam[j] + bet[j];
            for (int d = 0; d < rg.n_dst; ++d) rg.y[d][(long long)row * rg.ldy[d] + f] = y;
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
        if (rg.cat_src[c])
            for (int f = lane; f < rg.cat_n[c]; f += 64)
                rg.cat_dst[c][(long long)row * rg.cat_ldd[c] + f] = rg.cat_src[c][(long long)(row / rg.cat_div[c]) * rg.cat_lds[c] + f];
}

__device__ __forceinline__ void feature_head_epilogue(const HeadParams& h, int cloud, int C3, float* s_val, int tid, int nthreads) {
    const float none[kHeadFC][4] = {};
    feature_head_epilogue_t<false>(h, cloud, C3, s_val, tid, nthreads, none);
}

struct FwdParams {
    CloudParams cl;
    int S, tiles_total, tiles_per_seg;
    float eps;
    const float* packed;
    float* pooled;
    int* argmax;
    unsigned long long* partial;   // [B][S][C3] keys when S > 1
    HeadParams head;               // head.weight != NULL: feature head epilogue
};

// BF16: conv1 / conv2 contract bf16 operands (weights rounded once by the pack kernel, activations rounded as they are
// fed to the next layer) with fp32 accumulation; conv0, both LayerNorms and the max-pool are unchanged fp32 code.
// SPLIT (experimental, never together with BF16): conv1 / conv2 in ~fp32 accuracy on the bf16 matrix cores, every operand split
// into three bf16 terms (dense_layer_split); conv2's hi and mid weight images live in LDS, the lo image streams from L2.
#ifndef PCRL_FWD_RING
#define PCRL_FWD_RING 3
#endif
#ifndef PCRL_FWD_W1_LDS
#define PCRL_FWD_W1_LDS 1
#endif
// LDS bytes of everything but conv1 row blocks (small tables + the conv2 image; the split mode keeps two bf16 images = the same
// bytes as the fp32 one), and how many 32-row blocks of the fp32 conv1 image fit in what is left of the CU's 160 KB.
__host__ __device__ constexpr size_t fwd_lds_base_bytes(int T0, int C1, int C2, int C3, bool bf16) {
    return sizeof(ChanSrc) * PCRL_MAX_CHANNELS + 8 * (size_t)C3 +
           sizeof(float) * ((size_t)C3 * C2 / (bf16 ? 2 : 1) + (size_t)(C1 / 32) * T0 * 64 + C1 + 2 * C2 + 2 * C3);
}
__host__ __device__ constexpr int fwd_w1_lds_blocks(int T0, int C1, int C2, int C3, bool bf16, bool split) {
    if (bf16 || split || !PCRL_FWD_W1_LDS) return 0;
    const size_t room = 160 * 1024 - fwd_lds_base_bytes(T0, C1, C2, C3, false);
    const int fit = (int)(room / (sizeof(float) * 32 * (size_t)C1));
    return fit < C2 / 32 ? fit : C2 / 32;
}

#ifdef PCRL_FWD_STAMPS
// Development build only (-DPCRL_FWD_STAMPS): shader-clock stamps at the phase boundaries of the tile loop (first work item of
// every workgroup), read back with pcrl_debug_fwd_stamps (tools/fwd_stamps.py; profiles/r03_fwd_stamps.md).
__device__ unsigned long long g_fwd_stamps[256 * 8 * 8][8];
#define PCRL_FSTAMP(k) do { if (lane == 0 && work == (int)blockIdx.x && blockIdx.x < 256 && (tile - t_begin) / nwaves < 8) \
    g_fwd_stamps[(blockIdx.x * 8 + wave) * 8 + (tile - t_begin) / nwaves][k] = __builtin_readcyclecounter(); } while (0)
#else
#define PCRL_FSTAMP(k) do { } while (0)
#endif

template <int T0, int C1, int C2, int C3, bool BF16, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void encoder_fwd_kernel(const FwdParams p) {
    constexpr PackedLayout L{T0, C1, C2, C3};
    constexpr int MB1 = C1 / 32, MB2 = C2 / 32, MB3 = C3 / 32;
    // LDS image: small tables first (DS instructions carry a 16-bit offset, so everything that is
    // addressed with per-register constants must sit below 64 KB), the 128 KB conv2 image last.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ChanSrc* s_desc = reinterpret_cast<ChanSrc*>(smem);
    unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(s_desc + PCRL_MAX_CHANNELS);
    float* s_ln1 = reinterpret_cast<float*>(s_keys + C3);
    float* s_ln2 = s_ln1 + 2 * C2;
    float* s_b0 = s_ln2 + 2 * C3;
    float* s_w0 = s_b0 + C1;
    // exact fp32 mode: as many 32-row blocks of the conv1 image as the 160 KB of LDS still hold next to the conv2 image
    // (K1: 3 of 4) sit here too; the rest streams from L2 as before
    constexpr int W1L = fwd_w1_lds_blocks(T0, C1, C2, C3, BF16, SPLIT);
    float* s_w1 = s_w0 + MB1 * T0 * 64;
    float* s_w2 = s_w1 + W1L * 32 * C1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;

    {   // prologue: weights -> LDS, once per workgroup
        const f32x4* g = reinterpret_cast<const f32x4*>(p.packed + (SPLIT ? L.w2s(0) : BF16 ? L.w2b() : L.w2()));
        f32x4* s = reinterpret_cast<f32x4*>(s_w2);
        constexpr int N16 = BF16 ? C3 * C2 / 8 : C3 * C2 / 4;
        if (nthreads == 512) stage_to_lds<512, N16>(s, g, tid);
        else for (int i = tid; i < N16; i += nthreads) s[i] = g[i];
        for (int i = tid; i < MB1 * T0 * 64; i += nthreads) s_w0[i] = p.packed[L.w0() + i];
        if constexpr (W1L > 0) {
            const f32x4* g1 = reinterpret_cast<const f32x4*>(p.packed + L.w1());
            f32x4* s1 = reinterpret_cast<f32x4*>(s_w1);
            if (nthreads == 512) stage_to_lds<512, W1L * 8 * C1>(s1, g1, tid);
            else for (int i = tid; i < W1L * 8 * C1; i += nthreads) s1[i] = g1[i];
        }
        for (int i = tid; i < C1; i += nthreads) s_b0[i] = p.packed[L.b0() + i];
        ln_pair_table(s_ln1, p.packed + L.ln1(), C2, tid, nthreads);
        ln_pair_table(s_ln2, p.packed + L.ln2(), C3, tid, nthreads);
        if (tid < PCRL_MAX_CHANNELS) s_desc[tid] = p.cl.ch[tid];
    }
    const __amdgpu_buffer_rsrc_t r_packed = make_rsrc(p.packed, 4u * (unsigned)L.total());
    const unsigned lane16 = 16u * (unsigned)lane;
    const f32x4* s_w2v = reinterpret_cast<const f32x4*>(s_w2);
    const f32x4* s_w1v = reinterpret_cast<const f32x4*>(s_w1);
    // [r6] The conv1 / conv2 operand reads of the exact fp32 path address the 24 + 128 KB of LDS images as (lane's 16 bytes) + a compile-time
    // offset of up to 160 KB, and a DS instruction carries 16 bits of it.  Left to itself the compiler formed ~10 base registers for the
    // reads of one tile and, at 256 registers, SPILLED them: 12 scratch reloads inside conv2's MFMA stream, each in front of the
    // ds_read_b128 it feeds.  Three explicit window bases (lane16 + 0 / 64 KB / 128 KB), the upper two hidden from the optimiser, cover
    // everything: every read is base register + immediate.
    typedef __attribute__((address_space(3))) const f32x4 lds_f4;
    const unsigned lds_w0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem) + lane16;
    unsigned lds_w1 = lds_w0 + 65536u, lds_w2 = lds_w0 + 131072u;
    asm volatile("" : "+v"(lds_w1), "+v"(lds_w2));
    constexpr unsigned kW1Off = (unsigned)(sizeof(ChanSrc) * PCRL_MAX_CHANNELS + sizeof(unsigned long long) * C3 +
                                           sizeof(float) * (2 * C2 + 2 * C3 + C1 + MB1 * T0 * 64));
    constexpr unsigned kW2Off = kW1Off + (unsigned)sizeof(float) * W1L * 32 * C1;
    auto lds_image = [&](unsigned byte) -> f32x4 {       // `byte` is a constant after unrolling: the window select folds away
        const unsigned win = byte >> 16, imm = byte & 0xFFFFu;
        const unsigned base = win == 0 ? lds_w0 : win == 1 ? lds_w1 : lds_w2;
        return *(lds_f4*)(unsigned long)(base + imm);
    };

    // SUBSAMPLE with a count decided on the device (CloudParams::n_ptr): positions [n_pts, N) do not exist in this call -- their tiles are
    // skipped, and the lanes of the last live tile past n_pts hold copies of point n_pts - 1, exactly like the lanes past N of a ragged cloud
    const int n_pts = p.cl.n_ptr ? __builtin_amdgcn_readfirstlane(max(1, min(*p.cl.n_ptr, p.cl.N))) : p.cl.N;
    const int tiles_live = (n_pts + 31) >> 5;

    for (int work = blockIdx.x; work < p.cl.B * p.S; work += gridDim.x) {
        const int b = work / p.S, seg = work - b * p.S;
        const int t_begin = seg * p.tiles_per_seg;
        const int t_end = min(min(t_begin + p.tiles_per_seg, p.tiles_total), tiles_live);
        __syncthreads();   // previous read-out of s_keys (and the prologue) is complete
        // {value +0.0, point 0}: what a channel that is zero everywhere (ReLU-dead) must report, so zero maxima never
        // have to be written by anybody
        for (int i = tid; i < C3; i += nthreads) s_keys[i] = 0x00000000FFFFFFFFull;
        __syncthreads();

        for (int tile = t_begin + wave; tile < t_end; tile += nwaves) {
            const int pidx = tile * 32 + l31;
            const bool valid = pidx < n_pts;
            const int pc = valid ? pidx : n_pts - 1;

            PCRL_FSTAMP(0);
            // ---- preprocess (+ augmentation) ---------------------------------------------
            const f32x16 x = load_point<T0>(p.cl, s_desc, b, pc);

            const unsigned half_mask = half ? 0xFFFFFFFFu : 0u;
            // ---- conv0 + bias + ReLU ------------------------------------------------------
            f32x16 a0[MB1];
#pragma unroll
            for (int mb = 0; mb < MB1; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a0[mb][r] = s_b0[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
                for (int t = 0; t < T0; ++t) {
                    const float bop = half_select(x[2 * t], x[2 * t + 1], half_mask);
                    a0[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_w0[(mb * T0 + t) * 64 + lane], bop, a0[mb], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) a0[mb][r] = relu_nan(a0[mb][r]);
            }

            PCRL_FSTAMP(1);
            // ---- conv1 + LN + ReLU --------------------------------------------------------
            f32x16 a1[MB2];
            if (SPLIT)
                dense_layer_split<MB2, C1 / 16>(
                    a1, [&](int k, int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1s(k) + (mb * (C1 / 16) + g) * 256)); },
                    [&](int t) { return a0[t >> 4][t & 15]; });
            else if (BF16)
                dense_layer_bf16<MB2, C1 / 16>(
                    a1, [&](int mb, int g) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1b() + (mb * (C1 / 16) + g) * 256)); },
                    [&](int t) { return a0[t >> 4][t & 15]; });
            else
                // (the L2-fed row blocks' operand ring is pinned: left to the scheduler every load sank to just in front of its wait)
                dense_layer_mfma_stream<MB2, C1 / 8, PCRL_FWD_RING>(
                    a1, [&](int mb, int tq) {
                        return mb < W1L ? ((BF16 || SPLIT) ? s_w1v[(mb * (C1 / 8) + tq) * 64 + lane] : lds_image(kW1Off + 1024u * (unsigned)(mb * (C1 / 8) + tq)))
                                        : buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1() + (mb * (C1 / 8) + tq) * 256)); },
                    [&](int t) { return a0[t >> 4][t & 15]; });
            PCRL_FSTAMP(2);
            // ReLU as an integer max (one instruction per register instead of compare + select); a NaN of either sign is restored
            // below: a point with any NaN channel has a NaN variance, and LayerNorm then makes ALL its channels NaN
            const bool nan_pt1 = ln_relu_acc<C2, true>(a1, s_ln1, half, p.eps);
            if (__builtin_expect(__ballot(nan_pt1) != 0ull, 0)) {
#pragma unroll
                for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (nan_pt1) a1[mb][r] = u2f(0x7FC00000u);
            }

            PCRL_FSTAMP(3);
            // ---- conv2 + LN + ReLU --------------------------------------------------------
            f32x16 a2[MB3];
            if (SPLIT)
                dense_layer_split<MB3, C2 / 16>(
                    a2, [&](int k, int mb, int g) {
                        return k < 2 ? s_w2v[k * (C3 * C2 / 8) + (mb * (C2 / 16) + g) * 64 + lane]
                                     : buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w2s(2) + (mb * (C2 / 16) + g) * 256)); },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            else if (BF16)
                dense_layer_bf16<MB3, C2 / 16>(
                    a2, [&](int mb, int g) { return s_w2v[(mb * (C2 / 16) + g) * 64 + lane]; },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            else
                dense_layer_mfma<MB3, C2 / 8, 2>(
                    a2, [&](int mb, int tq) { return lds_image(kW2Off + 1024u * (unsigned)(mb * (C2 / 8) + tq)); },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            PCRL_FSTAMP(4);
#ifdef PCRL_FWD_ABLATE_TAIL
            // Development build only (tools/r4_fwd_ablate.sh): LayerNorm-2 and the max-pool REMOVED (the conv2 accumulators are
            // declared used, nothing is computed from them).  Not a forward pass: an upper bound for what any redesign of those two phases can reach
            // (profiles/r04_bf16_fwd_ceiling.md).
            {
#pragma unroll
                for (int mb = 0; mb < MB3; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(a2[mb][r]));      // the accumulators count as used: no instruction
                continue;
            }
#endif
            // No ReLU instructions after LayerNorm-2: the pool compares the raw bits as SIGNED integers, where every value <= 0
            // (and -0, and a NaN with the sign bit) sorts below the smallest positive float, i.e. below max(key, 1) -- exactly the
            // lanes the ReLU would have zeroed.  What reaches a key is positive, so the keys' unsigned order is unchanged.
            const bool nan_pt = ln_relu_acc<C3, true, true>(a2, s_ln2, half, p.eps);
            if (__builtin_expect(__ballot(nan_pt) != 0ull, 0)) {
                // torch: a NaN wins the max and the first NaN's index is returned.  Marker 0x7FFFFFFF: the largest signed
                // AND (among what reaches the keys) the largest unsigned pattern; read out as the canonical NaN.
#pragma unroll
                for (int mb = 0; mb < MB3; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (nan_pt) a2[mb][r] = u2f(0x7FFFFFFFu);
            }

            PCRL_FSTAMP(5);
            // ---- symmetric max-pool with first-index argmax --------------------------------
            // lanes past N hold a copy of point N - 1 and report that index, so no validity test is needed below
            const unsigned inv_idx = ~(unsigned)pc;
            if (tile == t_begin + wave) {
                // The wave's first tile of this cloud segment: the keys hold (almost) nothing yet, so reduce across the 32
                // lanes first and let only the winner of each half touch the key.
#pragma unroll
                for (int mb = 0; mb < MB3; ++mb) {
                    // 16 independent reductions advance together: each DPP step of one register fills the
                    // wait states of the others, and the 16 conditional key updates are issued back to back.
                    int m[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = (int)f2u(a2[mb][r]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = imax_(m[r], (int)dpp_u<0xB1>((unsigned)m[r]));
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = imax_(m[r], (int)dpp_u<0x4E>((unsigned)m[r]));
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = imax_(m[r], (int)dpp_u<0x141>((unsigned)m[r]));
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = imax_(m[r], (int)dpp_u<0x140>((unsigned)m[r]));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        auto sw = __builtin_amdgcn_permlane16_swap((unsigned)m[r], (unsigned)m[r], false, false);
                        // max(.., 1): a maximum that is not positive matches no lane (nobody reports it: the key's initial value
                        // already says "zero everywhere, point 0")
                        m[r] = imax_(imax_((int)sw[0], (int)sw[1]), 1);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned v = f2u(a2[mb][r]);
                        if ((int)v == m[r]) {
                            const int ch = acc_chan(mb * 16 + r, 0) + 4 * half;
                            atomicMax(&s_keys[ch], ((unsigned long long)v << 32) | inv_idx);
                        }
                    }
                }
            } else {
                // Later tiles: a point can only matter if it reaches the value already in the channel's key (which only
                // grows, so a stale read errs on the safe side; equality passes for the first-index rule).  That is rare
                // -- the k-th tile of a cloud holds the running maximum with probability ~1/k -- so the cross-lane
                // reduction is skipped and the few qualifying lanes update the key themselves: one LDS read and two
                // compares per register instead of four DPP steps, a lane swap and three more instructions.
                const unsigned* s_key_hi = reinterpret_cast<const unsigned*>(s_keys) + 1;
#pragma unroll
                for (int mb = 0; mb < MB3; ++mb) {
                    unsigned cur[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[r] = s_key_hi[2 * (acc_chan(mb * 16 + r, 0) + 4 * half)];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned v = f2u(a2[mb][r]);
                        const bool upd = (int)v >= imax_((int)cur[r], 1);   // key values are >= 0 as integers (positive floats, or the NaN marker)
                        // wave-uniform early-out: in a cloud's later tiles no lane of the wave reaches the key for most registers, and
                        // then nothing but the compare is issued (no exec-mask round trip, no predicated-off ds_max_u64)
                        if (__builtin_expect(__ballot(upd) != 0ull, 0)) {
                            if (upd) {
                                const int ch = acc_chan(mb * 16 + r, 0) + 4 * half;
                                atomicMax(&s_keys[ch], ((unsigned long long)v << 32) | inv_idx);
                            }
                        }
                    }
                }
            }
            PCRL_FSTAMP(6);
        }
        __syncthreads();
        for (int c = tid; c < C3; c += nthreads) {
            const unsigned long long key = s_keys[c];
            if (p.S == 1) {
                unsigned vb = (unsigned)(key >> 32);
                if (vb > 0x7F800000u) vb = 0x7FC00000u;
                p.pooled[(long long)b * C3 + c] = u2f(vb);
                p.argmax[(long long)b * C3 + c] = (int)~(unsigned)key;
            } else {
                p.partial[((long long)b * p.S + seg) * C3 + c] = key;
            }
        }
    }
    // Feature head (PointNet.final_mlp) of the clouds this workgroup finished -- after the tile loop, so that its parameters
    // are not live across it (inside the loop they cost the tile loop 17 spilled VGPRs / 55 SGPRs and 3 % of its speed).
    if (p.S == 1 && p.head.weight) {
        float* s_val = reinterpret_cast<float*>(s_keys);
        for (int b = blockIdx.x; b < p.cl.B; b += gridDim.x) {
            __syncthreads();
            for (int c = tid; c < C3; c += nthreads) s_val[2 * c] = p.pooled[(long long)b * C3 + c];
            feature_head_epilogue(p.head, b, C3, s_val, tid, nthreads);
        }
    }
}

// ---- wide last layer (c3 = 512 / 1024: the class default mlp_spec = [64, 128, 1024], pointnet.py:81) ------------------------------
// One point's c3 conv2 outputs no longer fit a lane's registers (c3 / 2 accumulators per lane), and LayerNorm-2 needs all of them
// twice (mean, then the centred sum of squares) before the first output can be formed.  The tile loop therefore runs conv2 THREE
// times in chunks of 256 output channels (8 row blocks = 128 accumulators): pass A accumulates the sum, pass B the centred
// squares, pass C normalises and pools.  The partial sums advance in exactly the order ln_center_rstd walks the row blocks
// (oracle/pcrl_oracle.c::half_sum), so the result is bit-identical to the oracle's as for the narrow shapes.  conv2's weights
// stream from the packed image in L2 (c3 x c2 fp32 = 512 KB does not fit LDS); fp32 only, no feature-head epilogue.  Three
// times the conv2 work of a (hypothetical) register-resident kernel -- no shipped SAC / DrQ config uses this shape.
template <int T0, int C1, int C2, int C3>
__global__ __launch_bounds__(512, 1) void encoder_fwd_wide_kernel(const FwdParams p) {
    constexpr PackedLayout L{T0, C1, C2, C3};
    constexpr int MB1 = C1 / 32, MB2 = C2 / 32, NCH = C3 / 256, MBC = 8;
    static_assert(C3 % 256 == 0 && C3 > 256, "wide kernel: c3 = 512, 768, 1024");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ChanSrc* s_desc = reinterpret_cast<ChanSrc*>(smem);
    unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(s_desc + PCRL_MAX_CHANNELS);
    float* s_ln1 = reinterpret_cast<float*>(s_keys + C3);
    float* s_ln2 = s_ln1 + 2 * C2;
    float* s_b0 = s_ln2 + 2 * C3;
    float* s_w0 = s_b0 + C1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    for (int i = tid; i < MB1 * T0 * 64; i += nthreads) s_w0[i] = p.packed[L.w0() + i];
    for (int i = tid; i < C1; i += nthreads) s_b0[i] = p.packed[L.b0() + i];
    ln_pair_table(s_ln1, p.packed + L.ln1(), C2, tid, nthreads);
    ln_pair_table(s_ln2, p.packed + L.ln2(), C3, tid, nthreads);
    if (tid < PCRL_MAX_CHANNELS) s_desc[tid] = p.cl.ch[tid];
    const __amdgpu_buffer_rsrc_t r_packed = make_rsrc(p.packed, 4u * (unsigned)L.total());
    const unsigned lane16 = 16u * (unsigned)lane;
    const f32x4* s_gb2 = reinterpret_cast<const f32x4*>(s_ln2);
    const unsigned* s_key_hi = reinterpret_cast<const unsigned*>(s_keys) + 1;

    // SUBSAMPLE with a count decided on the device (CloudParams::n_ptr): positions [n_pts, N) do not exist in this call -- their tiles are
    // skipped, and the lanes of the last live tile past n_pts hold copies of point n_pts - 1, exactly like the lanes past N of a ragged cloud
    const int n_pts = p.cl.n_ptr ? __builtin_amdgcn_readfirstlane(max(1, min(*p.cl.n_ptr, p.cl.N))) : p.cl.N;
    const int tiles_live = (n_pts + 31) >> 5;

    for (int work = blockIdx.x; work < p.cl.B * p.S; work += gridDim.x) {
        const int b = work / p.S, seg = work - b * p.S;
        const int t_begin = seg * p.tiles_per_seg;
        const int t_end = min(min(t_begin + p.tiles_per_seg, p.tiles_total), tiles_live);
        __syncthreads();
        for (int i = tid; i < C3; i += nthreads) s_keys[i] = 0x00000000FFFFFFFFull;      // {value +0.0, point 0}
        __syncthreads();
        for (int tile = t_begin + wave; tile < t_end; tile += nwaves) {
            const int pidx = tile * 32 + l31;
            const int pc = pidx < n_pts ? pidx : n_pts - 1;
            const f32x16 x = load_point<T0>(p.cl, s_desc, b, pc);
            const unsigned half_mask = half ? 0xFFFFFFFFu : 0u;
            f32x16 a0[MB1];
#pragma unroll
            for (int mb = 0; mb < MB1; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a0[mb][r] = s_b0[acc_chan(mb * 16 + r, 0) + 4 * half];
#pragma unroll
                for (int t = 0; t < T0; ++t) {
                    const float bop = half_select(x[2 * t], x[2 * t + 1], half_mask);
                    a0[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_w0[(mb * T0 + t) * 64 + lane], bop, a0[mb], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) a0[mb][r] = relu_nan(a0[mb][r]);
            }
            f32x16 a1[MB2];
            dense_layer_mfma<MB2, C1 / 8, 3>(
                a1, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, 4u * (unsigned)(L.w1() + (mb * (C1 / 8) + tq) * 256)); },
                [&](int t) { return a0[t >> 4][t & 15]; });
            const bool nan_pt1 = ln_relu_acc<C2, true>(a1, s_ln1, half, p.eps);
            if (__builtin_expect(__ballot(nan_pt1) != 0ull, 0)) {
#pragma unroll
                for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (nan_pt1) a1[mb][r] = u2f(0x7FC00000u);
            }
            // one chunk of conv2: row blocks 8 ch .. 8 ch + 7
            auto conv2_chunk = [&](f32x16 (&a2)[MBC], int ch) {
                const unsigned base = 4u * (unsigned)(L.w2() + ch * MBC * (C2 / 8) * 256);
                dense_layer_mfma<MBC, C2 / 8, 3>(
                    a2, [&](int mb, int tq) { return buf_load_f4(r_packed, lane16, base + 4u * (unsigned)((mb * (C2 / 8) + tq) * 256)); },
                    [&](int t) { return a1[t >> 4][t & 15]; });
            };
            // pass A: sum over the C3 channels (two half-sums of 4 interleaved partials, combined as ln_center_rstd does)
            f32x2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll 1
            for (int ch = 0; ch < NCH; ++ch) {
                f32x16 a2[MBC];
                conv2_chunk(a2, ch);
#pragma unroll
                for (int mb = 0; mb < MBC; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) { p01 = p01 + PCRL_PAIR(a2[mb], r); p23 = p23 + PCRL_PAIR(a2[mb], r + 2); }
            }
            float lo, hi;
            both_halves((p01[0] + p01[1]) + (p23[0] + p23[1]), lo, hi);
            const float mean = (lo + hi) / (float)C3;
            const f32x2 mean2 = {mean, mean};
            // pass B: centred sum of squares
            p01 = f32x2{0.f, 0.f}; p23 = f32x2{0.f, 0.f};
#pragma unroll 1
            for (int ch = 0; ch < NCH; ++ch) {
                f32x16 a2[MBC];
                conv2_chunk(a2, ch);
#pragma unroll
                for (int mb = 0; mb < MBC; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) {
                        const f32x2 c01 = PCRL_PAIR(a2[mb], r) - mean2, c23 = PCRL_PAIR(a2[mb], r + 2) - mean2;
                        p01 = __builtin_elementwise_fma(c01, c01, p01);
                        p23 = __builtin_elementwise_fma(c23, c23, p23);
                    }
            }
            both_halves((p01[0] + p01[1]) + (p23[0] + p23[1]), lo, hi);
            const float var = (lo + hi) / (float)C3;
            const bool nan_pt = var != var;
            const float rstd = 1.0f / __builtin_sqrtf(var + p.eps);
            const f32x2 rstd2 = {rstd, rstd};
            const bool any_nan = __ballot(nan_pt) != 0ull;
            const unsigned inv_idx = ~(unsigned)pc;
            // pass C: normalise (no ReLU instruction: the pool orders the raw bits as signed integers) and pool against the keys
#pragma unroll 1
            for (int ch = 0; ch < NCH; ++ch) {
                f32x16 a2[MBC];
                conv2_chunk(a2, ch);
#pragma unroll
                for (int mb = 0; mb < MBC; ++mb) {
                    unsigned cur[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[r] = s_key_hi[2 * (256 * ch + acc_chan(mb * 16 + r, 0) + 4 * half)];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x4 g4 = s_gb2[(256 * ch + acc_chan(mb * 16 + r, 0) + 4 * half) >> 1];
                        const f32x2 c = PCRL_PAIR(a2[mb], r) - mean2;
                        const f32x2 y = __builtin_elementwise_fma(c * rstd2, __builtin_shufflevector(g4, g4, 0, 1), __builtin_shufflevector(g4, g4, 2, 3));
                        a2[mb][r] = y[0]; a2[mb][r + 1] = y[1];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        unsigned v = f2u(a2[mb][r]);
                        if (any_nan && nan_pt) v = 0x7FFFFFFFu;          // torch: a NaN wins the max, the first NaN's index is returned
                        const bool upd = (int)v >= imax_((int)cur[r], 1);
                        if (__builtin_expect(__ballot(upd) != 0ull, 0)) {
                            if (upd) atomicMax(&s_keys[256 * ch + acc_chan(mb * 16 + r, 0) + 4 * half], ((unsigned long long)v << 32) | inv_idx);
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int c = tid; c < C3; c += nthreads) {
            const unsigned long long key = s_keys[c];
            if (p.S == 1) {
                unsigned vb = (unsigned)(key >> 32);
                if (vb > 0x7F800000u) vb = 0x7FC00000u;
                p.pooled[(long long)b * C3 + c] = u2f(vb);
                p.argmax[(long long)b * C3 + c] = (int)~(unsigned)key;
            } else {
                p.partial[((long long)b * p.S + seg) * C3 + c] = key;
            }
        }
    }
}

// Second stage of the split-cloud pool: max over the S partial keys of a cloud.
__global__ void encoder_merge_kernel(const unsigned long long* __restrict__ partial, int B, int S, int C3,
                                     float* __restrict__ pooled, int* __restrict__ argmax) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * C3) return;
    const int b = (int)(i / C3), c = (int)(i - (long long)b * C3);
    unsigned long long key = 0ull;
    for (int s = 0; s < S; ++s) {
        const unsigned long long k = partial[((long long)b * S + s) * C3 + c];
        key = k > key ? k : key;
    }
    unsigned vb = (unsigned)(key >> 32);
    if (vb > 0x7F800000u) vb = 0x7FC00000u;
    pooled[i] = u2f(vb);
    argmax[i] = (int)~(unsigned)key;
}

// The same merge for launches with a feature head: one workgroup per cloud (thread c owns channel c), then the head.
// NT threads: 256 without a feature head; 1 024 with one, so that every wave has ONE chunk of eight features (F <= 128) whose weights
// it requests first thing -- they do not depend on the keys -- and the keys arrive eight segments at a time instead of one by one.
template <int NT>
__global__ __launch_bounds__(NT) void encoder_merge_head_kernel(const unsigned long long* __restrict__ partial, int S, int C3,
                                                                float* __restrict__ pooled, int* __restrict__ argmax, const HeadParams head) {
    __shared__ float s_val[2 * 256];
    const int b = blockIdx.x, c = threadIdx.x;
    float first[kHeadFC][4];
    const bool early = head.weight != nullptr && (int)(threadIdx.x >> 6) * kHeadFC < head.F;
    if (early) feature_head_load_weights(head, C3, (int)(threadIdx.x >> 6) * kHeadFC, threadIdx.x & 63, first);
    if (c < C3) {
        unsigned long long key = 0ull;
        for (int s0 = 0; s0 < S; s0 += 8) {
            unsigned long long k[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) k[i] = partial[((long long)b * S + min(s0 + i, S - 1)) * C3 + c];
#pragma unroll
            for (int i = 0; i < 8; ++i) key = k[i] > key ? k[i] : key;
        }
        unsigned vb = (unsigned)(key >> 32);
        if (vb > 0x7F800000u) vb = 0x7FC00000u;
        pooled[(long long)b * C3 + c] = u2f(vb);
        argmax[(long long)b * C3 + c] = (int)~(unsigned)key;
        s_val[2 * c] = u2f(vb);
    }
    if (head.weight) feature_head_epilogue_t<true>(head, b, C3, s_val, threadIdx.x, NT, first);    // (a wave without a first chunk has no chunk at all)
    else feature_head_epilogue(head, b, C3, s_val, threadIdx.x, NT);
}

__global__ __launch_bounds__(256) void encoder_pack_kernel(const PackJob j) { encoder_pack_block(j, (int)blockIdx.x, (int)threadIdx.x); }

int fill_cloud_params(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug, int expect_channels, CloudParams* out) {
    if (!clouds) return fail(PCRL_E_ARG, "clouds is NULL");
    if (clouds->B < 0 || clouds->N < 1) return fail(PCRL_E_ARG, "bad cloud shape B=%d N=%d", clouds->B, clouds->N);
    if (clouds->nseg < 1 || clouds->nseg > PCRL_MAX_SEG) return fail(PCRL_E_ARG, "nseg=%d out of range", clouds->nseg);
    CloudParams& p = *out;
    p = CloudParams{};
    p.B = clouds->B; p.N = clouds->N;
    p.row_div = clouds->row_div > 1 ? clouds->row_div : 1;
    if (p.B % p.row_div) return fail(PCRL_E_ARG, "B=%d is not a multiple of row_div=%d", p.B, p.row_div);
    int c = 0;
    for (int s = 0; s < clouds->nseg; ++s) {
        const pcrl_feat_seg& sg = clouds->seg[s];
        if (!sg.ptr || sg.channels < 1) return fail(PCRL_E_ARG, "segment %d is empty", s);
        if (sg.dtype != PCRL_DT_F32 && sg.dtype != PCRL_DT_U8 && sg.dtype != PCRL_DT_BOOL) return fail(PCRL_E_ARG, "segment %d: bad dtype", s);
        const size_t esz = sg.dtype == PCRL_DT_F32 ? 4 : 1;
        for (int k = 0; k < sg.channels; ++k, ++c) {
            if (c >= PCRL_MAX_CHANNELS) return fail(PCRL_E_ARG, "more than %d channels", PCRL_MAX_CHANNELS);
            p.ch[c].base = static_cast<const char*>(sg.ptr) + esz * (size_t)k * sg.stride_c;
            p.ch[c].stride_b = sg.stride_b; p.ch[c].stride_n = sg.stride_n;
            p.ch[c].dtype = sg.dtype; p.ch[c].div255 = sg.div255;
        }
    }
    if (c != expect_channels) return fail(PCRL_E_ARG, "clouds carry %d channels, weights expect %d", c, expect_channels);
    p.C = c;
    if (aug && aug->flags) {
        if ((aug->flags & (PCRL_AUG_JITTER | PCRL_AUG_AFFINE)) && (clouds->seg[0].channels != 3 || clouds->seg[0].dtype != PCRL_DT_F32))
            return fail(PCRL_E_ARG, "augmentation needs segment 0 = xyz (3 x f32)");
        if (aug->n_index_ptr && !(aug->flags & PCRL_AUG_SUBSAMPLE)) return fail(PCRL_E_ARG, "n_index_ptr without SUBSAMPLE");
        if (aug->flags & PCRL_AUG_SUBSAMPLE) {
            if (!aug->point_index || aug->n_index < 1 || aug->n_index > clouds->N)
                return fail(PCRL_E_ARG, "SUBSAMPLE needs point_index and 1 <= n_index <= N (got %d of %d)", aug->n_index, clouds->N);
            p.point_index = aug->point_index;
            p.N = aug->n_index;
            p.n_ptr = aug->n_index_ptr;
        }
        if ((aug->flags & PCRL_AUG_AFFINE) && !aug->affine) return fail(PCRL_E_ARG, "AFFINE without matrix");
        if (aug->flags & PCRL_AUG_COLOR) {
            if (clouds->nseg < 2 || clouds->seg[0].channels != 3 || clouds->seg[1].channels != 3 || clouds->seg[1].dtype != PCRL_DT_U8 || !clouds->seg[1].div255)
                return fail(PCRL_E_ARG, "COLOR needs segment 0 = xyz (3 channels) and segment 1 = rgb (3 x uint8)");
            if (aug->flags & PCRL_AUG_SUBSAMPLE) return fail(PCRL_E_ARG, "COLOR + SUBSAMPLE: the contrast mean would have to be taken over the subsampled cloud");
            bool contrast = false;
            for (int k = 0; k < 4; ++k) contrast = contrast || ((aug->color_order >> (4 * k)) & 15) == PCRL_COLOR_CONTRAST;
            if (contrast && !aug->color_mean) return fail(PCRL_E_ARG, "COLOR with a contrast step needs color_mean (pcrl_color_contrast_mean_u8)");
            p.color_order = aug->color_order; p.color_mean = aug->color_mean;
            for (int k = 0; k < 4; ++k) { p.color_fac[k] = aug->color_factor[k]; p.color_omf[k] = aug->color_one_minus[k]; }
        }
        p.aug_flags = aug->flags; p.jitter_noise = aug->jitter_noise; p.affine = aug->affine;
        p.row_mul = aug->row_mul ? aug->row_mul : 1; p.row_add = aug->row_add;
        p.offset_ptr = reinterpret_cast<const unsigned long long*>(aug->offset_ptr);
        p.jitter_lo = aug->jitter_lo; p.jitter_hi = aug->jitter_hi; p.seed = aug->seed; p.offset = aug->offset;
    }
    return PCRL_OK;
}

static size_t fwd_lds_bytes(int T0, int C1, int C2, int C3, bool bf16, bool split) {
    return fwd_lds_base_bytes(T0, C1, C2, C3, bf16) + sizeof(float) * 32 * (size_t)C1 * fwd_w1_lds_blocks(T0, C1, C2, C3, bf16, split);
}

template <int T0, int C1, int C2, int C3, bool BF16, bool SPLIT = false>
static int launch_fwd(const FwdParams& p, int grid, hipStream_t stream) {
    const size_t lds = fwd_lds_bytes(T0, C1, C2, C3, BF16, SPLIT);
    auto kern = encoder_fwd_kernel<T0, C1, C2, C3, BF16, SPLIT>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
    PCRL_CHECK_LAUNCH("encoder_fwd_kernel");
    return PCRL_OK;
}

// mlp_spec of every shipped pn_* SAC / DrQ config: [64,128,256] (dm_control), [128,128,256] (maniskill), [32,64,128]
// (dm_control pn_motivating / pn_shift_motivating), and the class default [64,128,1024] (pointnet.py:81; used by none of them):
// c3 = 1024 does not fit the one-point-per-lane register layout and runs on encoder_fwd_wide_kernel (fp32 only).
bool encoder_dims_wide(int c1, int c2, int c3) { return c1 == 64 && c2 == 128 && c3 == 1024; }
bool encoder_dims_supported(int c1, int c2, int c3) {
    return ((c1 == 64 || c1 == 128) && c2 == 128 && c3 == 256) || (c1 == 32 && c2 == 64 && c3 == 128) || encoder_dims_wide(c1, c2, c3);
}

template <int T0, int C1, int C2, int C3>
static int launch_fwd_wide(const FwdParams& p, int grid, hipStream_t stream) {
    const size_t lds = sizeof(ChanSrc) * PCRL_MAX_CHANNELS + 8 * (size_t)C3 + sizeof(float) * (2 * C2 + 2 * C3 + C1 + (size_t)(C1 / 32) * T0 * 64);
    auto kern = encoder_fwd_wide_kernel<T0, C1, C2, C3>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
    PCRL_CHECK_LAUNCH("encoder_fwd_wide_kernel");
    return PCRL_OK;
}

}  // namespace pcrl

using namespace pcrl;

extern "C" int pcrl_encoder_packed_bytes(int32_t c_in, int32_t c1, int32_t c2, int32_t c3, size_t* bytes) {
    if (!bytes) return fail(PCRL_E_ARG, "bytes is NULL");
    if (c_in < 1 || c_in > PCRL_MAX_CHANNELS || !encoder_dims_supported(c1, c2, c3))
        return fail(PCRL_E_ARG, "unsupported encoder dims C=%d mlp_spec=[%d,%d,%d] (fused kernel: C<=16, [64|128,128,256], [32,64,128] or [64,128,1024])", c_in, c1, c2, c3);
    const PackedLayout L{(c_in + 1) / 2, c1, c2, c3};
    *bytes = sizeof(float) * (size_t)L.total();
    return PCRL_OK;
}

// min_tiles: tiles a workgroup should at least have.  8 (one per wave) for the bf16 kernel, whose two waves per SIMD overlap vector
// work with each other's MFMAs; 4 (one per SIMD) for the fp32 / split kernels, where the two waves of a SIMD only take turns on the
// matrix pipe -- a 32-cloud launch (a rank's share of K1 at 8 GPUs) then spreads over all 256 CUs with one tile per SIMD instead of
// 128 CUs with two (measured: the actor-phase forward of a 32-cloud step 54.8 -> see DESIGN.md section 4.1).
static void split_plan(int B, int N, int* S, int* tiles_total, int* tiles_per_seg, int min_tiles = 4) {
    const int tiles = (N + 31) / 32, cus = num_cus();
    int s = 1;
    if (B < cus) {
        s = cus / B;
        const int max_s = (tiles + min_tiles - 1) / min_tiles;
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
    }
    int tps = (tiles + s - 1) / s;
    s = (tiles + tps - 1) / tps;
    *S = s; *tiles_total = tiles; *tiles_per_seg = tps;
}

extern "C" int pcrl_encoder_fwd_workspace_bytes(int32_t B, int32_t N, int32_t c3, size_t* bytes) {
    if (!bytes || B < 1 || N < 1 || c3 < 1) return fail(PCRL_E_ARG, "bad arguments");
    int S, tt, tps;
    split_plan(B, N, &S, &tt, &tps);
    *bytes = S > 1 ? (size_t)B * S * c3 * sizeof(unsigned long long) : 0;
    return PCRL_OK;
}

// pcrl_encoder_pack_attach_cols: jobs handed over for the NEXT pack launch of this host thread
static thread_local ColGatherList t_colgather = {};
static thread_local int t_colgather_blocks = 0;

__global__ void col_gather_kernel(const ColGatherList cg) {       // the same jobs as a launch of their own (pcrl_encoder_pack_flush_cols)
    const int blk = (int)blockIdx.x;
    const ColGather& g = (cg.n > 1 && blk >= cg.job[1].blk_begin) ? cg.job[1] : cg.job[0];
    const int per_head = g.ncols * g.rows;
    const int e = (blk - g.blk_begin) * 256 + (int)threadIdx.x;
    if (e < g.heads * per_head) {
        const int h = e / per_head, r = e - h * per_head, j = r / g.rows, row = r - j * g.rows;
        g.dst[e] = g.src[h * g.head_stride + (long long)row * g.ld + g.col0 + j];
    }
}

extern "C" int pcrl_encoder_pack_attach_cols(const pcrl_col_gather* jobs, int32_t n) {
    if (n == 0) { t_colgather.n = 0; t_colgather_blocks = 0; return PCRL_OK; }
    if (!jobs || n < 0 || n > 2) return fail(PCRL_E_ARG, "column gather: 0 <= n <= 2 jobs");
    ColGatherList l{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const pcrl_col_gather& j = jobs[i];
        if (!j.src || !j.dst || j.heads < 1 || j.rows < 1 || j.ncols < 1 || j.col0 < 0 || j.col0 + j.ncols > j.ld)
            return fail(PCRL_E_ARG, "column gather: bad job %d", i);
        l.job[l.n++] = ColGather{j.src, j.head_stride, j.heads, j.rows, j.ld, j.col0, j.ncols, j.dst, blocks};
        blocks += (j.heads * j.ncols * j.rows + 255) / 256;
    }
    t_colgather = l; t_colgather_blocks = blocks;
    return PCRL_OK;
}

extern "C" int pcrl_encoder_pack_flush_cols(void* stream) {
    if (t_colgather.n == 0) return PCRL_OK;
    const ColGatherList l = t_colgather;
    const int blocks = t_colgather_blocks;
    t_colgather.n = 0; t_colgather_blocks = 0;
    hipLaunchKernelGGL(col_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, l);
    PCRL_CHECK_LAUNCH("col_gather_kernel");
    return PCRL_OK;
}

// The pack launch's work as a job (consumes the attached column gathers).
static int make_pack_job(const pcrl_encoder_weights* w, void* packed, size_t packed_bytes, PackJob* job) {
    if (!w || !packed) return fail(PCRL_E_ARG, "NULL argument");
    size_t need;
    if (int rc = pcrl_encoder_packed_bytes(w->c_in, w->c1, w->c2, w->c3, &need)) return rc;
    if (packed_bytes < need) return fail(PCRL_E_WORKSPACE, "packed buffer %zu < %zu bytes", packed_bytes, need);
    const int total = (int)(need / sizeof(float));
    const int extra = t_colgather.n ? t_colgather_blocks : 0;
    *job = PackJob{*w, (w->c_in + 1) / 2, static_cast<float*>(packed), t_colgather, (total + 255) / 256, (total + 255) / 256 + extra};
    t_colgather.n = 0; t_colgather_blocks = 0;       // attached column-gather jobs ride on this job and are consumed by it
    return PCRL_OK;
}

extern "C" int pcrl_encoder_pack_weights_f32(const pcrl_encoder_weights* w, void* packed, size_t packed_bytes, void* stream) {
    PackJob job;
    if (int rc = make_pack_job(w, packed, packed_bytes, &job)) return rc;
    hipLaunchKernelGGL(encoder_pack_kernel, dim3(job.total_blocks), dim3(256), 0, (hipStream_t)stream, job);
    PCRL_CHECK_LAUNCH("encoder_pack_kernel");
    return PCRL_OK;
}

// pcrl_encoder_pack_attach_to_gather: the same job, handed to this host thread's NEXT replay sampling launch
static thread_local PackJob t_pending_pack = {};
static thread_local bool t_pending_pack_on = false;

bool pcrl::take_pending_pack(PackJob* job) {
    if (!t_pending_pack_on) return false;
    *job = t_pending_pack;
    t_pending_pack_on = false;
    return true;
}

extern "C" int pcrl_encoder_pack_attach_to_gather(const pcrl_encoder_weights* w, void* packed, size_t packed_bytes) {
    if (t_pending_pack_on) return fail(PCRL_E_ARG, "a pack job is already pending: flush it first (pcrl_encoder_pack_flush_pending)");
    PackJob job;
    if (int rc = make_pack_job(w, packed, packed_bytes, &job)) return rc;
    t_pending_pack = job; t_pending_pack_on = true;
    return PCRL_OK;
}

extern "C" int pcrl_encoder_pack_drop_pending(void) {
    t_pending_pack_on = false;
    return PCRL_OK;
}

extern "C" int pcrl_encoder_pack_flush_pending(void* stream) {
    PackJob job;
    if (!pcrl::take_pending_pack(&job)) return PCRL_OK;
    hipLaunchKernelGGL(encoder_pack_kernel, dim3(job.total_blocks), dim3(256), 0, (hipStream_t)stream, job);
    PCRL_CHECK_LAUNCH("encoder_pack_kernel");
    return PCRL_OK;
}

static int encoder_fwd_impl(int mode /* 0 fp32, 1 bf16, 2 split */, const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                            const pcrl_encoder_weights* w, const void* packed,
                            float* pooled, int32_t* argmax,
                            void* workspace, size_t workspace_bytes, void* stream, const pcrl_feature_head* head = nullptr) {
    if (!clouds || !w || !packed || !pooled || !argmax) return fail(PCRL_E_ARG, "NULL argument");
    size_t need;
    if (int rc = pcrl_encoder_packed_bytes(w->c_in, w->c1, w->c2, w->c3, &need)) return rc;
    FwdParams p{};
    if (int rc = fill_cloud_params(clouds, aug, w->c_in, &p.cl)) return rc;
    if (p.cl.B == 0) return PCRL_OK;
    constexpr int min_tiles_f32 = 4;          // one tile per SIMD (2 measured worse, 8 is the bf16 kernel's: DESIGN.md section 4.1)
    split_plan(p.cl.B, p.cl.N, &p.S, &p.tiles_total, &p.tiles_per_seg, mode == 1 ? 8 : min_tiles_f32);
    if (p.S > 1) {
        const size_t ws = (size_t)p.cl.B * p.S * w->c3 * sizeof(unsigned long long);
        if (!workspace || workspace_bytes < ws) return fail(PCRL_E_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws);
        p.partial = static_cast<unsigned long long*>(workspace);
    }
    p.eps = w->eps; p.packed = static_cast<const float*>(packed); p.pooled = pooled; p.argmax = argmax;
    if (head) {
        if (!head->weight || !head->bias || !head->gamma || !head->beta) return fail(PCRL_E_ARG, "feature head: NULL parameter");
        if (head->F < 1 || head->F > 256 || head->F > w->c3 || head->n_ranges < 1 || head->n_ranges > 2)
            return fail(PCRL_E_ARG, "feature head: 1 <= F <= min(256, c3), 1 <= n_ranges <= 2");
        HeadParams& h = p.head;
        h.weight = head->weight; h.bias = head->bias; h.gamma = head->gamma; h.beta = head->beta; h.F = head->F; h.eps = head->eps;
        h.n_ranges = head->n_ranges;
        for (int i = 0; i < head->n_ranges; ++i) {
            const pcrl_ln_job& s = head->job[i];
            HeadRange& d = h.range[i];
            if (s.M < 0 || s.n_dst < 0 || s.n_dst > 4 || head->begin[i] < 0 || head->begin[i] + s.M > p.cl.B)
                return fail(PCRL_E_ARG, "feature head: bad range %d", i);
            if (i == 1 && head->begin[1] < head->begin[0] + head->job[0].M) return fail(PCRL_E_ARG, "feature head: ranges must be ascending and disjoint");
            d.begin = head->begin[i]; d.count = s.M; d.n_dst = s.n_dst; d.xhat = s.xhat; d.rstd = s.rstd;
            for (int k = 0; k < s.n_dst; ++k) { if (!s.dst[k]) return fail(PCRL_E_ARG, "NULL destination"); d.y[k] = s.dst[k]; d.ldy[k] = s.ld_dst[k]; }
            for (int c = 0; c < 2; ++c) {
                if (s.cat_n[c] > 0 && (!s.cat_src[c] || !s.cat_dst[c])) return fail(PCRL_E_ARG, "NULL pass-through columns");
                d.cat_src[c] = s.cat_n[c] > 0 ? s.cat_src[c] : nullptr; d.cat_dst[c] = s.cat_dst[c];
                d.cat_lds[c] = s.cat_ld_src[c]; d.cat_ldd[c] = s.cat_ld_dst[c]; d.cat_n[c] = s.cat_n[c];
                d.cat_div[c] = s.cat_row_div[c] > 1 ? s.cat_row_div[c] : 1;
            }
        }
    }

    const int grid = min(p.cl.B * p.S, num_cus());
    const int T0 = (p.cl.C + 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    int rc = PCRL_E_ARG;
    if (encoder_dims_wide(w->c1, w->c2, w->c3)) {
        if (mode != 0) return fail(PCRL_E_ARG, "mlp_spec=[%d,%d,%d]: only the fp32 kernel is built for the wide last layer", w->c1, w->c2, w->c3);
        if (head) return fail(PCRL_E_ARG, "mlp_spec=[%d,%d,%d]: the feature-head epilogue is built for c3 <= 256", w->c1, w->c2, w->c3);
#define PCRL_FWD_WIDE_CASE(T0_) if (T0 == T0_) rc = launch_fwd_wide<T0_, 64, 128, 1024>(p, grid, st);
        PCRL_FWD_WIDE_CASE(2) PCRL_FWD_WIDE_CASE(3) PCRL_FWD_WIDE_CASE(4) PCRL_FWD_WIDE_CASE(5)
#undef PCRL_FWD_WIDE_CASE
    }
#define PCRL_FWD_CASE(T0_, C1_, C2_, C3_)                                         \
    if (T0 == T0_ && w->c1 == C1_ && w->c2 == C2_ && w->c3 == C3_)                \
        rc = mode == 1 ? launch_fwd<T0_, C1_, C2_, C3_, true>(p, grid, st) : mode == 2 ? launch_fwd<T0_, C1_, C2_, C3_, false, true>(p, grid, st) \
                       : launch_fwd<T0_, C1_, C2_, C3_, false>(p, grid, st);
    PCRL_FWD_CASE(2, 64, 128, 256) PCRL_FWD_CASE(3, 64, 128, 256) PCRL_FWD_CASE(4, 64, 128, 256) PCRL_FWD_CASE(5, 64, 128, 256)
    PCRL_FWD_CASE(2, 128, 128, 256) PCRL_FWD_CASE(3, 128, 128, 256) PCRL_FWD_CASE(4, 128, 128, 256) PCRL_FWD_CASE(5, 128, 128, 256)
    PCRL_FWD_CASE(2, 32, 64, 128) PCRL_FWD_CASE(3, 32, 64, 128) PCRL_FWD_CASE(4, 32, 64, 128) PCRL_FWD_CASE(5, 32, 64, 128)
#undef PCRL_FWD_CASE
    if (rc == PCRL_E_ARG) return fail(PCRL_E_ARG, "no fused kernel for C=%d (supported: 3..10 channels)", p.cl.C);
    if (rc) return rc;
    if (p.S > 1 && head) {
        if (p.head.weight) hipLaunchKernelGGL(encoder_merge_head_kernel<1024>, dim3(p.cl.B), dim3(1024), 0, st, p.partial, p.S, w->c3, pooled, argmax, p.head);
        else hipLaunchKernelGGL(encoder_merge_head_kernel<256>, dim3(p.cl.B), dim3(256), 0, st, p.partial, p.S, w->c3, pooled, argmax, p.head);
        PCRL_CHECK_LAUNCH("encoder_merge_head_kernel");
    } else if (p.S > 1) {
        const long long n = (long long)p.cl.B * w->c3;
        hipLaunchKernelGGL(encoder_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           p.partial, p.cl.B, p.S, w->c3, pooled, argmax);
        PCRL_CHECK_LAUNCH("encoder_merge_kernel");
    }
    return PCRL_OK;
}

#define PCRL_FWD_HEAD_ENTRY(NAME, MODE)                                                                                             \
    extern "C" int NAME(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug, const pcrl_encoder_weights* w, const void* packed,  \
                        float* pooled, int32_t* argmax, const pcrl_feature_head* head, void* workspace, size_t workspace_bytes,       \
                        void* stream) {                                                                                               \
        if (!head) return fail(PCRL_E_ARG, "NULL argument");                                                                          \
        return encoder_fwd_impl(MODE, clouds, aug, w, packed, pooled, argmax, workspace, workspace_bytes, stream, head);              \
    }
PCRL_FWD_HEAD_ENTRY(pcrl_encoder_fwd_head_f32, 0)
PCRL_FWD_HEAD_ENTRY(pcrl_encoder_fwd_head_bf16, 1)
PCRL_FWD_HEAD_ENTRY(pcrl_encoder_fwd_head_f32split, 2)
#undef PCRL_FWD_HEAD_ENTRY

extern "C" int pcrl_encoder_fwd_f32(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                    const pcrl_encoder_weights* w, const void* packed,
                                    float* pooled, int32_t* argmax,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_fwd_impl(0, clouds, aug, w, packed, pooled, argmax, workspace, workspace_bytes, stream);
}

extern "C" int pcrl_encoder_fwd_bf16(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                     const pcrl_encoder_weights* w, const void* packed,
                                     float* pooled, int32_t* argmax,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_fwd_impl(1, clouds, aug, w, packed, pooled, argmax, workspace, workspace_bytes, stream);
}

extern "C" int pcrl_encoder_fwd_f32split(const pcrl_cloud_desc* clouds, const pcrl_aug_desc* aug,
                                         const pcrl_encoder_weights* w, const void* packed,
                                         float* pooled, int32_t* argmax,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_fwd_impl(2, clouds, aug, w, packed, pooled, argmax, workspace, workspace_bytes, stream);
}

#ifdef PCRL_FWD_STAMPS
extern "C" int pcrl_debug_fwd_stamps(unsigned long long* out, int n_rows) {
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pcrl::g_fwd_stamps), sizeof(unsigned long long) * 8 * (size_t)n_rows) != hipSuccess) return -3;
    return 0;
}
#endif
