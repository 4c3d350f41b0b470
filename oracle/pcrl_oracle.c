/*
 * pcrl_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement (plain C, gcc) of the arithmetic on the point-cloud
 * actor-critic hot path of lz1oceani/pointcloud_rl.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product (pointcloud_rl_amd/) never does.
 *
 * Parity status: PINNED by golden vectors produced by importing and running
 * the reference Python itself in the build container
 * (tools/gen_golden.py -> tests/golden/ *.npz, checked by
 * tests/test_oracle_golden.py).  The reference ships no tests of its own for
 * this path.
 *
 * Every function cites the reference lines it restates (paths relative to
 * the reference checkout).
 *
 * Floating-point order.  The reference leaves summation order to ATen; this
 * file fixes ONE order ("canonical order") so that the HIP kernels can be
 * compared bit-for-bit, not just within tolerance:
 *   - a dense layer output is a single fused-multiply-add chain over the
 *     input channels in the order pi() below (the order in which a CDNA
 *     32x32x2 f32 MFMA accumulator tile, re-used as the next layer's B
 *     operand, walks the channels);
 *   - a LayerNorm moment is two half-sums (channels whose (c & 4) == 0, and
 *     the rest), each the pairwise combination of 4 interleaved running sums.
 * Compile with -ffp-contract=off: every fused operation is written fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PCRL_ORACLE_API __attribute__((visibility("default")))

/* channel visited at position k (0 <= k < C, C % 32 == 0) of the canonical chain */
static inline int pi_chan(int k) {
    int t = k >> 1, h = k & 1;          /* MFMA k-step t, k-lane h          */
    int blk = t >> 4, r = t & 15;       /* 32-channel block, accumulator reg */
    return 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * h;
}

/* channel held by accumulator slot R (0 <= R < C/2) of half h */
static inline int acc_chan(int R, int h) {
    int blk = R >> 4, r = R & 15;
    return 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * h;
}

/* ReLU as torch.relu / nn.ReLU (block_utils.py:92-96): NaN propagates, -0 -> +0 */
static inline float relu_f(float x) { return !(x <= 0.0f) ? x : 0.0f; }

/* Half-sum over the C/2 channels of half h, 4 interleaved partials combined pairwise. */
static float half_sum(const float* x, int C, int h) {
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    for (int R = 0; R < C / 2; ++R) p[R & 3] = p[R & 3] + x[acc_chan(R, h)];
    return (p[0] + p[1]) + (p[2] + p[3]);
}
static float half_sumsq_centered(const float* x, float mean, int C, int h) {
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    for (int R = 0; R < C / 2; ++R) {
        float d = x[acc_chan(R, h)] - mean;
        p[R & 3] = fmaf(d, d, p[R & 3]);
    }
    return (p[0] + p[1]) + (p[2] + p[3]);
}

/*
 * Per-point LayerNorm over channels followed by ReLU, in place.
 * Restates LayerNormkD.forward (pyrl/networks/modules/nn_layer.py:207-219:
 * permute -> F.layer_norm(biased variance, eps inside the sqrt, affine) ->
 * permute) + ReLU (block_utils.py:92-96).  Returns 1 when the variance is NaN
 * (then every output of the point is NaN, as in torch).
 */
static int ln_relu_point(float* x, int C, const float* gamma, const float* beta, float eps,
                         float* mean_out, float* rstd_out, float* pre /* optional [C]: the value ReLU decides on */) {
    float mean = (half_sum(x, C, 0) + half_sum(x, C, 1)) / (float)C;
    float var = (half_sumsq_centered(x, mean, C, 0) + half_sumsq_centered(x, mean, C, 1)) / (float)C;
    float rstd = 1.0f / sqrtf(var + eps);
    for (int c = 0; c < C; ++c) {
        float d = x[c] - mean;
        float y = fmaf(d * rstd, gamma[c], beta[c]);
        if (pre) pre[c] = y;
        x[c] = relu_f(y);
    }
    if (mean_out) *mean_out = mean;
    if (rstd_out) *rstd_out = rstd;
    return var != var;
}

/*
 * Shared per-point MLP of the PointNet encoder for ONE point.
 * Restates ConvMLP as built for PointNet (pyrl/networks/backbones/mlp.py:43-56,
 * 103-108; pointnet.py:106-109 with ignore_first_ln=True, inactivated_output=False):
 *   conv0 (bias) + ReLU ; conv1 (no bias) + LN1d + ReLU ; conv2 (no bias) + LN1d + ReLU.
 * x: [C] input features (C may be odd; the chain treats a missing channel as 0).
 * h0:[c1] h1:[c2] h2:[c3] outputs (post-activation).  z1/z2 (optional): pre-LN values.
 */
static int mlp_point(const float* x, int C, int c1, int c2, int c3,
                     const float* w0, const float* b0,
                     const float* w1, const float* g1, const float* be1,
                     const float* w2, const float* g2, const float* be2, float eps,
                     float* h0, float* h1, float* h2,
                     float* z1, float* z2, float* stats /* mean1,rstd1,mean2,rstd2 or NULL */,
                     float* pre0, float* pre1, float* pre2 /* optional: what each layer's ReLU decides on */) {
    int Cp = (C + 1) & ~1;
    for (int o = 0; o < c1; ++o) {
        float a = b0[o];
        for (int k = 0; k < Cp; ++k) {
            float w = k < C ? w0[o * C + k] : 0.0f, xv = k < C ? x[k] : 0.0f;
            a = fmaf(w, xv, a);
        }
        if (pre0) pre0[o] = a;
        h0[o] = relu_f(a);
    }
    for (int o = 0; o < c2; ++o) {
        float a = 0.0f;
        for (int k = 0; k < c1; ++k) { int c = pi_chan(k); a = fmaf(w1[o * c1 + c], h0[c], a); }
        h1[o] = a;
    }
    if (z1) memcpy(z1, h1, sizeof(float) * c2);
    ln_relu_point(h1, c2, g1, be1, eps, stats ? stats + 0 : NULL, stats ? stats + 1 : NULL, pre1);
    for (int o = 0; o < c3; ++o) {
        float a = 0.0f;
        for (int k = 0; k < c2; ++k) { int c = pi_chan(k); a = fmaf(w2[o * c2 + c], h1[c], a); }
        h2[o] = a;
    }
    if (z2) memcpy(z2, h2, sizeof(float) * c3);
    int nan_pt = ln_relu_point(h2, c3, g2, be2, eps, stats ? stats + 2 : NULL, stats ? stats + 3 : NULL, pre2);
    if (nan_pt) for (int o = 0; o < c3; ++o) h2[o] = NAN;
    return nan_pt;
}

/*
 * PointNet encoder forward up to and including the symmetric max-pool.
 * Restates PointNet.forward (pyrl/networks/backbones/pointnet.py:148-151):
 *   feature = self.conv(feature); feature = feature.max(-1)[0]
 * plus the int64 argmax torch keeps for autograd (returned here as int32).
 * torch CPU max(dim) semantics: first index among equal values; a NaN wins
 * over every number and the first NaN's index is returned.
 *
 * feat   [B][C][N]  f32, the tensor PointCloudBase.preprocess builds (pointnet.py:49-73)
 * pooled [B][c3], argmax [B][c3]; prepool (optional) [B][c3][N].
 */
PCRL_ORACLE_API int pcrl_oracle_encoder_fwd_f32(
    const float* feat, int B, int C, int N, int c1, int c2, int c3,
    const float* w0, const float* b0, const float* w1, const float* g1, const float* be1,
    const float* w2, const float* g2, const float* be2, float eps,
    float* pooled, int32_t* argmax, float* prepool) {
    if (c1 % 32 || c2 % 32 || c3 % 32 || C < 1 || N < 1) return -1;
    float* x = (float*)malloc(sizeof(float) * (C + c1 + c2 + c3));
    float *h0 = x + C, *h1 = h0 + c1, *h2 = h1 + c2;
    for (int b = 0; b < B; ++b) {
        float* best = pooled + (size_t)b * c3;
        int32_t* bi = argmax + (size_t)b * c3;
        for (int n = 0; n < N; ++n) {
            for (int c = 0; c < C; ++c) x[c] = feat[((size_t)b * C + c) * N + n];
            mlp_point(x, C, c1, c2, c3, w0, b0, w1, g1, be1, w2, g2, be2, eps, h0, h1, h2, NULL, NULL, NULL, NULL, NULL, NULL);
            for (int o = 0; o < c3; ++o) {
                float v = h2[o];
                if (prepool) prepool[((size_t)b * c3 + o) * N + n] = v;
                if (n == 0) { best[o] = v; bi[o] = 0; }
                else if (!(best[o] != best[o]) && (v > best[o] || v != v)) { best[o] = v; bi[o] = n; }
            }
        }
    }
    free(x);
    return 0;
}

/*
 * The values the three ReLUs of the per-point MLP decide on (conv0 + bias; LayerNorm-1 and LayerNorm-2 outputs), for
 * n_pts given points x [n_pts][C], in the canonical order above -- i.e. the branch decisions the HIP forward kernel takes
 * (it is bit-identical to this file).  tests/test_fullsize_parity_gpu.py confirms every located "encoder event" with it:
 * a ReLU decision of the ATen restatement counts as flipped by the summation order only if THIS order decides otherwise.
 * Restates the same lines as mlp_point (mlp.py:43-56; nn_layer.py:207-219; block_utils.py:92-96).
 */
PCRL_ORACLE_API int pcrl_oracle_point_preacts_f32(
    const float* x, int n_pts, int C, int c1, int c2, int c3,
    const float* w0, const float* b0, const float* w1, const float* g1, const float* be1,
    const float* w2, const float* g2, const float* be2, float eps,
    float* pre0 /* [n_pts][c1] */, float* pre1 /* [n_pts][c2] */, float* pre2 /* [n_pts][c3] */) {
    if (c1 % 32 || c2 % 32 || c3 % 32 || C < 1 || n_pts < 0) return -1;
    float* h0 = (float*)malloc(sizeof(float) * (c1 + c2 + c3));
    float *h1 = h0 + c1, *h2 = h1 + c2;
    for (int i = 0; i < n_pts; ++i)
        mlp_point(x + (size_t)i * C, C, c1, c2, c3, w0, b0, w1, g1, be1, w2, g2, be2, eps, h0, h1, h2, NULL, NULL, NULL,
                  pre0 + (size_t)i * c1, pre1 + (size_t)i * c2, pre2 + (size_t)i * c3);
    free(h0);
    return 0;
}

/*
 * Stand-alone symmetric max-pool with first-index argmax over a materialised
 * [B][c][N] tensor (pointnet.py:151, `feature.max(-1)`), torch CPU tie/NaN rules.
 */
PCRL_ORACLE_API int pcrl_oracle_segmax_f32(const float* x, int B, int c, int N, float* out, int32_t* idx) {
    for (size_t r = 0; r < (size_t)B * c; ++r) {
        const float* row = x + r * N;
        float best = row[0]; int32_t bi = 0;
        for (int n = 1; n < N; ++n) {
            float v = row[n];
            if (!(best != best) && (v > best || v != v)) { best = v; bi = n; }
        }
        out[r] = best; idx[r] = bi;
    }
    return 0;
}
