// Head "tails" for gfx950: the last Linear of a head fused with what follows it, so that the update step does not pay a
// dependent launch for every 5-10 us kernel that moves a few MFLOP.
//
//   q_tail (critic)  Q heads' last layer for the target heads on s' and the online heads on s (LinearMLP's final Linear,
//                    pyrl/networks/backbones/mlp.py:97-100, n_out = 1), the TD target + critic loss (sac.py:125-157,
//                    drq.py:76-103) and the first backward stage of the online heads: dh2 = dq w2 (.) [h2 > 0], and
//                    per-workgroup partial sums of dW2 = dq^T h2, db2 = sum dq and of the logged statistics.
//                    Replaces three launches (a GEMM group, critic_loss_kernel, a GEMM group).
//   q_tail (actor)   Q heads' last layer on (s, pi(s)), d(-mean min_h q)/dq and dh2 (sac.py:177-183); partial sums of
//                    min_h q and of -log pi for actor_finalize.  Replaces a GEMM, actor_loss_kernel's per-row part, a GEMM.
//   actor_finalize   actor / temperature losses and d(alpha_loss)/d(log_alpha) from those partial sums (sac.py:183-195).
//   policy_tail_fwd  the policy's last layer (n_out = 2A) + TanhGaussianHead mode "max-entropy" (gaussian.py:83-87,
//                    distributions.py:89,116-127).  Replaces a GEMM + tanh_gaussian_fwd_kernel.
//   colsum_jobs      fixed-order column sums (or maxima) of per-workgroup partials for up to 12 jobs in one launch: the
//                    reductions the kernels above and layernorm_rows_bwd leave behind.
//
// Layout: a wave owns a row of the batch; a lane owns 4 consecutive columns of every 256-column chunk of the hidden vector
// (H % 256 == 0), so a row is read as H / 256 coalesced 1 KB pieces.  Reductions over lanes are butterflies in a fixed
// order, over waves / workgroups sums in index order: results are bit-reproducible run to run.
#include "common.h"

namespace pcrl {
// The row-split tail kernels (a row's hidden vector over the four waves of a workgroup) serve H = 1024 up to this many rows
// (methods/fused.py: FusedStep.tail_split_rows is the same number).
constexpr int kTailSplitMaxRows = 512;


constexpr int kTailMaxChunks = 8;                  // H <= 2048
constexpr float kTailHalfLog2Pi = 0.91893853320467274178f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int NC>
__device__ __forceinline__ void tail_load_row(const float* row, int lane, f32x4 (&v)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = *reinterpret_cast<const f32x4*>(row + 256 * c + 4 * lane);
}

template <int NC>
__device__ __forceinline__ float tail_dot(const f32x4 (&a)[NC], const f32x4 (&b)[NC]) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_fmaf(a[c][e], b[c][e], s);
    return wave_sum(s);
}

struct QTailParams {
    int mode;                                      // 0: critic, 1: actor
    const float* h2; long long h2_hs;              // online heads' last hidden activations [2][M][H] (head stride h2_hs)
    const float* w2; const float* b2; long long w_hs;   // online last layer: w2 + h * w_hs [H], b2 + h * w_hs [1]
    const float* h2_t; long long h2_t_hs;          // critic: target heads on s'
    const float* w2_t; const float* b2_t; long long w_t_hs;
    const float* nlp;                              // critic: -log pi(a'|s') [M]; actor: -log pi(a|s) [M]
    const float* rewards; const unsigned char* dones; int rd_div;
    const float* log_alpha; float gamma, reward_scale; int ignore_dones, group;
    int M, H;
    float* q; long long ld_q;                      // [M][2] (critic: Q(s,a); actor: Q(s,pi))
    float* q_target;                               // critic [M]
    float* dq; long long ld_dq;                    // [M][2]
    float* dh2; long long dh2_hs;                  // [2][M][H]
    float* part; int part_ld;                      // critic: [n_wg][2][part_ld]: dW2 partial (cols < H), db2 partial (col H)
    float* stat_part;                              // [n_wg][4]: critic {sum d^2, max |d|, sum min_h q, sum y}; actor {sum min_h q, sum nlp, -, -}
    float* d_neglogp;                              // actor: device scalar = -alpha / M
    // extra workgroups behind the row blocks (actor mode, optional): columns [cg_col0, cg_col0 + cg_ncols) of the two heads'
    // FIRST layer weight w0 [2][H][cg_ld] written as a compact image cg_dst [2][cg_ncols][H] -- the action columns, which
    // policy_tail_bwd_kernel contracts with dh1 (coalesced 16-byte rows instead of 4-byte loads cg_ld floats apart)
    const float* cg_w0; long long cg_hs; int cg_ld, cg_col0, cg_ncols; float* cg_dst;
};

template <int NC>
__global__ __launch_bounds__(256) void q_tail_kernel(const QTailParams p) {
    __shared__ float s_y[4];
    __shared__ float s_st[4][4];
    __shared__ float s_db[4][2];
    extern __shared__ __attribute__((aligned(16))) float s_dw[];          // critic: [4 waves][2][H]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_row_blocks = (p.M + 3) >> 2;
    if ((int)blockIdx.x >= n_row_blocks) {         // column-gather workgroups: one element per thread, coalesced stores
        const int per_head = p.cg_ncols * p.H;
        const int i = ((int)blockIdx.x - n_row_blocks) * 256 + (int)threadIdx.x;
        if (i < 2 * per_head) {
            const int h = i / per_head, r = i - h * per_head, j = r / p.H, col = r - j * p.H;
            p.cg_dst[i] = p.cg_w0[h * p.cg_hs + (long long)col * p.cg_ld + p.cg_col0 + j];
        }
        return;
    }
    const int m = blockIdx.x * 4 + wave;
    const bool live = m < p.M;
    const int mr = live ? m : p.M - 1;
    const float alpha = expf(p.log_alpha[0]);

    f32x4 hv[2][NC], wv[2][NC];
    float q[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        tail_load_row<NC>(p.h2 + h * p.h2_hs + (long long)mr * p.H, lane, hv[h]);
        tail_load_row<NC>(p.w2 + h * p.w_hs, lane, wv[h]);
        q[h] = tail_dot<NC>(hv[h], wv[h]) + p.b2[h * p.w_hs];
    }
    if (live && lane < 2) p.q[(long long)m * p.ld_q + lane] = q[lane];

    float dqv[2];
    if (p.mode == 0) {
        // ---- TD target (sac.py:125-134; drq.py:76-87 with the mean over `group` consecutive rows) ----
        float qn[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 tv[NC], tw[NC];
            tail_load_row<NC>(p.h2_t + h * p.h2_t_hs + (long long)mr * p.H, lane, tv);
            tail_load_row<NC>(p.w2_t + h * p.w_t_hs, lane, tw);
            qn[h] = tail_dot<NC>(tv, tw) + p.b2_t[h * p.w_t_hs];
        }
        const float mn = fminf(qn[0], qn[1]) + alpha * p.nlp[mr];
        const int e = mr / p.rd_div;
        const float r = p.rewards[e] * p.reward_scale;
        const float y = p.ignore_dones ? r + p.gamma * mn : r + (1.0f - (p.dones[e] ? 1.0f : 0.0f)) * p.gamma * mn;
        if (lane == 0) s_y[wave] = y;
        __syncthreads();
        float ybar = 0.0f;
        const int g0 = (wave / p.group) * p.group;              // 4 % group == 0: a group never straddles workgroups
        for (int a = 0; a < p.group; ++a) ybar += s_y[g0 + a];
        ybar = p.group > 1 ? ybar / (float)p.group : ybar;
        const float d0 = q[0] - ybar, d1 = q[1] - ybar;
        dqv[0] = 2.0f * d0 / (float)p.M; dqv[1] = 2.0f * d1 / (float)p.M;      // d/dq of mean_{b,h}(d^2) * H
        if (live && lane == 0) {
            p.q_target[m] = ybar;
            p.dq[(long long)m * p.ld_dq] = dqv[0]; p.dq[(long long)m * p.ld_dq + 1] = dqv[1];
        }
        if (lane == 0) {
            s_st[wave][0] = live ? d0 * d0 + d1 * d1 : 0.0f;
            s_st[wave][1] = live ? fmaxf(fabsf(d0), fabsf(d1)) : 0.0f;
            s_st[wave][2] = live ? fminf(q[0], q[1]) : 0.0f;
            s_st[wave][3] = live ? ybar : 0.0f;
            s_db[wave][0] = live ? dqv[0] : 0.0f; s_db[wave][1] = live ? dqv[1] : 0.0f;
        }
    } else {
        // ---- d(-mean_b min_h q)/dq: -1/M on the first minimal head (torch.min's index), sac.py:177-183 ----
        const int arg = q[1] < q[0] ? 1 : 0;
        dqv[0] = arg == 0 ? -1.0f / (float)p.M : 0.0f;
        dqv[1] = arg == 1 ? -1.0f / (float)p.M : 0.0f;
        if (live && lane == 0) { p.dq[(long long)m * p.ld_dq] = dqv[0]; p.dq[(long long)m * p.ld_dq + 1] = dqv[1]; }
        if (lane == 0) {
            s_st[wave][0] = live ? q[arg] : 0.0f;
            s_st[wave][1] = live ? p.nlp[mr] : 0.0f;
            s_st[wave][2] = 0.0f; s_st[wave][3] = 0.0f;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) p.d_neglogp[0] = -alpha / (float)p.M;
    }
    // ---- first backward stage of the online heads: dh2 = dq w2 (.) [h2 > 0] (the ReLU of the layer that produced h2) ----
    if (live) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float* dst = p.dh2 + h * p.dh2_hs + (long long)m * p.H;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = hv[h][c][e] > 0.0f ? dqv[h] * wv[h][c][e] : 0.0f;
                *reinterpret_cast<f32x4*>(dst + 256 * c + 4 * lane) = o;
            }
        }
    }
    if (p.mode == 0) {
        // this workgroup's share of dW2[h][:] = sum_rows dq[m][h] h2[h][m][:]: the four rows are added in wave order
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = live ? dqv[h] * hv[h][c][e] : 0.0f;
                *reinterpret_cast<f32x4*>(s_dw + (wave * 2 + h) * p.H + 256 * c + 4 * lane) = o;
            }
    }
    __syncthreads();
    if (p.mode == 0) {
        for (int i = threadIdx.x; i < 2 * p.H; i += 256) {
            const int h = i / p.H, col = i - h * p.H;
            const float v = ((s_dw[(0 * 2 + h) * p.H + col] + s_dw[(1 * 2 + h) * p.H + col]) + s_dw[(2 * 2 + h) * p.H + col]) + s_dw[(3 * 2 + h) * p.H + col];
            p.part[((long long)blockIdx.x * 2 + h) * p.part_ld + col] = v;
        }
        if (threadIdx.x < 2)
            p.part[((long long)blockIdx.x * 2 + threadIdx.x) * p.part_ld + p.H] =
                ((s_db[0][threadIdx.x] + s_db[1][threadIdx.x]) + s_db[2][threadIdx.x]) + s_db[3][threadIdx.x];
    }
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        float v;
        if (p.mode == 0 && k == 1) v = fmaxf(fmaxf(s_st[0][1], s_st[1][1]), fmaxf(s_st[2][1], s_st[3][1]));
        else v = ((s_st[0][k] + s_st[1][k]) + s_st[2][k]) + s_st[3][k];
        p.stat_part[(long long)blockIdx.x * 4 + k] = v;
    }
}

// q_tail_kernel for H = 1024 and small batches: the same four rows per workgroup (so the partial-sum layout is unchanged), but
// SIXTEEN waves -- a row's hidden vector is split over four of them (one 1 KB piece of every operand row per wave instead of four:
// the dependent-load chain is 4x shorter; at M = 32 the four-wave kernel took 13-14 us for eight workgroups).  The quarter dot
// products meet in LDS and are added in quarter order.
__global__ __launch_bounds__(1024) void q_tail_split_kernel(const QTailParams p) {
    __shared__ float s_dot[4][4][4];               // [row][quarter][q0, q1, qn0, qn1]
    __shared__ float s_y[4];
    __shared__ float s_st[4][4];
    __shared__ float s_db[4][2];
    extern __shared__ __attribute__((aligned(16))) float s_dw[];          // critic: [4 rows][2][H]
    const int w16 = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_row_blocks = (p.M + 3) >> 2;
    if ((int)blockIdx.x >= n_row_blocks) {         // column-gather workgroups (see q_tail_kernel)
        const int per_head = p.cg_ncols * p.H;
        const int i = ((int)blockIdx.x - n_row_blocks) * 1024 + (int)threadIdx.x;
        if (i < 2 * per_head) {
            const int h = i / per_head, r = i - h * per_head, j = r / p.H, col = r - j * p.H;
            p.cg_dst[i] = p.cg_w0[h * p.cg_hs + (long long)col * p.cg_ld + p.cg_col0 + j];
        }
        return;
    }
    const int row = w16 >> 2, qw = w16 & 3;
    const int m = blockIdx.x * 4 + row;
    const bool live = m < p.M;
    const int mr = live ? m : p.M - 1;
    const int col = 256 * qw + 4 * lane;
    // every scalar the row needs after the barriers is requested now, with the operand pieces (behind a barrier each would be a
    // round trip of its own: nlp was written by the previous launch, on another XCD)
    const float log_alpha_v = p.log_alpha[0];
    const float b2v[2] = {p.b2[0], p.b2[p.w_hs]};
    float b2tv[2] = {0.0f, 0.0f}, nlp_v = 0.0f, rew_v = 0.0f;
    bool done_v = false;
    if (p.mode == 0) {
        b2tv[0] = p.b2_t[0]; b2tv[1] = p.b2_t[p.w_t_hs];
        rew_v = p.rewards[mr / p.rd_div];
        done_v = !p.ignore_dones && p.dones[mr / p.rd_div] != 0;
    }
    nlp_v = p.nlp[mr];
    const float alpha = expf(log_alpha_v);
    f32x4 hv[2], wv[2];
    {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.0f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            hv[h] = *reinterpret_cast<const f32x4*>(p.h2 + h * p.h2_hs + (long long)mr * p.H + col);
            wv[h] = *reinterpret_cast<const f32x4*>(p.w2 + h * p.w_hs + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[h] = __builtin_fmaf(hv[h][e], wv[h][e], v[h]);
        }
        if (p.mode == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 tv = *reinterpret_cast<const f32x4*>(p.h2_t + h * p.h2_t_hs + (long long)mr * p.H + col);
                const f32x4 tw = *reinterpret_cast<const f32x4*>(p.w2_t + h * p.w_t_hs + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[2 + h] = __builtin_fmaf(tv[e], tw[e], v[2 + h]);
            }
        }
        allreduce_add32_x16(v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float lo, hi;
            both_halves(v[k], lo, hi);
            if (lane == 0) s_dot[row][qw][k] = lo + hi;
        }
    }
    __syncthreads();
    float q[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) q[h] = (((s_dot[row][0][h] + s_dot[row][1][h]) + s_dot[row][2][h]) + s_dot[row][3][h]) + b2v[h];
    const bool first = qw == 0 && lane == 0;       // one lane per row reports
    if (live && qw == 0 && lane < 2) p.q[(long long)m * p.ld_q + lane] = q[lane];
    float dqv[2];
    if (p.mode == 0) {
        float qn[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
            qn[h] = (((s_dot[row][0][2 + h] + s_dot[row][1][2 + h]) + s_dot[row][2][2 + h]) + s_dot[row][3][2 + h]) + b2tv[h];
        const float mn = fminf(qn[0], qn[1]) + alpha * nlp_v;
        const float r = rew_v * p.reward_scale;
        const float y = p.ignore_dones ? r + p.gamma * mn : r + (1.0f - (done_v ? 1.0f : 0.0f)) * p.gamma * mn;
        if (first) s_y[row] = y;
        __syncthreads();
        float ybar = 0.0f;
        const int g0 = (row / p.group) * p.group;
        for (int a = 0; a < p.group; ++a) ybar += s_y[g0 + a];
        ybar = p.group > 1 ? ybar / (float)p.group : ybar;
        const float d0 = q[0] - ybar, d1 = q[1] - ybar;
        dqv[0] = 2.0f * d0 / (float)p.M; dqv[1] = 2.0f * d1 / (float)p.M;
        if (live && first) {
            p.q_target[m] = ybar;
            p.dq[(long long)m * p.ld_dq] = dqv[0]; p.dq[(long long)m * p.ld_dq + 1] = dqv[1];
        }
        if (first) {
            s_st[row][0] = live ? d0 * d0 + d1 * d1 : 0.0f;
            s_st[row][1] = live ? fmaxf(fabsf(d0), fabsf(d1)) : 0.0f;
            s_st[row][2] = live ? fminf(q[0], q[1]) : 0.0f;
            s_st[row][3] = live ? ybar : 0.0f;
            s_db[row][0] = live ? dqv[0] : 0.0f; s_db[row][1] = live ? dqv[1] : 0.0f;
        }
    } else {
        const int arg = q[1] < q[0] ? 1 : 0;
        dqv[0] = arg == 0 ? -1.0f / (float)p.M : 0.0f;
        dqv[1] = arg == 1 ? -1.0f / (float)p.M : 0.0f;
        if (live && first) { p.dq[(long long)m * p.ld_dq] = dqv[0]; p.dq[(long long)m * p.ld_dq + 1] = dqv[1]; }
        if (first) {
            s_st[row][0] = live ? q[arg] : 0.0f;
            s_st[row][1] = live ? nlp_v : 0.0f;
            s_st[row][2] = 0.0f; s_st[row][3] = 0.0f;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) p.d_neglogp[0] = -alpha / (float)p.M;
    }
    if (live) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = hv[h][e] > 0.0f ? dqv[h] * wv[h][e] : 0.0f;
            *reinterpret_cast<f32x4*>(p.dh2 + h * p.dh2_hs + (long long)m * p.H + col) = o;
        }
    }
    if (p.mode == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = live ? dqv[h] * hv[h][e] : 0.0f;
            *reinterpret_cast<f32x4*>(s_dw + (row * 2 + h) * p.H + col) = o;
        }
    }
    __syncthreads();
    if (p.mode == 0) {
        for (int i = threadIdx.x; i < 2 * p.H; i += 1024) {
            const int h = i / p.H, c = i - h * p.H;
            const float v = ((s_dw[(0 * 2 + h) * p.H + c] + s_dw[(1 * 2 + h) * p.H + c]) + s_dw[(2 * 2 + h) * p.H + c]) + s_dw[(3 * 2 + h) * p.H + c];
            p.part[((long long)blockIdx.x * 2 + h) * p.part_ld + c] = v;
        }
        if (threadIdx.x < 2)
            p.part[((long long)blockIdx.x * 2 + threadIdx.x) * p.part_ld + p.H] =
                ((s_db[0][threadIdx.x] + s_db[1][threadIdx.x]) + s_db[2][threadIdx.x]) + s_db[3][threadIdx.x];
    }
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        float v;
        if (p.mode == 0 && k == 1) v = fmaxf(fmaxf(s_st[0][1], s_st[1][1]), fmaxf(s_st[2][1], s_st[3][1]));
        else v = ((s_st[0][k] + s_st[1][k]) + s_st[2][k]) + s_st[3][k];
        p.stat_part[(long long)blockIdx.x * 4 + k] = v;
    }
}

struct ActorFinalizeParams {
    const float* stat_part; int n_wg, M;
    const float* log_alpha; float target_entropy;
    float* alpha_grad; float* stats;               // stats [3]: actor_loss, entropy, alpha_loss
};
__device__ __forceinline__ void actor_finalize_wave(const ActorFinalizeParams& p, int lane) {
    // one wave; lane l sums workgroups l, l + 64, ... in order, then a fixed butterfly
    float sq = 0.0f, se = 0.0f;
    for (int i = lane; i < p.n_wg; i += 64) { sq += p.stat_part[(long long)i * 4]; se += p.stat_part[(long long)i * 4 + 1]; }
    sq = wave_sum(sq); se = wave_sum(se);
    if (lane == 0) {
        const float alpha = expf(p.log_alpha[0]);
        const float entropy = se / (float)p.M;
        p.stats[0] = -(sq / (float)p.M + alpha * entropy);
        p.stats[1] = entropy;
        const float al = alpha * (entropy - p.target_entropy);
        p.stats[2] = al;
        p.alpha_grad[0] = al;                      // d/d(log_alpha) of exp(log_alpha) * c = exp(log_alpha) * c
    }
}
__global__ __launch_bounds__(64) void actor_finalize_kernel(const ActorFinalizeParams p) { actor_finalize_wave(p, threadIdx.x); }

// ---- policy tail, backward -------------------------------------------------------------------------------------------------
// Three launches of the actor phase in one (sac.py:177-189 backward): the Q heads' first-layer data gradient restricted to the
// action columns, d_act = sum_h dh1_h W0_h[:, action columns] (a GEMM with n = A outputs), TanhGaussianHead's backward
// (tanh_gaussian_bwd_kernel: d(mean | log_std) from d_act and d(-log pi)), and the policy's last layer's data gradient
// dh2 = (dfeat W2) (.) [h2 > 0] (a GEMM with k = 2A).  A wave owns a row.  One more workgroup runs actor_finalize.
struct PolicyTailBwdParams {
    const float* dh1; long long dh1_hs;            // Q(s, pi) heads' dh1 [2][M][H], ReLU mask applied
    const float* w0a; long long w0a_hs;            // [2][A][H]: the action columns of the heads' first layer (q_tail's column gather)
    int M, H, A;
    const float* feat; long long ld_feat; const float* eps; const float* saved; const float* scale;
    float ls_min, ls_max, epsilon; const float* d_neglogp;
    float* dfeat; long long ld_dfeat;              // [M][2A]
    const float* h2; const float* w2; float* dh2;  // policy: h2 [M][H], w2 [2A][H], dh2 [M][H]
    int fin_on; ActorFinalizeParams fin;
};

template <int NC>
__global__ __launch_bounds__(256) void policy_tail_bwd_kernel(const PolicyTailBwdParams p) {
    __shared__ float s_v[4][64];                   // [wave]: d_act [A], then dfeat [2A]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_row_blocks = (p.M + 3) >> 2;
    if ((int)blockIdx.x >= n_row_blocks) {
        if (wave == 0 && p.fin_on) actor_finalize_wave(p.fin, lane);
        return;
    }
    const int m = blockIdx.x * 4 + wave;
    const bool live = m < p.M;
    const int mr = live ? m : p.M - 1;
    {
        f32x4 dv[2][NC];
#pragma unroll
        for (int h = 0; h < 2; ++h) tail_load_row<NC>(p.dh1 + h * p.dh1_hs + (long long)mr * p.H, lane, dv[h]);
        for (int j0 = 0; j0 < p.A; j0 += 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float s = 0.0f;
                if (j0 + k < p.A) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x4 wv[NC];
                        tail_load_row<NC>(p.w0a + h * p.w0a_hs + (long long)(j0 + k) * p.H, lane, wv);
#pragma unroll
                        for (int c = 0; c < NC; ++c)
#pragma unroll
                            for (int e = 0; e < 4; ++e) s = __builtin_fmaf(dv[h][c][e], wv[c][e], s);
                    }
                }
                v[k] = s;
            }
            allreduce_add32_x16(v);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float lo, hi;
                both_halves(v[k], lo, hi);
                if (lane == 0 && j0 + k < p.A) s_v[wave][j0 + k] = lo + hi;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // TanhGaussianHead backward, lane = action dimension (the arithmetic of tanh_gaussian_bwd_kernel)
    float g_u = 0.0f, g_ls = 0.0f;
    if (lane < p.A) {
        const int j = lane;
        const float ga = s_v[wave][j];
        const float t = p.saved[(long long)mr * 2 * p.A + j], std = p.saved[(long long)mr * 2 * p.A + p.A + j];
        const float e = p.eps[(long long)mr * p.A + j], sc = p.scale[j];
        const float ls = p.feat[(long long)mr * p.ld_feat + p.A + j];
        const float g_lp = -p.d_neglogp[0];
        const float omt2 = 1.0f - t * t;
        const float sq = 2.0f * sc * t * omt2 / (sc * omt2 + p.epsilon);
        g_u = ga * sc * omt2 + g_lp * sq;
        const float g_std = g_u * e + g_lp * (-1.0f / std);
        g_ls = (ls >= p.ls_min && ls <= p.ls_max) ? g_std * std : 0.0f;
        if (live) { p.dfeat[(long long)m * p.ld_dfeat + j] = g_u; p.dfeat[(long long)m * p.ld_dfeat + p.A + j] = g_ls; }
    }
    __builtin_amdgcn_wave_barrier();               // every lane has read its d_act before the slots are re-used
    if (lane < p.A) { s_v[wave][lane] = g_u; s_v[wave][p.A + lane] = g_ls; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // dh2[m][:] = sum_n dfeat[m][n] W2[n][:], masked by the ReLU that produced h2
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int n = 0; n < 2 * p.A; ++n) {
        const float g = s_v[wave][n];
        f32x4 wv[NC];
        tail_load_row<NC>(p.w2 + (long long)n * p.H, lane, wv);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][e] = __builtin_fmaf(g, wv[c][e], acc[c][e]);
    }
    if (live) {
        f32x4 hv[NC];
        tail_load_row<NC>(p.h2 + (long long)m * p.H, lane, hv);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = hv[c][e] > 0.0f ? acc[c][e] : 0.0f;
            *reinterpret_cast<f32x4*>(p.dh2 + (long long)m * p.H + 256 * c + 4 * lane) = o;
        }
    }
}

// ---- policy tail ----------------------------------------------------------------------------------------------------------
struct PolicyTailParams {
    const float* h2; int M, H;                     // [M][H]
    const float* w2; const float* b2; int A;       // [2A][H], [2A]
    const float* eps;                              // [M][A] or NULL (Philox draws, written to eps_out)
    float* eps_out; unsigned seed_lo, seed_hi; const int* step; int draw_id;
    const float* scale; const float* bias;         // [A]
    float ls_min, ls_max, epsilon;
    float* feat; long long ld_feat;                // [M][2A] (saved: the backward reads mean | log_std)
    float* act0; long long ld0; float* act1; long long ld1;
    float* neg_logp; float* saved;                 // [M], [M][2A]: tanh(u) | std
    // fold (split kernel only): the Q heads' FIRST layer finished here -- h1[h][m][:] = relu(pre[h][m][:] + sum_j action[m][j] w0a[h][j][:])
    // where pre = [feature | state] W0[:, :F+S]^T + b0 came from a GEMM launched earlier (no dependence on the action) and w0a is the
    // compact image of the action columns (pcrl_encoder_pack_attach_cols).  pre and h1 may be the same buffer.
    const float* fold_pre; long long fold_pre_hs; const float* fold_w0a; long long fold_w0a_hs; float* fold_h1; long long fold_h1_hs; int fold_heads;
};

__device__ __forceinline__ float tail_philox_normal(unsigned elem, unsigned draw, unsigned step, unsigned k0, unsigned k1) {
    uint32_t w[4];
    philox4x32_10(elem, draw, step, 0x5AC0FFEEu, k0, k1, w);
    const float u1 = ((float)(w[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// One wave = R consecutive rows (each slice of w2 is loaded once for them), one workgroup = 4 R rows.  R = 1 for small
// batches (the kernel is a latency chain: more, shorter waves), R = 4 for large ones (a quarter of the w2 traffic).
template <int NC, int R>
__global__ __launch_bounds__(256) void policy_tail_fwd_kernel(const PolicyTailParams p) {
    __shared__ float s_feat[4][R][64];             // [wave][row][output]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = (blockIdx.x * 4 + wave) * R;
    f32x4 hv[R][NC];
#pragma unroll
    for (int r = 0; r < R; ++r) tail_load_row<NC>(p.h2 + (long long)min(m0 + r, p.M - 1) * p.H, lane, hv[r]);
    const int n_out = 2 * p.A;
    constexpr int NG = 16 / R;                     // outputs per group
    // sixteen dot products at a time: per-lane partial sums first, then ONE batched cross-lane reduction (sixteen DPP
    // butterflies advancing together; a dependent shuffle chain per dot product cost ~600 cycles each)
    for (int n0 = 0; n0 < n_out; n0 += NG) {
        float v[16];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int n = min(n0 + k, n_out - 1);
            f32x4 wv[NC];
            tail_load_row<NC>(p.w2 + (long long)n * p.H, lane, wv);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = __builtin_fmaf(hv[r][c][e], wv[c][e], s);
                v[R * k + r] = s;
            }
        }
        allreduce_add32_x16(v);
#pragma unroll
        for (int k = 0; k < NG; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float lo, hi;
                both_halves(v[R * k + r], lo, hi);
                if (lane == 0 && n0 + k < n_out) s_feat[wave][r][n0 + k] = (lo + hi) + p.b2[n0 + k];
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int b = m0 + r;
        if (b >= p.M) break;
        for (int n = lane; n < n_out; n += 64) p.feat[(long long)b * p.ld_feat + n] = s_feat[wave][r][n];
        float lp = 0.0f;
        for (int j = lane; j < p.A; j += 64) {
            const float mean = s_feat[wave][r][j], ls = s_feat[wave][r][p.A + j];
            const float std = expf(fminf(fmaxf(ls, p.ls_min), p.ls_max));
            const float e = p.eps ? p.eps[(long long)b * p.A + j]
                                  : tail_philox_normal((unsigned)(b * p.A + j), (unsigned)p.draw_id, (unsigned)p.step[0], p.seed_lo, p.seed_hi);
            if (p.eps_out) p.eps_out[(long long)b * p.A + j] = e;
            const float u = mean + e * std;
            const float t = tanhf(u);
            const float s = p.scale[j];
            const float a = t * s + p.bias[j];
            const float diff = u - mean;
            lp += -(diff * diff) / (2.0f * (std * std)) - logf(std) - kTailHalfLog2Pi - logf(s * (1.0f - t * t) + p.epsilon);
            p.act0[(long long)b * p.ld0 + j] = a;
            if (p.act1) p.act1[(long long)b * p.ld1 + j] = a;
            if (p.saved) { p.saved[(long long)b * 2 * p.A + j] = t; p.saved[(long long)b * 2 * p.A + p.A + j] = std; }
        }
        lp = wave_sum(lp);
        if (lane == 0) p.neg_logp[b] = -lp;
    }
}

// policy_tail_bwd_kernel for small batches with H = 1024: one row per workgroup, a quarter of the hidden vector per wave (see
// policy_tail_fwd_split_kernel); every wave repeats the tiny TanhGaussianHead arithmetic so that one barrier is enough.
// A8 = ceil(A / 8) when all operand pieces of a phase are requested at once (8 A8 rows of both heads' action columns, then 16 A8 rows of
// w2 -- the second batch is issued before the barrier, under the TanhGaussianHead arithmetic); 0 = the loops, for A > 24.
template <int A8>
__global__ __launch_bounds__(256) void policy_tail_bwd_split_kernel(const PolicyTailBwdParams p) {
    __shared__ float s_part[4][32];
    __shared__ float s_df[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= p.M) {
        if (wave == 0 && p.fin_on) actor_finalize_wave(p.fin, lane);
        return;
    }
    const int m = blockIdx.x;
    const int col = 256 * wave + 4 * lane;
    constexpr int RA = A8 > 0 ? 8 * A8 : 1, RW = A8 > 0 ? 16 * A8 : 1;
    f32x4 w2v[RW];
    f32x4 hv;
    {
        f32x4 dv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) dv[h] = *reinterpret_cast<const f32x4*>(p.dh1 + h * p.dh1_hs + (long long)m * p.H + col);
        if constexpr (A8 > 0) {
            f32x4 wa[2][RA];
#pragma unroll
            for (int k = 0; k < RA; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    wa[h][k] = *reinterpret_cast<const f32x4*>(p.w0a + h * p.w0a_hs + (long long)min(k, p.A - 1) * p.H + col);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j0 = 0; j0 < RA; j0 += 16) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float s = 0.0f;
                    if (j0 + k < RA) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int e = 0; e < 4; ++e) s = __builtin_fmaf(dv[h][e], wa[h][j0 + k < RA ? j0 + k : 0][e], s);
                    }
                    v[k] = s;
                }
                allreduce_add32_x16(v);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float lo, hi;
                    both_halves(v[k], lo, hi);
                    if (lane == 0 && j0 + k < p.A) s_part[wave][j0 + k] = lo + hi;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the second phase's operands: nothing below depends on them until the last loop
#pragma unroll
            for (int n = 0; n < RW; ++n) w2v[n] = *reinterpret_cast<const f32x4*>(p.w2 + (long long)min(n, 2 * p.A - 1) * p.H + col);
            hv = *reinterpret_cast<const f32x4*>(p.h2 + (long long)m * p.H + col);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            for (int j0 = 0; j0 < p.A; j0 += 16) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float s = 0.0f;
                    if (j0 + k < p.A) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w0a + h * p.w0a_hs + (long long)(j0 + k) * p.H + col);
#pragma unroll
                            for (int e = 0; e < 4; ++e) s = __builtin_fmaf(dv[h][e], wv[e], s);
                        }
                    }
                    v[k] = s;
                }
                allreduce_add32_x16(v);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float lo, hi;
                    both_halves(v[k], lo, hi);
                    if (lane == 0 && j0 + k < p.A) s_part[wave][j0 + k] = lo + hi;
                }
            }
        }
    }
    __syncthreads();
    float g_u = 0.0f, g_ls = 0.0f;
    if (lane < p.A) {
        const int j = lane;
        const float ga = ((s_part[0][j] + s_part[1][j]) + s_part[2][j]) + s_part[3][j];
        const float t = p.saved[(long long)m * 2 * p.A + j], std = p.saved[(long long)m * 2 * p.A + p.A + j];
        const float e = p.eps[(long long)m * p.A + j], sc = p.scale[j];
        const float ls = p.feat[(long long)m * p.ld_feat + p.A + j];
        const float g_lp = -p.d_neglogp[0];
        const float omt2 = 1.0f - t * t;
        const float sq = 2.0f * sc * t * omt2 / (sc * omt2 + p.epsilon);
        g_u = ga * sc * omt2 + g_lp * sq;
        const float g_std = g_u * e + g_lp * (-1.0f / std);
        g_ls = (ls >= p.ls_min && ls <= p.ls_max) ? g_std * std : 0.0f;
        if (wave == 0) { p.dfeat[(long long)m * p.ld_dfeat + j] = g_u; p.dfeat[(long long)m * p.ld_dfeat + p.A + j] = g_ls; }
        s_df[wave][j] = g_u; s_df[wave][p.A + j] = g_ls;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (A8 > 0) {
#pragma unroll
        for (int n = 0; n < RW; ++n) {
            if (n < 2 * p.A) {
                const float g = s_df[wave][n];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(g, w2v[n][e], acc[e]);
            }
        }
    } else {
        for (int n = 0; n < 2 * p.A; ++n) {
            const float g = s_df[wave][n];
            const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w2 + (long long)n * p.H + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(g, wv[e], acc[e]);
        }
        hv = *reinterpret_cast<const f32x4*>(p.h2 + (long long)m * p.H + col);
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = hv[e] > 0.0f ? acc[e] : 0.0f;
    *reinterpret_cast<f32x4*>(p.dh2 + (long long)m * p.H + col) = o;
}

// The same head for SMALL batches with H = 1024: ONE row per workgroup, its four waves take a quarter of the hidden vector each
// (one 1 KB piece of h2 and of every w2 row per wave instead of four: the dependent-load chain of a wave is 4x shorter and M
// workgroups instead of M / 4 spread over the chip), partial dot products meet in LDS and are added in wave order.
// G = groups of sixteen outputs whose w2 pieces are ALL requested before the first dot product (G * 16 >= 2 A; 64 registers per group):
// the groups were G dependent L2 round trips, one behind the other, in a kernel that is nothing but latency (A = 22: 16.5 us for 23 MFLOP).
// G = 0: the loop, one group in flight (2 A > 48).  Same sums in the same order either way.
template <int G>
__global__ __launch_bounds__(256) void policy_tail_fwd_split_kernel(const PolicyTailParams p) {
    __shared__ float s_part[4][64];
    __shared__ float s_feat[64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x;
    const f32x4 hv = *reinterpret_cast<const f32x4*>(p.h2 + (long long)b * p.H + 256 * wave + 4 * lane);
    const int n_out = 2 * p.A;
    // the fold's operands (G = 1, i.e. A <= 8, two heads): requested with everything else instead of 2 A dependent round trips at the end
    constexpr int FJ = G == 1 ? 8 : 1;
    f32x4 fw[2][FJ], fpre[2];
    const bool fold_early = G == 1 && p.fold_h1 && p.fold_heads <= 2;
    if (fold_early) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hh = min(h, p.fold_heads - 1);
            fpre[h] = *reinterpret_cast<const f32x4*>(p.fold_pre + hh * p.fold_pre_hs + (long long)b * p.H + 256 * wave + 4 * lane);
#pragma unroll
            for (int j = 0; j < FJ; ++j)
                fw[h][j] = *reinterpret_cast<const f32x4*>(p.fold_w0a + hh * p.fold_w0a_hs + (long long)min(j, p.A - 1) * p.H + 256 * wave + 4 * lane);
        }
    }
    if constexpr (G > 0) {
        f32x4 wv[G][16];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                wv[g][k] = *reinterpret_cast<const f32x4*>(p.w2 + (long long)min(16 * g + k, n_out - 1) * p.H + 256 * wave + 4 * lane);
        __builtin_amdgcn_sched_barrier(0);      // the scheduler would sink every load to its first use ("operand rings are pinned")
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float s = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_fmaf(hv[e], wv[g][k][e], s);
                v[k] = s;
            }
            allreduce_add32_x16(v);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float lo, hi;
                both_halves(v[k], lo, hi);
                if (lane == 0 && 16 * g + k < n_out) s_part[wave][16 * g + k] = lo + hi;
            }
        }
    } else {
        for (int n0 = 0; n0 < n_out; n0 += 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int n = min(n0 + k, n_out - 1);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w2 + (long long)n * p.H + 256 * wave + 4 * lane);
                float s = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_fmaf(hv[e], wv[e], s);
                v[k] = s;
            }
            allreduce_add32_x16(v);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float lo, hi;
                both_halves(v[k], lo, hi);
                if (lane == 0 && n0 + k < n_out) s_part[wave][n0 + k] = lo + hi;
            }
        }
    }
    __syncthreads();
    __shared__ float s_act[32];
    if (wave == 0) {
        if (lane < n_out) s_feat[lane] = (((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane]) + p.b2[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int n = lane; n < n_out; n += 64) p.feat[(long long)b * p.ld_feat + n] = s_feat[n];
        float lp = 0.0f;
        if (lane < p.A) {
            const int j = lane;
            const float mean = s_feat[j], ls = s_feat[p.A + j];
            const float std = expf(fminf(fmaxf(ls, p.ls_min), p.ls_max));
            const float e = p.eps ? p.eps[(long long)b * p.A + j]
                                  : tail_philox_normal((unsigned)(b * p.A + j), (unsigned)p.draw_id, (unsigned)p.step[0], p.seed_lo, p.seed_hi);
            if (p.eps_out) p.eps_out[(long long)b * p.A + j] = e;
            const float u = mean + e * std;
            const float t = tanhf(u);
            const float sc = p.scale[j];
            const float a = t * sc + p.bias[j];
            const float diff = u - mean;
            lp = -(diff * diff) / (2.0f * (std * std)) - logf(std) - kTailHalfLog2Pi - logf(sc * (1.0f - t * t) + p.epsilon);
            p.act0[(long long)b * p.ld0 + j] = a;
            if (p.act1) p.act1[(long long)b * p.ld1 + j] = a;
            if (p.saved) { p.saved[(long long)b * 2 * p.A + j] = t; p.saved[(long long)b * 2 * p.A + p.A + j] = std; }
            s_act[j] = a;
        }
        lp = wave_sum(lp);
        if (lane == 0) p.neg_logp[b] = -lp;
    }
    if (!p.fold_h1) return;
    __syncthreads();
    // ---- the Q heads' first layer for this row, a quarter of the hidden vector per wave ----
    const int col = 256 * wave + 4 * lane;
    if (fold_early) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h >= p.fold_heads) break;
            f32x4 acc = fpre[h];
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                if (j < p.A) {
                    const float a = s_act[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(a, fw[h][j][e], acc[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc[e] > 0.0f ? acc[e] : 0.0f;
            *reinterpret_cast<f32x4*>(p.fold_h1 + h * p.fold_h1_hs + (long long)b * p.H + col) = acc;
        }
        return;
    }
    for (int h = 0; h < p.fold_heads; ++h) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(p.fold_pre + h * p.fold_pre_hs + (long long)b * p.H + col);
        for (int j = 0; j < p.A; ++j) {
            const float a = s_act[j];
            const f32x4 wv = *reinterpret_cast<const f32x4*>(p.fold_w0a + h * p.fold_w0a_hs + (long long)j * p.H + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(a, wv[e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] > 0.0f ? acc[e] : 0.0f;
        *reinterpret_cast<f32x4*>(p.fold_h1 + h * p.fold_h1_hs + (long long)b * p.H + col) = acc;
    }
}

// ---- fixed-order column reductions of per-workgroup partials (types and the block body: common.h) ---------------------------------
__global__ __launch_bounds__(256) void colsum_jobs_kernel(const ColsumParams p) { colsum_block(p, (int)blockIdx.x, (int)threadIdx.x); }

}  // namespace pcrl

using namespace pcrl;

extern "C" int pcrl_q_tail_workspace_floats(int32_t M, int32_t H, size_t* part_floats, size_t* stat_floats) {
    if (!part_floats || !stat_floats || M < 1 || H < 1) return fail(PCRL_E_ARG, "bad arguments");
    const size_t n_wg = (size_t)(M + 3) / 4;
    *part_floats = n_wg * 2 * (size_t)(H + 4);
    *stat_floats = n_wg * 4;
    return PCRL_OK;
}

static int q_tail_launch(const QTailParams& p, hipStream_t st) {
    if (p.H % 256 || p.H < 256 || p.H > 256 * kTailMaxChunks) return fail(PCRL_E_ARG, "head tail: H must be a multiple of 256, <= %d (got %d)", 256 * kTailMaxChunks, p.H);
    constexpr int split_max = kTailSplitMaxRows;
    if (p.H == 1024 && p.M <= split_max) {         // small batch: a row's hidden vector over four waves
        const int grid_s = (p.M + 3) / 4 + (p.cg_dst ? (2 * p.cg_ncols * p.H + 1023) / 1024 : 0);
        const size_t lds_s = p.mode == 0 ? sizeof(float) * 8 * (size_t)p.H : 0;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(q_tail_split_kernel), sizeof(float) * 8 * 1024)) return rc;
        hipLaunchKernelGGL(q_tail_split_kernel, dim3(grid_s), dim3(1024), lds_s, st, p);
        PCRL_CHECK_LAUNCH("q_tail_split_kernel");
        return PCRL_OK;
    }
    const int grid = (p.M + 3) / 4 + (p.cg_dst ? (2 * p.cg_ncols * p.H + 255) / 256 : 0);
    const size_t lds = p.mode == 0 ? sizeof(float) * 8 * (size_t)p.H : 0;
    switch (p.H / 256) {
#define PCRL_QT_CASE(NC_) case NC_: \
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(q_tail_kernel<NC_>), sizeof(float) * 8 * 256 * kTailMaxChunks)) return rc; \
        hipLaunchKernelGGL(q_tail_kernel<NC_>, dim3(grid), dim3(256), lds, st, p); break;
        PCRL_QT_CASE(1) PCRL_QT_CASE(2) PCRL_QT_CASE(3) PCRL_QT_CASE(4) PCRL_QT_CASE(5) PCRL_QT_CASE(6) PCRL_QT_CASE(7) PCRL_QT_CASE(8)
#undef PCRL_QT_CASE
    }
    PCRL_CHECK_LAUNCH("q_tail_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_q_tail_critic_f32(const float* h2, int64_t h2_head_stride, const float* w2, const float* b2, int64_t w_head_stride,
                                      const float* h2_target, int64_t h2_target_head_stride, const float* w2_target, const float* b2_target,
                                      int64_t w_target_head_stride, const float* neg_logp_next, const float* rewards, const uint8_t* dones,
                                      int32_t rd_row_div, const float* log_alpha, float gamma, float reward_scale, int32_t ignore_dones,
                                      int32_t group, int32_t M, int32_t H, float* q, int64_t ld_q, float* q_target, float* dq, int64_t ld_dq,
                                      float* dh2, int64_t dh2_head_stride, float* part, float* stat_part, void* stream) {
    if (!h2 || !w2 || !b2 || !h2_target || !w2_target || !b2_target || !neg_logp_next || !rewards || !log_alpha || !q || !q_target || !dq ||
        !dh2 || !part || !stat_part) return fail(PCRL_E_ARG, "NULL argument");
    if (!ignore_dones && !dones) return fail(PCRL_E_ARG, "dones is NULL");
    if (M < 1 || (group != 1 && group != 2 && group != 4) || M % group) return fail(PCRL_E_ARG, "head tail: group must be 1, 2 or 4 and divide M (M=%d group=%d)", M, group);
    QTailParams p{};
    p.mode = 0; p.h2 = h2; p.h2_hs = h2_head_stride; p.w2 = w2; p.b2 = b2; p.w_hs = w_head_stride;
    p.h2_t = h2_target; p.h2_t_hs = h2_target_head_stride; p.w2_t = w2_target; p.b2_t = b2_target; p.w_t_hs = w_target_head_stride;
    p.nlp = neg_logp_next; p.rewards = rewards; p.dones = dones; p.rd_div = rd_row_div > 1 ? rd_row_div : 1;
    p.log_alpha = log_alpha; p.gamma = gamma; p.reward_scale = reward_scale; p.ignore_dones = ignore_dones; p.group = group;
    p.M = M; p.H = H; p.q = q; p.ld_q = ld_q; p.q_target = q_target; p.dq = dq; p.ld_dq = ld_dq; p.dh2 = dh2; p.dh2_hs = dh2_head_stride;
    p.part = part; p.part_ld = H + 4; p.stat_part = stat_part;
    return q_tail_launch(p, (hipStream_t)stream);
}

extern "C" int pcrl_q_tail_actor_f32(const float* h2, int64_t h2_head_stride, const float* w2, const float* b2, int64_t w_head_stride,
                                     const float* neg_logp, const float* log_alpha, int32_t M, int32_t H, float* q, int64_t ld_q,
                                     float* dq, int64_t ld_dq, float* dh2, int64_t dh2_head_stride, float* d_neglogp, float* stat_part,
                                     void* stream) {
    if (!h2 || !w2 || !b2 || !neg_logp || !log_alpha || !q || !dq || !dh2 || !d_neglogp || !stat_part) return fail(PCRL_E_ARG, "NULL argument");
    if (M < 1) return fail(PCRL_E_ARG, "bad shape");
    QTailParams p{};
    p.mode = 1; p.h2 = h2; p.h2_hs = h2_head_stride; p.w2 = w2; p.b2 = b2; p.w_hs = w_head_stride; p.nlp = neg_logp; p.log_alpha = log_alpha;
    p.group = 1; p.rd_div = 1; p.M = M; p.H = H; p.q = q; p.ld_q = ld_q; p.dq = dq; p.ld_dq = ld_dq; p.dh2 = dh2; p.dh2_hs = dh2_head_stride;
    p.d_neglogp = d_neglogp; p.stat_part = stat_part;
    return q_tail_launch(p, (hipStream_t)stream);
}

extern "C" int pcrl_q_tail_actor_cols_f32(const float* h2, int64_t h2_head_stride, const float* w2, const float* b2, int64_t w_head_stride,
                                          const float* neg_logp, const float* log_alpha, int32_t M, int32_t H, float* q, int64_t ld_q,
                                          float* dq, int64_t ld_dq, float* dh2, int64_t dh2_head_stride, float* d_neglogp, float* stat_part,
                                          const float* w0, int64_t w0_head_stride, int32_t ld_w0, int32_t col0, int32_t ncols, float* cols_out,
                                          void* stream) {
    if (!h2 || !w2 || !b2 || !neg_logp || !log_alpha || !q || !dq || !dh2 || !d_neglogp || !stat_part) return fail(PCRL_E_ARG, "NULL argument");
    if (M < 1) return fail(PCRL_E_ARG, "bad shape");
    if (!w0 || !cols_out || ncols < 1 || col0 < 0 || col0 + ncols > ld_w0) return fail(PCRL_E_ARG, "q tail: bad column gather (col0=%d ncols=%d ld=%d)", col0, ncols, ld_w0);
    QTailParams p{};
    p.mode = 1; p.h2 = h2; p.h2_hs = h2_head_stride; p.w2 = w2; p.b2 = b2; p.w_hs = w_head_stride; p.nlp = neg_logp; p.log_alpha = log_alpha;
    p.group = 1; p.rd_div = 1; p.M = M; p.H = H; p.q = q; p.ld_q = ld_q; p.dq = dq; p.ld_dq = ld_dq; p.dh2 = dh2; p.dh2_hs = dh2_head_stride;
    p.d_neglogp = d_neglogp; p.stat_part = stat_part;
    p.cg_w0 = w0; p.cg_hs = w0_head_stride; p.cg_ld = ld_w0; p.cg_col0 = col0; p.cg_ncols = ncols; p.cg_dst = cols_out;
    return q_tail_launch(p, (hipStream_t)stream);
}

extern "C" int pcrl_policy_tail_bwd_f32(const float* dh1, int64_t dh1_head_stride, const float* w0_action_cols, int64_t w0a_head_stride,
                                        int32_t M, int32_t H, int32_t A, const float* feat, int64_t ld_feat, const float* eps, const float* saved,
                                        const float* scale, float log_std_min, float log_std_max, float epsilon, const float* d_neglogp,
                                        float* d_feat, int64_t ld_d_feat, const float* h2, const float* w2, float* dh2,
                                        const float* stat_part, const float* log_alpha, float target_entropy, float* alpha_grad, float* stats,
                                        void* stream) {
    if (!dh1 || !w0_action_cols || !feat || !eps || !saved || !scale || !d_neglogp || !d_feat || !h2 || !w2 || !dh2) return fail(PCRL_E_ARG, "NULL argument");
    if (M < 1 || A < 1 || 2 * A > 64) return fail(PCRL_E_ARG, "policy tail: 1 <= A <= 32 (got %d)", A);
    if (H % 256 || H < 256 || H > 1024) return fail(PCRL_E_ARG, "policy tail: H must be 256, 512, 768 or 1024 (got %d)", H);
    if (stat_part && (!log_alpha || !alpha_grad || !stats)) return fail(PCRL_E_ARG, "policy tail: the finalize part needs log_alpha, alpha_grad and stats");
    PolicyTailBwdParams p{};
    p.dh1 = dh1; p.dh1_hs = dh1_head_stride; p.w0a = w0_action_cols; p.w0a_hs = w0a_head_stride; p.M = M; p.H = H; p.A = A;
    p.feat = feat; p.ld_feat = ld_feat; p.eps = eps; p.saved = saved; p.scale = scale; p.ls_min = log_std_min; p.ls_max = log_std_max;
    p.epsilon = epsilon; p.d_neglogp = d_neglogp; p.dfeat = d_feat; p.ld_dfeat = ld_d_feat; p.h2 = h2; p.w2 = w2; p.dh2 = dh2;
    p.fin_on = stat_part != nullptr;
    p.fin = ActorFinalizeParams{stat_part, (M + 3) / 4, M, log_alpha, target_entropy, alpha_grad, stats};
    hipStream_t st = (hipStream_t)stream;
    constexpr int split_max = kTailSplitMaxRows;
    if (H == 1024 && M <= split_max && A <= 32) {
        constexpr bool prefetch = true;          // every operand piece of a phase requested up front (A <= 24); the <0> loops serve wider heads
        const dim3 g(M + (p.fin_on ? 1 : 0));
        const int a8 = prefetch ? (A + 7) / 8 : 0;
        if (a8 == 1) hipLaunchKernelGGL(policy_tail_bwd_split_kernel<1>, g, dim3(256), 0, st, p);
        else if (a8 == 2) hipLaunchKernelGGL(policy_tail_bwd_split_kernel<2>, g, dim3(256), 0, st, p);
        else if (a8 == 3) hipLaunchKernelGGL(policy_tail_bwd_split_kernel<3>, g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(policy_tail_bwd_split_kernel<0>, g, dim3(256), 0, st, p);
        PCRL_CHECK_LAUNCH("policy_tail_bwd_split_kernel");
        return PCRL_OK;
    }
    const int grid = (M + 3) / 4 + (p.fin_on ? 1 : 0);
    switch (H / 256) {
        case 1: hipLaunchKernelGGL(policy_tail_bwd_kernel<1>, dim3(grid), dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(policy_tail_bwd_kernel<2>, dim3(grid), dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(policy_tail_bwd_kernel<3>, dim3(grid), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(policy_tail_bwd_kernel<4>, dim3(grid), dim3(256), 0, st, p); break;
    }
    PCRL_CHECK_LAUNCH("policy_tail_bwd_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_actor_finalize_f32(const float* stat_part, int32_t M, const float* log_alpha, float target_entropy, float* alpha_grad,
                                       float* stats, void* stream) {
    if (!stat_part || !log_alpha || !alpha_grad || !stats || M < 1) return fail(PCRL_E_ARG, "bad arguments");
    ActorFinalizeParams p{stat_part, (M + 3) / 4, M, log_alpha, target_entropy, alpha_grad, stats};
    hipLaunchKernelGGL(actor_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("actor_finalize_kernel");
    return PCRL_OK;
}

static int policy_tail_fwd_impl(const float* h2, int32_t M, int32_t H, const float* w2, const float* b2, int32_t A, const float* eps,
                                uint64_t seed, const int32_t* step_counter, int32_t draw_id, float* eps_out, const float* scale,
                                const float* bias, float log_std_min, float log_std_max, float epsilon, float* feat, int64_t ld_feat,
                                float* action, int64_t ld_action, float* action2, int64_t ld_action2, float* neg_logp, float* saved,
                                const float* fold_pre, int64_t fold_pre_hs, const float* fold_w0a, int64_t fold_w0a_hs, int32_t fold_heads,
                                float* fold_h1, int64_t fold_h1_hs, void* stream);

extern "C" int pcrl_policy_tail_fwd_f32(const float* h2, int32_t M, int32_t H, const float* w2, const float* b2, int32_t A, const float* eps,
                                        uint64_t seed, const int32_t* step_counter, int32_t draw_id, float* eps_out, const float* scale,
                                        const float* bias, float log_std_min, float log_std_max, float epsilon, float* feat, int64_t ld_feat,
                                        float* action, int64_t ld_action, float* action2, int64_t ld_action2, float* neg_logp, float* saved,
                                        void* stream) {
    return policy_tail_fwd_impl(h2, M, H, w2, b2, A, eps, seed, step_counter, draw_id, eps_out, scale, bias, log_std_min, log_std_max, epsilon, feat,
                                ld_feat, action, ld_action, action2, ld_action2, neg_logp, saved, nullptr, 0, nullptr, 0, 0, nullptr, 0, stream);
}

extern "C" int pcrl_policy_tail_fwd_fold_f32(const float* h2, int32_t M, int32_t H, const float* w2, const float* b2, int32_t A, const float* eps,
                                             uint64_t seed, const int32_t* step_counter, int32_t draw_id, float* eps_out, const float* scale,
                                             const float* bias, float log_std_min, float log_std_max, float epsilon, float* feat, int64_t ld_feat,
                                             float* action, int64_t ld_action, float* action2, int64_t ld_action2, float* neg_logp, float* saved,
                                             const float* pre, int64_t pre_head_stride, const float* w0_action_cols, int64_t w0a_head_stride,
                                             int32_t n_heads, float* h1, int64_t h1_head_stride, void* stream) {
    if (!pre || !w0_action_cols || !h1 || n_heads < 1 || n_heads > 4) return fail(PCRL_E_ARG, "policy tail fold: NULL argument or n_heads outside [1, 4]");
    return policy_tail_fwd_impl(h2, M, H, w2, b2, A, eps, seed, step_counter, draw_id, eps_out, scale, bias, log_std_min, log_std_max, epsilon, feat,
                                ld_feat, action, ld_action, action2, ld_action2, neg_logp, saved, pre, pre_head_stride, w0_action_cols,
                                w0a_head_stride, n_heads, h1, h1_head_stride, stream);
}

static int policy_tail_fwd_impl(const float* h2, int32_t M, int32_t H, const float* w2, const float* b2, int32_t A, const float* eps,
                                uint64_t seed, const int32_t* step_counter, int32_t draw_id, float* eps_out, const float* scale,
                                const float* bias, float log_std_min, float log_std_max, float epsilon, float* feat, int64_t ld_feat,
                                float* action, int64_t ld_action, float* action2, int64_t ld_action2, float* neg_logp, float* saved,
                                const float* fold_pre, int64_t fold_pre_hs, const float* fold_w0a, int64_t fold_w0a_hs, int32_t fold_heads,
                                float* fold_h1, int64_t fold_h1_hs, void* stream) {
    if (!h2 || !w2 || !b2 || !scale || !bias || !feat || !action || !neg_logp) return fail(PCRL_E_ARG, "NULL argument");
    if (!eps && (!step_counter || !eps_out)) return fail(PCRL_E_ARG, "in-kernel draws need step_counter and eps_out");
    if (M < 1 || A < 1 || 2 * A > 64) return fail(PCRL_E_ARG, "policy tail: 1 <= A <= 32 (got %d)", A);
    if (H % 256 || H < 256 || H > 1024) return fail(PCRL_E_ARG, "policy tail: H must be 256, 512, 768 or 1024 (got %d)", H);
    PolicyTailParams p{h2, M, H, w2, b2, A, eps, eps_out, (unsigned)seed, (unsigned)(seed >> 32), step_counter, draw_id, scale, bias,
                       log_std_min, log_std_max, epsilon, feat, ld_feat, action, ld_action, action2, ld_action2, neg_logp, saved,
                       fold_pre, fold_pre_hs, fold_w0a, fold_w0a_hs, fold_h1, fold_h1_hs, fold_heads};
    hipStream_t st = (hipStream_t)stream;
    constexpr int split_max = kTailSplitMaxRows;
    if (fold_h1 && !(H == 1024 && M <= split_max && A <= 32))
        return fail(PCRL_E_ARG, "policy tail fold: built for H = 1024, M <= %d, A <= 32 (got H=%d M=%d A=%d)", split_max, H, M, A);
    if (H == 1024 && M <= split_max) {             // small batch: one row per workgroup, a quarter of the hidden vector per wave
        constexpr bool prefetch = true;
        const int groups = (2 * A + 15) / 16;
        if (prefetch && groups == 1) hipLaunchKernelGGL(policy_tail_fwd_split_kernel<1>, dim3(M), dim3(256), 0, st, p);
        else if (prefetch && groups == 2) hipLaunchKernelGGL(policy_tail_fwd_split_kernel<2>, dim3(M), dim3(256), 0, st, p);
        else if (prefetch && groups == 3) hipLaunchKernelGGL(policy_tail_fwd_split_kernel<3>, dim3(M), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(policy_tail_fwd_split_kernel<0>, dim3(M), dim3(256), 0, st, p);
        PCRL_CHECK_LAUNCH("policy_tail_fwd_split_kernel");
        return PCRL_OK;
    }
    const bool wide = M > 512;                     // rows per wave: 4 when the batch fills the chip anyway
    const int grid = wide ? (M + 15) / 16 : (M + 3) / 4;
#define PCRL_PT_CASE(NC_) \
    if (wide) hipLaunchKernelGGL((policy_tail_fwd_kernel<NC_, 4>), dim3(grid), dim3(256), 0, st, p); \
    else hipLaunchKernelGGL((policy_tail_fwd_kernel<NC_, 1>), dim3(grid), dim3(256), 0, st, p);
    switch (H / 256) {
        case 1: PCRL_PT_CASE(1) break;
        case 2: PCRL_PT_CASE(2) break;
        case 3: PCRL_PT_CASE(3) break;
        default: PCRL_PT_CASE(4) break;
    }
#undef PCRL_PT_CASE
    PCRL_CHECK_LAUNCH("policy_tail_fwd_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_colsum_jobs_f32(const pcrl_colsum_job* jobs, int32_t n, void* stream) {
    if (!jobs || n < 1 || n > kColsumJobs) return fail(PCRL_E_ARG, "colsum jobs: 1 <= n <= %d", kColsumJobs);
    ColsumParams p;
    const int blocks = colsum_fill(jobs, n, p);
    if (blocks < 0) return fail(PCRL_E_ARG, "bad colsum job");
    if (blocks == 0) return PCRL_OK;
    hipLaunchKernelGGL(colsum_jobs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    PCRL_CHECK_LAUNCH("colsum_jobs_kernel");
    return PCRL_OK;
}
