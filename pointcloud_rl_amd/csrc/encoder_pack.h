// encoder_pack.h -- the weights -> MFMA operand order pack of the encoder (one thread per packed float) and the column-gather jobs that
// ride on it, as a block body: run by encoder_pack_kernel (encoder_fwd.hip) and, as extra workgroups, by the replay's sampling launch
// (replay.hip: the step's first launch depends on nothing the pack reads, so the critic phase's re-pack needs no launch of its own).
#pragma once
#include "common.h"

namespace pcrl {

// Column-gather jobs riding on the pack launch (pcrl_encoder_pack_attach_cols): columns [col0, col0 + ncols) of `heads` weight
// matrices src + h * head_stride [rows][ld] written as the compact image dst [heads][ncols][rows] -- the action columns of the Q heads'
// first layer, which the policy tails contract row-wise (coalesced rows instead of 4-byte loads ld floats apart).
struct ColGather { const float* src; long long head_stride; int heads, rows, ld, col0, ncols; float* dst; int blk_begin; };
struct ColGatherList { ColGather job[2]; int n; };


// One pack launch's worth of work: blocks [0, main_blocks) pack, blocks [main_blocks, total_blocks) run the attached column gathers.
struct PackJob { pcrl_encoder_weights w; int T0; float* out; ColGatherList cg; int main_blocks, total_blocks; };

// Weights (reference state_dict layout) -> operand order.  One thread per packed float; 256 threads per block.
__device__ __forceinline__ void encoder_pack_block(const PackJob& job, int blk_id, int tid) {
    if (blk_id >= job.main_blocks) {
        const int blk = blk_id - job.main_blocks;
        const ColGather& g = (job.cg.n > 1 && blk >= job.cg.job[1].blk_begin) ? job.cg.job[1] : job.cg.job[0];
        const int per_head = g.ncols * g.rows;
        const int e = (blk - g.blk_begin) * 256 + tid;
        if (e < g.heads * per_head) {
            const int h = e / per_head, r = e - h * per_head, j = r / g.rows, row = r - j * g.rows;
            g.dst[e] = g.src[h * g.head_stride + (long long)row * g.ld + g.col0 + j];
        }
        return;
    }
    const pcrl_encoder_weights& w = job.w;
    float* __restrict__ out = job.out;
    const int T0 = job.T0;
    const PackedLayout L{T0, w.c1, w.c2, w.c3};
    const int i = blk_id * 256 + tid;
    if (i >= L.total()) return;
    float v = 0.0f;
    if (i < L.b0()) {                       // conv0: [mb][t][lane], natural k order, zero padded
        const int e = i - L.w0(), ln = e & 63, t = (e >> 6) % T0, mb = (e >> 6) / T0;
        const int row = 32 * mb + (ln & 31), k = 2 * t + (ln >> 5);
        v = k < w.c_in ? w.w0[row * w.c_in + k] : 0.0f;
    } else if (i < L.b0() + w.c1) {
        v = w.b0[i - L.b0()];
    } else if (i >= L.w1() && i < L.ln1()) { // conv1: [mb][tq][lane][4]
        const int e = i - L.w1(), j = e & 3, ln = (e >> 2) & 63, q = e >> 8;
        const int TQ = w.c1 / 8, tq = q % TQ, mb = q / TQ;
        v = w.w1[(32 * mb + (ln & 31)) * w.c1 + acc_chan(4 * tq + j, ln >> 5)];
    } else if (i >= L.ln1() && i < L.ln1() + 2 * w.c2) {
        const int e = i - L.ln1();
        v = (e & 1) ? w.be1[e >> 1] : w.g1[e >> 1];
    } else if (i >= L.w2() && i < L.ln2()) { // conv2: [mb][tq][lane][4]
        const int e = i - L.w2(), j = e & 3, ln = (e >> 2) & 63, q = e >> 8;
        const int TQ = w.c2 / 8, tq = q % TQ, mb = q / TQ;
        v = w.w2[(32 * mb + (ln & 31)) * w.c2 + acc_chan(4 * tq + j, ln >> 5)];
    } else if (i >= L.ln2() && i < L.ln2() + 2 * w.c3) {
        const int e = i - L.ln2();
        v = (e & 1) ? w.be2[e >> 1] : w.g2[e >> 1];
    } else if (i >= L.w2t() && i < L.w1t()) { // conv2 transposed (dX GEMM of the backward): rows = c2, k = c3
        const int e = i - L.w2t(), j = e & 3, ln = (e >> 2) & 63, q = e >> 8;
        const int TQ = w.c3 / 8, tq = q % TQ, mb = q / TQ;
        v = w.w2[acc_chan(4 * tq + j, ln >> 5) * w.c2 + 32 * mb + (ln & 31)];
    } else if (i >= L.w1t() && i < L.w1t() + w.c1 * w.c2) { // conv1 transposed: rows = c1, k = c2
        const int e = i - L.w1t(), j = e & 3, ln = (e >> 2) & 63, q = e >> 8;
        const int TQ = w.c2 / 8, tq = q % TQ, mb = q / TQ;
        v = w.w1[acc_chan(4 * tq + j, ln >> 5) * w.c1 + 32 * mb + (ln & 31)];
    } else if (i >= L.w1b() && i < L.w1b() + w.c1 * w.c2 / 2) {   // conv1, bf16: [mb][g][lane][8], two elements per slot
        unsigned bits = 0;
        for (int k = 0; k < 2; ++k) {
            const int e = 2 * (i - L.w1b()) + k, r = e & 7, ln = (e >> 3) & 63, q = e >> 9;
            const int G = w.c1 / 16, g = q % G, mb = q / G;
            bits |= bf16_rne_bits(w.w1[(32 * mb + (ln & 31)) * w.c1 + acc_chan(8 * g + r, ln >> 5)]) << (16 * k);
        }
        v = u2f(bits);
    } else if (i >= L.w2b() && i < L.w2b() + w.c2 * w.c3 / 2) {   // conv2, bf16
        unsigned bits = 0;
        for (int k = 0; k < 2; ++k) {
            const int e = 2 * (i - L.w2b()) + k, r = e & 7, ln = (e >> 3) & 63, q = e >> 9;
            const int G = w.c2 / 16, g = q % G, mb = q / G;
            bits |= bf16_rne_bits(w.w2[(32 * mb + (ln & 31)) * w.c2 + acc_chan(8 * g + r, ln >> 5)]) << (16 * k);
        }
        v = u2f(bits);
    } else if (i >= L.w2tb() && i < L.w2tb() + w.c2 * w.c3 / 2) {  // conv2 transposed, bf16: rows = c2, k = c3
        unsigned bits = 0;
        for (int k = 0; k < 2; ++k) {
            const int e = 2 * (i - L.w2tb()) + k, r = e & 7, ln = (e >> 3) & 63, q = e >> 9;
            const int G = w.c3 / 16, g = q % G, mb = q / G;
            bits |= bf16_rne_bits(w.w2[acc_chan(8 * g + r, ln >> 5) * w.c2 + 32 * mb + (ln & 31)]) << (16 * k);
        }
        v = u2f(bits);
    } else if (i >= L.w1tb() && i < L.w1tb() + w.c1 * w.c2 / 2) {  // conv1 transposed, bf16: rows = c1, k = c2
        unsigned bits = 0;
        for (int k = 0; k < 2; ++k) {
            const int e = 2 * (i - L.w1tb()) + k, r = e & 7, ln = (e >> 3) & 63, q = e >> 9;
            const int G = w.c2 / 16, g = q % G, mb = q / G;
            bits |= bf16_rne_bits(w.w1[acc_chan(8 * g + r, ln >> 5) * w.c1 + 32 * mb + (ln & 31)]) << (16 * k);
        }
        v = u2f(bits);
    }
    else {
        // split images: three bf16 terms of every weight, same element order as the bf16 images
        for (int layer = 1; layer <= 2; ++layer) {
            const int in_c = layer == 1 ? w.c1 : w.c2;                   // the layer's input channels = row length of its weight
            const int out_c = layer == 1 ? w.c2 : w.c3;
            const int n_img = in_c * out_c / 2;
            const float* src = layer == 1 ? w.w1 : w.w2;
            for (int term = 0; term < 3; ++term) {
                const int base = layer == 1 ? L.w1s(term) : L.w2s(term);         // forward: rows = outputs, contraction over inputs
                const int base_t = layer == 1 ? L.w1ts(term) : L.w2ts(term);     // transposed: rows = inputs, contraction over outputs
                if (i >= base && i < base + n_img) {
                    unsigned bits = 0;
                    for (int k = 0; k < 2; ++k) {
                        const int e = 2 * (i - base) + k, r = e & 7, ln = (e >> 3) & 63, q = e >> 9;
                        const int G = in_c / 16, g = q % G, mb = q / G;
                        bits |= bf16_split_bits(src[(32 * mb + (ln & 31)) * in_c + acc_chan(8 * g + r, ln >> 5)], term) << (16 * k);
                    }
                    v = u2f(bits);
                } else if (i >= base_t && i < base_t + n_img) {
                    unsigned bits = 0;
                    for (int k = 0; k < 2; ++k) {
                        const int e = 2 * (i - base_t) + k, r = e & 7, ln = (e >> 3) & 63, q = e >> 9;
                        const int G = out_c / 16, g = q % G, mb = q / G;
                        bits |= bf16_split_bits(src[acc_chan(8 * g + r, ln >> 5) * in_c + 32 * mb + (ln & 31)], term) << (16 * k);
                    }
                    v = u2f(bits);
                }
            }
        }
    }
    out[i] = v;
}

// The pack job a host thread has handed over for its next replay sampling launch (pcrl_encoder_pack_attach_to_gather), taken by that
// launch (replay.hip) or, if none came, by pcrl_encoder_pack_flush_pending.  Defined in encoder_fwd.hip.
bool take_pending_pack(PackJob* job);

}  // namespace pcrl
