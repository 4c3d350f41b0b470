// Device-resident replay sampling for gfx950: one launch gathers the sampled rows of every stored key
// (obs/xyz, obs/rgb, next_obs/..., actions, rewards, dones, ...) into the persistent staging batch the
// captured update step reads.  Replaces ReplayMemory.sample's numpy `take` per key plus the pageable
// host->device copy of 2*B clouds per step (pyrl/env/replay_buffer.py:297-322; pyrl/utils/data/
// dict_array.py:308-318; sac.py:104).  HBM-bound: row_bytes read + row_bytes written per sampled row.
#include "common.h"
#include "encoder_pack.h"

namespace pcrl {

constexpr int kMaxGatherSegs = 24;
struct GatherParams {
    const unsigned char* src[kMaxGatherSegs];
    unsigned char* dst[kMaxGatherSegs];
    long long row_bytes[kMaxGatherSegs];
    const int* idx; int B; long long capacity;
    // idx == NULL: row b is drawn in the kernel, uniform on [0, size), Philox4x32-10 keyed by seed, counter (b, draw)
    int* idx_out; unsigned size, seed_lo, seed_hi, draw_lo, draw_hi;
    // state != NULL: draw and size come from device memory ({draw, size, ticket, B row tickets}); the last workgroup advances draw
    unsigned long long* state;
    int n_y;              // grid rows that gather (1 + number of large keys); rows beyond them run an attached encoder pack job
};

// Copies nbytes (any alignment) with the 256 threads of the workgroup; the aligned path keeps four 16-byte loads of a thread in
// flight before the first store (a `d[i] = s[i]` loop waits for every load: 1 TB/s on K3's 39 MB, a quarter of what HBM gives).
__device__ __forceinline__ void gather_copy_row(const unsigned char* src, unsigned char* dst, long long nbytes, int tid) {
    if ((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)nbytes) & 15) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        const long long n16 = nbytes / 16;
        long long i = tid;
        for (; i + 3 * 256 < n16; i += 4 * 256) {
            const uint4 v0 = s4[i], v1 = s4[i + 256], v2 = s4[i + 512], v3 = s4[i + 768];
            d4[i] = v0; d4[i + 256] = v1; d4[i + 512] = v2; d4[i + 768] = v3;
        }
        for (; i < n16; i += 256) d4[i] = s4[i];
    } else if ((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)nbytes) & 3) == 0) {
        const unsigned* s1 = reinterpret_cast<const unsigned*>(src);
        unsigned* d1 = reinterpret_cast<unsigned*>(dst);
        for (long long i = tid; i < nbytes / 4; i += 256) d1[i] = s1[i];
    } else {
        for (long long i = tid; i < nbytes; i += 256) dst[i] = src[i];
    }
}

// grid (B, 1 + number of LARGE keys): workgroup (b, 0) copies row b of every SMALL key (row_bytes < kGatherSmall: actions, rewards,
// dones, robot state ... -- a workgroup per (row, key) for 4-byte rows was most of the launch's workgroups), workgroup (b, y > 0) the
// row of the y-th large key (the point-cloud tensors).
constexpr long long kGatherSmall = 1024;
__global__ __launch_bounds__(256) void replay_gather_kernel(const GatherParams p, const int n_segs, const PackJob pack) {
    const int b = blockIdx.x, seg = blockIdx.y;
    if (seg >= p.n_y) {                 // the encoder's re-pack riding on this launch (pcrl_encoder_pack_attach_to_gather): reads no sampled row
        const int blk = (seg - p.n_y) * (int)gridDim.x + b;
        if (blk < pack.total_blocks) encoder_pack_block(pack, blk, (int)threadIdx.x);
        return;
    }
    long long row;
    if (p.idx) {
        row = p.idx[b];
    } else {
        unsigned draw_lo = p.draw_lo, draw_hi = p.draw_hi, size = p.size;
        if (p.state) {
            const unsigned long long d = __builtin_nontemporal_load(p.state), n = __builtin_nontemporal_load(p.state + 1);
            draw_lo = (unsigned)d; draw_hi = (unsigned)(d >> 32);
            size = (unsigned)(n < 1 ? 1 : (n > (unsigned long long)p.capacity ? (unsigned long long)p.capacity : n));
        }
        uint32_t w[4];
        philox4x32_10((uint32_t)b, draw_lo, draw_hi, 0x52455055u, p.seed_lo, p.seed_hi, w);
        row = (long long)(((unsigned long long)w[0] * size) >> 32);       // multiply-shift: bias < size / 2^32
        if (seg == 0 && threadIdx.x == 0 && p.idx_out) p.idx_out[b] = (int)row;
        if (p.state) {
            // Every workgroup has read the draw count before it takes a ticket.  Tickets are two-level so that no address sees
            // more than gridDim.y + gridDim.x atomics (thousands on ONE address serialise in L2: 56 us for 2 816 workgroups):
            // the last workgroup of a row takes a global ticket, the last of those is the only workgroup left -- it advances
            // the count for the next launch; every counter is put back to zero by its last taker.
            // (No __threadfence here: on a multi-XCD part a device-scope release writes the XCD's whole L2 back -- measured
            // 50 us with the previous step's dirty lines in it.  None is needed: the barrier waits for the loads above, the
            // tickets are device-scope atomics, and the plain stores below only have to be visible to the NEXT launch.)
            __syncthreads();
            if (threadIdx.x == 0) {
                if (atomicAdd(p.state + 3 + b, 1ull) == (unsigned long long)p.n_y - 1) {
                    p.state[3 + b] = 0ull;
                    if (atomicAdd(p.state + 2, 1ull) == (unsigned long long)gridDim.x - 1) {
                        p.state[2] = 0ull;
                        p.state[0] = (((unsigned long long)draw_hi << 32) | draw_lo) + 1ull;
                    }
                }
            }
        }
    }
    row = row < 0 ? 0 : (row >= p.capacity ? p.capacity - 1 : row);       // never read outside the ring
    if (seg == 0) {                     // every small key of this row
        for (int k = 0; k < n_segs; ++k) {
            const long long nbytes = p.row_bytes[k];
            if (nbytes < kGatherSmall) gather_copy_row(p.src[k] + row * nbytes, p.dst[k] + (long long)b * nbytes, nbytes, threadIdx.x);
        }
        return;
    }
    int k = 0;                          // the seg-th large key
    for (int seen = 0; k < n_segs; ++k)
        if (p.row_bytes[k] >= kGatherSmall && ++seen == seg) break;
    const long long nbytes = p.row_bytes[k];
    gather_copy_row(p.src[k] + row * nbytes, p.dst[k] + (long long)b * nbytes, nbytes, threadIdx.x);
}

}  // namespace pcrl

using namespace pcrl;

static int gather_launch(const pcrl_gather_seg* segs, int32_t n_segs, GatherParams& p, void* stream) {
    if (!segs) return fail(PCRL_E_ARG, "NULL argument");
    if (n_segs < 0 || n_segs > kMaxGatherSegs) return fail(PCRL_E_ARG, "replay gather: at most %d keys per launch", kMaxGatherSegs);
    if (p.B < 0 || p.capacity < 1) return fail(PCRL_E_ARG, "bad batch / capacity");
    if (p.B == 0 || n_segs == 0) return PCRL_OK;
    for (int i = 0; i < n_segs; ++i) {
        if (!segs[i].src || !segs[i].dst || segs[i].row_bytes < 1) return fail(PCRL_E_ARG, "bad gather segment %d", i);
        p.src[i] = static_cast<const unsigned char*>(segs[i].src);
        p.dst[i] = static_cast<unsigned char*>(segs[i].dst);
        p.row_bytes[i] = segs[i].row_bytes;
    }
    int n_large = 0;
    for (int i = 0; i < n_segs; ++i) n_large += segs[i].row_bytes >= kGatherSmall ? 1 : 0;
    p.n_y = 1 + n_large;
    PackJob pack{};
    const int pack_rows = take_pending_pack(&pack) ? (pack.total_blocks + p.B - 1) / p.B : 0;     // this thread's pending pack job, if any
    hipLaunchKernelGGL(replay_gather_kernel, dim3(p.B, p.n_y + pack_rows), dim3(256), 0, (hipStream_t)stream, p, n_segs, pack);
    PCRL_CHECK_LAUNCH("replay_gather_kernel");
    return PCRL_OK;
}

extern "C" int pcrl_replay_sample_gather(const pcrl_gather_seg* segs, int32_t n_segs, int32_t B, int64_t size, int64_t capacity,
                                         uint64_t seed, uint64_t draw, int32_t* idx_out, void* stream) {
    if (size < 1 || size > capacity || size > 0xFFFFFFFFll) return fail(PCRL_E_ARG, "replay sample: 1 <= size <= capacity");
    GatherParams p{};
    p.idx = nullptr; p.B = B; p.capacity = capacity; p.idx_out = idx_out; p.size = (unsigned)size;
    p.seed_lo = (unsigned)seed; p.seed_hi = (unsigned)(seed >> 32); p.draw_lo = (unsigned)draw; p.draw_hi = (unsigned)(draw >> 32);
    return gather_launch(segs, n_segs, p, stream);
}

extern "C" int pcrl_replay_sample_gather_state(const pcrl_gather_seg* segs, int32_t n_segs, int32_t B, int64_t capacity, uint64_t seed,
                                               uint64_t* state, int64_t state_words, int32_t* idx_out, void* stream) {
    if (!state) return fail(PCRL_E_ARG, "NULL argument");
    if (B < 0 || state_words < 3 + (int64_t)B) return fail(PCRL_E_ARG, "replay sample: state holds %lld words, 3 + B = %lld needed",
                                                           (long long)state_words, 3ll + B);
    if (capacity < 1 || capacity > 0xFFFFFFFFll) return fail(PCRL_E_ARG, "replay sample: 1 <= capacity < 2^32");
    GatherParams p{};
    p.idx = nullptr; p.B = B; p.capacity = capacity; p.idx_out = idx_out; p.size = 1;
    p.seed_lo = (unsigned)seed; p.seed_hi = (unsigned)(seed >> 32);
    p.state = reinterpret_cast<unsigned long long*>(state);
    return gather_launch(segs, n_segs, p, stream);
}

extern "C" int pcrl_replay_gather(const pcrl_gather_seg* segs, int32_t n_segs, const int32_t* idx, int32_t B, int64_t capacity, void* stream) {
    if (!idx) return fail(PCRL_E_ARG, "NULL argument");
    GatherParams p{};
    p.idx = idx; p.B = B; p.capacity = capacity;
    return gather_launch(segs, n_segs, p, stream);
}
