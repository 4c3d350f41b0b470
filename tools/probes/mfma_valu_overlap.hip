// Probe: do MFMAs of one wave and VALU work of ANOTHER wave on the same SIMD overlap on gfx950?
// Workgroup = 8 waves; waves w and w+4 share SIMD w%4.  mode bit0: waves 0-3 run an MFMA stream (2 independent
// accumulators), bit1: waves 4-7 run a VALU stream (kind 0: v_fma chains, 1: DPP max chains, 2: LDS reads, 3: integer max
// chains (v_max_i32), 4: FP compare + select chains (v_cmp_nge_f32 + v_cndmask), 5: packed v_pk_fma_f32 chains).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int kind) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 512] = 1.0f;
    __syncthreads();
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 a0, a1;
            for (int i = 0; i < 16; ++i) { a0[i] = lane; a1[i] = -lane; }
            float x = lane * 0.001f, y = 1.0f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                }
            }
            r = a0[0] + a1[5];
        }
    } else if (mode & 2) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = lane + i;
        for (int it = 0; it < iters; ++it) {
            if (kind == 0) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            } else if (kind == 1) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        unsigned m = __builtin_bit_cast(unsigned, v[i]);
                        unsigned d = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, false);
                        v[i] = __builtin_bit_cast(float, m > d ? m : d + 1);
                    }
            } else if (kind == 2) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] += lds[(lane * 4 + i * 64 + u) & 4095];
            } else if (kind == 3) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        int m = __builtin_bit_cast(int, v[i]);
                        asm volatile("v_max_i32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(u + it));
                        v[i] = __builtin_bit_cast(float, m);
                    }
            } else if (kind == 4) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float t = v[i];
                        asm volatile("v_cmp_nge_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %1, vcc" : "=v"(t) : "v"(t), "v"(0.5f * u), "v"(1.0f) : "vcc");
                        v[i] = t;
                    }
            } else {
                typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        f2 t = {v[i], v[i + 1]};
                        t = __builtin_elementwise_fma(t, f2{1.0001f, 1.0001f}, f2{0.5f, 0.5f});
                        v[i] = t[0]; v[i + 1] = t[1];
                    }
            }
        }
        for (int i = 0; i < 16; ++i) r += v[i];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}
int main() {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char* kinds[] = {"v_fma", "dpp_max", "lds_read", "v_max_i32", "cmp+cndmask", "v_pk_fma"};
    for (int kind = 0; kind < 6; ++kind)
        for (int mode = 1; mode <= 3; ++mode) {
            if (kind > 0 && mode == 1) continue;
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 10, mode, kind);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode, kind);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-9s mode=%d (%s%s): %8.1f us   (MFMA stream alone = %d MFMAs x 64 cyc = %.1f us at 2.4 GHz)\n", kinds[kind], mode,
                   mode & 1 ? "MFMA " : "", mode & 2 ? "VALU" : "", ms * 1e3, iters * 16, iters * 16 * 64 / 2400.0);
        }
    return 0;
}
