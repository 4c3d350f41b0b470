// Probe: packed-fp32 VALU ops from ext_vector_type(2) arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* in, float* out) {
    f32x16 a;
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x * 16 + i];
    const float s = in[1000], m = in[1001];
    const f32x2 s2 = {s, s}, m2 = {m, m};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 c = f32x2{a[r], a[r + 1]} - m2;
        const f32x2 g = {in[1002 + r], in[1003 + r]}, b = {in[1020 + r], in[1021 + r]};
        const f32x2 y = __builtin_elementwise_fma(c * s2, g, b);
        a[r] = y[0]; a[r + 1] = y[1];
    }
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = a[i];
}
int main() {
    float h[2048], o[1024];
    for (int i = 0; i < 2048; ++i) h[i] = 0.37f * i - 3.0f * (i % 7);
    float *d, *e; hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) for (int r = 0; r < 16; ++r) {
        float want = __builtin_fmaf((h[t * 16 + r] - h[1001]) * h[1000], h[1002 + r], h[1020 + r]);
        if (want != o[t * 16 + r]) { if (bad < 4) printf("t%d r%d got %g want %g\n", t, r, o[t * 16 + r], want); ++bad; }
    }
    printf("mismatches: %d of 1024\n", bad);
    return 0;
}
