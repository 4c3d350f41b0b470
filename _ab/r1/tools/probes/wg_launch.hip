// Probe: how long does a round of (nearly) empty workgroups take on MI355X, as a function of
// workgroup size, static LDS and VGPR footprint?   hipcc --offload-arch=gfx950 -O3 wg_launch.hip -o wg_launch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int THREADS, int LDS_KB, int VG>
__global__ __launch_bounds__(THREADS) void k(float* out, int flag) {
    __shared__ float s[LDS_KB * 256 + 1];
    float v[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = threadIdx.x * 0.5f + i;
    if (flag) {   // never taken; keeps LDS and registers alive
        s[threadIdx.x] = v[0];
        __syncthreads();
        float a = 0;
#pragma unroll
        for (int i = 0; i < VG; ++i) a += v[i] * s[(threadIdx.x + i) % (LDS_KB * 256)];
        out[blockIdx.x * THREADS + threadIdx.x] = a;
    }
}
template <int THREADS, int LDS_KB, int VG>
void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s", name);
    for (int wgs : {256, 512, 544, 1024, 2048, 4096}) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<THREADS, LDS_KB, VG>), dim3(wgs), dim3(THREADS), 0, 0, out, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<THREADS, LDS_KB, VG>), dim3(wgs), dim3(THREADS), 0, 0, out, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" %5d:%6.1fus", wgs, ms * 1e3 / 200);
    }
    printf("\n");
}
int main() {
    float* out; hipMalloc(&out, 1 << 26);
    run<512, 64, 8>("512t 64KB vg8", out);
    run<512, 32, 8>("512t 32KB vg8", out);
    run<512, 1, 8>("512t 1KB vg8", out);
    run<512, 1, 120>("512t 1KB vg120", out);
    run<512, 64, 120>("512t 64KB vg120", out);
    run<256, 1, 8>("256t 1KB vg8", out);
    run<256, 32, 8>("256t 32KB vg8", out);
    run<256, 32, 120>("256t 32KB vg120", out);
    run<64, 1, 8>("64t 1KB vg8", out);
    return 0;
}
