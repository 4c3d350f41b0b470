// Probe: cost of reading a large by-value kernel argument through dependent scalar loads.
// hipcc --offload-arch=gfx950 -O3 kernarg_latency.hip -o kernarg_latency
#include <hip/hip_runtime.h>
#include <cstdio>
struct P { const float* a; const float* b; float* c; long long s[12]; int m, n, k; int begin; int pad[18]; };   // 208 bytes
struct G { P p[4]; int n; int total; int z; };

__global__ __launch_bounds__(512) void chase(const G g, float* out) {
    __shared__ float lds[16384];
    int wg = blockIdx.x;
    if (wg >= g.total) return;
    int gi = 0;
    for (int j = 1; j < 4; ++j) if (j < g.n && wg >= g.p[j].begin) gi = j;
    const P& p = g.p[gi];
    int local = wg - p.begin;
    int nx = local % p.n, my = (local / p.n) % p.m;
    long long off = nx * p.s[0] + my * p.s[1] + p.s[2 + (nx & 7)];
    if (g.z) { lds[threadIdx.x] = off; __syncthreads(); out[wg] = lds[(threadIdx.x + 1) & 511] + p.a[off] + p.b[off] + p.k; }
}
__global__ __launch_bounds__(512) void chase_dev(const G* gp, float* out) {
    __shared__ float lds[16384];
    const G& g = *gp;
    int wg = blockIdx.x;
    if (wg >= g.total) return;
    int gi = 0;
    for (int j = 1; j < 4; ++j) if (j < g.n && wg >= g.p[j].begin) gi = j;
    const P& p = g.p[gi];
    int local = wg - p.begin;
    int nx = local % p.n, my = (local / p.n) % p.m;
    long long off = nx * p.s[0] + my * p.s[1] + p.s[2 + (nx & 7)];
    if (g.z) { lds[threadIdx.x] = off; __syncthreads(); out[wg] = lds[(threadIdx.x + 1) & 511] + p.a[off] + p.b[off] + p.k; }
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    G g{}; g.n = 4; g.z = 0;
    for (int j = 0; j < 4; ++j) { g.p[j].n = 32; g.p[j].m = 8; g.p[j].begin = 256 * j; g.p[j].a = out; g.p[j].b = out; }
    G* gd; hipMalloc(&gd, sizeof(G));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int dev = 0; dev < 2; ++dev) {
        printf(dev ? "struct in device memory: " : "struct by value (kernarg): ");
        for (int wgs : {256, 512, 544, 1024, 2048}) {
            g.total = wgs;
            hipMemcpy(gd, &g, sizeof(G), hipMemcpyHostToDevice);
            for (int i = 0; i < 220; ++i) {
                if (i == 20) { hipDeviceSynchronize(); hipEventRecord(e0); }
                if (dev) hipLaunchKernelGGL(chase_dev, dim3(wgs), dim3(512), 0, 0, gd, out);
                else hipLaunchKernelGGL(chase, dim3(wgs), dim3(512), 0, 0, g, out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf(" %5d:%6.1fus", wgs, ms * 1e3 / 200);
        }
        printf("\n");
    }
    return 0;
}
