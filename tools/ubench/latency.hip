// Platform probe: effective shader clock (s_memtime ticks vs the 100 MHz s_memrealtime and vs a dependent-FMA chain)
// and the latency of a dependent global load (pointer chase) on an idle chip and with every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/latency.hip -o /tmp/latency && /tmp/latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void clocks(long long* out, int iters) {
    long long t0 = __builtin_readcyclecounter();
    long long r0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < iters; i++) x = fmaf(x, 1.0000001f, 1e-9f);
    long long t1 = __builtin_readcyclecounter();
    long long r1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (long long)x; }
}
// every lane chases its own chain: idx = next[idx]; `hops` dependent loads
__global__ void chase(const int* next, int n, int hops, long long* out, int* sink) {
    int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 997 % n;
    long long t0 = __builtin_readcyclecounter();
    for (int h = 0; h < hops; h++) idx = next[idx];
    long long t1 = __builtin_readcyclecounter();
    if (idx == -1) sink[0] = 1;
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)out, (unsigned long long)(t1 - t0));
}
int main() {
    long long* out; CK(hipMalloc(&out, 64)); int* sink; CK(hipMalloc(&sink, 4));
    long long h[3];
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const int iters = 1 << 20;
        CK(hipEventRecord(a)); clocks<<<1, 64>>>(out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
        printf("dependent fma x %d: %.3f ms wall, %lld s_memtime ticks, %lld realtime ticks (100 MHz) -> s_memtime = %.0f MHz, %.2f ticks per fma\n",
               iters, ms, h[0], h[1], h[0] / (h[1] / 100.0), (double)h[0] / iters);
    }
    const int n = 64 << 20;  // 256 MB of indices: beyond L2, inside the Infinity Cache after the first touch
    std::vector<int> nx(n);
    unsigned s = 12345;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; nx[i] = (int)((s >> 4) % n); }
    int* d; CK(hipMalloc(&d, (size_t)n * 4)); CK(hipMemcpy(d, nx.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    const int cfg[4][2] = {{1, 64}, {256, 64}, {1024, 256}, {4096, 256}};
    for (auto& c : cfg) {
        const int hops = 64;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(out, 0, 8));
            chase<<<c[0], c[1]>>>(d, n, hops, out, sink);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
        const double waves = (double)c[0] * (c[1] / 64);
        printf("pointer chase, %4d blocks x %3d threads: %.0f ticks per dependent gather (64 scattered dwords per wave)\n", c[0], c[1],
               (double)h[0] / waves / hops);
    }
    return 0;
}
