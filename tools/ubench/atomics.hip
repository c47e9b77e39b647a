// Micro-benchmark behind DESIGN.md's choice of XCD-local depth atomics: throughput of scattered 64-bit atomic-min on
// MI355X at agent scope (executed at the memory side: the 8 per-XCD L2s are not coherent) versus workgroup scope
// (executed in the issuing XCD's L2), plus a correctness check of the XCD-partitioned protocol: every address is only
// ever touched from one XCD (the workgroup reads HW_REG_XCC_ID and works on that XCD's partition).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/atomics.hip -o /tmp/atomics && /tmp/atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: agent-scope min  1: workgroup-scope min  2: plain store  3: agent-scope u32 or  4: workgroup-scope u32 or
// PART: restrict the addresses to the partition of the workgroup's XCD (region / 8 each)
template <int MODE, bool PART>
__global__ void __launch_bounds__(256) scatter(u64* buf, size_t npx, int W, int quads, unsigned seed, unsigned* xcd_hist) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    const unsigned x = xcc_id();
    if (threadIdx.x == 0 && xcd_hist) atomicAdd(&xcd_hist[x], 1u);
    const size_t part = PART ? npx / 8 : npx;
    u64* base = buf + (PART ? (size_t)x * part : 0);
    for (int k = 0; k < quads; k++) {
        unsigned h = hash32(t * 31u + k * 7919u + seed);
        size_t p = (size_t)(h % (unsigned)(part - W - 2));
        u64 val = ((u64)hash32(h) << 32) | t;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            size_t a = p + (j & 1) + (j >> 1) * W;
            u64 v = val + j;
            if (MODE == 0) __hip_atomic_fetch_min(&base[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 1) __hip_atomic_fetch_min(&base[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) base[a] = v;
            if (MODE == 3) __hip_atomic_fetch_or((unsigned*)&base[a], (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 4) __hip_atomic_fetch_or((unsigned*)&base[a], (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// correctness: NQ queues of work items; a workgroup only takes items of its own XCD's queue (dynamic claim), every
// item = 256 atomic-mins (workgroup scope) into that XCD's partition.  Result must equal the CPU minimum.
__global__ void __launch_bounds__(256) part_min(u64* buf, size_t part, int items_per_q, unsigned* qhead, unsigned seed) {
    __shared__ unsigned item;
    const unsigned x = xcc_id() & 7u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) item = atomicAdd(&qhead[x], 1u);
        __syncthreads();
        const unsigned it = item;
        if (it >= (unsigned)items_per_q) break;
        const unsigned id = (x * items_per_q + it) * 256 + threadIdx.x;
        const unsigned h = hash32(id + seed);
        const size_t a = (size_t)x * part + (h % 4096u);  // heavy contention on purpose: 4096 addresses per XCD
        const u64 v = ((u64)hash32(h ^ 0x9e3779b9u) << 32) | id;
        __hip_atomic_fetch_min(&buf[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
static unsigned h_hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE, bool PART>
float run(u64* buf, size_t npx, int W, int nblk, int quads, unsigned* hist) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(buf, 0xff, npx * 8));
    scatter<MODE, PART><<<nblk, 256>>>(buf, npx, W, quads, 1u, nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        CK(hipMemset(buf, 0xff, npx * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        scatter<MODE, PART><<<nblk, 256>>>(buf, npx, W, quads, 2u + r, r == 0 ? hist : nullptr);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    return best;
}

int main() {
    const int W = 1280;
    const size_t npx = (size_t)8 * 1024 * 1024;  // 64 MB of keys
    u64* buf; CK(hipMalloc(&buf, npx * 8));
    unsigned* hist; CK(hipMalloc(&hist, 64)); CK(hipMemset(hist, 0, 64));
    const int nblk = 1288, quads = 1;  // 329k threads x 4 = 1.3 M operations, like one config-3 step
    const double nops = (double)nblk * 256 * quads * 4;
    struct { const char* name; float ms; } res[] = {
        {"agent-scope u64 min, whole buffer", run<0, false>(buf, npx, W, nblk, quads, hist)},
        {"agent-scope u64 min, XCD partition", run<0, true>(buf, npx, W, nblk, quads, nullptr)},
        {"workgroup-scope u64 min, XCD partition", run<1, true>(buf, npx, W, nblk, quads, nullptr)},
        {"workgroup-scope u64 min, whole buffer (speed only)", run<1, false>(buf, npx, W, nblk, quads, nullptr)},
        {"plain u64 store, XCD partition", run<2, true>(buf, npx, W, nblk, quads, nullptr)},
        {"agent-scope u32 or, whole buffer", run<3, false>(buf, npx, W, nblk, quads, nullptr)},
        {"workgroup-scope u32 or, XCD partition", run<4, true>(buf, npx, W, nblk, quads, nullptr)},
    };
    for (auto& r : res) printf("%-52s %8.1f us  %7.2f G op/s\n", r.name, r.ms * 1e3, nops / (r.ms * 1e-3) / 1e9);
    unsigned h[16]; CK(hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost));
    printf("workgroups per XCC_ID:");
    for (int i = 0; i < 16; i++) printf(" %u", h[i]);
    printf("\n");
    // ---- correctness of the partitioned protocol
    const size_t part = npx / 8;
    const int items = 512;
    unsigned* qhead; CK(hipMalloc(&qhead, 64));
    int bad_total = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(buf, 0xff, npx * 8));
        CK(hipMemset(qhead, 0, 64));
        part_min<<<2048, 256>>>(buf, part, items, qhead, 77u + rep);
        CK(hipDeviceSynchronize());
        std::vector<u64> got(4096), want(4096);
        unsigned qh[8]; CK(hipMemcpy(qh, qhead, 32, hipMemcpyDeviceToHost));
        for (int x = 0; x < 8; x++) {
            CK(hipMemcpy(got.data(), buf + (size_t)x * part, 4096 * 8, hipMemcpyDeviceToHost));
            std::fill(want.begin(), want.end(), ~0ull);
            for (int it = 0; it < items; it++)
                for (int tid = 0; tid < 256; tid++) {
                    unsigned id = (x * items + it) * 256 + tid;
                    unsigned hh = h_hash32(id + 77u + rep);
                    u64 v = ((u64)h_hash32(hh ^ 0x9e3779b9u) << 32) | id;
                    want[hh % 4096u] = std::min(want[hh % 4096u], v);
                }
            int bad = 0;
            for (int i = 0; i < 4096; i++) bad += got[i] != want[i];
            if (qh[x] < (unsigned)items) bad += 1000000;  // queue never drained: no workgroup ran on this XCD
            bad_total += bad;
            if (bad) printf("rep %d XCD %d: %d mismatches (queue head %u)\n", rep, x, bad, qh[x]);
        }
    }
    printf("partitioned workgroup-scope atomic-min protocol: %s\n", bad_total ? "MISMATCH" : "exact (3 x 8 XCDs x 131072 contended updates)");
    return bad_total != 0;
}
