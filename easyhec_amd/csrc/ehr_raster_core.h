// ehr_raster_core.h -- triangle binning and LDS-tile coverage/z-test, shared by the drop-in rasterize kernel and
// the fused mask-loss kernel.
//
// Pipeline (all on one stream, no host round trip on the fused path):
//   bin_count  : one thread per (image, triangle): gather the clip-space vertices, clip / snap, and count the
//                triangle into every (image, tile, link) queue its pixel bounding box (+halo) touches.  Lanes of a
//                wave that hit the same queue are merged with ballot/popcount so a hot tile costs one atomic per
//                wave, not one per triangle.
//   bin_alloc  : one thread per tile: sums the tile's per-link counts, bump-allocates queue storage with one atomic
//                per wave (no scan kernel) and appends non-empty tiles to the work list.
//   bin_fill   : same traversal as bin_count, writes {triangle, v0, v1, v2} into the queues.
//   tile kernel: the tile's (+halo) depth/id buffer lives in LDS as 64-bit keys (ordered z/w << 32 | triangle id)
//                updated with ds_min_u64, so the z-test result does not depend on queue order.  A wave takes 64
//                queued triangles, prefix-sums their clamped bounding-box areas and splits the resulting pixel
//                sequence EVENLY over its 64 lanes (each lane walks a contiguous run, stepping integer edge
//                functions); covered fragments are compacted through an LDS ring with ballot/popcount and
//                depth-tested 64 at a time, so neither the coverage walk nor the depth math runs on idle lanes.
#pragma once
#include "ehr_device.h"

namespace ehr {

// ---- triangle source: clip-space vertices + indices -------------------------------------------------------------

struct ClipSource {
    const float4* pos;        // [B or 1][V] clip-space vertices
    const int32_t* tri;       // [T][3]
    const int32_t* tri_link;  // [T] link (queue) of each triangle, or nullptr = 0
    const int2* ranges;       // device [B] (start, count) or nullptr = all triangles
    int V, T, L;
    int image_stride;         // V in instance mode, 0 when all images share the vertices (range mode)
    __device__ __forceinline__ void range(int b, int& t0, int& t1) const {
        t0 = 0;
        t1 = T;
        if (ranges) {
            int2 r = ranges[b];
            t0 = max(r.x, 0);
            t1 = min(r.x + r.y, T);
        }
    }
    __device__ __forceinline__ const float4* verts(int b) const { return pos + (size_t)b * image_stride; }
    __device__ __forceinline__ bool indices(int t, int& v0, int& v1, int& v2, int& link) const {
        v0 = tri[3 * t];
        v1 = tri[3 * t + 1];
        v2 = tri[3 * t + 2];
        link = tri_link ? tri_link[t] : 0;
        return (unsigned)v0 < (unsigned)V && (unsigned)v1 < (unsigned)V && (unsigned)v2 < (unsigned)V &&
               (unsigned)link < (unsigned)L;
    }
};

// ---- binning -----------------------------------------------------------------------------------------------------

struct BinGeom {
    int W, H, ntx, nty, nt;  // tiles per row / column / image
    int L;                   // queues per tile
};

// meta words (device int[8]) shared by the bin / tile kernels
#define EHR_META_TOTAL 0     // entries allocated
#define EHR_META_OVERFLOW 1  // sticky overflow flag
#define EHR_META_NWORK 2     // non-empty tiles appended to the work list
#define EHR_META_NWORK_SLOW 3  // ... of which tiles that hold a triangle needing the 64-bit / clipping path
#define EHR_META_SPILL 4     // visibility-buffer chain: items allocated from the spill pool
#define EHR_META_INTS 48     // ints reserved for the meta block (8 words + profiling counters)

// A triangle is "slow" when it needs near-plane clipping or spans more than this many sub-pixels: then (and only
// then) its region-relative coordinates may not fit the 14 bits the 32-bit edge functions assume.  Tiles that hold a
// slow triangle are processed by the SLOW instantiation of the tile kernel; all other tiles by the lean one.
#define EHR_FAST_EXTENT (448 * 16)

// exact coverage test of every pixel centre of a (small) bounding box
#define EHR_CULL_BOX 9
__device__ __forceinline__ bool covers_any(const Coverage& cv, int W, int H) {
    EdgeEval ee = setup_edges(cv, cv.ix0, cv.iy0, W, H);
    bool hit = false;
    for (int iy = cv.iy0; iy <= cv.iy1; iy++) {
        i64 e0 = ee.e[0], e1 = ee.e[1], e2 = ee.e[2];
        for (int ix = cv.ix0; ix <= cv.ix1; ix++) {
            hit = hit || ((e0 | e1 | e2) >= 0);
            e0 += ee.sx[0];
            e1 += ee.sx[1];
            e2 += ee.sx[2];
        }
        ee.e[0] += ee.sy[0];
        ee.e[1] += ee.sy[1];
        ee.e[2] += ee.sy[2];
    }
    return hit;
}

// tile range touched by the triangle's pixel bounding box grown by HALO pixels
template <int HALO>
__device__ __forceinline__ bool tri_tile_range(const float4 p[3], int W, int H, int& tx0, int& tx1, int& ty0, int& ty1,
                                               bool& slow) {
    const ClipPoly c = clip_near_poly(p);
    const bool simple = (p[0].w > 0.f) && (p[1].w > 0.f) && (p[2].w > 0.f) && (p[0].z + p[0].w >= 0.f) &&
                        (p[1].z + p[1].w >= 0.f) && (p[2].z + p[2].w >= 0.f);
    slow = !simple;
    int ix0 = 0x7fffffff, iy0 = 0x7fffffff, ix1 = -1, iy1 = -1;
    bool any = false;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (s + 2 < c.n) {
            Coverage cv = (s == 0) ? setup_coverage(c.q0, c.q1, c.q2, W, H) : setup_coverage(c.q0, c.q2, c.q3, W, H);
            // a triangle whose (small) pixel box holds no covered pixel centre produces no fragment anywhere: drop it
            // here instead of paying its setup in every tile pass (about a fifth of the queued triangles of a
            // 1-px-triangle mesh).  Same exact edge functions as the rasterizer, so nothing visible is lost.
            if (cv.valid && (cv.ix1 - cv.ix0 + 1) * (cv.iy1 - cv.iy0 + 1) <= EHR_CULL_BOX && !covers_any(cv, W, H))
                cv.valid = false;
            if (cv.valid) {
                any = true;
                i64 ex = (i64)max(cv.X[0], max(cv.X[1], cv.X[2])) - min(cv.X[0], min(cv.X[1], cv.X[2]));
                i64 ey = (i64)max(cv.Y[0], max(cv.Y[1], cv.Y[2])) - min(cv.Y[0], min(cv.Y[1], cv.Y[2]));
                if (ex > EHR_FAST_EXTENT || ey > EHR_FAST_EXTENT) slow = true;
                ix0 = min(ix0, cv.ix0);
                iy0 = min(iy0, cv.iy0);
                ix1 = max(ix1, cv.ix1);
                iy1 = max(iy1, cv.iy1);
            }
        }
    }
    if (!any) return false;
    tx0 = max(ix0 - HALO, 0) / EHR_TILE_W;
    tx1 = min(ix1 + HALO, W - 1) / EHR_TILE_W;
    ty0 = max(iy0 - HALO, 0) / EHR_TILE_H;
    ty1 = min(iy1 + HALO, H - 1) / EHR_TILE_H;
    return true;
}

// FILL = false: count; FILL = true: write entries.  grid.x covers triangles, grid.y = image.
//
// Device-scope atomics on one address serialise at the memory side (~12 ns each on MI355X), and a dense tile's queue
// receives thousands of triangles, so the workgroup first merges its updates in an LDS hash (key -> count, LDS atomics)
// and then issues ONE global atomic per distinct queue; each triangle keeps the rank the LDS atomic returned, which
// becomes its slot inside the range the workgroup reserved.
#define EHR_BIN_SLOTS 1024   // LDS hash slots per workgroup (256 triangles x up to EHR_BIN_LOCAL tiles each)
#define EHR_BIN_LOCAL 3      // tiles per triangle merged through LDS; further tiles use direct global atomics

template <int HALO, bool FILL>
__global__ void __launch_bounds__(256) bin_kernel(ClipSource src, BinGeom g, int* __restrict__ counts,
                                                  int* __restrict__ cursors, const int* __restrict__ offsets,
                                                  int4* __restrict__ entries, int entries_cap, int* __restrict__ meta,
                                                  int* __restrict__ tile_slow) {
    __shared__ int hkey[EHR_BIN_SLOTS];
    __shared__ int hcnt[EHR_BIN_SLOTS];  // count, then (FILL) the base of the reserved range
    for (int i = threadIdx.x; i < EHR_BIN_SLOTS; i += 256) {
        hkey[i] = -1;
        hcnt[i] = 0;
    }
    const int b = blockIdx.y;
    int t0, t1;
    src.range(b, t0, t1);
    const int t = t0 + blockIdx.x * blockDim.x + threadIdx.x;
    int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1, link = 0, v0 = 0, v1 = 0, v2 = 0;
    if (t < t1 && src.indices(t, v0, v1, v2, link)) {
        const float4* pv = src.verts(b);
        float4 p[3] = {pv[v0], pv[v1], pv[v2]};
        bool slow = false;
        if (!tri_tile_range<HALO>(p, g.W, g.H, tx0, tx1, ty0, ty1, slow)) {
            tx1 = -1;
            ty1 = -1;
        }
        if (!FILL && slow && tile_slow)  // rare: flag every tile the triangle is queued in
            for (int ty = ty0; ty <= ty1; ty++)
                for (int tx = tx0; tx <= tx1; tx++) tile_slow[b * g.nt + ty * g.ntx + tx] = 1;
    }
    const int nx = tx1 - tx0 + 1;
    const int ntile = (tx1 >= tx0 && ty1 >= ty0) ? nx * (ty1 - ty0 + 1) : 0;
    __syncthreads();
    // phase 1: merge the first EHR_BIN_LOCAL tiles of every triangle in the LDS hash
    int slot[EHR_BIN_LOCAL], rank[EHR_BIN_LOCAL];
#pragma unroll
    for (int c = 0; c < EHR_BIN_LOCAL; c++) {
        slot[c] = -1;
        rank[c] = 0;
        if (c < ntile) {
            const int ty = ty0 + c / nx, tx = tx0 + c % nx;
            const int key = ((b * g.nt) + ty * g.ntx + tx) * g.L + link;
            unsigned h = ((unsigned)key * 2654435761u) >> 22;  // 10 bits
            for (;;) {
                int cur = hkey[h];
                if (cur == -1) {
                    int prev = atomicCAS(&hkey[h], -1, key);
                    cur = (prev == -1) ? key : prev;
                }
                if (cur == key) break;
                h = (h + 1) & (EHR_BIN_SLOTS - 1);
            }
            slot[c] = (int)h;
            rank[c] = atomicAdd(&hcnt[h], 1);
        }
    }
    __syncthreads();
    // phase 2: one global atomic per distinct queue touched by this workgroup
    if (FILL) {  // returning atomics: issue all of a thread's before consuming any (one round trip, not one per slot)
        int res[EHR_BIN_SLOTS / 256];
#pragma unroll
        for (int k = 0; k < EHR_BIN_SLOTS / 256; k++) {
            const int i = threadIdx.x + 256 * k;
            const int key = hkey[i];
            res[k] = (key >= 0) ? atomicAdd(&cursors[key], hcnt[i]) : 0;
        }
#pragma unroll
        for (int k = 0; k < EHR_BIN_SLOTS / 256; k++) hcnt[threadIdx.x + 256 * k] = res[k];
    } else {
        for (int i = threadIdx.x; i < EHR_BIN_SLOTS; i += 256) {
            const int key = hkey[i];
            if (key >= 0) atomicAdd(&counts[key], hcnt[i]);
        }
    }
    if (FILL) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < EHR_BIN_LOCAL; c++) {
            if (slot[c] >= 0) {
                const int key = hkey[slot[c]];
                const int at = offsets[key] + hcnt[slot[c]] + rank[c];
                if (at < entries_cap)
                    entries[at] = make_int4(t, v0, v1, v2);
                else
                    meta[EHR_META_OVERFLOW] = 1;
            }
        }
    }
    // phase 3: the remaining tiles of large triangles, directly
    if (FILL) {  // four tiles per round trip: the slots come back from returning atomics (~3 us each under load),
                 // and a serial chain of them per large triangle set this kernel's duration
        for (int c0 = EHR_BIN_LOCAL; c0 < ntile; c0 += 4) {
            int at[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = c0 + j;
                at[j] = -1;
                if (c < ntile) {
                    const int ty = ty0 + c / nx, tx = tx0 + c % nx;
                    const int key = ((b * g.nt) + ty * g.ntx + tx) * g.L + link;
                    at[j] = offsets[key] + atomicAdd(&cursors[key], 1);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (at[j] >= 0) {
                    if (at[j] < entries_cap)
                        entries[at[j]] = make_int4(t, v0, v1, v2);
                    else
                        meta[EHR_META_OVERFLOW] = 1;
                }
            }
        }
    } else {
        for (int c = EHR_BIN_LOCAL; c < ntile; c++) {
            const int ty = ty0 + c / nx, tx = tx0 + c % nx;
            const int key = ((b * g.nt) + ty * g.ntx + tx) * g.L + link;
            atomicAdd(&counts[key], 1);
        }
    }
}

// One thread per (image, tile): offsets for the tile's L queues, per-tile total, work lists of non-empty tiles
// (worklist[0 .. nwork) = lean tiles, worklist[ntiles .. ntiles + nwork_slow) = tiles flagged in tile_slow).
// Queue storage and work-list slots are bump-allocated with ONE device atomic per workgroup and counter (same-address
// device atomics serialise at ~12 ns each, so per-wave allocation dominated this kernel on many-view launches).
static __global__ void __launch_bounds__(256) bin_alloc_kernel(const int* __restrict__ counts, int* __restrict__ offsets,
                                                               int* __restrict__ tile_total, int* __restrict__ worklist,
                                                               const int* __restrict__ tile_slow, int ntiles, int L,
                                                               int* __restrict__ meta) {
    __shared__ int wsum[3][4];   // per wave: entries, lean tiles, slow tiles
    __shared__ int bbase[3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0;
    if (i < ntiles)
        for (int l = 0; l < L; l++) c += counts[(size_t)i * L + l];
    const bool is_slow = (i < ntiles) && c > 0 && tile_slow && tile_slow[i] != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const u64 ne = __ballot(c > 0 && !is_slow);
    const u64 ns = __ballot(is_slow);
    if (lane == 63) {
        wsum[0][wave] = incl;
        wsum[1][wave] = __popcll(ne);
        wsum[2][wave] = __popcll(ns);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const int tot = (wsum[k][0] + wsum[k][1]) + (wsum[k][2] + wsum[k][3]);
        const int word = (k == 0) ? EHR_META_TOTAL : (k == 1 ? EHR_META_NWORK : EHR_META_NWORK_SLOW);
        bbase[k] = (tot > 0 && (k == 0 || worklist)) ? atomicAdd(&meta[word], tot) : 0;
    }
    __syncthreads();
    int base = bbase[0], wbase = bbase[1], sbase = bbase[2];
    for (int w = 0; w < wave; w++) {
        base += wsum[0][w];
        wbase += wsum[1][w];
        sbase += wsum[2][w];
    }
    if (i < ntiles) {
        int run = base + incl - c;
        for (int l = 0; l < L; l++) {
            offsets[(size_t)i * L + l] = run;
            run += counts[(size_t)i * L + l];
        }
        if (tile_total) tile_total[i] = c;
        if (worklist && c > 0) {
            const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            if (is_slow)
                worklist[ntiles + sbase + __popcll(ns & lt)] = i;
            else
                worklist[wbase + __popcll(ne & lt)] = i;
        }
    }
}

// ---- LDS tile raster ---------------------------------------------------------------------------------------------

__device__ __forceinline__ void depth_test_write(const float4 p[3], int t, int ix, int iy, int W, int H, u64* slot) {
    const float xs = 2.f / (float)W, xo = 1.f / (float)W - 1.f;
    const float ys = 2.f / (float)H, yo = 1.f / (float)H - 1.f;
    float fx = (float)ix * xs + xo;
    float fy = (float)iy * ys + yo;
    float a0, a1, a2;
    eval_pixel(p, fx, fy, a0, a1, a2);
    float zw = eval_zw(p, a0, a1, a2);
    if (zw >= -1.f && zw <= 1.f) atomicMin(slot, ((u64)ord_key(zw) << 32) | (unsigned)t);
}

// Slow path (rare): a triangle that needs near-plane clipping or whose snapped vertices are too far from the region
// for 32-bit edge functions.  Rasterized serially by its own lane with 64-bit edge functions.
template <int RW, int RH>
__device__ __noinline__ void raster_lane_slow(float4 pa, float4 pb, float4 pc, int t, int W, int H, int rx0, int ry0,
                                              u64* __restrict__ key) {
    const float4 p[3] = {pa, pb, pc};
    const ClipPoly c = clip_near_poly(p);
    for (int s = 0; s + 2 < c.n; s++) {
        Coverage cv = (s == 0) ? setup_coverage(c.q0, c.q1, c.q2, W, H) : setup_coverage(c.q0, c.q2, c.q3, W, H);
        if (!cv.valid) continue;
        cv.ix0 = max(cv.ix0, rx0);
        cv.iy0 = max(cv.iy0, ry0);
        cv.ix1 = min(cv.ix1, rx0 + RW - 1);
        cv.iy1 = min(cv.iy1, ry0 + RH - 1);
        if (cv.ix0 > cv.ix1 || cv.iy0 > cv.iy1) continue;
        EdgeEval ee = setup_edges(cv, cv.ix0, cv.iy0, W, H);
        for (int iy = cv.iy0; iy <= cv.iy1; iy++) {
            i64 e0 = ee.e[0], e1 = ee.e[1], e2 = ee.e[2];
            for (int ix = cv.ix0; ix <= cv.ix1; ix++) {
                if ((e0 | e1 | e2) >= 0) depth_test_write(p, t, ix, iy, W, H, &key[(iy - ry0) * RW + (ix - rx0)]);
                e0 += ee.sx[0];
                e1 += ee.sx[1];
                e2 += ee.sx[2];
            }
            ee.e[0] += ee.sy[0];
            ee.e[1] += ee.sy[1];
            ee.e[2] += ee.sy[2];
        }
    }
}

// The fast (32-bit) rasterizer handles a triangle iff it needs no near-plane clipping and its snapped vertices lie
// within +-8192 sub-pixels (512 pixels) of the region origin.  Same predicate as in raster_wave.
template <int RW, int RH>
__device__ __forceinline__ bool needs_slow_path(const float4 p[3], int W, int H, int rx0, int ry0) {
    const bool simple = (p[0].w > 0.f) && (p[1].w > 0.f) && (p[2].w > 0.f) && (p[0].z + p[0].w >= 0.f) &&
                        (p[1].z + p[1].w >= 0.f) && (p[2].z + p[2].w >= 0.f);
    if (!simple) return true;
    Coverage cv = setup_coverage(p[0], p[1], p[2], W, H);
    if (!cv.valid) return false;
    int bx0 = max(cv.ix0, rx0), by0 = max(cv.iy0, ry0);
    int bx1 = min(cv.ix1, rx0 + RW - 1), by1 = min(cv.iy1, ry0 + RH - 1);
    if (bx0 > bx1 || by0 > by1) return false;
    const int ox = 16 * rx0 + 8 - 8 * W, oy = 16 * ry0 + 8 - 8 * H;
    bool fits = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        i64 rx = (i64)cv.X[k] - ox, ry = (i64)cv.Y[k] - oy;
        fits = fits && rx >= -8192 && rx <= 8192 && ry >= -8192 && ry <= 8192;
    }
    return !fits;
}

// Deterministic block-wide exclusive offset of `cnt` items per thread; total returned through `total`.
// wave_tot: LDS int[EHR_TILE_THREADS / 64].  Contains two barriers.
__device__ __forceinline__ int block_offset(int cnt, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < EHR_TILE_THREADS / 64; w++) {
        int v = wave_tot[w];
        if (w < wave) base += v;
        tot += v;
    }
    __syncthreads();
    total = tot;
    return base + incl - cnt;
}

// LDS scratch of the balanced rasterizer (one per workgroup).
struct BlockRaster {
    int pre[EHR_TILE_THREADS + 1];        // exclusive prefix of the jobs' clamped bounding-box areas (+ total)
    unsigned xy[EHR_TILE_THREADS][3];     // snapped vertices relative to the region origin, int16 x | int16 y << 16
    unsigned box[EHR_TILE_THREADS];       // bbox inside the region: x0 | y0 << 8 | w << 16 | h << 24
    int tri[EHR_TILE_THREADS];
    float4 pf[EHR_TILE_THREADS][3];       // clip-space vertices (depth)
    unsigned frag[EHR_TILE_THREADS / 64][128];  // per-wave ring: first pixel | 4-bit coverage << 12 | job << 16
    int wave_tot[EHR_TILE_THREADS / 64];
};

#define EHR_WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Edge functions of a staged job in 32-bit arithmetic, evaluated at the job's bbox origin.
struct Edge32 {
    int e[3], sx[3], sy[3];
};

__device__ __forceinline__ Edge32 job_edges(const unsigned xy[3], int bx0, int by0) {
    Edge32 ed;
    int X[3], Y[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        X[k] = (int)(short)(xy[k] & 0xffffu);
        Y[k] = (int)(short)(xy[k] >> 16);
    }
    const int Px = 16 * bx0, Py = 16 * by0;  // region pixel (0,0) has its centre at the origin
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int j = (k == 2) ? 0 : k + 1;
        const int dX = X[j] - X[k], dY = Y[j] - Y[k];
        const bool tl = (dY < 0) || (dY == 0 && dX < 0);
        ed.e[k] = dX * (Py - Y[k]) - dY * (Px - X[k]) - (tl ? 0 : 1);
        ed.sx[k] = -16 * dY;
        ed.sy[k] = 16 * dX;
    }
    return ed;
}

// Depth-test `n` (<= 64) ring entries, one per lane.  An entry is a run of up to 4 horizontally adjacent pixels of one
// job: first region pixel (bits 0-11) | 4-bit coverage mask (12-15) | job (16-31).
template <int RW>
__device__ __forceinline__ void drain_fragments(BlockRaster* br, const unsigned* ring, int head, int n, int W, int H,
                                                int rx0, int ry0, u64* __restrict__ key) {
    const int lane = lane_id();
    if (lane < n) {
        unsigned f = ring[(head + lane) & 127];
        int pix = f & 0xfffu, j = f >> 16;
        unsigned m4 = (f >> 12) & 15u;
        float4 p[3] = {br->pf[j][0], br->pf[j][1], br->pf[j][2]};
        const int t = br->tri[j];
        int py = pix / RW, px = pix - py * RW;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (m4 & (1u << i)) depth_test_write(p, t, rx0 + px + i, ry0 + py, W, H, &key[pix + i]);
    }
}

// Rasterize one queue (n entries at `ent`) into `key` with the whole workgroup.  256 triangles per round: every thread
// sets one up (gather, snap, clamp the box to the region), a block-wide prefix sum of the box areas splits the
// concatenated pixel sequence EVENLY over all threads (so all waves finish the walk together), each thread walks its
// contiguous run stepping 32-bit edge functions, covered fragments are compacted per wave (ballot/popcount) into an
// LDS ring and depth-tested 64 at a time.  Callers put barriers around it.
#ifdef EHR_PHASE_TIMING
#define EHR_SUBPHASE(i)                                                                       \
    do {                                                                                      \
        long long now_ = __builtin_readcyclecounter();                                        \
        if (threadIdx.x == 0 && meta)                                                         \
            atomicAdd((unsigned long long*)(meta + 8) + 8 + (i), (unsigned long long)(now_ - sub_last)); \
        sub_last = now_;                                                                      \
    } while (0)
#else
#define EHR_SUBPHASE(i) do { } while (0)
#endif

// What the caller already fetched for this thread's job of round 0 (software pipelining across passes): PRE = 1 the
// queue entry ent[tid], PRE = 2 the entry and its three clip-space vertices.
struct RoundZero {
    int4 e;
    float4 p0, p1, p2;
};

template <int RW, int RH, bool SLOW, int PRE = 0>
__device__ __forceinline__ void raster_queue(const ClipSource& src, int b, const int4* __restrict__ ent, int n, int W,
                                             int H, int rx0, int ry0, u64* __restrict__ key,
                                             BlockRaster* __restrict__ br, int* __restrict__ meta,
                                             const RoundZero& pre) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4* pv = src.verts(b);
    unsigned* ring = br->frag[wave];
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int nslow = 0;
#ifdef EHR_PHASE_TIMING
    long long sub_last = __builtin_readcyclecounter();
#endif
#pragma nounroll
    for (int base = 0; base < n; base += EHR_TILE_THREADS) {  // block-uniform
        const int i = base + tid;
        const bool active = i < n;
        // ---- per-thread setup
        float4 p[3];
        int t = 0;
        bool fast = false, slow = false;
        int area = 0;
        unsigned pxy[3] = {0, 0, 0}, pbox = 0;
        if (active) {
            int4 e = (PRE >= 1 && base == 0) ? pre.e : ent[i];
            t = e.x;
            if (PRE == 2 && base == 0) {
                p[0] = pre.p0;
                p[1] = pre.p1;
                p[2] = pre.p2;
            } else {
                p[0] = pv[e.y];
                p[1] = pv[e.z];
                p[2] = pv[e.w];
            }
#ifdef EHR_PHASE_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        }
        EHR_SUBPHASE(0);
        if (active) {
            const bool simple = (p[0].w > 0.f) && (p[1].w > 0.f) && (p[2].w > 0.f) && (p[0].z + p[0].w >= 0.f) &&
                                (p[1].z + p[1].w >= 0.f) && (p[2].z + p[2].w >= 0.f);
            if (simple) {
                Coverage cv = setup_coverage(p[0], p[1], p[2], W, H);
                if (cv.valid) {
                    int bx0 = max(cv.ix0, rx0), by0 = max(cv.iy0, ry0);
                    int bx1 = min(cv.ix1, rx0 + RW - 1), by1 = min(cv.iy1, ry0 + RH - 1);
                    if (bx0 <= bx1 && by0 <= by1) {
                        // region-relative snapped coordinates must fit 14 bits for the 32-bit edge functions
                        const int ox = 16 * rx0 + 8 - 8 * W, oy = 16 * ry0 + 8 - 8 * H;
                        bool fits = true;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            i64 rx = (i64)cv.X[k] - ox, ry = (i64)cv.Y[k] - oy;
                            fits = fits && rx >= -8192 && rx <= 8192 && ry >= -8192 && ry <= 8192;
                            pxy[k] = ((unsigned)(int)rx & 0xffffu) | ((unsigned)(int)ry << 16);
                        }
                        if (fits) {
                            fast = true;
                            int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
                            area = ((bw + 3) >> 2) * bh;  // work units: runs of 4 pixels along a row
                            pbox = (unsigned)(bx0 - rx0) | ((unsigned)(by0 - ry0) << 8) | ((unsigned)bw << 16) |
                                   ((unsigned)bh << 24);
                        } else {
                            slow = true;
                        }
                    }
                }
            } else {
                slow = true;
            }
        }
        nslow += slow ? 1 : 0;
        EHR_SUBPHASE(1);
        // ---- block-wide prefix sum of the areas, stage the jobs
        int S;
        const int excl = block_offset(area, br->wave_tot, S);
        EHR_SUBPHASE(2);
        if (S == 0) continue;  // block-uniform
        br->pre[tid] = excl;
        if (tid == EHR_TILE_THREADS - 1) br->pre[EHR_TILE_THREADS] = S;
        br->xy[tid][0] = pxy[0];
        br->xy[tid][1] = pxy[1];
        br->xy[tid][2] = pxy[2];
        br->box[tid] = pbox;
        br->tri[tid] = t;
        if (fast) {
            br->pf[tid][0] = p[0];
            br->pf[tid][1] = p[1];
            br->pf[tid][2] = p[2];
        }
        __syncthreads();
        EHR_SUBPHASE(3);
        // ---- every thread walks a contiguous run of K work units; a unit = 4 horizontally adjacent pixels of a job's
        //      bounding box (the per-unit bookkeeping -- ballot, ring push, row/job advance -- is paid once per 4 tests)
        const int K = (S + EHR_TILE_THREADS - 1) / EHR_TILE_THREADS;
        const int start = tid * K, end = min(start + K, S);
        int j = 0, bw = 1, bh = 1, gw = 1, gx = 0, dy = 0, pix = 0, rowpix = 0;
        Edge32 ed;
        int er0 = 0, er1 = 0, er2 = 0;  // edge values at the start of the current row
        ed.e[0] = ed.e[1] = ed.e[2] = -1;
        ed.sx[0] = ed.sx[1] = ed.sx[2] = 0;
        ed.sy[0] = ed.sy[1] = ed.sy[2] = 0;
        if (start < end) {
            int lo = 0, hi = EHR_TILE_THREADS - 1;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                int mid = (lo + hi + 1) >> 1;
                if (br->pre[mid] <= start)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            j = lo;
            unsigned bx = br->box[j];
            int bx0 = bx & 255, by0 = (bx >> 8) & 255;
            bw = (bx >> 16) & 255;
            bh = bx >> 24;
            gw = (bw + 3) >> 2;
            unsigned xy[3] = {br->xy[j][0], br->xy[j][1], br->xy[j][2]};
            ed = job_edges(xy, bx0, by0);
            int o = start - br->pre[j];
            dy = o / gw;
            gx = o - dy * gw;
            er0 = ed.e[0] + dy * ed.sy[0];
            er1 = ed.e[1] + dy * ed.sy[1];
            er2 = ed.e[2] + dy * ed.sy[2];
            ed.e[0] = er0 + 4 * gx * ed.sx[0];
            ed.e[1] = er1 + 4 * gx * ed.sx[1];
            ed.e[2] = er2 + 4 * gx * ed.sx[2];
            rowpix = (by0 + dy) * RW + bx0;
            pix = rowpix + 4 * gx;
        }
        int qhead = 0, qcount = 0;
        EHR_SUBPHASE(4);
#pragma nounroll
        for (int it = 0; it < K; it++) {
            const bool act = start + it < end;
            unsigned m4 = 0;
            if (act) {
                const int a1 = ed.e[0] + ed.sx[0], a2 = a1 + ed.sx[0], a3 = a2 + ed.sx[0];
                const int b1 = ed.e[1] + ed.sx[1], b2 = b1 + ed.sx[1], b3 = b2 + ed.sx[1];
                const int c1 = ed.e[2] + ed.sx[2], c2 = c1 + ed.sx[2], c3 = c2 + ed.sx[2];
                m4 = ((ed.e[0] | ed.e[1] | ed.e[2]) >= 0 ? 1u : 0u) | ((a1 | b1 | c1) >= 0 ? 2u : 0u) |
                     ((a2 | b2 | c2) >= 0 ? 4u : 0u) | ((a3 | b3 | c3) >= 0 ? 8u : 0u);
                const int rem = bw - 4 * gx;  // pixels of this unit that lie inside the bounding box
                if (rem < 4) m4 &= (1u << rem) - 1u;
            }
            const bool inside = m4 != 0;
            const u64 m = __ballot(inside);
            if (m) {
                if (inside)
                    ring[(qhead + qcount + __popcll(m & lt)) & 127] = (unsigned)pix | (m4 << 12) | ((unsigned)j << 16);
                qcount += __popcll(m);
                if (qcount >= 64) {
                    EHR_WAVE_LDS_FENCE();
                    drain_fragments<RW>(br, ring, qhead, 64, W, H, rx0, ry0, key);
                    qhead = (qhead + 64) & 127;
                    qcount -= 64;
                }
            }
            if (act && start + it + 1 < end) {
                gx++;
                pix += 4;
                ed.e[0] += 4 * ed.sx[0];
                ed.e[1] += 4 * ed.sx[1];
                ed.e[2] += 4 * ed.sx[2];
                if (gx == gw) {
                    gx = 0;
                    dy++;
                    rowpix += RW;
                    pix = rowpix;
                    er0 += ed.sy[0];
                    er1 += ed.sy[1];
                    er2 += ed.sy[2];
                    ed.e[0] = er0;
                    ed.e[1] = er1;
                    ed.e[2] = er2;
                    if (dy == bh) {  // next job with a non-empty box
                        do {
                            j++;
                        } while (br->pre[j + 1] == br->pre[j]);
                        unsigned bx = br->box[j];
                        int bx0 = bx & 255, by0 = (bx >> 8) & 255;
                        bw = (bx >> 16) & 255;
                        bh = bx >> 24;
                        gw = (bw + 3) >> 2;
                        unsigned xy[3] = {br->xy[j][0], br->xy[j][1], br->xy[j][2]};
                        ed = job_edges(xy, bx0, by0);
                        er0 = ed.e[0];
                        er1 = ed.e[1];
                        er2 = ed.e[2];
                        dy = 0;
                        rowpix = by0 * RW + bx0;
                        pix = rowpix;
                    }
                }
            }
        }
        EHR_SUBPHASE(5);
        if (qcount) {
            EHR_WAVE_LDS_FENCE();
            drain_fragments<RW>(br, ring, qhead, qcount, W, H, rx0, ry0, key);
        }
        __syncthreads();  // the job table is rewritten by the next round
        EHR_SUBPHASE(6);
    }
    if (!SLOW) {
        // tiles holding slow triangles are routed to the SLOW instantiation; if one shows up here the routing
        // predicate is broken -- report it (loss = NaN) rather than drop the triangle silently
        if (nslow && meta) meta[EHR_META_OVERFLOW] = 1;
    } else if (__syncthreads_or(nslow)) {  // rare: second sweep, slow triangles only, one per thread
#pragma nounroll
        for (int base = 0; base < n; base += EHR_TILE_THREADS) {
            const int i = base + tid;
            if (i < n) {
                int4 e = ent[i];
                float4 p[3] = {pv[e.y], pv[e.z], pv[e.w]};
                if (needs_slow_path<RW, RH>(p, W, H, rx0, ry0)) raster_lane_slow<RW, RH>(p[0], p[1], p[2], e.x, W, H, rx0, ry0, key);
            }
        }
    }
}

}  // namespace ehr
