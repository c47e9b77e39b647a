// ehr_raster_core.h -- triangle binning and LDS-tile coverage/z-test, shared by the drop-in rasterize kernel and
// the fused mask-loss kernel.
//
// Pipeline (all on one stream, no host round trip on the fused path):
//   bin_count  : one thread per (image, triangle): transform / clip / snap, atomically count the triangle into
//                every (image, tile, link) queue its pixel bounding box (+halo) touches
//   bin_alloc  : wave-aggregated bump allocation of queue storage (one atomic per wave, no scan kernel)
//   bin_fill   : same traversal as bin_count, writes triangle ids into the queues
//   tile kernel: one workgroup per 32x8-pixel tile; the tile's (+halo) depth/id buffer lives in LDS as 64-bit
//                keys (ordered z/w << 32 | triangle id) updated with ds_min_u64, so the z-test result does not
//                depend on queue order.  Micro-triangles (the median projected area is ~2 px) are rasterized by
//                one lane each; larger ones are swept by the whole 64-lane wave (ballot + readlane broadcast).
#pragma once
#include "ehr_device.h"

namespace ehr {

// ---- triangle sources --------------------------------------------------------------------------------------------

// drop-in rasterize: clip-space positions are given
struct PosSource {
    const float4* pos;
    const int32_t* tri;
    const int2* ranges;  // device [B] (start, count) or nullptr = all triangles
    int V, T, instance;
    __device__ __forceinline__ void range(int b, int& t0, int& t1) const {
        t0 = 0;
        t1 = T;
        if (ranges) {
            int2 r = ranges[b];
            t0 = max(r.x, 0);
            t1 = min(r.x + r.y, T);
        }
    }
    __device__ __forceinline__ bool fetch(int b, int t, float4 p[3], int& link) const {
        int v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2];
        link = 0;
        if ((unsigned)v0 >= (unsigned)V || (unsigned)v1 >= (unsigned)V || (unsigned)v2 >= (unsigned)V) return false;
        const float4* pb = pos + (instance ? (size_t)b * V : 0);
        p[0] = pb[v0];
        p[1] = pb[v1];
        p[2] = pb[v2];
        return true;
    }
};

// fused path: object-space vertices + one MVP per (view, link)
struct MvpSource {
    const float* verts;
    const int32_t* tri;
    const int32_t* tri_link;
    const float* mvp;  // [B, L, 16]
    int V, T, L;
    __device__ __forceinline__ void range(int, int& t0, int& t1) const {
        t0 = 0;
        t1 = T;
    }
    __device__ __forceinline__ float4 vertex(int b, int link, int v) const {
        const float* M = mvp + ((size_t)b * L + link) * 16;
        return transform_vertex(M, verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]);
    }
    __device__ __forceinline__ bool fetch(int b, int t, float4 p[3], int& link) const {
        int v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2];
        link = tri_link[t];
        if ((unsigned)v0 >= (unsigned)V || (unsigned)v1 >= (unsigned)V || (unsigned)v2 >= (unsigned)V ||
            (unsigned)link >= (unsigned)L)
            return false;
        p[0] = vertex(b, link, v0);
        p[1] = vertex(b, link, v1);
        p[2] = vertex(b, link, v2);
        return true;
    }
};

// ---- binning -----------------------------------------------------------------------------------------------------

struct BinGeom {
    int W, H, ntx, nty, nt;  // tiles per row / column / image
    int L;                   // queues per tile
};

// tile range touched by the triangle's pixel bounding box grown by HALO pixels
template <int HALO>
__device__ __forceinline__ bool tri_tile_range(const float4 p[3], int W, int H, int& tx0, int& tx1, int& ty0, int& ty1) {
    float4 q[4];
    int n = clip_near(p, q);
    int ix0 = 0x7fffffff, iy0 = 0x7fffffff, ix1 = -1, iy1 = -1;
    bool any = false;
    for (int s = 0; s + 2 < n; s++) {
        Coverage cv = setup_coverage(q[0], q[s + 1], q[s + 2], W, H);
        if (!cv.valid) continue;
        any = true;
        ix0 = min(ix0, cv.ix0);
        iy0 = min(iy0, cv.iy0);
        ix1 = max(ix1, cv.ix1);
        iy1 = max(iy1, cv.iy1);
    }
    if (!any) return false;
    tx0 = max(ix0 - HALO, 0) / EHR_TILE_W;
    tx1 = min(ix1 + HALO, W - 1) / EHR_TILE_W;
    ty0 = max(iy0 - HALO, 0) / EHR_TILE_H;
    ty1 = min(iy1 + HALO, H - 1) / EHR_TILE_H;
    return true;
}

// FILL = false: count; FILL = true: write ids.  grid.x covers triangles, grid.y = image.
template <class Src, int HALO, bool FILL>
__global__ void __launch_bounds__(256) bin_kernel(Src src, BinGeom g, int* __restrict__ counts, int* __restrict__ cursors,
                                                  const int* __restrict__ offsets, int* __restrict__ entries,
                                                  int entries_cap, int* __restrict__ meta) {
    int b = blockIdx.y;
    int t0, t1;
    src.range(b, t0, t1);
    int t = t0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= t1) return;
    float4 p[3];
    int link;
    if (!src.fetch(b, t, p, link)) return;
    int tx0, tx1, ty0, ty1;
    if (!tri_tile_range<HALO>(p, g.W, g.H, tx0, tx1, ty0, ty1)) return;
    for (int ty = ty0; ty <= ty1; ty++)
        for (int tx = tx0; tx <= tx1; tx++) {
            int key = ((b * g.nt) + ty * g.ntx + tx) * g.L + link;
            if (!FILL) {
                atomicAdd(&counts[key], 1);
            } else {
                int slot = atomicAdd(&cursors[key], 1);
                int at = offsets[key] + slot;
                if (at < entries_cap)
                    entries[at] = t;
                else
                    meta[1] = 1;  // overflow
            }
        }
}

// offsets[key] = start of the key's queue; meta[0] = total entries.  One atomic per wave.
static __global__ void __launch_bounds__(256) bin_alloc_kernel(const int* __restrict__ counts, int* __restrict__ offsets,
                                                        int nkeys, int* __restrict__ meta) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = (i < nkeys) ? counts[i] : 0;
    int lane = threadIdx.x & 63;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    int total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane == 63 && total > 0) base = atomicAdd(&meta[0], total);
    base = __shfl(base, 63, 64);
    if (i < nkeys) offsets[i] = base + incl - c;
}

// ---- LDS tile raster ---------------------------------------------------------------------------------------------

// Everything a lane needs to rasterize one (sub-)triangle into a region; broadcast with readlane for the
// cooperative path.
struct RasterJob {
    float4 p[3];  // parent clip-space vertices (depth)
    Coverage cv;  // snapped sub-triangle, bbox already clamped to the region
    int t;
};

template <class T>
__device__ __forceinline__ T bcast(T v, int src) {
    return __shfl(v, src, 64);
}
__device__ __forceinline__ float4 bcast4(float4 v, int src) {
    float4 r;
    r.x = __shfl(v.x, src, 64);
    r.y = __shfl(v.y, src, 64);
    r.z = __shfl(v.z, src, 64);
    r.w = __shfl(v.w, src, 64);
    return r;
}

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

// Rasterize this lane's triangle `t` (active lanes only) into the LDS key buffer of a RW x RH region whose lower-left
// pixel is (rx0, ry0).  Must be called by all 64 lanes of the wave (inactive lanes pass active = false).
template <int RW, int RH, int SMALL>
__device__ __forceinline__ void raster_wave(bool active, const float4 p[3], int t, int W, int H, int rx0, int ry0,
                                            u64* __restrict__ key) {
    float4 q[4];
    int n = active ? clip_near(p, q) : 0;
    Coverage cvs[2];
    unsigned big = 0;  // bit s: sub-triangle s is pending for the cooperative path
#pragma unroll
    for (int s = 0; s < 2; s++) {
        bool have = s + 2 < n;
        Coverage cv;
        cv.valid = false;
        if (have) {
            cv = setup_coverage(q[0], q[s + 1], q[s + 2], W, H);
            cv.ix0 = max(cv.ix0, rx0);
            cv.iy0 = max(cv.iy0, ry0);
            cv.ix1 = min(cv.ix1, rx0 + RW - 1);
            cv.iy1 = min(cv.iy1, ry0 + RH - 1);
            if (cv.ix0 > cv.ix1 || cv.iy0 > cv.iy1) cv.valid = false;
        }
        cvs[s] = cv;
        if (cv.valid) {
            int bw = cv.ix1 - cv.ix0 + 1, bh = cv.iy1 - cv.iy0 + 1;
            if (bw * bh <= SMALL) {
                EdgeEval ee = setup_edges(cv, cv.ix0, cv.iy0, W, H);
                for (int iy = cv.iy0; iy <= cv.iy1; iy++) {
                    i64 e0 = ee.e[0], e1 = ee.e[1], e2 = ee.e[2];
                    for (int ix = cv.ix0; ix <= cv.ix1; ix++) {
                        if ((e0 | e1 | e2) >= 0)
                            depth_test_write(p, t, ix, iy, W, H, &key[(iy - ry0) * RW + (ix - rx0)]);
                        e0 += ee.sx[0];
                        e1 += ee.sx[1];
                        e2 += ee.sx[2];
                    }
                    ee.e[0] += ee.sy[0];
                    ee.e[1] += ee.sy[1];
                    ee.e[2] += ee.sy[2];
                }
            } else {
                big |= 1u << s;
            }
        }
    }
    // cooperative sweep of the big ones: wave-uniform loop over (lane, sub-triangle)
    const int lane = lane_id();
#pragma unroll
    for (int s = 0; s < 2; s++) {
        u64 pending = __ballot((big >> s) & 1u);
        while (pending) {
            int src = __ffsll((long long)pending) - 1;
            pending &= pending - 1;
            float4 bp[3];
            bp[0] = bcast4(p[0], src);
            bp[1] = bcast4(p[1], src);
            bp[2] = bcast4(p[2], src);
            Coverage cv;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                cv.X[k] = bcast(cvs[s].X[k], src);
                cv.Y[k] = bcast(cvs[s].Y[k], src);
            }
            cv.ix0 = bcast(cvs[s].ix0, src);
            cv.ix1 = bcast(cvs[s].ix1, src);
            cv.iy0 = bcast(cvs[s].iy0, src);
            cv.iy1 = bcast(cvs[s].iy1, src);
            int bt = bcast(t, src);
            int bw = cv.ix1 - cv.ix0 + 1, bh = cv.iy1 - cv.iy0 + 1;
            EdgeEval ee = setup_edges(cv, cv.ix0, cv.iy0, W, H);
            for (int i = lane; i < bw * bh; i += 64) {
                int dy = i / bw, dx = i - dy * bw;
                i64 e0 = ee.e[0] + dx * ee.sx[0] + dy * ee.sy[0];
                i64 e1 = ee.e[1] + dx * ee.sx[1] + dy * ee.sy[1];
                i64 e2 = ee.e[2] + dx * ee.sx[2] + dy * ee.sy[2];
                if ((e0 | e1 | e2) >= 0) {
                    int ix = cv.ix0 + dx, iy = cv.iy0 + dy;
                    depth_test_write(bp, bt, ix, iy, W, H, &key[(iy - ry0) * RW + (ix - rx0)]);
                }
            }
        }
    }
}

}  // namespace ehr
