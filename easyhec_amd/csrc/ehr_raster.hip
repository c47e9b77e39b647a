// ehr_raster.hip -- drop-in dr.rasterize forward/backward (nvdiffrast_renderer.py:39) + context management.
#include <stdarg.h>

#include <string.h>

#include <algorithm>

#include "ehr_host.h"
#include "ehr_raster_core.h"
#include "ehr_pose_core.h"  // Dual<N>: forward-mode scalars (the rast_db backward below)

namespace ehr {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int Scratch::reserve(size_t bytes) {
    if (bytes <= cap) return EHR_OK;
    if (pinned && ptr)
        return fail(EHR_ERR_INVALID, "this context's rasterizer scratch (%zu bytes) is referenced by a captured graph and cannot grow "
                    "to %zu bytes: render other shapes through a RasterizeCudaContext of their own (or capture again on a fresh one)",
                    cap, bytes);
    moves++;
    if (ptr) {
        EHR_HIP(hipDeviceSynchronize());
        EHR_HIP(hipFree(ptr));
        ptr = nullptr;
        cap = 0;
    }
    size_t want = std::max(bytes, (size_t)4096);
    EHR_HIP(hipMalloc(&ptr, want));
    cap = want;
    return EHR_OK;
}

void Scratch::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

static __global__ void __launch_bounds__(256) zero_words_kernel(unsigned* __restrict__ p, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
static __global__ void __launch_bounds__(256) copy_words_kernel(unsigned* __restrict__ d, const unsigned* __restrict__ s, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
static __global__ void __launch_bounds__(256) copy_words4_kernel(uint4* __restrict__ d, const uint4* __restrict__ s, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
int fill_words(void* dst, size_t nwords, unsigned value, hipStream_t stream) {
    if (nwords == 0) return EHR_OK;
    const unsigned grid = (unsigned)std::min<size_t>((nwords + 255) / 256, 2048);
    zero_words_kernel<<<grid, 256, 0, stream>>>((unsigned*)dst, nwords, value);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}
int zero_words(void* dst, size_t nwords, hipStream_t stream) { return fill_words(dst, nwords, 0u, stream); }
int copy_words(void* dst, const void* src, size_t nwords, hipStream_t stream) {
    if (nwords == 0) return EHR_OK;
    if ((nwords & 3) == 0 && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const size_t n4 = nwords >> 2;
        copy_words4_kernel<<<(unsigned)std::min<size_t>((n4 + 255) / 256, 4096), 256, 0, stream>>>((uint4*)dst, (const uint4*)src, n4);
    } else {
        copy_words_kernel<<<(unsigned)std::min<size_t>((nwords + 255) / 256, 4096), 256, 0, stream>>>((unsigned*)dst, (const unsigned*)src, nwords);
    }
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

// One pixel's output from its winning key (depth | triangle): barycentrics, depth, id [, pixel differentials].
template <bool WITH_DB>
__device__ __forceinline__ void shade_pixel(const ClipSource& src, int b, u64 k, int ix, int iy, int W, int H,
                                            float4* __restrict__ rast, float4* __restrict__ rast_db) {
    const size_t pix = ((size_t)b * H + iy) * W + ix;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f), db = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k != ~0ull) {
        int t = (int)(unsigned)(k & 0xffffffffu);
        const float4* pv = src.verts(b);
        float4 p[3] = {pv[src.tri[3 * t]], pv[src.tri[3 * t + 1]], pv[src.tri[3 * t + 2]]};
        const float xs = 2.f / (float)W, xo = 1.f / (float)W - 1.f;
        const float ys = 2.f / (float)H, yo = 1.f / (float)H - 1.f;
        float fx = (float)ix * xs + xo;
        float fy = (float)iy * ys + yo;
        float a0, a1, a2;
        eval_pixel(p, fx, fy, a0, a1, a2);
        float at = (a0 + a1) + a2;
        float iw = 1.f / at;
        float b0 = sat01(a0 * iw);
        float b1 = sat01(a1 * iw);
        float zw = eval_zw(p, a0, a1, a2);
        zw = fmaxf(fminf(zw, 1.f), -1.f);
        out = make_float4(b0, b1, zw, tri_to_float(t + 1));
        if (WITH_DB) {
            float dfxdx = xs * iw;
            float dfydy = ys * iw;
            float da0dx = p[2].y * p[1].w - p[1].y * p[2].w;
            float da0dy = p[1].x * p[2].w - p[2].x * p[1].w;
            float da1dx = p[0].y * p[2].w - p[2].y * p[0].w;
            float da1dy = p[2].x * p[0].w - p[0].x * p[2].w;
            float da2dx = p[1].y * p[0].w - p[0].y * p[1].w;
            float da2dy = p[0].x * p[1].w - p[1].x * p[0].w;
            float datdx = (da0dx + da1dx) + da2dx;
            float datdy = (da0dy + da1dy) + da2dy;
            db.x = dfxdx * (b0 * datdx - da0dx);
            db.y = dfydy * (b0 * datdy - da0dy);
            db.z = dfxdx * (b1 * datdx - da1dx);
            db.w = dfydy * (b1 * datdy - da1dy);
        }
    }
    rast[pix] = out;
    if (WITH_DB) rast_db[pix] = db;
}

// ---- drop-in rasterize tile kernel -------------------------------------------------------------------------------

// One workgroup per (image, tile).  LDS: 256 x 8-byte keys + the waves' raster scratch.  Writes rast (and rast_db)
// for every pixel of the tile.
template <bool WITH_DB>
__global__ void __launch_bounds__(EHR_TILE_THREADS) raster_tile_kernel(ClipSource src, BinGeom g,
                                                                      int* __restrict__ counts,
                                                                      const int* __restrict__ offsets,
                                                                      const int4* __restrict__ entries, int entries_cap,
                                                                      float4* __restrict__ rast,
                                                                      float4* __restrict__ rast_db,
                                                                      int* __restrict__ meta) {
    __shared__ u64 key[EHR_TILE_W * EHR_TILE_H];
    __shared__ BlockRaster wscratch;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int tx = tile % g.ntx, ty = tile / g.ntx;
    const int rx0 = tx * EHR_TILE_W, ry0 = ty * EHR_TILE_H;
    const int tid = threadIdx.x;
    key[tid] = ~0ull;
    const int kidx = b * g.nt + tile;
    const int n = counts[kidx];
    const int off = offsets[kidx];
    __syncthreads();
    // this launch is the last reader of the call's counters: leave them zero for the next call (its own queue's count
    // and fill cursor; workgroup 0 the meta words), so that no call has to start with a fill kernel
    if (tid == 0) {
        counts[kidx] = 0;
        counts[gridDim.x * gridDim.y + kidx] = 0;  // cursors = counts + nkeys
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid < 8) meta[tid] = 0;
    if (off + n <= entries_cap) {
        if (n > 0) raster_queue<EHR_TILE_W, EHR_TILE_H, true>(src, b, entries + off, n, g.W, g.H, rx0, ry0, key, &wscratch, nullptr, RoundZero());
    } else {
        // This tile's queue did not fit the queue storage (the host skipped its size read-back because the previous frames of
        // this shape needed far less, or the call is being replayed from a captured graph): the tile finds its triangles
        // itself -- every triangle of the image is tested against the tile, 256 at a time, with the binning pass's own test,
        // and the hits go through the same rasterizer from a list in LDS.  The depth test is order independent, so the
        // result is the one the queue would have given, bit for bit; only slower (the host grows the storage as soon as it
        // sees the size: never an incomplete or NaN image).
        __shared__ int4 bf_ent[EHR_TILE_THREADS];
        __shared__ int bf_n;
        int t0, t1;
        src.range(b, t0, t1);
        for (int base = t0; base < t1; base += EHR_TILE_THREADS) {
            if (tid == 0) bf_n = 0;
            __syncthreads();
            const int t = base + tid;
            int v0 = 0, v1 = 0, v2 = 0, link = 0;
            if (t < t1 && src.indices(t, v0, v1, v2, link)) {
                const float4* pv = src.verts(b);
                const float4 p[3] = {pv[v0], pv[v1], pv[v2]};
                int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
                bool slow = false;
                if (tri_tile_range<0>(p, g.W, g.H, tx0, tx1, ty0, ty1, slow) && tx >= tx0 && tx <= tx1 && ty >= ty0 && ty <= ty1)
                    bf_ent[atomicAdd(&bf_n, 1)] = make_int4(t, v0, v1, v2);
            }
            __syncthreads();
            const int m = bf_n;
            if (m > 0) raster_queue<EHR_TILE_W, EHR_TILE_H, true>(src, b, bf_ent, m, g.W, g.H, rx0, ry0, key, &wscratch, nullptr, RoundZero());
            __syncthreads();
        }
    }
    __syncthreads();
    // shade: one thread per pixel
    const int lx = tid % EHR_TILE_W, ly = tid / EHR_TILE_W;
    const int ix = rx0 + lx, iy = ry0 + ly;
    if (ix >= g.W || iy >= g.H) return;
    shade_pixel<WITH_DB>(src, b, key[tid], ix, iy, g.W, g.H, rast, rast_db);
}

// ---- drop-in rasterize, direct form: small launches (a link's mesh in one image) ----------------------------------
//
// The queued form above is four dependent launches (count, allocate, fill, one workgroup per tile) plus 16 B per
// (triangle, tile) of queue traffic: right for a batch of views of a whole robot, ~40 us of mostly latency for the few
// thousand small triangles of ONE link in ONE 1280x720 image -- which is what the reference's schedule asks for 64 times
// per optimisation step (rb_solver.py:60-66).  The direct form is two launches: triangles are depth-tested straight
// into a 64-bit key image in global memory (the same key, the same per-pixel test, the same snapped edge functions as
// the tile rasterizer: depth_test_write / setup_coverage / setup_edges, so the result is the same bit for bit; the
// minimum is order independent), then one thread per pixel shades its key and re-arms it for the next call.
//   grid = (256 triangles, band of rows, image): every workgroup clips its triangles' boxes to its band, so a triangle
//   that covers the screen is spread over all bands and a small one costs the others a three-division row test each.
//   A lane walks a box of up to RD_OWN pixels itself; larger ones are staged in LDS with their finished setup and
//   walked by a whole wave, 8 x 8 pixels per step.
constexpr int RD_OWN = 96;     // the four lanes of a triangle walk a (band-clipped) box of up to this many pixels themselves
constexpr int RD_LIST = 128;   // larger boxes staged in LDS per workgroup, walked by a wave each, 8 x 8 pixels per step

struct RdEntry {  // one (sub-)triangle's finished setup: edge functions at its box origin, the box, the parent triangle
    i64 e[3], sx[3], sy[3];
    int ix0, iy0, bw, bh;
    float4 p[3];
    int t, pad[3];
};

// own lane: rows x columns, stepping the edge functions (adds only)
__device__ __forceinline__ void rd_walk_own(const float4 p[3], const EdgeEval& ee0, int ix0, int iy0, int bw, int bh, int t,
                                            int W, int H, u64* __restrict__ keyb, int q) {
    // lane q of the triangle's four takes rows q, q + 4, ...
    i64 r0 = ee0.e[0] + q * ee0.sy[0], r1 = ee0.e[1] + q * ee0.sy[1], r2 = ee0.e[2] + q * ee0.sy[2];
    for (int dy = q; dy < bh; dy += 4) {
        i64 e0 = r0, e1 = r1, e2 = r2;
        for (int dx = 0; dx < bw; dx++) {
            if ((e0 | e1 | e2) >= 0) depth_test_write(p, t, ix0 + dx, iy0 + dy, W, H, &keyb[(size_t)(iy0 + dy) * W + ix0 + dx]);
            e0 += ee0.sx[0];
            e1 += ee0.sx[1];
            e2 += ee0.sx[2];
        }
        r0 += 4 * ee0.sy[0];
        r1 += 4 * ee0.sy[1];
        r2 += 4 * ee0.sy[2];
    }
}

__global__ void __launch_bounds__(256) raster_direct_kernel(ClipSource src, int W, int H, int band_rows,
                                                            u64* __restrict__ key) {
    __shared__ RdEntry lst[RD_LIST];  // (RD_LIST >= 128: 64 triangles, two sub-triangles each at most)
    __shared__ int ln;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * band_rows, y1 = min(y0 + band_rows, H) - 1;
    if (tid == 0) ln = 0;
    __syncthreads();
    int t0, t1;
    src.range(b, t0, t1);
    // FOUR lanes per triangle (64 triangles per workgroup): the chip is far from full on a launch of a few thousand
    // triangles, what counts is the length of a wave's instruction chain -- the setup is computed four times over, the
    // rows of a box are dealt to the four lanes, and a wave has a quarter of the staged boxes to walk
    const int q = tid & 3;
    const int t = t0 + blockIdx.x * 64 + (tid >> 2);
    u64* const keyb = key + (size_t)b * W * H;
    int v0 = 0, v1 = 0, v2 = 0, link = 0;
    if (t < t1 && src.indices(t, v0, v1, v2, link)) {
        const float4* pv = src.verts(b);
        const float4 p[3] = {pv[v0], pv[v1], pv[v2]};
        // cheap band test first (every band's workgroup sees every triangle): rows of the three vertices, one pixel of
        // margin for the snapping; only for triangles in front of the eye, the others go through the clipper below
        bool maybe = true;
        if (p[0].w > 0.f && p[1].w > 0.f && p[2].w > 0.f) {
            const float hh = 0.5f * (float)H;
            const float r0 = p[0].y / p[0].w * hh + hh, r1 = p[1].y / p[1].w * hh + hh, r2 = p[2].y / p[2].w * hh + hh;
            const float lo = fminf(r0, fminf(r1, r2)), hi = fmaxf(r0, fmaxf(r1, r2));
            maybe = !(hi < (float)y0 - 1.f) && !(lo > (float)y1 + 2.f);  // (NaN: stays true)
        }
        if (maybe) {
            const ClipPoly c = clip_near_poly(p);
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (s + 2 >= c.n) continue;
                const Coverage cv = (s == 0) ? setup_coverage(c.q0, c.q1, c.q2, W, H) : setup_coverage(c.q0, c.q2, c.q3, W, H);
                if (!cv.valid) continue;
                const int iy0 = max(cv.iy0, y0), iy1 = min(cv.iy1, y1);
                if (iy0 > iy1) continue;
                const EdgeEval ee = setup_edges(cv, cv.ix0, iy0, W, H);
                const int bw = cv.ix1 - cv.ix0 + 1, bh = iy1 - iy0 + 1;
                int k = -1;
                if (bw * bh > RD_OWN) {
                    if (q == 0) k = atomicAdd(&ln, 1);
                    k = __shfl(k, lane & ~3, 64);
                    if (k >= RD_LIST) k = -1;  // list full: walked here after all (exact, only slower)
                }
                if (k >= 0 && q != 0) continue;  // (staged by lane 0 of the four)
                if (k >= 0) {
                    RdEntry& r = lst[k];
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        r.e[i] = ee.e[i];
                        r.sx[i] = ee.sx[i];
                        r.sy[i] = ee.sy[i];
                        r.p[i] = p[i];
                    }
                    r.ix0 = cv.ix0;
                    r.iy0 = iy0;
                    r.bw = bw;
                    r.bh = bh;
                    r.t = t;
                } else {
                    rd_walk_own(p, ee, cv.ix0, iy0, bw, bh, t, W, H, keyb, q);
                }
            }
        }
    }
    __syncthreads();
    const int n = min(ln, RD_LIST);
    const int lx = lane & 7, ly = lane >> 3;
    for (int j = wave; j < n; j += 4) {
        const RdEntry& r = lst[j];
        const float4 p[3] = {r.p[0], r.p[1], r.p[2]};
        const int t2 = r.t, ix0 = r.ix0, iy0 = r.iy0, bw = r.bw, bh = r.bh;
        const i64 sx0 = r.sx[0], sx1 = r.sx[1], sx2 = r.sx[2], sy0 = r.sy[0], sy1 = r.sy[1], sy2 = r.sy[2];
        i64 r0 = r.e[0] + lx * sx0 + ly * sy0, r1 = r.e[1] + lx * sx1 + ly * sy1, r2 = r.e[2] + lx * sx2 + ly * sy2;
        for (int cy = 0; cy < bh; cy += 8) {
            i64 e0 = r0, e1 = r1, e2 = r2;
            for (int cx = 0; cx < bw; cx += 8) {
                const int dx = cx + lx, dy = cy + ly;
                if (dx < bw && dy < bh && (e0 | e1 | e2) >= 0)
                    depth_test_write(p, t2, ix0 + dx, iy0 + dy, W, H, &keyb[(size_t)(iy0 + dy) * W + ix0 + dx]);
                e0 += 8 * sx0;
                e1 += 8 * sx1;
                e2 += 8 * sx2;
            }
            r0 += 8 * sy0;
            r1 += 8 * sy1;
            r2 += 8 * sy2;
        }
    }
}

// One workgroup per (image, 32 x 8 tile) -- a row of the tile is 512 contiguous bytes of `rast` --, so that the tile's
// flag (does any pixel hold a triangle?) is one __syncthreads_or and one byte store: nothing to initialise, no atomics.
template <bool WITH_DB>
__global__ void __launch_bounds__(256) raster_shade_kernel(ClipSource src, int B, int W, int H, u64* __restrict__ key,
                                                           float4* __restrict__ rast, float4* __restrict__ rast_db,
                                                           unsigned char* __restrict__ flags) {
    const int ntx = flag_ntx(W), nty = flag_nty(H);
    const int b = blockIdx.y, tile = blockIdx.x;
    const int ix = (tile % ntx) * EHR_FLAG_TW + (int)(threadIdx.x % EHR_FLAG_TW);
    const int iy = (tile / ntx) * EHR_FLAG_TH + (int)(threadIdx.x / EHR_FLAG_TW);
    bool drawn = false;
    if (ix < W && iy < H) {
        const size_t idx = ((size_t)b * H + iy) * W + ix;
        const u64 k = key[idx];
        drawn = k != ~0ull;
        if (drawn) key[idx] = ~0ull;  // re-armed for the next call: the key image is all ones between calls
        shade_pixel<WITH_DB>(src, b, k, ix, iy, W, H, rast, rast_db);
    }
    if (flags) {  // (kernel-uniform)
        const int any = __syncthreads_or(drawn ? 1 : 0);
        if (threadIdx.x == 0) flags[((size_t)b * nty) * ntx + tile] = any ? 1 : 0;
    }
}

// ---- rasterize backward ------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) raster_grad_kernel(const float4* __restrict__ pos, const int32_t* __restrict__ tri,
                                                          const float4* __restrict__ rast,
                                                          const float4* __restrict__ dy, int range_mode, int B, int V,
                                                          int T, int H, int W, float* __restrict__ grad_pos,
                                                          const unsigned char* __restrict__ flags) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t P = (size_t)H * W;
    if (idx >= P * B) return;
    int b = (int)(idx / P);
    int rem = (int)(idx - (size_t)b * P);
    int iy = rem / W, ix = rem - iy * W;
    if (!tile_occupied(flags, b, ix, iy, W, H)) return;  // nothing drawn in this tile: neither rast nor dy is read
    float4 r = rast[idx];
    int t = float_to_tri(r.w) - 1;
    if (t < 0 || t >= T) return;
    float4 g = dy[idx];
    if (g.x == 0.f && g.y == 0.f) return;
    int vi0 = tri[3 * t], vi1 = tri[3 * t + 1], vi2 = tri[3 * t + 2];
    if ((unsigned)vi0 >= (unsigned)V || (unsigned)vi1 >= (unsigned)V || (unsigned)vi2 >= (unsigned)V) return;
    size_t voff = range_mode ? 0 : (size_t)b * V;
    float4 p0 = pos[voff + vi0], p1 = pos[voff + vi1], p2 = pos[voff + vi2];
    const float xs = 2.f / (float)W, xo = 1.f / (float)W - 1.f;
    const float ys = 2.f / (float)H, yo = 1.f / (float)H - 1.f;
    float fx = (float)ix * xs + xo;
    float fy = (float)iy * ys + yo;
    float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    float a0 = p1x * p2y - p1y * p2x;
    float a1 = p2x * p0y - p2y * p0x;
    float a2 = p0x * p1y - p0y * p1x;
    float at = (a0 + a1) + a2;
    float ep = copysignf(1e-6f, at);
    float iw = 1.f / (at + ep);
    float b0 = a0 * iw, b1 = a1 * iw;
    float gb0 = g.x * iw, gb1 = g.y * iw;
    float gbb = gb0 * b0 + gb1 * b1;
    float gp0x = gbb * (p2y - p1y) - gb1 * p2y;
    float gp1x = gbb * (p0y - p2y) + gb0 * p2y;
    float gp2x = gbb * (p1y - p0y) - gb0 * p1y + gb1 * p0y;
    float gp0y = gbb * (p1x - p2x) + gb1 * p2x;
    float gp1y = gbb * (p2x - p0x) - gb0 * p2x;
    float gp2y = gbb * (p0x - p1x) + gb0 * p1x - gb1 * p0x;
    float gp0w = -fx * gp0x - fy * gp0y;
    float gp1w = -fx * gp1x - fy * gp1y;
    float gp2w = -fx * gp2x - fy * gp2y;
    float* gp = grad_pos + 4 * voff;
    atomicAdd(&gp[4 * vi0 + 0], gp0x); atomicAdd(&gp[4 * vi0 + 1], gp0y); atomicAdd(&gp[4 * vi0 + 3], gp0w);
    atomicAdd(&gp[4 * vi1 + 0], gp1x); atomicAdd(&gp[4 * vi1 + 1], gp1y); atomicAdd(&gp[4 * vi1 + 3], gp1w);
    atomicAdd(&gp[4 * vi2 + 0], gp2x); atomicAdd(&gp[4 * vi2 + 1], gp2y); atomicAdd(&gp[4 * vi2 + 3], gp2w);
}

// d(rast_db)/d(pos) contracted with ddb: the shading's expressions for (du/dX, du/dY, dv/dX, dv/dY) (shade_pixel) are
// re-evaluated in forward-mode arithmetic over the nine inputs (x, y, w of the triangle's vertices); like the (u, v) half
// above, the barycentrics' clamp is not differentiated.  EasyHeC discards rast_db; this completes the op.
__global__ void __launch_bounds__(256) raster_grad_db_kernel(const float4* __restrict__ pos, const int32_t* __restrict__ tri,
                                                             const float4* __restrict__ rast,
                                                             const float4* __restrict__ ddb, int range_mode, int B, int V,
                                                             int T, int H, int W, float* __restrict__ grad_pos) {
    typedef Dual<9> D9;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t P = (size_t)H * W;
    if (idx >= P * B) return;
    const int b = (int)(idx / P);
    const int rem = (int)(idx - (size_t)b * P);
    const int iy = rem / W, ix = rem - iy * W;
    const int t = float_to_tri(rast[idx].w) - 1;
    if (t < 0 || t >= T) return;
    const float4 g4 = ddb[idx];
    if (g4.x == 0.f && g4.y == 0.f && g4.z == 0.f && g4.w == 0.f) return;
    const int vi[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    if ((unsigned)vi[0] >= (unsigned)V || (unsigned)vi[1] >= (unsigned)V || (unsigned)vi[2] >= (unsigned)V) return;
    const size_t voff = range_mode ? 0 : (size_t)b * V;
    const float xs = 2.f / (float)W, xo = 1.f / (float)W - 1.f;
    const float ys = 2.f / (float)H, yo = 1.f / (float)H - 1.f;
    D9 X[3], Y[3], Wd[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 q = pos[voff + vi[k]];
        X[k] = dconst<9>(q.x);
        X[k].d[3 * k] = 1.f;
        Y[k] = dconst<9>(q.y);
        Y[k].d[3 * k + 1] = 1.f;
        Wd[k] = dconst<9>(q.w);
        Wd[k].d[3 * k + 2] = 1.f;
    }
    const D9 fx = dconst<9>((float)ix * xs + xo), fy = dconst<9>((float)iy * ys + yo);
    D9 px[3], py[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        px[k] = X[k] - fx * Wd[k];
        py[k] = Y[k] - fy * Wd[k];
    }
    const D9 a0 = px[1] * py[2] - py[1] * px[2];
    const D9 a1 = px[2] * py[0] - py[2] * px[0];
    const D9 a2 = px[0] * py[1] - py[0] * px[1];
    const D9 at = (a0 + a1) + a2;
    const D9 iw = dconst<9>(1.f) / at;
    const D9 b0 = a0 * iw, b1 = a1 * iw;
    const D9 dfx = dconst<9>(xs) * iw, dfy = dconst<9>(ys) * iw;
    const D9 da0x = Y[2] * Wd[1] - Y[1] * Wd[2], da0y = X[1] * Wd[2] - X[2] * Wd[1];
    const D9 da1x = Y[0] * Wd[2] - Y[2] * Wd[0], da1y = X[2] * Wd[0] - X[0] * Wd[2];
    const D9 da2x = Y[1] * Wd[0] - Y[0] * Wd[1], da2y = X[0] * Wd[1] - X[1] * Wd[0];
    const D9 datx = (da0x + da1x) + da2x, daty = (da0y + da1y) + da2y;
    D9 o[4];
    o[0] = dfx * (b0 * datx - da0x);
    o[1] = dfy * (b0 * daty - da0y);
    o[2] = dfx * (b1 * datx - da1x);
    o[3] = dfy * (b1 * daty - da1y);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    float* gp = grad_pos + 4 * voff;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float gx = 0.f, gy = 0.f, gw = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            gx += g[c] * o[c].d[3 * k];
            gy += g[c] * o[c].d[3 * k + 1];
            gw += g[c] * o[c].d[3 * k + 2];
        }
        atomicAdd(&gp[4 * vi[k] + 0], gx);
        atomicAdd(&gp[4 * vi[k] + 1], gy);
        atomicAdd(&gp[4 * vi[k] + 3], gw);
    }
}

}  // namespace ehr

using namespace ehr;

// ---- C ABI -------------------------------------------------------------------------------------------------------

extern "C" {

int ehr_version(void) { return 8; }

const char* ehr_last_error(void) { return g_last_error.c_str(); }

int ehr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ehr_device_arch(int dev) {
    static thread_local char arch[256];
    hipDeviceProp_t prop;
    arch[0] = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) snprintf(arch, sizeof(arch), "%s", prop.gcnArchName);
    return arch;
}

int ehr_ctx_create(int device, ehr_ctx** out) {
    if (!out) return fail(EHR_ERR_INVALID, "ehr_ctx_create: out is NULL");
    int n = ehr_device_count();
    if (device < 0 || device >= n) return fail(EHR_ERR_INVALID, "ehr_ctx_create: device %d out of range (%d visible)", device, n);
    int cur = 0;
    EHR_HIP(hipGetDevice(&cur));
    EHR_HIP(hipSetDevice(device));
    ehr_ctx* c = new ehr_ctx();
    c->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    }
    hipError_t e = hipHostMalloc((void**)&c->host_pinned, 8 * sizeof(int), hipHostMallocDefault);
    (void)hipSetDevice(cur);
    if (e != hipSuccess) {
        delete c;
        return fail(EHR_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return EHR_OK;
}

int ehr_ctx_destroy(ehr_ctx* c) {
    if (!c) return EHR_OK;
    (void)ehr_comm_destroy(c);
    (void)ehr_comm_p2p_close(c);
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(c->device);
    c->counts.release();
    c->ranges.release();
    c->rkeys.release();
    c->offsets.release();
    c->entries.release();
    c->sc_counts.release();
    c->sc_offsets.release();
    c->sc_entries.release();
    c->sc_posc.release();
    c->sc_clus.release();
    c->sc_misc.release();
    c->vb_clus.release();
    c->vb_idx.release();
    c->vb_heavy.release();
    c->vb_jobs.release();
    c->vb_boxes.release();
    c->vb_acc.release();
    c->vb_posc.release();
    c->vb_spill.release();
    c->vb_units.release();
    c->vb_refsum.release();
    c->vb_hstate.release();
    if (c->host_pinned) (void)hipHostFree(c->host_pinned);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; k++)
        if (c->ev_size[k]) (void)hipEventDestroy(c->ev_size[k]);
    if (c->gexec) (void)hipGraphExecDestroy(c->gexec);
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    (void)hipSetDevice(cur);
    delete c;
    return EHR_OK;
}

size_t ehr_ctx_scratch_bytes(ehr_ctx* c) {
    if (!c) return 0;
    size_t n = 0;
    for (const Scratch* s : {&c->counts, &c->ranges, &c->rkeys, &c->offsets, &c->entries, &c->vb_clus, &c->vb_heavy, &c->vb_idx, &c->vb_boxes, &c->vb_units,
                             &c->vb_acc, &c->vb_posc, &c->vb_jobs, &c->vb_spill, &c->vb_refsum, &c->vb_hstate, &c->sc_counts, &c->sc_offsets,
                             &c->sc_entries, &c->sc_posc, &c->sc_clus, &c->sc_misc})
        n += s->cap;
    return n;
}

size_t ehr_tile_flags_bytes(int B, int H, int W) {
    return (((size_t)std::max(B, 0) * flag_ntx(std::max(W, 1)) * flag_nty(std::max(H, 1))) + 3) & ~(size_t)3;
}

int ehr_rasterize_fwd(ehr_ctx* ctx, const float* pos, const int32_t* tri, const int32_t* ranges_host, int B, int V,
                      int T, int H, int W, float* rast, float* rast_db, unsigned char* tile_flags, void* stream_) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_rasterize_fwd: ctx is NULL");
    if (!pos || (!tri && T > 0) || !rast) return fail(EHR_ERR_INVALID, "ehr_rasterize_fwd: NULL tensor");
    if (B <= 0 || V < 0 || T < 0 || H <= 0 || W <= 0) return fail(EHR_ERR_INVALID, "ehr_rasterize_fwd: bad sizes");
    if (H > 32768 || W > 32768) return fail(EHR_ERR_INVALID, "ehr_rasterize_fwd: resolution above 32768 is unsupported");
    if (B > 65535) return fail(EHR_ERR_INVALID, "ehr_rasterize_fwd: %d images in one call (a grid extent holds 65535): split the batch", B);
    hipStream_t stream = (hipStream_t)stream_;
    bool capturing = false;
    {   // a call recorded into a stream capture bakes the scratch pointers into the graph: from then on they stay where they
        // are, for the life of the context (the graph is the caller's: the library never learns that it was destroyed)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cs);
        capturing = cs != hipStreamCaptureStatusNone;
        if (capturing)
            for (Scratch* sc : {&ctx->counts, &ctx->offsets, &ctx->entries, &ctx->rkeys, &ctx->ranges}) sc->pinned = true;
    }
    BinGeom g;
    g.W = W;
    g.H = H;
    g.ntx = (W + EHR_TILE_W - 1) / EHR_TILE_W;
    g.nty = (H + EHR_TILE_H - 1) / EHR_TILE_H;
    g.nt = g.ntx * g.nty;
    g.L = 1;
    const int nkeys = B * g.nt;
    int rc;
    if ((rc = ctx->counts.reserve(((size_t)2 * nkeys + EHR_META_INTS) * sizeof(int)))) return rc;
    if (ranges_host && (rc = ctx->ranges.reserve(2 * (size_t)B * sizeof(int)))) return rc;
    if ((rc = ctx->offsets.reserve((size_t)nkeys * sizeof(int)))) return rc;
    if (ctx->entries_cap == 0) {
        // (EHR_RASTER_MIN_ENTRIES: test hook for the undersized-storage path; default floor 1 M entries = 16 MB)
        static const size_t floor_entries = getenv("EHR_RASTER_MIN_ENTRIES") ? (size_t)std::max(1, atoi(getenv("EHR_RASTER_MIN_ENTRIES"))) : (size_t)1 << 20;
        size_t want = std::max(floor_entries, getenv("EHR_RASTER_MIN_ENTRIES") ? (size_t)0 : (size_t)B * (size_t)std::max(T, 1) * 2);
        if ((rc = ctx->entries.reserve(want * sizeof(int4)))) return rc;
        ctx->entries_cap = want;
    }
    int* counts = (int*)ctx->counts.ptr;
    int* cursors = counts + nkeys;
    int* meta = counts + 2 * nkeys;
    int* offsets = (int*)ctx->offsets.ptr;
    int2* ranges_dev = nullptr;
    int tmax = T;
    if (ranges_host) {
        ranges_dev = (int2*)ctx->ranges.ptr;
        const bool same = ctx->ranges_moves == ctx->ranges.moves && ctx->ranges_last.size() == 2 * (size_t)B &&
                          memcmp(ctx->ranges_last.data(), ranges_host, 2 * (size_t)B * sizeof(int32_t)) == 0;
        if (!same) {
            // The ranges live in ONE device buffer per context, and a captured graph holds no upload node:
            // new ranges under a capture would need a copy + a synchronisation on the capturing stream (which invalidates
            // the capture with an opaque HIP error), and new ranges on a context whose buffer a captured graph reads would
            // silently change what every later replay rasterizes.  Both are refused by name.
            if (capturing)
                return fail(EHR_ERR_INVALID, "ehr_rasterize_fwd: these ranges were not seen before the stream capture began; call once "
                            "with them outside the capture (warm-up), then capture");
            if (ctx->ranges.pinned && !ctx->ranges_last.empty())
                return fail(EHR_ERR_INVALID, "ehr_rasterize_fwd: this context's ranges are read by a captured graph; a call with other "
                            "ranges would change what its replays draw: use a RasterizeCudaContext of its own for them");
            // (from the context's own copy: the caller's array may be gone, or pageable, by the time the copy runs)
            ctx->ranges_last.assign(ranges_host, ranges_host + 2 * (size_t)B);
            ctx->ranges_moves = ~0ull;
            EHR_HIP(hipMemcpyAsync(ranges_dev, ctx->ranges_last.data(), (size_t)B * 2 * sizeof(int), hipMemcpyHostToDevice, stream));
            EHR_HIP(hipStreamSynchronize(stream));  // (once per new set of ranges: the host copy above must outlive the transfer)
            ctx->ranges_moves = ctx->ranges.moves;
        }
        tmax = 0;
        for (int b = 0; b < B; b++) tmax = std::max(tmax, ranges_host[2 * b + 1]);
        tmax = std::min(tmax, T);
    }
    ClipSource src;
    src.pos = (const float4*)pos;
    src.tri = tri;
    src.tri_link = nullptr;
    src.ranges = ranges_dev;
    src.V = V;
    src.T = T;
    src.L = 1;
    src.image_stride = ranges_host ? 0 : V;

    // Small launches (one link's mesh in one image: the reference's 64 calls per step) take the direct form: two kernels,
    // no queues, nothing for the host to size or wait for (its key image, 8 B per pixel of the call, stays with the context:
    // calls of more than 64 M pixels take the queued form).  EHR_RASTER_DIRECT_MAX (triangles x images; 0 = never) is a
    // test / tuning hook, read per call.
    {
        const char* e = getenv("EHR_RASTER_DIRECT_MAX");
        const size_t direct_max = e ? (size_t)std::max(0ll, atoll(e)) : ((size_t)1 << 17);
        // (range mode with small ranges -- a batch of (view, link) images over one concatenated mesh, the way nvdiffrast
        //  batches -- is many small launches in one: the direct form's cost per image is a latency chain that a batch hides)
        const bool small_ranges = ranges_host && tmax <= 16384 && (size_t)B * (size_t)tmax <= ((size_t)1 << 21) && !e;
        if (((size_t)B * (size_t)tmax <= direct_max || small_ranges) && (size_t)B * H * W <= ((size_t)1 << 26) && B <= 65535) {
            const size_t npix = (size_t)B * H * W;
            if ((rc = ctx->rkeys.reserve(npix * sizeof(u64)))) return rc;
            if (ctx->rkeys_clean != ctx->rkeys.moves) {
                if ((rc = fill_words(ctx->rkeys.ptr, ctx->rkeys.cap / sizeof(unsigned), 0xffffffffu, stream))) return rc;
            }
            ctx->rkeys_clean = ~0ull;  // (until the shade kernel is enqueued)
            u64* key = (u64*)ctx->rkeys.ptr;
            if (tmax > 0) {
                const int nbx = (tmax + 63) / 64;
                static const int rd_blocks = getenv("EHR_RD_BLOCKS") ? atoi(getenv("EHR_RD_BLOCKS")) : 16384;  // (tuning hook; bands are at least 8 rows)
                int Z = std::max(1, std::min(rd_blocks / std::max(1, nbx * B), (H + 7) / 8));
                const int band_rows = (H + Z - 1) / Z;
                Z = (H + band_rows - 1) / band_rows;
                raster_direct_kernel<<<dim3(nbx, Z, B), 256, 0, stream>>>(src, W, H, band_rows, key);
                EHR_LAUNCH_CHECK();
            }
            const dim3 sgrid((unsigned)(flag_ntx(W) * flag_nty(H)), (unsigned)B);
            if (rast_db)
                raster_shade_kernel<true><<<sgrid, 256, 0, stream>>>(src, B, W, H, key, (float4*)rast, (float4*)rast_db, tile_flags);
            else
                raster_shade_kernel<false><<<sgrid, 256, 0, stream>>>(src, B, W, H, key, (float4*)rast, nullptr, tile_flags);
            EHR_LAUNCH_CHECK();
            ctx->rkeys_clean = ctx->rkeys.moves;
            return EHR_OK;
        }
    }
    // (the queued form does not work out which tiles it drew into: every tile counts as occupied)
    if (tile_flags && (rc = fill_words(tile_flags, ehr_tile_flags_bytes(B, H, W) / sizeof(unsigned), 0x01010101u, stream))) return rc;
    // counts | cursors | meta are all zero between calls (raster_tile_kernel, the last kernel below, zeroes what a call
    // dirtied); only a fresh or moved buffer, or one a failed call left behind, is cleared here.
    if (ctx->counts_clean != ctx->counts.moves) {
        if ((rc = zero_words(counts, ctx->counts.cap / sizeof(int), stream))) return rc;
    }
    ctx->counts_clean = ~0ull;  // (until this call's last kernel is enqueued)
    dim3 bgrid((tmax + 255) / 256, B);
    if (tmax > 0) {
        bin_kernel<0, false><<<bgrid, 256, 0, stream>>>(src, g, counts, cursors, offsets, nullptr, 0, meta, nullptr);
        EHR_LAUNCH_CHECK();
    }
    bin_alloc_kernel<<<(nkeys + 255) / 256, 256, 0, stream>>>(counts, offsets, nullptr, nullptr, nullptr, nkeys, 1, meta);
    EHR_LAUNCH_CHECK();
    // Size read-back.  The queue storage must hold `total` entries, known only on the device.  Steady state (a solve
    // renders the same meshes again and again): the total travels to pinned host memory asynchronously and the NEXT call
    // looks at it -- if the last completed call of this shape needed at most half of the storage, this call does not
    // wait.  A frame that suddenly needs more than the storage holds is still rendered exactly: the tiles whose queues
    // did not fit find their triangles themselves (raster_tile_kernel's fallback; slower, never incomplete), and the
    // call after it sees the size and grows the storage.  Otherwise (first calls, new shape, tight storage) synchronise
    // once and grow, like nvdiffrast's own rasterizer.  Inside a stream capture (a torch.cuda.graphs capture of a whole
    // solver step) nothing on the host may wait or reallocate: the call is recorded with the storage as it is.
    if (!capturing) {
        const int slot = ctx->size_slot ^= 1;
        if (!ctx->ev_size[0]) {
            EHR_HIP(hipEventCreateWithFlags(&ctx->ev_size[0], hipEventDisableTiming));
            EHR_HIP(hipEventCreateWithFlags(&ctx->ev_size[1], hipEventDisableTiming));
        }
        const RasterShape shape = {B, V, T, H, W, ranges_host ? 1 : 0};
        bool wait = true;
        {
            const int prev = slot ^ 1;
            if (ctx->size_valid[prev] && ctx->size_shape[prev] == shape && hipEventQuery(ctx->ev_size[prev]) == hipSuccess &&
                2 * (size_t)ctx->host_pinned[6 + prev] + 1024 <= ctx->entries_cap)
                wait = false;
        }
        EHR_HIP(hipMemcpyAsync(ctx->host_pinned + 6 + slot, meta, sizeof(int), hipMemcpyDeviceToHost, stream));
        EHR_HIP(hipEventRecord(ctx->ev_size[slot], stream));
        ctx->size_valid[slot] = true;
        ctx->size_shape[slot] = shape;
        if (wait) {
            EHR_HIP(hipStreamSynchronize(stream));
            size_t total = (size_t)ctx->host_pinned[6 + slot];
            // (storage a captured graph points into stays where it is: raster_tile_kernel's fallback for undersized storage
            //  renders the frame all the same -- slower, never incomplete)
            if (2 * total + 1024 > ctx->entries_cap && !ctx->entries.pinned) {
                size_t want = 2 * total + total / 2 + 4096;
                if ((rc = ctx->entries.reserve(want * sizeof(int4)))) return rc;
                ctx->entries_cap = want;
            }
        }
    }
    int4* entries = (int4*)ctx->entries.ptr;
    if (tmax > 0) {
        bin_kernel<0, true><<<bgrid, 256, 0, stream>>>(src, g, counts, cursors, offsets, entries,
                                                       (int)std::min(ctx->entries_cap, (size_t)0x7fffffff), meta, nullptr);
        EHR_LAUNCH_CHECK();
    }
    dim3 tgrid(g.nt, B);
    int ecap = (int)std::min(ctx->entries_cap, (size_t)0x7fffffff);
    if (rast_db)
        raster_tile_kernel<true><<<tgrid, EHR_TILE_THREADS, 0, stream>>>(src, g, counts, offsets, entries, ecap,
                                                                        (float4*)rast, (float4*)rast_db, meta);
    else
        raster_tile_kernel<false><<<tgrid, EHR_TILE_THREADS, 0, stream>>>(src, g, counts, offsets, entries, ecap,
                                                                         (float4*)rast, nullptr, meta);
    EHR_LAUNCH_CHECK();
    ctx->counts_clean = ctx->counts.moves;
    return EHR_OK;
}

int ehr_rasterize_grad(const float* pos, const int32_t* tri, const float* rast, const float* dy, int range_mode, int B,
                       int V, int T, int H, int W, float* grad_pos, const unsigned char* tile_flags, void* stream_) {
    if (!pos || !tri || !rast || !dy || !grad_pos) return fail(EHR_ERR_INVALID, "ehr_rasterize_grad: NULL tensor");
    hipStream_t stream = (hipStream_t)stream_;
    size_t n = (size_t)B * H * W;
    if (n == 0) return EHR_OK;
    raster_grad_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>((const float4*)pos, tri, (const float4*)rast,
                                                                        (const float4*)dy, range_mode, B, V, T, H, W,
                                                                        grad_pos, tile_flags);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

int ehr_rasterize_grad_db(const float* pos, const int32_t* tri, const float* rast, const float* ddb, int range_mode, int B,
                          int V, int T, int H, int W, float* grad_pos, void* stream_) {
    if (!pos || !tri || !rast || !ddb || !grad_pos) return fail(EHR_ERR_INVALID, "ehr_rasterize_grad_db: NULL tensor");
    hipStream_t stream = (hipStream_t)stream_;
    size_t n = (size_t)B * H * W;
    if (n == 0) return EHR_OK;
    raster_grad_db_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>((const float4*)pos, tri, (const float4*)rast,
                                                                           (const float4*)ddb, range_mode, B, V, T, H, W,
                                                                           grad_pos);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

}  // extern "C"
