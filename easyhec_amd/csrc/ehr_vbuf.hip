// ehr_vbuf.hip -- the fused hot path: the kernels and the launch chain behind ehr_render_mask_loss / ehr_solver_step
// (host entry points: ehr_fused.hip) and the coverage-only chain of ehr_mask_variance (vbuf_score).  Restates
//   /root/reference/easyhec/modeling/models/rb_solve/rb_solver.py:60-72   (per-link render, sum, clamp, SSE)
//   /root/reference/easyhec/structures/nvdiffrast_renderer.py:33-47        (rasterize -> interpolate -> antialias -> flip)
//   /root/reference/easyhec/utils/nvdiffrast_utils.py:14-18                (transform_pos)
// without per-step triangle binning: no (tile, link) queues are built, counted, allocated or filled.  The meshes
// project to micro-triangles (median bounding box 8 px, a fifth of the non-empty boxes cover no pixel centre at all), for
// which building and draining per-tile queues cost more than the coverage tests themselves.  Instead ehr_fused_plan
// groups every link's triangles ONCE into clusters of 64 spatially close ones (recursive median split of the centroids
// in object space, valid for every pose), and a step is three launches per chunk of views (a fourth and a fifth only after a
// step has met a triangle for the general path):
//
//   vb_vertex_kernel    [pose forward] + clip-space vertices (posc) + one wave per cluster: transforms the cluster's
//                       triangles, snaps them, tests small boxes exactly (a triangle that covers no pixel centre is
//                       dropped here, once) and publishes per triangle its pixel box (tbox) and its raster record
//                       (integer edge functions with the tie rule folded in, depth-range class; trec); per cluster and
//                       per link the union of the boxes (cbox; lbox through integer atomics).
//   vb_job_kernel       one WAVE per job = (view, link, 32x8 tile the link's box touches), persistent waves over a job
//                       list that is never materialised.  A job culls cluster boxes, then triangle boxes, and rasterizes
//                       the survivors COVERAGE FIRST: a wave-wide balanced walker (units of 4 pixels for narrow boxes,
//                       exactly solved row spans for wide ones) ORs coverage into an LDS bitmap and defers the covered
//                       units; at the job's end only the units that hold a covered pixel with an uncovered neighbour
//                       are depth tested (ds_min_u64 on ordered(z/w) << 32 | triangle: order independent, hence the
//                       oracle's z-buffer bit for bit wherever anyone will look).  Last step's heaviest jobs go first
//                       on whole workgroups, its long ones as the waves' static first jobs.
//   vb_slow_kernel      jobs that met a triangle for the general path: near-plane clipping, 64-bit edges (normally none; the
//                       solver step launches it only once a step has needed it, see vb_put_aside).
//   (resolve stage)     covered/uncovered pixel pairs by bit arithmetic on the coverage bitmap, silhouette analysis of the
//                       hits, the link's 256 antialiased values + the blended pairs -> job slot: done by the job kernel's
//                       wave for the job it has just drawn, from LDS (vb_resolve_from_lds); vb_resolve_kernel, one wave per
//                       job, only for the jobs vb_slow_kernel redrew.
//   vb_composite_kernel one wave per tile that holds a job (every tile without a bound reference mask): sums the links'
//                       values in link order, clamps, frame loss, mask write, back-propagates the tile's blended pairs
//                       to 12 numbers per link in the view's fixed-point accumulators; its last-arriving workgroup runs
//                       the finish stage (accumulators -> loss / grad_mvp [-> pose backward -> Adam]).
//
// What runs once per rasterizer round or once per candidate cluster inside vb_job_kernel is kept free of LDS shuffles,
// integer divisions and searches: wave scans and reductions by DPP / v_permlane*_swap,
// a lane's first triangle by a scatter and a max-scan, packed hint ids, 24-bit multiplies.
//
// An earlier form kept the depth/id image in HBM and resolved visibility with one 64-bit global atomic-min per covered
// pixel; global atomics execute memory-side on this part (4.6 G/s with raster locality: 272 us for the 1.26 M fragments
// of the 8-view workload), see DESIGN.md section 6.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "ehr_fused_core.h"

namespace ehr {

constexpr int VB_RW = EHR_TILE_W + 2;  // tile + 1-pixel halo
constexpr int VB_RH = EHR_TILE_H + 2;
constexpr int VB_RN = VB_RW * VB_RH;   // 340
constexpr int VB_WORDS = (VB_RN + 63) / 64;  // 6 coverage words
constexpr int VB_LBOX_STRIDE = 16;     // ints per (view, link) box: min x, min y, max x, max y, padding to a 64-byte line
#define VB_HEAVY_T_DEFAULT 3000        // cost (4-pixel units walked + 256 per rasterizer round) from which a job counts as
                                      // heavy: next step a whole workgroup takes it.  Units, not triangles: a tile of 130 long
                                      // thin triangles (7000 units) keeps a wave busy for 60 us, one of 380 small ones for 25
constexpr int VB_HEAVY_CAP = 4096;    // heavy jobs remembered per step
constexpr int VB_MED_CAP = 2048;      // long single-wave jobs remembered per step (they are started first)
#define VB_MED_T_DEFAULT 1500          // cost from which a single-wave job counts as long
// Issue priority (s_setprio) of the waves on the jobs the kernel ends on: a long job started first runs beside three
// siblings per SIMD for most of its life (41 us instead of 30); with priority 48.5 instead of 50.0 us at 8 views.
#ifndef VB_PRIO_LONG
#define VB_PRIO_LONG 3
#endif
#ifndef VB_PRIO_HEAVY
#define VB_PRIO_HEAVY 2
#endif
constexpr int VB_JOB_ITEMS = 64;       // blended pairs kept in LDS per tile; the rest spills to a global pool
constexpr int VB_SPILL_BLOCK = 2048;   // items per spill allocation (one per overflowing tile)
constexpr u64 VB_EMPTY = ~0ull;
#define VB_FAST_EXTENT 8192            // snapped extent (1/16 px) up to which 32-bit edge functions are exact

// Counters that many waves hit with atomics each get a 128-byte line of their own behind the meta block (atomics on
// one line serialise memory-side at ~12 ns each): line xcd = job cursor of that XCD.
#define VB_LINES 36   // 0-7 job cursors of the XCDs, 8-15 composite arrival tickets of the XCDs, 16 the top ticket, 17 jobs put aside for vb_slow_kernel;
                      // merged kernel: 18-25 "this XCD's workgroups are through their jobs", 26 the same over the XCDs, 27-34 the XCDs' go flags (= generation)
__host__ __device__ __forceinline__ int* vb_line(int* meta, int k) {
    return (int*)((((uintptr_t)(meta + EHR_META_INTS)) + 127) & ~(uintptr_t)127) + 32 * k;
}

#define VB_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#ifdef VB_TIMELINE
#define VB_TL_BEGIN() const long long tl_t0_ = __builtin_readcyclecounter()
#define VB_TL_END(S_, i) do { if (lane_id() == 0) (S_).tl_c[i] += __builtin_readcyclecounter() - tl_t0_; } while (0)
#else
#define VB_TL_BEGIN() do { } while (0)
#define VB_TL_END(S_, i) do { } while (0)
#endif

struct VbItem {
    int packed;  // bits 0-9 q (region index of pixel0) | 10 d | 11-12 di | 13 tri1 | 14 (c1 - c0 > 0)
    int v1, v2;  // the two vertices of the crossing silhouette edge (global ids)
    float alpha;
};

// ---- compile-time bitmaps over the 34x10 region (bit i = region pixel i, row-major) ------------------------------
constexpr bool vb_interior(int i) {
    return (i % VB_RW) >= 1 && (i % VB_RW) <= EHR_TILE_W && (i / VB_RW) >= 1 && (i / VB_RW) <= EHR_TILE_H;
}
constexpr u64 vb_word_h(int k) {  // horizontal pair (i, i+1): same row, at least one pixel interior
    u64 w = 0;
    for (int j = 0; j < 64; j++) {
        const int i = 64 * k + j;
        if (i + 1 < VB_RN && (i % VB_RW) != VB_RW - 1 && (vb_interior(i) || vb_interior(i + 1))) w |= 1ull << j;
    }
    return w;
}
constexpr u64 vb_word_v(int k) {  // vertical pair (i, i+RW)
    u64 w = 0;
    for (int j = 0; j < 64; j++) {
        const int i = 64 * k + j;
        if (i + VB_RW < VB_RN && (vb_interior(i) || vb_interior(i + VB_RW))) w |= 1ull << j;
    }
    return w;
}

template <int S>
__device__ __forceinline__ void vb_shr(const u64 in[VB_WORDS], u64 out[VB_WORDS]) {
#pragma unroll
    for (int k = 0; k < VB_WORDS; k++) out[k] = (in[k] >> S) | (k + 1 < VB_WORDS ? in[k + 1] << (64 - S) : 0ull);
}

__device__ __forceinline__ int vb_readlane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float vb_readlane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float4 vb_readlane(const float4& v, int lane) {
    return make_float4(vb_readlane(v.x, lane), vb_readlane(v.y, lane), vb_readlane(v.z, lane), vb_readlane(v.w, lane));
}
__device__ __forceinline__ int vb_mbcnt(u64 m) {  // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// ---- accesses that are coherent between the eight XCDs INSIDE one launch ------------------------------------------
// The XCDs' L2s are not coherent with each other; between two launches the kernel boundary writes back and invalidates.
// The merged job + composite kernel (round 6) hands job slots from the wave that resolved a job to whichever wave
// composites its tile, possibly through another XCD's L2, in the middle of a launch: those words are written and read at
// AGENT scope (relaxed atomics: the stores write through, the loads miss the non-coherent levels), ordered by the arrival
// counters of the grid-wide barrier in between.  COH = false: plain accesses (the separate-launch forms).
template <bool COH>
__device__ __forceinline__ void vb_st_u64(void* p, u64 v) {
    if (COH)
        __hip_atomic_store((u64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *(u64*)p = v;
}
template <bool COH>
__device__ __forceinline__ u64 vb_ld_u64(const void* p) {
    return COH ? __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *(const u64*)p;
}
template <bool COH>
__device__ __forceinline__ void vb_st_i32(int* p, int v) {
    if (COH)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
template <bool COH>
__device__ __forceinline__ int vb_ld_i32(const int* p) {
    return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool COH>
__device__ __forceinline__ void vb_st_f4(float* p, const float4& v) {  // (16-byte aligned)
    if (COH) {
        vb_st_u64<true>(p, (u64)__float_as_uint(v.x) | ((u64)__float_as_uint(v.y) << 32));
        vb_st_u64<true>(p + 2, (u64)__float_as_uint(v.z) | ((u64)__float_as_uint(v.w) << 32));
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}
template <bool COH>
__device__ __forceinline__ float4 vb_ld_f4(const float* p) {
    if (COH) {
        const u64 a = vb_ld_u64<true>(p), b = vb_ld_u64<true>(p + 2);
        return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                           __uint_as_float((unsigned)(b >> 32)));
    }
    return *reinterpret_cast<const float4*>(p);
}
template <bool COH>
__device__ __forceinline__ void vb_st_item(VbItem* p, const VbItem& it) {
    if (COH) {
        vb_st_u64<true>(p, (u64)(unsigned)it.packed | ((u64)(unsigned)it.v1 << 32));
        vb_st_u64<true>((char*)p + 8, (u64)(unsigned)it.v2 | ((u64)__float_as_uint(it.alpha) << 32));
    } else {
        *p = it;
    }
}
template <bool COH>
__device__ __forceinline__ VbItem vb_ld_item(const VbItem* p) {
    if (COH) {
        const u64 a = vb_ld_u64<true>(p), b = vb_ld_u64<true>((const char*)p + 8);
        VbItem it;
        it.packed = (int)(unsigned)a;
        it.v1 = (int)(unsigned)(a >> 32);
        it.v2 = (int)(unsigned)b;
        it.alpha = __uint_as_float((unsigned)(b >> 32));
        return it;
    }
    return *p;
}

// ---- stage 1: vertices, screen boxes of triangles / clusters / links ---------------------------------------------

// Jobs that turned out heavy in the PREVIOUS step (poses move little between optimisation steps): remembered by their
// dense (view, link, tile) id, stamped into a table at the start of the step so that the single-wave enumeration skips
// them, and processed first, each by a whole workgroup.  Purely a scheduling hint: results do not depend on it.
struct VbHeavy {
    int* gen;     // [0] generation (one per call), [1..2] entries in list 0 / 1, [4..5] entries in mlist 0 / 1
    int* list;    // [2][VB_HEAVY_CAP] dense ids; list (gen & 1) is being written, the other one is being consumed
    int* mlist;   // [2][VB_MED_CAP] likewise: jobs below the heavy threshold that were long all the same
    int* stamp;   // [B * L * nt]  == generation: handled by a heavy workgroup this step; == -generation: a long job, dealt
                  // out as some wave's first job
    int mcap;     // long jobs consumed per step: min(VB_MED_CAP, 2 x workgroups of the job kernel) -- every one of them
                  // must find a wave with a static first job, and at least half the workgroups have those
    int heavy_base;  // the cost from which a job counts as heavy (gen[6] holds the value in force, never below this one)
    int heavy_max;   // heavy jobs a step can use (half the job kernel's workgroups): when a step records more, the value in
                     // force rises by an eighth, so that the heaviest jobs keep their workgroups instead of the whole phase
                     // being switched off (the Franka meshes at 1080p: 3000 jobs above 2500 in 8100; 112 instead of 124 us
                     // with the ~500 heaviest shared); it sinks back when fewer than half as many are recorded
};

// A remembered job in the hint lists: tile column | tile row << 10 | (view, link) << 22 -- no division on the way from a
// list entry to the first cluster box (the long jobs the kernel ends on start from these).
__device__ __forceinline__ int vb_hint_pack(int u, int tx, int ty) { return tx | (ty << 10) | (u << 22); }
__device__ __forceinline__ int vb_hint_dense(int id, int nt, int ntx) {
    return (int)((unsigned)id >> 22) * nt + ((id >> 10) & 4095) * ntx + (id & 1023);
}
static_assert(VB_MAX_UNITS <= 1024, "vb_hint_pack");
struct VbClusters {          // static acceleration index built by ehr_fused_plan (host): triangles grouped into
    const int32_t* ctri;     // [NC * 64] clusters of <= 64 spatially close triangles of one link (-1 = padding)
    const int32_t* clink;    // [NC] link of every cluster
    const int32_t* coff;     // [L + 1] first cluster of every link
    const float* laabb;      // [L][6] object-space bounding box of every link (min xyz, max xyz)
    const float4* cvert;     // [3][NC * 64] object-space corners of every cluster slot, packed (x0 y0 z0 x1 | y1 z1 x2 y2 |
                             // z2 valid - -): one coalesced round trip instead of the chain ctri -> tris -> verts
    int NC;
};

__device__ __forceinline__ uint2 vb_pack_box(int x0, int y0, int x1, int y1) {
    return make_uint2((unsigned)x0 | ((unsigned)y0 << 16), (unsigned)x1 | ((unsigned)y1 << 16));
}
#define VB_BOX_EMPTY make_uint2(0xffffffffu, 0u)  // x0 = y0 = 65535 > any pixel, x1 = y1 = 0: overlaps nothing

// grid = (nvb + ceil(NC / 4), B): the first nvb workgroups of a view transform its vertices (posc), the others take
// four clusters each -- one wave per cluster, one lane per triangle -- and publish the pixel bounding box of every
// triangle (tbox, empty for triangles that cover no pixel centre column/row), of every cluster (cbox) and, through a
// handful of integer atomics, of every link (lbox; reset by the finish kernel).  The cluster path transforms its own
// vertices with the same fma chain, so it does not wait for posc.  HEAD: the solver-step form (pose forward from dof,
// writes mvp / tc_jac / history row); otherwise mvp is an input.
// 32-bit edge functions of a triangle whose snapped extent is <= VB_FAST_EXTENT, relative to the centre of pixel
// (bx0, by0), which lies inside the triangle's bounding box: |coordinates| <= 2^13, products < 2^27.
struct VbEdges {
    int e0, e1, e2, sx0, sx1, sx2, sy0, sy1, sy2;
};
__device__ __forceinline__ VbEdges vb_edges(int X0, int Y0, int X1, int Y1, int X2, int Y2, int bx0, int by0, int W,
                                            int H) {
    VbEdges r;
    const int ox = 16 * bx0 + 8 - 8 * W, oy = 16 * by0 + 8 - 8 * H;
    X0 -= ox; X1 -= ox; X2 -= ox;
    Y0 -= oy; Y1 -= oy; Y2 -= oy;
    // (every factor is below 2^14 in magnitude: v_mul_i32_i24 is exact here and runs at full rate, v_mul_lo_u32 at a quarter)
    {
        const int dX = X1 - X0, dY = Y1 - Y0;
        r.e0 = __mul24(dX, 0 - Y0) - __mul24(dY, 0 - X0) - (((dY < 0) || (dY == 0 && dX < 0)) ? 0 : 1);
        r.sx0 = -16 * dY;
        r.sy0 = 16 * dX;
    }
    {
        const int dX = X2 - X1, dY = Y2 - Y1;
        r.e1 = __mul24(dX, 0 - Y1) - __mul24(dY, 0 - X1) - (((dY < 0) || (dY == 0 && dX < 0)) ? 0 : 1);
        r.sx1 = -16 * dY;
        r.sy1 = 16 * dX;
    }
    {
        const int dX = X0 - X2, dY = Y0 - Y2;
        r.e2 = __mul24(dX, 0 - Y2) - __mul24(dY, 0 - X2) - (((dY < 0) || (dY == 0 && dX < 0)) ? 0 : 1);
        r.sx2 = -16 * dY;
        r.sy2 = 16 * dX;
    }
    return r;
}

// Per-triangle raster record written by the cluster pass, read by the job waves (so a triangle is set up once per
// step, not once per tile it touches):
//   trec[2 i]     = { e0, e1, e2, dX0 | dY0 << 16 }   edge functions (tie rule folded in) at the centre of the box's
//   trec[2 i + 1] = { dX1 | dY1 << 16, dX2 | dY2 << 16, triangle id, kind }   first pixel; kind 0 = 32-bit fast path and
//                   every coverable pixel passes the depth-range test, 2 = fast path but the depth range must be tested
//                   per pixel, 1 = needs clipping / 64-bit (handled from the vertices)
// Depth is evaluated from the clip-space vertices (posc) of the few units that need it.
struct VbRecs {
    uint2* tbox;   // [B][NC][64] pixel box, VB_BOX_EMPTY if the triangle cannot cover a pixel centre
    uint2* cbox;   // [B][NC] union over the cluster
    int4* trec;    // [B][NC][64][2]: the two halves of a record side by side
    size_t n;      // B * NC * 64 (component stride)
};

// True if every pixel centre the triangle's SNAPPED outline can cover evaluates to a depth z/w inside [-1, 1] in
// vb_depth_test's arithmetic, so that coverage = the integer edge test alone.  All w > 0 (no near-plane crossing).
// z/w is affine over the screen; a covered pixel centre lies within 1/32 pixel (per axis) of the unsnapped triangle, and
// the float evaluation perturbs the barycentric weights by at most ~5e-5 pixel / thickness.  Hence: depth of the
// vertices, widened by the depth gradient over 0.6 sixteenth-pixels and by a quarter of the depth range, must stay
// 1e-5 inside the planes; thickness (2 area / longest edge) at least 0.05 sixteenth-pixels; w within a factor 4.
// Everything else -- NaNs included -- is "unsafe" and gets the exact per-pixel test.  For a robot in front of the
// camera (z/w = 0.998 at 1 m with near 1 mm, far 10 m; depth range of a triangle ~1e-5) only edge-on slivers fail.
// pos: the stricter class of the scoring op (mask = z/w of the nearest fragment > 0): every coverable pixel centre has
// a depth in (0, 1], so that coverage alone decides the mask.
__device__ __forceinline__ bool vb_depth_safe(const float4& p0, const float4& p1, const float4& p2, int W, int H, bool pos = false) {
    // (a classification, not a result: where it says "safe" the per-pixel depth-range test is skipped, and both ways draw the
    //  same pixels as long as "safe" is never said wrongly -- so the ten quotients are products with hardware reciprocals
    //  (1 ulp: 1e-7 of a depth near 1, absorbed by the margins below, which are a tenth wider than the bound they come from)
    //  instead of ten IEEE divisions of ~10 instructions each in the busiest loop of the vertex kernel)
    const float r0 = __builtin_amdgcn_rcpf(p0.w), r1 = __builtin_amdgcn_rcpf(p1.w), r2 = __builtin_amdgcn_rcpf(p2.w);
    const float z0 = p0.z * r0, z1 = p1.z * r1, z2 = p2.z * r2;
    const float sx = (float)(8 * W), sy = (float)(8 * H);
    const float x0 = p0.x * r0 * sx, y0 = p0.y * r0 * sy;
    const float d1x = p1.x * r1 * sx - x0, d1y = p1.y * r1 * sy - y0;
    const float d2x = p2.x * r2 * sx - x0, d2y = p2.y * r2 * sy - y0;
    const float q1 = fabsf(z1 - z0), q2 = fabsf(z2 - z0);
    const float A = fabsf(d1x * d2y - d2x * d1y);
    const float l1 = fabsf(d1x) + fabsf(d1y), l2 = fabsf(d2x) + fabsf(d2y);
    const float delta = 0.66f * (q1 * l2 + q2 * l1) * __builtin_amdgcn_rcpf(A) + 0.275f * (q1 + q2);
    const float zmax = fmaxf(z0, fmaxf(z1, z2)), zmin = fminf(z0, fminf(z1, z2));
    const float wmax = fmaxf(p0.w, fmaxf(p1.w, p2.w)), wmin = fminf(p0.w, fminf(p1.w, p2.w));
    return (A >= 0.055f * (l1 + l2) + 1.1e-3f) && (wmax <= 4.f * wmin) && (zmax + delta <= 1.f - 1.1e-5f) &&
           (zmin - delta >= (pos ? 1.1e-5f : -1.f + 1.1e-5f));
}

// two 16-bit unsigned minima / maxima in one instruction (v_pk_min_u16 / v_pk_max_u16)
typedef unsigned short vb_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned vb_pk_min_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(vb_u16x2, a), __builtin_bit_cast(vb_u16x2, b)));
}
__device__ __forceinline__ unsigned vb_pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(vb_u16x2, a), __builtin_bit_cast(vb_u16x2, b)));
}

#ifndef VB_SMALL_BOX
#define VB_SMALL_BOX 3  // boxes of up to this many pixel centres a side are tested exactly by the vertex kernel (4: the fourth
                        // row and column cost the VALU-bound kernel 7 % at Franka 16 x 1080p and drop nothing the job kernel
                        // notices; 2: the job kernel pays 2.3 us at 8 views for the 3 x 3 boxes that cover nothing)
#endif
#ifndef VB_VERTEX_WAVES
#define VB_VERTEX_WAVES 5  // (five workgroups per CU stay resident: the launch deals the work accordingly)
#endif
template <bool HEAD>
__global__ void __launch_bounds__(256, VB_VERTEX_WAVES)
vb_vertex_kernel(const float* __restrict__ verts, const int32_t* __restrict__ vert_link,
                 const int32_t* __restrict__ tris, VbClusters cl, StepHead head, float* __restrict__ mvp, int V, int nvb,
                 BinGeom g, float4* __restrict__ posc, VbRecs rc, int* __restrict__ lbox, int* __restrict__ zacc,
                 int nzacc, int* __restrict__ meta, int B, int gx, int xcd_views, VbHeavy hv, int chunk_role) {
    __shared__ float Tc[16];
    __shared__ float M[32][16];
    // 1-D grid of B * gx workgroups.  xcd_views > 0 (B a multiple of 8): workgroup w runs on XCD w % 8 (observed, used
    // for L2 locality only) and that XCD takes views [xcd * B / 8, (xcd + 1) * B / 8) -- the same views whose jobs stage 2
    // gives to that XCD, so the records it reads were written through the same L2.
    int b, bx;
    if (xcd_views > 0) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        b = xcd * xcd_views + k / gx;
        bx = k - (k / gx) * gx;
    } else {
        b = blockIdx.x / gx;
        bx = blockIdx.x - b * gx;
    }
    const int tid = threadIdx.x, L = g.L, H = g.H, W = g.W;
    // chunk_role: the views of a step go through the chain in chunks (one, unless views x links exceeds what a job
    // kernel handles): 1 = first chunk (the per-step housekeeping happens here), 2 = a later one (only the per-chunk
    // counters are re-armed)
    const bool pos_only = (chunk_role & 4) != 0;  // (the scoring op's chain)
    chunk_role &= 3;
    const bool first = bx == 0 && b == 0 && chunk_role == 1;
    const bool rearm = bx == 0 && b == 0 && chunk_role == 2;
    // A view's work items: [0, nvb) blocks of 256 vertices (-> posc; nvb = 0 where the plan computes clip-space vertices on
    // demand, see VbLazy), then groups of four clusters (one wave per cluster, one lane per triangle).  The workgroup takes items bx, bx + gx, ...: the pose head above every item (exponential,
    // matrices: ~3 us of dependent arithmetic) is paid once per workgroup, not once per 256 vertices -- with one item per
    // workgroup the Franka scene (375 k vertices x 16 views) needed 32 k workgroups and 125 us.
    const int nitems = nvb + (cl.NC + 3) / 4;
    const int lane = tid & 63;
    int item = bx;
    int l = -1, t = -1;
    float vx[3] = {0.f, 0.f, 0.f}, vy[3] = {0.f, 0.f, 0.f}, vz[3] = {0.f, 0.f, 0.f};
    bool have = false;
    // geometry loads first: they do not depend on the pose, so their latency hides under the pose arithmetic below
    auto load_geometry = [&](int it) {
        l = -1;
        t = -1;
        have = false;
        if (it < nvb) {
            const int v = it * 256 + tid;
            if (v < V) {
                l = vert_link[v];
                vx[0] = verts[3 * v];
                vy[0] = verts[3 * v + 1];
                vz[0] = verts[3 * v + 2];
                have = true;
            }
        } else {
            const int c = (it - nvb) * 4 + (tid >> 6);
            if (c < cl.NC) {
                const size_t cs = (size_t)c * 64 + lane, cn = (size_t)cl.NC * 64;
                t = cl.ctri[cs];
                l = cl.clink[c];
                const float4 q0 = cl.cvert[cs], q1 = cl.cvert[cn + cs], q2 = cl.cvert[2 * cn + cs];
                vx[0] = q0.x; vy[0] = q0.y; vz[0] = q0.z;
                vx[1] = q0.w; vy[1] = q1.x; vz[1] = q1.y;
                vx[2] = q1.z; vy[2] = q1.w; vz[2] = q2.x;
                have = t >= 0 && q2.y != 0.f;  // padding slots and triangles with a vertex index out of range draw nothing
            }
        }
    };
    if (item < nitems) load_geometry(item);
    __shared__ int lacc[32][4];  // pixel box of every link as far as this workgroup's clusters go
    if (tid < 128) lacc[tid >> 2][tid & 3] = (tid & 2) ? -1 : INT_MAX;
    // HEAD: the view's link poses and the intrinsics are requested NOW, together with dof, and parked in LDS: behind the
    // barrier below they were a second cold round trip on every workgroup's chain (dof -> exponential -> barrier -> link
    // poses -> matrices)
    __shared__ float LP[32][16];
    __shared__ float Kc[9];
    float dofv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (HEAD) {
        // (uniform: scalar loads into scalar registers, requested first; the exponential below waits for these alone)
#pragma unroll
        for (int k = 0; k < 6; k++) dofv[k] = head.dof[k];
        float lpv[2] = {0.f, 0.f};
        float kv = 0.f;
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (tid + 256 * j < L * 16) lpv[j] = head.link_poses[(size_t)b * L * 16 + tid + 256 * j];
        if (tid >= 64 && tid < 73) kv = head.K[tid - 64];
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (tid + 256 * j < L * 16) LP[(tid + 256 * j) >> 4][(tid + 256 * j) & 15] = lpv[j];
        if (tid >= 64 && tid < 73) Kc[tid - 64] = kv;
    }
    if (HEAD && tid < 6) {
        Dual<1> T6[16];
        se3_exp_dual<1>(dofv, 1e-4f, T6, tid);
        if (tid == 0)
            for (int i = 0; i < 16; i++) Tc[i] = T6[i].v;
        if (first) {
            for (int i = 0; i < 16; i++) {
                if (tid == 0) head.tc_jac[i] = T6[i].v;
                head.tc_jac[16 * (tid + 1) + i] = T6[i].d[0];
            }
            if (tid == 0 && head.history && head.hist_row) {  // rb_solver.py:50-51: the pose goes to the next free row
                int row = head.hist_row[0];
                if (head.hstate && head.adam_step) {
                    // no Adam step since the previous head: that step was reported, its row (the same pose) is this step's
                    // (and nobody has moved the cursor either -- a state loaded from outside sets both)
                    const int t1 = head.adam_step[0] + 1;
                    if (head.hstate[0] == t1 && head.hstate[1] && head.hstate[2] == row && row > 0) row -= 1;
                    head.hstate[0] = t1;
                }
                const bool adv = row >= 0 && row < head.history_rows;
                if (adv) {
                    for (int k = 0; k < 6; k++) head.history[6 * row + k] = dofv[k];
                    head.hist_row[0] = row + 1;
                }
                if (head.hstate) {
                    head.hstate[1] = adv ? 1 : 0;
                    head.hstate[2] = adv ? row + 1 : row;
                }
            }
        }
    }
    if (first) {
        // new generation of the heavy-job hint: stamp last step's heavy jobs, start this step's list empty
        __shared__ int s_gen;
        if (tid == 0) {
            s_gen = hv.gen[0] + 1;
            hv.gen[0] = s_gen;
            hv.gen[1 + (s_gen & 1)] = 0;
            hv.gen[4 + (s_gen & 1)] = 0;
            {   // the heavy threshold in force follows the number of heavy jobs the last step recorded
                const int nprev = hv.gen[1 + ((s_gen - 1) & 1)];
                int thr = max(hv.gen[6], hv.heavy_base);
                if (nprev > hv.heavy_max)
                    thr = min(thr + (thr >> 3), 1 << 24);
                else if (2 * nprev < hv.heavy_max)
                    thr = max(hv.heavy_base, thr - (thr >> 4));
                hv.gen[6] = thr;
            }
        }
        __syncthreads();
        {
            const int gen = s_gen, cur = (gen - 1) & 1;
            const int n = min(hv.gen[1 + cur], VB_HEAVY_CAP);
            for (int i = tid; i < n; i += 256) hv.stamp[min(vb_hint_dense(hv.list[cur * VB_HEAVY_CAP + i], g.nt, g.ntx), B * L * g.nt - 1)] = gen;
            const int n2 = min(hv.gen[4 + cur], hv.mcap);
            for (int i = tid; i < n2; i += 256) hv.stamp[min(vb_hint_dense(hv.mlist[cur * VB_MED_CAP + i], g.nt, g.ntx), B * L * g.nt - 1)] = -gen;
        }
        if (tid < 8) meta[tid] = 0;                          // overflow flag, spill cursor
        if (tid < VB_LINES) *vb_line(meta, tid) = 0;         // job cursors of the 8 XCDs, tickets, slow-job count
    }
    if (rearm) {  // a later chunk of the same step: cursors and tickets again, the overflow flag stays
        if (tid < 8 && tid != EHR_META_OVERFLOW) meta[tid] = 0;
        if (tid < VB_LINES) *vb_line(meta, tid) = 0;
    }
    if (bx == 0) {  // fixed-point accumulators of this view (a few KB)
        const int nzv = nzacc / B;
        for (int i = tid; i < nzv; i += 256) zacc[(size_t)b * nzv + i] = 0;
    }
    __syncthreads();
    if (HEAD) {
        if (tid < L) {
            float P[16], C[16];
            projection(Kc, H, W, head.n, head.f, P);
            mvp_from_pose(Tc, P, LP[tid], C);
            for (int k = 0; k < 16; k++) M[tid][k] = C[k];
            if (bx == 0)
                for (int k = 0; k < 16; k++) mvp[((size_t)b * L + tid) * 16 + k] = C[k];
        }
    } else {
        for (int i = tid; i < L * 16; i += 256) M[i >> 4][i & 15] = mvp[(size_t)b * L * 16 + i];
    }
    __syncthreads();
    for (; item < nitems;) {
    const bool lv = (unsigned)l < (unsigned)L;
    if (item < nvb) {
        const int v = item * 256 + tid;
        if (v < V) {
            float4 o = make_float4(0.f, 0.f, 0.f, -1.f);  // invalid link -> behind the camera, never drawn
            if (lv) o = transform_vertex(M[l], vx[0], vy[0], vz[0]);
            posc[(size_t)b * V + v] = o;
        }
    } else {
    const int c = (item - nvb) * 4 + (tid >> 6);
    const bool cvalid = c < cl.NC;
    int x0 = 0xffff, y0 = 0xffff, x1 = 0, y1 = 0;  // empty (overlaps nothing)
    int4 r0 = make_int4(-1, -1, -1, 0), r1 = make_int4(0, 0, t, 0);
    if (have && lv) {
        const float4 p0 = transform_vertex(M[l], vx[0], vy[0], vz[0]);
        const float4 p1 = transform_vertex(M[l], vx[1], vy[1], vz[1]);
        const float4 p2 = transform_vertex(M[l], vx[2], vy[2], vz[2]);
        const bool simple = (p0.w > 0.f) && (p1.w > 0.f) && (p2.w > 0.f) && (p0.z + p0.w >= 0.f) &&
                            (p1.z + p1.w >= 0.f) && (p2.z + p2.w >= 0.f);
        if (!simple) {  // crosses the near plane: the pixel box of its clipped pieces (the whole-wave path draws it)
            const float4 pp[3] = {p0, p1, p2};
            const ClipPoly cp = clip_near_poly(pp);
            for (int sidx = 0; sidx + 2 < cp.n; sidx++) {
                const Coverage cv = (sidx == 0) ? setup_coverage(cp.q0, cp.q1, cp.q2, W, H) : setup_coverage(cp.q0, cp.q2, cp.q3, W, H);
                if (!cv.valid) continue;
                const int ax = max(cv.ix0, 0), ay = max(cv.iy0, 0), bx1 = min(cv.ix1, W - 1), by1 = min(cv.iy1, H - 1);
                if (ax > bx1 || ay > by1) continue;
                x0 = min(x0, ax);
                y0 = min(y0, ay);
                x1 = max(x1, bx1);
                y1 = max(y1, by1);
            }
            r1.w = 1;
        } else {
            const Coverage cv = setup_coverage(p0, p1, p2, W, H);
            if (cv.valid) {
                x0 = cv.ix0; y0 = cv.iy0; x1 = cv.ix1; y1 = cv.iy1;
                // (snapped coordinates lie within +-2^30: the extent fits 32 unsigned bits)
                const unsigned ex = (unsigned)max(cv.X[0], max(cv.X[1], cv.X[2])) - (unsigned)min(cv.X[0], min(cv.X[1], cv.X[2]));
                const unsigned ey = (unsigned)max(cv.Y[0], max(cv.Y[1], cv.Y[2])) - (unsigned)min(cv.Y[0], min(cv.Y[1], cv.Y[2]));
                if (ex > VB_FAST_EXTENT || ey > VB_FAST_EXTENT) {
                    r1.w = 1;
                } else {
                    const VbEdges ed = vb_edges(cv.X[0], cv.Y[0], cv.X[1], cv.Y[1], cv.X[2], cv.Y[2], x0, y0, W, H);
                    // A small box (<= VB_SMALL_BOX pixel centres a side: nearly half of all triangles) is tested exactly, once, here: a fifth of
                    // the triangles with a non-empty box cover no pixel centre at all (70 % of the 1 x 1 boxes), and would
                    // otherwise be fetched, staged and walked by every job whose region their box touches.  The same
                    // integer edge functions as the rasterizer's walk, so nothing that could be drawn is dropped.
                    if (x1 - x0 < VB_SMALL_BOX && y1 - y0 < VB_SMALL_BOX) {
                        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
                        bool any = false;
                        int q0 = ed.e0, q1 = ed.e1, q2 = ed.e2;
#pragma unroll
                        for (int j = 0; j < VB_SMALL_BOX; j++) {
                            int a0 = q0, a1 = q1, a2 = q2;
#pragma unroll
                            for (int i = 0; i < VB_SMALL_BOX; i++) {
                                any = any || (i < bw && j < bh && (a0 | a1 | a2) >= 0);
                                a0 += ed.sx0;
                                a1 += ed.sx1;
                                a2 += ed.sx2;
                            }
                            q0 += ed.sy0;
                            q1 += ed.sy1;
                            q2 += ed.sy2;
                        }
                        if (!any) {
                            x0 = 0xffff; y0 = 0xffff; x1 = 0; y1 = 0;
                        }
                    }
                    r0 = make_int4(ed.e0, ed.e1, ed.e2, (int)(((unsigned)(ed.sy0 / 16) & 0xffffu) | ((unsigned)(-ed.sx0 / 16) << 16)));
                    r1.x = (int)(((unsigned)(ed.sy1 / 16) & 0xffffu) | ((unsigned)(-ed.sx1 / 16) << 16));
                    r1.y = (int)(((unsigned)(ed.sy2 / 16) & 0xffffu) | ((unsigned)(-ed.sx2 / 16) << 16));
                    r1.w = vb_depth_safe(p0, p1, p2, W, H, pos_only) ? 0 : 2;
                }
            }
        }
    }
    const size_t slot = ((size_t)b * cl.NC + (cvalid ? c : 0)) * 64 + lane;
    if (cvalid) rc.tbox[slot] = vb_pack_box(x0, y0, x1, y1);
    if (cvalid && x0 <= x1) {
        rc.trec[2 * slot] = r0;  // (the two halves side by side: one 32-byte piece of a cache line per triangle)
        rc.trec[2 * slot + 1] = r1;
    }
    // the cluster's box: a DPP reduction (four shifts inside the rows of 16 lanes, then the rows' results upwards: lane 63
    // ends up with everything; integers, so the order is free), no LDS.  The two words of the packed box ARE the operands: two
    // 16-bit minima / maxima per instruction (an empty box is (65535, 65535)-(0, 0): neutral both ways).
    const uint2 pbox = vb_pack_box(x0, y0, x1, y1);
    unsigned bmin = pbox.x, bmax = pbox.y;
#define VB_BOX_STEP(ctrl, rows)                                                                       \
    bmin = vb_pk_min_u16(bmin, (unsigned)__builtin_amdgcn_update_dpp((int)bmin, (int)bmin, ctrl, rows, 0xf, false)); \
    bmax = vb_pk_max_u16(bmax, (unsigned)__builtin_amdgcn_update_dpp((int)bmax, (int)bmax, ctrl, rows, 0xf, false));
    VB_BOX_STEP(0x111, 0xf) VB_BOX_STEP(0x112, 0xf) VB_BOX_STEP(0x114, 0xf) VB_BOX_STEP(0x118, 0xf)
    VB_BOX_STEP(0x142, 0xa) VB_BOX_STEP(0x143, 0xc)
#undef VB_BOX_STEP
    const int a = (int)(bmin & 0xffffu), bq = (int)(bmin >> 16), cc = (int)(bmax & 0xffffu), d = (int)(bmax >> 16);
    if (lane == 63) {
        const bool cne = cvalid && a <= cc;
        if (cvalid) rc.cbox[(size_t)b * cl.NC + c] = cne ? vb_pack_box(a, bq, cc, d) : VB_BOX_EMPTY;
        if (cne && lv) {  // link box: merged over everything this workgroup sees (LDS), published once at the end
            atomicMin(&lacc[l][0], a);
            atomicMin(&lacc[l][1], bq);
            atomicMax(&lacc[l][2], cc);
            atomicMax(&lacc[l][3], d);
        }
    }
    }
    item += gx;
    if (item < nitems) load_geometry(item);
    }
    // link boxes: integer atomics, one 64-byte line per (view, link): a workgroup's clusters belong to one or two links,
    // ~6 k atomics per step spread over B * L lines
    __syncthreads();
    if (tid < 4 * L && lacc[tid >> 2][0] != INT_MAX) {
        int* const lb = lbox + VB_LBOX_STRIDE * ((size_t)b * L + (tid >> 2));
        if (tid & 2)
            atomicMax(lb + (tid & 3), lacc[tid >> 2][tid & 3]);
        else
            atomicMin(lb + (tid & 3), lacc[tid >> 2][tid & 3]);
    }
}

// ---- stage 2: one wave per job: cull, coverage into LDS, depth only where the silhouette analysis will look ---------

// i / VB_RW for 0 <= i < 4096 by one full-rate 24-bit multiply (1928 = ceil(2^16 / 34), 1928 * 34 - 2^16 = 16)
__device__ __forceinline__ int vb_div_rw(int i) { return (int)(__umul24((unsigned)i, 1928u) >> 16); }
static_assert(VB_RW == 34, "vb_div_rw");
// ---- wave scans without LDS (DPP): four shifts inside the rows of 16 lanes, then the rows' totals into the rows above
#ifndef VB_FAST_SEARCH
#define VB_FAST_SEARCH 1  // 0: the scans as six ds_bpermute steps and the walkers' start as a binary search over the prefix table
#endif
#define VB_DPP_(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rows), 0xf, false)
__device__ __forceinline__ int vb_scan_add(int v) {  // inclusive sum
#if VB_FAST_SEARCH
    v += VB_DPP_(v, 0x111, 0xf);  // row_shr:1
    v += VB_DPP_(v, 0x112, 0xf);  // row_shr:2
    v += VB_DPP_(v, 0x114, 0xf);  // row_shr:4
    v += VB_DPP_(v, 0x118, 0xf);  // row_shr:8
    v += VB_DPP_(v, 0x142, 0xa);  // row_bcast:15 -> rows 1 and 3
    v += VB_DPP_(v, 0x143, 0xc);  // row_bcast:31 -> rows 2 and 3
#else
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
#endif
    return v;
}
__device__ __forceinline__ int vb_scan_max(int v) {  // inclusive maximum of values >= 0
    v = max(v, VB_DPP_(v, 0x111, 0xf));
    v = max(v, VB_DPP_(v, 0x112, 0xf));
    v = max(v, VB_DPP_(v, 0x114, 0xf));
    v = max(v, VB_DPP_(v, 0x118, 0xf));
    v = max(v, VB_DPP_(v, 0x142, 0xa));
    v = max(v, VB_DPP_(v, 0x143, 0xc));
    return v;
}
// x / d for 0 <= x < 2^16, 1 <= d <= 128 (inv = 1 / d to an ulp): (x + 0.5) / d is at least 0.5 / d away from the next
// integer, the float product at most 2^16 * 2^-22 -- exact
__device__ __forceinline__ int vb_div_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
// the next set bit of m above position j (there is one)
__device__ __forceinline__ int vb_next_set(u64 m, int j) { return j + __ffsll((unsigned long long)(m >> (j + 1))); }

struct VbRegion {
    int x0, y0, x1, y1;  // pixels of the region inside the image (inclusive); region origin = (rx0, ry0) below
};

// Clip-space vertices ON DEMAND (round 6).  Until round 5 the vertex kernel wrote every vertex's clip-space position per view
// (posc [B][V] float4) for the few thousand the depth tests, the silhouette analysis and the backward pass look up per step --
// at Franka 16 x 1080p (unshared vertices: V = 3 T) 168 MB of a 286 MB, bandwidth-bound launch.  The consumers now
// transform the object-space vertex themselves: the same fma chain (transform_vertex) on the same matrix (mvp[b][l], which
// the vertex kernel publishes) gives the same bits.  VbLazy is what travels (two pointers); VbVerts holds the matrix in
// scalar registers for the short stretch in which vertices are looked up.
struct VbLazy {
    const float* verts;  // [V][3] object-space vertices
    const float* M;      // mvp[b][l], 16 floats: the job's (view, link) -- wave-uniform
    const float4* pc;    // the view's clip-space vertices (posc + b V) where the plan keeps them (LAZY = false), else NULL
};
// LAZY = true: the matrix in scalar registers, a vertex = three loads + twelve fmas; false: a vertex = one 16-byte load.
// A template, not a run-time switch: the sixteen scalars of the lazy form cost the job kernel -- which has six registers to
// spare -- 19 more spilled VGPRs (8 views: 41.3 -> 44.8 us) whether they are used or not, so the plan picks the instantiation:
// lazy where vertices outnumber what is looked up by far (unshared vertices, V > 1.5 T: Franka 16 x 1080p, vertex stage
// 64.8 -> 46.8 us, step 175.5 -> 163.1 us), eager otherwise.
template <bool LAZY>
struct VbVertsT;
template <>
struct VbVertsT<true> {
    const float* verts;
    float M[16];
    __device__ __forceinline__ float4 operator[](int v) const {
        const float* a = verts + 3 * (size_t)v;
        return transform_vertex(M, a[0], a[1], a[2]);
    }
};
template <>
struct VbVertsT<false> {
    const float4* pc;
    __device__ __forceinline__ float4 operator[](int v) const { return pc[v]; }
};
template <bool LAZY>
__device__ __forceinline__ VbVertsT<LAZY> vb_verts(const VbLazy& z);
template <>
__device__ __forceinline__ VbVertsT<true> vb_verts<true>(const VbLazy& z) {
    VbVertsT<true> r;
    r.verts = z.verts;
#pragma unroll
    for (int k = 0; k < 16; k++) r.M[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(z.M[k])));
    return r;
}
template <>
__device__ __forceinline__ VbVertsT<false> vb_verts<false>(const VbLazy& z) {
    VbVertsT<false> r;
    r.pc = z.pc;
    return r;
}

// What stage 2 has to leave behind for a (view, link, region) is (a) which region pixels the link covers and (b) the
// nearest triangle at those covered pixels that have an uncovered 4-neighbour: only pairs of one covered and one
// uncovered pixel reach the silhouette analysis (with constant colour inside a link a blend between two covered pixels
// is alpha * (1 - 1) = 0 in value and in gradient), and the analysis looks at the covered pixel's triangle.  So the
// rasterizer works COVERAGE FIRST: the walker ORs 4-pixel coverage masks into a bitmap (one 64-bit word per region row)
// and appends the covered units to a deferred list; when the job's coverage is complete (or the list is full) the
// list is filtered against the pixels that can still matter and only the survivors -- typically a sixth of the
// fragments -- are depth tested (one IEEE division per pixel, a 64-bit LDS atomic min each).  Filtering against a PARTIAL
// coverage is conservative: a pixel that is interior to the partial coverage stays interior, so what a partial flush
// drops could never have been looked at.
//
// A covered pixel is only drawn if its depth z/w lies in [-1, 1] (depth_test_write in ehr_raster_core.h, the oracle's
// z-buffer loop): the vertex kernel marks a triangle SAFE (kind 0) when that holds for every pixel centre its snapped
// outline can cover -- vertex depths away from the planes by more than the plane's screen-space depth gradient over the
// snapping distance plus the evaluation's rounding (see vb_depth_safe) -- which is every triangle of a robot in front
// of the camera.  The others (kind 2: edge-on slivers, geometry at the far plane) take the same walker, but their
// units do not enter the bitmap: they are always depth tested, and a pixel that passes sets its coverage bit then.
#define VB_SPAN_GW 4                   // boxes from this many 4-pixel units per row are walked by rows (solved spans), not by units
#ifdef VB_TIMELINE
constexpr int VB_DL = 576;             // (profiling build: its per-wave counters need the room -- four workgroups per CU must still fit)
#else
constexpr int VB_DL = 640;             // deferred units per wave (LDS); a full list is flushed against the partial coverage
#endif
constexpr int VB_SQ = 256;             // ring of culling survivors per wave (LDS): a whole group of candidate clusters' worth
constexpr unsigned VB_ID_COVERED = 0xfffffffeu;  // published id of a covered pixel whose triangle nobody will ask for
constexpr u64 VB_ROW_MASK = (1ull << VB_RW) - 1ull;

struct alignas(16) VbRaster {  // staging area of one rasterizer round (64 candidate triangles)
    int e[64][3];          // edge functions at the first pixel of the job's box inside the region
    unsigned dxy[64][3];   // dX | dY << 16 of the three edges
    unsigned box[64];      // box inside the region: x0 | y0 << 8 | w << 16 | h << 24 (region-relative)
    unsigned ent[64];      // deferred-list entry of this triangle without pixel and mask: flag << 13 | link-relative slot << 14
    int pre[65];           // exclusive prefix of the jobs' work units (+ total)
};
struct alignas(16) VbWaveLds {   // per wave of the job kernel
    u64 key[VB_RN];              // depth/id of the region's pixels: ordered(z/w) << 32 | triangle, all-ones = never tested
    u64 cov[VB_RH];              // coverage bitmap, one word per region row (bit x = region column x)
    u64 need[VB_RH];             // flush: covered pixels with an uncovered 4-neighbour
    u64 intr[VB_RH];             // rounds: covered pixels whose four neighbours are covered too (as of the round's start)
    u64 intr_and[3][VB_RH];      // ... AND-ed over the rows r .. r + 2^k - 1 (k = 1, 2, 3; past the last row: all ones)
    VbRaster R;
    unsigned dl[VB_DL];          // deferred units: pixel (9 bits) | 4-bit coverage << 9 | R.ent
    unsigned sq[VB_SQ];          // survivors of the box culling waiting for a full round: record slots (ring)
    int bad;                     // scoring op: a flagged unit's pixel was drawn with a depth <= 0 (coverage cannot decide)
#ifdef VB_TIMELINE
    int tl_units, tl_rounds;     // profiling build: 4-pixel units walked / rounds run by this wave
    int tl_flushes, tl_tested, tl_deferred;
    int tl_cands, tl_groups;     // candidate clusters / groups of them (one round trip each)
    long long tl_c[8];           // cycles: staging, prefix + search, walk, flush, job total, claim + set-up, publish, -
#endif
};

// z/w at the centre of pixel (ix, iy) from the triangle's clip-space vertices, the oracle's arithmetic
// (depth_test_write); true if the pixel is drawn (depth inside [-1, 1]), and then the z-buffer is updated.
// (*nonpos, optional: set if a drawn pixel's depth is <= 0 -- the scoring op's mask is "depth of the nearest fragment > 0")
__device__ __forceinline__ bool vb_depth_test(const float4 p[3], int t, int ix, int iy, int W, int H, u64* slot,
                                              bool* nonpos = nullptr) {
    const float xs = 2.f / (float)W, xo = 1.f / (float)W - 1.f;
    const float ys = 2.f / (float)H, yo = 1.f / (float)H - 1.f;
    const float fx = (float)ix * xs + xo;
    const float fy = (float)iy * ys + yo;
    float a0, a1, a2;
    eval_pixel(p, fx, fy, a0, a1, a2);
    const float zw = eval_zw(p, a0, a1, a2);
    if (zw >= -1.f && zw <= 1.f) {
        if (nonpos) {
            if (!(zw > 0.f)) *nonpos = true;
        } else {
            atomicMin(slot, ((u64)ord_key(zw) << 32) | (unsigned)t);
        }
        return true;
    }
    return false;
}

// Static per-cluster-slot table (plan time): the three vertex ids and the triangle id of every cluster slot, so that a
// deferred unit finds its clip-space vertices (posc) with one 16-byte gather + three.
struct VbSlotIdx {
    const int4* cvidx;  // [NC * 64] {v0, v1, v2, triangle}; padding slots hold {0, 0, 0, -1}
};

// General path, whole wave per triangle, wave-uniform arguments: near-plane clipping, 64-bit edge functions, immediate
// depth test.  Rare (triangles crossing the near plane or spanning more than 512 pixels); only compiled into
// vb_job_slow, which redoes a job in which the lean code met such a triangle.
__device__ __forceinline__ void vb_raster_wide(float4 pa, float4 pb, float4 pc, int t, int W, int H, VbRegion rg, int rx0,
                                            int ry0, u64* key, u64* cov) {
    const int lane = lane_id();
    const float4 p[3] = {pa, pb, pc};
    const ClipPoly c = clip_near_poly(p);
    for (int s = 0; s + 2 < c.n; s++) {
        Coverage cv = (s == 0) ? setup_coverage(c.q0, c.q1, c.q2, W, H) : setup_coverage(c.q0, c.q2, c.q3, W, H);
        if (!cv.valid) continue;
        const int bx0 = max(cv.ix0, rg.x0), by0 = max(cv.iy0, rg.y0);
        const int bx1 = min(cv.ix1, rg.x1), by1 = min(cv.iy1, rg.y1);
        if (bx0 > bx1 || by0 > by1) continue;
        const int bw = bx1 - bx0 + 1, npx = bw * (by1 - by0 + 1);
        const EdgeEval ee = setup_edges(cv, bx0, by0, W, H);
        for (int i0 = 0; i0 < npx; i0 += 64) {
            const int i = i0 + lane;
            if (i < npx) {
                const int dy = i / bw, dx = i - dy * bw;
                const i64 e0 = ee.e[0] + dx * ee.sx[0] + dy * ee.sy[0];
                const i64 e1 = ee.e[1] + dx * ee.sx[1] + dy * ee.sy[1];
                const i64 e2 = ee.e[2] + dx * ee.sx[2] + dy * ee.sy[2];
                if ((e0 | e1 | e2) >= 0) {
                    const int ix = bx0 + dx, iy = by0 + dy;
                    if (vb_depth_test(p, t, ix, iy, W, H, &key[(iy - ry0) * VB_RW + (ix - rx0)]))
                        atomicOr((unsigned long long*)&cov[iy - ry0], 1ull << (ix - rx0));
                }
            }
        }
    }
}

// Flush of the deferred list (n entries): the pixels that can still matter under the CURRENT coverage (`cov`, complete
// or partial; shared by the four waves of a workgroup in the heavy-job phase) are the covered ones with an uncovered
// 4-neighbour inside the region; entries that touch none of them are dropped, the others are depth tested at exactly
// those pixels.  Units of unsafe triangles (flag) are always tested, at all their pixels, and set their coverage bits.
// COVER (scoring op): the list holds flagged units only; a pixel that passes the depth-range test is covered, and one whose
// depth is not positive raises *bad (the coverage-only chain cannot decide that pixel: the caller falls back).
template <bool COVER = false, bool LAZY = false>
__device__ __forceinline__ void vb_flush(VbWaveLds& S, u64* key, u64* cov, int n, const VbLazy& pvz,
                                      const int4* __restrict__ cvidx_link, int W, int H, int rx0, int ry0) {
    const int lane = lane_id();
    VB_TL_BEGIN();
    VB_WAVE_SYNC();
    if (lane < VB_RH) {
        const u64 c = cov[lane];
        const u64 up = (lane + 1 < VB_RH) ? cov[lane + 1] : VB_ROW_MASK;
        const u64 dn = (lane > 0) ? cov[lane - 1] : VB_ROW_MASK;
        // neighbour x + 1 is bit x of c >> 1, neighbour x - 1 bit x of c << 1; past the region's edge counts as covered
        const u64 inner = ((c >> 1) | (1ull << (VB_RW - 1))) & ((c << 1) | 1ull) & up & dn;
        S.need[lane] = c & ~inner & VB_ROW_MASK;
    }
    VB_WAVE_SYNC();
    int nk = 0;  // wave-uniform
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        unsigned e = 0;
        bool keep = false;
        if (i < n) {
            e = S.dl[i];
            const int pix = e & 511u, row = vb_div_rw(pix), col = pix - row * VB_RW;
            const unsigned m4 = (e >> 9) & 15u;
            const unsigned m = (e & (1u << 13)) ? m4 : (m4 & (unsigned)(S.need[row] >> col));
            keep = m != 0;
            e = (e & ~(15u << 9)) | (m << 9);
        }
        const u64 km = __ballot(keep);
        if (keep) S.dl[nk + vb_mbcnt(km)] = e;  // in place: nk <= base, and this round's reads are done
        nk += __popcll(km);
    }
    VB_WAVE_SYNC();
    const VbVertsT<LAZY> pv = vb_verts<LAZY>(pvz);  // (lazy: the matrix in scalar registers for the tests below)
    for (int base = 0; base < nk; base += 64) {
        const int i = base + lane;
        if (i < nk) {
            const unsigned e = S.dl[i];
            const int4 vi = cvidx_link[e >> 14];
            const float4 p[3] = {pv[vi.x], pv[vi.y], pv[vi.z]};
            const int pix = e & 511u, row = vb_div_rw(pix), col = pix - row * VB_RW;
            const unsigned m = (e >> 9) & 15u;
            const bool flag = (e >> 13) & 1u;
            // (one pass per set bit of the fullest mask in the wave: after the filter a unit has one or two pixels left)
            unsigned drawn = 0, mm = m;
            while (mm) {
                const int k = __ffs(mm) - 1;
                mm &= mm - 1;
                bool nonpos = false;
                if (vb_depth_test(p, vi.w, rx0 + col + k, ry0 + row, W, H, &key[pix + k], COVER ? &nonpos : nullptr)) drawn |= 1u << k;
                if (COVER && nonpos) S.bad = 1;
            }
            if (flag && drawn) atomicOr((unsigned long long*)&cov[row], (u64)drawn << col);
        }
    }
    VB_WAVE_SYNC();
#ifdef VB_TIMELINE
    if (lane == 0) {
        S.tl_flushes += 1;
        S.tl_tested += nk;
        S.tl_deferred += n;
    }
#endif
    VB_TL_END(S, 3);
}

// One round of the wave-level rasterizer: up to 64 candidate triangles (lane `sv` holds one: record slot `slot` of the
// view, `srel` the same relative to the link's first cluster slot) are rasterized into the job's coverage bitmap and
// their covered 4-pixel units appended to the wave's deferred list (n entries on entry; returns the new count).  The
// triangles' boxes (clamped to the region) are cut into units of 4 horizontally adjacent pixels; a wave-wide prefix sum
// splits the concatenated unit sequence EVENLY over the 64 lanes (a third of the candidates have boxes above 16
// pixels, so one lane per triangle would leave most lanes idle) and each lane walks its contiguous run stepping 32-bit
// edge functions.  Fed by the per-triangle records of the vertex kernel; nothing but LDS is touched inside the walk.
// COVER: the coverage-only form of the scoring op's chain -- no deferred units, no depth anywhere: every triangle must be
// of the class "coverage decides" (kind 0 under the positive-depth test), anything else aborts the job (-1); what is
// already covered (not just interior) hides a box.
template <bool WIDE, bool COVER = false, bool LAZY = false>
__device__ __forceinline__ int vb_raster_round(bool sv, size_t slot, unsigned srel, const VbRecs& rc, const VbRegion& rg,
                                               int rx0, int ry0, int W, int H, VbWaveLds& S, u64* key, u64* cov, int n,
                                               const VbLazy& pv, const int4* __restrict__ cvidx_link, bool& full,
                                               int& cost, int qh) {
    VbRaster& R = S.R;
    const int lane = lane_id();
    VB_TL_BEGIN();
    int units = 0, srows = 0;
    bool wide = false;
    uint2 bx = VB_BOX_EMPTY;
    int4 r0 = make_int4(0, 0, 0, 0), r1 = r0;
    if (sv) {
        bx = rc.tbox[slot];
        r0 = rc.trec[2 * slot];
        r1 = rc.trec[2 * slot + 1];
    }
    // Interior of the coverage so far: covered pixels whose four neighbours (inside the region) are covered too.  Nobody
    // will ever ask which triangle is visible there, and coverage cannot change there: a triangle whose box lies inside
    // the interior is skipped, units inside it are not deferred, and a region that is all interior ends the job (the
    // inner tiles of a link seen from close by: a few large triangles cover everything, the other layers add nothing).
    bool has_in = false;
    {
        u64 in = 0;
        if (lane < VB_RH) {
            const u64 c = cov[lane];
            const u64 up = (lane + 1 < VB_RH) ? cov[lane + 1] : VB_ROW_MASK;
            const u64 dn = (lane > 0) ? cov[lane - 1] : VB_ROW_MASK;
            in = COVER ? (c & VB_ROW_MASK) : (c & ((c >> 1) | (1ull << (VB_RW - 1))) & ((c << 1) | 1ull) & up & dn & VB_ROW_MASK);
            S.intr[lane] = in;
        }
        {
            // a sparse table over the rows (lanes 0-9 hold the rows; lanes 10-15 the identity): "is this box inside the
            // interior" is then two reads and an AND per triangle instead of a pass over all ten rows
            unsigned lo = (lane < VB_RH) ? (unsigned)in : 0xffffffffu, hi = (lane < VB_RH) ? (unsigned)(in >> 32) : 0xffffffffu;
#define VB_ROWS_AND(k, ctrl)                                                                              \
    lo &= (unsigned)__builtin_amdgcn_update_dpp(-1, (int)lo, ctrl, 0xf, 0xf, false);                      \
    hi &= (unsigned)__builtin_amdgcn_update_dpp(-1, (int)hi, ctrl, 0xf, 0xf, false);                      \
    if (lane < VB_RH) S.intr_and[k][lane] = (u64)lo | ((u64)hi << 32);
            VB_ROWS_AND(0, 0x101)  // row_shl:1: lane i <- lane i + 1
            VB_ROWS_AND(1, 0x102)  // row_shl:2
            VB_ROWS_AND(2, 0x104)  // row_shl:4
#undef VB_ROWS_AND
        }
        if (__ballot(lane < VB_RH && in != VB_ROW_MASK) == 0) {
            full = true;
            return n;
        }
        has_in = __ballot(lane < VB_RH && in != 0ull) != 0;  // (no interior yet -- most first rounds --: nothing can be hidden)
        VB_WAVE_SYNC();
    }
    if (sv) {
        if (r1.w == 1) {
            wide = true;
        } else {
            const int ix0 = bx.x & 0xffffu, iy0 = bx.x >> 16, ix1 = bx.y & 0xffffu, iy1 = bx.y >> 16;
            const int cx0 = max(ix0, rg.x0), cy0 = max(iy0, rg.y0);
            const int bw = min(ix1, rg.x1) - cx0 + 1, bh = min(iy1, rg.y1) - cy0 + 1;  // > 0: the boxes overlap
            const unsigned w[3] = {(unsigned)r0.w, (unsigned)r1.x, (unsigned)r1.y};
            const int ev[3] = {r0.x, r0.y, r0.z};
            const int ox = cx0 - ix0, oy = cy0 - iy0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int dX = (int)(short)(w[k] & 0xffffu), dY = (int)(short)(w[k] >> 16);
                R.e[lane][k] = ev[k] - 16 * __mul24(dY, ox) + 16 * __mul24(dX, oy);  // (24-bit operands: full-rate multiplies)
                R.dxy[lane][k] = w[k];
            }
            R.box[lane] = (unsigned)(cx0 - rx0) | ((unsigned)(cy0 - ry0) << 8) | ((unsigned)bw << 16) | ((unsigned)bh << 24);
            R.ent[lane] = (srel << 14) | ((r1.w == 2) ? (1u << 13) : 0u);
            const u64 bm = ((1ull << bw) - 1ull) << (cx0 - rx0);
            bool hidden = has_in;
            if (has_in) {
                // rows y0 .. y0 + bh - 1 = the 2^k rows from y0 and the 2^k rows up to the last, k = floor(log2(bh))
                const int y0r = cy0 - ry0, k = 31 - __clz(bh);
                const u64* const tab = (k == 0) ? S.intr : S.intr_and[k - 1];
                const u64 all = tab[y0r] & tab[y0r + bh - (1 << k)];
                hidden = (all & bm) == bm;
            }
            // a box of VB_SPAN_GW or more units per row goes to the span walker: work = its rows
            if (!hidden) {
                if (((bw + 3) >> 2) >= VB_SPAN_GW && !(COVER && r1.w == 2))  // (coverage-only form: flagged boxes stay with the units)
                    srows = bh;
                else
                    units = ((bw + 3) >> 2) * bh;
            }
        }
    }
    if ((!WIDE || COVER) && __ballot(wide)) return -1;  // the lean instantiation hands the whole job to vb_job_slow
    // one scan for both walkers: units in the low half (<= 64 x 90), span rows in the high half (<= 64 x 10)
    const int packed = units | (srows << 16);
    const int incl = vb_scan_add(packed);
    const int Ptot = vb_readlane(incl, 63);
    const int Stot = Ptot & 0xffff, Wtot = Ptot >> 16;
#if VB_FAST_SEARCH
    const u64 nzu = __ballot(units > 0), nzs = __ballot(srows > 0);  // triangles with units / with span rows
#endif
    cost += Stot + 3 * Wtot + 256;  // what the job costs a wave: a step per 64 units, three per 64 span rows, about four per round
#ifdef VB_TIMELINE
    if (lane == 0) {
        S.tl_units += Stot;
        S.tl_rounds += 1;
    }
#endif
    VB_TL_END(S, 0);
    if (Ptot > 0) {
        VB_TL_BEGIN();
        R.pre[lane] = incl - packed;
        if (lane == 63) R.pre[64] = Ptot;
        VB_WAVE_SYNC();
        const int K = (Stot + 63) >> 6;
        const int start = __mul24(lane, K), end = min(start + K, Stot);
        int j = 0, bw = 1, bh = 1, gw = 1, gx = 0, dy = 0, crow = 0, ccol0 = 0;
        unsigned eb = 0, lastm = 15u;  // lastm: the pixels of a row's last unit that lie inside the box
        int e0 = -1, e1 = -1, e2 = -1, sx0 = 0, sx1 = 0, sx2 = 0, sy0 = 0, sy1 = 0, sy2 = 0, er0 = 0, er1 = 0, er2 = 0;
#if VB_FAST_SEARCH
        // Which triangle does a lane start in?  Asked the other way round: triangle j's units begin at pre[j]; the first
        // lane to start at or after that is ceil(pre[j] / K), and if that start still lies inside the triangle, j puts its
        // number there.  A lane's triangle is the last number at or below its own place: a max-scan.  (One LDS round
        // trip instead of the six of a binary search over pre[] -- in every round of every job.  The table is the half of
        // the survivor ring this round has just emptied.)
        unsigned* const own = S.sq + qh;
        {
            const float invK = __builtin_amdgcn_rcpf((float)K);
            const int pj = (incl - packed) & 0xffff, uj = packed & 0xffff;
            const int wf = vb_div_small(pj + K - 1, invK);
            own[lane] = 0u;
            if (uj > 0 && __mul24(wf, K) < pj + uj) own[wf] = (unsigned)lane + 1u;
            VB_WAVE_SYNC();
            j = vb_scan_max((int)own[lane]) - 1;
        }
        if (start < end) {
#else
        if (start < end) {
            int lo = 0, hi = 63;
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int mid = (lo + hi + 1) >> 1;
                if ((R.pre[mid] & 0xffff) <= start)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            j = lo;
#endif
            const unsigned b4 = R.box[j];
            ccol0 = b4 & 255;
            const int y0r = (b4 >> 8) & 255;
            bw = (b4 >> 16) & 255;
            bh = b4 >> 24;
            gw = (bw + 3) >> 2;
            lastm = (1u << (bw - 4 * (gw - 1))) - 1u;
            eb = R.ent[j];
            const unsigned w0 = R.dxy[j][0], w1 = R.dxy[j][1], w2 = R.dxy[j][2];
            sx0 = -16 * (int)(short)(w0 >> 16); sy0 = 16 * (int)(short)(w0 & 0xffffu);
            sx1 = -16 * (int)(short)(w1 >> 16); sy1 = 16 * (int)(short)(w1 & 0xffffu);
            sx2 = -16 * (int)(short)(w2 >> 16); sy2 = 16 * (int)(short)(w2 & 0xffffu);
            const int o = start - (R.pre[j] & 0xffff);
#if VB_FAST_SEARCH
            dy = vb_div_small(o, __builtin_amdgcn_rcpf((float)gw));
#else
            dy = o / gw;
#endif
            gx = o - __mul24(dy, gw);  // (|s*| < 2^20, dy < 2^7, gx < 4: 24-bit operands, full-rate multiplies)
            er0 = R.e[j][0] + __mul24(dy, sy0);
            er1 = R.e[j][1] + __mul24(dy, sy1);
            er2 = R.e[j][2] + __mul24(dy, sy2);
            e0 = er0 + __mul24(4 * gx, sx0);
            e1 = er1 + __mul24(4 * gx, sx1);
            e2 = er2 + __mul24(4 * gx, sx2);
            crow = y0r + dy;
        }
        VB_TL_END(S, 1);
#ifdef VB_TIMELINE
        const long long tl_w0 = __builtin_readcyclecounter();
#endif
        for (int it0 = 0; it0 < K;) {
            // the deferred list takes at most 64 entries per step: walk as many steps as it has room for
            const int room = (VB_DL - n) >> 6;
            if (room == 0) {
                vb_flush<COVER, LAZY>(S, key, cov, n, pv, cvidx_link, W, H, rx0, ry0);
                n = 0;
                continue;
            }
            const int it1 = min(K, it0 + room);
#pragma nounroll
            for (int it = it0; it < it1; it++) {
                const bool act = start + it < end;
                unsigned m4 = 0;
                if (act) {
                    const int a1 = e0 + sx0, a2 = a1 + sx0, a3 = a2 + sx0;
                    const int b1 = e1 + sx1, b2 = b1 + sx1, b3 = b2 + sx1;
                    const int c1 = e2 + sx2, c2 = c1 + sx2, c3 = c2 + sx2;
                    // a pixel is inside if none of its three edge values is negative: the four sign bits, funnelled into
                    // one word by three double-word shifts
                    unsigned sg = (unsigned)(a3 | b3 | c3) >> 31;
                    sg = __builtin_amdgcn_alignbit(sg, (unsigned)(a2 | b2 | c2), 31);
                    sg = __builtin_amdgcn_alignbit(sg, (unsigned)(a1 | b1 | c1), 31);
                    sg = __builtin_amdgcn_alignbit(sg, (unsigned)(e0 | e1 | e2), 31);
                    m4 = ~sg & ((gx == gw - 1) ? lastm : 15u);  // (the row's last unit: only its pixels inside the box)
                }
                const bool inside = m4 != 0;
                const u64 m = __ballot(inside);
                if (COVER && !(__ballot(inside && (eb & (1u << 13))))) {
                    if (inside) atomicOr((unsigned long long*)&cov[crow], (u64)m4 << (ccol0 + 4 * gx));
                } else if (m) {
                    // deferred for a depth test: the unit's covered pixels outside the interior (as of the round's start)
                    unsigned md = 0;
                    if (inside) {
                        const int ccol = ccol0 + 4 * gx;
                        if (!(eb & (1u << 13))) atomicOr((unsigned long long*)&cov[crow], (u64)m4 << ccol);
                        md = COVER ? ((eb & (1u << 13)) ? m4 : 0u) : (m4 & ~(unsigned)(S.intr[crow] >> ccol));
                    }
                    const u64 ma = __ballot(md != 0);
                    if (md) S.dl[n + vb_mbcnt(ma)] = eb | (unsigned)(__mul24(crow, VB_RW) + ccol0 + 4 * gx) | (md << 9);
                    n += __popcll(ma);
                }
                if (act && start + it + 1 < end) {
                    gx++;
                    e0 += 4 * sx0;
                    e1 += 4 * sx1;
                    e2 += 4 * sx2;
                    if (gx == gw) {
                        gx = 0;
                        dy++;
                        crow++;
                        er0 += sy0;
                        er1 += sy1;
                        er2 += sy2;
                        e0 = er0;
                        e1 = er1;
                        e2 = er2;
                        if (dy == bh) {  // next job with a non-empty box
#if VB_FAST_SEARCH
                            j = vb_next_set(nzu, j);
#else
                            do {
                                j++;
                            } while (((R.pre[j + 1] ^ R.pre[j]) & 0xffff) == 0);
#endif
                            const unsigned b4 = R.box[j];
                            ccol0 = b4 & 255;
                            crow = (b4 >> 8) & 255;
                            bw = (b4 >> 16) & 255;
                            bh = b4 >> 24;
                            gw = (bw + 3) >> 2;
                            lastm = (1u << (bw - 4 * (gw - 1))) - 1u;
                            eb = R.ent[j];
                            const unsigned w0 = R.dxy[j][0], w1 = R.dxy[j][1], w2 = R.dxy[j][2];
                            sx0 = -16 * (int)(short)(w0 >> 16); sy0 = 16 * (int)(short)(w0 & 0xffffu);
                            sx1 = -16 * (int)(short)(w1 >> 16); sy1 = 16 * (int)(short)(w1 & 0xffffu);
                            sx2 = -16 * (int)(short)(w2 >> 16); sy2 = 16 * (int)(short)(w2 & 0xffffu);
                            er0 = e0 = R.e[j][0];
                            er1 = e1 = R.e[j][1];
                            er2 = e2 = R.e[j][2];
                            dy = 0;
                        }
                    }
                }
            }
            it0 = it1;
        }
        // ---- span walker: the triangles whose box is VB_SPAN_GW or more units wide (long thin ones mostly: 17 % of the rows
        //      but 45 % of the units, and most of those units empty).  A lane takes rows instead of units and solves each
        //      row's covered span [lo, hi] from the three edge functions: floor(E / |sx|) from a float estimate, made exact by
        //      one integer step (the estimate is within 1e-5 of the quotient wherever the result matters, |q| <= 40).  The
        //      span goes into the bitmap with one OR; its 4-pixel units that are not interior are deferred as usual.
        if (Wtot > 0) {
            const int K2 = (Wtot + 63) >> 6;
            const int s2 = __mul24(lane, K2), e2 = min(s2 + K2, Wtot);
            int j = 0, bw = 1, bh = 1, dy = 0, crow = 0, ccol0 = 0;
            unsigned eb = 0;
            int sx0 = 0, sx1 = 0, sx2 = 0, sy0 = 0, sy1 = 0, sy2 = 0, er0 = 0, er1 = 0, er2 = 0;
            float iv0 = 0.f, iv1 = 0.f, iv2 = 0.f;
            auto load_tri = [&](int jj) {
                const unsigned b4 = R.box[jj];
                ccol0 = b4 & 255;
                crow = (b4 >> 8) & 255;
                bw = (b4 >> 16) & 255;
                bh = b4 >> 24;
                eb = R.ent[jj];
                const unsigned w0 = R.dxy[jj][0], w1 = R.dxy[jj][1], w2 = R.dxy[jj][2];
                sx0 = -16 * (int)(short)(w0 >> 16); sy0 = 16 * (int)(short)(w0 & 0xffffu);
                sx1 = -16 * (int)(short)(w1 >> 16); sy1 = 16 * (int)(short)(w1 & 0xffffu);
                sx2 = -16 * (int)(short)(w2 >> 16); sy2 = 16 * (int)(short)(w2 & 0xffffu);
                iv0 = __builtin_amdgcn_rcpf((float)abs(sx0));
                iv1 = __builtin_amdgcn_rcpf((float)abs(sx1));
                iv2 = __builtin_amdgcn_rcpf((float)abs(sx2));
                er0 = R.e[jj][0];
                er1 = R.e[jj][1];
                er2 = R.e[jj][2];
                dy = 0;
            };
#if VB_FAST_SEARCH
            {
                const float invK = __builtin_amdgcn_rcpf((float)K2);
                const int pj = (incl - packed) >> 16, uj = packed >> 16;
                const int wf = vb_div_small(pj + K2 - 1, invK);
                VB_WAVE_SYNC();  // (the unit walker's reads of the table are complete)
                own[lane] = 0u;
                if (uj > 0 && __mul24(wf, K2) < pj + uj) own[wf] = (unsigned)lane + 1u;
                VB_WAVE_SYNC();
                j = vb_scan_max((int)own[lane]) - 1;
            }
            if (s2 < e2) {
#else
            if (s2 < e2) {
                int lo = 0, hi = 63;
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const int mid = (lo + hi + 1) >> 1;
                    if ((R.pre[mid] >> 16) <= s2)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                j = lo;
#endif
                load_tri(j);
                dy = s2 - (R.pre[j] >> 16);
                er0 += __mul24(dy, sy0);
                er1 += __mul24(dy, sy1);
                er2 += __mul24(dy, sy2);
                crow += dy;
            }
#pragma nounroll
            for (int it = 0; it < K2;) {
                const bool act = s2 + it < e2;
                int lo = 0, hi = -1;
                if (act) {
                    hi = bw - 1;
#define VB_SPAN_EDGE(E, SX, IV)                                                                    \
    if ((SX) == 0) {                                                                               \
        if ((E) < 0) hi = -1;                                                                      \
    } else {                                                                                       \
        const int asx = abs(SX);                                                                   \
        const float qf = fminf(fmaxf(floorf((float)(E) * (IV)), -40.f), 40.f);                     \
        int u = (int)qf;                                                                           \
        const int t = (E) - __mul24(u, asx);                                                       \
        u += (t < 0) ? -1 : ((t >= asx) ? 1 : 0);                                                  \
        if ((SX) > 0)                                                                              \
            lo = max(lo, -u);                                                                      \
        else                                                                                       \
            hi = min(hi, u);                                                                       \
    }
                    VB_SPAN_EDGE(er0, sx0, iv0)
                    VB_SPAN_EDGE(er1, sx1, iv1)
                    VB_SPAN_EDGE(er2, sx2, iv2)
#undef VB_SPAN_EDGE
                }
                u64 rm = 0;
                int c0 = 0;
                if (lo <= hi) {
                    c0 = ccol0 + lo;
                    rm = ((1ull << (hi - lo + 1)) - 1ull) << c0;
                }
                if (COVER) {
                    if (rm) atomicOr((unsigned long long*)&cov[crow], rm);
                    if (act && s2 + it + 1 < e2) {
                        dy++;
                        crow++;
                        er0 += sy0;
                        er1 += sy1;
                        er2 += sy2;
                        if (dy == bh) {
#if VB_FAST_SEARCH
                            j = vb_next_set(nzs, j);
#else
                            do {
                                j++;
                            } while ((R.pre[j + 1] >> 16) == (R.pre[j] >> 16));
#endif
                            load_tri(j);
                        }
                    }
                    it++;
                    continue;
                }
                // deferred: the span's units (aligned to the span's first pixel) with a pixel outside the interior
                const u64 md = rm ? ((rm & ~S.intr[crow]) >> c0) : 0ull;
                u64 nz = (md | (md >> 1) | (md >> 2) | (md >> 3)) & 0x1111111111111111ull;
                const int cnt = __popcll(nz);
                const int inc = vb_scan_add(cnt);
                const int T = vb_readlane(inc, 63);
                if (n + T > VB_DL) {  // no room for this step's entries: flush, then take the step again
                    vb_flush<COVER, LAZY>(S, key, cov, n, pv, cvidx_link, W, H, rx0, ry0);
                    n = 0;
                    continue;
                }
                if (rm && !(eb & (1u << 13))) atomicOr((unsigned long long*)&cov[crow], rm);
                int pos = n + inc - cnt;
                while (nz) {
                    const int bpos = __ffsll((unsigned long long)nz) - 1;
                    nz &= nz - 1;
                    S.dl[pos++] = eb | (unsigned)(crow * VB_RW + c0 + bpos) | ((unsigned)((md >> bpos) & 15ull) << 9);
                }
                n += T;
                if (act && s2 + it + 1 < e2) {
                    dy++;
                    crow++;
                    er0 += sy0;
                    er1 += sy1;
                    er2 += sy2;
                    if (dy == bh) {  // next triangle of this class
#if VB_FAST_SEARCH
                        j = vb_next_set(nzs, j);
#else
                        do {
                            j++;
                        } while ((R.pre[j + 1] >> 16) == (R.pre[j] >> 16));
#endif
                        load_tri(j);
                    }
                }
                it++;
            }
        }
        VB_WAVE_SYNC();  // the staging area is rewritten by the next round
#ifdef VB_TIMELINE
        if (lane == 0) S.tl_c[2] += __builtin_readcyclecounter() - tl_w0;
#endif
    }
    if (WIDE) {  // rare: near-plane clipping / very large extents, one triangle at a time, depth tested at once
        u64 wm = __ballot(wide);
        while (wm) {
            const int s = __ffsll((unsigned long long)wm) - 1;
            wm &= wm - 1;
            const int4 vi = cvidx_link[vb_readlane((int)srel, s)];
            const VbVertsT<LAZY> pvv = vb_verts<LAZY>(pv);
            vb_raster_wide(pvv[vi.x], pvv[vi.y], pvv[vi.z], vi.w, W, H, rg, rx0, ry0, key, cov);
        }
    }
    return n;
}

struct alignas(16) VbResolveLds {  // per wave of the resolve kernel
    unsigned ids[VB_RN];         // triangle id of each region pixel (all-ones = uncovered), copied from the job's slot
    float pairA[2 * VB_RN];      // blend weight of pair (q, d) at [d * RN + q]
    unsigned short hits[2 * VB_RN];
};

// tiles (+ 1-pixel halo) a link's pixel box touches: the jobs of that (view, link)
// (halo = 0: the scoring op's coverage-only jobs, which look at their tile alone)
__device__ __forceinline__ bool vb_unit_tiles(const int* __restrict__ bx, int W, int H, int& tx0, int& ty0, int& nx, int& ny,
                                              int halo = 1) {
    const int x0 = bx[0], y0 = bx[1], x1 = bx[2], y1 = bx[3];
    if (x0 > x1 || y0 > y1) return false;
    tx0 = max(x0 - halo, 0) / EHR_TILE_W;
    ty0 = max(y0 - halo, 0) / EHR_TILE_H;
    nx = min(x1 + halo, W - 1) / EHR_TILE_W - tx0 + 1;
    ny = min(y1 + halo, H - 1) / EHR_TILE_H - ty0 + 1;
    return true;
}

// Kernel-wide arguments of a job (what the job kernel's helpers need besides the job itself).
#ifndef VB_CULL_BATCHES
#define VB_CULL_BATCHES 2  // batches of 64 cluster boxes requested together
#endif
#ifndef VB_CULL_GROUP
#define VB_CULL_GROUP 8  // candidate clusters whose triangle boxes are requested together
#endif
struct VbJobArgs {
    VbRecs rc;
    const float* verts;   // [V][3] object-space vertices
    const float* mvp;     // [B][L][16] the chunk's (view, link) matrices: clip-space vertices are computed on demand (VbLazy) ...
    const float4* posc;   // ... or read from here ([B][V]) where the plan keeps them (NULL: lazy)
    const int4* cvidx;    // [NC * 64] {v0, v1, v2, triangle} of every cluster slot
    const int* lcoff;     // LDS: first cluster of every link
    unsigned* jid;        // job slots: triangle ids of the region's pixels
    u64* jcov;            //            coverage bitmap (region-linear, VB_WORDS words)
    int* jdesc;
    int* jn;
    int NC, V, W, H, L;
};

// Culls the link's clusters and triangles against the job's region and rasterizes this wave's share of them into the
// job's coverage bitmap `cov_` / depth-id buffer `key_` (LDS, initialised by the caller).  share / nshare: the wave takes
// every nshare-th candidate cluster -- 0 / 1 for a job of its own, wave / 4 when the whole workgroup works on one heavy
// job.  The survivors' covered units are left in the wave's deferred list (dln entries on return): the caller flushes it
// once the job's coverage is complete.  Returns 1 if anything survived the culling, 0 if not, and -1 (lean
// instantiation only) if a survivor needs the general path: the caller then redoes the job with vb_job_slow.
template <bool WIDE, bool COVER = false, bool LAZY = false>
__device__ __forceinline__ int vb_job_raster(const VbJobArgs& A, VbWaveLds& W_, u64* key_, u64* cov_, int b, int l,
                                             const VbRegion& rg, int rx0, int ry0, int share, int nshare, int& nsurv,
                                             int& dln) {
    const int lane = lane_id();
    const int c0 = A.lcoff[l], c1 = A.lcoff[l + 1];
    const uint2* const cb = A.rc.cbox + (size_t)b * A.NC;
    const unsigned rlo = (unsigned)rg.x0 | ((unsigned)rg.y0 << 16), rhi = (unsigned)rg.x1 | ((unsigned)rg.y1 << 16);
    // Survivors of the triangle-box test are queued (record slots) until 64 are waiting, so that every round of
    // the rasterizer is full; the triangle boxes of up to VB_CULL_GROUP candidate clusters are fetched per round trip.
    const size_t vbase = (size_t)b * A.NC * 64;
    const VbLazy pv = {A.verts, A.mvp + ((size_t)b * A.L + l) * 16, A.posc ? A.posc + (size_t)b * A.V : nullptr};
    const int4* const cvl = A.cvidx + (size_t)c0 * 64;  // the link's first cluster slot
    const unsigned srel0 = (unsigned)c0 * 64u;
    bool full = false;   // set by a round that finds the whole region interior
    int qh = 0, qn = 0;  // wave-uniform ring state
    int cord = 0;        // running ordinal of the candidate clusters
    bool drawn = false;
    // The cluster boxes of VB_CULL_BATCHES x 64 clusters are requested together (one round trip under load is ~1-2 us, and
    // all but one of the xArm7's links have between 65 and 128 clusters), then the batches are worked off one by one.
    for (int cb0 = c0; cb0 < c1; cb0 += 64 * VB_CULL_BATCHES) {
        u64 cmk[VB_CULL_BATCHES];
        {
            uint2 bxk[VB_CULL_BATCHES];
#pragma unroll
            for (int k = 0; k < VB_CULL_BATCHES; k++) {
                const int c = cb0 + 64 * k + lane;
                bxk[k] = VB_BOX_EMPTY;
                if (c < c1) bxk[k] = cb[c];
            }
#pragma unroll
            for (int k = 0; k < VB_CULL_BATCHES; k++)
                cmk[k] = __ballot((bxk[k].x & 0xffffu) <= (rhi & 0xffffu) && (bxk[k].y & 0xffffu) >= (rlo & 0xffffu) &&
                                  (bxk[k].x >> 16) <= (rhi >> 16) && (bxk[k].y >> 16) >= (rlo >> 16));
        }
#pragma nounroll
        for (int kb = 0; kb < VB_CULL_BATCHES; kb++) {
            const int cbase = cb0 + 64 * kb;
            u64 cm = cmk[0];
#pragma unroll
            for (int kk = 1; kk < VB_CULL_BATCHES; kk++)
                if (kb == kk) cm = cmk[kk];
            if (!cm) continue;
            if (nshare > 1) {  // cooperative job: this wave takes every nshare-th candidate cluster
                u64 mine = 0;
                u64 all = cm;
                while (all) {
                    const u64 low = all & (~all + 1);
                    if ((cord++ % nshare) == share) mine |= low;
                    all ^= low;
                }
                cm = mine;
            }
            while (cm) {  // wave-uniform
                // A group of up to VB_CULL_GROUP candidate clusters: their triangle boxes in one round trip, the survivors
                // of ALL of them pushed in one unrolled pass when the ring has room for them (the rule: the ring holds 256),
                // then as many rounds as are full.  Masks and cluster numbers are scalar registers that die before the
                // first round starts.  (Per cluster -- read the mask back, test, push, count, test for a round -- the loop
                // was ~30 instructions x 33 k candidates a step, and 25 of the 94 kcycles of the jobs the kernel ends on.)
                // Only when the ring cannot take the whole group do they wait in the LANES of three vector registers
                // (lane k: entry k) and go in one by one, a round whenever 64 are waiting.
                int ccv = 0, ng = 0, knext = 0;
                unsigned mlo = 0, mhi = 0;
#ifdef VB_TIMELINE
                const long long tl_g0 = __builtin_readcyclecounter();
#endif
                {
                    int cks[VB_CULL_GROUP];
                    uint2 tb[VB_CULL_GROUP];
#pragma unroll
                    for (int k = 0; k < VB_CULL_GROUP; k++) {
                        tb[k] = VB_BOX_EMPTY;
                        cks[k] = 0;
                        if (cm) {
                            cks[k] = cbase + __ffsll((unsigned long long)cm) - 1;
                            cm &= cm - 1;
                            tb[k] = A.rc.tbox[vbase + (size_t)cks[k] * 64 + lane];
                            ng = k + 1;
                        }
                    }
                    u64 mk[VB_CULL_GROUP];
                    int total = 0;
#pragma unroll
                    for (int k = 0; k < VB_CULL_GROUP; k++) {
                        mk[k] = 0;
                        if (k < ng) {
                            mk[k] = __ballot((tb[k].x & 0xffffu) <= (rhi & 0xffffu) && (tb[k].y & 0xffffu) >= (rlo & 0xffffu) &&
                                             (tb[k].x >> 16) <= (rhi >> 16) && (tb[k].y >> 16) >= (rlo >> 16));
                            total += __popcll(mk[k]);
                        }
                    }
#ifdef VB_TIMELINE
                    if (lane == 0) {
                        W_.tl_c[7] += __builtin_readcyclecounter() - tl_g0;
                        W_.tl_cands += ng;
                        W_.tl_groups += 1;
                    }
#endif
                    if (total == 0) continue;
                    drawn = true;
                    if (qn + total <= VB_SQ) {
#pragma unroll
                        for (int k = 0; k < VB_CULL_GROUP; k++) {
                            if (mk[k]) {
                                if ((mk[k] >> lane) & 1) W_.sq[(qh + qn + vb_mbcnt(mk[k])) & (VB_SQ - 1)] = (unsigned)(cks[k] * 64 + lane);
                                qn += __popcll(mk[k]);
                            }
                        }
                        knext = ng;
                    } else {
#pragma unroll
                        for (int k = 0; k < VB_CULL_GROUP; k++) {
                            ccv = (lane == k) ? cks[k] : ccv;
                            mlo = (lane == k) ? (unsigned)mk[k] : mlo;
                            mhi = (lane == k) ? (unsigned)(mk[k] >> 32) : mhi;
                        }
                    }
                }
                for (;;) {
                    // not unrolled, one call site: one copy of the rasterizer round (the kernel was 69 KB of code, more
                    // than the instruction cache two CUs share)
                    while (qn >= 64) {
                        VB_WAVE_SYNC();
                        const unsigned sl = W_.sq[(qh + lane) & (VB_SQ - 1)];
                        dln = vb_raster_round<WIDE, COVER, LAZY>(true, vbase + sl, sl - srel0, A.rc, rg, rx0, ry0, A.W, A.H, W_, key_, cov_, dln, pv, cvl, full, nsurv, qh);
                        if (dln < 0) return -1;
                        if (full) return 1;  // every pixel of the region is interior: nothing can change any more
                        qh = (qh + 64) & (VB_SQ - 1);
                        qn -= 64;
                    }
                    if (knext >= ng) break;
                    const u64 sm = (u64)(unsigned)__builtin_amdgcn_readlane((int)mlo, knext) | ((u64)(unsigned)__builtin_amdgcn_readlane((int)mhi, knext) << 32);
                    const int ck = __builtin_amdgcn_readlane(ccv, knext);
                    knext++;
                    if (sm) {
                        if ((sm >> lane) & 1) W_.sq[(qh + qn + vb_mbcnt(sm)) & (VB_SQ - 1)] = (unsigned)(ck * 64 + lane);
                        qn += __popcll(sm);
                    }
                }
            }
        }
    }
    if (qn) {
        VB_WAVE_SYNC();
        const bool sv = lane < qn;
        const unsigned sl = sv ? W_.sq[(qh + lane) & (VB_SQ - 1)] : srel0;
        dln = vb_raster_round<WIDE, COVER, LAZY>(sv, vbase + sl, sl - srel0, A.rc, rg, rx0, ry0, A.W, A.H, W_, key_, cov_, dln, pv, cvl, full, nsurv, qh);
        if (dln < 0) return -1;
    }
    return drawn ? 1 : 0;
}

// A drawn job leaves, in its slot, the coverage rows of its region and the triangle id of every pixel the depth test ran
// for (all-ones elsewhere), then its descriptor; the resolve kernel takes it from there (an undrawn job's descriptor is
// -1).  No list of drawn jobs: appending to one costs every job a returning atomic (~3 us under load), and three
// quarters of the jobs are drawn anyway.
// (part / nparts: the words this wave writes -- 0 / 1 for a job of its own; the four waves of a heavy job share them)
__device__ __forceinline__ void vb_publish(const VbJobArgs& A, const u64* key_, const u64* cov_, int job, int u, int tx,
                                           int ty, int part = 0, int nparts = 1) {
    const int lane = lane_id();
    VB_WAVE_SYNC();
    unsigned* const dst = A.jid + (size_t)job * VB_RN;
#pragma unroll
    for (int k = 0; k < VB_WORDS; k++) {
        if (nparts > 1 && (k % nparts) != part) continue;
        const unsigned i = 64u * k + lane;
        if (i < (unsigned)VB_RN) dst[i] = (unsigned)key_[i];  // low word = triangle id; all-ones stays all-ones
    }
    // coverage in region-linear order (bit i = region pixel i), what the resolve kernel's bit arithmetic works on
#pragma unroll
    for (int k = 0; k < VB_WORDS; k++) {
        if (nparts > 1 && (k % nparts) != part) continue;
        const unsigned i = 64u * k + lane;
        const unsigned row = (unsigned)vb_div_rw((int)i), col = i - row * VB_RW;
        const u64 w = __ballot(i < (unsigned)VB_RN && ((cov_[row < (unsigned)VB_RH ? row : 0] >> col) & 1ull));
        if (lane == 0) A.jcov[(size_t)job * VB_WORDS + k] = w;
    }
    if (lane == 0 && part == 0) A.jdesc[job] = u | (tx << 9) | (ty << 19);
}

// A whole job on one wave with the general triangle path compiled in (near-plane clipping, 64-bit edge functions): what a
// job falls back to when the lean code meets such a triangle.  Runs in a kernel of its own (vb_slow_kernel) over the list
// of such jobs: compiled into the job kernel -- inline or as a call, in the rounds or at the job loop's end -- the general
// path cost the lean code 50-70 spilled registers and 13 us at 8 views.
template <bool LAZY>
__device__ __forceinline__ void vb_job_slow(const VbJobArgs& A, VbWaveLds& S, int job, int u, int tx, int ty) {
    const int lane = lane_id();
    const int b = u / A.L, l = u - b * A.L;
    const int rx0 = tx * EHR_TILE_W - 1, ry0 = ty * EHR_TILE_H - 1;
    VbRegion rg;
    rg.x0 = max(rx0, 0);
    rg.y0 = max(ry0, 0);
    rg.x1 = min(rx0 + VB_RW - 1, A.W - 1);
    rg.y1 = min(ry0 + VB_RH - 1, A.H - 1);
    VB_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < VB_WORDS; k++) {
        const unsigned i = 64u * k + lane;
        if (i < (unsigned)VB_RN) S.key[i] = VB_EMPTY;
    }
    if (lane < VB_RH) S.cov[lane] = 0ull;
    VB_WAVE_SYNC();
    int nsurv = 0, dln = 0;
    const int drawn = vb_job_raster<true, false, LAZY>(A, S, S.key, S.cov, b, l, rg, rx0, ry0, 0, 1, nsurv, dln);
    if (dln > 0) vb_flush<false, LAZY>(S, S.key, S.cov, dln, VbLazy{A.verts, A.mvp + ((size_t)b * A.L + l) * 16, A.posc ? A.posc + (size_t)b * A.V : nullptr}, A.cvidx + (size_t)A.lcoff[l] * 64, A.W, A.H, rx0, ry0);
    if (drawn > 0) {
        vb_publish(A, S.key, S.cov, job, u, tx, ty);
    } else if (lane == 0) {
        A.jn[job] = -1;
        A.jdesc[job] = -1;
    }
}

// Stage 2: one WAVE per job = (view, link, 32x8 tile the link's screen box touches); persistent waves over the job list,
// which is never materialised (every workgroup derives it from the link boxes with a prefix sum over B * L counts).
// A job culls the link's cluster boxes, then the triangle boxes of the surviving clusters, and rasterizes the survivors
// into the wave's LDS depth/id buffer (tile + 1-pixel halo).  It leaves, in the job's slot (view, link, tile), the
// triangle id of every region pixel (jid) and a descriptor (jdesc; -1 and jn = -1 when nothing was drawn); the resolve
// kernel takes it from there.  No workgroup barriers after the prologue except in the heavy-job phase; thousands of
// independent waves hide each other's latency.
// A job that met a triangle for the general path goes on the list vb_slow_kernel works off.  The solver-step form of the
// chain does not launch that kernel until a step has needed it (slow_list == NULL: an empty launch costs the step 1.7 us
// and a robot in front of the camera never has such a triangle): then the job is marked empty and the step REPORTS it --
// overflow bit 4, so loss and gradient come out NaN and the optimiser state stays as it was; ehr_fused_status() returns
// EHR_ERR_RETRY and switches the pass on for the context's later calls.
#define VB_FLAG_NEED_SLOW 4
template <bool COH = false>
__device__ __forceinline__ void vb_put_aside(int4* __restrict__ slow_list, int* __restrict__ meta, int* __restrict__ jn,
                                             int* __restrict__ jdesc, int job, int u, int tx, int ty) {
    if (slow_list) {
        slow_list[atomicAdd(vb_line(meta, 17), 1)] = make_int4(job, u, tx, ty);
    } else {
        atomicOr(&meta[EHR_META_OVERFLOW], VB_FLAG_NEED_SLOW);
        vb_st_i32<COH>(&jn[job], -1);
        jdesc[job] = -1;
    }
}

// Kernel-wide arguments of the resolve stage (what vb_resolve_job needs besides the job itself).
struct VbResolveArgs {
    const float* verts;   // [V][3] object-space vertices
    const float* mvp;     // [B][L][16] (clip-space vertices on demand: VbLazy) ...
    const float4* posc;   // ... or kept: [B][V] (NULL: lazy)
    int L;
    const int4* tri4;     // [T] padded index table
    const int4* opp4;     // [T] opposite vertices (edge topology)
    int* jn;              // job slots: number of blended pairs (-1: the link contributes nothing to the tile)
    float* jval;          //            the link's 256 antialiased values
    VbItem* jitems;       //            blended pairs for the backward pass
    int* jspill;
    VbItem* spill;
    int* meta;
    int V, T, W, H, spill_cap, want_grad, dbg;
};

// The resolve stage of ONE drawn job, by one wave, from LDS: `ids` = triangle id of every region pixel (all-ones =
// uncovered, VB_ID_COVERED = covered but never depth tested: no uncovered neighbour), C = the region's coverage bitmap
// (bit i = region pixel i; wave-uniform).  Covered/uncovered pixel pairs by wave-uniform bit arithmetic, silhouette
// analysis of the compacted hits (restates nvdiffrast's antialias mesh kernel), gather of the link's antialiased value per
// pixel in the oracle's order.  Leaves in the job's slot the 256 values (jval), the blended pairs the backward pass needs
// (jitems) and their number (jn; -1 = the link contributes nothing here).  pairA [2 * VB_RN] and hits [2 * VB_RN] are the
// wave's LDS work areas.  Called by the job kernel right after a job's depth tests (the ids never leave LDS) and by
// vb_resolve_kernel for the jobs vb_slow_kernel drew.
template <bool COH = false, bool LAZY = false>
__device__ __forceinline__ void vb_resolve_job(const VbResolveArgs& Q, const unsigned* ids, float* pairA,
                                               unsigned short* hits, const u64 (&C)[VB_WORDS], size_t slot, int b,
                                               int l, int rx0, int ry0) {
    const int lane = lane_id();
    const int4* const tri4 = Q.tri4;
    const int4* const opp4 = Q.opp4;
    int* const jn = Q.jn;
    float* const jval = Q.jval;
    VbItem* const jitems = Q.jitems;
    int* const jspill = Q.jspill;
    VbItem* const spill = Q.spill;
    int* const meta = Q.meta;
    const int V = Q.V, T = Q.T, W = Q.W, H = Q.H, spill_cap = Q.spill_cap, want_grad = Q.want_grad, dbg = Q.dbg;
#define KT(i) (ids[i])
    const int r = lane >> 3, c4 = (lane & 7) * 4;
    const int myq = (r + 1) * VB_RW + (c4 + 1);
    const VbLazy pvz = {Q.verts, Q.mvp + ((size_t)b * Q.L + l) * 16, Q.posc ? Q.posc + (size_t)b * V : nullptr};
    int nitems = 0;       // wave-uniform
    int spill_base = -1;  // wave-uniform: first item of this job's spill block, once one was needed
    VB_WAVE_SYNC();
    for (int i = lane; i < 2 * VB_RN; i += 64) pairA[i] = 0.f;
    // region pixels inside the image (wave-uniform bitmap) and the pair-validity bitmaps derived from it
    u64 Iw[VB_WORDS];
#pragma unroll
    for (int k = 0; k < VB_WORDS; k++) {
        const unsigned i = 64u * k + lane;
        const int qy = vb_div_rw((int)i), qx = (int)i - qy * VB_RW;
        const int x = rx0 + qx, y = ry0 + qy;
        Iw[k] = __ballot(i < (unsigned)VB_RN && x >= 0 && x < W && y >= 0 && y < H);
    }
    u64 Vh[VB_WORDS], Vv[VB_WORDS];
    {
        // compile-time bitmaps (forced: a constexpr call with a loop index is otherwise evaluated at run time)
        constexpr u64 KH[VB_WORDS] = {vb_word_h(0), vb_word_h(1), vb_word_h(2), vb_word_h(3), vb_word_h(4), vb_word_h(5)};
        constexpr u64 KV[VB_WORDS] = {vb_word_v(0), vb_word_v(1), vb_word_v(2), vb_word_v(3), vb_word_v(4), vb_word_v(5)};
        static_assert(VB_WORDS == 6, "tables above");
        u64 s1[VB_WORDS], s34[VB_WORDS];
        vb_shr<1>(Iw, s1);
        vb_shr<VB_RW>(Iw, s34);
#pragma unroll
        for (int k = 0; k < VB_WORDS; k++) {
            Vh[k] = Iw[k] & s1[k] & KH[k];
            Vv[k] = Iw[k] & s34[k] & KV[k];
        }
    }
    // ---- pairs with exactly one covered pixel.  Only those can change the result: with constant colour inside a
    //      link a blend between two covered pixels is alpha * (1 - 1) = 0 in value and in gradient.
    u64 Hw[2 * VB_WORDS];
    int nh = 0;
    {
        u64 s1[VB_WORDS], s34[VB_WORDS];
        vb_shr<1>(C, s1);
        vb_shr<VB_RW>(C, s34);
#pragma unroll
        for (int k = 0; k < VB_WORDS; k++) {
            Hw[k] = (C[k] ^ s1[k]) & Vh[k];
            Hw[VB_WORDS + k] = (C[k] ^ s34[k]) & Vv[k];
            nh += __popcll(Hw[k]) + __popcll(Hw[VB_WORDS + k]);
        }
    }
    VB_WAVE_SYNC();
    float val[4];
#pragma unroll
    for (int j = 0; j < 4; j++) val[j] = (KT(myq + j) != 0xffffffffu) ? 1.f : 0.f;
    if (dbg & 2) nh = 0;
    if (nh != 0) {
        // ---- dense hit list, ordered by (direction, region index)
        {
            int base = 0;
#pragma unroll
            for (int s = 0; s < 2 * VB_WORDS; s++) {
                const u64 w = Hw[s];
                if (w) {
                    if ((w >> lane) & 1)
                        hits[base + vb_mbcnt(w)] = (unsigned short)(((s % VB_WORDS) * 64 + lane) | ((s / VB_WORDS) << 15));
                    base += __popcll(w);
                }
            }
        }
        VB_WAVE_SYNC();
        // ---- silhouette analysis of the hits (restates nvdiffrast's antialias mesh kernel), 64 per round
        const VbVertsT<LAZY> pv = vb_verts<LAZY>(pvz);
        for (int hbase = 0; hbase < nh; hbase += 64) {
            const int h = hbase + lane;
            VbItem it;
            it.packed = 0;
            it.v1 = 0;
            it.v2 = 0;
            it.alpha = 0.f;
            bool keep = false;
            if (h < nh) {
                const int hq = hits[h];
                const int d = hq >> 15, q = hq & 0x7fff;
                const int qy = vb_div_rw(q), qx = q - qy * VB_RW;
                const int nq = q + (d ? VB_RW : 1);
                const unsigned k0 = KT(q), k1 = KT(nq);
                const bool chose0 = k0 != 0xffffffffu;  // exactly one of the two is covered
                const int t = min((int)(chose0 ? k0 : k1) & 0x7fffffff, T - 1);  // (always a triangle id: the pixel has an uncovered neighbour)
                int px = rx0 + qx, py = ry0 + qy;
                if (!chose0) {
                    px += 1 - d;
                    py += d;
                }
                float4 p[3], o[3];
                const int4 ti = tri4[t], oi = opp4[t];  // one aligned 16-byte gather each
                const int vi[3] = {ti.x, ti.y, ti.z}, ov[3] = {oi.x, oi.y, oi.z};
#pragma unroll
                for (int k = 0; k < 3; k++) p[k] = pv[vi[k]];
#pragma unroll
                for (int k = 0; k < 3; k++) o[k] = ((unsigned)ov[k] < (unsigned)V) ? pv[ov[k]] : p[k];
                const AAPair a = aa_analyze(p, o, px, py, d, chose0, W, H);
                if (a.found) {
                    pairA[d * VB_RN + q] = a.alpha;
                    // keep for the backward pass if the destination pixel is interior to this tile
                    const int oq = (a.alpha > 0.f) ? q : nq;
                    const int oy = vb_div_rw(oq), ox = oq - oy * VB_RW;
                    const bool oi = ox >= 1 && ox <= EHR_TILE_W && oy >= 1 && oy <= EHR_TILE_H;
                    if (oi && a.alpha != 0.f) {
                        it.packed = q | (d << 10) | (a.di << 11) | (a.tri1 << 13) | ((chose0 ? 0 : 1) << 14);
                        it.v1 = (a.di == 0) ? vi[1] : (a.di == 1 ? vi[2] : vi[0]);  // edge di: v1-v2, v2-v0, v0-v1
                        it.v2 = (a.di == 0) ? vi[2] : (a.di == 1 ? vi[0] : vi[1]);
                        it.alpha = a.alpha;
                        keep = want_grad != 0;
                    }
                }
            }
            const u64 km = __ballot(keep);
            if (km) {
                const int at = nitems + vb_mbcnt(km);
                const int nnew = nitems + __popcll(km);
                if (nnew > VB_JOB_ITEMS && spill_base < 0) {  // wave-uniform: first overflow of this job
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&meta[EHR_META_SPILL], VB_SPILL_BLOCK);
                    spill_base = __builtin_amdgcn_readfirstlane(base);
                }
                // items this job can keep: its slot, then its block of the spill pool as far as the pool reaches
                int room = VB_JOB_ITEMS;
                if (spill_base >= 0 && spill_base < spill_cap) room += min(VB_SPILL_BLOCK, spill_cap - spill_base);
                if (keep) {
                    if (at < VB_JOB_ITEMS)
                        vb_st_item<COH>(&jitems[slot * VB_JOB_ITEMS + at], it);
                    else if (at < room)
                        vb_st_item<COH>(&spill[spill_base + (at - VB_JOB_ITEMS)], it);
                    else if (COH)
                        atomicOr(&meta[EHR_META_OVERFLOW], 1);
                    else
                        meta[EHR_META_OVERFLOW] = 1;  // reported through loss = NaN, never silent
                }
                nitems = min(nnew, room);  // never more than were stored: the composite kernel reads exactly these
            }
        }
        VB_WAVE_SYNC();
        // ---- gather the antialiased value of this link at my pixels (fixed order: down, left, right, up pair)
        {
            float cn[6], cd[4], cu[4];
#pragma unroll
            for (int j = 0; j < 6; j++) cn[j] = (KT(myq - 1 + j) != 0xffffffffu) ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cd[j] = (KT(myq - VB_RW + j) != 0xffffffffu) ? 1.f : 0.f;
                cu[j] = (KT(myq + VB_RW + j) != 0xffffffffu) ? 1.f : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float c = cn[j + 1];
                float v = c;
                float a;
                a = pairA[VB_RN + myq + j - VB_RW];
                if (a < 0.f) v += a * (c - cd[j]);
                a = pairA[myq + j - 1];
                if (a < 0.f) v += a * (c - cn[j]);
                a = pairA[myq + j];
                if (a > 0.f) v += a * (cn[j + 2] - c);
                a = pairA[VB_RN + myq + j];
                if (a > 0.f) v += a * (cu[j] - c);
                val[j] = v;
            }
        }
    }
    // ---- publish: the link's value at the tile's pixels (tile-local row-major), the number of blended pairs
    const bool nz = __ballot(val[0] != 0.f || val[1] != 0.f || val[2] != 0.f || val[3] != 0.f) != 0;
    if (nz) vb_st_f4<COH>(jval + slot * 256 + r * EHR_TILE_W + c4, make_float4(val[0], val[1], val[2], val[3]));
    if (lane == 0) {
        vb_st_i32<COH>(&jn[slot], nz ? nitems : -1);
        if (nitems > VB_JOB_ITEMS) vb_st_i32<COH>(&jspill[slot], spill_base);
    }
#undef KT
}

// The job kernel's call: the job's coverage rows (cov) and depth/id buffer (key) are complete in LDS; the ids move to the
// (now idle) deferred list, the key buffer becomes the pair table, and the wave resolves the job it has just drawn.
// Nothing of the job's region goes through global memory, and the resolve stage needs no launch of its own: it was a
// 10 us kernel of one dependent chain per job behind a boundary; here the chain runs while other waves still rasterize.
// (Resolve and rasterizer never overlap inside a wave: the live ranges of the two are disjoint, unlike round 2's fusion.)
template <bool COH = false, bool LAZY = false>
__device__ __forceinline__ void vb_resolve_from_lds(const VbResolveArgs& Q, VbWaveLds& S, u64* key, const u64* cov, size_t slot,
                                                    int b, int l, int rx0, int ry0) {
    const int lane = lane_id();
    static_assert(VB_RN <= VB_DL && 2 * VB_RN * sizeof(unsigned short) <= sizeof(VbRaster), "ids live in the deferred list's storage, the hit list in the rounds' staging area");
    VB_WAVE_SYNC();
    u64 C[VB_WORDS];
    unsigned idw[VB_WORDS];
#pragma unroll
    for (int k = 0; k < VB_WORDS; k++) {
        const unsigned i = 64u * k + lane;
        const unsigned row = (unsigned)vb_div_rw((int)i), col = i - row * VB_RW;
        const bool in = i < (unsigned)VB_RN;
        const bool cv = in && ((cov[in ? row : 0] >> col) & 1ull);
        const unsigned id = in ? (unsigned)key[i] : 0xffffffffu;  // low word = triangle id; all-ones stays all-ones
        C[k] = __ballot(cv);
        // covered pixels whose triangle was never asked for (no uncovered neighbour) carry a marker instead of an id
        idw[k] = (id != 0xffffffffu) ? id : (cv ? VB_ID_COVERED : 0xffffffffu);
    }
    VB_WAVE_SYNC();  // every lane has read its keys: the buffer is free
    unsigned* const ids = S.dl;
#pragma unroll
    for (int k = 0; k < VB_WORDS; k++) {
        const unsigned i = 64u * k + lane;
        if (i < (unsigned)VB_RN) ids[i] = idw[k];
    }
    vb_resolve_job<COH, LAZY>(Q, ids, reinterpret_cast<float*>(key), reinterpret_cast<unsigned short*>(&S.R), C, slot, b, l, rx0, ry0);
    VB_WAVE_SYNC();
}

// a lane's four pixels of a mask tile row := 0 (the composite kernel's layout: lane -> row lane / 8, columns 4 (lane % 8)..+3)
__device__ __forceinline__ void vb_zero_tile_row(float* __restrict__ mask, size_t im, bool row_in, int ix, int W, int vec_ok) {
    if (!row_in) return;
    if (vec_ok) {
        if (ix < W) *reinterpret_cast<float4*>(mask + im) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (ix + j < W) mask[im + j] = 0.f;
    }
}

struct VbCompArgs {  // what the composite stage needs besides its LDS tables
    BinGeom g;
    int B;
    const float* mvp;    // [B][L][16] (clip-space vertices on demand: VbLazy)
    int V;
    const float* verts;
    const int* jn;
    const float* jval;
    const VbItem* jitems;
    const int* jspill;
    int jcap;
    const float* ref;
    float* mask;
    long long* facc;
    int nls, want_grad, vec_ok;
    const VbItem* spill;
    int spill_cap;
    int* meta;
    int dbg;
    const long long* tsum;
};

// The composite stage's work items of ONE wave (slot `wslot` of the `nslots` wave slots its XCD has; `nwg` workgroups in
// all): see vb_composite_kernel.  s_jbase / s_utile: the (view, link) tables in LDS; gpix: 256 floats of LDS of this wave.
// COH: the job slots are read at agent scope (the merged kernel: they were written earlier in the SAME launch, possibly
// through another XCD's L2).
template <bool COH, bool FILL>
__device__ __forceinline__ void vb_composite_items(const VbCompArgs& C, const int* s_jbase, const unsigned* s_utile, float* gpix,
                                                   int xcd, int wslot, int nslots, int nwg) {
    const BinGeom& g = C.g;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int W = g.W, H = g.H, L = g.L, B = C.B, U = B * L;
    const bool sparse = C.tsum != nullptr;
    const int jcap = C.jcap, dbg = C.dbg, nls = C.nls, want_grad = C.want_grad, vec_ok = C.vec_ok, V = C.V, spill_cap = C.spill_cap;
    const float* const mvpc = C.mvp;
    const float* const verts = C.verts;
    const int* const jn = C.jn;
    const float* const jval = C.jval;
    const VbItem* const jitems = C.jitems;
    const int* const jspill = C.jspill;
    const VbItem* const spill = C.spill;
    const float* const ref = C.ref;
    float* const mask = C.mask;
    long long* const facc = C.facc;
    int* const meta = C.meta;
    const long long* const tsum = C.tsum;
    // work items: every tile of every view, or (bound reference) the JOBS -- a job stands for its tile if no link before
    // its own has a job there, so that every tile with a job comes up exactly once and the others never
    const int nitems = sparse ? min(s_jbase[U], jcap) : B * g.nt;
    // XCD-aware order (locality only): every XCD takes a contiguous run of items
    const int per_xcd = (nitems + 7) >> 3;
    const int ibeg = xcd * per_xcd, iend = min(ibeg + per_xcd, nitems);
    const int istep = nslots;
    const int acc_stride = 12 * L + nls * VB_LOSS_STRIDE;
    const int r = lane >> 3, c4 = (lane & 7) * 4;
    const float invL = __builtin_amdgcn_rcpf((float)L);  // (u / L by vb_div_small)
    for (int item = ibeg + wslot; item < iend; item += istep) {
    int b, tx, ty;
    if (sparse) {
        int lo = 0;
        if (U <= 64) {  // the last (view, link) whose first job is <= item: one LDS read per lane and a ballot
            lo = __popcll(__ballot(lane < U && s_jbase[lane] <= item)) - 1;
        } else {
            int hi = U - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_jbase[mid] <= item)
                    lo = mid;
                else
                    hi = mid - 1;
            }
        }
        b = vb_div_small(lo, invL);
        const unsigned ut = s_utile[lo];
        const int nx = (int)(ut >> 22), k = item - s_jbase[lo];
        const int kr = vb_div_small(k, __builtin_amdgcn_rcpf((float)nx));  // (k < 2^16 tiles, nx <= 128: exact)
        ty = (int)((ut >> 10) & 4095u) + kr;
        tx = (int)(ut & 1023u) + k - kr * nx;
        // owner of the tile = the job of the first link that has one there
        bool prior = false;
        if (lane < lo - b * L) {
            const int u2 = b * L + lane;
            const unsigned u2t = s_utile[u2];
            const int n2 = s_jbase[u2 + 1] - s_jbase[u2];
            const int ax0 = u2t & 1023u, ay0 = (u2t >> 10) & 4095u, anx = u2t >> 22;
            prior = n2 > 0 && tx >= ax0 && tx < ax0 + anx && ty >= ay0 && (ty - ay0) * anx < n2;  // (n2 = anx x the rows)
        }
        if (__ballot(prior)) continue;
    } else {
        b = item / g.nt;
        const int tile = item - b * g.nt;
        tx = tile % g.ntx;
        ty = tile / g.ntx;
    }
    const int tile = ty * g.ntx + tx;
    long long* const vacc = facc + (size_t)b * acc_stride;
    long long* const lacc = vacc + 12 * L + (tile % nls) * VB_LOSS_STRIDE;
    const int rx0 = tx * EHR_TILE_W - 1, ry0 = ty * EHR_TILE_H - 1;
    // links whose tile range (the tiles + halo their screen box touches: the jobs stage 2 ran) contains this tile (lane l
    // tests link l, from the tables in LDS)
    unsigned tmask;
    int myn = -1, myslot = 0;
    {
        if (lane < L && !(dbg & 1)) {
            const int u = b * L + lane;
            const unsigned ut = s_utile[u];
            const int j0 = s_jbase[u], n = s_jbase[u + 1] - j0;
            const int tx0 = ut & 1023u, ty0 = (ut >> 10) & 4095u, nx = ut >> 22;
            if (n > 0 && tx >= tx0 && tx < tx0 + nx && ty >= ty0 && (ty - ty0) * nx < n) {
                myslot = j0 + (ty - ty0) * nx + (tx - tx0);
                if (myslot < jcap) myn = vb_ld_i32<COH>(&jn[myslot]);
            }
        }
        tmask = (unsigned)__ballot(myn >= 0);  // links that contribute a value here
    }
    if (sparse && tmask == 0) {  // nothing drawn here: the tile's cached sum(ref^2) is part of vtot already
        if (FILL) {
            const int zx = tx * EHR_TILE_W + c4, zy = ty * EHR_TILE_H + r;
            vb_zero_tile_row(mask, ((size_t)b * H + (H - 1 - (zy < H ? zy : 0))) * W + zx, zy < H, zx, W, vec_ok);
        }
        continue;
    }
    const int ix = tx * EHR_TILE_W + c4, iy = ty * EHR_TILE_H + r;  // first of my 4 pixels (GL rows: y up)
    const bool row_in = iy < H;
    const size_t im = ((size_t)b * H + (H - 1 - (row_in ? iy : 0))) * W + ix;  // image convention: row 0 = top
    float rf[4] = {0.f, 0.f, 0.f, 0.f};
    bool pin[4];
#pragma unroll
    for (int j = 0; j < 4; j++) pin[j] = row_in && (ix + j) < W;
    if (vec_ok) {
        if (pin[0]) {
            const float4 r4 = *reinterpret_cast<const float4*>(ref + im);
            rf[0] = r4.x; rf[1] = r4.y; rf[2] = r4.z; rf[3] = r4.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (pin[j]) rf[j] = ref[im + j];
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned todo = tmask;
    while (todo) {  // sum in link order
        const int l = __ffs(todo) - 1;
        todo &= todo - 1;
        const size_t slot = (size_t)vb_readlane(myslot, l);
        const float4 v4 = vb_ld_f4<COH>(jval + slot * 256 + r * EHR_TILE_W + c4);
        acc[0] += v4.x;
        acc[1] += v4.y;
        acc[2] += v4.z;
        acc[3] += v4.w;
    }
    // ---- composite, loss, mask write (image convention: row 0 = top)
    float e2 = 0.f, gv[4];
    float mv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        gv[j] = 0.f;
        mv[j] = 0.f;
        if (pin[j]) {
            const float m = acc[j] > 1.f ? 1.f : acc[j];
            const float e = m - rf[j];
            e2 += e * e;
            gv[j] = (acc[j] <= 1.f) ? 2.f * e : 0.f;
            mv[j] = m;
        }
    }
    if (mask) {
        if (vec_ok) {
            if (pin[0]) *reinterpret_cast<float4*>(mask + im) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (pin[j]) mask[im + j] = mv[j];
        }
    }
    {
        const float s = wave_sum(e2);
        if (lane == 0) {
            if (sparse)
                fix_add_delta(lacc, s, tsum[(size_t)b * g.nt + tile], meta);
            else
                fix_add(lacc, s, meta);
        }
    }
    const unsigned bmask = (unsigned)__ballot(myn > 0);  // links with blended pairs to back-propagate
    if (!want_grad || bmask == 0 || (dbg & 4)) continue;

    // ---- backward: blended pairs -> rows (x, y, w) of d loss / d MVP, per link
    VB_WAVE_SYNC();  // the previous tile's reads of gpix are complete
#pragma unroll
    for (int j = 0; j < 4; j++) gpix[r * EHR_TILE_W + c4 + j] = gv[j];
    VB_WAVE_SYNC();
    unsigned links = bmask;
    while (links) {
        const int l = __ffs(links) - 1;
        links &= links - 1;
        const size_t slot = (size_t)vb_readlane(myslot, l);
        int n = vb_readlane(myn, l);
        int sbase = 0;
        if (n > VB_JOB_ITEMS) {  // the rest of the list lives in the spill pool; never read outside it
            sbase = vb_ld_i32<COH>(&jspill[slot]);
            if (sbase < 0 || sbase >= spill_cap) n = VB_JOB_ITEMS;
            else n = min(n, VB_JOB_ITEMS + (spill_cap - sbase));
        }
        const VbVertsT<true> pv = vb_verts<true>(VbLazy{verts, mvpc + ((size_t)b * L + l) * 16, nullptr});  // (the link's matrix: clip space on demand)
        float G[12];
#pragma unroll
        for (int k = 0; k < 12; k++) G[k] = 0.f;
        for (int i = lane; i < n; i += 64) {
            const VbItem itm = vb_ld_item<COH>((i < VB_JOB_ITEMS) ? &jitems[slot * VB_JOB_ITEMS + i] : &spill[sbase + (i - VB_JOB_ITEMS)]);
            const int q = itm.packed & 1023, d = (itm.packed >> 10) & 1;
            const int tri1 = (itm.packed >> 13) & 1;
            const float dc = ((itm.packed >> 14) & 1) ? 1.f : -1.f;
            const int nq = q + (d ? VB_RW : 1);
            const int oq = (itm.alpha > 0.f) ? q : nq;
            const int oy = vb_div_rw(oq), ox = oq - oy * VB_RW;
            const float gi = gpix[(oy - 1) * EHR_TILE_W + (ox - 1)];
            const float dd = gi * dc;
            if (gi == 0.f || dd == 0.f) continue;
            const int qy = vb_div_rw(q), qx = q - qy * VB_RW;
            int px = rx0 + qx, py = ry0 + qy;
            if (tri1) {
                px += 1 - d;
                py += d;
            }
            float g1[3], g2[3];
            const float* a1 = verts + 3 * (size_t)itm.v1;
            const float* a2 = verts + 3 * (size_t)itm.v2;
            const float h1[4] = {a1[0], a1[1], a1[2], 1.f}, h2[4] = {a2[0], a2[1], a2[2], 1.f};
            // (the object-space vertices are needed for the matrix gradient anyway: their clip-space positions come from them)
            aa_pos_grad(transform_vertex(pv.M, h1[0], h1[1], h1[2]), transform_vertex(pv.M, h2[0], h2[1], h2[2]), px, py, d,
                        itm.alpha, dd, W, H, g1, g2);
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
#pragma unroll
                for (int c = 0; c < 4; c++) G[4 * rr + c] += g1[rr] * h1[c] + g2[rr] * h2[c];
        }
        // the twelve sums in 14 shuffles (bit-equal to twelve wave_sum calls); one lane per element adds its sum
        const float mine = wave_sum12(G, lane);
        if ((lane & 3) == 0 && (lane & 12) != 12) fix_add(&vacc[12 * l + wave_sum12_element(lane)], mine, meta);
    }
    }
    // ---- bound reference AND a mask output: the tiles that hold a job were written by their owners above; every other
    //      tile of the images is zero.  The waves share them out (most have no item, or one): a wave tests a tile against
    //      the links' tile ranges -- the predicate that decides whether the tile HAS an owner -- and stores 1 KB of zeros if
    //      it has none.  Stores only: nothing here is waited for before the ticket's vmcnt.
    if (FILL) {
        const int total = B * g.nt;
        for (int t = (int)blockIdx.x * 4 + wave; t < total; t += nwg * 4) {
            const int b = t / g.nt, tile = t - b * g.nt;
            const int ty = tile / g.ntx, tx = tile - ty * g.ntx;
            bool owned = false;
            if (lane < L) {
                const int u = b * L + lane;
                const unsigned ut = s_utile[u];
                const int n = s_jbase[u + 1] - s_jbase[u];
                const int tx0 = ut & 1023u, ty0 = (ut >> 10) & 4095u, nx = ut >> 22;
                owned = n > 0 && tx >= tx0 && tx < tx0 + nx && ty >= ty0 && (ty - ty0) * nx < n;
            }
            if (__ballot(owned)) continue;
            const int ix = tx * EHR_TILE_W + c4, iy = ty * EHR_TILE_H + r;
            const bool row_in = iy < H;
            vb_zero_tile_row(mask, ((size_t)b * H + (H - 1 - (row_in ? iy : 0))) * W + ix, row_in, ix, W, vec_ok);
        }
    }
}

#ifndef VB_JOB_WAVES
#define VB_JOB_WAVES 4
#endif
#ifndef VB_INLINE_RESOLVE
#define VB_INLINE_RESOLVE 1   // 0: the round-4 chain (jobs publish ids + coverage, vb_resolve_kernel resolves every slot)
#endif
// The job kernel's parameters, ONE struct in the kernarg segment.  VB_PARAM_BLOCK = 1: the kernel does not name its
// parameter; it reads the fields through the kernarg segment pointer, made opaque to the optimiser at every use
// (vb_job_params), so that a field is a scalar load where it is needed instead of one of ~60 scalar registers filled at
// kernel entry and kept -- i.e. spilled to vector-register lanes and reloaded -- across the whole job loop (round 4: 237
// spilled SGPRs in this kernel).  VB_PARAM_BLOCK = 0: ordinary by-value use of the same struct (the A/B reference).
#ifndef VB_PARAM_BLOCK
#define VB_PARAM_BLOCK 1
#endif
struct VbJobParams {
    BinGeom g;
    int B;
    VbClusters cl;
    VbRecs rc;
    const int* lbox;
    int* jn;
    unsigned* jid;
    int* jdesc;
    int* jbase;
    unsigned* jutile;
    int jcap;
    int* meta;
    int dbg;
    VbHeavy hv;
    long long* timeline;
    const float* verts;  // object-space vertices and the chunk's matrices: clip-space vertices are computed on demand (VbLazy) ...
    const float* mvp;
    const float4* posc;  // ... or kept per view ([B][V]) by the eager instantiations (NULL: lazy)
    int V;
    VbSlotIdx si;
    u64* jcov;
    int4* slow_list;
    int heavy_t, med_t0;
    VbResolveArgs rq;
    // MERGE (round 6): the composite + finish stages run at the end of this launch
    VbCompArgs ca;
    const long long* vtot;
    const int* ref_flag;
    float* loss;
    float* grad_mvp;
    int* lbox_all;
    StepTail tail;
};
typedef const VbJobParams __attribute__((address_space(4)))* VbJobParamsPtr;
__device__ __forceinline__ VbJobParamsPtr vb_job_params() {
    VbJobParamsPtr p = (VbJobParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));  // (what the optimiser cannot see through it cannot hoist to the kernel's entry)
    return p;
}
// a whole parameter sub-struct out of the kernarg segment (word by word: the segment's address space has no copy constructor)
template <class T>
__device__ __forceinline__ T vb_load_pod(const T __attribute__((address_space(4)))* p) {
    static_assert(sizeof(T) % 4 == 0, "vb_load_pod");
    T out;
    const int __attribute__((address_space(4)))* const src = (const int __attribute__((address_space(4)))*)p;
    int* const dst = reinterpret_cast<int*>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) dst[i] = src[i];
    return out;
}
template <class T>
__device__ __forceinline__ T vb_load_pod(const T* p) { return *p; }
template <class P>
__device__ __forceinline__ VbResolveArgs vb_load_rq(P p) {
    VbResolveArgs q;
    q.verts = p->rq.verts; q.mvp = p->rq.mvp; q.posc = p->rq.posc; q.L = p->rq.L; q.tri4 = p->rq.tri4; q.opp4 = p->rq.opp4; q.jn = p->rq.jn; q.jval = p->rq.jval;
    q.jitems = p->rq.jitems; q.jspill = p->rq.jspill; q.spill = p->rq.spill; q.meta = p->rq.meta; q.V = p->rq.V;
    q.T = p->rq.T; q.W = p->rq.W; q.H = p->rq.H; q.spill_cap = p->rq.spill_cap; q.want_grad = p->rq.want_grad;
    q.dbg = p->rq.dbg;
    return q;
}
// COVER (the scoring op): jcov = the (view, tile) coverage words [B][nt][4] the jobs OR their tile's interior into, jn = one
// sticky flag raised by a job that met a triangle whose depth class does not let coverage decide; no slots, no lists.
// MERGE (round 6, solver step with a bound reference mask and no mask output, one chunk of views, general-triangle pass off):
// the composite and finish stages run at the END OF THIS LAUNCH instead of in a launch of their own.  All workgroups of the
// grid are resident together (four per CU by registers and LDS, the grid is four per CU), so a grid-wide arrival counter is
// safe: a workgroup whose waves have run out of jobs waits until its slots' stores have been performed, arrives (a ticket per
// XCD, then one over the XCDs), sleeps on its XCD's go flag, and then takes composite items like a workgroup of
// vb_composite_kernel would -- tables already in LDS, no launch boundary (cache write-back + invalidate, dispatch, tables:
// 13 of that launch's 18 us were not tile work).  Job slots travel at agent scope (vb_st_* / vb_ld_*<true>).  A wait that
// does not end (a grid that is not resident after all) is REPORTED through the overflow flag after ~50 ms, never a hang.
template <bool COVER, bool MERGE = false, bool LAZY = false>
__global__ void __launch_bounds__(256, VB_JOB_WAVES)
vb_job_kernel(VbJobParams prm_) {
    __shared__ VbWaveLds lds_all[4];
    // PRM(field): a kernel parameter, read where it is used (see VbJobParams)
#if VB_PARAM_BLOCK
#define PRM(f) (vb_job_params()->f)
#define VB_RQ() vb_load_rq(vb_job_params())
#else
#define PRM(f) (prm_.f)
#define VB_RQ() vb_load_rq(&prm_)
#endif
#ifdef VB_MERGE_TL
    const long long tl_start0 = wall_clock64();
#endif
    const int W = PRM(g.W), H = PRM(g.H), L = PRM(g.L), gnt = PRM(g.nt), gntx = PRM(g.ntx);
    const int B = PRM(B), V = PRM(V), dbg = PRM(dbg);
    int heavy_t = max(PRM(heavy_t), PRM(hv.gen)[6]);  // (the value in force: the vertex kernel adapts it, see VbHeavy)
    const float* const pverts = PRM(verts);
    const float* const pmvp = PRM(mvp);
    const float4* const posc = LAZY ? nullptr : PRM(posc);
#ifdef VB_TIMELINE  // profiling build only (-DVB_TIMELINE): a record per wave, printed by vbuf_meta_read under EHR_VB_PRINT
    const long long tl_start = wall_clock64();
#endif
    // the scheduling hint of the previous step is requested before anything else, in ONE round trip: which of the two
    // lists is the one to consume follows from the generation, so both lists' counts and this workgroup's entry of either are
    // requested together with it (they were a second, dependent round trip on every wave's way to its first job)
    const int* const hgen = PRM(hv.gen);
    const int gen = hgen[0], hc1 = hgen[1], hc2 = hgen[2], hm4 = hgen[4], hm5 = hgen[5];
    const int hida = PRM(hv.list)[min((int)blockIdx.x, VB_HEAVY_CAP - 1)];
    const int hidb = PRM(hv.list)[VB_HEAVY_CAP + min((int)blockIdx.x, VB_HEAVY_CAP - 1)];
    // (Not the long job a wave starts with: its index depends on the number of heavy workgroups.  Dealing the long jobs to
    //  the XCD's LAST workgroups instead, counted from the end, makes the index known here -- measured: 84.7 vs 83.1 us, the
    //  last workgroups of a 1024-workgroup grid start last.)
    const int hcur = (gen - 1) & 1, hnxt = gen & 1;
    const int nheavy_prev = hcur ? hc2 : hc1;
    const int hid_first = hcur ? hidb : hida;
    __shared__ int upre[VB_MAX_UNITS + 1];   // first job of every (view, link)
    __shared__ unsigned utile[VB_MAX_UNITS];  // its tile range: tx0 | ty0 << 10 | nx << 22
    __shared__ int lcoff[33];                 // first cluster of every link
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    VbWaveLds& S = lds_all[wave];
    const int U = B * L;
    const float invL = __builtin_amdgcn_rcpf((float)L);  // (u / L by vb_div_small)
    // ---- prologue (every workgroup, redundantly): tile range and job count of every (view, link), prefix sum
    int total;
    if (U <= 64) {
        // At most one (view, link) per lane: EVERY WAVE builds the tables by itself -- its own read of the link boxes, a wave
        // scan, the same values to the same LDS words as its three siblings -- and goes on without a barrier: the two
        // workgroup barriers and the LDS hand-over of the cooperative form below stood between every wave and its first job
        // (3.0-3.4 us into the kernel; the long jobs the kernel ends on are first jobs).
        if (lane <= L) lcoff[lane] = PRM(cl.coff)[lane];
        int cnt = 0;
        if (lane < U) {
            int tx0 = 0, ty0 = 0, nx = 0, ny = 0;
            const bool ne = vb_unit_tiles(PRM(lbox) + VB_LBOX_STRIDE * (size_t)lane, W, H, tx0, ty0, nx, ny, COVER ? 0 : 1);
            cnt = ne ? nx * ny : 0;
            utile[lane] = (unsigned)tx0 | ((unsigned)ty0 << 10) | ((unsigned)(ne ? nx : 1) << 22);
        }
        const int incl = vb_scan_add(cnt);
        if (lane < U) upre[lane + 1] = incl;
        if (lane == 0) upre[0] = 0;
        total = vb_readlane(incl, 63);
        VB_WAVE_SYNC();
    } else {
    if (tid <= L) lcoff[tid] = PRM(cl.coff)[tid];
    for (int u = tid; u < U; u += 256) {
        int tx0 = 0, ty0 = 0, nx = 0, ny = 0;
        const bool ne = vb_unit_tiles(PRM(lbox) + VB_LBOX_STRIDE * (size_t)u, W, H, tx0, ty0, nx, ny, COVER ? 0 : 1);
        upre[u + 1] = ne ? nx * ny : 0;
        utile[u] = (unsigned)tx0 | ((unsigned)ty0 << 10) | ((unsigned)(ne ? nx : 1) << 22);
    }
    if (tid == 0) upre[0] = 0;
    __syncthreads();
    if (wave == 0) {
        int run = 0;
        for (int base = 0; base < U; base += 64) {
            const int i = base + lane;
            int incl = (i < U) ? upre[i + 1] : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            if (i < U) upre[i + 1] = run + incl;
            run += vb_readlane(incl, 63);
        }
    }
    __syncthreads();
    total = upre[U];
    }
    if (blockIdx.x == 0) {
        // job slots are numbered like the jobs: stage 3 finds a (view, link, tile) slot from the link's first job
        // (every entry this wave may read here it has written itself, or a barrier lies in between: see above)
        if (!COVER) {
            for (int u = tid; u <= U; u += 256) PRM(jbase)[u] = upre[u];
            for (int u = tid; u < U; u += 256) PRM(jutile)[u] = utile[u];
        }
        if (tid == 0) {
            PRM(meta)[5] = total;  // number of jobs (the resolve kernel's loop bound)
            if (!COVER && total > PRM(jcap)) PRM(meta)[EHR_META_OVERFLOW] = 1;  // cannot happen with one slot per (view, link, tile)
        }
    }
    if (!COVER) total = min(total, PRM(jcap));
    // XCD-aware order: workgroup w runs on XCD w % 8 (observed, used for L2 locality only): every XCD takes a contiguous
    // eighth of the job list, so that a view's vertices, boxes and records stay in one L2.  Inside that eighth every wave
    // takes one job statically; the jobs beyond that are claimed (one returning atomic on the XCD's cursor) by whichever
    // wave finishes first.  Jobs differ by two orders of magnitude in cost: dealt statically (EHR_VB_DEBUG & 8) the kernel
    // waits for a wave that got a heavy SECOND job; claimed from the start, 500 waves hit each cursor at once (~12 ns per
    // same-address atomic) and the kernel is 20 % slower.
    const int per_xcd = (total + 7) >> 3, xcd = blockIdx.x & 7;
    const int jbeg = xcd * per_xcd, jend = min(jbeg + per_xcd, total);
    int* const cursor = vb_line(PRM(meta), xcd);
    VbJobArgs A;
    A.rc.tbox = PRM(rc.tbox);
    A.rc.cbox = PRM(rc.cbox);
    A.rc.trec = PRM(rc.trec);
    A.rc.n = PRM(rc.n);
    A.verts = pverts;
    A.mvp = pmvp;
    A.posc = posc;
    A.cvidx = PRM(si.cvidx);
    A.lcoff = lcoff;
    A.jid = PRM(jid);
    A.jcov = PRM(jcov);
    A.jdesc = PRM(jdesc);
    A.jn = PRM(jn);
    A.NC = PRM(cl.NC);
    A.V = V;
    A.W = W;
    A.H = H;
    A.L = L;
    // ---- heavy jobs first, one workgroup each: the four waves share the job's depth/id buffer (wave 0's) and split the
    //      candidate clusters; wave 0 publishes.  A job alone costs up to ~80 us on one wave (a thousand candidate
    //      triangles in one tile), which used to be the duration of this kernel at small batch sizes.
    __shared__ int s_heavy[2];  // drawn flag, survivors
    // Only when the machine is short of jobs (at most ~2 per wave): with many views per GPU the kernel is bound by the
    // sum of the jobs, not by the longest, and four waves on one job are less efficient than four jobs (measured: 64
    // views 8 % slower with the heavy phase, 8 views 10 % faster, 1 view 40 % faster).
    // Likewise when heavy jobs are the rule rather than the exception (more than one per two workgroups: the Franka
    // meshes at 1080p have ~3000 of them in 8100 jobs and run 17 % slower with the heavy phase; the 8-view xArm7
    // workload has ~220 in 5000).
    const int hmax = (dbg >> 8) ? (dbg >> 8) : (int)gridDim.x / 2;  // (EHR_VB_DEBUG bits 8..: experiment with the limit)
    const int nheavy = ((dbg & 64) || total > 2 * 4 * (int)gridDim.x || nheavy_prev > hmax) ? 0 : min(nheavy_prev, min(VB_HEAVY_CAP, (int)gridDim.x));  // (<= one heavy job per workgroup)
    // ... and when there are fewer jobs than waves (one view, small images) most workgroups are idle anyway: jobs count as
    // heavy from a proportionally lower cost (down to an eighth: a few rounds), so that the longest ones are shared
    {
        const float f = fminf(1.f, fmaxf(0.125f, (float)total / (float)(2 * (int)gridDim.x)));
        heavy_t = (int)((float)heavy_t * f);
    }
    auto remember_heavy = [&](int id) {
        const int at = atomicAdd(&PRM(hv.gen)[1 + hnxt], 1);
        if (at < VB_HEAVY_CAP) PRM(hv.list)[hnxt * VB_HEAVY_CAP + at] = id;
    };
    // Long jobs below the heavy threshold: a wave that claims one late (after two or three others) is what the kernel
    // ends on -- 30 us on one wave whenever it starts -- so they are remembered as well and dealt out as static FIRST jobs,
    // one per wave, in the next step.  Only when the machine is short of jobs (at most 1.5 per wave): with more, claiming
    // balances the waves anyway and the jobs are better off in their own XCD's eighth of the list (L2).
    const int med_t = max(PRM(med_t0), 1);
    const int nmed = ((dbg & (64 | 128)) || total > 6 * (int)gridDim.x) ? 0 : min(hcur ? hm5 : hm4, PRM(hv.mcap));
    auto remember_long = [&](int id) {
        const int at = atomicAdd(&PRM(hv.gen)[4 + hnxt], 1);
        if (at < VB_MED_CAP) PRM(hv.mlist)[hnxt * VB_MED_CAP + at] = id;
    };
    if ((int)blockIdx.x < nheavy) {  // (workgroup-uniform: the others go straight on, without a barrier)
        if (tid < 2) s_heavy[tid] = 0;
        __syncthreads();
    }
#if VB_PRIO_HEAVY
    if ((int)blockIdx.x < nheavy) __builtin_amdgcn_s_setprio(VB_PRIO_HEAVY);
#endif
    int hres_job = -1, hres_b = 0, hres_l = 0, hres_rx0 = 0, hres_ry0 = 0;  // the heavy job wave 0 resolves once the workgroup has split up again
    for (int hj = blockIdx.x; hj < nheavy; hj += gridDim.x) {  // workgroup-uniform (at most one turn: nheavy <= gridDim.x)
        const int id = (hj == (int)blockIdx.x) ? hid_first : PRM(hv.list)[hcur * VB_HEAVY_CAP + hj];
        const int u = min((int)((unsigned)id >> 22), U - 1), tx = id & 1023, ty = (id >> 10) & 4095;
        const unsigned ut = utile[u];
        const int tx0 = ut & 1023u, ty0 = (ut >> 10) & 4095u, nx = ut >> 22, n = upre[u + 1] - upre[u];
        const int job = upre[u] + (ty - ty0) * nx + (tx - tx0);
        // (n is nx times the rows: ty inside the rows <=> the tile's job number inside the link's jobs)
        if (n <= 0 || tx < tx0 || tx >= tx0 + nx || ty < ty0 || (ty - ty0) * nx >= n || job >= total) continue;  // the link moved away
        const int b = vb_div_small(u, invL), l = u - b * L;
        const int rx0 = tx * EHR_TILE_W - 1, ry0 = ty * EHR_TILE_H - 1;
        VbRegion rg;
        rg.x0 = max(rx0, 0);
        rg.y0 = max(ry0, 0);
        rg.x1 = min(rx0 + VB_RW - 1, W - 1);
        rg.y1 = min(ry0 + VB_RH - 1, H - 1);
        VbWaveLds& S0 = lds_all[0];
        for (int i = tid; i < VB_RN; i += 256) S0.key[i] = VB_EMPTY;
        if (tid < VB_RH) S0.cov[tid] = 0ull;
        __syncthreads();
        int nsurv = 0, dln = 0;
        const int drawn = vb_job_raster<false, false, LAZY>(A, S, S0.key, S0.cov, b, l, rg, rx0, ry0, wave, 4, nsurv, dln);
        if (lane == 0) {
            if (drawn > 0) atomicOr(&s_heavy[0], 1);
            if (drawn < 0) atomicOr(&s_heavy[0], 2);  // a triangle for the general path: wave 0 redoes the job alone
            atomicAdd(&s_heavy[1], nsurv);
        }
        __syncthreads();  // the job's coverage is complete: every wave filters its own deferred units against it
        const int any_drawn = s_heavy[0];
        int tot_surv = s_heavy[1];
        if (!(any_drawn & 2) && dln > 0)
            vb_flush<false, LAZY>(S, S0.key, S0.cov, dln, VbLazy{pverts, pmvp + ((size_t)b * L + l) * 16, LAZY ? nullptr : posc + (size_t)b * V}, PRM(si.cvidx) + (size_t)lcoff[l] * 64, W, H, rx0, ry0);
        __syncthreads();
#if VB_INLINE_RESOLVE
        if (!COVER && any_drawn == 1) {
            hres_job = job;
            hres_b = b;
            hres_l = l;
            hres_rx0 = rx0;
            hres_ry0 = ry0;
        }
#else
        if (any_drawn == 1) vb_publish(A, S0.key, S0.cov, job, u, tx, ty, wave, 4);  // every wave its share of the words
#endif
        if (wave == 0) {
            if (any_drawn & 2) {  // put aside for vb_slow_kernel
                if (lane == 0) vb_put_aside<MERGE>(PRM(slow_list), PRM(meta), PRM(jn), PRM(jdesc), job, u, tx, ty);
            } else if (any_drawn) {
            } else if (lane == 0) {
                vb_st_i32<MERGE>(&PRM(jn)[job], -1);
                PRM(jdesc)[job] = -1;
            }
            if (lane == 0) {
                s_heavy[0] = 0;
                s_heavy[1] = 0;
                if (tot_surv >= heavy_t)
                    remember_heavy(id);
                else if (tot_surv >= med_t)
                    remember_long(id);
#ifdef VB_TIMELINE
                S0.tl_flushes = tot_surv;  // (profiling build: wave 0's counters are reset after the heavy phase; parked here)
#endif
            }
        }
        __syncthreads();
    }

#if VB_INLINE_RESOLVE
    // (the loop's last barrier is behind us: waves 1-3 go on to their own jobs, nobody touches wave 0's LDS but wave 0)
    if (!COVER && wave == 0 && hres_job >= 0) vb_resolve_from_lds<MERGE, LAZY>(VB_RQ(), S, S.key, S.cov, (size_t)hres_job, hres_b, hres_l, hres_rx0, hres_ry0);
#endif
#if VB_PRIO_HEAVY
    __builtin_amdgcn_s_setprio(0);
#endif
    // workgroups that just spent their time on a heavy job take no static job: the first hk of this XCD's workgroups
    const int nhw = min(nheavy, (int)gridDim.x);
    const int hk = (nhw > xcd) ? (nhw - xcd + 7) >> 3 : 0;
    const int kx = blockIdx.x >> 3;                      // this workgroup's index inside its XCD
    const int G8 = (int)gridDim.x >> 3;                  // workgroups per XCD
#ifdef VB_TIMELINE
    const long long tl_heavy = wall_clock64();
    int tl_jobs = 0, tl_maxsurv = 0, tl_sumsurv = 0;
    long long tl_last[4] = {0, 0, 0, 0}, tl_dry = 0;  // the wave's LAST job: start, rounds done, depth tests done, resolved
#endif
#ifdef VB_TIMELINE
    const int tl_hcost = (wave == 0) ? lds_all[0].tl_flushes : 0;
    VB_WAVE_SYNC();
    if (lane == 0) {
        S.tl_units = S.tl_rounds = S.tl_flushes = S.tl_tested = S.tl_deferred = S.tl_cands = S.tl_groups = 0;
        for (int k = 0; k < 8; k++) S.tl_c[k] = 0;
    }
#endif
    // static first jobs: wave rx of this XCD's static waves (global rank 8 rx + xcd) takes long job number <rank> of the
    // previous step if there is one, else the (rx - nmx)-th job of the XCD's eighth; the rest of the eighth is claimed
    const int nmx = (nmed > xcd) ? (nmed - xcd + 7) >> 3 : 0;  // long jobs that go to this XCD's waves (<= the static waves, see mcap)
    // (the long jobs one per WORKGROUP first -- rank wave * (G8 - hk) + (kx - hk) -- instead of four to a workgroup: measured in
    //  round 6, no difference: 41.1 / 33.9 / 90.4 us job stage either way at 8 views / 1 view / Franka)
    const int rx = (kx - hk) * 4 + wave;
    bool list_first = kx >= hk && rx < nmx;
    bool first_job = kx >= hk && !list_first;
    const int nsn = (G8 - hk) * 4 - nmx;                       // static jobs of the eighth itself
    int sjob = jbeg + rx - nmx;
    for (;;) {
#ifdef VB_TIMELINE
        const long long tl_j0 = __builtin_readcyclecounter();
#endif
        int job = 0, u = -1, tx = 0, ty = 0;
        if (list_first) {
            list_first = false;
#if VB_PRIO_LONG
            __builtin_amdgcn_s_setprio(VB_PRIO_LONG);
#endif
            const int id = PRM(hv.mlist)[hcur * VB_MED_CAP + 8 * rx + xcd];
            u = min((int)((unsigned)id >> 22), U - 1);
            tx = id & 1023;
            ty = (id >> 10) & 4095;
            const unsigned ut = utile[u];
            const int tx0 = ut & 1023u, ty0 = (ut >> 10) & 4095u, nx = ut >> 22, n = upre[u + 1] - upre[u];
            job = upre[u] + (ty - ty0) * nx + (tx - tx0);
            if (n <= 0 || tx < tx0 || tx >= tx0 + nx || ty < ty0 || (ty - ty0) * nx >= n || job >= total) continue;  // the link moved away
        } else {
#if VB_PRIO_LONG
            __builtin_amdgcn_s_setprio(0);
#endif
            if (first_job || (dbg & 8)) {
                job = sjob;
                sjob += max(nsn, 1);
                first_job = false;
                if (job >= jbeg + nsn && !(dbg & 8)) job = jend;  // (more static waves than jobs)
            } else {  // whoever is done first takes the next one: the waves stuck with a heavy first job take no second
                if (lane == 0) job = jbeg + nsn + atomicAdd(cursor, 1);
                job = __builtin_amdgcn_readfirstlane(job);
            }
            if (job >= jend) {
#ifdef VB_TIMELINE
                tl_dry = wall_clock64();  // this wave found its XCD's list of jobs empty
#endif
                break;
            }
            if (U <= 64) {  // the last (view, link) whose first job is <= job: one LDS read per lane and a ballot
                u = __popcll(__ballot(lane < U && upre[lane] <= job)) - 1;
            } else {
                int lo = 0, hi = U - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (upre[mid] <= job)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                u = lo;
            }
            {
                const unsigned ut = utile[u];
                const int nx = (int)(ut >> 22), k = job - upre[u];
                const int kr = vb_div_small(k, __builtin_amdgcn_rcpf((float)nx));  // (k < 2^16 tiles, nx <= 128: exact)
                ty = (int)((ut >> 10) & 4095u) + kr;
                tx = (int)(ut & 1023u) + k - kr * nx;
            }
            const int st = (nheavy > 0 || nmed > 0) ? PRM(hv.stamp)[u * gnt + ty * gntx + tx] : 0;
            // a workgroup took this one in the heavy phase / it is some wave's first job
            if ((nheavy > 0 && st == gen) || (nmed > 0 && st == -gen)) continue;
        }
        const int b = vb_div_small(u, invL), l = u - b * L;
        const size_t slot = (size_t)job;
        const int hint_id = vb_hint_pack(u, tx, ty);
        const int rx0 = tx * EHR_TILE_W - 1, ry0 = ty * EHR_TILE_H - 1;
        VbRegion rg;  // tile + 1-pixel halo, inside the image (coverage-only form: the tile alone, same origin)
        rg.x0 = COVER ? rx0 + 1 : max(rx0, 0);
        rg.y0 = COVER ? ry0 + 1 : max(ry0, 0);
        rg.x1 = min(rx0 + VB_RW - 1 - (COVER ? 1 : 0), W - 1);
        rg.y1 = min(ry0 + VB_RH - 1 - (COVER ? 1 : 0), H - 1);
        VB_WAVE_SYNC();
        if (!COVER) {
#pragma unroll
            for (int k = 0; k < VB_WORDS; k++) {
                const unsigned i = 64u * k + lane;
                if (i < (unsigned)VB_RN) S.key[i] = VB_EMPTY;
            }
        }
        if (lane < VB_RH) S.cov[lane] = 0ull;
        if (COVER && lane == 0) S.bad = 0;
        VB_WAVE_SYNC();
#ifdef VB_TIMELINE
        const long long tl_j1 = __builtin_readcyclecounter();
        tl_last[0] = wall_clock64();
#endif
        int nsurv = 0, dln = 0;
        const int drawn = vb_job_raster<false, COVER, LAZY>(A, S, S.key, S.cov, b, l, rg, rx0, ry0, 0, 1, nsurv, dln);
#ifdef VB_TIMELINE
        tl_last[1] = tl_last[2] = tl_last[3] = wall_clock64();  // culling + rasterizer rounds done
#endif
        if (COVER) {
            // flagged units (their depth range must be tested per pixel: edge-on slivers mostly) are the only deferred ones
            if (drawn >= 0 && dln > 0)
                vb_flush<true, LAZY>(S, S.key, S.cov, dln, VbLazy{pverts, pmvp + ((size_t)b * L + l) * 16, LAZY ? nullptr : posc + (size_t)b * V}, PRM(si.cvidx) + (size_t)lcoff[l] * 64, W, H, rx0, ry0);
            VB_WAVE_SYNC();
            if (drawn < 0 || S.bad) {  // a triangle for the general path, or a drawn pixel with a depth <= 0: coverage cannot
                if (lane == 0) PRM(jn)[0] = 1;  // decide here and the caller falls back for the whole call
                continue;
            }
            if (drawn > 0) {
                // the tile's 32 x 8 interior of the coverage rows, OR-ed into the (view, tile) words (all links of a view)
                VB_WAVE_SYNC();
                if (lane < 4) {
                    const u64 w = ((S.cov[2 * lane + 1] >> 1) & 0xffffffffull) | (((S.cov[2 * lane + 2] >> 1) & 0xffffffffull) << 32);
                    // (layout [candidate][tile][pose][4]: the count kernel reads a candidate's words of a tile in one piece;
                    //  PRM(jcap) carries S, the poses per candidate, in this form)
                    if (w) atomicOr((unsigned long long*)&PRM(jcov)[((((size_t)(b / PRM(jcap)) * gnt + (size_t)ty * gntx + tx) * PRM(jcap) + (b % PRM(jcap))) * 4 + lane)], w);
                }
            }
            continue;
        }
        if (drawn < 0) {  // a triangle for the general path (near-plane clipping, huge extent): put the job aside
            if (lane == 0) vb_put_aside<MERGE>(PRM(slow_list), PRM(meta), PRM(jn), PRM(jdesc), job, u, tx, ty);
            continue;
        }
        if (dln > 0) vb_flush<false, LAZY>(S, S.key, S.cov, dln, VbLazy{pverts, pmvp + ((size_t)b * L + l) * 16, LAZY ? nullptr : posc + (size_t)b * V}, PRM(si.cvidx) + (size_t)lcoff[l] * 64, W, H, rx0, ry0);
#ifdef VB_TIMELINE
        tl_last[2] = tl_last[3] = wall_clock64();  // depth tests done
        tl_jobs++;
        tl_maxsurv = max(tl_maxsurv, nsurv);
        tl_sumsurv += nsurv;
#endif
        if (lane == 0) {
            if (nsurv >= heavy_t)
                remember_heavy(hint_id);
            else if (nsurv >= med_t)
                remember_long(hint_id);
        }
        if (drawn == 0) {  // the link's box touches this tile, its triangles do not
            if (lane == 0) {
                vb_st_i32<MERGE>(&PRM(jn)[slot], -1);
                PRM(jdesc)[slot] = -1;
            }
            continue;
        }
#ifdef VB_TIMELINE
        const long long tl_j2 = __builtin_readcyclecounter();
#endif
#if VB_INLINE_RESOLVE
        vb_resolve_from_lds<MERGE, LAZY>(VB_RQ(), S, S.key, S.cov, slot, b, l, rx0, ry0);
#else
        vb_publish(A, S.key, S.cov, job, u, tx, ty);
#endif
#ifdef VB_TIMELINE
        tl_last[3] = wall_clock64();  // resolved
        if (lane == 0) {
            const long long now = __builtin_readcyclecounter();
            S.tl_c[4] += now - tl_j0;
            S.tl_c[5] += tl_j1 - tl_j0;
            S.tl_c[6] += now - tl_j2;
        }
#endif
    }
#ifdef VB_TIMELINE
    if (lane == 0 && PRM(timeline)) {
        const size_t gw = (size_t)blockIdx.x * 4 + wave;
        const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_ID: wave slot, SIMD, CU, SH, SE
        const unsigned xccid = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // XCC_ID
        PRM(timeline)[4 * gw] = tl_start;
        PRM(timeline)[4 * gw + 1] = wall_clock64();
        PRM(timeline)[4 * gw + 2] = tl_heavy;
        PRM(timeline)[4 * gw + 3] = (long long)(tl_jobs & 0xff) | ((long long)(tl_maxsurv & 0xfff) << 8) | ((long long)(tl_sumsurv & 0xfff) << 20) |
                               ((long long)(hwid & 0xffff) << 32) | ((long long)(xccid & 0xf) << 48);
        long long* const tx = PRM(timeline) + 4 * (size_t)gridDim.x * 4 + 12 * gw;
        tx[0] = S.tl_units;
        tx[1] = S.tl_rounds;
        tx[2] = (long long)S.tl_flushes | ((long long)S.tl_tested << 16) | ((long long)S.tl_deferred << 40);
        for (int k = 0; k < 4; k++) tx[3 + k] = S.tl_c[k];
        tx[7] = tl_hcost;
        for (int k = 0; k < 3; k++) tx[8 + k] = S.tl_c[4 + k];
        tx[11] = (S.tl_c[7] & 0xffffffffll) | ((long long)S.tl_cands << 32) | ((long long)S.tl_groups << 48);
        long long* const ty = PRM(timeline) + 16 * (size_t)gridDim.x * 4 + 5 * gw;
        for (int k = 0; k < 4; k++) ty[k] = tl_last[k];
        ty[4] = tl_dry;
    }
#endif
    if (MERGE && !COVER) {
        // ---- the grid-wide hand-over.  Every wave first waits until its own slot stores have been performed (agent scope:
        //      written through), the workgroup arrives, then sleeps on its XCD's go flag (= this step's generation).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* const meta = PRM(meta);
#ifdef VB_MERGE_TL  // profiling build: when this workgroup arrived, was let go, finished its items, [finished the finish stage]
        long long* const mtl = PRM(timeline) + 8 * (size_t)blockIdx.x;
        if (tid == 0) {
            mtl[0] = tl_start0;
            mtl[1] = wall_clock64();
        }
#endif
        if (tid == 0) {
            if (atomicAdd(vb_line(meta, 18 + xcd), 1) == G8 - 1 && atomicAdd(vb_line(meta, 26), 1) == 7)
                for (int k = 0; k < 8; k++) __hip_atomic_store(vb_line(meta, 27 + k), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(vb_line(meta, 27 + xcd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 16)) {  // (~50-100 ms: the grid is not resident together after all -- reported, never a hang)
                    atomicOr(&meta[EHR_META_OVERFLOW], 8);
                    break;
                }
            }
        }
        __syncthreads();
#ifdef VB_MERGE_TL
        if (tid == 0) mtl[2] = wall_clock64();
#endif
        // ---- composite stage (vb_composite_kernel's items, this XCD's eighth of them dealt over its waves)
        const VbCompArgs C = vb_load_pod(&PRM(ca));
        {   // the links' screen boxes start "empty" in the next step (nobody reads them after the prologue above)
            int* const lbox_all = PRM(lbox_all);
            for (int i = (int)blockIdx.x * 256 + tid; i < 16 * B * L; i += (int)gridDim.x * 256) lbox_all[i] = (i & 2) ? INT_MIN : INT_MAX;
        }
        // (items beyond a wave's first one claimed from a per-XCD cursor instead of dealt: measured, 72.2 -> 75.3 us at 8 views and
        //  278 -> 326 us at 64 -- every wave ends on a claim, 512 same-address atomics per XCD at ~12 ns each)
        vb_composite_items<true, false>(C, upre, utile, reinterpret_cast<float*>(S.key), xcd, kx * 4 + wave, G8 * 4, (int)gridDim.x);
        // ---- the workgroup whose atomics are performed last runs the finish stage (as in vb_composite_kernel)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef VB_MERGE_TL
        if (tid == 0) {
            mtl[3] = wall_clock64();
            mtl[4] = 0;
        }
#endif
        if (tid == 0) {
            int last = 0;
            if (atomicAdd(vb_line(meta, 8 + xcd), 1) == G8 - 1) last = atomicAdd(vb_line(meta, 16), 1) == 7;
            s_heavy[0] = last;  // (the heavy phase's flag word: idle since that phase)
        }
        __syncthreads();
        if (!s_heavy[0]) return;
        const int* const ref_flag = PRM(ref_flag);
        if (ref_flag && tid == 0 && ref_flag[0]) atomicOr(&meta[EHR_META_OVERFLOW], 1);  // the bound reference's own sums overflowed
        __syncthreads();
        const StepTail tail = vb_load_pod(&PRM(tail));
        finish_body<true>(C.g, B, C.facc, PRM(vtot), PRM(loss), PRM(grad_mvp), meta, tail, C.nls, nullptr, VB_LOSS_STRIDE,
                          reinterpret_cast<float*>(lds_all[0].key), reinterpret_cast<double(*)[17]>(lds_all[1].key),
                          reinterpret_cast<float*>(lds_all[2].key), reinterpret_cast<float(*)[16]>(lds_all[3].key));
#ifdef VB_MERGE_TL
        __syncthreads();
        if (tid == 0) mtl[4] = wall_clock64();
#endif
    }
}
#undef PRM
#undef VB_RQ

// Stage 2a (normally empty): the jobs the lean code put aside, one wave each, with the general triangle path.
__global__ void __launch_bounds__(256)
vb_slow_kernel(BinGeom g, VbClusters cl, VbRecs rc, const float* __restrict__ verts, const float* __restrict__ mvp,
               const float4* __restrict__ posc, int V, VbSlotIdx si,
               unsigned* __restrict__ jid, u64* __restrict__ jcov, int* __restrict__ jdesc, int* __restrict__ jn,
               const int4* __restrict__ slow_list, const int* __restrict__ meta) {
    const int n = *vb_line(const_cast<int*>(meta), 17);
    if (n == 0) return;
    __shared__ VbWaveLds lds_all[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    VbJobArgs A;
    A.rc = rc;
    A.verts = verts;
    A.mvp = mvp;
    A.posc = posc;
    A.cvidx = si.cvidx;
    A.lcoff = cl.coff;
    A.jid = jid;
    A.jcov = jcov;
    A.jdesc = jdesc;
    A.jn = jn;
    A.NC = cl.NC;
    A.V = V;
    A.W = g.W;
    A.H = g.H;
    A.L = g.L;
    (void)lane;
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        const int4 e = slow_list[i];
        if (posc)  // (not a hot kernel: both forms compiled in, chosen per launch)
            vb_job_slow<false>(A, lds_all[wave], e.x, e.y, e.z, e.w);
        else
            vb_job_slow<true>(A, lds_all[wave], e.x, e.y, e.z, e.w);
    }
}

// Stage 2b, as a kernel of its own: one WAVE per drawn job whose coverage and triangle ids were PUBLISHED to its slot
// (jid / jcov / jdesc) instead of being resolved by the wave that drew it.  Since round 5 the job kernel resolves its jobs
// itself, straight from LDS (vb_resolve_job above); what is left for this kernel are the jobs vb_slow_kernel redrew with
// the general triangle path (slow_list != NULL: exactly those), i.e. normally nothing -- the solver step only launches it
// together with vb_slow_kernel.  slow_list == NULL: every job slot of the chunk (the round-4 form of the chain, kept as
// the A/B reference: -DVB_INLINE_RESOLVE=0).
__global__ void __launch_bounds__(256)
vb_resolve_kernel(BinGeom g, int B, const float* __restrict__ verts, const float* __restrict__ mvp,
                  const float4* __restrict__ posc, int V, int T, const int4* __restrict__ tri4,
                  const int4* __restrict__ opp4, const unsigned* __restrict__ jid, const u64* __restrict__ jcov,
                  const int* __restrict__ jdesc,
                  int* __restrict__ jn, float* __restrict__ jval, VbItem* __restrict__ jitems,
                  int* __restrict__ jspill, int jcap, int want_grad, VbItem* __restrict__ spill, int spill_cap,
                  int* __restrict__ meta, int dbg, const int4* __restrict__ slow_list) {
    __shared__ VbResolveLds lds_all[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    VbResolveLds& S = lds_all[wave];
    const int W = g.W, H = g.H, L = g.L;
    (void)B;
    // XCD-aware like the job kernel (workgroup w runs on XCD w % 8; the slots of an eighth of the job list were written
    // through that XCD's L2), one job per wave and turn
    const bool listed = slow_list != nullptr;
    const int total = min(meta[5], jcap), per_xcd = (total + 7) >> 3, xcd = blockIdx.x & 7;
    const int jbeg = xcd * per_xcd, jend = min(jbeg + per_xcd, total);
    const int it0 = listed ? (int)blockIdx.x * 4 + wave : jbeg + (int)(blockIdx.x >> 3) * 4 + wave;
    const int it1 = listed ? *vb_line(meta, 17) : jend;
    const int step = listed ? (int)gridDim.x * 4 : (int)(gridDim.x >> 3) * 4;
    for (int it = it0; it < it1; it += step) {
        const int job = listed ? slow_list[it].x : it;
        if (job >= jcap) continue;
        const size_t slot = (size_t)job;
        // the ids are requested together with the descriptor (one round trip; an undrawn slot holds stale ids, unused)
        unsigned idw[VB_WORDS];
        {
            const unsigned* const src = jid + slot * VB_RN;
#pragma unroll
            for (int k = 0; k < VB_WORDS; k++) {
                const unsigned i = 64u * k + lane;
                idw[k] = (i < (unsigned)VB_RN) ? src[i] : 0xffffffffu;
            }
        }
        const u64 mycw = (lane < VB_WORDS) ? jcov[slot * VB_WORDS + lane] : 0ull;  // coverage bitmap of the region
        const int de = jdesc[job];
        if (de < 0) continue;  // nothing drawn: the job kernel has already marked the slot
        const int u = de & 511, tx = (de >> 9) & 1023, ty = (de >> 19) & 4095;
        const int b = u / L;
        const int rx0 = tx * EHR_TILE_W - 1, ry0 = ty * EHR_TILE_H - 1;
        u64 C[VB_WORDS];  // bit i = region pixel i is covered
#pragma unroll
        for (int k = 0; k < VB_WORDS; k++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mycw, k);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mycw >> 32), k);
            C[k] = ((u64)hi << 32) | lo;
        }
        VB_WAVE_SYNC();  // the previous job's reads of S are complete
#pragma unroll
        for (int k = 0; k < VB_WORDS; k++) {
            const unsigned i = 64u * k + lane;
            // covered pixels whose triangle was never asked for (no uncovered neighbour) carry a marker instead of an id
            if (i < (unsigned)VB_RN)
                S.ids[i] = (idw[k] != 0xffffffffu) ? idw[k] : (((C[k] >> lane) & 1ull) ? VB_ID_COVERED : 0xffffffffu);
        }
        VbResolveArgs Q;  // (wave-uniform; the compiler keeps what it needs in scalar registers)
        Q.verts = verts; Q.mvp = mvp; Q.posc = posc; Q.L = L; Q.tri4 = tri4; Q.opp4 = opp4; Q.jn = jn; Q.jval = jval; Q.jitems = jitems; Q.jspill = jspill;
        Q.spill = spill; Q.meta = meta; Q.V = V; Q.T = T; Q.W = W; Q.H = H; Q.spill_cap = spill_cap; Q.want_grad = want_grad;
        Q.dbg = dbg;
        if (posc)
            vb_resolve_job<false, false>(Q, S.ids, S.pairA, S.hits, C, slot, b, u - b * L, rx0, ry0);
        else
            vb_resolve_job<false, true>(Q, S.ids, S.pairA, S.hits, C, slot, b, u - b * L, rx0, ry0);
    }
}

// Per (view, tile) of a BOUND reference mask: the fixed-point value the composite kernel would add to the view's frame
// loss for a tile no link touches (mask = 0, so the tile contributes sum(ref^2), summed exactly like vb_composite_kernel
// sums it: four pixels per lane, then the shuffle tree), and per view the total over all tiles.  The reference mask does
// not change during a solve, so these are constants of the solve (ehr_fused_bind_ref); with them the composite kernel only
// visits tiles inside the view's link boxes and adds fix(new) - cached there.  Integer sums are associative: bit-identical
// to streaming every tile.
__global__ void __launch_bounds__(256)
vb_refsum_kernel(BinGeom g, int B, const float* __restrict__ ref, int vec_ok, long long* __restrict__ tsum,
                 long long* __restrict__ vtot, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= B * g.nt) return;
    const int W = g.W, H = g.H;
    const int b = gw / g.nt, tile = gw - b * g.nt;
    const int tx = tile % g.ntx, ty = tile / g.ntx;
    const int r = lane >> 3, c4 = (lane & 7) * 4;
    const int ix = tx * EHR_TILE_W + c4, iy = ty * EHR_TILE_H + r;
    const bool row_in = iy < H;
    const size_t im = ((size_t)b * H + (H - 1 - (row_in ? iy : 0))) * W + ix;
    float rf[4] = {0.f, 0.f, 0.f, 0.f};
    bool pin[4];
#pragma unroll
    for (int j = 0; j < 4; j++) pin[j] = row_in && (ix + j) < W;
    if (vec_ok) {
        if (pin[0]) {
            const float4 r4 = *reinterpret_cast<const float4*>(ref + im);
            rf[0] = r4.x; rf[1] = r4.y; rf[2] = r4.z; rf[3] = r4.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (pin[j]) rf[j] = ref[im + j];
    }
    float e2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (pin[j]) {
            const float e = 0.f - rf[j];
            e2 += e * e;
        }
    const float s = wave_sum(e2);
    if (lane == 0) {
        long long q = 0;
        if (!(fabsf(s) < 1.0e9f))
            atomicOr(flag, 1);  // every step on this reference reports the overflow (loss = NaN)
        else if (s != 0.f)
            q = fix_of(s);
        tsum[gw] = q;
        if (q != 0) atomicAdd((unsigned long long*)&vtot[b], (unsigned long long)q);
    }
}

// Stage 3: one WAVE per 32x8 tile (4 pixels per lane, float4 image accesses, no workgroup barriers): sums the links'
// values in link order, clamps, accumulates the frame loss, writes the mask, and back-propagates the tile's blended
// pairs to 12 numbers per link which go to the view's fixed-point accumulators.  Persistent waves over
//   tsum == NULL: every tile of every view (tiles no link box touches just stream: mask = 0, loss += ref^2);
//   tsum != NULL (bound reference mask): the tiles of the views' link rectangles only; a tile that no link contributes
//                 to is skipped without reading the reference -- its cached sum is already in vtot.  With a mask output
//                 the tiles outside every rectangle are then filled with zeros (stores only, no reference read, no sums).
// The workgroup that finishes LAST (a ticket per XCD, then one over the XCDs) runs the finish stage: accumulators ->
// loss / grad_mvp [-> pose backward -> Adam], re-arms the link boxes.  vec_ok: W % 4 == 0 and 16-byte aligned images.
template <bool TAIL, bool FILL>
__global__ void __launch_bounds__(256, FILL ? 5 : 6)
vb_composite_kernel(BinGeom g, int B, const float* __restrict__ mvp, int V, const float* __restrict__ verts,
                    int* __restrict__ lbox, const int* __restrict__ jn, const float* __restrict__ jval,
                    const VbItem* __restrict__ jitems, const int* __restrict__ jspill, const int* __restrict__ jbase,
                    const unsigned* __restrict__ jutile, int jcap, const float* __restrict__ ref,
                    float* __restrict__ mask, long long* __restrict__ facc, int nls, int want_grad, int vec_ok,
                    const VbItem* __restrict__ spill, int spill_cap, int* __restrict__ meta, int dbg,
                    const long long* __restrict__ tsum, const long long* __restrict__ vtot,
                    const int* __restrict__ ref_flag, float* __restrict__ loss,
                    float* __restrict__ grad_mvp, StepTail tail, int do_finish, int B_all,
                    const long long* __restrict__ facc_all, const long long* __restrict__ vtot_all,
                    int* __restrict__ lbox_all) {
    __shared__ float gpix_all[4][EHR_TILE_W * EHR_TILE_H];
    extern __shared__ int s_dyn[];  // [U + 1] first job of every (view, link) | [U] its tile range
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* const gpix = gpix_all[wave];
    const int W = g.W, H = g.H, L = g.L, U = B * L;
    int* const s_jbase = s_dyn;
    unsigned* const s_utile = reinterpret_cast<unsigned*>(s_dyn + U + 1);
    const bool sparse = tsum != nullptr;
    for (int i = tid; i <= U; i += 256) s_jbase[i] = jbase[i];
    for (int i = tid; i < U; i += 256) s_utile[i] = jutile[i];
    // The links' screen boxes start "empty" in the next step.  Nobody reads them after the job kernel's prologue, so the
    // call's last composite launch re-arms them here, a store per thread of its first workgroups, instead of in the
    // finish stage at its end (where it was 32 stores per thread on the step's critical path).
    if (do_finish && lbox_all)
        for (int i = (int)blockIdx.x * 256 + tid; i < 16 * B_all * g.L; i += (int)gridDim.x * 256)
            lbox_all[i] = (i & 2) ? INT_MIN : INT_MAX;  // 16 ints (one line) per box: min x, min y, max x, max y, padding
    __syncthreads();
    VbCompArgs C;
    C.g = g; C.B = B; C.mvp = mvp; C.V = V; C.verts = verts; C.jn = jn; C.jval = jval; C.jitems = jitems; C.jspill = jspill;
    C.jcap = jcap; C.ref = ref; C.mask = mask; C.facc = facc; C.nls = nls; C.want_grad = want_grad; C.vec_ok = vec_ok;
    C.spill = spill; C.spill_cap = spill_cap; C.meta = meta; C.dbg = dbg; C.tsum = tsum;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7;
    vb_composite_items<false, FILL>(C, s_jbase, s_utile, gpix, xcd, (int)(blockIdx.x >> 3) * 4 + wave, (nwg >> 3) * 4, nwg);
    // ---- the workgroup whose atomics are performed last runs the finish stage.  Every wave first waits until its own
    //      atomics have been performed (vmcnt covers them), then one lane takes a ticket on the XCD's counter and the
    //      last of an XCD one on the top counter: two levels, because a few thousand arrivals on ONE address serialise
    //      at ~12 ns each.  The accumulators are only ever touched by agent-scope atomics and read back with agent-scope
    //      loads (acc_load), so no cache maintenance is needed between the two.
    if (!do_finish) return;  // (not the last chunk of views: the kernel boundary orders its sums before the last chunk's finish)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int s_last;
    if (tid == 0) {
        int last = 0;
        const int per = nwg >> 3;  // workgroups per XCD counter (nwg is a multiple of 8)
        if (atomicAdd(vb_line(meta, 8 + xcd), 1) == per - 1) last = atomicAdd(vb_line(meta, 16), 1) == 7;
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (ref_flag && tid == 0 && ref_flag[0]) atomicOr(&meta[EHR_META_OVERFLOW], 1);  // the bound reference's own sums overflowed
    __syncthreads();
    __shared__ double S[4][17];
    __shared__ float red_lds[8];
#ifndef VB_NO_FINISH
    __shared__ float Js[6][16];
    finish_body<TAIL>(g, B_all, facc_all, sparse ? vtot_all : nullptr, loss, grad_mvp, meta, tail, nls, nullptr, VB_LOSS_STRIDE,
                      gpix_all[0], S, red_lds, Js);
#endif
}

// Content hash of the scoring op's mesh arrays: the cluster index holds copies of the vertex positions, so a mesh edited
// in place under the same pointers must rebuild it.  Every (word, position) pair goes through a 64-bit finaliser
// (murmur3's fmix64) before it is summed: a sum of the raw words, however weighted, is linear, and two compensating
// edits would cancel; the sum of the mixed values is still order independent.
__global__ void __launch_bounds__(256)
vb_hash_kernel(const unsigned* __restrict__ a, size_t na, const unsigned* __restrict__ b, size_t nb, const unsigned* __restrict__ c,
               size_t nc, unsigned long long* __restrict__ out) {
    unsigned long long h = 0;
    const size_t n = na + nb + nc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned w = i < na ? a[i] : (i < na + nb ? b[i - na] : c[i - na - nb]);
        unsigned long long x = (unsigned long long)w + 0x9e3779b97f4a7c15ull * ((unsigned long long)i + 1ull);
        x ^= x >> 33;
        x *= 0xff51afd7ed558ccdull;
        x ^= x >> 33;
        x *= 0xc4ceb9fe1a85ec53ull;
        x ^= x >> 33;
        h += x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

// Scoring op, last stage: one wave per (candidate, tile), four pixels per lane: c = number of the candidate's S views
// whose coverage word has the pixel's bit, score[q] += sum c (S - c) (space_explorer.py:152-165: the summed unbiased
// variance of S binary masks is that integer over S (S - 1)); optionally the count image (row 0 = top).  Also re-arms
// the link boxes for the next chunk of views.
__global__ void __launch_bounds__(256)
vb_score_count_kernel(BinGeom g, int nq, int S, const u64* __restrict__ tcov, unsigned long long* __restrict__ sacc,
                      unsigned char* __restrict__ count_img, int* __restrict__ lbox, int nlbox,
                      unsigned long long* __restrict__ prev_sacc, long long* __restrict__ prev_score, int prev_nq) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nlbox; i += gridDim.x * 256) lbox[i] = (i & 2) ? INT_MIN : INT_MAX;
    // The sums of a chunk are collected in scratch, one 128-byte line per candidate (a few hundred atomics per candidate:
    // on the caller's packed int64 array they would all hit one line and serialise at ~25 ns each), and stored to the
    // caller's array by the NEXT launch of this kernel (the last one has nq = 0).
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < prev_nq; i += 256) {
            prev_score[i] = (long long)prev_sacc[16 * i];
            prev_sacc[16 * i] = 0ull;
        }
    const int r = lane >> 3, c4 = (lane & 7) * 4;
    const int sh = (r & 1) * 32 + c4;
    // persistent waves (a workgroup per four items would be bound by the rate workgroups are dispatched at)
    for (int item = blockIdx.x * 4 + wave; item < nq * g.nt; item += gridDim.x * 4) {
    const int q = item / g.nt, tile = item - q * g.nt;
    int c[4] = {0, 0, 0, 0};
    // one round trip per 16 views: lane 4 s + k requests word k of view s, the words then go round by readlane
    for (int s0 = 0; s0 < S; s0 += 16) {
        const int ls = s0 + (lane >> 2);
        const u64 mine = (ls < S) ? tcov[(((size_t)q * g.nt + tile) * S + ls) * 4 + (lane & 3)] : 0ull;
        if (__ballot(mine != 0ull) == 0) continue;  // (most tiles hold nothing)

        const int ns = min(16, S - s0);
        for (int k = 0; k < ns; k++) {
            const int src = 4 * k + (r >> 1);
            const unsigned lo = (unsigned)__shfl((int)(unsigned)mine, src, 64), hi = (unsigned)__shfl((int)(unsigned)(mine >> 32), src, 64);
            const unsigned bits = (unsigned)(((((u64)hi) << 32) | lo) >> sh) & 15u;
#pragma unroll
            for (int j = 0; j < 4; j++) c[j] += (bits >> j) & 1u;
        }
    }
    int v = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) v += c[j] * (S - c[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0 && v) atomicAdd(&sacc[16 * q], (unsigned long long)v);
    if (count_img) {
        const int tx = tile % g.ntx, ty = tile / g.ntx;
        const int ix = tx * EHR_TILE_W + c4, iy = ty * EHR_TILE_H + r;
        if (iy < g.H)
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ix + j < g.W && c[j]) count_img[((size_t)q * g.H + (g.H - 1 - iy)) * g.W + ix + j] = (unsigned char)c[j];
    }
    }
}

// [T][3] int32 -> [T] int4 (one aligned 16-byte gather per triangle in the silhouette analysis)
__global__ void vb_pad_kernel(const int32_t* __restrict__ a, int T, int4* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) out[t] = make_int4(a[3 * t], a[3 * t + 1], a[3 * t + 2], 0);
}

}  // namespace ehr
using namespace ehr;

// ---- host side ------------------------------------------------------------------------------------------------------

// Reorders idx[0..n) (indices into the centroid array) in place: median split along the longest axis, left part
// rounded to a multiple of 64, recursively.  Against a Morton curve of the centroids (first version) the clusters'
// screen boxes touch a third fewer tiles (xArm7 at 720p: 4.4 k instead of 6.5 k cluster-tile pairs per view), i.e. a third
// fewer candidate clusters per job.
static void vb_kd_order(const float* cen, int* idx, int n) {
    if (n <= 64) return;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            const float c = cen[3 * (size_t)idx[i] + k];
            lo[k] = std::min(lo[k], c);
            hi[k] = std::max(hi[k], c);
        }
    int ax = 0;
    if (hi[1] - lo[1] > hi[ax] - lo[ax]) ax = 1;
    if (hi[2] - lo[2] > hi[ax] - lo[ax]) ax = 2;
    const int half = std::min(std::max(((n / 2 + 63) / 64) * 64, 64), n - 1);
    std::nth_element(idx, idx + half, idx + n, [&](int a, int b) {
        const float ca = cen[3 * (size_t)a + ax], cb = cen[3 * (size_t)b + ax];
        return ca < cb || (ca == cb && a < b);  // index as tie-break: the same scene always gives the same clusters
    });
    vb_kd_order(cen, idx, half);
    vb_kd_order(cen, idx + half, n - half);
}

// Static acceleration index of a scene: per link, the triangles ordered by a median-split tree over their centroids (object
// space, so it holds for every pose) and cut into clusters of 64.  Triangle ids stay the caller's: depth ties and the
// antialias topology are unaffected.
// tri_link == NULL: the links come from vert_link (the scoring op's packed mesh): a triangle belongs to the link of its
// vertices; *mixed is set if a triangle's vertices disagree or the triangles are not grouped by link (the caller then
// keeps to the per-triangle path).
static int vb_build_clusters(Scratch& out, int& nc_out, int L, int V, int T, const float* verts, const int32_t* tris,
                             const int32_t* tri_link, const int32_t* vert_link = nullptr, bool* mixed = nullptr) {
    std::vector<float> hv((size_t)3 * std::max(V, 1));
    std::vector<int32_t> ht((size_t)3 * std::max(T, 1)), hl((size_t)std::max(T, 1));
    if (V > 0) EHR_HIP(hipMemcpy(hv.data(), verts, (size_t)3 * V * sizeof(float), hipMemcpyDeviceToHost));
    if (T > 0) {
        EHR_HIP(hipMemcpy(ht.data(), tris, (size_t)3 * T * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (tri_link) {
            EHR_HIP(hipMemcpy(hl.data(), tri_link, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost));
        } else {
            std::vector<int32_t> vl((size_t)std::max(V, 1), 0);
            if (V > 0) EHR_HIP(hipMemcpy(vl.data(), vert_link, (size_t)V * sizeof(int32_t), hipMemcpyDeviceToHost));
            int prev = 0;
            for (int t = 0; t < T; t++) {
                int lk[3];
                for (int j = 0; j < 3; j++) {
                    const int v = ht[3 * (size_t)t + j];
                    lk[j] = ((unsigned)v < (unsigned)V) ? vl[v] : -1;
                }
                if (lk[0] != lk[1] || lk[0] != lk[2] || (unsigned)lk[0] >= (unsigned)L || lk[0] < prev) {
                    *mixed = true;
                    return EHR_OK;
                }
                prev = lk[0];
                hl[t] = lk[0];
            }
        }
    }
    std::vector<int32_t> ctri, clink, coff((size_t)L + 1, 0);
    std::vector<float> aabb((size_t)6 * L);
    for (int l = 0; l < L; l++)
        for (int k = 0; k < 3; k++) {
            aabb[6 * (size_t)l + k] = 3.4e38f;
            aabb[6 * (size_t)l + 3 + k] = -3.4e38f;
        }
    int t = 0;
    for (int l = 0; l < L; l++) {
        coff[l] = (int32_t)clink.size();
        const int t0 = t;
        while (t < T && hl[t] == l) t++;
        if (t < T && hl[t] < l) return fail(EHR_ERR_INVALID, "ehr_fused_plan: tri_link must be sorted by link");
        const int n = t - t0;
        if (n == 0) continue;
        std::vector<float> cen((size_t)3 * n);
        float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                float c = 0.f;
                for (int j = 0; j < 3; j++) {
                    const int v = ht[3 * (size_t)(t0 + i) + j];
                    const float x = ((unsigned)v < (unsigned)V) ? hv[3 * (size_t)v + k] : 0.f;
                    c += x;
                    if ((unsigned)v < (unsigned)V) {
                        aabb[6 * (size_t)l + k] = std::min(aabb[6 * (size_t)l + k], x);
                        aabb[6 * (size_t)l + 3 + k] = std::max(aabb[6 * (size_t)l + 3 + k], x);
                    }
                }
                c *= (1.f / 3.f);
                if (!(c == c)) c = 0.f;
                cen[3 * (size_t)i + k] = c;
                lo[k] = std::min(lo[k], c);
                hi[k] = std::max(hi[k], c);
            }
        // order the link's triangles so that every run of 64 is spatially compact: recursive median split along the
        // longest axis of the centroids, the left part a multiple of 64 (so that clusters never straddle a split)
        std::vector<int> order((size_t)n);
        for (int i = 0; i < n; i++) order[i] = i;
        vb_kd_order(cen.data(), order.data(), n);
        for (int i = 0; i < n; i++) {
            if ((i & 63) == 0) clink.push_back(l);
            ctri.push_back(t0 + order[i]);
        }
        while (ctri.size() & 63) ctri.push_back(-1);
    }
    if (t != T) return fail(EHR_ERR_INVALID, "ehr_fused_plan: tri_link holds a link outside [0, %d) or is not sorted", L);
    coff[L] = (int32_t)clink.size();
    const int NC = (int)clink.size();
    nc_out = NC;
    int rc;
    const size_t n_ctri = (size_t)std::max(NC, 1) * 64;
    std::vector<float> cvert(12 * n_ctri, 0.f);  // three float4 planes
    for (size_t i = 0; i < ctri.size(); i++) {
        const int tt = ctri[i];
        if (tt < 0) continue;
        const int v0 = ht[3 * (size_t)tt], v1 = ht[3 * (size_t)tt + 1], v2 = ht[3 * (size_t)tt + 2];
        if ((unsigned)v0 >= (unsigned)V || (unsigned)v1 >= (unsigned)V || (unsigned)v2 >= (unsigned)V) continue;
        const float c9[9] = {hv[3 * (size_t)v0], hv[3 * (size_t)v0 + 1], hv[3 * (size_t)v0 + 2], hv[3 * (size_t)v1], hv[3 * (size_t)v1 + 1],
                             hv[3 * (size_t)v1 + 2], hv[3 * (size_t)v2], hv[3 * (size_t)v2 + 1], hv[3 * (size_t)v2 + 2]};
        for (int k = 0; k < 4; k++) cvert[4 * i + k] = c9[k];
        for (int k = 0; k < 4; k++) cvert[4 * (n_ctri + i) + k] = c9[4 + k];
        cvert[4 * (2 * n_ctri + i)] = c9[8];
        cvert[4 * (2 * n_ctri + i) + 1] = 1.f;
    }
    std::vector<int32_t> cvidx(4 * n_ctri, 0);  // {v0, v1, v2, triangle} of every cluster slot
    for (size_t i = 0; i < n_ctri; i++) {
        const int tt = i < ctri.size() ? ctri[i] : -1;
        cvidx[4 * i + 3] = tt;
        if (tt < 0) continue;
        for (int k = 0; k < 3; k++) {
            const int v = ht[3 * (size_t)tt + k];
            cvidx[4 * i + k] = ((unsigned)v < (unsigned)V) ? v : 0;  // (such a triangle draws nothing: never looked up)
        }
    }
    for (int l = 0; l < L; l++)  // a deferred unit names its triangle by an 18-bit slot relative to its link's first cluster
        if (coff[l + 1] - coff[l] > 4096)
            return fail(EHR_ERR_INVALID, "ehr_fused_plan: link %d has more than 262144 triangles", l);
    if ((rc = out.reserve((n_ctri + std::max(NC, 1) + L + 1 + 6 * (size_t)L) * sizeof(int32_t) + 32 + cvert.size() * sizeof(float) +
                          cvidx.size() * sizeof(int32_t)))) return rc;
    int32_t* d = (int32_t*)out.ptr;
    if (NC > 0) {
        EHR_HIP(hipMemcpy(d, ctri.data(), (size_t)NC * 64 * sizeof(int32_t), hipMemcpyHostToDevice));
        EHR_HIP(hipMemcpy(d + n_ctri, clink.data(), (size_t)NC * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    EHR_HIP(hipMemcpy(d + n_ctri + std::max(NC, 1), coff.data(), ((size_t)L + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    EHR_HIP(hipMemcpy(d + n_ctri + std::max(NC, 1) + L + 1, aabb.data(), aabb.size() * sizeof(float), hipMemcpyHostToDevice));
    {
        const uintptr_t at = ((uintptr_t)(d + n_ctri + std::max(NC, 1) + L + 1 + 6 * (size_t)L) + 15) & ~(uintptr_t)15;
        EHR_HIP(hipMemcpy((void*)at, cvert.data(), cvert.size() * sizeof(float), hipMemcpyHostToDevice));
        EHR_HIP(hipMemcpy((char*)at + cvert.size() * sizeof(float), cvidx.data(), cvidx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return EHR_OK;
}

int ehr::vbuf_plan(ehr_ctx* ctx, int B, int L, int V, int T, int H, int W, float slack, const float* verts,
                   const int32_t* tris, const int32_t* tri_link, const int32_t* opp) {
    if (H > 32760 || W > 32736)  // tile counts are packed into 10 (columns) and 12 (rows) bits
        return fail(EHR_ERR_INVALID, "ehr_fused_plan: resolution above 32736 x 32760 (W x H) is unsupported");
    if ((V > 0 && !verts) || (T > 0 && (!tris || !tri_link || !opp)))
        return fail(EHR_ERR_INVALID, "ehr_fused_plan: the scene arrays (verts, tris, tri_link, opp) are required");
    int rc;
    if ((rc = vb_build_clusters(ctx->vb_clus, ctx->vb_nc, L, V, T, verts, tris, tri_link))) return rc;
    const int NC = std::max(ctx->vb_nc, 1);
    // The views of a call go through the chain in CHUNKS of Bc views (one chunk in the common case): the job kernel
    // keeps its (view, link) tables in LDS (VB_MAX_UNITS entries), and the per-chunk scratch -- clip-space vertices,
    // raster records, job slots (3.5 KB each) -- is bounded to 24 GB of the 288 (EHR_VB_SCRATCH_MB) however many views a
    // call brings: a chunk costs a pass of the chain with its own tails, so chunks are as large as they may be (the reference
    // batches all frames of a data set in one step, configs/xarm7/example.yaml: batch_size 100).  A job = a (link, tile)
    // pair whose boxes touch; by default (`slack` <= 0) one slot per (link, tile) is provided, so nothing can overflow;
    // `slack` > 0 provides `slack` jobs per view tile instead (a fraction is fine: a robot's links touch a twentieth of
    // the tiles of a typical frame; less scratch, larger chunks); a view that needs more (every pixel under more than
    // `slack` link boxes on average) is reported (loss = NaN, ehr_fused_status).
    BinGeom gp = make_geom(H, W, L);
    const size_t slot_bytes = 256 * sizeof(float) + VB_JOB_ITEMS * sizeof(VbItem) + VB_WORDS * sizeof(u64) + 2 * sizeof(int) +
                              VB_RN * sizeof(unsigned) + sizeof(int) + sizeof(int4);
    const double jobs_per_view = std::max(1.0, ((slack > 0.f) ? std::min((double)L, (double)slack) : (double)L) * gp.nt);
    const double view_bytes = jobs_per_view * slot_bytes + (double)NC * (64 * 40 + 8) + (double)std::max(V, 1) * 16 +
                              (double)L * gp.nt * 4;
    static const double budget = getenv("EHR_VB_SCRATCH_MB") ? atof(getenv("EHR_VB_SCRATCH_MB")) * 1048576.0 : 24576.0 * 1048576.0;
    int Bc = std::min(B, std::max(1, VB_MAX_UNITS / L));
    Bc = std::max(1, std::min(Bc, (int)(budget / view_bytes)));
    if (jobs_per_view * Bc > 2.0e9) return fail(EHR_ERR_INVALID, "ehr_fused_plan: views x links x tiles of a chunk exceeds 2e9");
    ctx->vb_chunk = Bc;
    if ((rc = ctx->vb_acc.reserve(((size_t)B * (12 * (size_t)L + VB_LOSS_SLOTS * VB_LOSS_STRIDE)) * sizeof(long long) + EHR_META_INTS * sizeof(int) + (VB_LINES + 1) * 128))) return rc;
    {   // Clip-space vertices: kept per view (posc, 16 B per vertex and view, written by the vertex kernel) or computed where
        // they are looked up (VbLazy).  Lazy pays where vertices outnumber the look-ups by far -- meshes with unshared
        // vertices (V ~ 3 T): Franka 16 x 1080p spends 168 of the vertex launch's 286 MB on them -- and costs the job kernel
        // registers it does not have (8 views xArm7: +3.5 us), hence a choice per plan.  EHR_VB_LAZY=0/1 overrides.
        const char* e = getenv("EHR_VB_LAZY");
        ctx->vb_lazy = e ? atoi(e) != 0 : (double)V > 1.5 * (double)std::max(T, 1);
        if (ctx->vb_lazy != (ctx->vb_posc.ptr == nullptr) && ctx->gexec) {  // (a captured chain has the form baked in)
            EHR_HIP(hipGraphExecDestroy(ctx->gexec));
            ctx->gexec = nullptr;
        }
        if (!ctx->vb_lazy && (rc = ctx->vb_posc.reserve((size_t)Bc * std::max(V, 1) * sizeof(float4)))) return rc;
    }
    // per step and (view, cluster slot): trec 32 B | tbox 8 B, then cbox 8 B per (view, cluster)
    if ((rc = ctx->vb_boxes.reserve((size_t)Bc * NC * (64 * 40 + 8)))) return rc;
    {  // pool of blended pairs for jobs that exceed their slot (EHR_VB_SPILL_ITEMS: test hook for the overflow path)
        const char* e = getenv("EHR_VB_SPILL_ITEMS");
        ctx->vb_spill_cap = e ? std::max(0, atoi(e)) : VB_SPILL_ITEMS;
        // (the -DVB_TIMELINE profiling build parks its per-wave records here: keep room for them)
        if ((rc = ctx->vb_spill.reserve(std::max((size_t)ctx->vb_spill_cap * sizeof(VbItem), (size_t)1 << 20)))) return rc;
    }
    if ((rc = ctx->vb_units.reserve((size_t)VB_LBOX_STRIDE * B * L * sizeof(int)))) return rc;  // link boxes (one 64-byte line each), all views
    {  // a bound reference mask's cached sums: tsum [B][nt] | vtot [B] | flag
        if ((rc = ctx->vb_refsum.reserve(((size_t)B * gp.nt + B + 1) * sizeof(long long)))) return rc;
        ctx->vb_ref = nullptr;  // a new plan forgets the binding
    }
    {  // job slots of a chunk, compact (numbered like the jobs): value tile 1 KB | items 1 KB | coverage words | count | spill
       // base | region ids 1.36 KB | descriptor | an entry of the slow-job list; then the links' first jobs and tile ranges
        ctx->vb_jcap = (int)(jobs_per_view * Bc);
        const size_t nslot = (size_t)ctx->vb_jcap;
        if ((rc = ctx->vb_jobs.reserve(nslot * slot_bytes + (2 * (size_t)Bc * L + 1) * sizeof(int) + 32))) return rc;
    }
    {  // heavy-job hint: generation + two counts | two lists | stamp table (dense ids of a chunk)
        const size_t ints = 8 + 2 * (size_t)VB_HEAVY_CAP + 2 * (size_t)VB_MED_CAP + (size_t)Bc * L * gp.nt;
        if ((rc = ctx->vb_heavy.reserve(ints * sizeof(int)))) return rc;
        EHR_HIP(hipMemset(ctx->vb_heavy.ptr, 0, ints * sizeof(int)));
    }
    if ((rc = ctx->vb_idx.reserve((size_t)2 * std::max(T, 1) * sizeof(int4)))) return rc;
    if (T > 0) {
        vb_pad_kernel<<<(T + 255) / 256, 256>>>(tris, T, (int4*)ctx->vb_idx.ptr);
        vb_pad_kernel<<<(T + 255) / 256, 256>>>(opp, T, (int4*)ctx->vb_idx.ptr + T);
        EHR_LAUNCH_CHECK();
    }
    if (!ctx->vb_hstate.ptr) {  // (once per context: a re-plan after a reported step must not forget that step)
        if ((rc = ctx->vb_hstate.reserve(16 * sizeof(int)))) return rc;
        EHR_HIP(hipMemset(ctx->vb_hstate.ptr, 0, 16 * sizeof(int)));
    }
    ctx->vb_plan_tris = tris;
    ctx->vb_plan_opp = opp;
    ctx->vb_plan_verts = verts;
    EHR_HIP(hipMemset(ctx->vb_acc.ptr, 0, ctx->vb_acc.cap));
    std::vector<int> boxes((size_t)VB_LBOX_STRIDE * B * L);  // link boxes start "empty"; the finish kernel re-arms them
    for (size_t i = 0; i < boxes.size(); i++) boxes[i] = (i & 2) ? INT_MIN : INT_MAX;
    EHR_HIP(hipMemcpy(ctx->vb_units.ptr, boxes.data(), boxes.size() * sizeof(int), hipMemcpyHostToDevice));
    EHR_HIP(hipDeviceSynchronize());
    return EHR_OK;
}

int ehr::vbuf_meta_read(ehr_ctx* ctx, int* meta4) {
    const size_t off = (size_t)ctx->pB * (12 * (size_t)ctx->pL + VB_LOSS_SLOTS * VB_LOSS_STRIDE) * sizeof(long long);
    EHR_HIP(hipMemcpy(meta4, (char*)ctx->vb_acc.ptr + off, 4 * sizeof(int), hipMemcpyDeviceToHost));
    if (getenv("EHR_VB_PRINT")) {  // diagnostics
        int m8[8];
        EHR_HIP(hipMemcpy(m8, (char*)ctx->vb_acc.ptr + off, sizeof(m8), hipMemcpyDeviceToHost));
        fprintf(stderr, "[ehr vbuf] overflow %d spill %d jobs %d\n", m8[EHR_META_OVERFLOW], m8[EHR_META_SPILL], m8[5]);
        int hg[3];
        EHR_HIP(hipMemcpy(hg, ctx->vb_heavy.ptr, sizeof(hg), hipMemcpyDeviceToHost));
        fprintf(stderr, "[ehr vbuf] heavy jobs: generation %d, lists %d / %d\n", hg[0], hg[1], hg[2]);
        int cur[8];
        for (int k = 0; k < 8; k++)
            EHR_HIP(hipMemcpy(&cur[k], vb_line((int*)((char*)ctx->vb_acc.ptr + off), k), sizeof(int), hipMemcpyDeviceToHost));
        fprintf(stderr, "[ehr vbuf] job cursors %d %d %d %d %d %d %d %d\n", cur[0], cur[1], cur[2], cur[3], cur[4], cur[5], cur[6], cur[7]);
#ifdef VB_MERGE_TL
        {   // merged kernel: per workgroup [start, arrived, let go, items done, finish done] on the 100 MHz clock
            const int nwg = ((ctx->num_cus * 4) + 7) & ~7;
            std::vector<long long> tl((size_t)8 * nwg);
            EHR_HIP(hipMemcpy(tl.data(), ctx->vb_spill.ptr, tl.size() * sizeof(long long), hipMemcpyDeviceToHost));
            long long t0 = tl[0], a_last = 0, go_first = 1ll << 62, go_last = 0, it_last = 0, fin = 0;
            double a_mean = 0, it_mean = 0;
            for (int i = 0; i < nwg; i++) t0 = std::min(t0, tl[8 * i]);
            for (int i = 0; i < nwg; i++) {
                a_last = std::max(a_last, tl[8 * i + 1] - t0);
                a_mean += (double)(tl[8 * i + 1] - t0) / nwg;
                go_first = std::min(go_first, tl[8 * i + 2] - t0);
                go_last = std::max(go_last, tl[8 * i + 2] - t0);
                it_last = std::max(it_last, tl[8 * i + 3] - t0);
                it_mean += (double)(tl[8 * i + 3] - tl[8 * i + 2]) / nwg;
                if (tl[8 * i + 4]) fin = tl[8 * i + 4] - t0;
            }
            fprintf(stderr, "[ehr merge] workgroups arrive: mean %.1f last %.1f us; let go: first %.1f last %.1f us; items done: last %.1f us (mean %.1f us of work); finish done %.1f us\n",
                    a_mean * 0.01, a_last * 0.01, go_first * 0.01, go_last * 0.01, it_last * 0.01, it_mean * 0.01, fin * 0.01);
        }
#endif
#ifdef VB_TIMELINE
        {
            // Timeline of the job kernel's waves (100 MHz clock), written into the (otherwise idle) spill pool: when
            // they started, left the heavy phase and ended; what their single-wave jobs amounted to; where they ran.
            const int nw = 4 * (((ctx->num_cus * 4) + 7) & ~7);
            std::vector<long long> tl((size_t)21 * nw);
            EHR_HIP(hipMemcpy(tl.data(), ctx->vb_spill.ptr, tl.size() * sizeof(long long), hipMemcpyDeviceToHost));
            long long t0 = tl[0], t1 = tl[1];
            for (int i = 0; i < nw; i++) {
                t0 = std::min(t0, tl[4 * i]);
                t1 = std::max(t1, tl[4 * i + 1]);
            }
            int hs[16] = {0}, he[16] = {0}, hh[16] = {0};
            double busy = 0;
            std::vector<std::pair<long long, int>> order;
            std::vector<long long> per(8 * 8 * 2 * 16 * 4, 0);  // survivors per SIMD
            for (int i = 0; i < nw; i++) {
                hs[std::min<long long>((tl[4 * i] - t0) / 500, 15)]++;
                he[std::min<long long>((tl[4 * i + 1] - t0) / 500, 15)]++;
                hh[std::min<long long>((tl[4 * i + 2] - t0) / 500, 15)]++;
                busy += (double)(tl[4 * i + 1] - tl[4 * i]);
                order.push_back(std::make_pair(-tl[4 * i + 1], i));
                const long long x = tl[4 * i + 3];
                const unsigned hw = (unsigned)(x >> 32) & 0xffff;
                per[((((size_t)((x >> 48) & 7) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)) * 4 + ((hw >> 4) & 3)] += (x >> 20) & 0xfff;
            }
            std::sort(order.begin(), order.end());
            fprintf(stderr, "[ehr timeline] job kernel: %d waves, span %.1f us, mean wave life %.1f us\n", nw, (t1 - t0) * 0.01, busy / nw * 0.01);
            fprintf(stderr, "[ehr timeline] starts per 5 us:");
            for (int i = 0; i < 16; i++) fprintf(stderr, " %d", hs[i]);
            fprintf(stderr, "\n[ehr timeline] heavy phase left per 5 us:");
            for (int i = 0; i < 16; i++) fprintf(stderr, " %d", hh[i]);
            fprintf(stderr, "\n[ehr timeline] ends per 5 us:");
            for (int i = 0; i < 16; i++) fprintf(stderr, " %d", he[i]);
            fprintf(stderr, "\n[ehr timeline] last waves: wave (workgroup): start, heavy phase left, end [us]; single-wave jobs, max / sum survivors; place\n");
            for (int k = 0; k < 12 && k < nw; k++) {
                const int i = order[k].second;
                const long long x = tl[4 * i + 3];
                const unsigned hw = (unsigned)(x >> 32) & 0xffff;
                const long long* tx = &tl[4 * (size_t)nw + 12 * i];
                fprintf(stderr, "   %5d (%4d): %5.1f %5.1f %5.1f ; %lld jobs, %lld / %lld ; %lld units in %lld rounds ; %lld flushes: %lld of %lld units tested ; kcycles stage %.0f search %.0f walk %.0f flush %.0f resolve %.0f total %.0f cull-wait %.0f (%lld clusters, %lld groups) ; xcc %lld cu %u simd %u\n", i, i / 4,
                        (tl[4 * i] - t0) * 0.01, (tl[4 * i + 2] - t0) * 0.01, (tl[4 * i + 1] - t0) * 0.01, x & 0xff, (x >> 8) & 0xfff,
                        (x >> 20) & 0xfff, tx[0], tx[1], tx[2] & 0xffff, (tx[2] >> 16) & 0xffffff, tx[2] >> 40, tx[3] * 1e-3, tx[4] * 1e-3, tx[5] * 1e-3,
                        tx[6] * 1e-3, tx[10] * 1e-3, tx[8] * 1e-3, (tx[11] & 0xffffffffll) * 1e-3, (tx[11] >> 32) & 0xffff, (tx[11] >> 48) & 0xffff, (x >> 48) & 0xf, (hw >> 8) & 15, (hw >> 4) & 3);
            }
            {   // where the waves' cycles go, all waves together (single-wave jobs; the phases nest as in the per-wave lines)
                double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < nw; i++) {
                    const long long* tx = &tl[4 * (size_t)nw + 12 * i];
                    ph[0] += (double)tx[3]; ph[1] += (double)tx[4]; ph[2] += (double)tx[5]; ph[3] += (double)tx[6];
                    ph[4] += (double)tx[10]; ph[5] += (double)tx[8]; ph[6] += (double)(tx[11] & 0xffffffffll);
                    ph[7] += (double)tx[9];
                }
                fprintf(stderr, "[ehr timeline] all waves, Mcycles: stage %.1f search %.1f walk %.1f flush %.1f resolve %.1f total %.1f claim + cull %.1f (cull-wait %.1f) ; wave life %.1f\n",
                        ph[0] * 1e-6, ph[1] * 1e-6, ph[2] * 1e-6, ph[3] * 1e-6, ph[4] * 1e-6, ph[5] * 1e-6, ph[7] * 1e-6, ph[6] * 1e-6, busy * 24.0 * 1e-6);
            }
            {   // Where a late helper could still help: the last job of the waves the kernel ends on, against the moment the
                // lists of jobs ran dry (task 1b: late sharing inside a workgroup)
                std::vector<long long> dry;
                for (int i = 0; i < nw; i++)
                    if (tl[16 * (size_t)nw + 5 * i + 4]) dry.push_back(tl[16 * (size_t)nw + 5 * i + 4] - t0);
                std::sort(dry.begin(), dry.end());
                const double d10 = dry.empty() ? 0.0 : dry[dry.size() / 10] * 0.01, d50 = dry.empty() ? 0.0 : dry[dry.size() / 2] * 0.01;
                fprintf(stderr, "[ehr timeline] waves find their list of jobs empty: first %.1f, 10 %% %.1f, median %.1f us (%zu waves)\n",
                        dry.empty() ? 0.0 : dry[0] * 0.01, d10, d50, dry.size());
                fprintf(stderr, "[ehr timeline] last job of the last waves: start, rounds done, depth tests done, resolved [us]; rounds still to run when 10 %% / half of the waves were idle\n");
                double left10 = 0, left50 = 0, tail10 = 0, tail50 = 0;
                const int NL = std::min(64, nw);
                for (int k = 0; k < NL; k++) {
                    const int i = order[k].second;
                    const long long* ty = &tl[16 * (size_t)nw + 5 * i];
                    if (!ty[0]) continue;  // (a wave without a single-wave job)
                    const double a = (ty[0] - t0) * 0.01, b = (ty[1] - t0) * 0.01, c = (ty[2] - t0) * 0.01, d = (ty[3] - t0) * 0.01;
                    if (k < 12) fprintf(stderr, "   %5d: %5.1f %5.1f %5.1f %5.1f ; %4.1f / %4.1f us\n", i, a, b, c, d, std::max(0.0, b - std::max(a, d10)), std::max(0.0, b - std::max(a, d50)));
                    left10 += std::max(0.0, b - std::max(a, d10)) / NL;
                    left50 += std::max(0.0, b - std::max(a, d50)) / NL;
                    tail10 += std::max(0.0, d - std::max(b, d10)) / NL;
                    tail50 += std::max(0.0, d - std::max(b, d50)) / NL;
                }
                fprintf(stderr, "[ehr timeline] mean over the %d last waves: rounds still to run %.1f / %.1f us, depth tests + resolve after that %.1f / %.1f us\n", NL, left10, left50, tail10, tail50);
            }
            long long mx = 0, sum = 0;
            int used = 0;
            for (size_t k = 0; k < per.size(); k++) {
                mx = std::max(mx, per[k]);
                sum += per[k];
                used += per[k] > 0;
            }
            fprintf(stderr, "[ehr timeline] single-wave survivors per SIMD: %d SIMDs with work, mean %.0f, max %lld\n", used, used ? (double)sum / used : 0.0, mx);
            {
                long long us = 0, rs = 0, umax = 0;
                long long fl = 0, te = 0, de = 0;
                double cyc[4] = {0, 0, 0, 0}, jc[3] = {0, 0, 0}, cw = 0;
                long long ncand = 0, ngrp = 0;
                for (int i = 0; i < nw; i++) {
                    const long long* tx = &tl[4 * (size_t)nw + 12 * i];
                    us += tx[0];
                    rs += tx[1];
                    umax = std::max(umax, tx[0]);
                    fl += tx[2] & 0xffff;
                    te += (tx[2] >> 16) & 0xffffff;
                    de += tx[2] >> 40;
                    for (int k = 0; k < 4; k++) cyc[k] += (double)tx[3 + k];
                    for (int k = 0; k < 3; k++) jc[k] += (double)tx[8 + k];
                    cw += (double)(tx[11] & 0xffffffffll);
                    ncand += (tx[11] >> 32) & 0xffff;
                    ngrp += (tx[11] >> 48) & 0xffff;
                }
                fprintf(stderr, "[ehr timeline] single-wave jobs: %lld units in %lld rounds (%.0f units per round), at most %lld units on one wave\n", us, rs, rs ? (double)us / rs : 0.0, umax);
                {   // heavy phase: cost of the job a workgroup shared vs the time it took
                    std::vector<std::pair<long long, long long>> hp;
                    for (int i = 0; i < nw; i += 4) {
                        const long long* tx = &tl[4 * (size_t)nw + 12 * i];
                        if (tx[7] > 0) hp.push_back(std::make_pair(tl[4 * i + 2] - tl[4 * i], tx[7]));
                    }
                    std::sort(hp.begin(), hp.end());
                    fprintf(stderr, "[ehr timeline] heavy phase: %zu workgroups; (us, cost) of every 16th by duration:", hp.size());
                    for (size_t k = 0; k < hp.size(); k += std::max<size_t>(1, hp.size() / 16)) fprintf(stderr, " (%.1f, %lld)", hp[k].first * 0.01, hp[k].second);
                    if (!hp.empty()) fprintf(stderr, " (%.1f, %lld)", hp.back().first * 0.01, hp.back().second);
                    fprintf(stderr, "\n");
                }
                fprintf(stderr, "[ehr timeline] single-wave jobs (drawn ones): %.1f wave-Mcycles in total, claim + set-up %.1f, publish / resolve %.1f; triangle boxes of %lld candidate clusters in %lld groups: %.1f waiting\n", jc[0] * 1e-6, jc[1] * 1e-6, jc[2] * 1e-6, ncand, ngrp, cw * 1e-6);
                fprintf(stderr, "[ehr timeline] flushes %lld, deferred units %lld, depth-tested units %lld; wave-Mcycles: staging %.1f, prefix+search %.1f, walk %.1f, flush %.1f (of %.1f in total)\n",
                        fl, de, te, cyc[0] * 1e-6, cyc[1] * 1e-6, cyc[2] * 1e-6, cyc[3] * 1e-6, busy * 0.01 * 2100.0 * 1e-6);
            }
        }
#endif
    }
    return EHR_OK;
}

int ehr::vbuf_chain(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* tri_link,
                    const int32_t* vert_link, const int32_t* opp, float* mvp, const float* ref, int B, int L, int V,
                    int T, int H, int W, float* mask, float* loss, float* grad_mvp, const StepHead* head,
                    const StepTail* tail, hipStream_t stream) {
    (void)tri_link;
    if (tris != ctx->vb_plan_tris || opp != ctx->vb_plan_opp || verts != ctx->vb_plan_verts)
        return fail(EHR_ERR_INVALID, "fused op: the scene arrays differ from the planned ones; call ehr_fused_plan again");
    BinGeom g = make_geom(H, W, L);
    long long* const facc_all = (long long*)ctx->vb_acc.ptr;
    const int acc_stride = 12 * L + VB_LOSS_SLOTS * VB_LOSS_STRIDE;
    int* meta = (int*)(facc_all + (size_t)B * acc_stride);
    int* const lbox_all = (int*)ctx->vb_units.ptr;
    VbItem* spill = (VbItem*)ctx->vb_spill.ptr;
    const int NC = ctx->vb_nc, NC1 = std::max(NC, 1);
    VbClusters cl;
    cl.ctri = (const int32_t*)ctx->vb_clus.ptr;
    cl.clink = cl.ctri + (size_t)NC1 * 64;
    cl.coff = cl.clink + NC1;
    cl.laabb = (const float*)(cl.coff + L + 1);
    cl.cvert = (const float4*)(((uintptr_t)(cl.laabb + 6 * (size_t)L) + 15) & ~(uintptr_t)15);
    cl.NC = NC;
    VbSlotIdx si;
    si.cvidx = (const int4*)(cl.cvert + 3 * (size_t)NC1 * 64);
    VbHeavy hv;
    hv.gen = (int*)ctx->vb_heavy.ptr;
    hv.list = hv.gen + 8;
    hv.mlist = hv.list + 2 * VB_HEAVY_CAP;
    hv.stamp = hv.mlist + 2 * VB_MED_CAP;

    hipEvent_t* ev = nullptr;
    if (ctx->timing) {
        const size_t need = ctx->ev_used + EHR_FUSED_STAGES + 1;
        while (ctx->ev.size() < need) {
            hipEvent_t e;
            EHR_HIP(hipEventCreate(&e));
            ctx->ev.push_back(e);
        }
        ev = ctx->ev.data() + ctx->ev_used;
        ctx->ev_used = need;
        EHR_HIP(hipEventRecord(ev[0], stream));
    }
    const int vec_ok = ((W & 3) == 0) && (((uintptr_t)ref & 15) == 0) && (!mask || ((uintptr_t)mask & 15) == 0);
    static const int dbg_env = getenv("EHR_VB_DEBUG") ? atoi(getenv("EHR_VB_DEBUG")) : 0;  // measurement aid only
    static const int job_grid = getenv("EHR_VB_JOB_GRID") ? atoi(getenv("EHR_VB_JOB_GRID")) : 4;   // tuning knob
    static const int heavy_t = getenv("EHR_VB_HEAVY_T") ? atoi(getenv("EHR_VB_HEAVY_T")) : VB_HEAVY_T_DEFAULT;  // tuning knob
    static const int med_t = getenv("EHR_VB_MED_T") ? atoi(getenv("EHR_VB_MED_T")) : VB_MED_T_DEFAULT;        // tuning knob
    hv.mcap = std::min(VB_MED_CAP, 2 * (((ctx->num_cus * std::max(1, job_grid)) + 7) & ~7));
    hv.heavy_base = heavy_t;
    hv.heavy_max = ((((ctx->num_cus * std::max(1, job_grid)) + 7) & ~7)) / 2;
    static const int vertex_grid = getenv("EHR_VB_VERTEX_GRID") ? atoi(getenv("EHR_VB_VERTEX_GRID")) : 5;  // tuning knob
    static const int xcd_align = getenv("EHR_VB_XCD") ? atoi(getenv("EHR_VB_XCD")) : 1;  // tuning knob
    static const int res_grid = getenv("EHR_VB_RESOLVE_GRID") ? atoi(getenv("EHR_VB_RESOLVE_GRID")) : 5;  // tuning knob (5 workgroups per CU are resident)
    static const int no_sparse = getenv("EHR_VB_NO_SPARSE") ? atoi(getenv("EHR_VB_NO_SPARSE")) : 0;  // A/B aid
    static const int no_sparse_mask = getenv("EHR_VB_NO_SPARSE_MASK") ? atoi(getenv("EHR_VB_NO_SPARSE_MASK")) : 0;  // A/B aid
    static const int comp_grid = getenv("EHR_VB_COMPOSITE_GRID") ? atoi(getenv("EHR_VB_COMPOSITE_GRID")) : 6;  // tuning knob (6 resident per CU)
    const bool sparse = !no_sparse && (!mask || no_sparse_mask == 0) && ctx->vb_ref != nullptr && ctx->vb_ref == ref;
    const long long* const tsum_all = sparse ? (const long long*)ctx->vb_refsum.ptr : nullptr;
    const long long* const vtot_all = sparse ? tsum_all + (size_t)B * g.nt : nullptr;
    const int* const ref_flag = sparse ? (const int*)(vtot_all + B) : nullptr;
    const int Bc = std::max(1, std::min(ctx->vb_chunk, B));
    // (the heavy-job hint names jobs by their dense id inside a chunk: with more than one chunk it is switched off)
    const int dbg = dbg_env | (Bc < B ? 64 : 0);
    const size_t nslot = (size_t)ctx->vb_jcap;
    float* jval = (float*)ctx->vb_jobs.ptr;
    VbItem* jitems = (VbItem*)(jval + nslot * 256);
    u64* jcov = (u64*)(jitems + nslot * VB_JOB_ITEMS);
    int* jn = (int*)(jcov + nslot * VB_WORDS);
    int* jspill = jn + nslot;
    unsigned* jid = (unsigned*)(jspill + nslot);
    int* jdesc = (int*)(jid + nslot * VB_RN);
    int* jbase = jdesc + nslot;                                 // [Bc * L + 1] first job of every (view, link) of the chunk
    unsigned* jutile = (unsigned*)(jbase + (size_t)Bc * L + 1);  // [Bc * L] its tile range
    int4* slow_list = (int4*)(((uintptr_t)(jutile + (size_t)Bc * L) + 15) & ~(uintptr_t)15);  // [nslot] jobs for vb_slow_kernel
    const bool lazy = ctx->vb_lazy;
    float4* const posc = lazy ? nullptr : (float4*)ctx->vb_posc.ptr;
    const int nvb = lazy ? 0 : (std::max(V, 1) + 255) / 256;  // (no per-vertex work items where clip-space vertices are computed on demand)
    const int nitems = nvb + (NC + 3) / 4;
    for (int b0 = 0; b0 < B; b0 += Bc) {  // chunks of views (one, unless views x links / scratch say otherwise)
        const int Bk = std::min(Bc, B - b0);
        const bool first_chunk = b0 == 0, last_chunk = b0 + Bk == B;
        const bool time_it = ev && last_chunk;  // (stage times of a multi-chunk call: those of its last chunk)
        long long* facc = facc_all + (size_t)b0 * acc_stride;
        int* lbox = lbox_all + (size_t)VB_LBOX_STRIDE * b0 * L;
        float* mvp_k = mvp + (size_t)b0 * L * 16;
        const float* ref_k = ref + (size_t)b0 * H * W;
        float* mask_k = mask ? mask + (size_t)b0 * H * W : nullptr;
        VbRecs recs;
        recs.n = (size_t)Bk * NC1 * 64;
        recs.trec = (int4*)ctx->vb_boxes.ptr;
        recs.tbox = (uint2*)(recs.trec + recs.n * 2);
        recs.cbox = recs.tbox + recs.n;
        // stage 0: [pose forward] + vertices + screen boxes
        // workgroups per view: as many as stay resident together (5 per CU), every one with the same number of work items
        const int per_view_cap = std::max(8, (ctx->num_cus * std::max(1, vertex_grid)) / std::max(Bk, 1));
        const int items_per_wg = (nitems + per_view_cap - 1) / per_view_cap;
        const int gx = std::max(1, (nitems + items_per_wg - 1) / std::max(items_per_wg, 1));
        const int xcd_views = (xcd_align && (Bk % 8) == 0) ? Bk / 8 : 0;
        const dim3 vgrid(gx * Bk);
        const int nacc_ints = 2 * Bk * acc_stride;
        const int role = first_chunk ? 1 : 2;
        if (head) {
            StepHead hk = *head;
            hk.link_poses = head->link_poses + (size_t)b0 * L * 16;
            vb_vertex_kernel<true><<<vgrid, 256, 0, stream>>>(verts, vert_link, tris, cl, hk, mvp_k, V, nvb, g, posc, recs, lbox,
                                                             (int*)facc, nacc_ints, meta, Bk, gx, xcd_views, hv, role);
        } else {
            StepHead none = {};
            vb_vertex_kernel<false><<<vgrid, 256, 0, stream>>>(verts, vert_link, tris, cl, none, mvp_k, V, nvb, g, posc, recs,
                                                              lbox, (int*)facc, nacc_ints, meta, Bk, gx, xcd_views, hv, role);
        }
        EHR_LAUNCH_CHECK();
        if (time_it) EHR_HIP(hipEventRecord(ev[1], stream));
        // stage 1: jobs = (view, link, tile) -> coverage and the triangle ids the silhouette analysis will ask for
        const int job_wgs = ((ctx->num_cus * std::max(1, job_grid)) + 7) & ~7;
        // stage 1a (below) is launched by the stateless render call always, by the solver step only once a step needed it
        const bool with_slow = !tail || ctx->vb_slow_needed;
        VbResolveArgs rq;  // the resolve stage runs inside the job kernel, on the wave that drew the job
        rq.verts = verts;
        rq.mvp = mvp_k;
        rq.posc = posc;
        rq.L = L;
        rq.tri4 = (const int4*)ctx->vb_idx.ptr;
        rq.opp4 = (const int4*)ctx->vb_idx.ptr + T;
        rq.jn = jn;
        rq.jval = jval;
        rq.jitems = jitems;
        rq.jspill = jspill;
        rq.spill = spill;
        rq.meta = meta;
        rq.V = V;
        rq.T = T;
        rq.W = W;
        rq.H = H;
        rq.spill_cap = ctx->vb_spill_cap;
        rq.want_grad = grad_mvp ? 1 : 0;
        rq.dbg = dbg;
        VbJobParams jp;
        jp.g = g;
        jp.B = Bk;
        jp.cl = cl;
        jp.rc = recs;
        jp.lbox = lbox;
        jp.jn = jn;
        jp.jid = jid;
        jp.jdesc = jdesc;
        jp.jbase = jbase;
        jp.jutile = jutile;
        jp.jcap = ctx->vb_jcap;
        jp.meta = meta;
        jp.dbg = dbg;
        jp.hv = hv;
        jp.timeline = (long long*)ctx->vb_spill.ptr;
        jp.verts = verts;
        jp.mvp = mvp_k;
        jp.posc = posc;
        jp.V = V;
        jp.si = si;
        jp.jcov = jcov;
        jp.slow_list = with_slow ? slow_list : nullptr;
        jp.heavy_t = heavy_t;
        jp.med_t0 = med_t;
        jp.rq = rq;
        // Merged form (round 6, EHR_VB_MERGE=1; OFF by default): the composite + finish stages at the end of the job kernel's
        // launch, for the solver step's default chain -- bound reference, no mask output, one chunk, general-triangle pass off.
        // Measured (profiles/r06_merged_composite_ab.txt): the same step time as the launch of its own at 8 views (72.2 us both),
        // 3 % slower at 64 views -- what follows the last job is a tile's chain of dependent round trips, two levels of arrival
        // tickets and the finish stage's own chain, none of which a launch boundary adds to; and the merged grid has fewer waves
        // than there are items.  Kept as the A/B record and as the carrier of the grid-wide hand-over (agent-scope job slots).
        static const int merge_env = getenv("EHR_VB_MERGE") ? atoi(getenv("EHR_VB_MERGE")) : 0;
        const bool merged = merge_env && !lazy && tail && sparse && !mask_k && Bk == B && !with_slow && (job_wgs & 7) == 0;
        if (merged) {
            VbCompArgs& C = jp.ca;
            C.g = g; C.B = Bk; C.mvp = mvp_k; C.V = V; C.verts = verts; C.jn = jn; C.jval = jval; C.jitems = jitems;
            C.jspill = jspill; C.jcap = ctx->vb_jcap; C.ref = ref_k; C.mask = nullptr; C.facc = facc; C.nls = VB_LOSS_SLOTS;
            C.want_grad = grad_mvp ? 1 : 0; C.vec_ok = vec_ok; C.spill = spill; C.spill_cap = ctx->vb_spill_cap; C.meta = meta;
            C.dbg = dbg; C.tsum = tsum_all;
            jp.vtot = vtot_all;
            jp.ref_flag = ref_flag;
            jp.loss = loss;
            jp.grad_mvp = grad_mvp;
            jp.lbox_all = lbox_all;
            jp.tail = *tail;
            vb_job_kernel<false, true><<<job_wgs, 256, 0, stream>>>(jp);
            EHR_LAUNCH_CHECK();
            if (time_it)
                for (int k = 2; k <= 4; k++) EHR_HIP(hipEventRecord(ev[k], stream));
            continue;
        }
        if (lazy)
            vb_job_kernel<false, false, true><<<job_wgs, 256, 0, stream>>>(jp);
        else
            vb_job_kernel<false><<<job_wgs, 256, 0, stream>>>(jp);
        EHR_LAUNCH_CHECK();
        // stage 1a: jobs with a triangle that crosses the near plane or spans > 512 pixels (normally none: the kernel returns at once)
        static const int slow_grid = getenv("EHR_VB_SLOW_GRID") ? atoi(getenv("EHR_VB_SLOW_GRID")) : 32;  // tuning knob
        if (with_slow) {
            vb_slow_kernel<<<std::max(1, slow_grid), 256, 0, stream>>>(g, cl, recs, verts, mvp_k, posc, V, si, jid, jcov, jdesc, jn, slow_list, meta);
            EHR_LAUNCH_CHECK();
        }
        if (time_it) EHR_HIP(hipEventRecord(ev[2], stream));
        // stage 1b: drawn jobs -> per-link values and blended pairs.  The job kernel has done that for the jobs it drew itself;
        // only the jobs vb_slow_kernel redrew are left (none, normally: a launch of 32 workgroups that read a counter)
#if VB_INLINE_RESOLVE
        if (with_slow) {
            vb_resolve_kernel<<<std::max(8, slow_grid), 256, 0, stream>>>(g, Bk, verts, mvp_k, posc, V, T, rq.tri4, rq.opp4, jid, jcov, jdesc, jn, jval,
                                                                          jitems, jspill, ctx->vb_jcap, rq.want_grad, spill,
                                                                          ctx->vb_spill_cap, meta, dbg, slow_list);
            EHR_LAUNCH_CHECK();
        }
        (void)res_grid;
#else
        const int res_wgs = ((ctx->num_cus * std::max(1, res_grid)) + 7) & ~7;
        vb_resolve_kernel<<<res_wgs, 256, 0, stream>>>(g, Bk, verts, mvp_k, posc, V, T, (const int4*)ctx->vb_idx.ptr,
                                                       (const int4*)ctx->vb_idx.ptr + T, jid, jcov, jdesc, jn, jval, jitems, jspill,
                                                       ctx->vb_jcap, grad_mvp ? 1 : 0, spill, ctx->vb_spill_cap, meta, dbg, nullptr);
        EHR_LAUNCH_CHECK();
#endif
        if (time_it) {
            for (int k = 3; k <= 4; k++) EHR_HIP(hipEventRecord(ev[k], stream));
        }
        // stage 2: composite, loss, mask, backward.  In the call's last chunk its last-arriving workgroup runs the finish
        // stage over ALL views (accumulators -> loss / grad_mvp, + pose backward and Adam in the solver-step form; re-arms
        // the link boxes).  With a bound reference mask and no mask output only tiles that hold a job are visited.
        int nwg = ctx->num_cus * std::max(1, comp_grid);
        if (!sparse) nwg = std::min(nwg, (Bk * g.nt + 3) / 4);
        nwg = std::max(8, (nwg + 7) & ~7);  // a multiple of 8: the XCD split and the two-level arrival ticket rely on it
        const long long* tsum = sparse ? tsum_all + (size_t)b0 * g.nt : nullptr;
        const long long* vtot = sparse ? vtot_all + b0 : nullptr;
        const size_t dyn = (2 * (size_t)Bk * L + 1) * sizeof(int);  // link tables in LDS
        const bool fill = sparse && mask_k != nullptr;  // (FILL: the zero fill of the tiles no job owns, a variant of its own
                                                          //  so that the form without a mask output keeps its registers)
#define VB_COMPOSITE(TAILV, FILLV, tailarg)                                                                                  \
    vb_composite_kernel<TAILV, FILLV><<<nwg, 256, dyn, stream>>>(                                                            \
        g, Bk, mvp_k, V, verts, lbox, jn, jval, jitems, jspill, jbase, jutile, ctx->vb_jcap, ref_k, mask_k, facc,            \
        VB_LOSS_SLOTS, grad_mvp ? 1 : 0, vec_ok, spill, ctx->vb_spill_cap, meta, dbg, tsum, vtot, ref_flag, loss, grad_mvp,  \
        tailarg, last_chunk ? 1 : 0, B, facc_all, vtot_all, lbox_all)
        StepTail none = {};
        if (tail) {
            if (fill) VB_COMPOSITE(true, true, *tail); else VB_COMPOSITE(true, false, *tail);
        } else {
            if (fill) VB_COMPOSITE(false, true, none); else VB_COMPOSITE(false, false, none);
        }
#undef VB_COMPOSITE
        EHR_LAUNCH_CHECK();
    }
    if (ev) {
        for (int k = 5; k <= 7; k++) EHR_HIP(hipEventRecord(ev[k], stream));
    }
    return EHR_OK;
}


// Binds a reference mask to the plan (ehr_fused_bind_ref): one pass stores per (view, tile) the fixed-point sum(ref^2)
// exactly as the composite kernel would add it for a tile no link touches, and per view the total.  ref == NULL unbinds.
// The scoring op (ehr_mask_variance, csrc/ehr_score.hip) on the solver's machinery: the static cluster index of the packed
// mesh, then per chunk of candidates the vertex kernel (records + link boxes; depth class "coverage decides": every
// coverable pixel of the triangle has z/w in (0, 1]), the job kernel in its coverage-only form (no deferred units, no
// depth, no slots: a job ORs its tile's coverage into the view's word) and the count kernel.  *handled = 0 if the call
// cannot take this road (then nothing the caller sees was touched beyond `score` / `count`, which it rewrites): links
// not grouped, too many views per candidate for one chunk, or -- known only afterwards -- a triangle whose depth class
// needs the exact z-buffer (geometry within two near-plane distances of the camera, or crossing the far plane).
int ehr::vbuf_score(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* vert_link, const float* mvp, int Q,
                    int S, int L, int V, int T, int H, int W, long long* score, unsigned char* count, hipStream_t stream,
                    int* handled) {
    *handled = 0;
    // EHR_SCORE_PATH (tests, A/B): "tile" = always the per-triangle queue path of ehr_score.hip, "chain" = this one or an error
    const char* const want = getenv("EHR_SCORE_PATH");
    const bool must = want && !strcmp(want, "chain");
    if (want && !strcmp(want, "tile")) return EHR_OK;
    if (!vert_link || L > 32 || S * L > VB_MAX_UNITS || H > 32760 || W > 32736)
        return must ? fail(EHR_ERR_INVALID, "ehr_mask_variance: EHR_SCORE_PATH=chain, but the call's sizes do not allow it") : EHR_OK;
    int rc;
    unsigned long long hash = 0;
    {
        if ((rc = ctx->sc_misc.reserve(4096))) return rc;
        unsigned long long* dh = (unsigned long long*)ctx->sc_misc.ptr;
        EHR_HIP(hipMemsetAsync(dh, 0, sizeof(unsigned long long), stream));
        vb_hash_kernel<<<64, 256, 0, stream>>>((const unsigned*)verts, (size_t)3 * V, (const unsigned*)tris, (size_t)3 * T,
                                                (const unsigned*)vert_link, (size_t)V, dh);
        EHR_LAUNCH_CHECK();
        EHR_HIP(hipMemcpyAsync(ctx->host_pinned, dh, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        EHR_HIP(hipStreamSynchronize(stream));
        hash = *(unsigned long long*)ctx->host_pinned;
    }
    if (ctx->sc_key[0] != verts || ctx->sc_key[1] != tris || ctx->sc_key[2] != vert_link || ctx->sc_key_n[0] != V ||
        ctx->sc_key_n[1] != T || ctx->sc_key_n[2] != L || ctx->sc_hash != hash) {
        bool mixed = false;
        ctx->sc_key[0] = nullptr;
        if ((rc = vb_build_clusters(ctx->sc_clus, ctx->sc_nc, L, V, T, verts, tris, nullptr, vert_link, &mixed))) {
            if (rc == EHR_ERR_INVALID) return EHR_OK;  // (a link with too many triangles for a slot id: the other path takes it)
            return rc;
        }
        ctx->sc_mixed = mixed;
        ctx->sc_key[0] = verts;
        ctx->sc_key[1] = tris;
        ctx->sc_key[2] = vert_link;
        ctx->sc_key_n[0] = V;
        ctx->sc_key_n[1] = T;
        ctx->sc_key_n[2] = L;
        ctx->sc_hash = hash;
    }
    if (ctx->sc_mixed || ctx->sc_nc <= 0)
        return must ? fail(EHR_ERR_INVALID, "ehr_mask_variance: EHR_SCORE_PATH=chain, but the mesh's links are not grouped") : EHR_OK;
    const BinGeom g = make_geom(H, W, L);
    const int NC = ctx->sc_nc;
    VbClusters cl;
    cl.ctri = (const int32_t*)ctx->sc_clus.ptr;
    cl.clink = cl.ctri + (size_t)NC * 64;
    cl.coff = cl.clink + NC;
    cl.laabb = (const float*)(cl.coff + L + 1);
    cl.cvert = (const float4*)(((uintptr_t)(cl.laabb + 6 * (size_t)L) + 15) & ~(uintptr_t)15);
    cl.NC = NC;
    VbSlotIdx si;
    si.cvidx = (const int4*)(cl.cvert + 3 * (size_t)NC * 64);
    // candidates per chunk: (view, link) units of a chunk fit the job kernel's LDS tables; scratch bounded like the solver's
    const double view_bytes = (double)NC * (64 * 40 + 8) + (double)V * 16 + (double)g.nt * 32;
    int Qc = std::max(1, (VB_MAX_UNITS / L) / S);
    Qc = std::max(1, std::min(Qc, (int)(8192.0 * 1048576.0 / (view_bytes * S))));
    Qc = std::min(Qc, Q);
    const int Bc = Qc * S;
    const size_t n_hv = 8 + 2 * (size_t)VB_HEAVY_CAP + 2 * (size_t)VB_MED_CAP;
    const size_t misc_ints = (size_t)VB_LBOX_STRIDE * Bc * L + EHR_META_INTS + (VB_LINES + 2) * 32 + n_hv + 16 + 64 + 2 * 32 * (size_t)Qc;
    if ((rc = ctx->sc_posc.reserve((size_t)Bc * V * sizeof(float4)))) return rc;
    if ((rc = ctx->sc_entries.reserve((size_t)Bc * NC * (64 * 40 + 8)))) return rc;
    if ((rc = ctx->sc_misc.reserve(misc_ints * sizeof(int) + 64 + (size_t)Bc * g.nt * 4 * sizeof(u64)))) return rc;
    ctx->sc_entries_cap = 0;  // (the per-triangle path sizes its queues again if it runs after this)
    int* lbox = (int*)ctx->sc_misc.ptr;
    int* meta = lbox + (size_t)VB_LBOX_STRIDE * Bc * L;
    int* hvp = meta + EHR_META_INTS + (VB_LINES + 2) * 32;
    int* sticky = hvp + n_hv;
    unsigned long long* sacc0 = (unsigned long long*)(((uintptr_t)(sticky + 16) + 127) & ~(uintptr_t)127);  // [2][Qc][16]
    u64* tcov = (u64*)(sacc0 + 2 * 16 * (size_t)Qc);
    float4* posc = (float4*)ctx->sc_posc.ptr;
    VbHeavy hv;
    hv.gen = hvp;
    hv.list = hv.gen + 8;
    hv.mlist = hv.list + 2 * VB_HEAVY_CAP;
    hv.stamp = hv.gen;  // (never touched: the hint is off)
    hv.mcap = 0;
    hv.heavy_base = 0;
    hv.heavy_max = 0;
    EHR_HIP(hipMemsetAsync(hvp, 0, (n_hv + 16) * sizeof(int), stream));
    EHR_HIP(hipMemsetAsync(sacc0, 0, 2 * 16 * (size_t)Qc * sizeof(unsigned long long), stream));
    EHR_HIP(hipMemsetAsync(score, 0, (size_t)Q * sizeof(long long), stream));
    if (count) EHR_HIP(hipMemsetAsync(count, 0, (size_t)Q * H * W, stream));
    static const int vertex_grid = getenv("EHR_VB_VERTEX_GRID") ? atoi(getenv("EHR_VB_VERTEX_GRID")) : 5;
    static const int job_grid = getenv("EHR_VB_JOB_GRID") ? atoi(getenv("EHR_VB_JOB_GRID")) : 4;
    const int nvb = (std::max(V, 1) + 255) / 256;
    const int nitems = nvb + (NC + 3) / 4;
    const int job_wgs = ((ctx->num_cus * std::max(1, job_grid)) + 7) & ~7;
    // link boxes start empty (afterwards the count kernel re-arms them)
    vb_score_count_kernel<<<64, 256, 0, stream>>>(g, 0, S, tcov, sacc0, nullptr, lbox, VB_LBOX_STRIDE * Bc * L, sacc0, score, 0);
    EHR_LAUNCH_CHECK();
    unsigned long long* prev_sacc = sacc0;
    long long* prev_score = score;
    int prev_nq = 0, flip = 0;
    for (int q0 = 0; q0 < Q; q0 += Qc) {
        const int nq = std::min(Qc, Q - q0), Bk = nq * S;
        VbRecs recs;
        recs.n = (size_t)Bk * NC * 64;
        recs.trec = (int4*)ctx->sc_entries.ptr;
        recs.tbox = (uint2*)(recs.trec + recs.n * 2);
        recs.cbox = recs.tbox + recs.n;
        const int per_view_cap = std::max(8, (ctx->num_cus * std::max(1, vertex_grid)) / std::max(Bk, 1));
        const int items_per_wg = (nitems + per_view_cap - 1) / per_view_cap;
        const int gx = std::max(1, (nitems + items_per_wg - 1) / std::max(items_per_wg, 1));
        const int xcd_views = ((Bk % 8) == 0) ? Bk / 8 : 0;
        StepHead none = {};
        vb_vertex_kernel<false><<<dim3(gx * Bk), 256, 0, stream>>>(verts, vert_link, tris, cl, none, const_cast<float*>(mvp) + (size_t)q0 * S * L * 16,
                                                                V, nvb, g, posc, recs, lbox, (int*)tcov, Bk * g.nt * 8, meta, Bk, gx,
                                                                xcd_views, hv, 1 | 4);
        EHR_LAUNCH_CHECK();
        VbJobParams jp = {};
        jp.g = g;
        jp.B = Bk;
        jp.cl = cl;
        jp.rc = recs;
        jp.lbox = lbox;
        jp.jn = sticky;
        jp.jcap = S;
        jp.meta = meta;
        jp.dbg = 64 | 128;
        jp.hv = hv;
        jp.verts = verts;
        jp.mvp = mvp + (size_t)q0 * S * L * 16;
        jp.posc = posc;
        jp.V = V;
        jp.si = si;
        jp.jcov = tcov;
        jp.heavy_t = 0x7fffffff;
        jp.med_t0 = 0x7fffffff;
        vb_job_kernel<true><<<job_wgs, 256, 0, stream>>>(jp);
        EHR_LAUNCH_CHECK();
        static const int count_grid = getenv("EHR_SCORE_COUNT_GRID") ? atoi(getenv("EHR_SCORE_COUNT_GRID")) : 4;  // tuning knob
        unsigned long long* const sacc = sacc0 + (size_t)flip * 16 * Qc;
        vb_score_count_kernel<<<std::min((nq * g.nt + 3) / 4, std::max(1, count_grid) * ctx->num_cus), 256, 0, stream>>>(
            g, nq, S, tcov, sacc, count ? count + (size_t)q0 * H * W : nullptr, lbox, VB_LBOX_STRIDE * Bc * L, prev_sacc, prev_score, prev_nq);
        EHR_LAUNCH_CHECK();
        prev_sacc = sacc;
        prev_score = score + q0;
        prev_nq = nq;
        flip ^= 1;
    }
    vb_score_count_kernel<<<1, 256, 0, stream>>>(g, 0, S, tcov, sacc0, nullptr, lbox, 0, prev_sacc, prev_score, prev_nq);
    EHR_LAUNCH_CHECK();
    EHR_HIP(hipMemcpyAsync(ctx->host_pinned, sticky, sizeof(int), hipMemcpyDeviceToHost, stream));
    EHR_HIP(hipMemcpyAsync(ctx->host_pinned + 1, meta + EHR_META_OVERFLOW, sizeof(int), hipMemcpyDeviceToHost, stream));
    EHR_HIP(hipStreamSynchronize(stream));
    if (ctx->host_pinned[0] || ctx->host_pinned[1])  // a triangle needs the exact z-buffer: the other path redoes the call
        return must ? fail(EHR_ERR_INVALID, "ehr_mask_variance: EHR_SCORE_PATH=chain, but a triangle needs the exact z-buffer") : EHR_OK;
    *handled = 1;
    return EHR_OK;
}

int ehr::vbuf_bind_ref(ehr_ctx* ctx, const float* ref, hipStream_t stream) {
    ctx->vb_ref = nullptr;
    if (!ref) return EHR_OK;
    const int B = ctx->pB, H = ctx->pH, W = ctx->pW;
    BinGeom g = make_geom(H, W, ctx->pL);
    long long* tsum = (long long*)ctx->vb_refsum.ptr;
    long long* vtot = tsum + (size_t)B * g.nt;
    EHR_HIP(hipMemsetAsync(vtot, 0, ((size_t)B + 1) * sizeof(long long), stream));
    const int vec_ok = ((W & 3) == 0) && (((uintptr_t)ref & 15) == 0);
    const int nw = B * g.nt;
    vb_refsum_kernel<<<(nw + 3) / 4, 256, 0, stream>>>(g, B, ref, vec_ok, tsum, vtot, (int*)(vtot + B));
    EHR_LAUNCH_CHECK();
    ctx->vb_ref = ref;
    return EHR_OK;
}
