// ehr_interp_aa.hip -- drop-in dr.interpolate (nvdiffrast_renderer.py:42) and dr.antialias (:43), fwd + bwd,
// plus the edge-topology build that replaces dr.antialias_construct_topology_hash.
//
// PROVENANCE: the per-pair arithmetic is ehr_device.h's aa_analyze / aa_pos_grad (see the provenance note there: written
// from knowledge of nvdiffrast's antialias.cu, which is under the NVIDIA Source Code License; not present in
// /root/reference), and the int4 work items handed from the forward to the backward pass follow that file's work buffer.
// The kernel structure is this repo's own since round 4: nvdiffrast (and rounds 1-3 here) discover the pixel pairs, analyse
// them and scatter the blends with float atomics in three launches; here ONE launch per 32 x 8 tile lists the pairs in LDS,
// analyses them one per lane and lets every pixel gather its blends in a serial sweep's order (no atomics, bit-exact against
// the oracle).  The topology is a sorted edge table, not nvdiffrast's hash.
#include <algorithm>

#include "ehr_device.h"
#include "ehr_host.h"

namespace ehr {

// ---- interpolate -------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) interp_fwd_kernel(const float* __restrict__ attr, const float4* __restrict__ rast,
                                                         const int32_t* __restrict__ tri, int B, int Ba, int V, int T,
                                                         int A, size_t P, float* __restrict__ out, int H, int W,
                                                         const unsigned char* __restrict__ flags) {
    // one workgroup per (image, 32 x 8 tile): grid (tiles in x, tiles in y, images) -- a pixel's coordinates without a
    // division (the linear form spent ~150 instructions per pixel on a 64-bit and a 32-bit one before it knew its tile)
    const int ix = (int)blockIdx.x * EHR_FLAG_TW + ((int)threadIdx.x & (EHR_FLAG_TW - 1));
    const int iy = (int)blockIdx.y * EHR_FLAG_TH + ((int)threadIdx.x / EHR_FLAG_TW);
    const int b = (int)blockIdx.z;
    if (ix >= W || iy >= H) return;
    const size_t pix = (size_t)b * P + (size_t)iy * W + ix;
    float* o = out + pix * A;
    if (flags && !flags[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x]) {  // a tile nothing was drawn into:
        for (int k = 0; k < A; k++) o[k] = 0.f;                                              // zeros, and `rast` is not read
        return;
    }
    float4 r = rast[pix];
    int t = float_to_tri(r.w) - 1;
    bool ok = t >= 0 && t < T;
    int v0 = 0, v1 = 0, v2 = 0;
    if (ok) {
        v0 = tri[3 * t];
        v1 = tri[3 * t + 1];
        v2 = tri[3 * t + 2];
        ok = (unsigned)v0 < (unsigned)V && (unsigned)v1 < (unsigned)V && (unsigned)v2 < (unsigned)V;
    }
    if (!ok) {
        for (int k = 0; k < A; k++) o[k] = 0.f;
        return;
    }
    const float* ab = attr + (Ba == 1 ? 0 : (size_t)b * V * A);
    float b0 = r.x, b1 = r.y;
    float b2 = (1.f - b0) - b1;
    for (int k = 0; k < A; k++)
        o[k] = fmaf(b2, ab[(size_t)v2 * A + k], fmaf(b1, ab[(size_t)v1 * A + k], b0 * ab[(size_t)v0 * A + k]));
}

__global__ void __launch_bounds__(256) interp_grad_kernel(const float* __restrict__ attr, const float4* __restrict__ rast,
                                                          const int32_t* __restrict__ tri, const float* __restrict__ dy,
                                                          int B, int Ba, int V, int T, int A, size_t P,
                                                          float* __restrict__ grad_attr, float4* __restrict__ grad_rast, int H,
                                                          int W, const unsigned char* __restrict__ flags) {
    const int ix = (int)blockIdx.x * EHR_FLAG_TW + ((int)threadIdx.x & (EHR_FLAG_TW - 1));  // (grid as interp_fwd_kernel's)
    const int iy = (int)blockIdx.y * EHR_FLAG_TH + ((int)threadIdx.x / EHR_FLAG_TW);
    const int b = (int)blockIdx.z;
    if (ix >= W || iy >= H) return;
    const size_t pix = (size_t)b * P + (size_t)iy * W + ix;
    if (flags && !flags[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x]) {  // a tile nothing was drawn into:
        grad_rast[pix] = make_float4(0.f, 0.f, 0.f, 0.f);                                    // no gradient, nothing is read
        return;
    }
    float4 r = rast[pix];
    int t = float_to_tri(r.w) - 1;
    float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0 && t < T) {
        int v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2];
        if ((unsigned)v0 < (unsigned)V && (unsigned)v1 < (unsigned)V && (unsigned)v2 < (unsigned)V) {
            size_t aoff = (Ba == 1 ? 0 : (size_t)b * V * A);
            float b0 = r.x, b1 = r.y;
            float b2 = (1.f - b0) - b1;
            float g0 = 0.f, g1 = 0.f;
            for (int k = 0; k < A; k++) {
                float d = dy[pix * A + k];
                float a0 = attr[aoff + (size_t)v0 * A + k], a1 = attr[aoff + (size_t)v1 * A + k],
                      a2 = attr[aoff + (size_t)v2 * A + k];
                if (grad_attr && d != 0.f) {  // grad_attr == NULL: the attributes are constants (EasyHeC's all-ones colours)
                    atomicAdd(&grad_attr[aoff + (size_t)v0 * A + k], b0 * d);
                    atomicAdd(&grad_attr[aoff + (size_t)v1 * A + k], b1 * d);
                    atomicAdd(&grad_attr[aoff + (size_t)v2 * A + k], b2 * d);
                }
                g0 += d * (a0 - a2);
                g1 += d * (a1 - a2);
            }
            gr.x = g0;
            gr.y = g1;
        }
    }
    grad_rast[pix] = gr;
}

// ---- topology ----------------------------------------------------------------------------------------------------
// Open-addressing hash of undirected edges.  Each slot keeps the two smallest record ids (3*t + k) that contain
// the edge, so the result equals "the first two triangles in index order" regardless of insertion order.

struct EdgeSlot {
    u64 key;        // (min(va,vb) << 32 | max(va,vb)) + 1, 0 = empty
    unsigned r0;    // smallest record id
    unsigned r1;    // second smallest record id
};

__device__ __forceinline__ unsigned edge_hash(u64 k, unsigned mask) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (unsigned)k & mask;
}

__device__ __forceinline__ u64 edge_key(const int32_t* tri, int t, int k, int& vopp) {
    int v[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    int va = v[(k + 1) % 3], vb = v[(k + 2) % 3];
    vopp = v[k];
    if (va == vb) return 0;
    unsigned lo = (unsigned)min(va, vb), hi = (unsigned)max(va, vb);
    return (((u64)lo << 32) | hi) + 1;
}

// pass 0: claim slot + r0 = min record; pass 1: r1 = min record != r0; pass 2: resolve opp
__global__ void __launch_bounds__(256) topo_kernel(const int32_t* __restrict__ tri, int T, EdgeSlot* __restrict__ tab,
                                                   unsigned mask, int pass, int32_t* __restrict__ opp) {
    int rec = blockIdx.x * blockDim.x + threadIdx.x;
    if (rec >= 3 * T) return;
    int t = rec / 3, k = rec - 3 * t;
    int vopp;
    u64 key = edge_key(tri, t, k, vopp);
    if (key == 0) {
        if (pass == 2) opp[rec] = -1;
        return;
    }
    unsigned h = edge_hash(key, mask);
    for (;;) {
        u64 cur = tab[h].key;
        if (cur == 0 && pass == 0) {
            u64 prev = atomicCAS(&tab[h].key, 0ull, key);
            cur = (prev == 0) ? key : prev;
        }
        if (cur == key) break;
        if (cur == 0) {  // cannot happen after pass 0; never spin forever
            if (pass == 2) opp[rec] = -1;
            return;
        }
        h = (h + 1) & mask;
    }
    if (pass == 0) {
        atomicMin(&tab[h].r0, (unsigned)rec);
    } else if (pass == 1) {
        if (tab[h].r0 != (unsigned)rec) atomicMin(&tab[h].r1, (unsigned)rec);
    } else {
        unsigned r0 = tab[h].r0, r1 = tab[h].r1;
        int s0 = tri[r0], s1 = (r1 != 0xffffffffu) ? tri[r1] : -1;  // tri[3t+k] is the record's opposite vertex
        int r = -1;
        if (s0 == vopp)
            r = s1;
        else if (s1 == vopp)
            r = s0;
        opp[rec] = r;
    }
}

__global__ void topo_init_kernel(EdgeSlot* tab, unsigned n) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        tab[i].key = 0;
        tab[i].r0 = 0xffffffffu;
        tab[i].r1 = 0xffffffffu;
    }
}

// ---- attribute pixel differentials (dr.interpolate(..., rast_db=, diff_attrs=); not on EasyHeC's path) -----------------
// out_da [B,H,W,2D]: (d attr_j / dX, d attr_j / dY) for j = diff_idx[i] (NULL = all attributes) at channels 2i, 2i+1:
// attr = u a0 + v a1 + (1-u-v) a2  =>  d attr / dX = du/dX (a0 - a2) + dv/dX (a1 - a2), with rast_db = (du/dX, du/dY,
// dv/dX, dv/dY).  Same products and sums as the oracle's ehro_interpolate_da_fwd.
__global__ void __launch_bounds__(256) interp_da_fwd_kernel(const float* __restrict__ attr, const float4* __restrict__ rast,
                                                            const float4* __restrict__ rast_db,
                                                            const int32_t* __restrict__ tri,
                                                            const int32_t* __restrict__ diff_idx, int B, int Ba, int V, int T,
                                                            int A, int D, size_t P, float* __restrict__ out_da) {
    size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= P * B) return;
    const int b = (int)(pix / P);
    const int t = float_to_tri(rast[pix].w) - 1;
    float* o = out_da + pix * 2 * D;
    bool ok = t >= 0 && t < T;
    int v0 = 0, v1 = 0, v2 = 0;
    if (ok) {
        v0 = tri[3 * t];
        v1 = tri[3 * t + 1];
        v2 = tri[3 * t + 2];
        ok = (unsigned)v0 < (unsigned)V && (unsigned)v1 < (unsigned)V && (unsigned)v2 < (unsigned)V;
    }
    const float4 db = ok ? rast_db[pix] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* ab = attr + (Ba == 1 ? 0 : (size_t)b * V * A);
    for (int k = 0; k < D; k++) {
        const int j = diff_idx ? diff_idx[k] : k;
        float ox = 0.f, oy = 0.f;
        if (ok && (unsigned)j < (unsigned)A) {
            const float a2 = ab[(size_t)v2 * A + j];
            const float d0 = ab[(size_t)v0 * A + j] - a2, d1 = ab[(size_t)v1 * A + j] - a2;
            const float mx0 = db.x * d0, mx1 = db.z * d1, my0 = db.y * d0, my1 = db.w * d1;
            ox = mx0 + mx1;
            oy = my0 + my1;
        }
        o[2 * k] = ox;
        o[2 * k + 1] = oy;
    }
}

__global__ void __launch_bounds__(256) interp_da_grad_kernel(const float* __restrict__ attr, const float4* __restrict__ rast,
                                                             const float4* __restrict__ rast_db,
                                                             const int32_t* __restrict__ tri,
                                                             const int32_t* __restrict__ diff_idx,
                                                             const float* __restrict__ dy_da, int B, int Ba, int V, int T,
                                                             int A, int D, size_t P, float* __restrict__ grad_attr,
                                                             float4* __restrict__ grad_db) {
    size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= P * B) return;
    const int b = (int)(pix / P);
    const int t = float_to_tri(rast[pix].w) - 1;
    bool ok = t >= 0 && t < T;
    int v0 = 0, v1 = 0, v2 = 0;
    if (ok) {
        v0 = tri[3 * t];
        v1 = tri[3 * t + 1];
        v2 = tri[3 * t + 2];
        ok = (unsigned)v0 < (unsigned)V && (unsigned)v1 < (unsigned)V && (unsigned)v2 < (unsigned)V;
    }
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        const float4 db = rast_db[pix];
        const size_t aoff = (Ba == 1 ? 0 : (size_t)b * V * A);
        for (int k = 0; k < D; k++) {
            const int j = diff_idx ? diff_idx[k] : k;
            if ((unsigned)j >= (unsigned)A) continue;
            const float gx = dy_da[pix * 2 * D + 2 * k], gy = dy_da[pix * 2 * D + 2 * k + 1];
            if (gx == 0.f && gy == 0.f) continue;
            const float a2 = attr[aoff + (size_t)v2 * A + j];
            const float d0 = attr[aoff + (size_t)v0 * A + j] - a2, d1 = attr[aoff + (size_t)v1 * A + j] - a2;
            g.x += gx * d0;
            g.y += gy * d0;
            g.z += gx * d1;
            g.w += gy * d1;
            if (grad_attr) {
                const float c0 = gx * db.x + gy * db.y, c1 = gx * db.z + gy * db.w;
                atomicAdd(&grad_attr[aoff + (size_t)v0 * A + j], c0);
                atomicAdd(&grad_attr[aoff + (size_t)v1 * A + j], c1);
                atomicAdd(&grad_attr[aoff + (size_t)v2 * A + j], -(c0 + c1));
            }
        }
    }
    if (grad_db) grad_db[pix] = g;
}

static unsigned topo_slots(int T) {
    unsigned need = (unsigned)std::max(3 * (size_t)std::max(T, 1) * 2, (size_t)64);
    unsigned n = 64;
    while (n < need) n <<= 1;
    return n;
}

// ---- antialias ---------------------------------------------------------------------------------------------------

// work buffer (forward -> backward): one SEGMENT per workgroup of the forward kernel (a 32 x 8 tile): int4 header
// {count, 0, 0, 0} followed by up to 512 int4 items {px, py, flags, alpha bits} -- the BLENDED pairs whose first pixel
// the workgroup owns.  Every workgroup writes its own header, so nothing has to be zeroed beforehand.
// flags: bits 0-1 di, bit 2 d (vertical pair), bit 3 chosen triangle is pixel1's, bits 16.. image
#define AA_FLAG_D 4
#define AA_FLAG_TRI1 8
constexpr int AA_SEG = 513;  // int4 per segment

// One pixel pair (pix0 = (px, py) of image b, pix1 = its right (d = 0) or upper (d = 1) neighbour) whose triangle ids
// differ: pick the nearer triangle, find the silhouette edge that crosses between the two pixel centres.
struct AAHit {
    bool found;
    float alpha;
    int flags;  // di | AA_FLAG_TRI1
};
__device__ __forceinline__ AAHit aa_pair(const float4* __restrict__ rast, const float4* __restrict__ pos,
                                         const int32_t* __restrict__ tri, const int32_t* __restrict__ opp, int range_mode,
                                         int V, int T, int H, int W, size_t P, int b, int px, int py, int d) {
    AAHit h;
    h.found = false;
    h.alpha = 0.f;
    h.flags = 0;
    const size_t pix0 = (size_t)b * P + (size_t)py * W + px;
    const size_t pix1 = pix0 + (d ? (size_t)W : 1);
    const float4 r0 = rast[pix0], r1 = rast[pix1];
    const int tri0 = float_to_tri(r0.w) - 1, tri1 = float_to_tri(r1.w) - 1;
    int t = (tri0 >= 0) ? tri0 : tri1;
    if (tri0 >= 0 && tri1 >= 0) t = (r0.z < r1.z) ? tri0 : tri1;
    const bool chose0 = !(t == tri1);
    int cx = px, cy = py;
    if (!chose0) {
        cx += 1 - d;
        cy += d;
    }
    if (t < 0 || t >= T) return h;
    // (the triangle's corners and its neighbours' opposite corners are requested together: both follow from t alone)
    const int vi[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    const int ov[3] = {opp[3 * t], opp[3 * t + 1], opp[3 * t + 2]};
    if ((unsigned)vi[0] >= (unsigned)V || (unsigned)vi[1] >= (unsigned)V || (unsigned)vi[2] >= (unsigned)V) return h;
    const float4* pb = pos + (range_mode ? 0 : (size_t)b * V);
    float4 p[3], o[3];
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = pb[vi[k]];
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] = ((unsigned)ov[k] < (unsigned)V) ? pb[ov[k]] : p[k];
    const AAPair a = aa_analyze(p, o, cx, cy, d, chose0, W, H);
    if (!a.found) return h;
    h.found = true;
    h.alpha = a.alpha;
    h.flags = a.di | (a.tri1 ? AA_FLAG_TRI1 : 0);
    return h;
}

// The whole forward pass in ONE launch, one workgroup per 32 x 8 tile, gathering instead of scattering.
//   1. every pixel looks at its right and upper neighbour (the tile's first column / row also at the left / lower one)
//      and lists the pairs whose triangle ids differ in LDS;
//   2. the pairs are analysed one per lane, all at once -- a pair is a chain of four dependent reads (ids -> triangle
//      -> vertices -> neighbour triangles' vertices), and a pixel that analysed its four pairs itself, one after the
//      other, held the launch for four such chains;
//   3. every pixel adds the blends of the (up to four) pairs it belongs to that land on it -- its own right / upper pair
//      when alpha > 0, its left / lower neighbour's pair when alpha <= 0 -- in the order a serial sweep over first
//      pixels would (no atomics: bit-reproducible, and equal to the oracle's sweep bit for bit).
// Pairs across a tile border are analysed by both tiles.  Round 3 had three launches here (zero the list's counter,
// discover the pairs + copy the colour, analyse and scatter with float atomics).
constexpr int AA_TW = 32, AA_TH = 8;
static_assert(AA_TW == EHR_FLAG_TW && AA_TH == EHR_FLAG_TH, "the tile flags of dr.rasterize are per antialias tile");
constexpr int AA_MAXP = 2 * AA_TW * AA_TH + AA_TW + AA_TH;  // own pairs + the halo pairs of the first column and row

__global__ void __launch_bounds__(256) aa_fwd_kernel(const float* __restrict__ color, const float4* __restrict__ rast,
                                                     const float4* __restrict__ pos, const int32_t* __restrict__ tri,
                                                     const int32_t* __restrict__ opp, int range_mode, int B, int V, int T,
                                                     int H, int W, int C, int ntx, int nty, float* __restrict__ out,
                                                     int4* __restrict__ work, const unsigned char* __restrict__ flags,
                                                     float4* __restrict__ zero, int zero_n) {
    __shared__ int s_npair, s_count;
    __shared__ unsigned s_pair[AA_MAXP];   // px - tx0 + 1 (6 bits) | (py - ty0 + 1) << 6 (4 bits) | d << 10
    __shared__ float s_alpha[AA_MAXP];     // 0 = nothing lands anywhere
    __shared__ short s_slot[4][AA_TW * AA_TH];  // per pixel: slot of its R, U, L, D pair or -1
    const int tid = threadIdx.x;
    // grid (tiles in x, tiles in y, images): no division between a workgroup and its tile
    const int ttx = (int)blockIdx.x, tty = (int)blockIdx.y, b = (int)blockIdx.z;
    const int tile = tty * ntx + ttx;
    const int segi = b * ntx * nty + tile;  // the tile's segment of `work`
    if (zero) {
        // the buffer the backward pass will accumulate pos's gradient into, cleared here: the fill a caller would launch
        // before ehr_antialias_grad (one per (view, link) image in EasyHeC's schedule) rides on a launch that exists anyway
        const int nwg = ntx * nty * B;
        for (int i = segi * 256 + tid; i < zero_n; i += nwg * 256) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int tx0 = ttx * AA_TW, ty0 = tty * AA_TH;
    const int lx = tid % AA_TW, ly = tid / AA_TW;
    const int px = tx0 + lx, py = ty0 + ly;
    const size_t P = (size_t)H * W;
    const bool in = px < W && py < H;
    const size_t idx = (size_t)b * P + (size_t)py * W + px;
    if (flags) {
        // Pairs need two different triangle ids: a tile that holds no triangle, and none of whose four neighbours does,
        // has no pair and is part of none (its first column / row looks at the left / lower tile, its last at the right /
        // upper one).  Its pixels keep their colour; `rast` is not read.  (workgroup-uniform: five bytes)
        const unsigned char* const f = flags + (size_t)b * ntx * nty;
        const bool any = f[tile] | (ttx > 0 ? f[tile - 1] : 0) | (ttx + 1 < ntx ? f[tile + 1] : 0) |
                         (tty > 0 ? f[tile - ntx] : 0) | (tty + 1 < nty ? f[tile + ntx] : 0);
        if (!any) {
            if (in)
                for (int k = 0; k < C; k++) out[idx * C + k] = color[idx * C + k];
            if (tid == 0) work[(size_t)segi * AA_SEG] = make_int4(0, 0, 0, 0);
            return;
        }
    }
    if (tid == 0) {
        s_npair = 0;
        s_count = 0;
    }
    __syncthreads();
    // ---- 1. list the pairs
    int sR = -1, sU = -1, sL = -1, sD = -1;
    if (in) {
        const float t0 = rast[idx].w;
        if (px + 1 < W && rast[idx + 1].w != t0) {
            sR = atomicAdd(&s_npair, 1);
            s_pair[sR] = (unsigned)(lx + 1) | ((unsigned)(ly + 1) << 6);
        }
        if (py + 1 < H && rast[idx + W].w != t0) {
            sU = atomicAdd(&s_npair, 1);
            s_pair[sU] = (unsigned)(lx + 1) | ((unsigned)(ly + 1) << 6) | (1u << 10);
        }
        if (lx == 0 && px > 0 && rast[idx - 1].w != t0) {  // the pair (left neighbour, this pixel) belongs to the tile on the left
            sL = atomicAdd(&s_npair, 1);
            s_pair[sL] = (unsigned)(lx + 0) | ((unsigned)(ly + 1) << 6);
        }
        if (ly == 0 && py > 0 && rast[idx - W].w != t0) {
            sD = atomicAdd(&s_npair, 1);
            s_pair[sD] = (unsigned)(lx + 1) | ((unsigned)(ly + 0) << 6) | (1u << 10);
        }
    }
    s_slot[0][tid] = (short)sR;
    s_slot[1][tid] = (short)sU;
    s_slot[2][tid] = (short)sL;
    s_slot[3][tid] = (short)sD;
    __syncthreads();
    // ---- 2. analyse them, one per lane
    const int np = s_npair;
    int4* const seg = work + (size_t)segi * AA_SEG;
    for (int i = tid; i < np; i += 256) {
        const unsigned pr = s_pair[i];
        const int qx = tx0 + (int)(pr & 63u) - 1, qy = ty0 + (int)((pr >> 6) & 15u) - 1, d = (int)(pr >> 10);
        const AAHit h = aa_pair(rast, pos, tri, opp, range_mode, V, T, H, W, P, b, qx, qy, d);
        const float al = (h.found) ? h.alpha : 0.f;
        s_alpha[i] = al;
        // the list for the backward pass holds every blended pair once: the tile that owns its first pixel records it
        if (al != 0.f && qx >= tx0 && qy >= ty0)
            seg[1 + atomicAdd(&s_count, 1)] = make_int4(qx, qy, (b << 16) | (d ? AA_FLAG_D : 0) | h.flags, __float_as_int(al));
    }
    __syncthreads();
    // ---- 3. gather
    if (in) {
        // slots of the four pairs: own R / U; the left neighbour's R pair (its slot, or this pixel's halo pair in the
        // first column); the lower neighbour's U pair likewise
        const int kR = sR, kU = sU;
        const int kL = (lx > 0) ? (int)s_slot[0][tid - 1] : sL;
        const int kD = (ly > 0) ? (int)s_slot[1][tid - AA_TW] : sD;
        float aR = (kR >= 0) ? s_alpha[kR] : 0.f, aU = (kU >= 0) ? s_alpha[kU] : 0.f;
        float aL = (kL >= 0) ? s_alpha[kL] : 0.f, aD = (kD >= 0) ? s_alpha[kD] : 0.f;
        if (!(aR > 0.f)) aR = 0.f;  // own pairs land here (pix0) when alpha > 0
        if (!(aU > 0.f)) aU = 0.f;
        if (aL > 0.f) aL = 0.f;     // neighbours' pairs land here (pix1) when alpha <= 0
        if (aD > 0.f) aD = 0.f;
        for (int k = 0; k < C; k++) {
            const float c = color[idx * C + k];
            float v = c;
            // every term is alpha (c1 - c0) of its pair: this pixel is pix1 of its neighbours' pairs, pix0 of its own.
            // Order = the pairs' order in a sweep over first pixels, horizontal before vertical (the oracle's, and a
            // serial implementation's): lower neighbour's vertical pair, left neighbour's horizontal one, own two.
            if (aD != 0.f) v += aD * (c - color[(idx - W) * C + k]);
            if (aL != 0.f) v += aL * (c - color[(idx - 1) * C + k]);
            if (aR != 0.f) v += aR * (color[(idx + 1) * C + k] - c);
            if (aU != 0.f) v += aU * (color[(idx + W) * C + k] - c);
            out[idx * C + k] = v;
        }
    }
    if (tid == 0) seg[0] = make_int4(s_count, 0, 0, 0);
}

// backward: one wave per segment of the forward pass's list
__global__ void __launch_bounds__(256) aa_grad_kernel(const float* __restrict__ color, const float4* __restrict__ rast,
                                                      const float4* __restrict__ pos, const int32_t* __restrict__ tri,
                                                      const float* __restrict__ dy, const int4* __restrict__ work, int nseg,
                                                      int range_mode, int V, int T, int H, int W, int C,
                                                      float* __restrict__ grad_color, float* __restrict__ grad_pos) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nseg) return;
    const int4* const seg = work + (size_t)s * AA_SEG;
    const int count = seg[0].x;
    size_t P = (size_t)H * W;
    for (int i = threadIdx.x & 63; i < count; i += 64) {
        int4 item = seg[1 + i];
        float alpha = __int_as_float(item.w);
        if (alpha == 0.f) continue;
        int px = item.x, py = item.y, b = item.z >> 16, d = (item.z & AA_FLAG_D) ? 1 : 0;
        int di = item.z & 3;
        bool tri1 = (item.z & AA_FLAG_TRI1) != 0;
        size_t pix0 = (size_t)b * P + (size_t)py * W + px;
        size_t pix1 = pix0 + (d ? (size_t)W : 1);
        const float* c0 = color + pix0 * C;
        const float* c1 = color + pix1 * C;
        const float* g = dy + (alpha > 0.f ? pix0 : pix1) * C;
        float dd = 0.f;
        for (int k = 0; k < C; k++) {
            float gk = g[k];
            if (gk != 0.f) {
                dd += gk * (c1[k] - c0[k]);
                if (grad_color) {
                    float v = alpha * gk;
                    atomicAdd(&grad_color[pix0 * C + k], -v);
                    atomicAdd(&grad_color[pix1 * C + k], v);
                }
            }
        }
        if (dd == 0.f) continue;
        int t = float_to_tri(rast[tri1 ? pix1 : pix0].w) - 1;
        if (t < 0 || t >= T) continue;
        int vi[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
        int i1 = (di < 2) ? di + 1 : 0;
        int i2 = (i1 < 2) ? i1 + 1 : 0;
        int v1 = vi[i1], v2 = vi[i2];
        if ((unsigned)v1 >= (unsigned)V || (unsigned)v2 >= (unsigned)V) continue;
        size_t voff = range_mode ? 0 : (size_t)b * V;
        int cx = px, cy = py;
        if (tri1) {
            cx += 1 - d;
            cy += d;
        }
        float g1[3], g2[3];
        aa_pos_grad(pos[voff + v1], pos[voff + v2], cx, cy, d, alpha, dd, W, H, g1, g2);
        float* gp = grad_pos + 4 * voff;
        atomicAdd(&gp[4 * v1 + 0], g1[0]); atomicAdd(&gp[4 * v1 + 1], g1[1]); atomicAdd(&gp[4 * v1 + 3], g1[2]);
        atomicAdd(&gp[4 * v2 + 0], g2[0]); atomicAdd(&gp[4 * v2 + 1], g2[1]); atomicAdd(&gp[4 * v2 + 3], g2[2]);
    }
}

}  // namespace ehr

using namespace ehr;

extern "C" {

int ehr_interpolate_fwd(const float* attr, const float* rast, const int32_t* tri, int B, int Ba, int V, int T, int A,
                        int H, int W, float* out, const unsigned char* tile_flags, void* stream_) {
    if (!attr || !rast || !tri || !out) return fail(EHR_ERR_INVALID, "ehr_interpolate_fwd: NULL tensor");
    if (Ba != 1 && Ba != B) return fail(EHR_ERR_INVALID, "ehr_interpolate_fwd: attr batch %d must be 1 or %d", Ba, B);
    if (B > 65535) return fail(EHR_ERR_INVALID, "ehr_interpolate_fwd: %d images in one call (the grid's z extent holds 65535): split the batch", B);
    size_t P = (size_t)H * W, n = P * B;
    if (n == 0) return EHR_OK;
    interp_fwd_kernel<<<dim3(flag_ntx(W), flag_nty(H), B), EHR_FLAG_TW * EHR_FLAG_TH, 0, (hipStream_t)stream_>>>(
        attr, (const float4*)rast, tri, B, Ba, V, T, A, P, out, H, W, tile_flags);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

int ehr_interpolate_grad(const float* attr, const float* rast, const int32_t* tri, const float* dy, int B, int Ba, int V,
                         int T, int A, int H, int W, float* grad_attr, float* grad_rast, const unsigned char* tile_flags,
                         void* stream_) {
    if (!attr || !rast || !tri || !dy || !grad_rast)
        return fail(EHR_ERR_INVALID, "ehr_interpolate_grad: NULL tensor");
    if (Ba != 1 && Ba != B) return fail(EHR_ERR_INVALID, "ehr_interpolate_grad: attr batch %d must be 1 or %d", Ba, B);
    if (B > 65535) return fail(EHR_ERR_INVALID, "ehr_interpolate_grad: %d images in one call (the grid's z extent holds 65535): split the batch", B);
    size_t P = (size_t)H * W, n = P * B;
    if (n == 0) return EHR_OK;
    interp_grad_kernel<<<dim3(flag_ntx(W), flag_nty(H), B), EHR_FLAG_TW * EHR_FLAG_TH, 0, (hipStream_t)stream_>>>(
        attr, (const float4*)rast, tri, dy, B, Ba, V, T, A, P, grad_attr, (float4*)grad_rast, H, W, tile_flags);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

int ehr_interpolate_da_fwd(const float* attr, const float* rast, const float* rast_db, const int32_t* tri,
                           const int32_t* diff_idx, int B, int Ba, int V, int T, int A, int D, int H, int W, float* out_da,
                           void* stream_) {
    if (!attr || !rast || !rast_db || !tri || !out_da) return fail(EHR_ERR_INVALID, "ehr_interpolate_da_fwd: NULL tensor");
    if (Ba != 1 && Ba != B) return fail(EHR_ERR_INVALID, "ehr_interpolate_da_fwd: attr batch %d must be 1 or %d", Ba, B);
    if (D < 0 || (!diff_idx && D != A)) return fail(EHR_ERR_INVALID, "ehr_interpolate_da_fwd: D must equal A when diff_idx is NULL");
    size_t P = (size_t)H * W, n = P * B;
    if (n == 0 || D == 0) return EHR_OK;
    interp_da_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream_>>>(
        attr, (const float4*)rast, (const float4*)rast_db, tri, diff_idx, B, Ba, V, T, A, D, P, out_da);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

int ehr_interpolate_da_grad(const float* attr, const float* rast, const float* rast_db, const int32_t* tri,
                            const int32_t* diff_idx, const float* dy_da, int B, int Ba, int V, int T, int A, int D, int H,
                            int W, float* grad_attr, float* grad_rast_db, void* stream_) {
    if (!attr || !rast || !rast_db || !tri || !dy_da) return fail(EHR_ERR_INVALID, "ehr_interpolate_da_grad: NULL tensor");
    if (Ba != 1 && Ba != B) return fail(EHR_ERR_INVALID, "ehr_interpolate_da_grad: attr batch %d must be 1 or %d", Ba, B);
    if (D < 0 || (!diff_idx && D != A)) return fail(EHR_ERR_INVALID, "ehr_interpolate_da_grad: D must equal A when diff_idx is NULL");
    size_t P = (size_t)H * W, n = P * B;
    if (n == 0 || (!grad_attr && !grad_rast_db)) return EHR_OK;
    interp_da_grad_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream_>>>(
        attr, (const float4*)rast, (const float4*)rast_db, tri, diff_idx, dy_da, B, Ba, V, T, A, D, P, grad_attr,
        (float4*)grad_rast_db);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

size_t ehr_topology_scratch_bytes(int T) { return (size_t)topo_slots(T) * sizeof(EdgeSlot); }

int ehr_antialias_topology(const int32_t* tri, int T, int32_t* opp, void* scratch, size_t scratch_bytes, void* stream_) {
    if (!tri || !opp || !scratch) return fail(EHR_ERR_INVALID, "ehr_antialias_topology: NULL tensor");
    unsigned n = topo_slots(T);
    if (scratch_bytes < (size_t)n * sizeof(EdgeSlot))
        return fail(EHR_ERR_INVALID, "ehr_antialias_topology: scratch too small (%zu < %zu)", scratch_bytes,
                    (size_t)n * sizeof(EdgeSlot));
    if (T <= 0) return EHR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    EdgeSlot* tab = (EdgeSlot*)scratch;
    topo_init_kernel<<<(n + 255) / 256, 256, 0, stream>>>(tab, n);
    EHR_LAUNCH_CHECK();
    unsigned grid = (unsigned)((3 * (size_t)T + 255) / 256);
    for (int pass = 0; pass < 3; pass++) {
        topo_kernel<<<grid, 256, 0, stream>>>(tri, T, tab, n - 1, pass, opp);
        EHR_LAUNCH_CHECK();
    }
    return EHR_OK;
}

static size_t aa_segments(int B, int H, int W) {  // one per (image, 32 x 8 tile)
    return (size_t)B * ((W + AA_TW - 1) / AA_TW) * ((H + AA_TH - 1) / AA_TH);
}
size_t ehr_antialias_work_bytes(int B, int H, int W) { return std::max<size_t>(aa_segments(B, H, W), 1) * AA_SEG * sizeof(int4); }

int ehr_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri, const int32_t* opp,
                      int range_mode, int B, int V, int T, int H, int W, int C, float* out, void* work,
                      const unsigned char* tile_flags, void* stream_) {
    return ehr_antialias_fwd_zg(color, rast, pos, tri, opp, range_mode, B, V, T, H, W, C, out, work, tile_flags, nullptr, stream_);
}

int ehr_antialias_fwd_zg(const float* color, const float* rast, const float* pos, const int32_t* tri, const int32_t* opp,
                         int range_mode, int B, int V, int T, int H, int W, int C, float* out, void* work,
                         const unsigned char* tile_flags, float* grad_pos_zero, void* stream_) {
    if (!color || !rast || !pos || !tri || !opp || !out || !work)
        return fail(EHR_ERR_INVALID, "ehr_antialias_fwd: NULL tensor");
    if (out == color) return fail(EHR_ERR_INVALID, "ehr_antialias_fwd: out must not alias color (every pixel reads its neighbours' colours)");
    if (B >= 32768) return fail(EHR_ERR_INVALID, "ehr_antialias_fwd: batch too large");
    hipStream_t stream = (hipStream_t)stream_;
    size_t n = (size_t)B * H * W;
    const size_t nzero = (size_t)(range_mode ? 1 : B) * V;  // float4s of grad_pos_zero (pos's shape)
    if (nzero > 0x7fffffffu) return fail(EHR_ERR_INVALID, "ehr_antialias_fwd: vertex array too large");
    if (n == 0) {
        if (grad_pos_zero && nzero) EHR_HIP(hipMemsetAsync(grad_pos_zero, 0, nzero * 16, stream));
        return EHR_OK;
    }
    aa_fwd_kernel<<<dim3((W + AA_TW - 1) / AA_TW, (H + AA_TH - 1) / AA_TH, B), 256, 0, stream>>>(
        color, (const float4*)rast, (const float4*)pos, tri, opp, range_mode, B, V, T, H, W, C, (W + AA_TW - 1) / AA_TW,
        (H + AA_TH - 1) / AA_TH, out, (int4*)work, tile_flags, (float4*)grad_pos_zero, (int)nzero);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

int ehr_antialias_grad(const float* color, const float* rast, const float* pos, const int32_t* tri, const float* dy,
                       const void* work, int range_mode, int B, int V, int T, int H, int W, int C, float* grad_color,
                       float* grad_pos, void* stream_) {
    if (!color || !rast || !pos || !tri || !dy || !work || !grad_pos)  // (grad_color may be NULL: the caller does not need it)
        return fail(EHR_ERR_INVALID, "ehr_antialias_grad: NULL tensor");
    hipStream_t stream = (hipStream_t)stream_;
    size_t n = (size_t)B * H * W;
    if (n == 0) return EHR_OK;
    int rc;
    if (grad_color && (rc = copy_words(grad_color, dy, n * C, stream))) return rc;
    const int nseg = (int)aa_segments(B, H, W);
    aa_grad_kernel<<<(nseg + 3) / 4, 256, 0, stream>>>(color, (const float4*)rast, (const float4*)pos, tri, dy, (const int4*)work,
                                                       nseg, range_mode, V, T, H, W, C, grad_color, grad_pos);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

}  // extern "C"
