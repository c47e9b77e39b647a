// ehr_fused.hip -- the fused hot path: B views x L links -> composite mask, per-frame SSE loss and
// d(loss_b)/d(MVP[b,l]) in ONE pass over the image, restating
//   /root/reference/easyhec/modeling/models/rb_solve/rb_solver.py:60-72   (per-link render, sum, clamp, SSE)
//   /root/reference/easyhec/structures/nvdiffrast_renderer.py:33-47        (rasterize -> interpolate -> antialias -> flip)
//   /root/reference/easyhec/utils/nvdiffrast_utils.py:14-18                (transform_pos)
// and the backward of all of it down to the 4x4 matrix of every (view, link).
//
// One workgroup owns a 32x8-pixel tile plus a 1-pixel halo.  For every link that touches the tile it rasterizes the
// link's queued triangles into an LDS depth/id buffer (ds_min_u64), finds the pixel pairs whose triangle ids differ,
// runs the silhouette analysis on a densely packed hit list (block-wide deterministic compaction), and GATHERS the
// antialias blend per interior pixel in a fixed order -- no global or LDS float atomics, so results are
// bit-reproducible.  Blended pairs are kept as compact items in LDS; once all links are composited the per-pixel
// loss gradient is known and the same workgroup back-propagates the items to 12 numbers per link (rows x, y, w of
// d loss / d MVP).  HBM traffic per pixel is one read of the reference mask and one write of the rendered mask.
#include <stdlib.h>

#include <algorithm>

#include "ehr_fused_core.h"

namespace ehr {

constexpr int RW = EHR_TILE_W + 2;  // region = tile + 1-pixel halo
constexpr int RH = EHR_TILE_H + 2;
constexpr int RN = RW * RH;          // 340
constexpr int CAND_PER_THREAD = (2 * RN + EHR_TILE_THREADS - 1) / EHR_TILE_THREADS;  // 3
constexpr int MAX_ITEMS = 704;      // blended pairs kept per tile (all links); overflow is reported, never silent
constexpr int MAX_LINKS = 32;
#ifndef EHR_LEAN_WAVES
#define EHR_LEAN_WAVES 4  // waves per SIMD the lean tile kernel is compiled for (4 workgroups per CU)
#endif

struct Item {
    int packed;  // bits 0-9 q (region index of pixel0) | 10 d | 11-12 di | 13 tri1 | 14 (c1 - c0 > 0)
    int v1, v2;  // the two vertices of the crossing silhouette edge (global ids)
    float alpha;
};

// clip-space vertices of every (view, vertex): posc[b][v] = MVP[b, vert_link[v]] * [x, y, z, 1]
// (easyhec/utils/nvdiffrast_utils.py:14-18 for all links of a view at once)
__global__ void __launch_bounds__(256) fused_vertex_kernel(const float* __restrict__ verts,
                                                           const int32_t* __restrict__ vert_link,
                                                           const float* __restrict__ mvp, int V, int L,
                                                           float4* __restrict__ posc) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (v >= V) return;
    const int l = vert_link[v];
    float4 o = make_float4(0.f, 0.f, 0.f, -1.f);  // invalid link -> behind the camera, never drawn
    if ((unsigned)l < (unsigned)L) o = transform_vertex(mvp + ((size_t)b * L + l) * 16, verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]);
    posc[(size_t)b * V + v] = o;
}

// Merged first stage of a solver step: pose_forward (6 threads, one partial each) + zeroing of the queue counters +
// vertex transform, in one launch.  grid = (ceil(V / 256), B).
__global__ void __launch_bounds__(256)
step_vertex_kernel(const float* __restrict__ verts, const int32_t* __restrict__ vert_link, const float* __restrict__ dof,
                   const float* __restrict__ K, const float* __restrict__ link_poses, int V, int L, int H, int W, float n,
                   float f, float4* __restrict__ posc, float* __restrict__ mvp, float* __restrict__ tc_jac,
                   const int* __restrict__ step, float* __restrict__ history, int history_rows, int* __restrict__ zero,
                   int nzero, int* __restrict__ zero2, int nzero2) {
    __shared__ float Tc[16];
    __shared__ float M[MAX_LINKS][16];
    const int b = blockIdx.y, tid = threadIdx.x;
    const bool first = blockIdx.x == 0 && b == 0;
    if (tid < 6) {
        Dual<1> T[16];
        se3_exp_dual<1>(dof, 1e-4f, T, tid);
        if (tid == 0)
            for (int i = 0; i < 16; i++) Tc[i] = T[i].v;
        if (first) {
            for (int i = 0; i < 16; i++) {
                if (tid == 0) tc_jac[i] = T[i].v;
                tc_jac[16 * (tid + 1) + i] = T[i].d[0];
            }
            if (tid == 0 && history && step) {
                int row = step[0];
                if (row >= 0 && row < history_rows)
                    for (int k = 0; k < 6; k++) history[6 * row + k] = dof[k];
            }
        }
    }
    // clear this block's slice of the queue counters (count | cursor | slow flags | meta) for the step
    {
        const int nblk = gridDim.x * gridDim.y, blk = b * gridDim.x + blockIdx.x;
        const int per = (nzero + nblk - 1) / nblk;
        const int z0 = blk * per, z1 = min(z0 + per, nzero);
        for (int i = z0 + tid; i < z1; i += 256) zero[i] = 0;
        if (blk == 0)
            for (int i = tid; i < nzero2; i += 256) zero2[i] = 0;  // the fixed-point accumulators (a few KB)
    }
    __syncthreads();
    if (tid < L) {
        float P[16], C[16];
        projection(K, H, W, n, f, P);
        mvp_from_pose(Tc, P, link_poses + ((size_t)b * L + tid) * 16, C);
        for (int k = 0; k < 16; k++) M[tid][k] = C[k];
        if (blockIdx.x == 0)
            for (int k = 0; k < 16; k++) mvp[((size_t)b * L + tid) * 16 + k] = C[k];
    }
    __syncthreads();
    const int v = blockIdx.x * 256 + tid;
    if (v >= V) return;
    const int l = vert_link[v];
    float4 o = make_float4(0.f, 0.f, 0.f, -1.f);
    if ((unsigned)l < (unsigned)L) o = transform_vertex(M[l], verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]);
    posc[(size_t)b * V + v] = o;
}

// Streaming pass over the tiles no triangle touches (~90 % of the image): mask = 0, loss += ref^2.
// grid = (tile rows, views); one wave handles one 32x8 tile as 64 float4 accesses (full 128-byte lines).
__global__ void __launch_bounds__(256) fused_empty_kernel(BinGeom g, const int* __restrict__ tile_total,
                                                          const float* __restrict__ ref, float* __restrict__ mask,
                                                          long long* __restrict__ facc, int acc_stride,
                                                          int* __restrict__ meta) {
    const int ty = blockIdx.x, b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ly = lane >> 3, lx4 = (lane & 7) * 4;  // 8 rows x 8 float4
    const int iy = ty * EHR_TILE_H + ly;
    const bool vec_ok = (g.W & 3) == 0;
    long long wsum = 0;  // this wave's share of the view's loss, fixed point
    for (int tx = wave; tx < g.ntx; tx += 4) {
        const int tile = ty * g.ntx + tx;
        if (tile_total[b * g.nt + tile] != 0) continue;  // wave-uniform
        const int ix = tx * EHR_TILE_W + lx4;
        float s = 0.f;
        if (iy < g.H && ix < g.W) {
            const size_t im = ((size_t)b * g.H + (g.H - 1 - iy)) * g.W + ix;
            if (vec_ok) {
                float4 r = *reinterpret_cast<const float4*>(ref + im);
                s = ((r.x * r.x + r.y * r.y) + r.z * r.z) + r.w * r.w;
                if (mask) *reinterpret_cast<float4*>(mask + im) = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                for (int k = 0; k < 4 && ix + k < g.W; k++) {
                    float r = ref[im + k];
                    s += r * r;
                    if (mask) mask[im + k] = 0.f;
                }
            }
        }
        s = wave_sum(s);
        if (lane == 0) {
            if (!(s < 1.0e9f)) meta[EHR_META_OVERFLOW] = 1;
            wsum += __double2ll_rn((double)s * EHR_FIX_SCALE);
        }
    }
    if (lane == 0 && wsum) atomicAdd((unsigned long long*)&facc[(size_t)b * acc_stride + acc_stride - 1], (unsigned long long)wsum);
}

// Heavy tiles: persistent workgroups walk the work list of non-empty tiles.  SLOW = false is the lean instantiation
// (no 64-bit / clipping path, <= 128 VGPRs -> 4 workgroups per CU); tiles that hold a triangle needing that path are
// on the second work list and run through the SLOW = true instantiation.
template <bool SLOW>
__global__ void __launch_bounds__(EHR_TILE_THREADS, SLOW ? 1 : EHR_LEAN_WAVES)
fused_tile_kernel(ClipSource src, BinGeom g, const float* __restrict__ verts, const int* __restrict__ counts,
                  const int* __restrict__ offsets, const int4* __restrict__ entries, int entries_cap,
                  const int* __restrict__ worklist, const int32_t* __restrict__ opp, const float* __restrict__ ref,
                  float* __restrict__ mask, long long* __restrict__ facc, int want_grad, int* __restrict__ meta,
                  int dbg) {
    __shared__ u64 key[RN];
    __shared__ float pairA[2][RN];
    __shared__ unsigned short hits[2 * RN];
    __shared__ Item items[MAX_ITEMS];
    __shared__ BlockRaster wscratch;
    __shared__ int seg_end[MAX_LINKS];
    __shared__ int cnt_l[MAX_LINKS];
    __shared__ int off_l[MAX_LINKS];
    __shared__ float gpix[EHR_TILE_W * EHR_TILE_H];

    const int tid = threadIdx.x;
    const int lx = tid % EHR_TILE_W, ly = tid / EHR_TILE_W;
    const int L = g.L, W = g.W, H = g.H;
    const int acc_stride = 12 * L + 1;  // per view: 12 numbers per link, then the frame loss
    const int myq = (ly + 1) * RW + (lx + 1);
    const int nwork = meta[SLOW ? EHR_META_NWORK_SLOW : EHR_META_NWORK];
#ifdef EHR_PHASE_TIMING
    // profiling build only (tools/phase_profile.sh): cycles spent up to each phase marker, summed over workgroups
    long long ph_last = __builtin_readcyclecounter();
#define EHR_PHASE(i)                                                                   \
    do {                                                                               \
        long long now_ = __builtin_readcyclecounter();                                 \
        if (tid == 0) atomicAdd((unsigned long long*)(meta + 8) + (i), (unsigned long long)(now_ - ph_last)); \
        ph_last = now_;                                                                \
    } while (0)
#else
#define EHR_PHASE(i) do { } while (0)
#endif

    for (int wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const int gt = worklist[wi];  // b * nt + tile
        const int b = gt / g.nt, tile = gt - b * g.nt;
        const int tx = tile % g.ntx, ty = tile / g.ntx;
        const int rx0 = tx * EHR_TILE_W - 1, ry0 = ty * EHR_TILE_H - 1;
        const int ix = tx * EHR_TILE_W + lx, iy = ty * EHR_TILE_H + ly;
        const bool in_img = ix < W && iy < H;
        const int kidx = gt * L;
        long long* vacc = facc + (size_t)b * acc_stride;
        const float4* pv = src.verts(b);

        __syncthreads();  // previous tile's LDS users are done
        if (tid < L) {
            cnt_l[tid] = counts[kidx + tid];
            off_l[tid] = offsets[kidx + tid];
        }
        // the reference-mask pixel is only needed after all links are composited: fetch it now, use it later
        float refv = 0.f;
        size_t im = 0;
        if (in_img) {
            im = ((size_t)b * H + (H - 1 - iy)) * W + ix;
            refv = ref[im];
        }
        __syncthreads();

        float acc = 0.f;
        int nitems = 0;  // uniform across the block

        // queue entries of the NEXT link pass are fetched while the current one is processed (one global round trip
        // per pass off the critical path)
        int4 pre_e = make_int4(0, 0, 0, 0);
        {
            int lf = 0;
            while (lf < L && cnt_l[lf] == 0) lf++;
            if (lf < L && tid < cnt_l[lf] && off_l[lf] + tid < entries_cap) pre_e = entries[off_l[lf] + tid];
        }

        for (int l = 0; l < L; l++) {
            int n = cnt_l[l];
            if (n == 0) {
                if (tid == 0) seg_end[l] = nitems;
                continue;
            }
            const int off = off_l[l];
            if (off + n > entries_cap) n = max(entries_cap - off, 0);
            const int4 cur_e = pre_e;
            {
                int ln = l + 1;
                while (ln < L && cnt_l[ln] == 0) ln++;
                if (ln < L && tid < cnt_l[ln] && off_l[ln] + tid < entries_cap) pre_e = entries[off_l[ln] + tid];
            }
            for (int i = tid; i < RN; i += EHR_TILE_THREADS) key[i] = ~0ull;
            __syncthreads();
            // ---- coverage + z-test of the link's queued triangles
            EHR_PHASE(0);
            if (!(dbg & 2)) {
                RoundZero pre;
                pre.e = cur_e;
                raster_queue<RW, RH, SLOW, 1>(src, b, entries + off, n, W, H, rx0, ry0, key, &wscratch, meta, pre);
            }
            __syncthreads();
            EHR_PHASE(1);
            // ---- pixel pairs with different triangle ids -> dense hit list (deterministic order)
            unsigned myhit[CAND_PER_THREAD];
            int nh = 0;
#pragma unroll
            for (int j = 0; j < CAND_PER_THREAD; j++) {
                myhit[j] = 0xffffffffu;
                int c = tid + j * EHR_TILE_THREADS;
                if (c < 2 * RN) {
                    int d = c >= RN ? 1 : 0;
                    int q = c - d * RN;
                    int qx = q % RW, qy = q / RW;
                    int nx = qx + 1 - d, ny = qy + d;
                    pairA[d][q] = 0.f;
                    bool ok = nx < RW && ny < RH;
                    int ax = rx0 + qx, ay = ry0 + qy, bx = rx0 + nx, by = ry0 + ny;
                    ok = ok && ax >= 0 && ay >= 0 && bx < W && by < H;  // both pixels inside the image
                    bool qi = qx >= 1 && qx <= EHR_TILE_W && qy >= 1 && qy <= EHR_TILE_H;
                    bool ni = nx >= 1 && nx <= EHR_TILE_W && ny >= 1 && ny <= EHR_TILE_H;
                    ok = ok && (qi || ni);  // at least one of them interior to the tile
                    if (ok) {
                        // Only pairs with exactly one covered pixel can change the result: with constant colour
                        // inside a link, a blend between two covered pixels is alpha * (1 - 1) = 0 in value and in
                        // gradient, so those (the vast majority of id changes) need no silhouette analysis.
                        const bool c0 = key[q] != ~0ull, c1 = key[ny * RW + nx] != ~0ull;
                        if (c0 != c1) {
                            myhit[j] = (unsigned)(q | (d << 15));
                            nh++;
                        }
                    }
                }
            }
            int nhits;
            if (dbg & 1) nh = 0;
            int hoff = block_offset(nh, wscratch.wave_tot, nhits);
#pragma unroll
            for (int j = 0; j < CAND_PER_THREAD; j++)
                if (myhit[j] != 0xffffffffu) hits[hoff++] = (unsigned short)myhit[j];
            __syncthreads();
            EHR_PHASE(2);
            // ---- silhouette analysis of the hits (restates nvdiffrast's antialias mesh kernel), 256 per round
            for (int hbase = 0; hbase < nhits; hbase += EHR_TILE_THREADS) {
                const int h = hbase + tid;
                Item it;
                it.packed = 0;
                it.v1 = 0;
                it.v2 = 0;
                it.alpha = 0.f;
                int keep = 0;
                if (h < nhits) {
                    int hq = hits[h];
                    int d = hq >> 15, q = hq & 0x7fff;
                    int qx = q % RW, qy = q / RW;
                    int nq = q + (d ? RW : 1);
                    u64 k0 = key[q], k1 = key[nq];
                    int tri0 = (k0 == ~0ull) ? -1 : (int)(unsigned)k0;
                    int tri1 = (k1 == ~0ull) ? -1 : (int)(unsigned)k1;
                    float zw0 = ord_unkey((unsigned)(k0 >> 32)), zw1 = ord_unkey((unsigned)(k1 >> 32));
                    int t = (tri0 >= 0) ? tri0 : tri1;
                    if (tri0 >= 0 && tri1 >= 0) t = (zw0 < zw1) ? tri0 : tri1;
                    bool chose0 = !(t == tri1);
                    int px = rx0 + qx, py = ry0 + qy;
                    if (!chose0) {
                        px += 1 - d;
                        py += d;
                    }
                    float4 p[3], o[3];
                    int vi[3], ov[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        vi[k] = src.tri[3 * t + k];
                        ov[k] = opp[3 * t + k];
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++) p[k] = pv[vi[k]];
#pragma unroll
                    for (int k = 0; k < 3; k++) o[k] = ((unsigned)ov[k] < (unsigned)src.V) ? pv[ov[k]] : p[k];
                    AAPair a = aa_analyze(p, o, px, py, d, chose0, W, H);
                    if (a.found) {
                        pairA[d][q] = a.alpha;
                        // keep for the backward pass if the destination pixel is interior to this tile
                        int oq = (a.alpha > 0.f) ? q : nq;
                        int ox = oq % RW, oy = oq / RW;
                        bool oi = ox >= 1 && ox <= EHR_TILE_W && oy >= 1 && oy <= EHR_TILE_H;
                        if (oi && a.alpha != 0.f && (tri0 >= 0) != (tri1 >= 0)) {
                            it.packed = q | (d << 10) | (a.di << 11) | (a.tri1 << 13) | ((tri1 >= 0 ? 1 : 0) << 14);
                            it.v1 = (a.di == 0) ? vi[1] : (a.di == 1 ? vi[2] : vi[0]);  // edge di: v1-v2, v2-v0, v0-v1
                            it.v2 = (a.di == 0) ? vi[2] : (a.di == 1 ? vi[0] : vi[1]);
                            it.alpha = a.alpha;
                            keep = want_grad;
                        }
                    }
                }
                int nfound;
                int ioff = block_offset(keep, wscratch.wave_tot, nfound);
                if (keep) {
                    int at = nitems + ioff;
                    if (at < MAX_ITEMS)
                        items[at] = it;
                    else
                        meta[EHR_META_OVERFLOW] = 1;  // reported through loss = NaN
                }
                nitems = min(nitems + nfound, MAX_ITEMS);
            }
            if (tid == 0) seg_end[l] = nitems;
            __syncthreads();
            EHR_PHASE(3);
            // ---- gather the antialiased value of this link at my pixel (fixed order: down, left, right, up pair)
            {
                float cq = (key[myq] != ~0ull) ? 1.f : 0.f;
                float val = cq;
                float a;
                a = pairA[1][myq - RW];
                if (a < 0.f) val += a * (cq - ((key[myq - RW] != ~0ull) ? 1.f : 0.f));
                a = pairA[0][myq - 1];
                if (a < 0.f) val += a * (cq - ((key[myq - 1] != ~0ull) ? 1.f : 0.f));
                a = pairA[0][myq];
                if (a > 0.f) val += a * (((key[myq + 1] != ~0ull) ? 1.f : 0.f) - cq);
                a = pairA[1][myq];
                if (a > 0.f) val += a * (((key[myq + RW] != ~0ull) ? 1.f : 0.f) - cq);
                acc += val;
            }
            __syncthreads();
            EHR_PHASE(4);
        }

        // ---- composite, loss, mask write (image convention: row 0 = top)
        float e2 = 0.f, gval = 0.f;
        if (in_img) {
            float m = acc > 1.f ? 1.f : acc;
            float e = m - refv;
            e2 = e * e;
            gval = (acc <= 1.f) ? 2.f * e : 0.f;
            if (mask) mask[im] = m;
        }
        gpix[tid] = gval;
        {
            float s = wave_sum(e2);
            if ((tid & 63) == 0) fix_add(&vacc[12 * L], s, meta);
            __syncthreads();  // gpix is read by the backward pass
        }
        EHR_PHASE(5);
        if (!want_grad) continue;

        // ---- backward: blended pairs -> rows (x, y, w) of d loss / d MVP, per link
        int seg0 = 0;
        for (int l = 0; l < L; l++) {
            if (cnt_l[l] == 0) continue;
            const int seg1 = seg_end[l];
            float G[12];
#pragma unroll
            for (int k = 0; k < 12; k++) G[k] = 0.f;
            for (int it = seg0 + tid; it < seg1; it += EHR_TILE_THREADS) {
                const Item im = items[it];
                int q = im.packed & 1023, d = (im.packed >> 10) & 1;
                int tri1 = (im.packed >> 13) & 1;
                float dc = ((im.packed >> 14) & 1) ? 1.f : -1.f;
                int nq = q + (d ? RW : 1);
                int oq = (im.alpha > 0.f) ? q : nq;
                int ox = oq % RW - 1, oy = oq / RW - 1;
                float gi = gpix[oy * EHR_TILE_W + ox];
                float dd = gi * dc;
                if (gi == 0.f || dd == 0.f) continue;
                int v1 = im.v1, v2 = im.v2;
                int qx = q % RW, qy = q / RW;
                int px = rx0 + qx, py = ry0 + qy;
                if (tri1) {
                    px += 1 - d;
                    py += d;
                }
                float g1[3], g2[3];
                aa_pos_grad(pv[v1], pv[v2], px, py, d, im.alpha, dd, W, H, g1, g2);
                const float* a1 = verts + 3 * (size_t)v1;
                const float* a2 = verts + 3 * (size_t)v2;
                float h1[4] = {a1[0], a1[1], a1[2], 1.f}, h2[4] = {a2[0], a2[1], a2[2], 1.f};
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) G[4 * r + c] += g1[r] * h1[c] + g2[r] * h2[c];
            }
            seg0 = seg1;
            // every wave adds its 12 sums itself (lane k holds number k): no LDS staging, no barriers
            float mine = 0.f;
#pragma unroll
            for (int k = 0; k < 12; k++) {
                float s = wave_sum(G[k]);
                if ((tid & 63) == k) mine = s;
            }
            if ((tid & 63) < 12) fix_add(&vacc[12 * l + (tid & 63)], mine, meta);
        }
        EHR_PHASE(6);
    }
}

}  // namespace ehr

using namespace ehr;

extern "C" {

int ehr_fused_plan(ehr_ctx* ctx, int B, int L, int V, int T, int H, int W, float slack, const float* verts,
                   const int32_t* tris, const int32_t* tri_link, const int32_t* opp) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_fused_plan: ctx is NULL");
    if (B <= 0 || L <= 0 || L > MAX_LINKS || V < 0 || T < 0 || H <= 0 || W <= 0 || H > 32768 || W > 32768)
        return fail(EHR_ERR_INVALID, "ehr_fused_plan: bad sizes (1 <= L <= %d)", MAX_LINKS);
    if (!(slack >= 1.f)) slack = 4.f;
    if (ctx->gexec) {  // a captured chain holds the old plan's pointers and shape
        EHR_HIP(hipGraphExecDestroy(ctx->gexec));
        ctx->gexec = nullptr;
    }
    if (ctx->path_vbuf) {
        int rc0 = vbuf_plan(ctx, B, L, V, T, H, W, slack, verts, tris, tri_link, opp);
        if (rc0) return rc0;
        ctx->pB = B;
        ctx->pL = L;
        ctx->pV = V;
        ctx->pT = T;
        ctx->pH = H;
        ctx->pW = W;
        return EHR_OK;
    }
    BinGeom g = make_geom(H, W, L);
    size_t nkeys = (size_t)B * g.nt * L;
    if (nkeys > 0x3fffffff) return fail(EHR_ERR_INVALID, "ehr_fused_plan: too many (view, tile, link) queues");
    int rc;
    if ((rc = ctx->counts.reserve((2 * nkeys + (size_t)B * g.nt + EHR_META_INTS) * sizeof(int)))) return rc;
    if ((rc = ctx->offsets.reserve(nkeys * sizeof(int)))) return rc;
    // queue storage: a triangle is queued once per tile its bounding box (+1 pixel) touches
    size_t want = (size_t)((double)slack * (double)B * (double)std::max(T, 1)) + 65536;
    want = std::min(want, (size_t)0x7fffffff / sizeof(int4));
    if (want > ctx->entries_cap) {
        if ((rc = ctx->entries.reserve(want * sizeof(int4)))) return rc;
        ctx->entries_cap = want;
    }
    if ((rc = ctx->tile_part.reserve((size_t)B * (12 * (size_t)L + 1) * sizeof(long long)))) return rc;  // fixed-point sums
    if ((rc = ctx->tile_list.reserve((size_t)3 * B * g.nt * sizeof(int)))) return rc;  // tile totals | lean work list | slow work list
    if ((rc = ctx->posc.reserve((size_t)B * std::max(V, 1) * sizeof(float4)))) return rc;
    int dev = 0;
    EHR_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    EHR_HIP(hipGetDeviceProperties(&prop, dev));
    ctx->num_cus = prop.multiProcessorCount;
    if (!ctx->side) {
        EHR_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        EHR_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        EHR_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        EHR_HIP(hipEventCreateWithFlags(&ctx->ev_fill, hipEventDisableTiming));
    }
    ctx->pB = B;
    ctx->pL = L;
    ctx->pV = V;
    ctx->pT = T;
    ctx->pH = H;
    ctx->pW = W;
    return EHR_OK;
}

// The launch chain of the fused op.  head/tail == nullptr: generic form (mvp given, stops at loss / grad_mvp).
// head/tail != nullptr: solver-step form (pose forward merged into the vertex kernel, pose backward (+ Adam) merged into
// the reduction): 7 launches instead of 12.
static int fused_chain(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* tri_link,
                       const int32_t* vert_link, const int32_t* opp, float* mvp, const float* ref, int B, int L, int V,
                       int T, int H, int W, float* mask, float* loss, float* grad_mvp, const StepHead* head,
                       const StepTail* tail, void* stream_) {
    if (!ctx) return fail(EHR_ERR_INVALID, "fused op: ctx is NULL");
    if (!verts || !tris || !tri_link || !vert_link || !opp || !mvp || !ref || !loss)
        return fail(EHR_ERR_INVALID, "fused op: NULL tensor");
    if (ctx->pB != B || ctx->pL != L || ctx->pV != V || ctx->pT != T || ctx->pH != H || ctx->pW != W)
        return fail(EHR_ERR_INVALID, "fused op: shape differs from the planned one; call ehr_fused_plan first");
    hipStream_t stream = (hipStream_t)stream_;
    if (ctx->path_vbuf)
        return vbuf_chain(ctx, verts, tris, tri_link, vert_link, opp, mvp, ref, B, L, V, T, H, W, mask, loss, grad_mvp,
                          head, tail, stream);
    BinGeom g = make_geom(H, W, L);
    const int ntiles = B * g.nt;
    const int nkeys = ntiles * L;
    int* counts = (int*)ctx->counts.ptr;
    int* cursors = counts + nkeys;
    int* tile_slow = counts + 2 * nkeys;       // [ntiles]
    int* meta = tile_slow + ntiles;            // [EHR_META_INTS]
    int* offsets = (int*)ctx->offsets.ptr;
    int4* entries = (int4*)ctx->entries.ptr;
    int* tile_total = (int*)ctx->tile_list.ptr;
    int* worklist = tile_total + ntiles;
    float4* posc = (float4*)ctx->posc.ptr;
    long long* facc = (long long*)ctx->tile_part.ptr;
    const int acc_stride = 12 * L + 1, nacc_ints = 2 * B * acc_stride;
    const int ecap = (int)std::min(ctx->entries_cap, (size_t)0x7fffffff);
    ClipSource src;
    src.pos = posc;
    src.tri = tris;
    src.tri_link = tri_link;
    src.ranges = nullptr;
    src.V = V;
    src.T = T;
    src.L = L;
    src.image_stride = V;

    // optional per-stage events (measurement hook)
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
    // stage 0: clear queues, transform vertices, count
        // counts | cursors | tile_slow | meta[0..8); the profiling counters behind meta[8] accumulate across calls
    const int nzero = 2 * nkeys + ntiles + 8;
    if (head && V > 0) {
        step_vertex_kernel<<<dim3((V + 255) / 256, B), 256, 0, stream>>>(
            verts, vert_link, head->dof, head->K, head->link_poses, V, L, H, W, head->n, head->f, posc, mvp, head->tc_jac,
            head->step, head->history, head->history_rows, counts, nzero, (int*)facc, nacc_ints);
        EHR_LAUNCH_CHECK();
    } else {
        EHR_HIP(hipMemsetAsync(counts, 0, (size_t)nzero * sizeof(int), stream));
        EHR_HIP(hipMemsetAsync(facc, 0, (size_t)nacc_ints * sizeof(int), stream));
        if (V > 0) {
            fused_vertex_kernel<<<dim3((V + 255) / 256, B), 256, 0, stream>>>(verts, vert_link, mvp, V, L, posc);
            EHR_LAUNCH_CHECK();
        }
    }
    dim3 bgrid((T + 255) / 256, B);
    if (T > 0) {
        bin_kernel<1, false><<<bgrid, 256, 0, stream>>>(src, g, counts, cursors, offsets, nullptr, 0, meta, tile_slow);
        EHR_LAUNCH_CHECK();
    }
    if (ev) EHR_HIP(hipEventRecord(ev[1], stream));
    // stage 1: queue allocation + work list
    bin_alloc_kernel<<<(ntiles + 255) / 256, 256, 0, stream>>>(counts, offsets, tile_total, worklist, tile_slow, ntiles, L,
                                                               meta);
    EHR_LAUNCH_CHECK();
    if (ev) EHR_HIP(hipEventRecord(ev[2], stream));
    // the empty-tile stream only needs the tile totals: fork it onto the side stream so that it runs under the queue
    // fill and the tile kernels (it is bandwidth-bound, they are latency-bound); joined before the reduction
    const bool overlap = !ctx->timing && ctx->side != nullptr;
    if (overlap) {
        EHR_HIP(hipEventRecord(ctx->ev_fork, stream));
        EHR_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        fused_empty_kernel<<<dim3(g.nty, B), 256, 0, ctx->side>>>(g, tile_total, ref, mask, facc, acc_stride, meta);
        EHR_LAUNCH_CHECK();
    }
    // stage 2: fill
    if (T > 0) {
        bin_kernel<1, true><<<bgrid, 256, 0, stream>>>(src, g, counts, cursors, offsets, entries, ecap, meta, nullptr);
        EHR_LAUNCH_CHECK();
    }
    if (ev) EHR_HIP(hipEventRecord(ev[3], stream));
    if (overlap) {  // the side stream's second kernel (slow tiles, below) needs the filled queues
        EHR_HIP(hipEventRecord(ctx->ev_fill, stream));
        EHR_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fill, 0));
    }
    // stage 3: tiles -- streaming pass over the empty ones, persistent workgroups over the work list
    if (!overlap) {
        fused_empty_kernel<<<dim3(g.nty, B), 256, 0, stream>>>(g, tile_total, ref, mask, facc, acc_stride, meta);
        EHR_LAUNCH_CHECK();
    }
    if (ev) EHR_HIP(hipEventRecord(ev[4], stream));
    static const int grid_mult = getenv("EHR_TILE_GRID_MULT") ? atoi(getenv("EHR_TILE_GRID_MULT")) : 6;  // tuning knob
    const int tgrid = std::max(1, std::min(ntiles, ctx->num_cus * std::max(1, grid_mult)));
    static const int dbg_skip = getenv("EHR_DEBUG_SKIP") ? atoi(getenv("EHR_DEBUG_SKIP")) : 0;  // profiling aid only
    fused_tile_kernel<false><<<tgrid, EHR_TILE_THREADS, 0, stream>>>(src, g, verts, counts, offsets, entries, ecap,
                                                                    worklist, opp, ref, mask, facc,
                                                                    grad_mvp ? 1 : 0, meta, dbg_skip);
    EHR_LAUNCH_CHECK();
    if (ev) EHR_HIP(hipEventRecord(ev[5], stream));
    // tiles holding a near-clipped or very large triangle (normally none): same kernel with the 64-bit path compiled in.
    // Disjoint tiles, so it runs beside the lean kernel on the side stream (behind the empty-tile pass) instead of
    // adding its launch + drain (~4 us even when its work list is empty) to the critical path.
    static const int side_slow = getenv("EHR_SIDE_SLOW") ? atoi(getenv("EHR_SIDE_SLOW")) : 1;  // tuning knob
    hipStream_t sstream = (overlap && side_slow) ? ctx->side : stream;
    fused_tile_kernel<true><<<std::max(1, std::min(ntiles, ctx->num_cus / 8)), EHR_TILE_THREADS, 0, sstream>>>(
        src, g, verts, counts, offsets, entries, ecap, worklist + ntiles, opp, ref, mask, facc, grad_mvp ? 1 : 0,
        meta, dbg_skip);
    EHR_LAUNCH_CHECK();
    if (overlap) {
        EHR_HIP(hipEventRecord(ctx->ev_join, ctx->side));
        EHR_HIP(hipStreamWaitEvent(stream, ctx->ev_join, 0));
    }
    if (ev) EHR_HIP(hipEventRecord(ev[6], stream));
    // stage 4: accumulators -> loss / grad_mvp (+ pose backward and Adam in the solver-step form), one workgroup
    if (tail) {
        fused_finish_kernel<true><<<1, 256, 0, stream>>>(g, B, facc, loss, grad_mvp, meta, *tail, 1, nullptr, 1);
    } else {
        StepTail none = {};
        fused_finish_kernel<false><<<1, 256, 0, stream>>>(g, B, facc, loss, grad_mvp, meta, none, 1, nullptr, 1);
    }
    EHR_LAUNCH_CHECK();
    if (ev) EHR_HIP(hipEventRecord(ev[7], stream));
    return EHR_OK;
}

int ehr_render_mask_loss(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* tri_link,
                         const int32_t* vert_link, const int32_t* opp, const float* mvp, const float* ref, int B,
                         int L, int V, int T, int H, int W, float* mask, float* loss, float* grad_mvp, void* stream) {
    return fused_chain(ctx, verts, tris, tri_link, vert_link, opp, const_cast<float*>(mvp), ref, B, L, V, T, H, W, mask,
                       loss, grad_mvp, nullptr, nullptr, stream);
}

int ehr_solver_step(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* tri_link,
                    const int32_t* vert_link, const int32_t* opp, const float* K, const float* link_poses,
                    const float* ref, int B, int L, int V, int T, int H, int W, float near_, float far_, float* dof,
                    float* adam_m, float* adam_v, int32_t* step, float* history, int history_rows, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float* mvp, float* tc_jac, float* mask, float* loss_b,
                    float* grad_mvp, float* red, float* loss_out, float* grad_out, int defer_adam, void* stream) {
    if (!K || !link_poses || !dof || !adam_m || !adam_v || !step || !tc_jac || !grad_mvp || !red)
        return fail(EHR_ERR_INVALID, "ehr_solver_step: NULL tensor");
    if (L > MAX_LINKS) return fail(EHR_ERR_INVALID, "ehr_solver_step: more than %d links", MAX_LINKS);
    StepHead head;
    head.dof = dof;
    head.K = K;
    head.link_poses = link_poses;
    head.tc_jac = tc_jac;
    head.step = step;
    head.history = history;
    head.history_rows = history_rows;
    head.n = near_;
    head.f = far_;
    StepTail tail;
    tail.K = K;
    tail.link_poses = link_poses;
    tail.tc_jac = tc_jac;
    tail.red = red;
    tail.dof = dof;
    tail.m = adam_m;
    tail.v = adam_v;
    tail.step = step;
    tail.loss_out = loss_out;
    tail.grad_out = grad_out;
    tail.n = near_;
    tail.f = far_;
    tail.lr = lr;
    tail.b1 = beta1;
    tail.b2 = beta2;
    tail.eps = eps;
    tail.wd = weight_decay;
    tail.defer_adam = defer_adam;
    return fused_chain(ctx, verts, tris, tri_link, vert_link, opp, mvp, ref, B, L, V, T, H, W, mask, loss_b, grad_mvp,
                       &head, &tail, stream);
}

int ehr_fused_status(ehr_ctx* ctx) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_fused_status: ctx is NULL");
    if (ctx->pB == 0) return EHR_OK;
    if (ctx->path_vbuf) {
        if (!ctx->vb_acc.ptr) return EHR_OK;
        EHR_HIP(hipDeviceSynchronize());
        int m4[4] = {0, 0, 0, 0};
        int rc0 = vbuf_meta_read(ctx, m4);
        if (rc0) return rc0;
        if (m4[EHR_META_OVERFLOW])
            return fail(EHR_ERR_OVERFLOW, "fused path: an accumulator or the blended-pair spill pool overflowed");
        return EHR_OK;
    }
    if (!ctx->counts.ptr) return EHR_OK;
    EHR_HIP(hipDeviceSynchronize());
    BinGeom g = make_geom(ctx->pH, ctx->pW, ctx->pL);
    const size_t nkeys = (size_t)ctx->pB * g.nt * ctx->pL;
    const size_t meta_off = 2 * nkeys + (size_t)ctx->pB * g.nt;
    int meta[4] = {0, 0, 0, 0};
    EHR_HIP(hipMemcpy(meta, (int*)ctx->counts.ptr + meta_off, sizeof(meta), hipMemcpyDeviceToHost));
#ifdef EHR_PHASE_TIMING
    {
        unsigned long long ph[16];
        EHR_HIP(hipMemcpy(ph, (int*)ctx->counts.ptr + meta_off + 8, sizeof(ph), hipMemcpyDeviceToHost));
        const char* names[8] = {"pre-raster", "raster", "hit-discovery", "analysis", "gather", "composite", "backward", ""};
        unsigned long long tot = 0;
        for (int i = 0; i < 7; i++) tot += ph[i];
        for (int i = 0; i < 7; i++)
            fprintf(stderr, "[ehr phase] %-14s %12llu cycles  %5.1f %%\n", names[i], ph[i], tot ? 100.0 * ph[i] / tot : 0.0);
        const char* sub[7] = {"r:load-wait", "r:setup", "r:prefix-sum", "r:stage+sync", "r:search", "r:walk", "r:tail+sync"};
        for (int i = 0; i < 7; i++)
            fprintf(stderr, "[ehr phase]   %-14s %12llu cycles  %5.1f %%\n", sub[i], ph[8 + i], tot ? 100.0 * ph[8 + i] / tot : 0.0);
        EHR_HIP(hipMemset((int*)ctx->counts.ptr + meta_off + 8, 0, sizeof(ph)));
    }
#endif
    if (meta[EHR_META_OVERFLOW])
        return fail(EHR_ERR_OVERFLOW, "fused path: a bin queue or a tile's blend list overflowed (%d queued, capacity %zu); "
                                      "re-plan with a larger slack", meta[0], ctx->entries_cap);
    return EHR_OK;
}

// ---- hipGraph capture of library launch chains ------------------------------------------------------------------

int ehr_graph_begin(ehr_ctx* ctx, void** capture_stream) {
    if (!ctx || !capture_stream) return fail(EHR_ERR_INVALID, "ehr_graph_begin: NULL argument");
    if (ctx->capturing) return fail(EHR_ERR_INVALID, "ehr_graph_begin: a capture is already open on this context");
    if (ctx->timing) return fail(EHR_ERR_INVALID, "ehr_graph_begin: disable ehr_fused_timing first");
    if (!ctx->cap_stream) EHR_HIP(hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
    if (ctx->gexec) {
        EHR_HIP(hipGraphExecDestroy(ctx->gexec));
        ctx->gexec = nullptr;
    }
    EHR_HIP(hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    *capture_stream = (void*)ctx->cap_stream;
    return EHR_OK;
}

int ehr_graph_end(ehr_ctx* ctx) {
    if (!ctx || !ctx->capturing) return fail(EHR_ERR_INVALID, "ehr_graph_end: no open capture");
    ctx->capturing = false;
    hipGraph_t graph = nullptr;
    EHR_HIP(hipStreamEndCapture(ctx->cap_stream, &graph));
    if (!graph) return fail(EHR_ERR_HIP, "ehr_graph_end: the capture was invalidated");
    hipError_t e = hipGraphInstantiate(&ctx->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        ctx->gexec = nullptr;
        return fail(EHR_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    ctx->gexec_reallocs = Scratch::reallocs;
    return EHR_OK;
}

int ehr_graph_launch(ehr_ctx* ctx, void* stream) {
    if (!ctx || !ctx->gexec) return fail(EHR_ERR_INVALID, "ehr_graph_launch: no instantiated graph");
    if (ctx->gexec_reallocs != Scratch::reallocs)  // a drop-in op or a re-plan moved scratch the graph points into
        return fail(EHR_ERR_INVALID, "ehr_graph_launch: library scratch was reallocated after the capture; capture again");
    EHR_HIP(hipGraphLaunch(ctx->gexec, (hipStream_t)stream));
    return EHR_OK;
}

int ehr_graph_release(ehr_ctx* ctx) {
    if (!ctx) return EHR_OK;
    if (ctx->capturing) {
        hipGraph_t graph = nullptr;
        (void)hipStreamEndCapture(ctx->cap_stream, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        ctx->capturing = false;
    }
    if (ctx->gexec) {
        EHR_HIP(hipGraphExecDestroy(ctx->gexec));
        ctx->gexec = nullptr;
    }
    return EHR_OK;
}

int ehr_fused_timing(ehr_ctx* ctx, int enable) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_fused_timing: ctx is NULL");
    ctx->timing = enable != 0;
    ctx->ev_used = 0;
    return EHR_OK;
}

int ehr_fused_timing_read(ehr_ctx* ctx, float* ms, int* ncalls) {
    if (!ctx || !ms || !ncalls) return fail(EHR_ERR_INVALID, "ehr_fused_timing_read: NULL argument");
    EHR_HIP(hipDeviceSynchronize());
    const size_t per = EHR_FUSED_STAGES + 1;
    const size_t n = ctx->ev_used / per;
    for (int s = 0; s < EHR_FUSED_STAGES; s++) ms[s] = 0.f;
    for (size_t c = 0; c < n; c++)
        for (int s = 0; s < EHR_FUSED_STAGES; s++) {
            float t = 0.f;
            EHR_HIP(hipEventElapsedTime(&t, ctx->ev[c * per + s], ctx->ev[c * per + s + 1]));
            ms[s] += t;
        }
    *ncalls = (int)n;
    ctx->ev_used = 0;
    return EHR_OK;
}

}  // extern "C"
