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
#include <algorithm>

#include "ehr_host.h"
#include "ehr_raster_core.h"

namespace ehr {

constexpr int RW = EHR_TILE_W + 2;  // region = tile + 1-pixel halo
constexpr int RH = EHR_TILE_H + 2;
constexpr int RN = RW * RH;          // 340
constexpr int CAND_PER_THREAD = (2 * RN + EHR_TILE_THREADS - 1) / EHR_TILE_THREADS;  // 3
constexpr int MAX_ITEMS = 1536;      // blended pairs kept per tile (all links); overflow is reported, never silent
constexpr int MAX_LINKS = 64;

struct Item {
    int packed;  // bits 0-9 q (region index of pixel0) | 10 d | 11-12 di | 13 tri1 | 14 (c1 - c0 > 0)
    int tri;     // chosen triangle (global index)
    float alpha;
};

// Deterministic block-wide exclusive offset of `cnt` items per thread; total returned through `total`.
// wave_tot: LDS int[4].  Contains two barriers.
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

__global__ void __launch_bounds__(EHR_TILE_THREADS)
fused_tile_kernel(MvpSource src, BinGeom g, const int* __restrict__ counts, const int* __restrict__ offsets,
                  const int* __restrict__ entries, int entries_cap, const int32_t* __restrict__ opp,
                  const float* __restrict__ ref, float* __restrict__ mask, float* __restrict__ tile_part,
                  int want_grad, int* __restrict__ meta) {
    __shared__ u64 key[RN];
    __shared__ float pairA[2][RN];
    __shared__ unsigned short hits[2 * RN];
    __shared__ Item items[MAX_ITEMS];
    __shared__ int seg_end[MAX_LINKS];
    __shared__ int cnt_l[MAX_LINKS];
    __shared__ float gpix[EHR_TILE_W * EHR_TILE_H];
    __shared__ int wave_tot[4];
    __shared__ float wred[4][12];

    const int tile = blockIdx.x, b = blockIdx.y;
    const int tx = tile % g.ntx, ty = tile / g.ntx;
    const int rx0 = tx * EHR_TILE_W - 1, ry0 = ty * EHR_TILE_H - 1;
    const int tid = threadIdx.x;
    const int lx = tid % EHR_TILE_W, ly = tid / EHR_TILE_W;
    const int ix = tx * EHR_TILE_W + lx, iy = ty * EHR_TILE_H + ly;
    const bool in_img = ix < g.W && iy < g.H;
    const int L = g.L, W = g.W, H = g.H;
    const int kidx = (b * g.nt + tile) * L;
    const int part_stride = 1 + 12 * L;
    float* part = tile_part + (size_t)(b * g.nt + tile) * part_stride;

    if (tid < L) cnt_l[tid] = counts[kidx + tid];
    __syncthreads();

    float acc = 0.f;
    int nitems = 0;  // uniform across the block
    const int myq = (ly + 1) * RW + (lx + 1);

    for (int l = 0; l < L; l++) {
        const int n = cnt_l[l];
        if (n == 0) {
            if (tid == 0) seg_end[l] = nitems;
            continue;
        }
        const int off = offsets[kidx + l];
        for (int i = tid; i < RN; i += EHR_TILE_THREADS) key[i] = ~0ull;
        __syncthreads();
        // ---- coverage + z-test of the link's queued triangles
        for (int base = 0; base < n; base += EHR_TILE_THREADS) {
            int i = base + tid;
            bool active = i < n && off + i < entries_cap;
            float4 p[3];
            int t = 0, link;
            if (active) {
                t = entries[off + i];
                active = src.fetch(b, t, p, link);
            }
            raster_wave<RW, RH, 12>(active, p, t, W, H, rx0, ry0, key);
        }
        __syncthreads();
        // ---- pixel pairs with different triangle ids -> dense hit list (deterministic order)
        unsigned short myhit[CAND_PER_THREAD];
        int nh = 0;
#pragma unroll
        for (int j = 0; j < CAND_PER_THREAD; j++) {
            int c = tid + j * EHR_TILE_THREADS;
            if (c < 2 * RN) {
                int d = c >= RN ? 1 : 0;
                int q = c - d * RN;
                int qx = q % RW, qy = q / RW;
                int nx = qx + 1 - d, ny = qy + d;
                pairA[d][q] = 0.f;
                bool ok = nx < RW && ny < RH;
                // both pixels inside the image
                int ax = rx0 + qx, ay = ry0 + qy, bx = rx0 + nx, by = ry0 + ny;
                ok = ok && ax >= 0 && ay >= 0 && bx < W && by < H;
                // at least one of them interior to the tile (only those can receive or own a blend we need)
                bool qi = qx >= 1 && qx <= EHR_TILE_W && qy >= 1 && qy <= EHR_TILE_H;
                bool ni = nx >= 1 && nx <= EHR_TILE_W && ny >= 1 && ny <= EHR_TILE_H;
                ok = ok && (qi || ni);
                if (ok) {
                    u64 k0 = key[q], k1 = key[ny * RW + nx];
                    unsigned t0 = (k0 == ~0ull) ? 0xffffffffu : (unsigned)k0;
                    unsigned t1 = (k1 == ~0ull) ? 0xffffffffu : (unsigned)k1;
                    if (t0 != t1) myhit[nh++] = (unsigned short)(q | (d << 15));
                }
            }
        }
        int nhits;
        int hoff = block_offset(nh, wave_tot, nhits);
        for (int j = 0; j < nh; j++) hits[hoff + j] = myhit[j];
        __syncthreads();
        // ---- silhouette analysis of the hits (restates nvdiffrast's antialias mesh kernel)
        Item mine[CAND_PER_THREAD];
        int nm = 0;
#pragma unroll
        for (int j = 0; j < CAND_PER_THREAD; j++) {
            int h = tid + j * EHR_TILE_THREADS;
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
                int vi[3] = {src.tri[3 * t], src.tri[3 * t + 1], src.tri[3 * t + 2]};
                float4 p[3], o[3];
#pragma unroll
                for (int k = 0; k < 3; k++) p[k] = src.vertex(b, l, vi[k]);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    int ov = opp[3 * t + k];
                    o[k] = ((unsigned)ov < (unsigned)src.V) ? src.vertex(b, l, ov) : p[k];
                }
                AAPair a = aa_analyze(p, o, px, py, d, chose0, W, H);
                if (a.found) {
                    pairA[d][q] = a.alpha;
                    // keep for the backward pass if the destination pixel is interior to this tile
                    int oq = (a.alpha > 0.f) ? q : nq;
                    int ox = oq % RW, oy = oq / RW;
                    bool oi = ox >= 1 && ox <= EHR_TILE_W && oy >= 1 && oy <= EHR_TILE_H;
                    if (oi && a.alpha != 0.f) {
                        float c0 = (tri0 >= 0) ? 1.f : 0.f, c1 = (tri1 >= 0) ? 1.f : 0.f;
                        if (c0 != c1) {
                            Item it;
                            it.packed = q | (d << 10) | (a.di << 11) | (a.tri1 << 13) | ((c1 > c0 ? 1 : 0) << 14);
                            it.tri = t;
                            it.alpha = a.alpha;
                            mine[nm++] = it;
                        }
                    }
                }
            }
        }
        int nfound;
        int ioff = block_offset(want_grad ? nm : 0, wave_tot, nfound);
        if (want_grad) {
            for (int j = 0; j < nm; j++) {
                int at = nitems + ioff + j;
                if (at < MAX_ITEMS)
                    items[at] = mine[j];
                else
                    meta[1] = 1;  // item overflow: reported through loss = NaN
            }
        }
        nitems = min(nitems + nfound, MAX_ITEMS);
        if (tid == 0) seg_end[l] = nitems;
        __syncthreads();
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
    }

    // ---- composite, loss, mask write (image convention: row 0 = top)
    float e2 = 0.f, gval = 0.f;
    if (in_img) {
        size_t im = ((size_t)b * H + (H - 1 - iy)) * W + ix;
        float m = acc > 1.f ? 1.f : acc;
        float e = m - ref[im];
        e2 = e * e;
        gval = (acc <= 1.f) ? 2.f * e : 0.f;
        if (mask) mask[im] = m;
    }
    gpix[tid] = gval;
    {
        float s = wave_sum(e2);
        if ((tid & 63) == 0) wred[tid >> 6][0] = s;
        __syncthreads();
        if (tid == 0) part[0] = ((wred[0][0] + wred[1][0]) + wred[2][0]) + wred[3][0];
        __syncthreads();
    }
    if (!want_grad) return;

    // ---- backward: blended pairs -> rows (x, y, w) of d loss / d MVP, per link
    int seg0 = 0;
    for (int l = 0; l < L; l++) {
        if (cnt_l[l] == 0) continue;
        const int seg1 = seg_end[l];
        float G[12];
#pragma unroll
        for (int k = 0; k < 12; k++) G[k] = 0.f;
        for (int it = seg0 + tid; it < seg1; it += EHR_TILE_THREADS) {
            Item im = items[it];
            int q = im.packed & 1023, d = (im.packed >> 10) & 1, di = (im.packed >> 11) & 3;
            int tri1 = (im.packed >> 13) & 1;
            float dc = ((im.packed >> 14) & 1) ? 1.f : -1.f;
            int nq = q + (d ? RW : 1);
            int oq = (im.alpha > 0.f) ? q : nq;
            int ox = oq % RW - 1, oy = oq / RW - 1;
            float gi = gpix[oy * EHR_TILE_W + ox];
            float dd = gi * dc;
            if (gi == 0.f || dd == 0.f) continue;
            int t = im.tri;
            int vi[3] = {src.tri[3 * t], src.tri[3 * t + 1], src.tri[3 * t + 2]};
            int i1 = (di < 2) ? di + 1 : 0;
            int i2 = (i1 < 2) ? i1 + 1 : 0;
            int v1 = vi[i1], v2 = vi[i2];
            int qx = q % RW, qy = q / RW;
            int px = rx0 + qx, py = ry0 + qy;
            if (tri1) {
                px += 1 - d;
                py += d;
            }
            float g1[3], g2[3];
            aa_pos_grad(src.vertex(b, l, v1), src.vertex(b, l, v2), px, py, d, im.alpha, dd, W, H, g1, g2);
            const float* a1 = src.verts + 3 * (size_t)v1;
            const float* a2 = src.verts + 3 * (size_t)v2;
            float h1[4] = {a1[0], a1[1], a1[2], 1.f}, h2[4] = {a2[0], a2[1], a2[2], 1.f};
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) G[4 * r + c] += g1[r] * h1[c] + g2[r] * h2[c];
        }
        seg0 = seg1;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            float s = wave_sum(G[k]);
            if ((tid & 63) == 0) wred[tid >> 6][k] = s;
        }
        __syncthreads();
        if (tid < 12) part[1 + 12 * l + tid] = ((wred[0][tid] + wred[1][tid]) + wred[2][tid]) + wred[3][tid];
        __syncthreads();
    }
}

// Fixed-order reduction of the per-tile partials.  grid = (L + 1, B): block (j, b) reduces link j's 12 numbers
// (j < L) or the loss (j == L).
__global__ void __launch_bounds__(256) fused_reduce_kernel(BinGeom g, const int* __restrict__ counts,
                                                           const float* __restrict__ tile_part,
                                                           float* __restrict__ loss, float* __restrict__ grad_mvp,
                                                           const int* __restrict__ meta) {
    __shared__ double red[256][12];
    const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, L = g.L;
    const int part_stride = 1 + 12 * L;
    double s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0.0;
    if (j == L) {
        for (int t = tid; t < g.nt; t += 256) s[0] += (double)tile_part[(size_t)(b * g.nt + t) * part_stride];
    } else {
        if (!grad_mvp) return;
        for (int t = tid; t < g.nt; t += 256) {
            if (counts[(b * g.nt + t) * L + j] == 0) continue;
            const float* p = tile_part + (size_t)(b * g.nt + t) * part_stride + 1 + 12 * j;
#pragma unroll
            for (int k = 0; k < 12; k++) s[k] += (double)p[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 12; k++) red[tid][k] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int k = 0; k < 12; k++) red[tid][k] += red[tid + o][k];
        }
        __syncthreads();
    }
    if (j == L) {
        if (tid == 0) {
            float v = (float)red[0][0];
            if (meta[1]) v = __int_as_float(0x7fc00000);  // overflow => NaN, never a silently wrong loss
            loss[b] = v;
        }
    } else if (tid < 16) {
        // rows x, y, w of the 4x4 gradient; the z row never receives gradient on this path
        int r = tid >> 2, c = tid & 3;
        float v = 0.f;
        if (r == 0) v = (float)red[0][c];
        if (r == 1) v = (float)red[0][4 + c];
        if (r == 3) v = (float)red[0][8 + c];
        grad_mvp[((size_t)b * L + j) * 16 + tid] = v;
    }
}

}  // namespace ehr

using namespace ehr;

extern "C" {

static BinGeom make_geom(int H, int W, int L) {
    BinGeom g;
    g.W = W;
    g.H = H;
    g.ntx = (W + EHR_TILE_W - 1) / EHR_TILE_W;
    g.nty = (H + EHR_TILE_H - 1) / EHR_TILE_H;
    g.nt = g.ntx * g.nty;
    g.L = L;
    return g;
}

int ehr_fused_plan(ehr_ctx* ctx, int B, int L, int V, int T, int H, int W, float slack) {
    (void)V;
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_fused_plan: ctx is NULL");
    if (B <= 0 || L <= 0 || L > MAX_LINKS || T < 0 || H <= 0 || W <= 0 || H > 32768 || W > 32768)
        return fail(EHR_ERR_INVALID, "ehr_fused_plan: bad sizes (1 <= L <= %d)", MAX_LINKS);
    if (!(slack >= 1.f)) slack = 4.f;
    BinGeom g = make_geom(H, W, L);
    size_t nkeys = (size_t)B * g.nt * L;
    if (nkeys > 0x3fffffff) return fail(EHR_ERR_INVALID, "ehr_fused_plan: too many (view, tile, link) queues");
    int rc;
    if ((rc = ctx->counts.reserve((2 * nkeys + 4) * sizeof(int)))) return rc;
    if ((rc = ctx->offsets.reserve(nkeys * sizeof(int)))) return rc;
    // queue storage: every triangle lands in >= 1 tile; micro-triangles average ~1.3 tiles, big ones more.
    size_t want = (size_t)((double)slack * (double)B * (double)std::max(T, 1)) + 65536;
    want = std::min(want, (size_t)0x7fffffff);
    if (want > ctx->entries_cap) {
        if ((rc = ctx->entries.reserve(want * sizeof(int)))) return rc;
        ctx->entries_cap = want;
    }
    if ((rc = ctx->tile_part.reserve((size_t)B * g.nt * (1 + 12 * (size_t)L) * sizeof(float)))) return rc;
    ctx->pB = B;
    ctx->pL = L;
    ctx->pT = T;
    ctx->pH = H;
    ctx->pW = W;
    return EHR_OK;
}

int ehr_render_mask_loss(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* tri_link,
                         const int32_t* opp, const float* mvp, const float* ref, int B, int L, int V, int T, int H,
                         int W, float* mask, float* loss, float* grad_mvp, void* stream_) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_render_mask_loss: ctx is NULL");
    if (!verts || !tris || !tri_link || !opp || !mvp || !ref || !loss)
        return fail(EHR_ERR_INVALID, "ehr_render_mask_loss: NULL tensor");
    if (ctx->pB != B || ctx->pL != L || ctx->pT != T || ctx->pH != H || ctx->pW != W)
        return fail(EHR_ERR_INVALID, "ehr_render_mask_loss: shape differs from the planned one; call ehr_fused_plan first");
    hipStream_t stream = (hipStream_t)stream_;
    BinGeom g = make_geom(H, W, L);
    const int nkeys = B * g.nt * L;
    int* counts = (int*)ctx->counts.ptr;
    int* cursors = counts + nkeys;
    int* meta = counts + 2 * nkeys;
    int* offsets = (int*)ctx->offsets.ptr;
    int* entries = (int*)ctx->entries.ptr;
    const int ecap = (int)std::min(ctx->entries_cap, (size_t)0x7fffffff);
    MvpSource src;
    src.verts = verts;
    src.tri = tris;
    src.tri_link = tri_link;
    src.mvp = mvp;
    src.V = V;
    src.T = T;
    src.L = L;

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
    EHR_HIP(hipMemsetAsync(counts, 0, ((size_t)2 * nkeys + 4) * sizeof(int), stream));
    dim3 bgrid((T + 255) / 256, B);
    if (T > 0) {
        bin_kernel<MvpSource, 1, false><<<bgrid, 256, 0, stream>>>(src, g, counts, cursors, offsets, nullptr, 0, meta);
        EHR_LAUNCH_CHECK();
    }
    if (ev) EHR_HIP(hipEventRecord(ev[1], stream));
    bin_alloc_kernel<<<(nkeys + 255) / 256, 256, 0, stream>>>(counts, offsets, nkeys, meta);
    EHR_LAUNCH_CHECK();
    if (ev) EHR_HIP(hipEventRecord(ev[2], stream));
    if (T > 0) {
        bin_kernel<MvpSource, 1, true><<<bgrid, 256, 0, stream>>>(src, g, counts, cursors, offsets, entries, ecap, meta);
        EHR_LAUNCH_CHECK();
    }
    if (ev) EHR_HIP(hipEventRecord(ev[3], stream));
    dim3 tgrid(g.nt, B);
    fused_tile_kernel<<<tgrid, EHR_TILE_THREADS, 0, stream>>>(src, g, counts, offsets, entries, ecap, opp, ref, mask,
                                                             (float*)ctx->tile_part.ptr, grad_mvp ? 1 : 0, meta);
    EHR_LAUNCH_CHECK();
    if (ev) EHR_HIP(hipEventRecord(ev[4], stream));
    dim3 rgrid(L + 1, B);
    fused_reduce_kernel<<<rgrid, 256, 0, stream>>>(g, counts, (const float*)ctx->tile_part.ptr, loss, grad_mvp, meta);
    EHR_LAUNCH_CHECK();
    if (ev) EHR_HIP(hipEventRecord(ev[5], stream));
    return EHR_OK;
}

int ehr_fused_status(ehr_ctx* ctx) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_fused_status: ctx is NULL");
    if (!ctx->counts.ptr || ctx->pB == 0) return EHR_OK;
    EHR_HIP(hipDeviceSynchronize());
    BinGeom g = make_geom(ctx->pH, ctx->pW, ctx->pL);
    const size_t nkeys = (size_t)ctx->pB * g.nt * ctx->pL;
    int meta[4] = {0, 0, 0, 0};
    EHR_HIP(hipMemcpy(meta, (int*)ctx->counts.ptr + 2 * nkeys, sizeof(meta), hipMemcpyDeviceToHost));
    if (meta[1])
        return fail(EHR_ERR_OVERFLOW, "fused path: a bin queue or a tile's blend list overflowed (%d queued, capacity %zu); "
                                      "re-plan with a larger slack", meta[0], ctx->entries_cap);
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
