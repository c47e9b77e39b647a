// ehr_score.hip -- next-best-view scoring of the space explorer, restating
//   /root/reference/easyhec/modeling/models/rb_solve/space_explorer.py:152-165
//     for every candidate joint configuration q: render the non-antialiased robot mask under each of the S sampled
//     camera poses (render_api.py:155-192 -> :70-96 -> nvdiffrast_renderer.py:50-72: all link meshes merged into one
//     mesh, one rasterize, mask = rast[..., 2] > 0), stack, torch.var over the S masks per pixel, sum over pixels.
//
// For binary masks the unbiased per-pixel variance is c (S - c) / (S (S - 1)) with c = number of poses that cover the
// pixel, so the score is an INTEGER sum: score[q] = sum_pixels c (S - c).  The reference does 10 000 full-image
// renders + a [S, H*W] float reduction per exploration round; here one workgroup owns a (q, 32x8 tile) pair, rasterizes
// the tile under each pose into an LDS depth/id buffer (same binning + balanced rasterizer as the solver path), keeps
// c in a register and adds its tile's sum with one 64-bit integer atomic -- no image ever reaches HBM (unless the
// caller asks for the count image), and the result is exact and order independent.
#include <algorithm>

#include "ehr_host.h"
#include "ehr_raster_core.h"

namespace ehr {

constexpr int SCORE_MAX_S = 255;  // counts are kept (and optionally written) as 8-bit values
// meta words of the (q, tile) work lists (bin_alloc_kernel owns EHR_META_NWORK / _NWORK_SLOW for its per-view lists)
constexpr int META_PAIRS = 4, META_PAIRS_SLOW = 5;

// posc[view][v] = MVP[view, vert_link[v]] * [x, y, z, 1]   (nvdiffrast_utils.py:14-18; render_api.py:84-90)
__global__ void __launch_bounds__(256) score_vertex_kernel(const float* __restrict__ verts,
                                                           const int32_t* __restrict__ vert_link,
                                                           const float* __restrict__ mvp, int V, int L,
                                                           float4* __restrict__ posc) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (v >= V) return;
    const int l = vert_link ? vert_link[v] : 0;
    float4 o = make_float4(0.f, 0.f, 0.f, -1.f);
    if ((unsigned)l < (unsigned)L) o = transform_vertex(mvp + ((size_t)b * L + l) * 16, verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]);
    posc[(size_t)b * V + v] = o;
}

// One thread per (q, tile): does any of the S views queue a triangle there?  Two work lists, as in the solver path:
// worklist[0 .. nwork) lean pairs, worklist[nitems .. nitems + nwork_slow) pairs holding a triangle that needs the
// clipping / 64-bit rasterizer in at least one view.
__global__ void __launch_bounds__(256) score_work_kernel(const int* __restrict__ counts, const int* __restrict__ tile_slow,
                                                         int Q, int S, int nt, int* __restrict__ worklist,
                                                         int* __restrict__ meta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nitems = Q * nt;
    int any = 0, slow = 0;
    if (i < nitems) {
        const int q = i / nt, tile = i - q * nt;
        for (int s = 0; s < S; s++) {
            const size_t idx = (size_t)(q * S + s) * nt + tile;
            any |= counts[idx];
            slow |= tile_slow[idx];
        }
    }
    const bool is_slow = any != 0 && slow != 0, is_lean = any != 0 && slow == 0;
    const int lane = threadIdx.x & 63;
    const u64 ne = __ballot(is_lean), ns = __ballot(is_slow);
    int wbase = 0, sbase = 0;
    if (lane == 0) {
        if (ne) wbase = atomicAdd(&meta[META_PAIRS], __popcll(ne));
        if (ns) sbase = atomicAdd(&meta[META_PAIRS_SLOW], __popcll(ns));
    }
    wbase = __shfl(wbase, 0, 64);
    sbase = __shfl(sbase, 0, 64);
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (is_lean) worklist[wbase + __popcll(ne & lt)] = i;
    if (is_slow) worklist[nitems + sbase + __popcll(ns & lt)] = i;
}

// One workgroup per work-list entry (q, tile); thread = pixel of the tile.
template <bool SLOW>
__global__ void __launch_bounds__(EHR_TILE_THREADS, SLOW ? 1 : 4)
score_tile_kernel(ClipSource src, BinGeom g, const int* __restrict__ counts, const int* __restrict__ offsets,
                  const int4* __restrict__ entries, int entries_cap, const int* __restrict__ worklist, int S, int q0,
                  unsigned long long* __restrict__ score, unsigned char* __restrict__ count_img,
                  int* __restrict__ meta) {
    __shared__ u64 key[EHR_TILE_W * EHR_TILE_H];
    __shared__ BlockRaster wscratch;
    __shared__ int cnt_s[SCORE_MAX_S + 1];
    __shared__ int off_s[SCORE_MAX_S + 1];
    __shared__ int wred[EHR_TILE_THREADS / 64];
    const int tid = threadIdx.x;
    const int item = worklist[blockIdx.x];  // q_local * nt + tile
    const int ql = item / g.nt, tile = item - ql * g.nt;
    const int tx = tile % g.ntx, ty = tile / g.ntx;
    const int rx0 = tx * EHR_TILE_W, ry0 = ty * EHR_TILE_H;
    if (tid < S) {
        const size_t kidx = (size_t)(ql * S + tid) * g.nt + tile;
        cnt_s[tid] = counts[kidx];
        off_s[tid] = offsets[kidx];
    }
    __syncthreads();
    // Poses are rasterized one after the other into the same LDS buffer.  Each pass is a short dependent chain (queue
    // entries -> vertex gather -> setup -> walk), so the loads of the following passes are issued ahead: entries two
    // passes ahead, vertices one pass ahead.
    auto next_active = [&](int s) {
        do s++;
        while (s < S && cnt_s[s] == 0);
        return s;
    };
    auto load_entry = [&](int s) {
        const int n = min(cnt_s[s], max(entries_cap - off_s[s], 0));
        return (tid < n) ? entries[off_s[s] + tid] : make_int4(0, 0, 0, 0);
    };
    auto load_verts = [&](int s, RoundZero& r) {
        const int n = min(cnt_s[s], max(entries_cap - off_s[s], 0));
        if (tid < n) {
            const float4* pv = src.verts(ql * S + s);
            r.p0 = pv[r.e.y];
            r.p1 = pv[r.e.z];
            r.p2 = pv[r.e.w];
        }
    };
    int c = 0;
    int s = next_active(-1);
    RoundZero cur, nxt;
    if (s < S) {
        cur.e = load_entry(s);
        load_verts(s, cur);
    }
    int sn = (s < S) ? next_active(s) : S;
    if (sn < S) nxt.e = load_entry(sn);
    while (s < S) {  // block-uniform
        int snn = S;
        int4 e2 = make_int4(0, 0, 0, 0);
        if (sn < S) {
            load_verts(sn, nxt);
            snn = next_active(sn);
            if (snn < S) e2 = load_entry(snn);
        }
        const int off = off_s[s];
        const int n = min(cnt_s[s], max(entries_cap - off, 0));
        key[tid] = ~0ull;
        __syncthreads();
        raster_queue<EHR_TILE_W, EHR_TILE_H, SLOW, 2>(src, ql * S + s, entries + off, n, g.W, g.H, rx0, ry0, key,
                                                      &wscratch, meta, cur);
        __syncthreads();
        // nvdiffrast_renderer.py:70: mask = rast_out[0, :, :, 2] > 0 (z/w of the nearest fragment; 0 where empty)
        const u64 k = key[tid];
        if (k != ~0ull && ord_unkey((unsigned)(k >> 32)) > 0.f) c++;
        __syncthreads();
        s = sn;
        sn = snn;
        cur = nxt;
        nxt.e = e2;
    }
    int v = c * (S - c);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((tid & 63) == 0) wred[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        const int tot = (wred[0] + wred[1]) + (wred[2] + wred[3]);
        if (tot) atomicAdd(&score[q0 + ql], (unsigned long long)tot);
    }
    if (count_img) {
        const int ix = rx0 + tid % EHR_TILE_W, iy = ry0 + tid / EHR_TILE_W;
        if (ix < g.W && iy < g.H && c)  // image convention: row 0 = top (nvdiffrast_renderer.py:71 flip)
            count_img[((size_t)(q0 + ql) * g.H + (g.H - 1 - iy)) * g.W + ix] = (unsigned char)c;
    }
}

}  // namespace ehr

using namespace ehr;

extern "C" int ehr_mask_variance(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* vert_link,
                                 const float* mvp, int Q, int S, int L, int V, int T, int H, int W, int64_t* score,
                                 uint8_t* count, int chunk_views, void* stream_) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_mask_variance: ctx is NULL");
    if (!verts || !tris || !mvp || !score) return fail(EHR_ERR_INVALID, "ehr_mask_variance: NULL tensor");
    if (Q <= 0 || L <= 0 || V <= 0 || T <= 0 || H <= 0 || W <= 0) return fail(EHR_ERR_INVALID, "ehr_mask_variance: bad sizes");
    if (S < 1 || S > SCORE_MAX_S) return fail(EHR_ERR_INVALID, "ehr_mask_variance: S must be in [1, %d]", SCORE_MAX_S);
    if (H > 32768 || W > 32768) return fail(EHR_ERR_INVALID, "ehr_mask_variance: resolution above 32768 is unsupported");
    hipStream_t stream = (hipStream_t)stream_;
    {   // the coverage-only chain on the solver's cluster / job machinery, where the call allows it (ehr_vbuf.hip)
        int handled = 0;
        const int rc2 = vbuf_score(ctx, verts, tris, vert_link, mvp, Q, S, L, V, T, H, W, (long long*)score, count, stream, &handled);
        if (rc2) return rc2;
        if (handled) return EHR_OK;
    }
    BinGeom g;
    g.W = W;
    g.H = H;
    g.ntx = (W + EHR_TILE_W - 1) / EHR_TILE_W;
    g.nty = (H + EHR_TILE_H - 1) / EHR_TILE_H;
    g.nt = g.ntx * g.nty;
    g.L = 1;
    // candidate configurations per pass: bounded by the scratch a pass needs (clip-space vertices dominate)
    if (chunk_views <= 0) chunk_views = 512;
    const int Qc = std::max(1, std::min(Q, chunk_views / S));
    const size_t views = (size_t)Qc * S, nkeys = views * g.nt;
    if (nkeys > (size_t)0x7fffffff / 4) return fail(EHR_ERR_INVALID, "ehr_mask_variance: chunk too large");
    int rc;
    // counts | cursors | tile_slow | meta          offsets | worklist (2 x Qc x nt)
    if ((rc = ctx->sc_counts.reserve((3 * nkeys + 2 * EHR_META_INTS) * sizeof(int)))) return rc;
    if ((rc = ctx->sc_offsets.reserve((nkeys + 2 * (size_t)Qc * g.nt) * sizeof(int)))) return rc;
    if ((rc = ctx->sc_posc.reserve(views * (size_t)V * sizeof(float4)))) return rc;
    if (ctx->sc_entries_cap == 0) {
        size_t want = std::max((size_t)1 << 20, views * (size_t)T);
        if ((rc = ctx->sc_entries.reserve(want * sizeof(int4)))) return rc;
        ctx->sc_entries_cap = want;
    }
    int* counts = (int*)ctx->sc_counts.ptr;
    int* cursors = counts + nkeys;
    int* tile_slow = counts + 2 * nkeys;
    int* meta = counts + 3 * nkeys;          // reset every pass: totals and work-list lengths
    int* smeta = meta + EHR_META_INTS;       // never reset inside a call: sticky overflow flag (+ profiling counters)
    int* offsets = (int*)ctx->sc_offsets.ptr;
    int* worklist = offsets + nkeys;
    float4* posc = (float4*)ctx->sc_posc.ptr;

    EHR_HIP(hipMemsetAsync(score, 0, (size_t)Q * sizeof(int64_t), stream));
    if (count) EHR_HIP(hipMemsetAsync(count, 0, (size_t)Q * H * W, stream));
    EHR_HIP(hipMemsetAsync(smeta, 0, EHR_META_INTS * sizeof(int), stream));
    for (int q0 = 0; q0 < Q; q0 += Qc) {
        const int qn = std::min(Qc, Q - q0);
        const int nv = qn * S;                  // views of this pass
        const int nk = nv * g.nt;
        EHR_HIP(hipMemsetAsync(counts, 0, (3 * nkeys + EHR_META_INTS) * sizeof(int), stream));
        ClipSource src;
        src.pos = posc;
        src.tri = tris;
        src.tri_link = nullptr;
        src.ranges = nullptr;
        src.V = V;
        src.T = T;
        src.L = 1;
        src.image_stride = V;
        score_vertex_kernel<<<dim3((V + 255) / 256, nv), 256, 0, stream>>>(verts, vert_link, mvp + (size_t)q0 * S * L * 16,
                                                                          V, L, posc);
        EHR_LAUNCH_CHECK();
        dim3 bgrid((T + 255) / 256, nv);
        bin_kernel<0, false><<<bgrid, 256, 0, stream>>>(src, g, counts, cursors, offsets, nullptr, 0, meta, tile_slow);
        EHR_LAUNCH_CHECK();
        bin_alloc_kernel<<<(nk + 255) / 256, 256, 0, stream>>>(counts, offsets, nullptr, nullptr, nullptr, nk, 1, meta);
        EHR_LAUNCH_CHECK();
        score_work_kernel<<<(qn * g.nt + 255) / 256, 256, 0, stream>>>(counts, tile_slow, qn, S, g.nt, worklist, meta);
        EHR_LAUNCH_CHECK();
        // size read-back (the one synchronisation per pass): queue storage needed and the work-list lengths
        EHR_HIP(hipMemcpyAsync(ctx->host_pinned, meta, 6 * sizeof(int), hipMemcpyDeviceToHost, stream));
        EHR_HIP(hipStreamSynchronize(stream));
        const size_t total = (size_t)ctx->host_pinned[EHR_META_TOTAL];
        const int nwork = ctx->host_pinned[META_PAIRS], nslow = ctx->host_pinned[META_PAIRS_SLOW];
        if (total > (size_t)0x7fffffff) return fail(EHR_ERR_INVALID, "ehr_mask_variance: %zu queue entries in one pass; lower chunk_views", total);
        if (total > ctx->sc_entries_cap) {
            size_t want = total + total / 4;
            if ((rc = ctx->sc_entries.reserve(want * sizeof(int4)))) return rc;
            ctx->sc_entries_cap = want;
        }
        int4* entries = (int4*)ctx->sc_entries.ptr;
        const int ecap = (int)std::min(ctx->sc_entries_cap, (size_t)0x7fffffff);
        bin_kernel<0, true><<<bgrid, 256, 0, stream>>>(src, g, counts, cursors, offsets, entries, ecap, smeta, nullptr);
        EHR_LAUNCH_CHECK();
        if (nwork > 0) {
            score_tile_kernel<false><<<nwork, EHR_TILE_THREADS, 0, stream>>>(src, g, counts, offsets, entries, ecap, worklist,
                                                                            S, q0, (unsigned long long*)score, count, smeta);
            EHR_LAUNCH_CHECK();
        }
        if (nslow > 0) {
            score_tile_kernel<true><<<nslow, EHR_TILE_THREADS, 0, stream>>>(src, g, counts, offsets, entries, ecap,
                                                                           worklist + (size_t)qn * g.nt, S, q0,
                                                                           (unsigned long long*)score, count, smeta);
            EHR_LAUNCH_CHECK();
        }
    }
    EHR_HIP(hipMemcpyAsync(ctx->host_pinned, smeta, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
    EHR_HIP(hipStreamSynchronize(stream));
    const int sticky = ctx->host_pinned[EHR_META_OVERFLOW];
    if (sticky) return fail(EHR_ERR_OVERFLOW, "ehr_mask_variance: internal queue overflow (results are invalid)");
    return EHR_OK;
}
