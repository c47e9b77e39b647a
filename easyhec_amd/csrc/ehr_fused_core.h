// ehr_fused_core.h -- pieces shared by the two launch chains of the fused op (the LDS-tile path in ehr_fused.hip and the
// visibility-buffer path in ehr_vbuf.hip): fixed-point accumulators, the head / tail descriptors of a solver step and
// the one-workgroup finish kernel (accumulators -> loss / grad_mvp -> pose backward -> Adam).
#pragma once
#include "ehr_host.h"
#include "ehr_pose_core.h"
#include "ehr_raster_core.h"

namespace ehr {

// Per-view sums (frame loss, 12 gradient numbers per link) are accumulated in 64-bit FIXED POINT with integer atomics:
// integer addition is associative, so the result does not depend on which workgroup adds first -- bit-reproducible like
// a fixed-order reduction, but without a reduction pass over all tiles.  Scale 2^32: addends are rounded to 2.3e-10
// (absolute), sums up to +-2.1e9 fit; larger magnitudes raise the overflow flag (loss = NaN), never wrap silently.
#define EHR_FIX_SCALE 4294967296.0
__device__ __forceinline__ void fix_add(long long* acc, float v, int* meta) {
    if (!(fabsf(v) < 1.0e9f)) {  // also catches NaN
        meta[EHR_META_OVERFLOW] = 1;
        return;
    }
    if (v != 0.f) atomicAdd((unsigned long long*)acc, (unsigned long long)__double2ll_rn((double)v * EHR_FIX_SCALE));
}
__device__ __forceinline__ float fix_get(long long q) { return (float)((double)q * (1.0 / EHR_FIX_SCALE)); }

// Last stage, ONE workgroup: fixed-point accumulators -> loss[B] and grad_mvp[B,L,16]; with TAIL also the rest of a
// solver step (d sum(loss) / d dof = the 8 floats a data-parallel job all-reduces, then Adam unless deferred).
struct StepTail {  // what the solver-step form needs (all device pointers)
    const float* K;
    const float* link_poses;
    const float* tc_jac;
    float* red;
    float* dof;
    float* m;
    float* v;
    int* step;
    float* loss_out;
    float* grad_out;
    float n, f, lr, b1, b2, eps, wd;
    int defer_adam;
};

template <bool TAIL>
__global__ void __launch_bounds__(256) fused_finish_kernel(BinGeom g, int B, const long long* __restrict__ facc,
                                                           float* __restrict__ loss, float* __restrict__ grad_mvp,
                                                           const int* __restrict__ meta, StepTail tail, int nls,
                                                           int* __restrict__ lbox, int lstride) {
    // per view: 12 numbers per link, then `nls` partial sums of the frame loss (several slots so that thousands of
    // tiles do not serialise on one address; integer sums, so the split does not change the result)
    const int tid = threadIdx.x, L = g.L;
    // (`lstride` i64 apart: the visibility-buffer chain gives every slot a 128-byte line of its own)
    const int acc_stride = 12 * L + nls * lstride;
    __shared__ float vloss[256];  // frame loss of up to 256 views per pass (summed once, read many times below)
    if (lbox)  // visibility-buffer chain: the links' screen boxes start "empty" in the next step
        for (int i = tid; i < 16 * B * L; i += 256) lbox[i] = (i & 2) ? INT_MIN : INT_MAX;  // 16 ints (one line) per box
    auto view_loss_slow = [&](int b) {
        long long s = 0;
        for (int k = 0; k < nls; k++) s += facc[(size_t)b * acc_stride + 12 * L + k * lstride];
        return fix_get(s);
    };
    if (nls == 32) {  // one slot per lane, half a wave per view: a single round trip instead of 32 dependent adds
        for (int base = 0; base < min(B, 256); base += 8) {
            const int b = base + (tid >> 5), k = tid & 31;
            long long s = (b < B) ? facc[(size_t)b * acc_stride + 12 * L + k * lstride] : 0;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (k == 0 && b < B && b < 256) vloss[b] = fix_get(s);
        }
    } else if (tid < B) {
        vloss[tid] = view_loss_slow(tid);
    }
    __syncthreads();
    auto view_loss = [&](int b) { return b < 256 ? vloss[b] : view_loss_slow(b); };
    const bool bad = meta[EHR_META_OVERFLOW] != 0;  // overflow => NaN, never a silently wrong loss
    const float nanv = __int_as_float(0x7fc00000);
    for (int i = tid; i < B; i += 256) loss[i] = bad ? nanv : view_loss(i);
    if (grad_mvp) {
        for (int i = tid; i < B * L * 16; i += 256) {
            // rows x, y, w of the 4x4 gradient; the z row never receives gradient on this path
            const int bl = i >> 4, e = i & 15, r = e >> 2, c = e & 3;
            const int b = bl / L, l = bl - b * L;
            float v = 0.f;
            if (r != 2) v = fix_get(facc[(size_t)b * acc_stride + 12 * l + 4 * (r == 3 ? 2 : r) + c]);
            grad_mvp[i] = bad ? nanv : v;
        }
    }
    if (TAIL) {
        // One dependent round trip in total: the optimiser state and the Jacobian are requested up front, the
        // gradients come straight from the accumulators (the stores above are fire-and-forget), and Adam reads the
        // 8 reduced floats back from LDS.
        __shared__ double S[256][16];
        __shared__ double lsum[256];
        __shared__ float red_lds[8];
        AdamState st;
        if (!tail.defer_adam) st = pose_adam_fetch(tail.dof, tail.m, tail.v, tail.step);
        pose_backward_block_t(
            [&](int i, float* G) {
                const int b = i / L, l = i - b * L;
                const long long* a = facc + (size_t)b * acc_stride + 12 * l;
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int r = e >> 2, c = e & 3;
                    float v = 0.f;
                    if (r != 2) v = fix_get(a[4 * (r == 3 ? 2 : r) + c]);
                    G[e] = bad ? nanv : v;
                }
            },
            [&](int b) { return bad ? nanv : view_loss(b); }, tail.K, tail.link_poses,
            tail.tc_jac, B, L, g.H, g.W, tail.n, tail.f, tail.red, S, lsum, red_lds);
        __syncthreads();
        if (!tail.defer_adam)
            pose_adam_apply(st, tail.dof, tail.m, tail.v, tail.step, red_lds, tail.lr, tail.b1, tail.b2, tail.eps, tail.wd,
                            tail.loss_out, tail.grad_out);
    }
}


static inline BinGeom make_geom(int H, int W, int L) {
    BinGeom g;
    g.W = W;
    g.H = H;
    g.ntx = (W + EHR_TILE_W - 1) / EHR_TILE_W;
    g.nty = (H + EHR_TILE_H - 1) / EHR_TILE_H;
    g.nt = g.ntx * g.nty;
    g.L = L;
    return g;
}

struct StepHead {  // inputs of the merged first stage (pose forward inside the vertex kernel)
    const float* dof;
    const float* K;
    const float* link_poses;
    float* tc_jac;
    const int* step;
    float* history;
    int history_rows;
    float n, f;
};


}  // namespace ehr
