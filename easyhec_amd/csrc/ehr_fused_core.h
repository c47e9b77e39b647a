// ehr_fused_core.h -- pieces of the fused op's launch chain (ehr_vbuf.hip) that are not rasterization: fixed-point
// accumulators, the head / tail descriptors of a solver step and the finish stage (accumulators -> loss / grad_mvp ->
// pose backward -> Adam), which runs inside the composite kernel's last-arriving workgroup.
#pragma once
#include "ehr_host.h"
#include "ehr_pose_core.h"
#include "ehr_raster_core.h"

namespace ehr {

// Per-view sums (frame loss, 12 gradient numbers per link) are accumulated in 64-bit FIXED POINT with integer atomics:
// integer addition is associative, so the result does not depend on which workgroup adds first -- bit-reproducible like
// a fixed-order reduction, but without a reduction pass over all tiles.  Scale 2^32: addends are rounded to 2.3e-10
// (absolute), sums up to +-2.1e9 fit; larger magnitudes raise the overflow flag (loss = NaN), never wrap silently.
// The flag is raised with an atomic: the finish stage reads it inside the same launch.
#define EHR_FIX_SCALE 4294967296.0
__device__ __forceinline__ long long fix_of(float v) { return __double2ll_rn((double)v * EHR_FIX_SCALE); }
__device__ __forceinline__ void fix_add(long long* acc, float v, int* meta) {
    if (!(fabsf(v) < 1.0e9f)) {  // also catches NaN
        atomicOr(&meta[EHR_META_OVERFLOW], 1);
        return;
    }
    if (v != 0.f) atomicAdd((unsigned long long*)acc, (unsigned long long)fix_of(v));
}
// acc += fix(v) - cached: the caller has bound a constant whose fixed-point value `cached` is already part of the
// view's total (the sum of ref^2 over a tile no link touches).  Integer arithmetic: the same bits as adding fix(v) to
// an accumulator that never held `cached`.
__device__ __forceinline__ void fix_add_delta(long long* acc, float v, long long cached, int* meta) {
    if (!(fabsf(v) < 1.0e9f)) {
        atomicOr(&meta[EHR_META_OVERFLOW], 1);
        return;
    }
    const long long d = ((v != 0.f) ? fix_of(v) : 0ll) - cached;
    if (d != 0) atomicAdd((unsigned long long*)acc, (unsigned long long)d);
}
__device__ __forceinline__ float fix_get(long long q) { return (float)((double)q * (1.0 / EHR_FIX_SCALE)); }
// accumulators are read inside the launch that adds to them (finish stage in the last-arriving workgroup): agent-scope
// loads, which are served by the memory side the atomics were performed on, never by this CU's L1
__device__ __forceinline__ long long acc_load(const long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// facc: per view 12 numbers per link, then `nls` partial sums of the frame loss, `lstride` i64 apart (several slots, one
// 128-byte line each, so that thousands of tiles do not serialise on one address; integer sums, so the split does not
// change the result).  vtot (optional): per view a constant that belongs to the frame loss (the bound reference mask's
// cached part).  Called by all 256 threads of one workgroup after every other workgroup's atomics were performed.
template <bool TAIL>
__device__ __forceinline__ void finish_body(const BinGeom& g, int B, const long long* __restrict__ facc,
                                            const long long* __restrict__ vtot, float* __restrict__ loss,
                                            float* __restrict__ grad_mvp, int* __restrict__ meta, const StepTail& tail,
                                            int nls, int* __restrict__ lbox, int lstride, float* vloss /* LDS [256] */,
                                            double (*S)[17] /* LDS [4][17] */, float* red_lds /* LDS [8] */,
                                            float (*Js)[16] /* LDS [6][16] */) {
    const int tid = threadIdx.x, L = g.L;
    const int acc_stride = 12 * L + nls * lstride;
    if (lbox)  // the links' screen boxes start "empty" in the next step
        for (int i = tid; i < 16 * B * L; i += 256) lbox[i] = (i & 2) ? INT_MIN : INT_MAX;  // 16 ints (one line) per box
    AdamState st;
    if (TAIL && !tail.defer_adam) st = pose_adam_fetch(tail.dof, tail.m, tail.v, tail.step);
    auto view_loss_slow = [&](int b) {
        long long s = vtot ? vtot[b] : 0ll;
        for (int k = 0; k < nls; k++) s += acc_load(&facc[(size_t)b * acc_stride + 12 * L + k * lstride]);
        return fix_get(s);
    };
    // Frame losses.  nls == 32: one slot per lane, half a wave per view -- a single round trip instead of 32 dependent
    // adds -- and the lane that ends up with a view's sum stores loss[b] and keeps it as ITS share of sum_b loss_b: no
    // hand-over through LDS, no barrier, so the reads below (gradient accumulators, link poses, Jacobian) travel with
    // these instead of after them.
    const bool bad = __hip_atomic_load(&meta[EHR_META_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const float nanv = __int_as_float(0x7fc00000);  // overflow => NaN, never a silently wrong loss
    double la_mine = 0.0;
    if (nls == 32) {
        const int k = tid & 31;
        for (int base = 0; base < B; base += 32) {  // four groups of 8 views per round trip
            long long s4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int b = base + 8 * j + (tid >> 5);
                s4[j] = (b < B) ? acc_load(&facc[(size_t)b * acc_stride + 12 * L + k * lstride]) : 0;
            }
            // (Adam's bias corrections while the slots are on their way: they need the step count only)
            if (TAIL && base == 0 && !tail.defer_adam) pose_adam_bias(st, tail.lr, tail.b1, tail.b2);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int b = base + 8 * j + (tid >> 5);
                if (base + 8 * j >= B) break;  // (workgroup-uniform)
                long long s = s4[j];
                s += wave_xor<16>(s);  // (the 32 slots of a view: half a wave)
                s += wave_xor<8>(s);
                s += wave_xor<4>(s);
                s += wave_xor<2>(s);
                s += wave_xor<1>(s);
                if (k == 0 && b < B) {
                    const float lv = bad ? nanv : fix_get(s + (vtot ? vtot[b] : 0ll));
                    loss[b] = lv;
                    la_mine += (double)lv;
                }
            }
        }
    } else {
        for (int b = tid; b < B; b += 256) {
            const float lv = bad ? nanv : view_loss_slow(b);
            loss[b] = lv;
            la_mine += (double)lv;
        }
    }
    auto grad16 = [&](int bl, float* G) {
        // rows x, y, w of the 4x4 gradient; the z row never receives gradient on this path
        const int b = bl / L, l = bl - b * L;
        const long long* a = facc + (size_t)b * acc_stride + 12 * l;
        long long q[12];
#pragma unroll
        for (int e = 0; e < 12; e++) q[e] = acc_load(a + e);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int r = e >> 2, c = e & 3;
            float v = 0.f;
            if (r != 2) v = fix_get(q[4 * (r == 3 ? 2 : r) + c]);
            G[e] = bad ? nanv : v;
        }
    };
    if (!TAIL) {
        if (grad_mvp)
            for (int i = tid; i < B * L; i += 256) {
                float G[16];
                grad16(i, G);
#pragma unroll
                for (int e = 0; e < 16; e++) grad_mvp[(size_t)i * 16 + e] = G[e];
            }
        return;
    }
    // One dependent round trip in total: the optimiser state and the Jacobian are requested up front, the gradients
    // come straight from the accumulators, and Adam reads the 8 reduced floats back from LDS.
    pose_backward_block_t(
        [&](int i, float* G) {
            grad16(i, G);
            if (grad_mvp)
#pragma unroll
                for (int e = 0; e < 16; e++) grad_mvp[(size_t)i * 16 + e] = G[e];
        },
        [&](int) { return 0.f; }, tail.K, tail.link_poses, tail.tc_jac, B, L, g.H, g.W, tail.n, tail.f, tail.red, S, red_lds,
        &la_mine, nls == 32, Js);
    __syncthreads();
    if (!tail.defer_adam)
        pose_adam_apply(st, tail.dof, tail.m, tail.v, tail.step, red_lds, tail.lr, tail.b1, tail.b2, tail.eps, tail.wd,
                        tail.loss_out, tail.grad_out);
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
    int* hist_row;  // [1] row of `history` this step's pose goes to; advanced here (its own counter, not Adam's)
    float* history;
    int history_rows;
    float n, f;
    // A REPORTED step (overflow: loss NaN, pose and optimiser untouched -- on every rank of a data-parallel job, the NaN
    // travels with the exchanged sum) has written the unchanged pose to a row; the next step's head sees that Adam's
    // counter has not moved since the previous head and takes the same row again: the history keeps one row per EFFECTIVE
    // step with no host-side rewind (which had to guess which rows to give back).
    const int* adam_step;  // [1] the optimiser's step counter
    int* hstate;           // [2] ctx->vb_hstate
};


}  // namespace ehr
