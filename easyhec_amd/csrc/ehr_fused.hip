// ehr_fused.hip -- host entry points of the fused hot path (include/ehr.h: ehr_fused_plan, ehr_render_mask_loss,
// ehr_solver_step, ehr_fused_status, ehr_graph_*, ehr_fused_timing*): B views x L links -> composite mask, per-frame SSE
// loss and d(loss_b)/d(MVP[b,l]), restating
//   /root/reference/easyhec/modeling/models/rb_solve/rb_solver.py:60-72   (per-link render, sum, clamp, SSE)
//   /root/reference/easyhec/structures/nvdiffrast_renderer.py:33-47        (rasterize -> interpolate -> antialias -> flip)
//   /root/reference/easyhec/utils/nvdiffrast_utils.py:14-18                (transform_pos)
// and the backward of all of it down to the 4x4 matrix of every (view, link).  The kernels live in ehr_vbuf.hip (the
// visibility-buffer chain); round 1's queue-based LDS-tile chain that used to sit here was retired in round 3 (one hot
// path, one implementation; it is in the git history).
#include <stdlib.h>

#include <algorithm>

#include "ehr_fused_core.h"

namespace ehr {
constexpr int MAX_LINKS = 32;
}  // namespace ehr

using namespace ehr;

extern "C" {

int ehr_fused_plan(ehr_ctx* ctx, int B, int L, int V, int T, int H, int W, float slack, const float* verts,
                   const int32_t* tris, const int32_t* tri_link, const int32_t* opp) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_fused_plan: ctx is NULL");
    if (B <= 0 || L <= 0 || L > MAX_LINKS || V < 0 || T < 0 || H <= 0 || W <= 0 || H > 32768 || W > 32768)
        return fail(EHR_ERR_INVALID, "ehr_fused_plan: bad sizes (1 <= L <= %d)", MAX_LINKS);
    if (!(slack > 0.f)) slack = 0.f;  // default (also NaN): a slot for every (view, link, tile)
    if (ctx->gexec) {  // a captured chain holds the old plan's pointers and shape
        EHR_HIP(hipGraphExecDestroy(ctx->gexec));
        ctx->gexec = nullptr;
    }
    int rc0 = vbuf_plan(ctx, B, L, V, T, H, W, slack, verts, tris, tri_link, opp);
    if (rc0) return rc0;
    int dev = 0;
    EHR_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    EHR_HIP(hipGetDeviceProperties(&prop, dev));
    ctx->num_cus = prop.multiProcessorCount;
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
// the last stage).
static int fused_chain(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* tri_link,
                       const int32_t* vert_link, const int32_t* opp, float* mvp, const float* ref, int B, int L, int V,
                       int T, int H, int W, float* mask, float* loss, float* grad_mvp, const StepHead* head,
                       const StepTail* tail, void* stream_) {
    if (!ctx) return fail(EHR_ERR_INVALID, "fused op: ctx is NULL");
    if (!verts || !tris || !tri_link || !vert_link || !opp || !mvp || !ref || !loss)
        return fail(EHR_ERR_INVALID, "fused op: NULL tensor");
    if (ctx->pB != B || ctx->pL != L || ctx->pV != V || ctx->pT != T || ctx->pH != H || ctx->pW != W)
        return fail(EHR_ERR_INVALID, "fused op: shape differs from the planned one; call ehr_fused_plan first");
    return vbuf_chain(ctx, verts, tris, tri_link, vert_link, opp, mvp, ref, B, L, V, T, H, W, mask, loss, grad_mvp, head,
                      tail, (hipStream_t)stream_);
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
                    float* adam_m, float* adam_v, int32_t* step, float* history, int history_rows, int32_t* history_row,
                    float lr, float beta1,
                    float beta2, float eps, float weight_decay, float* mvp, float* tc_jac, float* mask, float* loss_b,
                    float* grad_mvp, float* red, float* loss_out, float* grad_out, int defer_adam, void* stream) {
    if (!K || !link_poses || !dof || !adam_m || !adam_v || !step || !tc_jac || !grad_mvp || !red)
        return fail(EHR_ERR_INVALID, "ehr_solver_step: NULL tensor");
    if (L > MAX_LINKS) return fail(EHR_ERR_INVALID, "ehr_solver_step: more than %d links", MAX_LINKS);
    if (history && !history_row) return fail(EHR_ERR_INVALID, "ehr_solver_step: history given without history_row");
    StepHead head;
    head.dof = dof;
    head.K = K;
    head.link_poses = link_poses;
    head.tc_jac = tc_jac;
    head.hist_row = history_row;
    head.history = history;
    head.history_rows = history_rows;
    head.n = near_;
    head.f = far_;
    head.adam_step = step;
    head.hstate = (int*)ctx->vb_hstate.ptr;
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

int ehr_fused_bind_ref(ehr_ctx* ctx, const float* ref, void* stream) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_fused_bind_ref: ctx is NULL");
    if (ctx->capturing) return fail(EHR_ERR_INVALID, "ehr_fused_bind_ref: not inside a graph capture");
    // A captured chain has the bound-reference form (and the cached sums' pointers) baked into its kernel arguments: binding,
    // re-binding or unbinding under it would make its replays use stale sums.  The graph goes; ehr_graph_launch then fails
    // loudly ("no instantiated graph") until the chain is captured again.
    if (ctx->gexec) {
        EHR_HIP(hipGraphExecDestroy(ctx->gexec));
        ctx->gexec = nullptr;
    }
    if (!ref) {  // unbind
        ctx->vb_ref = nullptr;
        return EHR_OK;
    }
    if (ctx->pB == 0) return fail(EHR_ERR_INVALID, "ehr_fused_bind_ref: call ehr_fused_plan first");
    return vbuf_bind_ref(ctx, ref, (hipStream_t)stream);
}

int ehr_fused_status(ehr_ctx* ctx) {
    if (!ctx) return fail(EHR_ERR_INVALID, "ehr_fused_status: ctx is NULL");
    if (ctx->pB == 0) return EHR_OK;
    if (!ctx->vb_acc.ptr) return EHR_OK;
    EHR_HIP(hipDeviceSynchronize());
    int m4[4] = {0, 0, 0, 0};
    int rc0 = vbuf_meta_read(ctx, m4);
    if (rc0) return rc0;
    if (m4[EHR_META_OVERFLOW] & 4) ctx->vb_slow_needed = true;  // (VB_FLAG_NEED_SLOW; a captured chain must be re-captured)
    if (m4[EHR_META_OVERFLOW] & ~4)
        return fail(EHR_ERR_OVERFLOW, "fused path: an accumulator or the blended-pair spill pool overflowed");
    if (m4[EHR_META_OVERFLOW] & 4)
        return fail(EHR_ERR_RETRY, "fused path: the step met triangles for the general-triangle pass (near-plane clipping or "
                    "more than 512 pixels wide), which the solver step had not been launching; it is switched on now: run the "
                    "step again (its NaN left the optimiser state untouched; re-capture a captured chain)");
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
    ctx->gexec_reallocs = ctx->scratch_moves();
    return EHR_OK;
}

int ehr_graph_launch(ehr_ctx* ctx, void* stream) {
    if (!ctx || !ctx->gexec) return fail(EHR_ERR_INVALID, "ehr_graph_launch: no instantiated graph");
    if (ctx->gexec_reallocs != ctx->scratch_moves())  // a drop-in op or a re-plan moved scratch the graph points into
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
