/*
 * ehr.h -- C ABI of libehr_hip.so, the MI355X (gfx950) silhouette-rasterizer behind EasyHeC's mask-render hot path.
 *
 * Drop-in boundary: the reference reaches its renderer only through four nvdiffrast.torch entry points
 *     dr.RasterizeCudaContext()                      /root/reference/easyhec/structures/nvdiffrast_renderer.py:23
 *     dr.rasterize(glctx, pos, tri, resolution)      .../nvdiffrast_renderer.py:39  (and :64)
 *     dr.interpolate(attr, rast, tri)                .../nvdiffrast_renderer.py:42  (and :67)
 *     dr.antialias(color, rast, pos, tri)            .../nvdiffrast_renderer.py:43  (and :68)
 * nvdiffrast's own torch plugin binds those to C++ functions taking torch tensors; this header is what a
 * ctypes / pybind stub binds instead (easyhec_amd/_lib.py).  Plain pointers and sizes only.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the ctx's / current HIP device unless the name ends in _host;
 *   - all floats are fp32, all indices int32, tensors are dense row-major ("contiguous");
 *   - every call enqueues on `stream` (a hipStream_t; NULL = the null stream) and never blocks, EXCEPT where a
 *     function's comment says "synchronises";
 *   - return value: 0 = ok, negative = error; ehr_last_error() returns the message for the calling thread;
 *   - image rows follow nvdiffrast (row 0 = bottom) for the three drop-in ops; the fused op takes and returns
 *     images in the reference's final convention (row 0 = top, i.e. after nvdiffrast_renderer.py:47's flip).
 */
#ifndef EHR_H
#define EHR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EHR_OK 0
#define EHR_ERR_INVALID (-1)   /* bad argument */
#define EHR_ERR_HIP (-2)       /* a HIP runtime call failed */
#define EHR_ERR_OVERFLOW (-3)  /* internal work buffer overflow (reported, never silent) */
#define EHR_ERR_RETRY (-4)     /* the step was reported as NaN for a reason the library has already fixed: run it again */

typedef struct ehr_ctx ehr_ctx;

/* library / device --------------------------------------------------------------------------------------------- */
int ehr_version(void);                 /* ABI version, currently 8 (8: ehr_antialias_fwd_zg; 7: ehr_comm_p2p_*; 6: tile flags -- ehr_tile_flags_bytes, a tile_flags argument on ehr_rasterize_fwd / _grad, ehr_interpolate_fwd / _grad, ehr_antialias_fwd; 2: ehr_fused_plan takes the scene arrays; 3: ehr_fused_bind_ref, ehr_comm_*, history_row; 4: ehr_ctx_scratch_bytes; 5: EHR_ERR_RETRY from ehr_fused_status, ehr_antialias_fwd needs no zeroed work buffer and out != color, ehr_interpolate_da_*, ehr_rasterize_grad_db) */
const char* ehr_last_error(void);      /* message of the last failing call on this thread ("" if none) */
int ehr_device_count(void);            /* number of visible HIP devices (0 if none) */
const char* ehr_device_arch(int dev);  /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */

/* replaces dr.RasterizeCudaContext() -- nvdiffrast_renderer.py:23.  Owns binning scratch, grown on demand. */
int ehr_ctx_create(int device, ehr_ctx** out);
int ehr_ctx_destroy(ehr_ctx* ctx);
/* Device memory the context holds at the moment (its scratch buffers; they grow on demand and are released with it). */
size_t ehr_ctx_scratch_bytes(ehr_ctx* ctx);

/* replaces dr.rasterize -- nvdiffrast_renderer.py:39.
 * instance mode (ranges_host == NULL): pos [B,V,4], every image draws all T triangles.
 * range mode: pos [V,4], image b draws triangles ranges_host[2b] .. +ranges_host[2b+1] (HOST int32 [B,2]).
 * rast [B,H,W,4] = (u, v, z/w, triangle_id+1); rast_db [B,H,W,4] = (du/dx, du/dy, dv/dx, dv/dy) or NULL.
 * Two forms, same bits.  Small launches (B x T <= 131072: one link's mesh in one image, the reference's call) are
 * depth-tested straight into a key image in global memory and shaded by a second kernel: no queues, never a host wait.
 * Larger ones go through per-tile queues (count, allocate, fill, one workgroup per tile); that form synchronises only
 * while it is sizing its queue storage (first calls of a shape): in steady state the size of a frame reaches the host
 * asynchronously and is looked at by the next call, and a frame that outgrows the storage is still rendered exactly (the
 * tiles whose queues did not fit find their triangles themselves; slower, never incomplete) before the storage grows.
 * Neither form enqueues a fill: the context's counters / key image are left clean by the call's last kernel. */
/* tile_flags (optional, NULL = not wanted): ehr_tile_flags_bytes(B, H, W) bytes, one per (image, 32 x 8 pixel tile, row-major,
 * ceil(W / 32) per row), set to non-zero iff some pixel of the tile holds a triangle (the queued form sets them all).  The
 * ops below that take `tile_flags` leave `rast` unread where the flag is zero: a robot link covers a few per cent of a
 * frame, and the reference's schedule is five full-image passes per (view, link).  Passing NULL there is always valid. */
size_t ehr_tile_flags_bytes(int B, int H, int W);
int ehr_rasterize_fwd(ehr_ctx* ctx, const float* pos, const int32_t* tri, const int32_t* ranges_host, int B, int V,
                      int T, int H, int W, float* rast, float* rast_db, unsigned char* tile_flags, void* stream);

/* backward of dr.rasterize w.r.t. pos through (u,v); dy = grad of rast [B,H,W,4] (z/w and id carry no gradient).
 * grad_pos (pos's shape) is ACCUMULATED into: the caller zero-fills it. */
int ehr_rasterize_grad(const float* pos, const int32_t* tri, const float* rast, const float* dy, int range_mode, int B,
                       int V, int T, int H, int W, float* grad_pos, const unsigned char* tile_flags, void* stream);
/* ... and w.r.t. pos through rast_db: ddb = grad of rast_db [B,H,W,4]; grad_pos is ACCUMULATED into.  (EasyHeC discards
 * rast_db, nvdiffrast_renderer.py:39; this completes the op.) */
int ehr_rasterize_grad_db(const float* pos, const int32_t* tri, const float* rast, const float* ddb, int range_mode, int B,
                          int V, int T, int H, int W, float* grad_pos, void* stream);

/* replaces dr.interpolate -- nvdiffrast_renderer.py:42.  attr [Ba,V,A] with Ba == B or 1; out [B,H,W,A]. */
int ehr_interpolate_fwd(const float* attr, const float* rast, const int32_t* tri, int B, int Ba, int V, int T, int A,
                        int H, int W, float* out, const unsigned char* tile_flags, void* stream);
/* grad_attr [Ba,V,A] is ACCUMULATED into (caller zero-fills) and may be NULL when the attributes need no gradient
 * (EasyHeC interpolates constant colours); grad_rast [B,H,W,4] is overwritten. */
int ehr_interpolate_grad(const float* attr, const float* rast, const int32_t* tri, const float* dy, int B, int Ba, int V,
                         int T, int A, int H, int W, float* grad_attr, float* grad_rast, const unsigned char* tile_flags,
                         void* stream);

/* dr.interpolate's second output, the attribute pixel differentials (interpolate(attr, rast, tri, rast_db, diff_attrs);
 * EasyHeC does not ask for them -- nvdiffrast_renderer.py:42 passes neither -- they complete the op's signature).
 * rast_db [B,H,W,4] from ehr_rasterize_fwd; diff_idx [D] int32 DEVICE attribute indices, NULL = all (then D == A);
 * out_da [B,H,W,2D] = (d attr_j / dX, d attr_j / dY) for j = diff_idx[i] at channels 2i, 2i+1; 0 where no triangle.
 * grad: grad_attr [Ba,V,A] is ACCUMULATED into (may be NULL), grad_rast_db [B,H,W,4] overwritten (may be NULL); rast
 * itself receives no gradient from the differentials. */
int ehr_interpolate_da_fwd(const float* attr, const float* rast, const float* rast_db, const int32_t* tri,
                           const int32_t* diff_idx, int B, int Ba, int V, int T, int A, int D, int H, int W, float* out_da,
                           void* stream);
int ehr_interpolate_da_grad(const float* attr, const float* rast, const float* rast_db, const int32_t* tri,
                            const int32_t* diff_idx, const float* dy_da, int B, int Ba, int V, int T, int A, int D, int H,
                            int W, float* grad_attr, float* grad_rast_db, void* stream);

/* replaces dr.antialias_construct_topology_hash(tri).  Writes opp [T,3] int32: for triangle t and edge k
 * (k=0: v1-v2, k=1: v2-v0, k=2: v0-v1) the third vertex of the other triangle sharing that edge, or -1.
 * scratch: ehr_topology_scratch_bytes(T) bytes of device memory. */
size_t ehr_topology_scratch_bytes(int T);
int ehr_antialias_topology(const int32_t* tri, int T, int32_t* opp, void* scratch, size_t scratch_bytes, void* stream);

/* replaces dr.antialias -- nvdiffrast_renderer.py:43.  color/out [B,H,W,C]; pos [B,V,4] or [V,4] (range_mode).
 * work: ehr_antialias_work_bytes(B,H,W) bytes, filled by fwd and consumed by grad (nvdiffrast's work buffer); it needs
 * no initialisation.  out must not alias color (every pixel reads its neighbours' colours).  The forward result is
 * deterministic: every pixel adds the blends that land on it in the order of a serial sweep over pixel pairs. */
size_t ehr_antialias_work_bytes(int B, int H, int W);
int ehr_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri, const int32_t* opp,
                      int range_mode, int B, int V, int T, int H, int W, int C, float* out, void* work,
                      const unsigned char* tile_flags, void* stream);
/* The same forward pass, which also clears grad_pos_zero (pos's shape; may be NULL) on its way: the buffer a later
 * ehr_antialias_grad accumulates into then needs no fill of its own -- one launch less per (view, link) image of the
 * reference's schedule (ABI 8).  A caller that runs the backward pass twice zero-fills the buffer itself the second time. */
int ehr_antialias_fwd_zg(const float* color, const float* rast, const float* pos, const int32_t* tri, const int32_t* opp,
                         int range_mode, int B, int V, int T, int H, int W, int C, float* out, void* work,
                         const unsigned char* tile_flags, float* grad_pos_zero, void* stream);
/* grad_color [B,H,W,C] is overwritten (may be NULL: not wanted); grad_pos (pos's shape) is ACCUMULATED into (caller zero-fills). */
int ehr_antialias_grad(const float* color, const float* rast, const float* pos, const int32_t* tri, const float* dy,
                       const void* work, int range_mode, int B, int V, int T, int H, int W, int C, float* grad_color,
                       float* grad_pos, void* stream);

/* Fused hot path: what RBSolver.forward + loss.backward() compute per optimisation step
 * (/root/reference/easyhec/modeling/models/rb_solve/rb_solver.py:60-72 through nvdiffrast_renderer.py:33-47),
 * for B views x L links in one pass:
 *     si[b,l]  = flip_y(antialias(interpolate(1, rasterize(MVP[b,l] * verts_l, tri_l))))
 *     mask[b]  = min(sum_l si[b,l], 1)                 -> mask [B,H,W] (may be NULL)
 *     loss[b]  = sum_pixels (mask[b] - ref[b])^2       -> loss [B]
 *     grad_mvp[b,l] = d loss[b] / d MVP[b,l]           -> grad_mvp [B,L,16] (may be NULL: forward only)
 * Scene = all links concatenated: verts [V,3]; tris [T,3] with GLOBAL vertex indices, sorted by link;
 * tri_link [T] / vert_link [V] int32 link of each triangle / vertex; opp [T,3] from ehr_antialias_topology on the
 * concatenated mesh (links share no vertices, so the per-link topology is preserved).
 * ehr_fused_plan prepares the context for one scene and shape (it synchronises and reads the scene back once): it
 * sizes the scratch for (B,L,V,T,H,W) and builds the static acceleration index of the scene's triangles (clusters of 64
 * spatially close triangles per link, padded index tables), so it takes the scene arrays -- the SAME device arrays that
 * are later passed to ehr_render_mask_loss / ehr_solver_step (they are checked by address; call it again when the
 * scene, the shape or the arrays change -- also when only the CONTENTS of verts / tris change: the index holds a copy
 * of every triangle's corners).  Limits, checked here: L <= 32, W <= 32736, H <= 32760, <= 262144 triangles per link.
 * Any number of views: a call's views go through the launch chain in chunks (one chunk up to 512 / L views; fewer per
 * chunk when the chunk's scratch -- clip-space vertices, raster records, 3.5 KB per (view, link, tile) job slot: 0.1 GB per
 * 1280x720 view of an 8-link robot -- would exceed 24 GB, EHR_VB_SCRATCH_MB), all inside the one call.  The hot calls never synchronise or allocate, so they can be
 * captured in a hipGraph; a new plan invalidates a captured graph.  `slack` <= 0 (default): one job slot per (view, link,
 * tile), nothing can overflow except the fixed-point accumulators (|sum| > 2^31) and a 16 MB spill pool for tiles with
 * more than 64 blended pairs per link; `slack` > 0 provides only `slack` job slots per view tile (fractions allowed:
 * 0.5 = half as many slots as the view has tiles; less scratch, larger chunks).  An overflow makes loss[] NaN (never a silently wrong image), leaves the optimiser state untouched, and
 * ehr_fused_status() returns EHR_ERR_OVERFLOW after synchronising. */
int ehr_fused_plan(ehr_ctx* ctx, int B, int L, int V, int T, int H, int W, float slack, const float* verts,
                   const int32_t* tris, const int32_t* tri_link, const int32_t* opp);
int ehr_render_mask_loss(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* tri_link,
                         const int32_t* vert_link, const int32_t* opp, const float* mvp, const float* ref, int B,
                         int L, int V, int T, int H, int W, float* mask, float* loss, float* grad_mvp, void* stream);
int ehr_fused_status(ehr_ctx* ctx); /* synchronises the device; 0, EHR_ERR_OVERFLOW or EHR_ERR_RETRY (see ehr_solver_step) */

/* Binds a reference-mask batch ref [B,H,W] (the planned shape) to the plan.  The reference masks of a solve do not
 * change (rb_solver.py:70 compares every step with the same dps['mask']), so the part of the frame loss that comes from
 * image tiles no link touches -- sum(ref^2) over those tiles, 90 % of a 1280x720 frame -- is a constant of the solve.
 * This call stores it once (one pass over ref: per tile the fixed-point value the composite stage would add, per view the
 * total); afterwards ehr_render_mask_loss / ehr_solver_step calls that pass THIS pointer visit only the tiles inside the
 * views' link boxes and correct the cached total by integer differences (with a mask output they also store zeros to
 * every other tile of `mask`, without reading ref there).  The sums are 64-bit fixed point, so loss, gradients and masks
 * are bit-identical to the unbound path; it is an exact algebraic saving, not skipped work.  The caller promises not to
 * modify ref's contents while it is bound: call again after changing them, or with ref == NULL to unbind.
 * ehr_fused_plan unbinds.  Calls with another pointer take the unbound path.  Enqueues on `stream`; not to be called
 * inside a graph capture. */
int ehr_fused_bind_ref(ehr_ctx* ctx, const float* ref, void* stream);

/* Measurement hook (bench.py's roofline leg): when enabled, every ehr_render_mask_loss / ehr_solver_step call records
 * hipEvents around its kernels on the launch stream.  ehr_fused_timing_read synchronises, writes the ACCUMULATED
 * milliseconds per stage since the last read and the number of calls covered, then resets.  Stages of the default
 * (visibility-buffer) chain: ms[0] vertex kernel (pose forward, clip-space vertices, per-triangle raster records, cluster
 * and link boxes), ms[1] job kernel (one wave per (view, link, tile): box culling, LDS rasterizer, depth tests where the
 * silhouette analysis will look, and -- since round 5 -- the resolve stage of the job it has just drawn: silhouette analysis,
 * antialiased values, blended pairs; the dominant kernel), ms[4] composite kernel (link sum, clamp, loss, mask, backward; its
 * last-arriving workgroup runs the finish stage: accumulators -> loss / grad_mvp [-> pose backward -> Adam]).  ms[2] brackets
 * the general-triangle pass and the resolve kernel of the jobs it redrew, which the solver step launches only once a step
 * has needed them: in the default chain it is an EMPTY pair of events and measures what a pair costs on this stack (every
 * other figure includes about as much); ms[3], ms[5], ms[6] are unused (~0).
 * Not for use under graph capture. */
#define EHR_FUSED_STAGES 7
int ehr_fused_timing(ehr_ctx* ctx, int enable);
int ehr_fused_timing_read(ehr_ctx* ctx, float* ms, int* ncalls);

/* The two ends of one optimisation step around the renderer (trainer/rbsolver.py:29-43), each a single tiny kernel so
 * that a whole step is a handful of launches with no host round trip:
 *   ehr_pose_forward : dof[6] -> Tc_c2b = se3_exp_map(dof) (utils/pytorch3d_se3.py:46-130) ->
 *                      mvp[b,l] = proj(K) @ opencv2blender @ Tc_c2b @ link_poses[b,l]   (rb_solver.py:52,63;
 *                      nvdiffrast_renderer.py:33-37).  tc_jac [7,16]: Tc_c2b and d Tc_c2b / d dof_i (forward mode).
 *                      If history != NULL, row step[0] of history [history_rows,6] receives dof (rb_solver.py:50-51).
 *   ehr_pose_backward: grad_mvp [B,L,16] (= d loss_b / d mvp[b,l]) and loss [B] -> red[8] =
 *                      {d(sum_b loss_b)/d dof (6), sum_b loss_b, B}: the 8 floats one all-reduce(sum) exchanges.
 *   ehr_pose_adam    : torch.optim.Adam step (L2 weight decay added to the gradient) on dof with the MEAN-loss
 *                      gradient red[0..5] / red[7]; m, v [6] and step [1] are the optimiser state; loss_out[0] =
 *                      red[6] / red[7]; grad_out [6] optional.  K is the 3x3 row-major intrinsics matrix. */
int ehr_pose_forward(const float* dof, const float* K, const float* link_poses, int B, int L, int H, int W, float n,
                     float f, float* mvp, float* tc_jac, const int32_t* step, float* history, int history_rows,
                     void* stream);
int ehr_pose_backward(const float* grad_mvp, const float* loss, const float* K, const float* link_poses,
                      const float* tc_jac, int B, int L, int H, int W, float n, float f, float* red, void* stream);
int ehr_pose_adam(float* dof, float* m, float* v, int32_t* step, const float* red, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float* loss_out, float* grad_out, void* stream);

/* One whole optimisation step (trainer/rbsolver.py:29-43) as a chain of 3 launches (vertex + raster records; jobs, which
 * resolve themselves; composite + finish.  Two more -- the general-triangle pass for jobs with a triangle that crosses the
 * near plane or is wider than 512 pixels, and the resolve kernel of what it redrew -- join the chain once a step has needed them: that step is reported as NaN like an overflow -- loss,
 * gradient NaN, optimiser state untouched --, ehr_fused_status() then returns EHR_ERR_RETRY and switches the pass on for
 * the context's later calls; run the step again, and re-capture the chain if it was captured in a graph.  A robot in
 * front of the camera never has such a triangle; the stateless ehr_render_mask_loss always launches the pass):
 * ehr_pose_forward is merged into the vertex kernel (which also does the per-step housekeeping) and ehr_pose_backward +
 * ehr_pose_adam into the finish stage of the composite kernel.  Same arithmetic and outputs as calling the pieces one by one:
 * mvp [B,L,16], tc_jac [7,16], loss_b [B], grad_mvp [B,L,16], red [8], loss_out [1], grad_out [6] are all written.
 * `step` [1] is Adam's step count (bias correction); `history_row` [1] is the row of history [history_rows,6] that
 * receives this step's pose (rb_solver.py:50-51: the first free row) and is advanced by the call -- two counters,
 * because a solver built on a loaded model starts a fresh optimiser but keeps appending to the history.  history ==
 * NULL records nothing.  A REPORTED step (overflow: NaN loss, pose and optimiser state untouched) has recorded the unchanged
 * pose; the next call of this context notices that neither `step` nor `history_row` has moved since and writes into the
 * SAME row again, so the history holds one row per effective step -- on every rank of a data-parallel job alike (the NaN
 * travels with the exchanged sums), without the caller rewinding anything (ABI 7).
 * defer_adam != 0 stops after `red` so that the caller can exchange it across ranks and then apply Adam: ehr_comm_p2p_step
 * (exchange + Adam in one launch), or ehr_comm_allreduce / any all-reduce followed by ehr_pose_adam.
 * Requires ehr_fused_plan for (B,L,V,T,H,W); never synchronises or allocates. */
int ehr_solver_step(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* tri_link,
                    const int32_t* vert_link, const int32_t* opp, const float* K, const float* link_poses,
                    const float* ref, int B, int L, int V, int T, int H, int W, float near_plane, float far_plane,
                    float* dof, float* adam_m, float* adam_v, int32_t* step, float* history, int history_rows,
                    int32_t* history_row, float lr, float beta1, float beta2, float eps, float weight_decay, float* mvp,
                    float* tc_jac, float* mask, float* loss_b, float* grad_mvp, float* red, float* loss_out,
                    float* grad_out, int defer_adam, void* stream);

/* The data-parallel exchange (replaces the DDP gradient all-reduce of trainer/base.py:349-352 under the
 * one-process-per-GPU launch of tools/run_easyhec.py:41-50).  Views shard over the ranks; per step every rank runs
 * ehr_solver_step(defer_adam = 1) on its views, the ranks sum the 8-float vector red = [d sum(loss)/d dof (6), sum(loss),
 * n_views] and then run ehr_pose_adam, bit-identically.  The library owns an RCCL communicator per context and issues
 * ncclAllReduce on the stream it is given -- the chain's own stream -- so the step is [solver step, all-reduce, Adam] on ONE
 * stream with no host round trip, and the three calls can be captured together in a hipGraph (ehr_graph_*).
 *   ehr_comm_unique_id : rank 0 obtains the 128-byte ncclUniqueId; the caller ships it to every rank (any transport);
 *   ehr_comm_init      : every rank, with the same id: ncclCommInitRank on the context's device (blocks until all ranks
 *                        have joined); nranks == 1 is legal (a single-GPU communicator);
 *   ehr_comm_allreduce : in-place sum of red[0..count) (device, fp32) over the ranks, enqueued on `stream`;
 *   ehr_comm_destroy   : releases the communicator (ehr_ctx_destroy does it too).
 * RCCL is resolved with dlopen at the first call (the librccl a host process already holds is reused). */
int ehr_comm_unique_id(void* id128_host);
int ehr_comm_init(ehr_ctx* ctx, const void* id128_host, int nranks, int rank);
int ehr_comm_allreduce(ehr_ctx* ctx, float* red, int count, void* stream);
int ehr_comm_destroy(ehr_ctx* ctx);

/* The same exchange WITHOUT a collective library (ABI 7; SURVEY 5 / 8e's one-shot exchange over xGMI peer memory): a step of
 * ~70 us has no room for a 10-30 us ncclAllReduce followed by a launch of its own for Adam.  Every rank owns a mailbox in its
 * device memory; the exchange is one single-workgroup kernel per rank and step that stores the rank's 8 floats and a sequence
 * number into its slot of every peer's mailbox (system-scope stores), waits for its own mailbox to fill, adds the slots up in
 * rank order (bit-identical sums on every rank) and goes on to the Adam update of ehr_pose_adam in the same launch.
 *   ehr_comm_p2p_export : allocates (once) and clears this context's mailbox and returns its 64-byte hipIpcMemHandle; the
 *                         caller ships every rank's handle to every rank (any transport, e.g. an all_gather);
 *   ehr_comm_p2p_open   : handles [nranks][64] in rank order (the own entry is ignored); opens the peers' mailboxes
 *                         (hipIpcOpenMemHandle: ranks on different GPUs of one node, or -- for tests -- on the same GPU);
 *                         nranks <= EHR_P2P_MAX_RANKS.  Every rank must have exported before any rank opens, and opened
 *                         before any rank exchanges (the caller's barrier).
 *   ehr_comm_p2p_step   : red[8] (device) <- sum over the ranks, in place; with dof != NULL the Adam update of ehr_pose_adam
 *                         follows in the same kernel (same arguments).  Every rank must make the same sequence of calls.
 *                         Enqueued on `stream`, capturable.  A peer that never answers is REPORTED after ~half a minute: the sums are NaN
 *                         and the optimiser state stays as it was.
 *   ehr_comm_p2p_close  : closes the peers' mailboxes and frees the own one (ehr_ctx_destroy does it too). */
#define EHR_P2P_MAX_RANKS 64
int ehr_comm_p2p_export(ehr_ctx* ctx, void* handle64_host);
int ehr_comm_p2p_open(ehr_ctx* ctx, const void* handles_host, int nranks, int rank);
int ehr_comm_p2p_step(ehr_ctx* ctx, float* red, float* dof, float* m, float* v, int32_t* step, float lr, float beta1,
                      float beta2, float eps, float weight_decay, float* loss_out, float* grad_out, void* stream);
int ehr_comm_p2p_close(ehr_ctx* ctx);

/* hipGraph capture of launch chains.  ehr_graph_begin opens a capture on a stream the context owns and returns it; every
 * library call made with THAT stream until ehr_graph_end (e.g. one ehr_solver_step, or ehr_solver_step with defer_adam
 * followed by ehr_pose_adam) is recorded instead of executed, including the chain's side-stream fork/join.  All device
 * pointers and scalars of the recorded calls are baked in: the buffers must stay alive and in place; iteration state
 * (dof, Adam moments, step counter, history row) lives on the device, so replays advance the optimisation exactly like
 * eager calls.  ehr_graph_launch enqueues one replay on `stream`.  One graph per context; ehr_graph_begin discards the
 * previous one.  The plan (ehr_fused_plan) must not change between capture and replay. */
int ehr_graph_begin(ehr_ctx* ctx, void** capture_stream);
int ehr_graph_end(ehr_ctx* ctx);
int ehr_graph_launch(ehr_ctx* ctx, void* stream);
int ehr_graph_release(ehr_ctx* ctx);

/* Next-best-view scoring of the space explorer (modeling/models/rb_solve/space_explorer.py:152-165): for each of Q
 * candidate joint configurations, the robot (all links merged into one mesh, utils/render_api.py:70-96) is rendered
 * WITHOUT antialiasing under S camera poses (mask = rast[..., 2] > 0, structures/nvdiffrast_renderer.py:70) and the
 * per-pixel unbiased variance over the S masks is summed over the image (torch.var(masks, dim=0).sum()).  For binary
 * masks that is  sum_px c (S - c) / (S (S - 1))  with c = number of poses covering the pixel, so the kernel returns the
 * exact integer  score[q] = sum_px c (S - c)  and the caller divides.
 *   verts [V,3], tris [T,3] (global vertex ids), vert_link [V] (link of each vertex; NULL = all 0),
 *   mvp [Q,S,L,16] = proj(K) @ opencv2blender @ Tc_c2b[s] @ link_pose[q,l], row-major;
 *   score [Q] int64 (device); count [Q,H,W] uint8 (device, optional; image convention row 0 = top): c per pixel.
 * Two implementations, same integers.  Where the mesh's triangles are grouped by link (vert_link given, every triangle's
 * vertices in one link, links ascending), L <= 32 and S x L <= 512, the call runs on the solver's machinery: the
 * static cluster index of the mesh (built at the first call, rebuilt when the arrays' contents change: a content hash is
 * taken every call), then per chunk of candidates the vertex kernel, the job kernel in a coverage-only form (a triangle
 * all of whose coverable pixels have a depth in (0, 1] needs no depth test at all; the few others are tested per pixel)
 * and a count kernel; one synchronisation at the start (hash) and one at the end.  If a drawn pixel turns out to have a
 * depth <= 0, or a triangle crosses the near plane or spans > 512 pixels -- geometry within two near-plane distances of
 * the camera --, the call is redone by the other implementation: per-triangle tile queues and an exact z-buffer, in
 * passes of about chunk_views rendered views (<= 0: default 512), one synchronisation per pass.
 * EHR_SCORE_PATH=tile / =chain (environment) forces one of them (chain: an error where it cannot decide). */
int ehr_mask_variance(ehr_ctx* ctx, const float* verts, const int32_t* tris, const int32_t* vert_link,
                      const float* mvp, int Q, int S, int L, int V, int T, int H, int W, int64_t* score,
                      uint8_t* count, int chunk_views, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EHR_H */
