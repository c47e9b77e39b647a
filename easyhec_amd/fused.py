"""Fused mask-loss op: all (view, link) silhouettes, the link composite, the SSE loss and its gradient down to each
(view, link) 4x4 matrix in one pass (``ehr_render_mask_loss`` in include/ehr.h).

It computes exactly what RBSolver.forward's double loop computes through the three drop-in ops
(/root/reference/easyhec/modeling/models/rb_solve/rb_solver.py:60-72 over
/root/reference/easyhec/structures/nvdiffrast_renderer.py:33-47), but touches HBM once per pixel instead of
~2 kB per pixel.  HIP only; no fallback."""
import ctypes

import torch

from . import _lib, dr

__all__ = ["LinkScene", "render_mask_loss", "mvp_matrices"]


class LinkScene:
    """All link meshes concatenated on the device: verts [V,3], tris [T,3] (global vertex ids, sorted by link),
    tri_link [T], edge topology [T,3].  Built once per robot (rb_solver.py:23-28 keeps one buffer per link)."""

    def __init__(self, vertices, faces, device):
        assert len(vertices) == len(faces) and len(vertices) > 0
        self.device = torch.device(device)
        self.num_links = len(vertices)
        vs, fs, ls, vls = [], [], [], []
        self.vert_offsets, self.tri_offsets = [0], [0]
        for l, (v, f) in enumerate(zip(vertices, faces)):
            v = torch.as_tensor(v, dtype=torch.float32).reshape(-1, 3)
            f = torch.as_tensor(f, dtype=torch.int32).reshape(-1, 3)
            vs.append(v)
            fs.append(f + self.vert_offsets[-1])
            ls.append(torch.full((f.shape[0],), l, dtype=torch.int32))
            vls.append(torch.full((v.shape[0],), l, dtype=torch.int32))
            self.vert_offsets.append(self.vert_offsets[-1] + v.shape[0])
            self.tri_offsets.append(self.tri_offsets[-1] + f.shape[0])
        self.verts = torch.cat(vs).contiguous().to(self.device)
        self.tris = torch.cat(fs).contiguous().to(self.device)
        self.tri_link = torch.cat(ls).contiguous().to(self.device)
        self.vert_link = torch.cat(vls).contiguous().to(self.device)
        self.num_verts, self.num_tris = self.verts.shape[0], self.tris.shape[0]
        # links share no vertices, so the topology of the concatenation is the per-link topology
        self.opp = dr.antialias_construct_topology_hash(self.tris).opp
        self.geometry_bytes = 12 * self.num_verts + 12 * self.num_tris


class _Plan:
    """Per-context plan for one (B, L, T, H, W) shape: sizes the ctx scratch once so the hot call never allocates."""

    def __init__(self, glctx, scene, B, H, W, slack=0.0):
        self.key = _plan_key(scene, B, H, W)
        self.slack = float(slack)
        with torch.cuda.device(glctx.device):
            _lib.check(_lib.lib().ehr_fused_plan(glctx.handle, B, scene.num_links, scene.num_verts, scene.num_tris, H, W,
                                                 ctypes.c_float(slack), _lib.ptr(scene.verts), _lib.ptr(scene.tris),
                                                 _lib.ptr(scene.tri_link), _lib.ptr(scene.opp)), "ehr_fused_plan")


def _plan_key(scene, B, H, W):
    # the plan also holds a static index of the scene's triangles, so it is tied to the scene's buffers
    return (B, H, W, scene.num_links, scene.num_verts, scene.num_tris, scene.verts.data_ptr(), scene.tris.data_ptr())


def _ensure_plan(glctx, scene, B, H, W, slack=None):
    """slack: job slots per view tile (``ehr_fused_plan``); 0 = one per (view, link, tile), which can never overflow.  None
    (the autograd path): a plan of this shape that exists is kept whatever its budget, a new one gets every slot (a NaN
    gradient would poison torch's Adam); the launch chain of :class:`easyhec_amd.fast.FusedPoseStep` asks for 1 (an eighth
    of the scratch at 8 links) and recovers from the reported overflow by planning again with 0."""
    plan = getattr(glctx, "_plan", None)
    same = plan is not None and plan.key == _plan_key(scene, B, H, W)
    if not same or (slack is not None and getattr(plan, "slack", 0.0) != float(slack)):
        glctx._plan = _Plan(glctx, scene, B, H, W, slack=0.0 if slack is None else slack)
        glctx._bound_ref = None  # ehr_fused_plan forgets a bound reference mask


def check_status(glctx):
    """Synchronises; raises if a bin queue / blend list overflowed since the plan (loss is NaN in that case)."""
    with torch.cuda.device(glctx.device):
        _lib.check(_lib.lib().ehr_fused_status(glctx.handle), "fused render")


def _launch(glctx, scene, mvp, ref, mask, loss, grad_mvp):
    B, L = mvp.shape[0], mvp.shape[1]
    H, W = ref.shape[1], ref.shape[2]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    with torch.cuda.device(glctx.device):
        _lib.check(_lib.lib().ehr_render_mask_loss(
            glctx.handle, _lib.ptr(scene.verts), _lib.ptr(scene.tris), _lib.ptr(scene.tri_link),
            _lib.ptr(scene.vert_link), _lib.ptr(scene.opp),
            _lib.ptr(mvp), _lib.ptr(ref), B, L, scene.num_verts, scene.num_tris, H, W, _lib.ptr(mask), _lib.ptr(loss),
            _lib.ptr(grad_mvp), stream), "ehr_render_mask_loss")


class _RenderMaskLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, glctx, scene, mvp, ref, want_mask):
        B, L = mvp.shape[0], mvp.shape[1]
        H, W = ref.shape[1], ref.shape[2]
        _ensure_plan(glctx, scene, B, H, W)
        mask = torch.empty((B, H, W), dtype=torch.float32, device=mvp.device) if want_mask else None
        loss = torch.empty((B,), dtype=torch.float32, device=mvp.device)
        need_grad = mvp.requires_grad
        grad = torch.empty((B, L, 4, 4), dtype=torch.float32, device=mvp.device) if need_grad else None
        _launch(glctx, scene, mvp, ref, mask, loss, grad)
        if need_grad:
            ctx.save_for_backward(grad)
        if mask is None:
            mask = torch.empty((0,), dtype=torch.float32, device=mvp.device)
        ctx.mark_non_differentiable(mask)
        return mask, loss

    @staticmethod
    def backward(ctx, _gmask, gloss):
        (grad,) = ctx.saved_tensors
        return None, None, grad * gloss[:, None, None, None], None, None


def render_mask_loss(glctx, scene, mvp, ref, want_mask=True):
    """mvp [B,L,4,4] (= proj @ opencv2blender @ Tc_c2b @ link_pose), ref [B,H,W] float masks (row 0 = top).
    Returns ``(mask [B,H,W], loss [B])`` with ``loss[b] = sum((mask[b] - ref[b])**2)``; differentiable w.r.t. mvp
    (the rendered mask is returned detached, as the reference only back-propagates the loss)."""
    dr._check_dev("mvp", mvp, torch.float32)
    dr._check_dev("ref", ref, torch.float32)
    dr._require(mvp.dim() == 4 and mvp.shape[1] == scene.num_links and mvp.shape[2:] == (4, 4),
                "mvp must have shape [B, num_links, 4, 4]")
    dr._require(ref.dim() == 3 and ref.shape[0] == mvp.shape[0], "ref must have shape [B, H, W]")
    dr._require(mvp.device == scene.device and ref.device == scene.device, "mvp/ref must be on the scene's device")
    return _RenderMaskLoss.apply(glctx, scene, mvp.contiguous(), ref.contiguous(), want_mask)


def mvp_matrices(K, H, W, Tc_c2b, link_poses, n=0.001, f=10.0):
    """[B,L,4,4] clip matrices ``proj @ opencv2blender @ Tc_c2b @ link_poses[b,l]`` -- the product
    nvdiffrast_renderer.py:33-37 forms per call with rb_solver.py:63's ``Tc_c2l``; differentiable w.r.t. Tc_c2b."""
    from .nvdiffrast_utils import K_to_projection, opencv2blender
    proj = K_to_projection(K, H, W, n=n, f=f).to(Tc_c2b.device)
    o2b = opencv2blender(device=Tc_c2b.device)
    Tc_c2l = Tc_c2b[None, None] @ link_poses          # rb_solver.py:63
    return proj @ (o2b @ Tc_c2l)                      # nvdiffrast_renderer.py:35,37 (same association)


# kernels bracketed by ehr_fused_timing's hipEvents, in ms[] order (include/ehr.h): vertex + records, jobs, resolve,
# composite (+ the finish stage in its last workgroup); slots 3, 5, 6 are unused.
STAGES = ("vertex", "job", "resolve", "unused3", "composite", "unused5", "unused6")
DOMINANT_STAGE, DOMINANT_KERNEL = "job", "vb_job_kernel"


def bind_ref(glctx, scene, ref):
    """Bind a reference-mask batch to the context's plan (``ehr_fused_bind_ref``): the loss contribution of the tiles no
    link touches is cached once, and later calls with THIS tensor and no mask output only visit the tiles inside the
    views' link boxes -- bit-identical results (64-bit fixed-point sums).  The caller must not modify ``ref`` in place
    while it is bound; ``ref=None`` unbinds.  The context keeps a reference to the tensor."""
    if ref is None:
        _lib.check(_lib.lib().ehr_fused_bind_ref(glctx.handle, None, None), "ehr_fused_bind_ref")
        glctx._bound_ref = None
        return
    dr._check_dev("ref", ref, torch.float32)
    dr._require(ref.dim() == 3 and ref.is_contiguous(), "ref must be a contiguous [B, H, W] tensor")
    _ensure_plan(glctx, scene, ref.shape[0], ref.shape[1], ref.shape[2])
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    with torch.cuda.device(glctx.device):
        _lib.check(_lib.lib().ehr_fused_bind_ref(glctx.handle, _lib.ptr(ref), stream), "ehr_fused_bind_ref")
    glctx._bound_ref = ref


def set_timing(glctx, enable):
    """Measurement hook (include/ehr.h ``ehr_fused_timing``): hipEvents around each kernel of the fused op."""
    _lib.check(_lib.lib().ehr_fused_timing(glctx.handle, int(bool(enable))), "ehr_fused_timing")


def read_timing(glctx):
    """-> ({stage: accumulated ms}, number of calls); synchronises and resets."""
    ms = (ctypes.c_float * len(STAGES))()
    n = ctypes.c_int(0)
    with torch.cuda.device(glctx.device):
        _lib.check(_lib.lib().ehr_fused_timing_read(glctx.handle, ms, ctypes.byref(n)), "ehr_fused_timing_read")
    return {s: float(ms[i]) for i, s in enumerate(STAGES)}, int(n.value)
