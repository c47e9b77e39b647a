"""Drop-in for the four ``nvdiffrast.torch`` entry points EasyHeC uses, backed by hand-written HIP kernels.

The reference does ``import nvdiffrast.torch as dr`` and calls
(/root/reference/easyhec/structures/nvdiffrast_renderer.py):

    dr.RasterizeCudaContext()                              :23
    dr.rasterize(glctx, pos, tri, resolution=[H, W])       :39, :64
    dr.interpolate(attr, rast, tri)                        :42, :67
    dr.antialias(color, rast, pos, tri)                    :43, :68

``import easyhec_amd.dr as dr`` gives the same names, positional order, return tuples and autograd behaviour.
Everything runs through the C ABI of include/ehr.h (libehr_hip.so); tensors must live on a HIP device -- there is no
CPU or PyTorch fallback, a missing library or a CPU tensor raises.
"""
import ctypes

import torch

from . import _lib

__all__ = ["RasterizeCudaContext", "RasterizeGLContext", "rasterize", "interpolate", "antialias",
           "antialias_construct_topology_hash", "TopologyHash", "carry_tile_flags"]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


# Tile flags (ABI 6): dr.rasterize leaves one byte per (image, 32 x 8 tile) -- "a triangle was drawn here" -- in a small
# tensor that rides on the `rast` tensor object it returns (a Python attribute: it follows the object through
# .contiguous() of a contiguous tensor, not through views or detach()).  dr.interpolate / dr.antialias and the backward
# passes look for it on the `rast` they are handed and skip the empty nine tenths of a link's image; without it (a `rast`
# that came from somewhere else) they process every pixel, as before.  The results are identical either way.
_FLAGS_ATTR = "_ehr_tile_flags"


_FLAGS_OF = "_ehr_tile_flags_of"   # (storage address, version counter) of the rasterizer output the flags describe


def _attach_flags(rast, flags):
    # (a `rast` edited in place afterwards -- ids composited into empty tiles, say -- or flags carried onto another tensor's
    #  storage are not trusted any more: interpolate / antialias then process every pixel)
    setattr(rast, _FLAGS_ATTR, flags)
    setattr(rast, _FLAGS_OF, (rast.data_ptr(), rast._version))


def _flags_of(rast):
    f, of = getattr(rast, _FLAGS_ATTR, None), getattr(rast, _FLAGS_OF, None)
    if f is None or of is None:
        return None
    B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
    ok = (f.is_cuda and f.device == rast.device and f.numel() == _lib.lib().ehr_tile_flags_bytes(B, H, W)
          and of == (rast.data_ptr(), rast._version))
    return f if ok else None


def carry_tile_flags(src, dst):
    """Hand ``src``'s tile flags (if any) on to ``dst`` -- a detached or otherwise re-wrapped tensor over the SAME rasterizer
    output (same storage, same version counter) -- and return ``dst``."""
    f, of = getattr(src, _FLAGS_ATTR, None), getattr(src, _FLAGS_OF, None)
    if f is not None and of is not None and dst.shape[:3] == src.shape[:3] and dst.data_ptr() == of[0]:
        setattr(dst, _FLAGS_ATTR, f)
        setattr(dst, _FLAGS_OF, of)
    return dst


def _check_dev(name, t, dtype):
    _require(isinstance(t, torch.Tensor), f"{name} must be a torch.Tensor")
    _require(t.is_cuda, f"{name} must be a CUDA/HIP tensor (easyhec_amd has no CPU path)")
    _require(t.dtype == dtype, f"{name} must have dtype {dtype}, got {t.dtype}")


class RasterizeCudaContext:
    """Replaces ``dr.RasterizeCudaContext(device=None)``: owns the rasterizer's binning scratch on one device."""

    def __init__(self, device=None):
        _require(torch.cuda.is_available(), "RasterizeCudaContext: no HIP device is available")
        if device is None:
            idx = torch.cuda.current_device()
        else:
            dev = torch.device(device)
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.device_index = idx
        self.device = torch.device("cuda", idx)
        handle = ctypes.c_void_p()
        with torch.cuda.device(idx):
            _lib.check(_lib.lib().ehr_ctx_create(idx, ctypes.byref(handle)), "ehr_ctx_create")
        self._h = handle
        self._plan = None

    @property
    def handle(self):
        return self._h

    def scratch_bytes(self):
        """Device memory this context holds at the moment (``ehr_ctx_scratch_bytes``)."""
        return int(_lib.lib().ehr_ctx_scratch_bytes(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().ehr_ctx_destroy(h)
            except Exception:
                pass


# nvdiffrast's OpenGL context plays the same role; there is no GL on this path, the HIP rasterizer serves both.
RasterizeGLContext = RasterizeCudaContext


class _RasterizeFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, glctx, pos, tri, resolution, ranges, grad_db, flags):
        H, W = int(resolution[0]), int(resolution[1])
        if ranges is None:
            B, V = pos.shape[0], pos.shape[1]
            rptr = None
        else:
            B, V = ranges.shape[0], pos.shape[0]
            ranges = ranges.contiguous()
            rptr = ctypes.c_void_p(ranges.data_ptr())
        T = tri.shape[0]
        rast = torch.empty((B, H, W, 4), dtype=torch.float32, device=pos.device)
        db = torch.empty((B, H, W, 4), dtype=torch.float32, device=pos.device) if grad_db else None
        with torch.cuda.device(pos.device):
            _lib.check(_lib.lib().ehr_rasterize_fwd(glctx.handle, _lib.ptr(pos), _lib.ptr(tri), rptr, B, V, T, H, W,
                                                    _lib.ptr(rast), _lib.ptr(db), _lib.ptr(flags), _stream()), "rasterize")
        if db is None:
            db = torch.empty((B, H, W, 0), dtype=torch.float32, device=pos.device)
        ctx.save_for_backward(pos, tri, rast)
        ctx.flags = flags
        ctx.range_mode = ranges is not None
        if not grad_db:
            ctx.mark_non_differentiable(db)
        # no gradient reaches `rast` on EasyHeC's path (the colour is constant, antialias returns none for it): without this
        # autograd would materialise a zero image and run the backward kernel on it, per (view, link)
        ctx.set_materialize_grads(False)
        return rast, db

    @staticmethod
    def backward(ctx, dy, ddb):
        if dy is None and ddb is None:
            return None, None, None, None, None, None, None
        pos, tri, rast = ctx.saved_tensors
        B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
        V, T = pos.shape[-2], tri.shape[0]
        g = torch.zeros_like(pos)
        with torch.cuda.device(pos.device):
            if dy is not None:   # through (u, v)
                dy = dy.contiguous()
                _lib.check(_lib.lib().ehr_rasterize_grad(_lib.ptr(pos), _lib.ptr(tri), _lib.ptr(rast), _lib.ptr(dy),
                                                         int(ctx.range_mode), B, V, T, H, W, _lib.ptr(g),
                                                         _lib.ptr(ctx.flags), _stream()), "rasterize backward")
            if ddb is not None:  # through the pixel differentials of (u, v) (nobody on EasyHeC's path asks for this)
                ddb = ddb.contiguous()
                _lib.check(_lib.lib().ehr_rasterize_grad_db(_lib.ptr(pos), _lib.ptr(tri), _lib.ptr(rast), _lib.ptr(ddb),
                                                            int(ctx.range_mode), B, V, T, H, W, _lib.ptr(g), _stream()),
                           "rasterize backward (rast_db)")
        return None, g, None, None, None, None, None


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """``dr.rasterize``: pos [B,V,4] (instance mode) or [V,4] with ``ranges`` [B,2] int32 CPU tensor (range mode);
    tri [T,3] int32; resolution (H, W).  Returns ``(rast [B,H,W,4] = (u, v, z/w, triangle_id + 1), rast_db)``.

    ``rast_db`` holds the screen-space derivatives of (u, v) when ``grad_db`` is set; gradients flow back to ``pos``
    through both outputs (EasyHeC never consumes rast_db: nvdiffrast_renderer.py:39 discards it)."""
    _require(isinstance(glctx, RasterizeCudaContext), "glctx must be a RasterizeCudaContext")
    _check_dev("pos", pos, torch.float32)
    _check_dev("tri", tri, torch.int32)
    _require(len(resolution) == 2, "resolution must be [height, width]")
    _require(tri.dim() == 2 and tri.shape[1] == 3, "tri must have shape [>0, 3]")
    _require(int(resolution[0]) > 0 and int(resolution[1]) > 0, "resolution must be [>0, >0]")
    _require(pos.device.index == glctx.device_index, "pos must be on the context's device")
    if ranges is None:
        _require(pos.dim() == 3 and pos.shape[0] > 0 and pos.shape[2] == 4,
                 "instance mode - pos must have shape [>0, >0, 4]")
    else:
        _require(pos.dim() == 2 and pos.shape[1] == 4, "range mode - pos must have shape [>0, 4]")
        _require(isinstance(ranges, torch.Tensor) and not ranges.is_cuda and ranges.dtype == torch.int32 and
                 ranges.dim() == 2 and ranges.shape[1] == 2,
                 "range mode - ranges must be a CPU int32 tensor with shape [>0, 2]")
    B = pos.shape[0] if ranges is None else ranges.shape[0]
    flags = torch.empty((_lib.lib().ehr_tile_flags_bytes(B, int(resolution[0]), int(resolution[1])),), dtype=torch.uint8,
                        device=pos.device)
    rast, db = _RasterizeFunc.apply(glctx, pos.contiguous(), tri.contiguous(), resolution, ranges, bool(grad_db), flags)
    _attach_flags(rast, flags)
    return rast, db


class _InterpolateFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri, flags):
        B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
        Ba, V, A = attr.shape
        out = torch.empty((B, H, W, A), dtype=torch.float32, device=rast.device)
        with torch.cuda.device(rast.device):
            _lib.check(_lib.lib().ehr_interpolate_fwd(_lib.ptr(attr), _lib.ptr(rast), _lib.ptr(tri), B, Ba, V,
                                                      tri.shape[0], A, H, W, _lib.ptr(out), _lib.ptr(flags), _stream()),
                       "interpolate")
        ctx.save_for_backward(attr, rast, tri)
        ctx.flags = flags
        return out

    @staticmethod
    def backward(ctx, dy):
        attr, rast, tri = ctx.saved_tensors
        B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
        Ba, V, A = attr.shape
        # EasyHeC interpolates constant all-ones colours (nvdiffrast_renderer.py:41-42): no attribute gradient is
        # wanted, so neither the zero-fill nor the per-pixel atomics are paid
        g_attr = torch.zeros_like(attr) if ctx.needs_input_grad[0] else None
        g_rast = torch.empty_like(rast)
        dy = dy.contiguous()
        with torch.cuda.device(rast.device):
            _lib.check(_lib.lib().ehr_interpolate_grad(_lib.ptr(attr), _lib.ptr(rast), _lib.ptr(tri), _lib.ptr(dy), B,
                                                       Ba, V, tri.shape[0], A, H, W, _lib.ptr(g_attr),
                                                       _lib.ptr(g_rast), _lib.ptr(ctx.flags), _stream()), "interpolate backward")
        return g_attr, g_rast, None, None


class _InterpolateDaFunc(torch.autograd.Function):
    """Attribute pixel differentials: out_da = d attr / d(X, Y) from rast_db (interpolate's second output)."""

    @staticmethod
    def forward(ctx, attr, rast, rast_db, tri, idx):
        B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
        Ba, V, A = attr.shape
        D = A if idx is None else int(idx.shape[0])
        out = torch.empty((B, H, W, 2 * D), dtype=torch.float32, device=rast.device)
        with torch.cuda.device(rast.device):
            _lib.check(_lib.lib().ehr_interpolate_da_fwd(_lib.ptr(attr), _lib.ptr(rast), _lib.ptr(rast_db), _lib.ptr(tri),
                                                         _lib.ptr(idx), B, Ba, V, tri.shape[0], A, D, H, W, _lib.ptr(out),
                                                         _stream()), "interpolate (differentials)")
        ctx.save_for_backward(attr, rast, rast_db, tri, idx)
        return out

    @staticmethod
    def backward(ctx, dy):
        attr, rast, rast_db, tri, idx = ctx.saved_tensors
        B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
        Ba, V, A = attr.shape
        D = A if idx is None else int(idx.shape[0])
        g_attr = torch.zeros_like(attr) if ctx.needs_input_grad[0] else None
        g_db = torch.empty_like(rast_db) if ctx.needs_input_grad[2] else None
        if g_attr is not None or g_db is not None:
            dy = dy.contiguous()
            with torch.cuda.device(rast.device):
                _lib.check(_lib.lib().ehr_interpolate_da_grad(_lib.ptr(attr), _lib.ptr(rast), _lib.ptr(rast_db), _lib.ptr(tri),
                                                              _lib.ptr(idx), _lib.ptr(dy), B, Ba, V, tri.shape[0], A, D, H,
                                                              W, _lib.ptr(g_attr), _lib.ptr(g_db), _stream()),
                           "interpolate (differentials) backward")
        return g_attr, None, g_db, None, None  # (the differentials do not depend on (u, v): nothing for rast)


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """``dr.interpolate``: attr [B or 1, V, A] (or [V, A] in range mode), rast from :func:`rasterize`, tri [T,3].
    Returns ``(out [B,H,W,A], out_da)``.  ``out_da`` is empty unless ``diff_attrs`` ('all' or a list of attribute indices)
    and ``rast_db`` (rasterize's second output) are given: then it holds the attributes' pixel differentials
    [B,H,W,2*len(diff_attrs)] = (d/dX, d/dY) per listed attribute.  (EasyHeC passes neither: nvdiffrast_renderer.py:42.)"""
    _check_dev("attr", attr, torch.float32)
    _check_dev("rast", rast, torch.float32)
    _check_dev("tri", tri, torch.int32)
    _require(rast.dim() == 4 and rast.shape[3] == 4, "rast must have shape [>0, >0, >0, 4]")
    _require(tri.dim() == 2 and tri.shape[1] == 3, "tri must have shape [>0, 3]")
    if attr.dim() == 2:
        attr = attr[None]
    _require(attr.dim() == 3, "attr must have shape [>0, >0, >0] or [>0, >0]")
    _require(attr.shape[0] in (1, rast.shape[0]), "attr batch must be 1 or match rast")
    flags = _flags_of(rast)
    attr, rast, tri = attr.contiguous(), rast.contiguous(), tri.contiguous()
    out = _InterpolateFunc.apply(attr, rast, tri, flags)
    if diff_attrs is None:
        return out, torch.empty((out.shape[0], out.shape[1], out.shape[2], 0), dtype=torch.float32, device=out.device)
    _require(rast_db is not None, "interpolate: diff_attrs needs rast_db (the second output of rasterize)")
    _check_dev("rast_db", rast_db, torch.float32)
    _require(rast_db.shape == rast.shape, "rast_db must have the shape of rast (rasterize with grad_db=True)")
    A = attr.shape[2]
    if isinstance(diff_attrs, str):
        _require(diff_attrs == "all", "diff_attrs must be 'all' or a list of attribute indices")
        idx = None
    else:
        lst = [int(i) for i in diff_attrs]
        _require(len(lst) > 0 and all(0 <= i < A for i in lst), "diff_attrs: attribute indices out of range")
        idx = None if lst == list(range(A)) else torch.tensor(lst, dtype=torch.int32).to(rast.device)
    out_da = _InterpolateDaFunc.apply(attr, rast, rast_db.contiguous(), tri, idx)
    return out, out_da


class TopologyHash:
    """Result of :func:`antialias_construct_topology_hash`: the opposite vertex across every triangle edge."""

    def __init__(self, opp, num_triangles):
        self.opp = opp
        self.num_triangles = num_triangles


_TOPO_CACHE = {}


def antialias_construct_topology_hash(tri):
    """``dr.antialias_construct_topology_hash(tri)``: build once per mesh, pass as ``topology_hash=``."""
    _check_dev("tri", tri, torch.int32)
    _require(tri.dim() == 2 and tri.shape[1] == 3, "tri must have shape [>0, 3]")
    tri = tri.contiguous()
    T = tri.shape[0]
    lib = _lib.lib()
    opp = torch.empty((T, 3), dtype=torch.int32, device=tri.device)
    nbytes = lib.ehr_topology_scratch_bytes(T)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=tri.device)
    with torch.cuda.device(tri.device):
        _lib.check(lib.ehr_antialias_topology(_lib.ptr(tri), T, _lib.ptr(opp), _lib.ptr(scratch), nbytes, _stream()),
                   "antialias_construct_topology_hash")
    return TopologyHash(opp, T)


class _AntialiasFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, opp, boost, flags):
        B, H, W, C = color.shape
        V, T = pos.shape[-2], tri.shape[0]
        range_mode = int(pos.dim() == 2)
        lib = _lib.lib()
        out = torch.empty_like(color)
        work = torch.empty((lib.ehr_antialias_work_bytes(B, H, W),), dtype=torch.uint8, device=color.device)
        # the buffer backward() accumulates pos's gradient into is cleared by the forward kernel on its way (a fill per call
        # otherwise: 64 launches per step of the reference's schedule); backward() takes it once
        g_pos = torch.empty_like(pos) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(color.device):
            _lib.check(lib.ehr_antialias_fwd_zg(_lib.ptr(color), _lib.ptr(rast), _lib.ptr(pos), _lib.ptr(tri),
                                                _lib.ptr(opp), range_mode, B, V, T, H, W, C, _lib.ptr(out),
                                                _lib.ptr(work), _lib.ptr(flags), _lib.ptr(g_pos), _stream()), "antialias")
        ctx.save_for_backward(color, rast, pos, tri, work)
        ctx.g_pos = g_pos
        ctx.boost = float(boost)
        return out

    @staticmethod
    def backward(ctx, dy):
        color, rast, pos, tri, work = ctx.saved_tensors
        B, H, W, C = color.shape
        V, T = pos.shape[-2], tri.shape[0]
        range_mode = int(pos.dim() == 2)
        dy = dy.contiguous()
        # (EasyHeC's colour is interpolated from constant attributes: nothing asks for its gradient, and the full-image copy
        #  of dy it starts from is a kernel per (view, link))
        g_color = torch.empty_like(color) if ctx.needs_input_grad[0] else None
        g_pos, ctx.g_pos = ctx.g_pos, None   # cleared by the forward pass; a second backward() of a retained graph fills its own
        if g_pos is None:
            g_pos = torch.zeros_like(pos)
        with torch.cuda.device(color.device):
            _lib.check(_lib.lib().ehr_antialias_grad(_lib.ptr(color), _lib.ptr(rast), _lib.ptr(pos), _lib.ptr(tri),
                                                     _lib.ptr(dy), _lib.ptr(work), range_mode, B, V, T, H, W, C,
                                                     _lib.ptr(g_color), _lib.ptr(g_pos), _stream()),
                       "antialias backward")
        if ctx.boost != 1.0:
            g_pos = g_pos * ctx.boost
        return g_color, None, g_pos, None, None, None, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """``dr.antialias``: blends across silhouette edges so that coverage becomes differentiable w.r.t. ``pos``.
    Without ``topology_hash`` the edge topology is rebuilt on every call, as nvdiffrast does
    (nvdiffrast_renderer.py:43 passes none)."""
    _check_dev("color", color, torch.float32)
    _check_dev("rast", rast, torch.float32)
    _check_dev("pos", pos, torch.float32)
    _check_dev("tri", tri, torch.int32)
    _require(color.dim() == 4 and color.shape[3] > 0, "color must have shape [>0, >0, >0, >0]")
    _require(rast.dim() == 4 and rast.shape[3] == 4 and rast.shape[:3] == color.shape[:3],
             "rast must have shape [B, H, W, 4] matching color")
    _require(tri.dim() == 2 and tri.shape[1] == 3, "tri must have shape [>0, 3]")
    _require((pos.dim() == 3 and pos.shape[2] == 4 and pos.shape[0] == color.shape[0]) or
             (pos.dim() == 2 and pos.shape[1] == 4), "pos must have shape [B, >0, 4] or [>0, 4]")
    tri = tri.contiguous()
    if topology_hash is None:
        # nvdiffrast rebuilds the topology on every such call; the result only depends on tri, so the last few are kept
        # (keyed by the tensor's storage and version counter: an in-place edit of tri invalidates the entry)
        key = (tri.data_ptr(), tri._version, tri.shape[0], tri.device.index)
        topology_hash = _TOPO_CACHE.get(key)
        if topology_hash is None:
            topology_hash = antialias_construct_topology_hash(tri)
            if len(_TOPO_CACHE) >= 64:
                _TOPO_CACHE.clear()
            _TOPO_CACHE[key] = topology_hash
            topology_hash._keepalive = tri  # the key's address stays this tensor's while the entry lives
    _require(isinstance(topology_hash, TopologyHash) and topology_hash.num_triangles == tri.shape[0],
             "topology_hash does not belong to tri")
    return _AntialiasFunc.apply(color.contiguous(), rast.contiguous(), pos.contiguous(), tri, topology_hash.opp,
                                pos_gradient_boost, _flags_of(rast))
