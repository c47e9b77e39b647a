"""ctypes/numpy binding of oracle/libehr_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
(easyhec_amd/) never does.  Each function cites the reference call site it restates; the arithmetic itself is a
restatement of nvdiffrast's published algorithm (not in /root/reference) -- parity unpinned, see ehr_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libehr_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile the C restatement (gcc).  Called by __graft_entry__.build()."""
    src = os.path.join(_HERE, "ehr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _fp(a):
    return None if a is None else a.ctypes.data_as(_f)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_i)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed with {rc}")


def rasterize(pos, tri, resolution, ranges=None, grad_db=True):
    """dr.rasterize (easyhec/structures/nvdiffrast_renderer.py:39).  pos [B,V,4] or [V,4]+ranges [B,2]."""
    pos, tri = _f32(pos), _i32(tri)
    H, W = int(resolution[0]), int(resolution[1])
    if ranges is None:
        assert pos.ndim == 3
        B, V = pos.shape[0], pos.shape[1]
        rg = None
    else:
        assert pos.ndim == 2
        rg = _i32(ranges)
        B, V = rg.shape[0], pos.shape[0]
    T = tri.shape[0]
    rast = np.empty((B, H, W, 4), np.float32)
    db = np.empty((B, H, W, 4), np.float32) if grad_db else None
    _chk(lib().ehro_rasterize_fwd(_fp(pos), _ip(tri), _ip(rg), B, V, T, H, W, _fp(rast), _fp(db)), "rasterize_fwd")
    return rast, db


def rasterize_grad(pos, tri, rast, dy, range_mode=False):
    pos, tri, rast, dy = _f32(pos), _i32(tri), _f32(rast), _f32(dy)
    B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
    V = pos.shape[-2]
    g = np.zeros_like(pos)
    _chk(lib().ehro_rasterize_grad(_fp(pos), _ip(tri), _fp(rast), _fp(dy), int(range_mode), B, V, tri.shape[0], H, W,
                                   _fp(g)), "rasterize_grad")
    return g


def rasterize_grad_db(pos, tri, rast, ddb, range_mode=False):
    """d(rast_db)/d(pos) contracted with ddb (the gradient w.r.t. rasterize's second output)."""
    pos, tri, rast, ddb = _f32(pos), _i32(tri), _f32(rast), _f32(ddb)
    B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
    V = pos.shape[-2]
    g = np.zeros_like(pos)
    _chk(lib().ehro_rasterize_grad_db(_fp(pos), _ip(tri), _fp(rast), _fp(ddb), int(range_mode), B, V, tri.shape[0], H, W,
                                      _fp(g)), "rasterize_grad_db")
    return g


def interpolate(attr, rast, tri):
    """dr.interpolate (nvdiffrast_renderer.py:42).  attr [1 or B,V,A]."""
    attr, rast, tri = _f32(attr), _f32(rast), _i32(tri)
    B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
    Ba, V, A = attr.shape
    out = np.empty((B, H, W, A), np.float32)
    _chk(lib().ehro_interpolate_fwd(_fp(attr), _fp(rast), _ip(tri), B, Ba, V, tri.shape[0], A, H, W, _fp(out)),
         "interpolate_fwd")
    return out


def interpolate_grad(attr, rast, tri, dy):
    attr, rast, tri, dy = _f32(attr), _f32(rast), _i32(tri), _f32(dy)
    B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
    Ba, V, A = attr.shape
    ga = np.zeros_like(attr)
    gr = np.empty_like(rast)
    _chk(lib().ehro_interpolate_grad(_fp(attr), _fp(rast), _ip(tri), _fp(dy), B, Ba, V, tri.shape[0], A, H, W,
                                     _fp(ga), _fp(gr)), "interpolate_grad")
    return ga, gr


def interpolate_da(attr, rast, rast_db, tri, diff_attrs="all"):
    """dr.interpolate's second output: attribute pixel differentials [B,H,W,2D] for the attribute indices ``diff_attrs``
    ('all' or a list); rast_db from :func:`rasterize`."""
    attr, rast, rast_db, tri = _f32(attr), _f32(rast), _f32(rast_db), _i32(tri)
    B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
    Ba, V, A = attr.shape
    idx = None if isinstance(diff_attrs, str) else _i32(np.asarray(diff_attrs))
    D = A if idx is None else idx.shape[0]
    out = np.empty((B, H, W, 2 * D), np.float32)
    _chk(lib().ehro_interpolate_da_fwd(_fp(attr), _fp(rast), _fp(rast_db), _ip(tri), None if idx is None else _ip(idx),
                                       B, Ba, V, tri.shape[0], A, D, H, W, _fp(out)), "interpolate_da_fwd")
    return out


def interpolate_da_grad(attr, rast, rast_db, tri, dy_da, diff_attrs="all"):
    attr, rast, rast_db, tri, dy_da = _f32(attr), _f32(rast), _f32(rast_db), _i32(tri), _f32(dy_da)
    B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
    Ba, V, A = attr.shape
    idx = None if isinstance(diff_attrs, str) else _i32(np.asarray(diff_attrs))
    D = A if idx is None else idx.shape[0]
    ga = np.zeros_like(attr)
    gdb = np.empty_like(rast_db)
    _chk(lib().ehro_interpolate_da_grad(_fp(attr), _fp(rast), _fp(rast_db), _ip(tri), None if idx is None else _ip(idx),
                                        _fp(dy_da), B, Ba, V, tri.shape[0], A, D, H, W, _fp(ga), _fp(gdb)),
         "interpolate_da_grad")
    return ga, gdb


def topology(tri):
    """Opposite vertex per (triangle, edge); -1 = boundary edge (nvdiffrast's topology hash, as a table)."""
    tri = _i32(tri)
    opp = np.empty((tri.shape[0], 3), np.int32)
    _chk(lib().ehro_topology(_ip(tri), tri.shape[0], _ip(opp)), "topology")
    return opp


def antialias(color, rast, pos, tri, opp=None):
    """dr.antialias (nvdiffrast_renderer.py:43)."""
    color, rast, pos, tri = _f32(color), _f32(rast), _f32(pos), _i32(tri)
    opp = topology(tri) if opp is None else _i32(opp)
    B, H, W, C = color.shape
    range_mode = int(pos.ndim == 2)
    V = pos.shape[-2]
    out = np.empty_like(color)
    _chk(lib().ehro_antialias_fwd(_fp(color), _fp(rast), _fp(pos), _ip(tri), _ip(opp), range_mode, B, V, tri.shape[0],
                                  H, W, C, _fp(out)), "antialias_fwd")
    return out


def antialias_grad(color, rast, pos, tri, dy, opp=None):
    color, rast, pos, tri, dy = _f32(color), _f32(rast), _f32(pos), _i32(tri), _f32(dy)
    opp = topology(tri) if opp is None else _i32(opp)
    B, H, W, C = color.shape
    range_mode = int(pos.ndim == 2)
    V = pos.shape[-2]
    gc = np.empty_like(color)
    gp = np.zeros_like(pos)
    _chk(lib().ehro_antialias_grad(_fp(color), _fp(rast), _fp(pos), _ip(tri), _ip(opp), _fp(dy), range_mode, B, V,
                                   tri.shape[0], H, W, C, _fp(gc), _fp(gp)), "antialias_grad")
    return gc, gp


def transform_pos(mtx, verts):
    """easyhec/utils/nvdiffrast_utils.py:14-18 -> [1,V,4]."""
    mtx, verts = _f32(mtx), _f32(verts)
    pos = np.empty((verts.shape[0], 4), np.float32)
    _chk(lib().ehro_transform_pos(_fp(mtx), _fp(verts), verts.shape[0], _fp(pos)), "transform_pos")
    return pos[None]


def render_mask_loss(verts, tris, tri_off, vert_off, mvp, ref, exact_interp=False, want_grad=True, want_mask=True):
    """Fused restatement of rb_solver.py:60-72 + backward to the per-(view,link) MVP.

    verts [V,3]; tris [T,3] global vertex indices sorted by link; tri_off/vert_off [L+1]; mvp [B,L,4,4];
    ref [B,H,W] (row 0 = top).  Returns mask [B,H,W], loss [B] (per-frame SSE), grad_mvp [B,L,4,4] = d loss_b/d mvp.
    """
    verts, tris, mvp, ref = _f32(verts), _i32(tris), _f32(mvp), _f32(ref)
    tri_off, vert_off = _i32(tri_off), _i32(vert_off)
    B, L = mvp.shape[0], mvp.shape[1]
    H, W = ref.shape[1], ref.shape[2]
    mask = np.empty((B, H, W), np.float32) if want_mask else None
    loss = np.empty((B,), np.float32)
    g = np.empty((B, L, 4, 4), np.float32) if want_grad else None
    _chk(lib().ehro_render_mask_loss(_fp(verts), _ip(tris), _ip(tri_off), _ip(vert_off), _fp(mvp), _fp(ref), B, L,
                                     verts.shape[0], tris.shape[0], H, W, int(exact_interp), _fp(mask), _fp(loss),
                                     _fp(g)), "render_mask_loss")
    return mask, loss, g


def mask_variance(verts, tris, vert_link, mvp, H, W, return_counts=False):
    """Space-explorer score (easyhec/modeling/models/rb_solve/space_explorer.py:152-165): for every candidate q the
    merged robot mesh is rasterized without antialiasing under the S camera poses (render_api.py:70-96,
    nvdiffrast_renderer.py:50-72, mask = rast[..., 2] > 0) and the unbiased per-pixel variance over the S binary masks
    is summed: sum_px c (S - c) / (S (S - 1)).  Returns the exact integer numerator score [Q] = sum_px c (S - c)
    (and the count images [Q,H,W] uint8, row 0 = top).

    verts [V,3]; tris [T,3]; vert_link [V]; mvp [Q,S,L,4,4]."""
    verts, tris, mvp = _f32(verts), _i32(tris), _f32(mvp)
    vert_link = np.asarray(vert_link).astype(np.int64)
    Q, S, L = mvp.shape[:3]
    score = np.zeros((Q,), np.int64)
    counts = np.zeros((Q, H, W), np.uint8) if return_counts else None
    for q in range(Q):
        pos = np.empty((S, verts.shape[0], 4), np.float32)
        for s in range(S):
            for l in range(L):
                sel = np.nonzero(vert_link == l)[0]
                if sel.size:
                    pos[s, sel] = transform_pos(mvp[q, s, l], verts[sel])[0]
        rast, _ = rasterize(pos, tris, (H, W), grad_db=False)
        c = (rast[..., 2] > 0).sum(0).astype(np.int64)  # [H,W], row 0 = bottom
        score[q] = int((c * (S - c)).sum())
        if return_counts:
            counts[q] = c[::-1].astype(np.uint8)
    return (score, counts) if return_counts else score


def set_num_threads(n):
    lib().ehro_set_num_threads(int(n))


def num_threads():
    return int(lib().ehro_num_threads())
