#!/usr/bin/env python
"""Dump REAL nvdiffrast outputs for this repo's committed inputs, so that parity stops being "unpinned".

Run this on any machine with an NVIDIA GPU, PyTorch and nvdiffrast (the package EasyHeC's requirements.txt:29 installs
from git HEAD) from a checkout of THIS repository -- it imports torch, nvdiffrast and numpy, nothing else, and reads only
data files of this repository:

    pip install git+https://github.com/NVlabs/nvdiffrast.git
    python tools/dump_nvdiffrast_golden.py            # writes tests/golden/nvdiffrast_*.npz

and commit the files it writes.  Every file carries its inputs next to nvdiffrast's outputs, plus the versions that
produced them.  `tests/test_nvdiffrast_golden.py` (CPU: the oracle) and `tests/test_gpu_nvdiffrast_golden.py` (the HIP
ops and the fused path) pick the files up when they are present and are skipped when they are not; they report
mismatching-pixel counts and L-infinity differences of rast / antialiased colour / position gradients.

What is dumped (the four entry points the reference calls, in its order -- structures/nvdiffrast_renderer.py:33-47:
rasterize -> interpolate(ones) -> antialias -> flip; rb_solver.py:60-72: per-link render, sum, clamp, SSE):

  ops_random        the inputs of tests/golden/ops_random_72x104.npz (random shared-vertex soup, 72x104, arbitrary attributes):
                    rast, rast_db, interpolated colour, antialiased colour, d/d pos and d/d attr for the stored dy
  fused_xarm7       the inputs of tests/golden/fused_xarm7_160x120.npz (xArm7, 8 links, 2 views): per (view, link) triangle-id
                    image, the composite mask, the per-view loss and d loss / d MVP
  config1           the inputs of tests/golden/config1_zeropos_320x240.npz (xarm7_zeropos mesh as one link, 320x240)
  adv_depth         ADVERSARIAL: two interleaved sheets of one mesh, 1e-4 apart in depth, cut by a silhouette -- which
                    triangle wins the depth test at a silhouette pixel decides which edge antialias inspects
                    (nvdiffrast's CUDA rasterizer tests fixed-point depth from snapped vertices; this repo tests float z/w
                    from the unsnapped ones: DESIGN.md section 3, "known open deviation")
  adv_slivers       ADVERSARIAL: slivers thinner than 1/16 pixel at random slopes and sub-pixel offsets, tiny triangles around
                    pixel centres (snapping, top-left rule, degenerate-after-snapping culling)
  adv_centres       ADVERSARIAL: axis-aligned and 45-degree edges exactly through pixel centres, shared edges (single
                    ownership), both windings
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def versions(dr, module):
    top = __import__(module.split(".")[0])
    return {"module": module, "version": str(getattr(top, "__version__", "unknown")), "torch": torch.__version__,
            "cuda": str(torch.version.cuda), "hip": str(getattr(torch.version, "hip", None)),
            "gpu": torch.cuda.get_device_name(0), "context": type(dr_ctx(dr)).__name__}


_CTX = {}


def dr_ctx(dr):
    if "ctx" not in _CTX:
        _CTX["ctx"] = dr.RasterizeCudaContext()
    return _CTX["ctx"]


def three_ops(dr, pos, tri, attr, H, W, dy=None):
    """rasterize -> interpolate -> antialias on one image; gradients for dy if given.  pos [V,4], tri [T,3], attr [V,C]."""
    dev = torch.device("cuda")
    tp = torch.tensor(pos[None], device=dev, requires_grad=True)
    tt = torch.tensor(tri, device=dev)
    ta = torch.tensor(attr[None], device=dev, requires_grad=True)
    rast, db = dr.rasterize(dr_ctx(dr), tp, tt, resolution=[H, W])
    col, _ = dr.interpolate(ta, rast, tt)
    aa = dr.antialias(col, rast, tp, tt)
    out = {"rast": rast.detach().cpu().numpy(), "db": db.detach().cpu().numpy(), "col": col.detach().cpu().numpy(),
           "aa": aa.detach().cpu().numpy()}
    if dy is not None:
        (aa * torch.tensor(dy, device=dev)).sum().backward()
        out["grad_pos"] = tp.grad.cpu().numpy()
        out["grad_attr"] = ta.grad.cpu().numpy()
    return out


def link_masks(dr, link_meshes, mvp, ref, H, W):
    """The reference's per-link silhouette render and composite for B views x L links, from the clip matrices.
    link_meshes: [(verts [V,3], faces [T,3] int32)]; mvp [B,L,4,4]; ref [B,H,W].  Returns mask, loss, grad_mvp and the
    per-(view, link) triangle-id images (rast[..., 3], GL row order)."""
    dev = torch.device("cuda")
    B, L = mvp.shape[:2]
    tm = torch.tensor(mvp, device=dev, requires_grad=True)
    tref = torch.tensor(ref, device=dev)
    masks, ids = [], np.zeros((B, L, H, W), np.float32)
    clip = [[None] * L for _ in range(B)]
    for b in range(B):
        per_link = []
        for l, (v, f) in enumerate(link_meshes):
            tv = torch.tensor(v, device=dev)
            tf = torch.tensor(f, device=dev)
            posw = torch.cat([tv, torch.ones([tv.shape[0], 1], device=dev)], dim=1)
            pos_clip = torch.matmul(posw, tm[b, l].t())[None, ...]
            rast, _ = dr.rasterize(dr_ctx(dr), pos_clip, tf, resolution=[H, W])
            col, _ = dr.interpolate(torch.ones_like(tv)[None, ...], rast, tf)
            col = dr.antialias(col, rast, pos_clip, tf)
            per_link.append(torch.flip(col[0, :, :, 0], dims=[0]))
            ids[b, l] = rast[0, :, :, 3].detach().cpu().numpy()
            clip[b][l] = pos_clip[0].detach().cpu().numpy()
        masks.append(torch.stack(per_link).sum(0).clamp(max=1))
    mask = torch.stack(masks)
    loss_b = ((mask - tref) ** 2).sum(dim=(1, 2))
    loss_b.sum().backward()
    # the clip-space positions torch.matmul produced here (cuBLAS rounds the 4-term dot products its own way): with them a
    # consumer can run its three ops on IDENTICAL inputs -- renderer semantics apart from GEMM rounding, which alone flips
    # isolated razor-edge pixels (this repo's fused path transforms with an fma chain; its own three ops fed by torch.matmul
    # differ from it at 1 pixel of 38 400 on the 160x120 fixture)
    pos_clip_all = np.stack([np.concatenate(clip[b], axis=0) for b in range(B)])
    return {"mask": mask.detach().cpu().numpy(), "loss": loss_b.detach().cpu().numpy(), "grad_mvp": tm.grad.cpu().numpy(),
            "tri_ids": ids, "pos_clip": pos_clip_all}


def load_links(name):
    z = np.load(os.path.join(ROOT, "easyhec_amd", "assets", f"{name}.npz"), allow_pickle=False)
    vo, to = z["vert_offsets"], z["tri_offsets"]
    return [(z["vertices"][vo[i]:vo[i + 1]].astype(np.float32), z["faces"][to[i]:to[i + 1]].astype(np.int32))
            for i in range(len(vo) - 1)]


# ---- adversarial inputs (deterministic; stored in the output next to nvdiffrast's answers) ---------------------------

def adv_depth(H=96, W=128):
    """Two interleaved triangle sheets of ONE mesh, 1e-4 apart in z/w, tilted against each other so that the nearer one
    changes along the silhouette; a diagonal cut gives the silhouette sub-pixel variety."""
    rng = np.random.default_rng(7)
    n = 14
    xs = np.linspace(-0.7, 0.7, n + 1)
    pts, tris = [], []
    for sheet in range(2):
        base = len(pts)
        for j in range(n + 1):
            for i in range(n + 1):
                x = xs[i] + rng.uniform(-0.012, 0.012) + 0.013 * sheet
                y = xs[j] + rng.uniform(-0.012, 0.012) - 0.009 * sheet
                z = 0.30 + (1e-4 if sheet else 0.0) + (2e-4 * x if sheet else -2e-4 * x)
                w = 1.0 + 0.15 * x + 0.1 * y
                pts.append([x * w, y * w, z * w, w])
        for j in range(n):
            for i in range(n):
                if xs[i] + xs[j] > 0.55:  # the cut: a diagonal silhouette through the sheets
                    continue
                a = base + j * (n + 1) + i
                tris.append([a, a + 1, a + n + 2])
                tris.append([a, a + n + 2, a + n + 1])
    return np.array(pts, np.float32), np.array(tris, np.int32), H, W


def adv_slivers(H=64, W=96):
    rng = np.random.default_rng(11)
    pts, tris = [], []
    for k in range(160):  # slivers: two long edges, width from 1/64 to 1/4 pixel
        cx, cy = rng.uniform(-0.9, 0.9, 2)
        ang = rng.uniform(0, np.pi)
        ln = rng.uniform(0.05, 0.5)
        wd = rng.uniform(1.0 / 64, 0.25) * 2.0 / W
        d = np.array([np.cos(ang), np.sin(ang)])
        nrm = np.array([-d[1], d[0]])
        p0, p1, p2 = np.array([cx, cy]) - d * ln / 2, np.array([cx, cy]) + d * ln / 2, np.array([cx, cy]) + nrm * wd
        z = rng.uniform(-0.5, 0.5)
        b = len(pts)
        for p in (p0, p1, p2):
            pts.append([p[0], p[1], z, 1.0])
        tris.append([b, b + 1, b + 2] if k % 2 else [b, b + 2, b + 1])
    for k in range(120):  # tiny triangles around pixel centres
        ix, iy = rng.integers(2, W - 2), rng.integers(2, H - 2)
        c = np.array([(ix + 0.5) * 2.0 / W - 1.0, (iy + 0.5) * 2.0 / H - 1.0]) + rng.uniform(-0.6, 0.6, 2) * np.array([2.0 / W, 2.0 / H])
        z = rng.uniform(-0.5, 0.5)
        b = len(pts)
        for _ in range(3):
            q = c + rng.uniform(-0.7, 0.7, 2) * np.array([2.0 / W, 2.0 / H])
            pts.append([q[0], q[1], z, 1.0])
        tris.append([b, b + 1, b + 2])
    return np.array(pts, np.float32), np.array(tris, np.int32), H, W


def adv_centres(H=48, W=64):
    """Quads whose edges run exactly through pixel centres (x = (i + 0.5) 2 / W - 1), shared diagonals at 45 degrees
    through pixel centres, both windings, abutting quads (each covered pixel must have exactly one owner)."""
    pts, tris = [], []

    def cx(i):
        return (i + 0.5) * 2.0 / W - 1.0

    def cy(j):
        return (j + 0.5) * 2.0 / H - 1.0

    def quad(x0, y0, x1, y1, z, flip):
        b = len(pts)
        for (x, y) in ((x0, y0), (x1, y0), (x1, y1), (x0, y1)):
            pts.append([x, y, z, 1.0])
        if flip:
            tris.extend([[b, b + 2, b + 1], [b, b + 3, b + 2]])
        else:
            tris.extend([[b, b + 1, b + 2], [b, b + 2, b + 3]])

    quad(cx(4), cy(4), cx(20), cy(16), 0.1, False)     # edges through pixel centres, square -> 45-degree diagonal
    quad(cx(20), cy(4), cx(36), cy(16), 0.1, True)     # abutting, other winding (separate vertices: no shared topology)
    quad(cx(4), cy(16), cx(36), cy(28), 0.2, False)    # abutting from above
    quad(cx(10), cy(30), cx(10) + 16 * 2.0 / W, cy(30) + 12 * 2.0 / H, -0.2, False)
    b = len(pts)                                        # a fan with shared vertices (topology exists): centre on a pixel centre
    pts.append([cx(48), cy(24), 0.0, 1.0])
    ring = [(cx(56), cy(24)), (cx(54), cy(30)), (cx(48), cy(32)), (cx(42), cy(30)), (cx(40), cy(24)), (cx(42), cy(18)),
            (cx(48), cy(16)), (cx(54), cy(18))]
    for (x, y) in ring:
        pts.append([x, y, 0.0, 1.0])
    for k in range(8):
        tris.append([b, b + 1 + k, b + 1 + (k + 1) % 8])
    return np.array(pts, np.float32), np.array(tris, np.int32), H, W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=GOLD)
    ap.add_argument("--module", default="nvdiffrast.torch", help="the module that provides the four entry points.  The default is "
                    "the real thing; `easyhec_amd.dr` makes a DRY RUN of this script and of the consumer tests on an AMD box "
                    "(write it somewhere else with --out: such files prove nothing about parity and must not be committed)")
    ap.add_argument("--prefix", default="nvdiffrast")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        sys.exit("needs a GPU (nvdiffrast's RasterizeCudaContext)")
    if args.module != "nvdiffrast.torch" and os.path.abspath(args.out) == os.path.abspath(GOLD):
        sys.exit("a dry run with another module must not write into tests/golden (use --out)")
    import importlib
    sys.path.insert(0, ROOT)
    dr = importlib.import_module(args.module)
    ver = versions(dr, args.module)
    vstr = np.array([f"{k}={v}" for k, v in sorted(ver.items())])
    os.makedirs(args.out, exist_ok=True)

    g = np.load(os.path.join(GOLD, "ops_random_72x104.npz"))
    H, W = g["rast"].shape[1:3]
    o = three_ops(dr, g["pos"], g["tri"], g["attr"][0], H, W, dy=g["dy"])
    np.savez_compressed(os.path.join(args.out, f"{args.prefix}_ops_random_72x104.npz"), kind="ops", versions=vstr, pos=g["pos"],
                        tri=g["tri"], attr=g["attr"], dy=g["dy"], H=H, W=W, **o)

    g = np.load(os.path.join(GOLD, "fused_xarm7_160x120.npz"))
    H, W = int(g["H"]), int(g["W"])
    ref = np.unpackbits(g["ref"])[:2 * H * W].reshape(2, H, W).astype(np.float32)
    o = link_masks(dr, load_links("xarm7"), g["mvp"], ref, H, W)
    np.savez_compressed(os.path.join(args.out, f"{args.prefix}_fused_xarm7_160x120.npz"), kind="fused", robot="xarm7", versions=vstr,
                        mvp=g["mvp"], ref=np.packbits(ref > 0.5), H=H, W=W, **o)

    g = np.load(os.path.join(GOLD, "config1_zeropos_320x240.npz"))
    z = np.load(os.path.join(GOLD, "xarm7_zeropos.npz"))
    H, W = g["mask"].shape[1:]
    o = link_masks(dr, [(z["vertices"].astype(np.float32), z["faces"].astype(np.int32))], g["mvp"], np.zeros((1, H, W), np.float32), H, W)
    np.savez_compressed(os.path.join(args.out, f"{args.prefix}_config1_zeropos_320x240.npz"), kind="fused", robot="zeropos", versions=vstr,
                        mvp=g["mvp"], ref=np.packbits(np.zeros((1, H, W), bool)), H=H, W=W, **o)

    for name, make in (("adv_depth", adv_depth), ("adv_slivers", adv_slivers), ("adv_centres", adv_centres)):
        pos, tri, H, W = make()
        rng = np.random.default_rng(len(name))
        attr = np.ones((pos.shape[0], 1), np.float32)  # the reference's constant colour: coverage is the signal
        dy = rng.normal(size=(1, H, W, 1)).astype(np.float32)
        o = three_ops(dr, pos, tri, attr, H, W, dy=dy)
        np.savez_compressed(os.path.join(args.out, f"{args.prefix}_{name}.npz"), kind="ops", versions=vstr, pos=pos, tri=tri,
                            attr=attr[None], dy=dy, H=H, W=W, **o)
        print(name, "covered", int((o["rast"][..., 3] > 0).sum()), "fractional", int(((o["aa"] > 0) & (o["aa"] < 1)).sum()))
    print("wrote", sorted(f for f in os.listdir(args.out) if f.startswith(args.prefix + "_")))


if __name__ == "__main__":
    main()
