"""Ad-hoc GPU bring-up check: HIP ops vs the CPU oracle on random scenes and the xArm7 workload."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from easyhec_amd import dr, fused  # noqa: E402
from easyhec_amd.robot import load_robot  # noqa: E402
from easyhec_amd.synthetic import WORKLOADS, camera_Tc_c2b, make_views  # noqa: E402
from oracle import oracle as o  # noqa: E402

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))
ctx = dr.RasterizeCudaContext()
rng = np.random.default_rng(0)


def cmp(name, a, b, tol=0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b).max() if a.size else 0.0
    print(f"  {name}: max abs diff {d:.3e}  (ref max {np.abs(b).max():.3e})  {'OK' if d <= tol else 'DIFF'}")
    return d


for (H, W, nt) in [(64, 64, 50), (120, 200, 400), (256, 256, 3000)]:
    print("scene", H, W, nt)
    pos, tri = helpers.random_mesh(rng, nt)
    pos = pos[None]
    r_ref, db_ref = o.rasterize(pos, tri, [H, W])
    tp = torch.tensor(pos, device=dev, requires_grad=True)
    tt = torch.tensor(tri, device=dev)
    r, db = dr.rasterize(ctx, tp, tt, [H, W])
    torch.cuda.synchronize()
    cmp("tri id", r[..., 3].detach().cpu().numpy(), r_ref[..., 3])
    cmp("uvz", r[..., :3].detach().cpu().numpy(), r_ref[..., :3], 1e-6)
    cmp("db", db.detach().cpu().numpy(), db_ref, 1e-3)
    attr = rng.uniform(0, 1, size=(1, pos.shape[1], 3)).astype(np.float32)
    ta = torch.tensor(attr, device=dev, requires_grad=True)
    c, _ = dr.interpolate(ta, r, tt)
    c_ref = o.interpolate(attr, r_ref, tri)
    cmp("interp", c.detach().cpu().numpy(), c_ref, 1e-6)
    opp_ref = o.topology(tri)
    th = dr.antialias_construct_topology_hash(tt)
    cmp("topology", th.opp.cpu().numpy(), opp_ref)
    aa = dr.antialias(c, r, tp, tt)
    aa_ref = o.antialias(c_ref, r_ref, pos, tri)
    cmp("antialias", aa.detach().cpu().numpy(), aa_ref, 1e-5)
    gy = rng.normal(size=aa_ref.shape).astype(np.float32)
    (aa * torch.tensor(gy, device=dev)).sum().backward()
    gc_ref, gp_ref = o.antialias_grad(c_ref, r_ref, pos, tri, gy)
    ga_ref, gr_ref = o.interpolate_grad(attr, r_ref, tri, gc_ref)
    gp2_ref = o.rasterize_grad(pos, tri, r_ref, gr_ref)
    cmp("grad attr", ta.grad.cpu().numpy(), ga_ref, 1e-3)
    cmp("grad pos", tp.grad.cpu().numpy(), gp_ref + gp2_ref, 1e-2)

print("xarm7 fused")
rb = load_robot("xarm7")
for name, nv in [("xarm7_640x480_1view", 1), ("xarm7_1280x720_8view", 2)]:
    wl = WORKLOADS[name]
    H, W, K = wl["H"], wl["W"], wl["K"]
    q, lp = make_views(rb, nv)
    Tc = camera_Tc_c2b(radius=wl["radius"], lift=wl["lift"])
    mvp = helpers.mvp_numpy(K, H, W, Tc, lp)
    verts, tris, toff, voff = helpers.scene_arrays(rb)
    ref = (rng.uniform(size=(nv, H, W)) > 0.9).astype(np.float32)
    t0 = time.time()
    m_ref, l_ref, g_ref = o.render_mask_loss(verts, tris, toff, voff, mvp, ref)
    print("  oracle s", time.time() - t0)
    scene = fused.LinkScene([v for v, _ in rb.meshes], [f for _, f in rb.meshes], dev)
    tm = torch.tensor(mvp, device=dev, requires_grad=True)
    tr = torch.tensor(ref, device=dev)
    mask, loss = fused.render_mask_loss(ctx, scene, tm, tr)
    loss.sum().backward()
    torch.cuda.synchronize()
    fused.check_status(ctx)
    cmp("mask", mask.cpu().numpy(), m_ref, 1e-6)
    print("  loss", loss.detach().cpu().numpy(), l_ref)
    cmp("grad mvp", tm.grad.cpu().numpy(), g_ref, 1e-1)
    print("  rel grad err", np.abs(tm.grad.cpu().numpy() - g_ref).max() / np.abs(g_ref).max())
    for _ in range(3):
        mask, loss = fused.render_mask_loss(ctx, scene, tm, tr)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        mask, loss = fused.render_mask_loss(ctx, scene, tm, tr)
    torch.cuda.synchronize()
    print("  fused ms/step", (time.time() - t0) / 20 * 1e3)
