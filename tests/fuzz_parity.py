"""Randomised parity sweep (test infrastructure, run by hand on the GPU box; not collected by pytest): the fused op, the scoring op and
the drop-in ops (both forms of dr.rasterize, dr.antialias forward) against the CPU oracle on random resolutions, zooms,
camera distances (down to a few centimetres from the robot) and joint configurations -- masks, counts, rasterizer
outputs and antialiased images bit-exact, losses / gradients within the suite's tolerances.  python tests/fuzz_parity.py [--cases 120] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import helpers  # noqa: E402
from easyhec_amd import dr, fused, space_explorer as se  # noqa: E402
from easyhec_amd.config import XARM7_K_1280x720  # noqa: E402
from easyhec_amd.robot import load_robot  # noqa: E402
from easyhec_amd.synthetic import camera_Tc_c2b, make_views, perturb_pose, scaled_K  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    robots = {n: load_robot(n) for n in ("xarm7", "franka")}
    scenes = {n: fused.LinkScene([v for v, _ in r.meshes], [f for _, f in r.meshes], dev) for n, r in robots.items()}
    ctx = dr.RasterizeCudaContext()
    t0 = time.time()
    worst = dict(loss=0.0, grad=0.0)
    nops = 0
    for case in range(a.cases):
        name = "xarm7" if rng.uniform() < 0.75 else "franka"
        rb, scene = robots[name], scenes[name]
        W = int(rng.integers(48, 420))
        H = int(rng.integers(40, 300))
        B = int(rng.integers(1, 4))
        scale = float(rng.uniform(0.04, 0.35)) * (12.0 if rng.uniform() < 0.1 else 1.0)
        radius = float(rng.choice([0.1, 0.2, 0.45, 0.9, 1.3, 2.5]))
        K = scaled_K(XARM7_K_1280x720, scale, W, H, True)
        _, lp = make_views(rb, B, seed=int(rng.integers(1 << 30)), qpos_scale=float(rng.uniform(0.2, 1.0)))
        Tc = perturb_pose(camera_Tc_c2b(radius=radius, lift=float(rng.uniform(0.0, 0.4))), dt=rng.normal(0, 0.03, 3),
                          drot_deg=rng.normal(0, 4.0, 3))
        mvp = helpers.mvp_numpy(K, H, W, Tc, lp)
        ref = (rng.uniform(size=(B, H, W)) > rng.uniform(0.2, 0.9)).astype(np.float32)
        verts, tris, toff, voff = helpers.scene_arrays(rb)
        m_ref, l_ref, g_ref = oracle.render_mask_loss(verts, tris, toff, voff, mvp, ref)
        tm = torch.tensor(mvp, device=dev, requires_grad=True)
        mask, loss = fused.render_mask_loss(ctx, scene, tm, torch.tensor(ref, device=dev))
        loss.sum().backward()
        torch.cuda.synchronize()
        fused.check_status(ctx)
        mask, loss, grad = mask.cpu().numpy(), loss.detach().cpu().numpy(), tm.grad.cpu().numpy()
        tag = f"case {case}: {name} {W}x{H} B={B} scale={scale:.3f} radius={radius}"
        assert (mask == m_ref).all(), tag + f": {(mask != m_ref).sum()} mask pixels differ"
        el = np.abs(loss - l_ref).max() / max(np.abs(l_ref).max(), 1e-30)
        eg = np.abs(grad - g_ref).max() / max(np.abs(g_ref).max(), 1e-30)
        assert el <= 1e-6 and eg <= 1e-5, tag + f": loss {el:.2e} grad {eg:.2e}"
        worst["loss"], worst["grad"] = max(worst["loss"], el), max(worst["grad"], eg)
        if case % 3 == 0:   # the scoring op on the same poses: B candidates x 1..3 camera poses
            S = int(rng.integers(1, 4))
            mv = np.stack([helpers.mvp_numpy(K, H, W, perturb_pose(Tc, dt=rng.normal(0, 0.02, 3), drot_deg=rng.normal(0, 2, 3)), lp)
                           for _ in range(S)], axis=1).astype(np.float32)
            vl = np.concatenate([np.full(v.shape[0], l, np.int32) for l, (v, _) in enumerate(rb.meshes)])
            s_ref, c_ref = oracle.mask_variance(verts, tris, vl, mv, H, W, return_counts=True)
            _, sc, cn = se.mask_variance(ctx, scene, torch.tensor(mv, device=dev), H, W, return_counts=True)
            assert (cn.cpu().numpy() == c_ref).all() and (sc.cpu().numpy() == s_ref).all(), tag + ": scoring op differs"
        if case % 4 == 1:   # the drop-in ops on one link of the same view: both forms of dr.rasterize, dr.antialias forward
            l = int(rng.integers(len(rb.meshes)))
            v, f = rb.meshes[l]
            pos = oracle.transform_pos(mvp[0, l], v)                       # [1, V, 4]
            r_ref, db_ref = oracle.rasterize(pos, f, [H, W])
            tp, tf = torch.tensor(pos, device=dev), torch.tensor(f, device=dev)
            old = os.environ.get("EHR_RASTER_DIRECT_MAX")
            for form in ("1000000000", "0"):
                os.environ["EHR_RASTER_DIRECT_MAX"] = form
                r, db = dr.rasterize(ctx, tp, tf, [H, W])
                assert (r.cpu().numpy() == r_ref).all() and (db.detach().cpu().numpy() == db_ref).all(), tag + f": dr.rasterize form {form} link {l}"
            if old is None:
                os.environ.pop("EHR_RASTER_DIRECT_MAX", None)
            else:
                os.environ["EHR_RASTER_DIRECT_MAX"] = old
            col = (r_ref[..., 3:4] > 0).astype(np.float32)
            aa = dr.antialias(torch.tensor(col, device=dev), r, tp, tf)
            assert (aa.cpu().numpy() == oracle.antialias(col, r_ref, pos, f)).all(), tag + f": dr.antialias link {l}"
            nops += 1
    print(f"fuzz ok: {a.cases} cases in {time.time() - t0:.1f} s, worst loss rel {worst['loss']:.1e}, grad rel {worst['grad']:.1e}; "
          f"{nops} of them also through the drop-in ops (both rasterizer forms, antialias forward: bit-equal)")


if __name__ == "__main__":
    main()
