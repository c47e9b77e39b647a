"""Robustness soak (run on the GPU box): long solver runs eager and as a replayed hipGraph, re-planning across shapes,
scorer calls with changing batch shapes (scratch growth paths), interleaved contexts.  Prints one line per phase;
any HIP error, NaN or status failure raises.  python tools/soak.py [--scale 1.0]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from easyhec_amd import dr, fused, space_explorer as se  # noqa: E402
from easyhec_amd.config import XARM7_K_1280x720, Cfg  # noqa: E402
from easyhec_amd.rb_solver import RBSolver  # noqa: E402
from easyhec_amd.robot import load_robot  # noqa: E402
from easyhec_amd.synthetic import camera_Tc_c2b, make_views, perturb_pose, scaled_K  # noqa: E402
from easyhec_amd.trainer import RBSolverTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="multiplies the iteration counts")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rb = load_robot("xarm7")
    Tc = camera_Tc_c2b()
    t_all = time.time()

    def problem(B, H, W, s):
        K = scaled_K(XARM7_K_1280x720, s, W, H, s != 1.0)
        _, lp = make_views(rb, B, seed=B)
        cfg = Cfg()
        cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W
        cfg.model.rbsolver.init_Tc_c2b = perturb_pose(Tc).tolist()
        model = RBSolver(cfg, meshes=rb.meshes).to(dev)
        ctx, scene = model._ensure_renderer().glctx, model._ensure_scene()
        mvp = torch.tensor(helpers.mvp_numpy(K, H, W, Tc, lp), device=dev)
        ref = (fused.render_mask_loss(ctx, scene, mvp, torch.zeros((B, H, W), device=dev))[0] > 0.5).float()
        batch = {"mask": ref, "link_poses": torch.tensor(lp, device=dev),
                 "K": torch.tensor(K, dtype=torch.float32, device=dev)[None].repeat(B, 1, 1)}
        return cfg, model, batch

    # 1. long eager run and long graph-replay run on two contexts, interleaved
    cfg, m1, batch = problem(8, 720, 1280, 1.0)
    _, m2, _ = problem(8, 720, 1280, 1.0)
    t1, t2 = RBSolverTrainer(cfg, m1, batch, fast=True), RBSolverTrainer(cfg, m2, batch, fast=True, graph=True)
    n = int(6000 * a.scale)
    t0 = time.time()
    for i in range(n):
        t1.step()
        t2.step()
        if i % 1000 == 999:
            torch.cuda.synchronize()
            assert torch.equal(m1.dof.data, m2.dof.data), f"eager and graph runs diverged at {i}"
            fused.check_status(t1.fast.glctx)
            fused.check_status(t2.fast.glctx)
    torch.cuda.synchronize()
    print(f"solver: 2 x {n} steps (eager + graph replay, bit-identical) in {time.time() - t0:.1f} s, "
          f"loss {float(t1.last_loss):.2f}", flush=True)

    # 2. re-planning across shapes on ONE context
    ctx = dr.RasterizeCudaContext()
    scene = fused.LinkScene([v for v, _ in rb.meshes], [f for _, f in rb.meshes], dev)
    rng = np.random.default_rng(0)
    t0 = time.time()
    for it in range(int(40 * a.scale)):
        B = int(rng.integers(1, 12))
        H, W = int(rng.integers(40, 900)), int(rng.integers(40, 1400))
        s = W / 1280.0
        K = scaled_K(XARM7_K_1280x720, s, W, H, True)
        _, lp = make_views(rb, B, seed=it)
        mvp = torch.tensor(helpers.mvp_numpy(K, H, W, perturb_pose(Tc), lp), device=dev, requires_grad=True)
        ref = torch.zeros((B, H, W), device=dev)
        mask, loss = fused.render_mask_loss(ctx, scene, mvp, ref)
        loss.sum().backward()
        torch.cuda.synchronize()
        fused.check_status(ctx)
        assert torch.isfinite(loss).all() and torch.isfinite(mvp.grad).all()
        assert abs(float(loss.detach().sum()) - float((mask.detach().double() ** 2).sum())) <= 1e-5 * max(1.0, float(loss.detach().sum()))
    print(f"re-plan: {int(40 * a.scale)} random shapes in {time.time() - t0:.1f} s", flush=True)

    # 3. scorer with changing batch shapes and chunk sizes
    t0 = time.time()
    for it in range(int(30 * a.scale)):
        Q, S = int(rng.integers(1, 40)), int(rng.integers(1, 12))
        H, W = int(rng.integers(60, 500)), int(rng.integers(60, 700))
        s = W / 1280.0
        K = scaled_K(XARM7_K_1280x720, s, W, H, True)
        _, lp = make_views(rb, Q, seed=100 + it)
        mv = np.stack([helpers.mvp_numpy(K, H, W, perturb_pose(Tc, dt=rng.normal(0, 0.02, 3), drot_deg=rng.normal(0, 2, 3)), lp)
                       for _ in range(S)], axis=1)
        var, score, counts = se.mask_variance(ctx, scene, torch.tensor(mv, device=dev), H, W, return_counts=True,
                                              chunk_views=int(rng.integers(1, 200)))
        c = counts.long()
        assert (score == (c * (S - c)).sum(dim=(1, 2))).all() and int(c.max()) <= S
    print(f"scorer: {int(30 * a.scale)} random shapes in {time.time() - t0:.1f} s", flush=True)
    print(f"soak ok in {time.time() - t_all:.1f} s")


if __name__ == "__main__":
    main()
