"""BASELINE configs[1] (xArm7, 640x480, ONE view, 200 Adam iterations): where do runs end?

A single silhouette leaves the pose weakly observed along some directions, and constant-LR Adam (lr 3e-3, the
reference's setting) jitters around whatever it reaches, so "the converged pose" is a cloud, not a point.  This tool
measures that cloud: several initial perturbations, HIP launch chain, mean of the last 20 iterates, and optionally the
same run driven by the CPU oracle (slow).  Its numbers are quoted in BASELINE.md / DESIGN.md.

    python tools/config2_study.py [--oracle] [--views 1]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PERTURBATIONS = [((0.02, -0.015, 0.02), (3.0, -2.0, 2.0)),      # the one SURVEY 8d names
                 ((-0.015, 0.02, -0.01), (-2.0, 3.0, -1.5)),
                 ((0.01, 0.01, -0.025), (1.5, 2.5, 3.0))]


def pose_error(Ta, Tb):
    dt = np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]) * 1000.0
    R = Ta[:3, :3].T @ Tb[:3, :3]
    return dt, np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    from easyhec_amd import fused
    from easyhec_amd.config import XARM7_K_1280x720, Cfg
    from easyhec_amd.rb_solver import RBSolver
    from easyhec_amd.robot import load_robot
    from easyhec_amd.se3 import se3_exp_map
    from easyhec_amd.synthetic import camera_Tc_c2b, make_views, perturb_pose, scaled_K
    from easyhec_amd.trainer import RBSolverTrainer
    dev = torch.device("cuda:0")
    rb = load_robot("xarm7")
    H, W, B = 480, 640, args.views
    K = scaled_K(XARM7_K_1280x720, 0.5, W, H, True)
    _, lp = make_views(rb, B, seed=0)
    Tc = camera_Tc_c2b()
    Kt = torch.tensor(K, dtype=torch.float32, device=dev)
    lpt = torch.tensor(lp, device=dev)
    ends = []
    for dt, dr in PERTURBATIONS:
        init = perturb_pose(Tc, dt=dt, drot_deg=dr)
        cfg = Cfg()
        cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W
        cfg.model.rbsolver.init_Tc_c2b = init.tolist()
        model = RBSolver(cfg, meshes=rb.meshes).to(dev)
        with torch.no_grad():
            gt, _ = fused.render_mask_loss(model._ensure_renderer().glctx, model._ensure_scene(), fused.mvp_matrices(
                Kt, H, W, torch.tensor(Tc, dtype=torch.float32, device=dev), lpt), torch.zeros((B, H, W), device=dev))
        ref = (gt > 0.5).float()
        batch = {"mask": ref, "link_poses": lpt, "K": Kt[None].repeat(B, 1, 1)}
        tr = RBSolverTrainer(cfg, model, batch, fast=True)
        dofs, losses = [], []
        for it in range(args.iters):
            losses.append(float(tr.step()[1]))
            dofs.append(model.dof.detach().clone())
        mean_dof = torch.stack(dofs[-20:]).mean(0).cpu()
        Tm = se3_exp_map(mean_dof[None]).permute(0, 2, 1)[0].numpy().astype(np.float64)
        e0, e1 = pose_error(init, Tc), pose_error(Tm, Tc)
        l0, l1 = losses[0], float(np.mean(losses[-20:]))
        jit = torch.stack(dofs[-20:]).std(0).cpu().numpy()
        print(f"init err {e0[0]:6.2f} mm {e0[1]:5.2f} deg -> mean of last 20: {e1[0]:6.2f} mm {e1[1]:5.3f} deg | loss {l0:9.1f} -> {l1:8.1f} "
              f"| jitter (std of dof) trans {jit[:3].max() * 1e3:.2f} mm rot {np.degrees(jit[3:].max()):.3f} deg", flush=True)
        ends.append(Tm)
        if args.oracle:
            from oracle_backend import OracleRBSolver
            cpu = OracleRBSolver(rb, init, H, W)
            cb = {"mask": ref.cpu(), "link_poses": torch.tensor(lp), "K": torch.tensor(K, dtype=torch.float32)[None].repeat(B, 1, 1)}
            ctr = RBSolverTrainer(cfg, cpu, cb)
            cd = []
            for it in range(args.iters):
                ctr.step()
                cd.append(cpu.dof.detach().clone())
            Tcm = se3_exp_map(torch.stack(cd[-20:]).mean(0)[None]).permute(0, 2, 1)[0].numpy().astype(np.float64)
            d = pose_error(Tm, Tcm)
            eo = pose_error(Tcm, Tc)
            print(f"   oracle-driven run ends {eo[0]:6.2f} mm {eo[1]:5.3f} deg from the truth; HIP vs oracle: {d[0]:.2f} mm {d[1]:.3f} deg", flush=True)
    for i in range(len(ends)):
        for j in range(i + 1, len(ends)):
            d = pose_error(ends[i], ends[j])
            print(f"runs {i} / {j} end {d[0]:.2f} mm {d[1]:.3f} deg apart")


if __name__ == "__main__":
    main()
