"""End-to-end on the GPU: BASELINE configs[1] (xArm7, 640x480, 1 view, 200 Adam iterations on Tc_c2b) converges, and
the HIP-driven optimisation tracks the oracle-driven one to <= 1 mm / 0.1 deg (north_star's pose bar)."""
import os
import sys

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def pose_error(Ta, Tb):
    dt = np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]) * 1000.0
    R = Ta[:3, :3].T @ Tb[:3, :3]
    ang = np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))
    return dt, ang


def test_config2_converges_and_matches_oracle_driven_run(xarm7, oracle):
    assert torch.cuda.is_available()
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_backend import OracleRBSolver
    from easyhec_amd import fused
    from easyhec_amd.config import XARM7_K_1280x720, Cfg
    from easyhec_amd.rb_solver import RBSolver
    from easyhec_amd.se3 import se3_exp_map
    from easyhec_amd.synthetic import camera_Tc_c2b, make_views, perturb_pose, scaled_K
    from easyhec_amd.trainer import RBSolverTrainer
    dev = torch.device("cuda:0")
    H, W, B, iters = 480, 640, 1, 200
    K = scaled_K(XARM7_K_1280x720, 0.5, W, H, True)
    _, lp = make_views(xarm7, B, seed=0)
    Tc = camera_Tc_c2b()
    cfg = Cfg()
    cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W
    cfg.model.rbsolver.init_Tc_c2b = perturb_pose(Tc).tolist()
    model = RBSolver(cfg, meshes=xarm7.meshes).to(dev)
    ren, scene = model._ensure_renderer(), model._ensure_scene()
    Kt = torch.tensor(K, dtype=torch.float32, device=dev)
    lpt = torch.tensor(lp, device=dev)
    with torch.no_grad():
        gt, _ = fused.render_mask_loss(ren.glctx, scene, fused.mvp_matrices(
            Kt, H, W, torch.tensor(Tc, dtype=torch.float32, device=dev), lpt), torch.zeros((B, H, W), device=dev))
    ref = (gt > 0.5).float()
    batch = {"mask": ref, "link_poses": lpt, "K": Kt[None], "Tc_c2b": torch.tensor(Tc, dtype=torch.float32, device=dev)[None]}
    tr = RBSolverTrainer(cfg, model, batch)
    losses = [float(tr.step()[1]) for _ in range(iters)]
    T_gpu = se3_exp_map(model.dof.detach().cpu()[None]).permute(0, 2, 1)[0].numpy().astype(np.float64)
    e0 = pose_error(perturb_pose(Tc), Tc)
    e1 = pose_error(T_gpu, Tc)
    assert losses[-1] < 0.2 * losses[0]
    assert e1[0] < 0.35 * e0[0] and e1[1] < 0.35 * e0[1], (e0, e1)
    # oracle-driven run of the same optimisation on the CPU
    cpu = OracleRBSolver(xarm7, perturb_pose(Tc), H, W)
    cb = {"mask": ref.cpu(), "link_poses": torch.tensor(lp), "K": torch.tensor(K, dtype=torch.float32)[None]}
    ctr = RBSolverTrainer(cfg, cpu, cb)
    closs = [float(ctr.step()[1]) for _ in range(iters)]
    T_cpu = se3_exp_map(cpu.dof.detach()[None]).permute(0, 2, 1)[0].numpy().astype(np.float64)
    dmm, ddeg = pose_error(T_gpu, T_cpu)
    assert dmm <= 1.0 and ddeg <= 0.1, (dmm, ddeg)
    assert abs(losses[-1] - closs[-1]) <= 0.05 * closs[-1] + 1.0
