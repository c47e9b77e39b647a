import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from easyhec_amd.config import Cfg
from easyhec_amd.rb_solver import RBSolver
from easyhec_amd.robot import load_robot
from easyhec_amd.trainer import RBSolverTrainer
dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "franka_offline_example.npz"))
shape = tuple(z["shape"]); masks = np.unpackbits(z["masks"])[:int(np.prod(shape))].reshape(shape).astype(bool)
robot = load_robot("franka")
lp = np.stack([robot.link_poses(q) for q in z["qpos"]]).astype(np.float32)
def iou(a, b): return (a & b).sum() / max(1, (a | b).sum())
def run(scale, iters, meshes, lr=3e-3, tag=""):
    H, W = 480 * scale, 640 * scale
    K = z["K"].copy(); K[:2] *= scale
    m = np.repeat(np.repeat(masks, scale, axis=1), scale, axis=2)
    cfg = Cfg(); cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W; cfg.solver.max_lr = lr
    cfg.model.rbsolver.init_Tc_c2b = z["init_Tc_c2b"].tolist()
    model = RBSolver(cfg, meshes=meshes).to(dev)
    batch = {"mask": torch.tensor(m, dtype=torch.float32, device=dev), "link_poses": torch.tensor(lp, device=dev),
             "K": torch.tensor(K, dtype=torch.float32, device=dev)[None].repeat(10, 1, 1)}
    tr = RBSolverTrainer(cfg, model, batch, fast=True)
    t0 = time.time(); ls = []
    for it in range(iters):
        _, l = tr.step()
        if it % (iters // 5) == 0: ls.append(round(float(l)))
    with torch.no_grad():
        out, ld = model(batch)
    r = out["rendered_masks"].cpu().numpy() > 0.5
    print(tag, "scale", scale, "its", iters, "losses", ls, "final", round(float(ld["mask_loss"])), "IoU", round(float(np.mean([iou(r[i], m[i]) for i in range(10)])), 3), "time", round(time.time() - t0, 2), flush=True)
    return out["tsfm"].numpy()
boxes = [helpers.box_mesh(v) for v, _ in robot.meshes]
run(1, 2000, robot.meshes, tag="full")
run(2, 2000, robot.meshes, tag="full")
run(4, 2000, robot.meshes, tag="full")
run(1, 2000, boxes, tag="boxes")
