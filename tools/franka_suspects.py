"""The reference's Franka example ends at IoU 0.69 with the base aligned and the distal links off.
This script tries the obvious MODEL-side suspects once each and prints loss / IoU of the optimum reached from the
documented initial pose (configs/franka/example_franka_offline.yaml:5-8), 1000 Adam iterations each:

  * a mesh-frame convention error of ONE link (Collada up-axis / node-matrix handling): the link's mesh rotated by
    +-90 deg about x, y or z, or by 180 deg about z, in its own frame;
  * a constant offset of ONE joint reading (+-2 deg, +-5 deg);
  * the hand mesh flipped (180 deg about z) -- the hand is the link whose Collada file has the most involved node tree.

Run on the GPU box:  python tools/franka_suspects.py  > gpurun_out/franka_suspects.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easyhec_amd.config import Cfg  # noqa: E402
from easyhec_amd.rb_solver import RBSolver  # noqa: E402
from easyhec_amd.robot import load_robot  # noqa: E402
from easyhec_amd.trainer import RBSolverTrainer  # noqa: E402

dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "franka_offline_example.npz"))
shape = tuple(z["shape"])
masks = np.unpackbits(z["masks"])[:int(np.prod(shape))].reshape(shape).astype(bool)
robot = load_robot("franka")
qpos = np.asarray(z["qpos"], dtype=np.float64)
H, W = 480, 640


def iou(a, b):
    return (a & b).sum() / max(1, (a | b).sum())


def rot(axis, deg):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    R = np.eye(4)
    i, j = [(1, 2), (2, 0), (0, 1)]["xyz".index(axis)]
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
    return R


def solve(lp, iters=1000):
    cfg = Cfg()
    cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W
    cfg.model.rbsolver.init_Tc_c2b = z["init_Tc_c2b"].tolist()
    model = RBSolver(cfg, meshes=robot.meshes).to(dev)
    batch = {"mask": torch.tensor(masks, dtype=torch.float32, device=dev),
             "link_poses": torch.tensor(lp.astype(np.float32), device=dev),
             "K": torch.tensor(z["K"], dtype=torch.float32, device=dev)[None].repeat(shape[0], 1, 1)}
    tr = RBSolverTrainer(cfg, model, batch, fast=True)
    for _ in range(iters):
        tr.step()
    with torch.no_grad():
        out, ld = model(batch)
    r = out["rendered_masks"].cpu().numpy() > 0.5
    return float(ld["mask_loss"]), float(np.mean([iou(r[i], masks[i]) for i in range(shape[0])]))


def main():
    lp0 = np.stack([robot.link_poses(q) for q in qpos])
    base = solve(lp0)
    print("baseline                          loss %8.0f  IoU %.3f" % base)
    rows = []
    L = lp0.shape[1]
    for l in range(L):
        for axis, deg in (("x", 90), ("x", -90), ("y", 90), ("y", -90), ("z", 90), ("z", -90), ("z", 180)):
            lp = lp0.copy()
            lp[:, l] = lp[:, l] @ rot(axis, deg)
            rows.append(("link %d mesh R%s(%+d)" % (l, axis, deg),) + solve(lp))
    nj = min(7, qpos.shape[1])
    for j in range(nj):
        for d in (-5, -2, 2, 5):
            q = qpos.copy()
            q[:, j] += np.deg2rad(d)
            rows.append(("joint %d offset %+d deg" % (j, d),) + solve(np.stack([robot.link_poses(x) for x in q])))
    rows.sort(key=lambda r: r[1])
    print("variants sorted by final loss (baseline %.0f):" % base[0])
    for name, loss, i in rows[:12]:
        print("  %-30s loss %8.0f  IoU %.3f" % (name, loss, i))
    print("  ...")
    for name, loss, i in rows[-3:]:
        print("  %-30s loss %8.0f  IoU %.3f" % (name, loss, i))
    better = [r for r in rows if r[1] < 0.9 * base[0]]
    print("variants with a loss more than 10 %% below the baseline: %d" % len(better))


if __name__ == "__main__":
    main()
