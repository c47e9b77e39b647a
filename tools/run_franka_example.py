"""Runs the reference's offline Franka example (assets/franka_offline_example.zip, re-packed in tests/golden/) end to end:
dataset directory -> XarmRealDataset -> RBSolver -> 1000 Adam iterations (configs/franka/example_franka_offline.yaml)."""
import os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easyhec_amd.config import Cfg
from easyhec_amd.data import XarmRealDataset, collate_all
from easyhec_amd.rb_solver import RBSolver
from easyhec_amd.robot import load_robot
from easyhec_amd.trainer import RBSolverTrainer


def write_example_dir(dst):
    from PIL import Image
    z = np.load(os.path.join(ROOT, "tests", "golden", "franka_offline_example.npz"))
    shape = tuple(z["shape"])
    masks = np.unpackbits(z["masks"])[:int(np.prod(shape))].reshape(shape).astype(bool)
    os.makedirs(os.path.join(dst, "mask")); os.makedirs(os.path.join(dst, "qpos"))
    for i in range(shape[0]):
        Image.fromarray((masks[i] * 255).astype(np.uint8)).save(os.path.join(dst, "mask", f"{i:06d}.png"))
        np.savetxt(os.path.join(dst, "qpos", f"{i:06d}.txt"), z["qpos"][i])
    np.savetxt(os.path.join(dst, "K.txt"), z["K"])
    return z["init_Tc_c2b"], masks


def iou(a, b):
    return (a & b).sum() / max(1, (a | b).sum())


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    with tempfile.TemporaryDirectory() as d:
        init, masks = write_example_dir(d)
        robot = load_robot("franka")
        ds = XarmRealDataset(d, robot)
        batch = collate_all(ds, dev)
    cfg = Cfg()
    cfg.model.rbsolver.H, cfg.model.rbsolver.W = 480, 640
    cfg.model.rbsolver.init_Tc_c2b = init.tolist()
    model = RBSolver(cfg, meshes=robot.meshes).to(dev)
    tr = RBSolverTrainer(cfg, model, batch, fast=True)
    with torch.no_grad():
        out0, ld0 = model(batch)
    m0 = out0["rendered_masks"].cpu().numpy() > 0.5
    print("init  loss", float(ld0["mask_loss"]), "IoU", np.mean([iou(m0[i], masks[i]) for i in range(len(masks))]))
    t0 = time.time()
    for it in range(iters):
        _, l = tr.step()
        if it % 100 == 0:
            print(it, float(l), flush=True)
    torch.cuda.synchronize(); print("time", time.time() - t0)
    with torch.no_grad():
        out1, ld1 = model(batch)
    m1 = out1["rendered_masks"].cpu().numpy() > 0.5
    print("final loss", float(ld1["mask_loss"]), "IoU", np.mean([iou(m1[i], masks[i]) for i in range(len(masks))]))
    print("Tc_c2b\n", out1["tsfm"].numpy())
