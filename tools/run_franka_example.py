"""Runs the reference's offline Franka example (assets/franka_offline_example.zip, re-packed in tests/golden/) end to end:
dataset directory -> XarmRealDataset -> RBSolver -> 1000 Adam iterations (configs/franka/example_franka_offline.yaml),
then puts the RESIDUAL on the table: per-frame IoU / loss of the reached optimum and, with --overlay DIR, one PNG per
frame (red = shipped mask only, green = rendered only, yellow = both) -- the optimum reached from the documented init
pose fits the base and mis-fits the distal links in the long-reach frames (mean IoU 0.69).

    python tools/run_franka_example.py [iterations] [--overlay gpurun_out/franka_overlay]"""
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
    argv = [a for a in sys.argv[1:]]
    overlay = None
    if "--overlay" in argv:
        overlay = argv[argv.index("--overlay") + 1]
        del argv[argv.index("--overlay"):argv.index("--overlay") + 2]
    iters = int(argv[0]) if argv else 1000
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
    soft = out1["rendered_masks"].cpu().numpy()
    print("frame  IoU   loss(SSE)  fg(mask) fg(render)  centroid shift px (render - mask)")
    for i in range(len(masks)):
        ys, xs = np.nonzero(masks[i])
        yr, xr = np.nonzero(m1[i])
        shift = (xr.mean() - xs.mean(), yr.mean() - ys.mean()) if len(xs) and len(xr) else (float("nan"),) * 2
        print(f"{i:5d} {iou(m1[i], masks[i]):5.3f} {((soft[i] - masks[i]) ** 2).sum():10.1f} {masks[i].mean():8.3f} {m1[i].mean():9.3f}"
              f"   ({shift[0]:+6.1f}, {shift[1]:+6.1f})")
    if overlay:
        from PIL import Image
        os.makedirs(overlay, exist_ok=True)
        for i in range(len(masks)):
            rgb = np.zeros(masks[i].shape + (3,), np.uint8)
            rgb[..., 0] = masks[i] * 255
            rgb[..., 1] = m1[i] * 255
            Image.fromarray(rgb).save(os.path.join(overlay, f"frame_{i:02d}.png"))
        print("overlays written to", overlay)
