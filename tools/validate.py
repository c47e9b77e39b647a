"""Visual acceptance check, mirroring /root/reference/tools/validate.py:13-48: load the solved pose from the newest
checkpoint (``ckpt['model']['dof']``), render the robot mask for every frame of a dataset directory
(``color/*.png``, ``qpos/*.txt``, ``K.txt``) through :mod:`easyhec_amd.render_api`, and write red overlays
(easyhec/utils/plt_utils.py:163-200 ``vis_mask``: 40 % tint + 1-pixel border) as PNGs.

    python tools/validate.py --ckpt_dir models/xarm7/example --data_dir data/xarm7/example [--robot xarm7] [--out dbg/validate]
"""
import argparse
import glob
import os
import os.path as osp
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from easyhec_amd import render_api  # noqa: E402
from easyhec_amd.se3 import se3_exp_map  # noqa: E402


def vis_mask(img, mask, color=(255, 0, 0), alpha=0.4, border_alpha=0.5):
    img = img.astype(np.float32)
    m = mask.astype(bool)
    img[m] = img[m] * (1.0 - alpha) + alpha * np.asarray(color, np.float32)
    # border: mask pixels with an unmasked 4-neighbour (stands in for cv2.findContours + drawContours, thickness 1)
    pad = np.pad(m, 1)
    inner = pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
    edge = m & ~inner
    img[edge] = img[edge] * (1.0 - border_alpha) + border_alpha * np.asarray(color, np.float32)
    return img.clip(0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt_dir", default="models/xarm7/example")
    ap.add_argument("--data_dir", default="data/xarm7/example", help="data dir to validate on")
    ap.add_argument("--robot", default="xarm7", choices=["xarm7", "franka"])
    ap.add_argument("--urdf", default=None, help="URDF for the kinematics (default: the chain packaged with the robot)")
    ap.add_argument("--out", default="dbg/validate")
    a = ap.parse_args()
    ckpt_path = sorted(glob.glob(osp.join(a.ckpt_dir, "*pth")))[-1]
    print(f"using ckpt path {ckpt_path}")
    ckpt = torch.load(ckpt_path, map_location="cpu")
    dof6 = ckpt["model"]["dof"]
    Tc_c2b = se3_exp_map(dof6[None]).permute(0, 2, 1)[0].cpu().numpy()
    np.set_printoptions(suppress=True, precision=3)
    print("Tc_c2b", np.array2string(Tc_c2b, separator=","))
    rgb_paths = sorted(glob.glob(osp.join(a.data_dir, "color/*.png")))
    qpos_paths = sorted(glob.glob(osp.join(a.data_dir, "qpos/*.txt")))
    K = np.loadtxt(osp.join(a.data_dir, "K.txt"))
    os.makedirs(a.out, exist_ok=True)
    render = render_api.nvdiffrast_render_xarm_api if a.robot == "xarm7" else render_api.nvdiffrast_render_franka_api
    for i, (rp, qp) in enumerate(zip(rgb_paths, qpos_paths)):
        rgb = np.asarray(Image.open(rp).convert("RGB"))
        H, W = rgb.shape[:2]
        mask = render(a.urdf, Tc_c2b, np.loadtxt(qp), H, W, K)
        Image.fromarray(vis_mask(rgb, mask)).save(osp.join(a.out, f"rendered_mask_{i:06d}.png"))
    print(f"wrote {len(rgb_paths)} overlays to {a.out}")


if __name__ == "__main__":
    main()
