"""Dataset directory format of the reference -- the step immediately before the hot path.

Mirrors /root/reference/easyhec/data/datasets/xarm_real.py:16-84: a directory with ``color/*.png`` (optional here),
``mask/*.png`` (foreground = any non-zero value, read like ``cv2.imread(path, 2) > 0``), ``qpos/*.txt`` (one joint
vector per frame, zero-padded to the articulation's dof), ``K.txt`` (3x3) and optionally ``Tc_c2b.txt`` (4x4, else
identity).  Link poses come from the URDF chain (:mod:`easyhec_amd.kinematics`) instead of SAPIEN.
``__getitem__`` returns the same keys as the reference (``rgb`` only if colour images are present and requested)."""
import glob
import os

import numpy as np
import torch

__all__ = ["XarmRealDataset", "collate_all"]


def _read_mask(path):
    """``cv2.imread(path, 2) > 0`` (xarm_real.py:40; flag 2 = IMREAD_ANYDEPTH: one grey channel, 16-bit kept): alpha is
    dropped, palettes are resolved, colour is reduced with OpenCV's BGR2GRAY fixed-point weights (so a pixel such as
    (1, 0, 0) is background, as it is for the reference)."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ("P", "PA"):
            im = im.convert("RGBA" if "transparency" in im.info or im.mode == "PA" else "RGB")
        if im.mode in ("RGBA", "LA"):
            im = im.convert(im.mode[:-1])  # drop alpha
        a = np.asarray(im)
    if a.ndim == 3:
        a = a.astype(np.int64)
        a = (a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14  # cv::COLOR_RGB2GRAY, 8-bit
    return a > 0


class XarmRealDataset(torch.utils.data.Dataset):
    def __init__(self, data_dir, robot, ds_len=-1, load_rgb=False):
        """robot: :class:`easyhec_amd.robot.Robot` (URDF chain + ``use_links``, the reference's
        ``cfg.dataset.xarm_real.{urdf_path, use_links}``)."""
        self.data_dir = data_dir
        if ds_len < 0:
            ds_len = 1000000
        rgb_paths = sorted(glob.glob(f"{data_dir}/color/*.png"))[:ds_len]
        mask_paths = sorted(glob.glob(f"{data_dir}/mask/*.png"))[:ds_len]
        qpos_paths = sorted(glob.glob(f"{data_dir}/qpos/*.txt"))[:ds_len]
        if not mask_paths or len(mask_paths) != len(qpos_paths):
            raise FileNotFoundError(f"{data_dir}: need matching mask/*.png and qpos/*.txt files")
        self.nimgs = len(mask_paths)
        self.images = []
        if load_rgb:
            from PIL import Image
            for p in rgb_paths:
                with Image.open(p) as im:
                    self.images.append(np.asarray(im)[..., :3])
        self.masks = torch.from_numpy(np.stack([_read_mask(p) for p in mask_paths])).float()
        self.qpos = [np.loadtxt(p) for p in qpos_paths]
        self.link_poses = torch.from_numpy(
            np.stack([robot.link_poses(q) for q in self.qpos])).float()   # xarm_real.py:42-58
        self.K = torch.from_numpy(np.loadtxt(f"{data_dir}/K.txt")).float()
        tc = f"{data_dir}/Tc_c2b.txt"
        self.Tc_c2b = torch.from_numpy(np.loadtxt(tc) if os.path.exists(tc) else np.eye(4)).float()

    def __len__(self):
        return self.nimgs

    def __getitem__(self, idx):
        d = {"qpos": self.qpos[idx], "K": self.K, "link_poses": self.link_poses[idx], "Tc_c2b": self.Tc_c2b,
             "mask": self.masks[idx]}
        if self.images:
            d["rgb"] = self.images[idx]
        return d


def collate_all(ds, device=None):
    """The single batch the reference trains on (batch_size=100 >= #frames, example.yaml:45) as device tensors."""
    B = len(ds)
    batch = {"mask": ds.masks, "link_poses": ds.link_poses, "K": ds.K[None].repeat(B, 1, 1),
             "Tc_c2b": ds.Tc_c2b[None].repeat(B, 1, 1)}
    if device is not None:
        batch = {k: v.to(device) for k, v in batch.items()}
    return batch
