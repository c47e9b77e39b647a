"""Projection / clip-space helpers, mirroring /root/reference/easyhec/utils/nvdiffrast_utils.py:5-18.

Same names, arguments and results; unlike the reference they stay on the input's device and never force a
device->host sync (the reference's ``torch.tensor([... cuda scalars ...])`` costs four per call)."""
import numpy as np
import torch

__all__ = ["K_to_projection", "transform_pos", "opencv2blender"]


def K_to_projection(K, H, W, n=0.001, f=10.0):
    """OpenGL projection of a pinhole camera (nvdiffrast_utils.py:5-11): pixel centre at u = ix + 0.5."""
    K = torch.as_tensor(K)
    dev = K.device
    K = K.to(torch.float32)
    fu, fv, cu, cv = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    zero = torch.zeros((), dtype=torch.float32, device=dev)
    one = torch.ones((), dtype=torch.float32, device=dev)
    rows = [
        torch.stack([2 * fu / W, zero, -2 * cu / W + 1, zero]),
        torch.stack([zero, 2 * fv / H, 2 * cv / H - 1, zero]),
        torch.stack([zero, zero, one * (-(f + n) / (f - n)), one * (-2 * f * n / (f - n))]),
        torch.stack([zero, zero, -one, zero]),
    ]
    return torch.stack(rows)


def transform_pos(mtx, pos):
    """(x,y,z) -> (x,y,z,1) @ mtx^T, shape [1,V,4] (nvdiffrast_utils.py:14-18)."""
    t_mtx = torch.from_numpy(mtx).to(pos.device) if isinstance(mtx, np.ndarray) else mtx
    posw = torch.cat([pos, torch.ones([pos.shape[0], 1], dtype=pos.dtype, device=pos.device)], dim=1)
    return torch.matmul(posw, t_mtx.t())[None, ...]


def opencv2blender(device=None, dtype=torch.float32):
    """diag(1,-1,-1,1): OpenCV camera (x right, y down, z forward) -> GL camera
    (nvdiffrast_renderer.py:18-22; the matrix is its own inverse)."""
    d = torch.ones(4, dtype=dtype, device=device)  # (built on the device: a host list is a pageable copy, which a stream
    d[1:3] = -1.0                                  #  capture rejects)
    return torch.diag(d)
