"""SE(3) exponential / logarithm used to parameterise ``Tc_c2b`` by 6 numbers.

Restates /root/reference/easyhec/utils/pytorch3d_se3.py:12-41 (_so3_exp_map), :46-130 (se3_exp_map),
:218-258 (_se3_V_matrix, _get_se3_V_input) and /root/reference/easyhec/utils/utils_3d.py:303-335
(se3_exp_map wrapper, se3_log_map with backend='opencv') without pytorch3d / OpenCV.

Convention (PyTorch3D): a transform is stored TRANSPOSED, ``[[R^T, 0], [t, 1]]`` -- the reference re-transposes after
every call (rb_solver.py:31-34, :52) and so do the callers here.  dof = [log_translation(3), log_rotation(3)].
"""
import numpy as np
import torch

__all__ = ["hat", "se3_exp_map", "se3_log_map", "so3_log_numpy"]


def hat(v):
    """[N,3] -> [N,3,3] skew-symmetric matrices (pytorch3d.transforms.so3.hat)."""
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    o = torch.zeros_like(x)
    return torch.stack([torch.stack([o, -z, y], dim=1), torch.stack([z, o, -x], dim=1),
                        torch.stack([-y, x, o], dim=1)], dim=1)


def _so3_exp_map(log_rot, eps=1e-4):
    nrms = (log_rot * log_rot).sum(1)
    rot_angles = torch.clamp(nrms, eps).sqrt()  # squared-angle clamp, pytorch3d_se3.py:26
    rot_angles_inv = 1.0 / rot_angles
    fac1 = rot_angles_inv * rot_angles.sin()
    fac2 = rot_angles_inv * rot_angles_inv * (1.0 - rot_angles.cos())
    skews = hat(log_rot)
    skews_square = torch.bmm(skews, skews)
    R = fac1[:, None, None] * skews + fac2[:, None, None] * skews_square + \
        torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None]
    return R, rot_angles, skews, skews_square


def _se3_V_matrix(log_rotation, log_rotation_hat, log_rotation_hat_square, rotation_angles, eps=1e-4):
    return (torch.eye(3, dtype=log_rotation.dtype, device=log_rotation.device)[None]
            + log_rotation_hat * ((1 - torch.cos(rotation_angles)) / (rotation_angles ** 2))[:, None, None]
            + log_rotation_hat_square *
            ((rotation_angles - torch.sin(rotation_angles)) / (rotation_angles ** 3))[:, None, None])


def se3_exp_map(log_transform, eps=1e-4):
    """[N,6] -> [N,4,4] (transposed convention), pytorch3d_se3.py:46-130."""
    if log_transform.ndim != 2 or log_transform.shape[1] != 6:
        raise ValueError("Expected input to be of shape (N, 6).")
    N = log_transform.shape[0]
    log_translation = log_transform[..., :3]
    log_rotation = log_transform[..., 3:]
    R, rotation_angles, log_rotation_hat, log_rotation_hat_square = _so3_exp_map(log_rotation, eps=eps)
    V = _se3_V_matrix(log_rotation, log_rotation_hat, log_rotation_hat_square, rotation_angles, eps=eps)
    T = torch.bmm(V, log_translation[:, :, None])[:, :, 0]
    top = torch.cat([R, T[:, :, None]], dim=2)
    # (generated on the device: a host list would be a pageable host-to-device copy, which a stream capture rejects)
    bottom = torch.eye(4, dtype=log_transform.dtype, device=log_transform.device)[3]
    transform = torch.cat([top, bottom[None, None, :].expand(N, 1, 4)], dim=1)
    return transform.permute(0, 2, 1)


def so3_log_numpy(R):
    """Rotation vector of a 3x3 rotation matrix in float64 (what ``cv2.Rodrigues(R)[0]`` returns)."""
    R = np.asarray(R, dtype=np.float64)
    u, _, vt = np.linalg.svd(R)  # OpenCV also projects onto SO(3) first
    R = u @ vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r * r).sum() * 0.25)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        # theta ~ pi: R + I = 2 a a^T, take the best-conditioned column; sign is free at exactly pi
        M = R + np.eye(3)
        k = int(np.argmax(np.diag(M)))
        a = M[:, k] / np.linalg.norm(M[:, k])
        if np.dot(a, r) < 0:
            a = -a
        return a * theta
    return r * (0.5 * theta / s)


def se3_log_map(transform, eps=1e-4, cos_bound=1e-4, backend="opencv", test_acc=True):
    """[N,4,4] (transposed convention) -> [N,6]; utils_3d.py:308-335 with backend='opencv'.

    The rotation part is ``-Rodrigues(transform[:3,:3])`` evaluated on the host in float64 exactly like the
    reference's OpenCV call (utils_3d.py:322); ``test_acc`` re-exponentiates and raises on error > 0.1 (:331-334)."""
    del cos_bound
    if backend not in ("opencv", None):
        raise NotImplementedError(f"se3_log_map backend {backend!r}")
    log_rotation = []
    for tsfm in transform:
        rv = -so3_log_numpy(tsfm[:3, :3].detach().cpu().numpy())
        log_rotation.append(torch.from_numpy(rv.reshape(-1)).to(transform.device).float())
    log_rotation = torch.stack(log_rotation, dim=0)
    T = transform[:, 3, :3]
    nrms = (log_rotation ** 2).sum(-1)
    rotation_angles = torch.clamp(nrms, eps).sqrt()
    lh = hat(log_rotation)
    V = _se3_V_matrix(log_rotation, lh, torch.bmm(lh, lh), rotation_angles, eps=eps)
    log_translation = torch.linalg.solve(V, T[:, :, None])[:, :, 0]
    dof6 = torch.cat((log_translation, log_rotation), dim=1)
    if test_acc:
        err = (se3_exp_map(dof6) - transform).abs().max()
        if err > 0.1:
            raise RuntimeError("se3_log_map: exp(log(T)) differs from T by more than 0.1")
    return dof6
