"""Synthetic workloads for the BASELINE.json configs: joint angles, link poses, camera pose and
intrinsics.  There is no network / dataset here, so inputs are generated deterministically (seed 0) from the
packaged robot geometry; reference masks are rendered by the HIP renderer itself at the ground-truth pose."""
import numpy as np

from .config import FRANKA_K_1920x1080, XARM7_K_1280x720

__all__ = ["lookat_pose", "camera_Tc_c2b", "scaled_K", "perturb_pose", "make_views", "WORKLOADS"]


def lookat_pose(phi, theta, radius):
    """Camera-to-base pose looking at the origin (restates /root/reference/easyhec/utils/utils_3d.py:359-394,
    ``calc_pose_from_lookat`` for one view): phi from +z, theta azimuth, OpenCV camera axes."""
    c = np.array([radius * np.sin(theta) * np.sin(phi), -radius * np.cos(theta) * np.sin(phi), radius * np.cos(phi)])
    fwd = c / (np.linalg.norm(c) + 1e-10)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(up, fwd)
    right = right / (np.linalg.norm(right) + 1e-10)
    if (right ** 2).sum() < 0.01:
        right = np.array([0.0, 1.0, 0.0])
    up = np.cross(fwd, right)
    up = up / (np.linalg.norm(up) + 1e-10)
    pose = np.eye(4)
    pose[:3, :3] = np.stack([right, up, fwd], axis=-1)
    pose[:3, 3] = c
    return pose @ np.diag([1.0, -1.0, -1.0, 1.0])  # blender -> opencv


def camera_Tc_c2b(phi_deg=60.0, theta_deg=20.0, radius=1.3, lift=0.3):
    """Tc_c2b = inverse of the look-at pose with the camera raised by ``lift`` metres
    (recipe of /root/reference/tools/manual_tune_franka_init.py:20-26)."""
    Tb_b2c = lookat_pose(np.radians(phi_deg), np.radians(theta_deg), radius)
    Tb_b2c[2, 3] += lift
    return np.linalg.inv(Tb_b2c)


def scaled_K(K, scale, W=None, H=None, recentre=False):
    K = np.array(K, dtype=np.float64)
    K[:2] *= scale
    if recentre:
        K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    return K


def perturb_pose(Tc_c2b, dt=(0.02, -0.015, 0.02), drot_deg=(3.0, -2.0, 2.0)):
    """GT o exp([dt, drot]) -- the initial pose error of config 2."""
    w = np.radians(np.asarray(drot_deg, dtype=np.float64))
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * (K @ K) if th > 0 else np.eye(3)
    D = np.eye(4)
    D[:3, :3] = R
    D[:3, 3] = dt
    return Tc_c2b @ D


def make_views(robot, n_views, seed=0, qpos_scale=0.5):
    """(qpos [n,7], link_poses [n,L,4,4] float32) for n synthetic views of one camera."""
    rng = np.random.default_rng(seed)
    q = robot.sample_qpos(n_views, rng, scale=qpos_scale)
    lp = np.stack([robot.link_poses(qi) for qi in q]).astype(np.float32)
    return q, lp


# name -> (robot, H, W, K, number of views, camera radius, lift)
WORKLOADS = {
    # config 1: plumbing case, single merged mesh is handled by the tests directly (320x240)
    "xarm7_640x480_1view": dict(robot="xarm7", H=480, W=640, K=scaled_K(XARM7_K_1280x720, 0.5, 640, 480, True),
                                views=1, radius=1.3, lift=0.3),
    "xarm7_1280x720_8view": dict(robot="xarm7", H=720, W=1280, K=np.array(XARM7_K_1280x720), views=8, radius=1.3,
                                 lift=0.3),
    "franka_1920x1080_16view": dict(robot="franka", H=1080, W=1920, K=np.array(FRANKA_K_1920x1080), views=16,
                                    radius=2.6, lift=0.5),
    "xarm7_1280x720_64view": dict(robot="xarm7", H=720, W=1280, K=np.array(XARM7_K_1280x720), views=64, radius=1.3,
                                  lift=0.3),
}
