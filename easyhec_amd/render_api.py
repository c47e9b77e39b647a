"""numpy-in / numpy-out mask rendering helpers with the names, argument order and return conventions of
/root/reference/easyhec/utils/render_api.py:16-192 (consumers: tools/validate.py:13-48, space_explorer.py:152-162).

A "mesh" is anything with ``.vertices`` [N,3] and ``.faces`` [M,3] (a trimesh.Trimesh in the reference) or a
``(vertices, faces)`` pair.  Everything renders through the HIP ops of :mod:`easyhec_amd.dr`; there is no CPU path."""
import os
from typing import List

import numpy as np
import torch

from .renderer import NVDiffrastRenderer
from .robot import load_robot
from .kinematics import UrdfChain

__all__ = ["nvdiffrast_render_mesh_api", "nvdiffrast_render_meshes_api", "nvdiffrast_parallel_render_meshes_api",
           "nvdiffrast_render_xarm_api", "nvdiffrast_render_franka_api", "nvdiffrast_parallel_render_xarm_api"]


class NVdiffrastRenderMeshApiHelper:
    """One cached renderer per image size (render_api.py:16-24)."""
    _renderer = None
    H, W = None, None

    @staticmethod
    def get_renderer(H, W):
        h = NVdiffrastRenderMeshApiHelper
        if h._renderer is None or H != h.H or W != h.W:
            h._renderer = NVDiffrastRenderer((H, W))
            h.H, h.W = H, W
        return h._renderer


def _vf(mesh):
    if isinstance(mesh, (tuple, list)):
        v, f = mesh
    else:
        v, f = mesh.vertices, mesh.faces
    return np.asarray(v), np.asarray(f)


def _dev(renderer, a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(renderer.device)


def nvdiffrast_render_mesh_api(mesh, object_pose, H, W, K, anti_aliasing=True):
    """render_api.py:27-45: mask of one mesh; object_pose = camera<-object (OpenCV axes).  Returns bool [H,W]
    (the antialiased float mask is cast with ``astype(bool)``, i.e. any partially covered pixel counts)."""
    renderer = NVdiffrastRenderMeshApiHelper.get_renderer(H, W)
    v, f = _vf(mesh)
    mask = renderer.render_mask(_dev(renderer, v, torch.float32), _dev(renderer, f, torch.int32),
                                _dev(renderer, K, torch.float32), _dev(renderer, object_pose, torch.float32),
                                anti_aliasing=anti_aliasing)
    return mask.detach().cpu().numpy().astype(bool)


def nvdiffrast_render_meshes_api(meshes: List, object_poses, H, W, K, return_ndarray=True):
    """render_api.py:48-67: one non-antialiased render per mesh, ``stack.sum(0).clamp(max=1)``."""
    renderer = NVdiffrastRenderMeshApiHelper.get_renderer(H, W)
    if len(meshes) != len(object_poses):  # dl_ext.primitive.safe_zip
        raise ValueError("meshes and object_poses differ in length")
    Kt = _dev(renderer, K, torch.float32)
    masks = []
    for mesh, object_pose in zip(meshes, object_poses):
        v, f = _vf(mesh)
        masks.append(renderer.render_mask(_dev(renderer, v, torch.float32), _dev(renderer, f, torch.int32), Kt,
                                          _dev(renderer, object_pose, torch.float32), anti_aliasing=False))
    mask = torch.stack(masks).float().sum(0).clamp(max=1)
    if return_ndarray:
        mask = mask.cpu().numpy().astype(bool)
    return mask


def nvdiffrast_parallel_render_meshes_api(meshes: List, object_poses, H, W, K, return_ndarray=True):
    """render_api.py:70-96: vertices moved to the camera frame, all meshes packed into one (pytorch3d ``Meshes``
    verts_packed / faces_packed = concatenation with index offsets), ONE rasterize."""
    renderer = NVdiffrastRenderMeshApiHelper.get_renderer(H, W)
    if len(meshes) != len(object_poses):
        raise ValueError("meshes and object_poses differ in length")
    Kt = _dev(renderer, K, torch.float32)
    poses = _dev(renderer, np.stack(object_poses), torch.float32)
    verts_list, faces_list, base = [], [], 0
    for mesh, pose in zip(meshes, poses):
        v, f = _vf(mesh)
        v = _dev(renderer, v, torch.float32)
        vh = torch.cat([v, torch.ones_like(v[:, :1])], dim=1) @ pose.T      # utils_3d.transform_points
        verts_list.append(vh[:, :3] / vh[:, 3:])
        faces_list.append(_dev(renderer, f, torch.int32) + base)
        base += v.shape[0]
    verts, faces = torch.cat(verts_list).contiguous(), torch.cat(faces_list).contiguous()
    mask = renderer.batch_render_mask(verts, faces, Kt, anti_aliasing=False).float()
    if return_ndarray:
        mask = mask.cpu().numpy().astype(bool)
    return mask


class _RobotHelper:
    """render_api.py:99-142 (RenderXarmApiHelper / RenderFrankaApiHelper): cached meshes + kinematics."""
    _robots, _chains = {}, {}

    @staticmethod
    def robot(name):
        if name not in _RobotHelper._robots:
            _RobotHelper._robots[name] = load_robot(name)
        return _RobotHelper._robots[name]

    @staticmethod
    def chain(name, urdf_path):
        """FK from ``urdf_path`` when the file exists, else from the chain packaged with the robot asset."""
        if urdf_path and os.path.exists(urdf_path):
            if urdf_path not in _RobotHelper._chains:
                _RobotHelper._chains[urdf_path] = UrdfChain(urdf_path)
            return _RobotHelper._chains[urdf_path]
        return _RobotHelper.robot(name).chain


def _robot_meshes_poses(name, urdf_path, Tc_c2b, qpos):
    robot = _RobotHelper.robot(name)
    chain = _RobotHelper.chain(name, urdf_path)
    lp = chain.link_poses(qpos, robot.use_links)
    Tc = np.asarray(Tc_c2b, dtype=np.float64)
    return robot.meshes, [(Tc @ lp[i]).astype(np.float32) for i in range(len(robot.meshes))]


def nvdiffrast_render_xarm_api(urdf_path, robot_pose, qpos, H, W, K, return_ndarray=True):
    """render_api.py:145-159: xArm7 links 1..8 (link_base, link1..link7), one render per link."""
    meshes, poses = _robot_meshes_poses("xarm7", urdf_path, robot_pose, qpos)
    return nvdiffrast_render_meshes_api(meshes, poses, H, W, K, return_ndarray=return_ndarray)


def nvdiffrast_render_franka_api(urdf_path, Tc_c2b, qpos, H, W, K, return_ndarray=True):
    """render_api.py:162-177: Franka link0..7 + hand (links without a visual mesh are skipped)."""
    meshes, poses = _robot_meshes_poses("franka", urdf_path, Tc_c2b, qpos)
    return nvdiffrast_render_meshes_api(meshes, poses, H, W, K, return_ndarray=return_ndarray)


def nvdiffrast_parallel_render_xarm_api(urdf_path, robot_pose, qpos, H, W, K, return_ndarray=True):
    """render_api.py:180-192: same as the xArm API but through the packed single-rasterize path."""
    meshes, poses = _robot_meshes_poses("xarm7", urdf_path, robot_pose, qpos)
    return nvdiffrast_parallel_render_meshes_api(meshes, poses, H, W, K, return_ndarray=return_ndarray)
