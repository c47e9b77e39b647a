"""Forward kinematics of a URDF serial chain -- the producer of the hot path's ``link_poses`` input.

The reference obtains link poses from SAPIEN/Pinocchio
(/root/reference/easyhec/structures/sapien_kin.py:26-30, called per frame and link in
/root/reference/easyhec/data/datasets/xarm_real.py:42-56).  Only revolute/prismatic/fixed joints of a tree whose
root is fixed at the identity are needed for the xArm7 and Franka URDFs, so that is what is implemented.
Link indices follow SAPIEN's articulation order, which for these URDFs is the order in which <link> elements are
reached by a breadth-first walk from the root (== document order for the arm links).
"""
import xml.etree.ElementTree as ET

import numpy as np

__all__ = ["UrdfChain", "rpy_to_matrix"]


def rpy_to_matrix(rpy):
    r, p, y = [float(v) for v in rpy]
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]], dtype=np.float64)


def _origin(elem):
    T = np.eye(4)
    if elem is not None:
        T[:3, :3] = rpy_to_matrix((elem.get("rpy") or "0 0 0").split())
        T[:3, 3] = [float(v) for v in (elem.get("xyz") or "0 0 0").split()]
    return T


def _axis_angle(axis, q):
    a = np.asarray(axis, dtype=np.float64)
    a = a / (np.linalg.norm(a) + 1e-30)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(4)
    R[:3, :3] = np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)
    return R


class UrdfChain:
    """Minimal URDF kinematic tree: ``compute_forward_kinematics(qpos) -> {link_name: 4x4}``."""

    def __init__(self, urdf_path=None, spec=None):
        if spec is None:
            spec = self.parse(urdf_path)
        self.links = list(spec["links"])
        self.joints = [dict(j) for j in spec["joints"]]
        for j in self.joints:
            j["origin"] = np.asarray(j["origin"], dtype=np.float64).reshape(4, 4)
            j["axis"] = np.asarray(j["axis"], dtype=np.float64)
        children = {j["child"] for j in self.joints}
        roots = [l for l in self.links if l not in children]
        self.root = roots[0]
        # articulation order: BFS from the root, joints in document order
        order, queue = [], [self.root]
        self._joint_of_child = {j["child"]: j for j in self.joints}
        while queue:
            l = queue.pop(0)
            order.append(l)
            queue.extend(j["child"] for j in self.joints if j["parent"] == l)
        self.link_order = order
        self.active = [self._joint_of_child[l] for l in order if l in self._joint_of_child and
                       self._joint_of_child[l]["type"] in ("revolute", "continuous", "prismatic")]
        self.dof = len(self.active)

    @staticmethod
    def parse(urdf_path):
        root = ET.parse(urdf_path).getroot()
        links = [l.get("name") for l in root.findall("link")]
        joints = []
        for j in root.findall("joint"):
            ax = j.find("axis")
            lim = j.find("limit")
            joints.append({
                "name": j.get("name"), "type": j.get("type"),
                "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
                "origin": _origin(j.find("origin")).tolist(),
                "axis": [float(v) for v in (ax.get("xyz") if ax is not None else "1 0 0").split()],
                "lower": float(lim.get("lower", "0")) if lim is not None else 0.0,
                "upper": float(lim.get("upper", "0")) if lim is not None else 0.0,
            })
        return {"links": links, "joints": joints}

    def spec(self):
        return {"links": self.links,
                "joints": [{**{k: v for k, v in j.items() if k not in ("origin", "axis")},
                            "origin": j["origin"].tolist(), "axis": j["axis"].tolist()} for j in self.joints]}

    def limits(self):
        return np.array([[j["lower"], j["upper"]] for j in self.active], dtype=np.float64)

    def compute_forward_kinematics(self, qpos):
        """qpos: ``dof`` joint values in articulation order (shorter vectors are zero-padded, as
        xarm_real.py:47-48 does).  Returns {link name: base<-link 4x4 float64}."""
        q = np.zeros(self.dof)
        qpos = np.asarray(qpos, dtype=np.float64).reshape(-1)
        q[:min(self.dof, qpos.size)] = qpos[:self.dof]
        qmap = {id(j): q[i] for i, j in enumerate(self.active)}
        poses = {self.root: np.eye(4)}
        for l in self.link_order[1:]:
            j = self._joint_of_child[l]
            T = poses[j["parent"]] @ j["origin"]
            if j["type"] in ("revolute", "continuous"):
                T = T @ _axis_angle(j["axis"], qmap[id(j)])
            elif j["type"] == "prismatic":
                P = np.eye(4)
                P[:3, 3] = j["axis"] / (np.linalg.norm(j["axis"]) + 1e-30) * qmap[id(j)]
                T = T @ P
            poses[l] = T
        return poses

    def link_poses_batch(self, qposes, link_indices):
        """[N, len(link_indices), 4, 4] float64 for N joint vectors at once (same arithmetic as
        :meth:`compute_forward_kinematics`, vectorised over the batch: the space explorer needs FK of ~1000 candidate
        configurations per round, space_explorer.py:98-150)."""
        Q = np.atleast_2d(np.asarray(qposes, dtype=np.float64))
        N = Q.shape[0]
        q = np.zeros((N, self.dof))
        q[:, :min(self.dof, Q.shape[1])] = Q[:, :self.dof]
        col = {id(j): i for i, j in enumerate(self.active)}
        eye = np.broadcast_to(np.eye(4), (N, 4, 4))
        poses = {self.root: eye}
        for l in self.link_order[1:]:
            j = self._joint_of_child[l]
            T = poses[j["parent"]] @ j["origin"]
            if j["type"] in ("revolute", "continuous"):
                a = j["axis"] / (np.linalg.norm(j["axis"]) + 1e-30)
                Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
                ang = q[:, col[id(j)]]
                R = np.tile(np.eye(4), (N, 1, 1))
                R[:, :3, :3] = np.eye(3) + np.sin(ang)[:, None, None] * Kx + (1 - np.cos(ang))[:, None, None] * (Kx @ Kx)
                T = T @ R
            elif j["type"] == "prismatic":
                P = np.tile(np.eye(4), (N, 1, 1))
                P[:, :3, 3] = (j["axis"] / (np.linalg.norm(j["axis"]) + 1e-30))[None] * q[:, col[id(j)], None]
                T = T @ P
            poses[l] = T
        return np.stack([poses[self.link_order[i]] for i in link_indices], axis=1).astype(np.float64)

    def link_poses(self, qpos, link_indices):
        """[len(link_indices),4,4] poses for SAPIEN-style link indices (``use_links`` in the reference configs)."""
        poses = self.compute_forward_kinematics(qpos)
        return np.stack([poses[self.link_order[i]] for i in link_indices]).astype(np.float64)
