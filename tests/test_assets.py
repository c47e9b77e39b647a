"""Geometry fixtures: the packaged link meshes + URDF chain reproduce the reference's own FK fixture
(assets/xarm7_zeropos.ply == FK(q=0) o link0..7.STL), and the loaders round-trip small files."""
import os
import struct

import numpy as np

from easyhec_amd.kinematics import UrdfChain, rpy_to_matrix
from easyhec_amd.mesh_io import load_dae, load_ply, load_stl, merge_corners, merge_vertices
from easyhec_amd.robot import load_robot

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_xarm7_fk_reproduces_the_reference_zero_pose_ply(xarm7):
    z = np.load(os.path.join(GOLD, "xarm7_zeropos.npz"))
    pv, pf = z["vertices"], z["faces"]
    assert xarm7.num_tris == 41096 == pf.shape[0] and xarm7.num_verts == 20525 == pv.shape[0]
    assert [f.shape[0] for _, f in xarm7.meshes] == [6094, 3682, 4510, 4740, 4872, 4580, 7986, 4632]
    poses = xarm7.link_poses(np.zeros(7))
    V, F, nv = [], [], 0
    for (v, f), T in zip(xarm7.meshes, poses):
        V.append(v.astype(np.float64) @ T[:3, :3].T + T[:3, 3])
        F.append(f + nv)
        nv += v.shape[0]
    V, F = np.concatenate(V), np.concatenate(F)
    assert np.abs(V[F] - pv[pf]).max() < 1e-5  # same face order, metres
    c = z["corner_3d"]
    assert np.abs(V.min(0) - c.min(0)).max() < 1e-5 and np.abs(V.max(0) - c.max(0)).max() < 1e-5


def test_link_indices_follow_the_reference_config(xarm7):
    assert xarm7.use_links == [1, 2, 3, 4, 5, 6, 7, 8]  # link_base, link1..7 (defaults.py:60 uses [2..8] = link1..7)
    assert xarm7.chain.link_order[:9] == ["world", "link_base", "link1", "link2", "link3", "link4", "link5",
                                          "link6", "link7"]
    fr = load_robot("franka")
    assert fr.num_tris == 133676 and fr.num_links == 9 and fr.use_links == [0, 1, 2, 3, 4, 5, 6, 7, 9]
    assert fr.chain.link_order[9] == "panda_hand"


def test_fk_joint_rotation_is_about_the_joint_axis(xarm7):
    q = np.zeros(7)
    q[0] = 0.7
    p0 = xarm7.link_poses(np.zeros(7))
    p1 = xarm7.link_poses(q)
    assert np.abs(p0[0] - p1[0]).max() < 1e-12  # base unaffected
    R = p0[1][:3, :3].T @ p1[1][:3, :3]
    assert abs(R[0, 0] - np.cos(0.7)) < 1e-9 and abs(R[1, 0] - np.sin(0.7)) < 1e-9 and abs(R[2, 2] - 1) < 1e-9
    assert abs(np.linalg.det(rpy_to_matrix([0.3, -0.2, 1.1])) - 1) < 1e-12


def test_stl_ply_dae_round_trip(tmp_path):
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    faces = np.array([[0, 1, 2], [0, 1, 3], [1, 2, 3]], np.int32)
    p = tmp_path / "t.stl"
    with open(p, "wb") as f:
        f.write(b"\0" * 80 + struct.pack("<I", len(faces)))
        for tri in faces:
            f.write(struct.pack("<3f", 0, 0, 0))
            for vi in tri:
                f.write(struct.pack("<3f", *verts[vi]))
            f.write(struct.pack("<H", 0))
    v, f = load_stl(str(p))
    assert v.shape == (4, 3) and f.shape == (3, 3) and np.abs(v[f] - verts[faces]).max() == 0
    p2 = tmp_path / "t.ply"
    with open(p2, "wb") as fh:
        fh.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\n"
                 b"property float z\nelement face 3\nproperty list uchar int vertex_indices\nend_header\n")
        fh.write(verts.tobytes())
        for tri in faces:
            fh.write(struct.pack("<B3i", 3, *tri))
    v2, f2 = load_ply(str(p2))
    assert np.abs(v2 - verts).max() == 0 and (f2 == faces).all()
    p3 = tmp_path / "t.dae"
    p3.write_text("""<?xml version="1.0"?><COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema">
<library_geometries><geometry id="g"><mesh><source id="p"><float_array id="pa" count="12">0 0 0 1 0 0 0 1 0 0 0 1</float_array>
<technique_common><accessor source="#pa" count="4" stride="3"/></technique_common></source>
<vertices id="v"><input semantic="POSITION" source="#p"/></vertices>
<triangles count="3"><input semantic="VERTEX" source="#v" offset="0"/><p>0 1 2 0 1 3 1 2 3</p></triangles></mesh></geometry>
</library_geometries><library_visual_scenes><visual_scene id="s"><node><matrix>1 0 0 0 0 1 0 0 0 0 1 -0.5 0 0 0 1</matrix>
<instance_geometry url="#g"/></node></visual_scene></library_visual_scenes></COLLADA>""")
    v3, f3 = load_dae(str(p3))
    assert np.abs(v3[f3] - (verts[faces] + [0, 0, -0.5])).max() < 1e-12  # node matrix applied (franka link1.dae case)


def test_merge_vertices_keeps_face_order_and_drops_unreferenced():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 0, 0], [5, 5, 5], [0, 1, 1e-10]])
    f = np.array([[0, 1, 2], [3, 5, 0]])
    mv, mf = merge_vertices(v, f)
    assert mv.shape[0] == 3 and mf.tolist() == [[0, 1, 2], [1, 2, 0]]


def test_urdf_chain_spec_round_trip(xarm7):
    ch2 = UrdfChain(spec=xarm7.chain.spec())
    q = np.linspace(-0.5, 0.5, 7)
    assert np.abs(ch2.link_poses(q, xarm7.use_links) - xarm7.link_poses(q)).max() == 0


def test_merge_corners_respects_normals_like_trimesh():
    """Two triangles sharing an edge: merged by position when their corner normals agree to 1e-2, kept apart when the
    faces carry different (flat) normals -- trimesh.merge_vertices(merge_norm=False, digits_norm=2)."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], float)
    up = np.tile([0, 0, 1.0], (6, 1))
    v, f = merge_corners(pos, up)
    assert v.shape[0] == 4 and f.tolist() == [[0, 1, 2], [1, 3, 2]]
    tilted = up.copy()
    tilted[3:] = [0, 0.1, 0.995]
    v2, f2 = merge_corners(pos, tilted)
    assert v2.shape[0] == 6 and f2.tolist() == [[0, 1, 2], [3, 4, 5]]
    almost = up.copy()
    almost[3:] = [0, 0.001, 1.0]                      # rounds to the same 1e-2 cell
    assert merge_corners(pos, almost)[0].shape[0] == 4
    assert merge_corners(pos, None)[0].shape[0] == 4  # no normals: position only (the STL case)


def test_franka_collada_meshes_stay_nearly_unmerged():
    """The Franka visual meshes carry flat per-face normals, so the reference (trimesh) renders them almost as a
    vertex soup: ~2.8 corners per triangle survive the merge.  This is what makes dr.antialias see nearly every edge
    as a silhouette edge and gives the dense gradients the reference's Franka example converges with."""
    fr = load_robot("franka")
    assert fr.num_tris == 133676
    assert 2.5 * fr.num_tris < fr.num_verts <= 3 * fr.num_tris
    xa = load_robot("xarm7")
    assert xa.num_verts < 0.6 * xa.num_tris            # STL: merged by position


def test_batched_fk_equals_per_configuration_fk(xarm7):
    """Vectorised FK (space explorer: ~1000 candidate configurations per round) against the per-configuration chain."""
    from easyhec_amd.robot import load_robot
    for robot in (xarm7, load_robot("franka")):
        q = robot.sample_qpos(50, np.random.default_rng(3), scale=1.0)
        a = np.stack([robot.link_poses(x) for x in q])
        b = robot.link_poses_batch(q)
        assert a.shape == b.shape == (50, robot.num_links, 4, 4) and np.abs(a - b).max() <= 1e-12
        assert np.abs(robot.link_poses_batch(q[:1, :3])[0] - robot.link_poses(q[0, :3])).max() <= 1e-12  # zero padding
