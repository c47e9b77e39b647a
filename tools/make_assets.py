"""Generates the geometry fixtures shipped in easyhec_amd/assets/ from the reference's robot descriptions.

Run in the build container (needs /root/reference):  python tools/make_assets.py
Outputs are DATA (merged link meshes as float32/int32 arrays + the URDF joint table as JSON), not source:
  easyhec_amd/assets/xarm7.npz   link0..7.STL of assets/xarm_description/meshes/xarm7/visual (41 096 triangles)
                                 + joint chain of assets/xarm7_with_gripper_reduced_dof.urdf
  easyhec_amd/assets/franka.npz  link0..7.dae + hand.dae of assets/franka/franka_description/meshes/visual
                                 + joint chain of assets/franka/urdf/franka.urdf
and tests/golden/xarm7_zeropos_fk.npz (the reference's own FK fixture, assets/xarm7_zeropos.ply, decimated to the
per-link bounding boxes used by tests/test_assets.py).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easyhec_amd.kinematics import UrdfChain  # noqa: E402
from easyhec_amd.mesh_io import load_mesh, load_ply  # noqa: E402

REF = "/root/reference/assets"


def pack(name, mesh_paths, urdf, use_links, out):
    verts, faces, voff, toff = [], [], [0], [0]
    for p in mesh_paths:
        v, f = load_mesh(os.path.join(REF, p))
        verts.append(v.astype(np.float32))
        faces.append(f.astype(np.int32))
        voff.append(voff[-1] + v.shape[0])
        toff.append(toff[-1] + f.shape[0])
    chain = UrdfChain(os.path.join(REF, urdf))
    np.savez_compressed(out, vertices=np.concatenate(verts), faces=np.concatenate(faces),
                        vert_offsets=np.array(voff, np.int32), tri_offsets=np.array(toff, np.int32),
                        mesh_paths=np.array(mesh_paths), use_links=np.array(use_links, np.int32),
                        chain=np.array(json.dumps(chain.spec())), name=np.array(name))
    print(out, "links", len(mesh_paths), "verts", voff[-1], "tris", toff[-1], os.path.getsize(out), "bytes")


def main():
    adir = os.path.join(ROOT, "easyhec_amd", "assets")
    os.makedirs(adir, exist_ok=True)
    # xArm7: defaults.py:66-74 mesh list (link0..7), URDF link indices 1..8 (link_base, link1..7)
    pack("xarm7", [f"xarm_description/meshes/xarm7/visual/link{i}.STL" for i in range(8)],
         "xarm7_with_gripper_reduced_dof.urdf", list(range(1, 9)), os.path.join(adir, "xarm7.npz"))
    # Franka: configs/franka/example_franka_offline.yaml:10-18 and :39 (use_links [0..7, 9])
    pack("franka", [f"franka/franka_description/meshes/visual/link{i}.dae" for i in range(8)] +
         ["franka/franka_description/meshes/visual/hand.dae"],
         "franka/urdf/franka.urdf", [0, 1, 2, 3, 4, 5, 6, 7, 9], os.path.join(adir, "franka.npz"))
    # the reference's FK fixture: zero-pose PLY
    pv, pf = load_ply(os.path.join(REF, "xarm7_zeropos.ply"))
    meta = np.load(os.path.join(REF, "xarm7_meta.npy"), allow_pickle=True).item()
    gdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gdir, exist_ok=True)
    np.savez_compressed(os.path.join(gdir, "xarm7_zeropos.npz"), vertices=pv.astype(np.float32),
                        faces=pf.astype(np.int32), corner_3d=meta["corner_3d"], K=meta["K"])
    print("zeropos", pv.shape, pf.shape)
    # the reference's only real image dataset (assets/franka_offline_example.zip): masks + qpos + K as arrays
    import io
    import zipfile
    from PIL import Image
    z = zipfile.ZipFile(os.path.join(REF, "franka_offline_example.zip"))
    masks, qpos = [], []
    for i in range(10):
        with Image.open(io.BytesIO(z.read(f"offline_example/mask/{i:06d}.png"))) as im:
            a = np.asarray(im)
        masks.append((a.max(axis=2) if a.ndim == 3 else a) > 0)
        qpos.append(np.loadtxt(io.BytesIO(z.read(f"offline_example/qpos/{i:06d}.txt"))))
    K = np.loadtxt(io.BytesIO(z.read("offline_example/K.txt")))
    masks = np.stack(masks)
    # init pose of configs/franka/example_franka_offline.yaml:5-8
    init = np.array([[9.3969262e-01, 3.4202009e-01, 6.4914198e-09, -6.4085639e-01],
                     [1.7101002e-01, -4.6984622e-01, -8.6602539e-01, 4.9582830e-01],
                     [-2.9619810e-01, 8.1379771e-01, -4.9999991e-01, 1.2412001e+00],
                     [0, 0, 0, 1]])
    np.savez_compressed(os.path.join(gdir, "franka_offline_example.npz"), masks=np.packbits(masks), shape=masks.shape,
                        qpos=np.stack(qpos), K=K, init_Tc_c2b=init)
    print("franka offline example", masks.shape, "foreground", masks.mean(axis=(1, 2)).round(3))


if __name__ == "__main__":
    main()
