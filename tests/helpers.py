"""Shared scene builders for the tests (numpy only)."""
import numpy as np


def random_mesh(rng, n_tri, spread=0.9, size=0.25, zrange=(-0.5, 0.5), w_range=(0.8, 1.6), shared=True):
    """Random triangle soup in clip space with perspective w; vertices partly shared so that edges have neighbours."""
    if shared:
        nv = max(4, n_tri // 2 + 2)
        centers = rng.uniform(-spread, spread, size=(nv, 2))
        z = rng.uniform(*zrange, size=(nv, 1))
        w = rng.uniform(*w_range, size=(nv, 1))
        pos = np.concatenate([centers * w, z * w, w], axis=1).astype(np.float32)
        # connect near neighbours to keep triangles small
        tri = []
        for _ in range(n_tri):
            a = rng.integers(nv)
            d = np.linalg.norm(centers - centers[a], axis=1)
            near = np.argsort(d)[1:8]
            b, c = rng.choice(near, 2, replace=False)
            tri.append([a, b, c])
        return pos, np.array(tri, dtype=np.int32)
    pts = []
    for _ in range(n_tri):
        c = rng.uniform(-spread, spread, size=2)
        for _ in range(3):
            xy = c + rng.uniform(-size, size, size=2)
            w = rng.uniform(*w_range)
            z = rng.uniform(*zrange)
            pts.append([xy[0] * w, xy[1] * w, z * w, w])
    pos = np.array(pts, dtype=np.float32)
    tri = np.arange(3 * n_tri, dtype=np.int32).reshape(-1, 3)
    return pos, tri


def grid_mesh(n, z=0.0, lo=-0.6, hi=0.6, jitter=0.0, rng=None, w=1.0):
    """(n x n)-quad planar grid -> 2 n^2 triangles with shared vertices."""
    xs = np.linspace(lo, hi, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    P = np.stack([X.ravel(), Y.ravel()], axis=1)
    if jitter and rng is not None:
        P = P + rng.uniform(-jitter, jitter, size=P.shape)
    pos = np.concatenate([P * w, np.full((P.shape[0], 1), z * w), np.full((P.shape[0], 1), w)], axis=1)
    tri = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i
            tri.append([a, a + 1, a + n + 2])
            tri.append([a, a + n + 2, a + n + 1])
    return pos.astype(np.float32), np.array(tri, dtype=np.int32)


def projection(K, H, W, n=0.001, f=10.0):
    """numpy restatement of easyhec/utils/nvdiffrast_utils.py:5-11 (tests only)."""
    return np.array([[2 * K[0, 0] / W, 0, -2 * K[0, 2] / W + 1, 0],
                     [0, 2 * K[1, 1] / H, 2 * K[1, 2] / H - 1, 0],
                     [0, 0, -(f + n) / (f - n), -2 * f * n / (f - n)],
                     [0, 0, -1, 0]], dtype=np.float64)


def scene_arrays(robot):
    """Concatenated scene of a Robot: verts, tris (global ids), tri_off, vert_off."""
    verts = np.concatenate([v for v, _ in robot.meshes]).astype(np.float32)
    voff = np.cumsum([0] + [v.shape[0] for v, _ in robot.meshes]).astype(np.int32)
    toff = np.cumsum([0] + [f.shape[0] for _, f in robot.meshes]).astype(np.int32)
    tris = np.concatenate([f + voff[i] for i, (_, f) in enumerate(robot.meshes)]).astype(np.int32)
    return verts, tris, toff, voff


def mvp_numpy(K, H, W, Tc_c2b, link_poses):
    """[B,L,4,4] float32 = proj @ opencv2blender @ Tc_c2b @ link_pose in float64, rounded once."""
    proj = projection(np.asarray(K, dtype=np.float64), H, W)
    o2b = np.diag([1.0, -1.0, -1.0, 1.0])
    return (proj @ o2b @ np.asarray(Tc_c2b, dtype=np.float64) @ np.asarray(link_poses, dtype=np.float64)).astype(
        np.float32)


def box_mesh(v):
    """Axis-aligned bounding box of a vertex set as a closed 12-triangle mesh (coarse stand-in for a link)."""
    lo, hi = np.asarray(v).min(0), np.asarray(v).max(0)
    c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]],
                 np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5],
                  [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]], np.int32)
    return c, f
