"""Writes tests/golden/*.npz from the CPU oracle (run in the build container: python tools/make_golden.py).

The reference ships no golden vectors for this path and nvdiffrast cannot run here, so these fixtures are
ORACLE-generated regression pins (inputs + expected outputs); the oracle itself is pinned by the analytic tests in
tests/test_oracle_*.py.  Inputs are stored (or regenerated from the packaged robot + fixed seeds) so that the GPU
tests can replay them without the oracle being the only witness."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from easyhec_amd.config import XARM7_K_1280x720  # noqa: E402
from easyhec_amd.robot import load_robot  # noqa: E402
from easyhec_amd.synthetic import camera_Tc_c2b, lookat_pose, make_views, perturb_pose, scaled_K  # noqa: E402
from oracle import oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def config1():
    """BASELINE configs[0]: xArm7 zero-pose PLY as ONE link, 320x240, single view."""
    z = np.load(os.path.join(GOLD, "xarm7_zeropos.npz"))
    verts, faces = z["vertices"], z["faces"]
    H, W = 240, 320
    K = scaled_K(XARM7_K_1280x720, 0.25, W, H, recentre=True)
    Tb_b2c = lookat_pose(np.radians(60), np.radians(20), 1.0)
    Tb_b2c[2, 3] += 0.3
    Tc = np.linalg.inv(Tb_b2c)
    mvp = helpers.mvp_numpy(K, H, W, Tc, np.eye(4)[None, None])
    ref = np.zeros((1, H, W), np.float32)
    toff = np.array([0, faces.shape[0]], np.int32)
    voff = np.array([0, verts.shape[0]], np.int32)
    mask, loss, g = oracle.render_mask_loss(verts, faces, toff, voff, mvp, ref)
    np.savez_compressed(os.path.join(GOLD, "config1_zeropos_320x240.npz"), K=K, Tc_c2b=Tc, mvp=mvp, mask=mask,
                        loss=loss, grad_mvp=g)
    print("config1: covered", int((mask > 0.5).sum()), "fractional", int(((mask > 0) & (mask < 1)).sum()), "loss", loss)


def fused_small():
    rb = load_robot("xarm7")
    H, W, B = 120, 160, 2
    K = scaled_K(XARM7_K_1280x720, 0.125, W, H, True)
    _, lp = make_views(rb, B, seed=1)
    Tc = camera_Tc_c2b()
    mvp = helpers.mvp_numpy(K, H, W, perturb_pose(Tc), lp)
    verts, tris, toff, voff = helpers.scene_arrays(rb)
    ref = (oracle.render_mask_loss(verts, tris, toff, voff, helpers.mvp_numpy(K, H, W, Tc, lp),
                                   np.zeros((B, H, W), np.float32), want_grad=False)[0] > 0.5).astype(np.float32)
    mask, loss, g = oracle.render_mask_loss(verts, tris, toff, voff, mvp, ref)
    np.savez_compressed(os.path.join(GOLD, "fused_xarm7_160x120.npz"), mvp=mvp, ref=np.packbits(ref > 0.5),
                        mask=mask, loss=loss, grad_mvp=g, H=H, W=W)
    print("fused_small: loss", loss)


def ops_random():
    rng = np.random.default_rng(42)
    H, W = 72, 104  # not multiples of the 32x8 tile
    pos, tri = helpers.random_mesh(rng, 300)
    attr = rng.uniform(0, 1, size=(1, pos.shape[0], 2)).astype(np.float32)
    rast, db = oracle.rasterize(pos[None], tri, [H, W])
    col = oracle.interpolate(attr, rast, tri)
    aa = oracle.antialias(col, rast, pos[None], tri)
    dy = rng.normal(size=aa.shape).astype(np.float32)
    gc, gp = oracle.antialias_grad(col, rast, pos[None], tri, dy)
    ga, gr = oracle.interpolate_grad(attr, rast, tri, gc)
    gp2 = oracle.rasterize_grad(pos[None], tri, rast, gr)
    np.savez_compressed(os.path.join(GOLD, "ops_random_72x104.npz"), pos=pos, tri=tri, attr=attr, rast=rast, db=db,
                        col=col, aa=aa, dy=dy, grad_attr=ga, grad_pos=gp + gp2, opp=oracle.topology(tri))
    print("ops_random: covered", int((rast[..., 3] > 0).sum()))


def score_small():
    """Space-explorer score (oracle.mask_variance): 4 candidate configurations x 5 camera poses at 160x120."""
    rb = load_robot("xarm7")
    H, W, Q, S = 120, 160, 4, 5
    K = scaled_K(XARM7_K_1280x720, 0.125, W, H, True)
    _, lp = make_views(rb, Q, seed=7, qpos_scale=0.8)
    rng = np.random.default_rng(8)
    Tc0 = camera_Tc_c2b()
    mvp = np.stack([helpers.mvp_numpy(K, H, W, perturb_pose(Tc0, dt=rng.normal(0, 0.02, 3),
                                                            drot_deg=rng.normal(0, 2.0, 3)), lp) for _ in range(S)], axis=1)
    verts, tris, _, _ = helpers.scene_arrays(rb)
    vl = np.concatenate([np.full(v.shape[0], l, np.int32) for l, (v, _) in enumerate(rb.meshes)])
    score, counts = oracle.mask_variance(verts, tris, vl, mvp, H, W, return_counts=True)
    np.savez_compressed(os.path.join(GOLD, "score_xarm7_160x120.npz"), mvp=mvp, score=score, counts=counts, H=H, W=W)
    print("score_small:", score)


if __name__ == "__main__":
    which = sys.argv[1:] or ["config1", "fused_small", "ops_random", "score_small"]
    for name in which:
        globals()[name]()
