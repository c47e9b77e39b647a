"""GPU parity of the space-explorer scoring kernel (ehr_mask_variance through the C ABI) and of the render_api
facade against the CPU oracle: integer results, so the bar is exact equality."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(xarm7):
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from easyhec_amd import dr, fused, space_explorer
    dev = torch.device("cuda:0")
    ctx = dr.RasterizeCudaContext()
    scene = fused.LinkScene([v for v, _ in xarm7.meshes], [f for _, f in xarm7.meshes], dev)
    return space_explorer, ctx, scene, dev


def candidate_mvps(xarm7, H, W, scale, Q, S, seed, radius=1.3, spread=1.0):
    """mvp [Q,S,L,4,4]: Q random joint configurations seen from S cameras scattered around one look-at pose (the
    spread of the optimiser's pose history, space_explorer.py:87-91)."""
    from easyhec_amd.config import XARM7_K_1280x720
    from easyhec_amd.synthetic import camera_Tc_c2b, make_views, perturb_pose, scaled_K
    K = scaled_K(XARM7_K_1280x720, scale, W, H, scale != 1.0)
    _, lp = make_views(xarm7, Q, seed=seed, qpos_scale=0.8)
    rng = np.random.default_rng(seed + 1)
    Tc0 = camera_Tc_c2b(radius=radius)
    out = np.empty((Q, S) + lp.shape[1:], np.float32)
    for s in range(S):
        Tc = perturb_pose(Tc0, dt=rng.normal(0, 0.02 * spread, 3), drot_deg=rng.normal(0, 2.0 * spread, 3))
        out[:, s] = helpers.mvp_numpy(K, H, W, Tc, lp)
    return out


def vert_link_of(xarm7):
    return np.concatenate([np.full(v.shape[0], l, np.int32) for l, (v, _) in enumerate(xarm7.meshes)])


@pytest.mark.parametrize("H,W,scale,Q,S,chunk", [(120, 160, 0.125, 5, 4, 0), (240, 320, 0.25, 7, 3, 6),
                                                 (100, 150, 0.12, 3, 10, 10), (64, 96, 0.07, 4, 1, 0)])
def test_mask_variance_matches_oracle(env, oracle, xarm7, H, W, scale, Q, S, chunk):
    se, ctx, scene, dev = env
    mvp = candidate_mvps(xarm7, H, W, scale, Q, S, seed=H)
    verts, tris, _, _ = helpers.scene_arrays(xarm7)
    s_ref, c_ref = oracle.mask_variance(verts, tris, vert_link_of(xarm7), mvp, H, W, return_counts=True)
    var, score, counts = se.mask_variance(ctx, scene, torch.tensor(mvp, device=dev), H, W, return_counts=True,
                                          chunk_views=chunk)
    assert (counts.cpu().numpy() == c_ref).all()
    assert (score.cpu().numpy() == s_ref).all()
    if S > 1:
        assert s_ref.min() > 0
        assert np.allclose(var.cpu().numpy(), s_ref / (S * (S - 1.0)), rtol=1e-6)
    else:
        assert (s_ref == 0).all() and c_ref.max() == 1


def test_mask_variance_slow_tiles_match_oracle(env, oracle, xarm7):
    """Cameras a few centimetres from the robot / 12x zoom: near-plane clipping and triangles of several hundred
    pixels route their tiles through the 64-bit instantiation."""
    se, ctx, scene, dev = env
    from easyhec_amd.config import XARM7_K_1280x720
    from easyhec_amd.synthetic import camera_Tc_c2b, make_views, scaled_K
    H, W, Q = 240, 320, 2
    K = scaled_K(XARM7_K_1280x720, 0.25, W, H, True)
    Kz = K.copy()
    Kz[:2, :2] *= 12.0
    _, lp = make_views(xarm7, Q, seed=4)
    cams = [(K, camera_Tc_c2b(radius=0.12, lift=0.15)), (Kz, camera_Tc_c2b(radius=0.45, lift=0.2)),
            (K, camera_Tc_c2b(radius=0.9))]
    mvp = np.stack([helpers.mvp_numpy(k, H, W, tc, lp) for k, tc in cams], axis=1)
    verts, tris, _, _ = helpers.scene_arrays(xarm7)
    s_ref, c_ref = oracle.mask_variance(verts, tris, vert_link_of(xarm7), mvp, H, W, return_counts=True)
    assert (c_ref > 0).mean() > 0.2
    _, score, counts = se.mask_variance(ctx, scene, torch.tensor(mvp, device=dev), H, W, return_counts=True)
    assert (counts.cpu().numpy() == c_ref).all() and (score.cpu().numpy() == s_ref).all()


def test_mask_variance_equals_torch_var_of_three_op_renders(env, xarm7):
    """The reference's own formulation: S non-antialiased renders of the packed mesh through dr.rasterize
    (render_api.py:70-96), stacked, torch.var(dim=0).sum() (space_explorer.py:163-164)."""
    se, ctx, scene, dev = env
    from easyhec_amd import dr
    H, W, Q, S = 240, 320, 3, 5
    mvp = candidate_mvps(xarm7, H, W, 0.25, Q, S, seed=11)
    tm = torch.tensor(mvp, device=dev)
    var, score, counts = se.mask_variance(ctx, scene, tm, H, W, return_counts=True)
    ones = torch.ones((scene.num_verts, 1), device=dev)
    for q in range(Q):
        masks = []
        for s in range(S):
            M = tm[q, s][scene.vert_link.long()]                             # [V,4,4]
            pos = (M @ torch.cat([scene.verts, ones], 1)[..., None])[..., 0]  # [V,4]
            rast, _ = dr.rasterize(ctx, pos[None].contiguous(), scene.tris, resolution=[H, W])
            masks.append(torch.flip(rast[0, :, :, 2] > 0, dims=[0]))
        masks = torch.stack(masks)
        ref = torch.var(masks.reshape(S, -1).float(), dim=0).sum()
        assert abs(float(var[q]) - float(ref)) <= 2e-4 * float(ref)
        # torch.matmul rounds positions differently from the kernel's fma chain: allow a few boundary pixels
        assert (masks.sum(0).to(torch.uint8) != counts[q]).float().mean() < 1e-4


def test_mask_variance_properties_at_full_size(env, xarm7):
    """1280x720, 10 poses (the reference's cfg.model.space_explorer defaults): size-independent identities."""
    se, ctx, scene, dev = env
    H, W, Q, S = 720, 1280, 24, 10
    mvp = torch.tensor(candidate_mvps(xarm7, H, W, 1.0, Q, S, seed=5), device=dev)
    var, score, counts = se.mask_variance(ctx, scene, mvp, H, W, return_counts=True)
    c = counts.long()
    assert (score == (c * (S - c)).sum(dim=(1, 2))).all()           # score is the sum over the count image
    assert int(c.max()) <= S and float((c > 0).float().mean()) > 0.02
    # pose order does not matter; chunking does not matter
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(0)).to(dev)
    _, score_p = se.mask_variance(ctx, scene, mvp[:, perm], H, W, chunk_views=40)
    assert (score_p == score).all()
    # identical poses -> zero variance, and the count image is S x the single mask
    same = mvp[:, :1].expand(-1, S, -1, -1, -1).contiguous()
    _, score0, counts0 = se.mask_variance(ctx, scene, same, H, W, return_counts=True)
    assert (score0 == 0).all() and set(torch.unique(counts0).tolist()) <= {0, S}
    # pairwise identity: sum_px c (S - c) = sum_{s < s'} |mask_s xor mask_s'|
    _, s2 = se.mask_variance(ctx, scene, mvp[:2, :2], H, W)
    _, _, ca = se.mask_variance(ctx, scene, mvp[:2, :1], H, W, return_counts=True)
    _, _, cb = se.mask_variance(ctx, scene, mvp[:2, 1:2], H, W, return_counts=True)
    assert (s2 == (ca != cb).sum(dim=(1, 2))).all()


def test_space_explorer_picks_the_best_candidate(env, xarm7):
    se, ctx, scene, dev = env
    from easyhec_amd.config import XARM7_K_1280x720
    from easyhec_amd.synthetic import camera_Tc_c2b, perturb_pose
    H, W = 360, 640
    K = np.array(XARM7_K_1280x720) * 0.5
    K[2, 2] = 1.0
    ex = se.SpaceExplorer(xarm7, K, H, W, device=dev)
    rng = np.random.default_rng(0)
    # a fake optimisation history: 300 poses jittering around one camera pose
    Tc0 = camera_Tc_c2b()
    hist = np.zeros((1000, 6), np.float32)
    for i in range(300):
        Tc = perturb_pose(Tc0, dt=rng.normal(0, 0.01, 3), drot_deg=rng.normal(0, 1.0, 3))
        hist[i] = _log(Tc)
    qposes = xarm7.sample_qpos(12, rng, scale=0.8)
    valid = np.ones(12, bool)
    valid[3] = False
    out = ex.forward(qposes, hist, start=200, sample=10, valid=valid, generator=torch.Generator().manual_seed(1))
    v = out["variances"].numpy()
    assert v[3] == 0 and (np.delete(v, 3) > 0).all()
    assert int(out["qpos_idx"]) == int(np.argmax(v)) and float(out["variance"]) == float(v.max())
    assert float(out["var_min"]) == float(v[v > 0].min()) and np.allclose(out["qpos"], qposes[int(np.argmax(v))])


def _log(T):
    from easyhec_amd.se3 import se3_log_map
    return se3_log_map(torch.tensor(np.asarray(T, np.float32).T[None]), backend="opencv", test_acc=False)[0].numpy()


def test_render_api_matches_oracle(env, oracle, xarm7):
    """render_api facade (render_api.py:27-96,145-192): per-mesh and packed non-antialiased masks."""
    se, ctx, scene, dev = env
    from easyhec_amd import render_api
    from easyhec_amd.config import XARM7_K_1280x720
    from easyhec_amd.synthetic import camera_Tc_c2b, scaled_K
    H, W = 240, 320
    K = scaled_K(XARM7_K_1280x720, 0.25, W, H, True)
    Tc = camera_Tc_c2b()
    q = xarm7.sample_qpos(1, np.random.default_rng(2), scale=0.6)[0]
    m_links = render_api.nvdiffrast_render_xarm_api(None, Tc, q, H, W, K)
    m_packed = render_api.nvdiffrast_parallel_render_xarm_api(None, Tc, q, H, W, K)
    assert m_links.dtype == bool and m_links.shape == (H, W) and m_packed.dtype == bool
    # oracle: rasterize each link with the same float32 clip matrix the facade builds
    lp = xarm7.link_poses(q)
    proj = helpers.projection(np.asarray(K, np.float64), H, W).astype(np.float32)
    o2b = np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)
    ref = np.zeros((H, W), bool)
    for l, (v, f) in enumerate(xarm7.meshes):
        pose = (Tc @ lp[l]).astype(np.float32)
        mtx = proj @ (o2b @ pose)
        rast, _ = oracle.rasterize(oracle.transform_pos(mtx, v), f, (H, W), grad_db=False)
        ref |= (rast[0, ::-1, :, 2] > 0)
    assert (m_links != ref).mean() < 2e-4        # torch.matmul vs fma chain: a few boundary pixels at most
    assert (m_packed != ref).mean() < 2e-4
    assert 0.02 < ref.mean() < 0.5
    one = render_api.nvdiffrast_render_mesh_api(xarm7.meshes[0], (Tc @ lp[0]).astype(np.float32), H, W, K)
    hard = render_api.nvdiffrast_render_mesh_api(xarm7.meshes[0], (Tc @ lp[0]).astype(np.float32), H, W, K,
                                                 anti_aliasing=False)
    assert (one | hard == one).all() and one.sum() >= hard.sum() > 0   # AA mask cast to bool only adds pixels


def test_mask_variance_golden_fixture(env):
    import os
    se, ctx, scene, dev = env
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "score_xarm7_160x120.npz"))
    _, score, counts = se.mask_variance(ctx, scene, torch.tensor(g["mvp"], device=dev), int(g["H"]), int(g["W"]),
                                        return_counts=True)
    assert (score.cpu().numpy() == g["score"]).all() and (counts.cpu().numpy() == g["counts"]).all()


def test_mask_variance_edge_cases(env, oracle, xarm7):
    """Argument validation at the C ABI, a single candidate, the maximum number of poses, nothing in view."""
    se, ctx, scene, dev = env
    H, W = 64, 96
    mvp = torch.tensor(candidate_mvps(xarm7, H, W, 0.07, 1, 2, seed=1), device=dev)
    with pytest.raises(RuntimeError, match="S must be in"):
        se.mask_variance(ctx, scene, mvp[:, :1].expand(-1, 256, -1, -1, -1).contiguous(), H, W)
    with pytest.raises(ValueError):
        se.mask_variance(ctx, scene, mvp[:, :, :3], H, W)               # wrong number of links
    with pytest.raises(RuntimeError):
        se.mask_variance(ctx, scene, mvp.cpu(), H, W)                    # no CPU path
    # S = 255 (the 8-bit count limit): 255 copies of two alternating poses -> c in {0, 127, 128, 255}
    big = mvp[:, [0, 1] * 127 + [0]].contiguous()
    _, score, counts = se.mask_variance(ctx, scene, big, H, W, return_counts=True, chunk_views=64)
    _, _, ca = se.mask_variance(ctx, scene, mvp[:, :1], H, W, return_counts=True)
    _, _, cb = se.mask_variance(ctx, scene, mvp[:, 1:], H, W, return_counts=True)
    expect = 128 * ca.long() + 127 * cb.long()
    assert (counts.long() == expect).all()
    assert int(score[0]) == int((expect * (255 - expect)).sum())
    # nothing in view: all-zero counts and score, no error
    far = mvp.clone()
    far[..., 3, :] *= -1.0                                               # w < 0 for every vertex
    _, s0, c0 = se.mask_variance(ctx, scene, far, H, W, return_counts=True)
    assert int(s0.abs().sum()) == 0 and int(c0.sum()) == 0


def test_mask_variance_chain_equals_tile_path(env, oracle, xarm7, monkeypatch):
    """The coverage-only chain on the solver's cluster / job machinery (the default where the call allows it) against the
    per-triangle queue path and the oracle: counts and scores integer-equal, several chunks of candidates, a count image;
    then the mesh is edited IN PLACE (same device pointers): the cached cluster index must notice and follow."""
    se, ctx, _, dev = env
    from easyhec_amd import fused
    H, W, Q, S = 180, 320, 9, 10
    mvp = candidate_mvps(xarm7, H, W, 0.25, Q, S, seed=21)
    scene = fused.LinkScene([v.copy() for v, _ in xarm7.meshes], [f for _, f in xarm7.meshes], dev)
    verts, tris, _, _ = helpers.scene_arrays(xarm7)
    mvp_t = torch.tensor(mvp, device=dev)

    def both():
        monkeypatch.setenv("EHR_SCORE_PATH", "chain")      # an error if the chain cannot take the call
        _, s1, c1 = se.mask_variance(ctx, scene, mvp_t, H, W, return_counts=True)
        monkeypatch.setenv("EHR_SCORE_PATH", "tile")
        _, s2, c2 = se.mask_variance(ctx, scene, mvp_t, H, W, return_counts=True)
        monkeypatch.delenv("EHR_SCORE_PATH")
        return s1.cpu().numpy(), c1.cpu().numpy(), s2.cpu().numpy(), c2.cpu().numpy()

    s1, c1, s2, c2 = both()
    s_ref, c_ref = oracle.mask_variance(verts, tris, vert_link_of(xarm7), mvp, H, W, return_counts=True)
    assert (s1 == s_ref).all() and (c1 == c_ref).all() and (s2 == s_ref).all() and (c2 == c_ref).all() and s_ref.min() > 0
    scene.verts.mul_(0.8)                                   # same pointers, other geometry
    s1b, c1b, s2b, c2b = both()
    s_refb, c_refb = oracle.mask_variance(verts * np.float32(0.8), tris, vert_link_of(xarm7), mvp, H, W, return_counts=True)
    assert (s1b == s_refb).all() and (c1b == c_refb).all() and (s2b == s_refb).all() and (c2b == c_refb).all()
    assert (s_refb != s_ref).any()


def test_mask_variance_chain_falls_back_near_the_camera(env, oracle, xarm7, monkeypatch):
    """Cameras a few centimetres from the robot and a 12x zoom hold triangles that cross the near plane or span hundreds
    of pixels: coverage alone cannot decide there.  The chain reports that (an error when it is demanded), the default
    call falls back to the exact path, results as the oracle's."""
    se, ctx, scene, dev = env
    from easyhec_amd.config import XARM7_K_1280x720
    from easyhec_amd.synthetic import camera_Tc_c2b, make_views, scaled_K
    H, W, Q = 240, 320, 2
    K = scaled_K(XARM7_K_1280x720, 0.25, W, H, True)
    Kz = K.copy()
    Kz[:2, :2] *= 12.0
    _, lp = make_views(xarm7, Q, seed=4)
    cams = [(K, camera_Tc_c2b(radius=0.12, lift=0.15)), (Kz, camera_Tc_c2b(radius=0.45, lift=0.2)),
            (K, camera_Tc_c2b(radius=0.9))]
    mvp = np.stack([helpers.mvp_numpy(k, H, W, tc, lp) for k, tc in cams], axis=1)
    verts, tris, _, _ = helpers.scene_arrays(xarm7)
    s_ref, c_ref = oracle.mask_variance(verts, tris, vert_link_of(xarm7), mvp, H, W, return_counts=True)
    _, s, c = se.mask_variance(ctx, scene, torch.tensor(mvp, device=dev), H, W, return_counts=True)
    assert (s.cpu().numpy() == s_ref).all() and (c.cpu().numpy() == c_ref).all()
    monkeypatch.setenv("EHR_SCORE_PATH", "chain")
    with pytest.raises(RuntimeError):
        se.mask_variance(ctx, scene, torch.tensor(mvp, device=dev), H, W, return_counts=True)
