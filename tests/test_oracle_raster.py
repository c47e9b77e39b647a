"""Pins the CPU oracle's rasterize/interpolate restatement with analytic known answers
(the reference holds no golden vectors for this path, so these are the pins)."""
import numpy as np
import pytest

import helpers


def quad(x0, y0, x1, y1, z=0.0, w=1.0):
    pos = np.array([[x0, y0, z, 1], [x1, y0, z, 1], [x1, y1, z, 1], [x0, y1, z, 1]], np.float32) * w
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return pos, tri


def ndc(px, n):  # pixel-edge coordinate (0..n) -> NDC
    return 2.0 * px / n - 1.0


def test_axis_aligned_quad_coverage_is_exact(oracle):
    H, W = 16, 24
    # covers pixel centres with 3.2 <= x+0.5 <= 9.7  and 2.6 <= y+0.5 <= 10.4
    pos, tri = quad(ndc(3.2, W), ndc(2.6, H), ndc(9.7, W), ndc(10.4, H))
    rast, _ = oracle.rasterize(pos[None], tri, [H, W])
    cov = rast[0, :, :, 3] > 0
    exp = np.zeros((H, W), bool)
    exp[3:10, 3:10] = True  # centres 3.5..9.5 in x, rows 3.5..9.5 in y (row 0 = bottom)
    assert (cov == exp).all()


def test_shared_edge_through_pixel_centres_is_assigned_once(oracle):
    H, W = 8, 8
    # diagonal of the quad passes exactly through pixel centres; each pixel must belong to exactly one triangle
    pos, tri = quad(ndc(1, W), ndc(1, H), ndc(7, W), ndc(7, H))
    rast, _ = oracle.rasterize(pos[None], tri, [H, W])
    ids = rast[0, :, :, 3]
    assert (ids[1:7, 1:7] > 0).all() and (ids > 0).sum() == 36
    # vertical shared edge at a pixel-centre column
    pos = np.array([[ndc(1, W), ndc(1, H), 0, 1], [ndc(3.5, W), ndc(1, H), 0, 1], [ndc(3.5, W), ndc(7, H), 0, 1],
                    [ndc(1, W), ndc(7, H), 0, 1], [ndc(6, W), ndc(1, H), 0, 1], [ndc(6, W), ndc(7, H), 0, 1]],
                   np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3], [1, 4, 5], [1, 5, 2]], np.int32)
    rast, _ = oracle.rasterize(pos[None], tri, [H, W])
    col = rast[0, 1:7, 3, 3]
    assert (col > 0).all()  # centre x = 3.5 lies on the shared edge: covered exactly once, by one side only
    left = np.isin(col, [1, 2])
    assert left.all() or (~left).all()


def test_both_windings_render(oracle):
    H, W = 16, 16
    pos, tri = quad(-0.5, -0.5, 0.5, 0.5)
    r0, _ = oracle.rasterize(pos[None], tri, [H, W])
    r1, _ = oracle.rasterize(pos[None], tri[:, ::-1].copy(), [H, W])
    assert ((r0[..., 3] > 0) == (r1[..., 3] > 0)).all()


def test_nearest_depth_wins_and_ties_go_to_lower_index(oracle):
    H, W = 16, 16
    p0, t0 = quad(-0.6, -0.6, 0.6, 0.6, z=0.3)
    p1, t1 = quad(-0.3, -0.3, 0.9, 0.9, z=-0.2)
    pos = np.concatenate([p0, p1])
    tri = np.concatenate([t0, t1 + 4])
    rast, _ = oracle.rasterize(pos[None], tri, [H, W])
    ids, zw = rast[0, :, :, 3], rast[0, :, :, 2]
    assert np.isin(ids[8, 8], [3, 4]) and abs(zw[8, 8] + 0.2) < 1e-6
    assert np.isin(ids[4, 4], [1, 2]) and abs(zw[4, 4] - 0.3) < 1e-6
    # identical geometry twice: the first copy wins everywhere
    pos2 = np.concatenate([p0, p0])
    tri2 = np.concatenate([t0, t0 + 4])
    r2, _ = oracle.rasterize(pos2[None], tri2, [H, W])
    assert r2[0, :, :, 3].max() <= 2


def test_barycentrics_depth_and_derivatives(oracle):
    H, W = 32, 32
    pos = np.array([[-0.8, -0.7, 0.1, 1.0], [0.9, -0.6, 0.4, 1.5], [0.1, 0.8, -0.3, 0.7]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    rast, db = oracle.rasterize(pos[None], tri, [H, W])
    ys, xs = np.nonzero(rast[0, :, :, 3] > 0)
    assert len(xs) > 50
    P = pos.astype(np.float64)

    def uv(fx, fy):
        # solve  sum_i b_i * (x_i, y_i, w_i) ~ (fx, fy, 1)  projectively
        A = np.stack([P[:, 0] - fx * P[:, 3], P[:, 1] - fy * P[:, 3], np.ones(3)])
        b = np.linalg.solve(A, np.array([0, 0, 1.0]))
        zw = (b * P[:, 2]).sum() / (b * P[:, 3]).sum()
        return b[0], b[1], zw

    for x, y in list(zip(xs, ys))[::17]:
        fx, fy = (x + 0.5) / W * 2 - 1, (y + 0.5) / H * 2 - 1
        u, v, zw = uv(fx, fy)
        assert abs(rast[0, y, x, 0] - u) < 2e-6 and abs(rast[0, y, x, 1] - v) < 2e-6
        assert abs(rast[0, y, x, 2] - zw) < 2e-6
        e = 1e-4
        ux = (uv(fx + e, fy)[0] - uv(fx - e, fy)[0]) / (2 * e) * 2 / W
        vy = (uv(fx, fy + e)[1] - uv(fx, fy - e)[1]) / (2 * e) * 2 / H
        assert abs(db[0, y, x, 0] - ux) < 1e-4 and abs(db[0, y, x, 3] - vy) < 1e-4


def test_opencv_projection_known_answer(oracle):
    """SURVEY 8a known answer: camera-frame vertex (x,y,z) lands at u = fu*x/z + cu, v = fv*y/z + cv and, after the
    wrapper's flip, image row r covers v in [r, r+1)."""
    H, W = 48, 64
    K = np.array([[50.0, 0, 31.3], [0, 52.0, 22.8], [0, 0, 1]])
    proj = helpers.projection(K, H, W) @ np.diag([1.0, -1, -1, 1])
    # a small camera-frame square around the point that projects to pixel (u0, v0)
    z = 2.0
    u0, v0 = 40.5, 10.5
    c = np.array([(u0 - K[0, 2]) * z / K[0, 0], (v0 - K[1, 2]) * z / K[1, 1], z])
    h = 0.6 * z / K[0, 0]  # +-0.6 px
    verts = np.array([c + [-h, -h, 0], c + [h, -h, 0], c + [h, h, 0], c + [-h, h, 0]], np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    pos = oracle.transform_pos(proj.astype(np.float32), verts)
    rast, _ = oracle.rasterize(pos, tri, [H, W])
    mask = rast[0, ::-1, :, 3] > 0  # flip to image rows
    ys, xs = np.nonzero(mask)
    assert set(zip(xs.tolist(), ys.tolist())) == {(40, 10)}
    zw = rast[0, ::-1][10, 40, 2]
    n, f = 0.001, 10.0
    assert abs(zw - ((f + n) / (f - n) - 2 * f * n / ((f - n) * z))) < 1e-5


def test_near_plane_clipping_and_behind_camera(oracle):
    H, W = 32, 32
    # one vertex behind the eye (w < 0): the visible part must still rasterize, with depth inside [-1, 1]
    pos = np.array([[-0.5, -0.5, 0.2, 1.0], [0.5, -0.5, 0.2, 1.0], [0.0, 3.0, -2.0, -0.5]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    rast, _ = oracle.rasterize(pos[None], tri, [H, W])
    cov = rast[0, :, :, 3] > 0
    assert cov.sum() > 10 and np.abs(rast[0, :, :, 2]).max() <= 1.0
    # fully behind: nothing
    pos2 = pos.copy()
    pos2[:, 3] = -1.0
    r2, _ = oracle.rasterize(pos2[None], tri, [H, W])
    assert (r2 == 0).all()
    # beyond the far plane: rejected per pixel
    pos3 = np.array([[-0.5, -0.5, 1.5, 1.0], [0.5, -0.5, 1.5, 1.0], [0.0, 0.5, 1.5, 1.0]], np.float32)
    r3, _ = oracle.rasterize(pos3[None], tri, [H, W])
    assert (r3 == 0).all()


def test_empty_degenerate_and_invalid_inputs(oracle):
    H, W = 8, 8
    pos = np.array([[0, 0, 0, 1], [0.5, 0.5, 0, 1], [1, 1, 0, 1], [np.nan, 0, 0, 1]], np.float32)
    tri = np.array([[0, 1, 2], [0, 0, 1], [0, 1, 7], [0, 1, 3]], np.int32)  # collinear, repeated, out of range, NaN
    rast, db = oracle.rasterize(pos[None], tri, [H, W])
    assert (rast == 0).all() and (db == 0).all()
    rast, _ = oracle.rasterize(pos[None], np.zeros((0, 3), np.int32), [H, W])
    assert (rast == 0).all()


def test_range_mode_equals_instance_mode(oracle):
    rng = np.random.default_rng(3)
    pos, tri = helpers.random_mesh(rng, 60)
    H, W = 40, 56
    full, _ = oracle.rasterize(pos[None], tri, [H, W])
    ranges = np.array([[0, 60], [10, 25], [59, 1]], np.int32)
    rr, _ = oracle.rasterize(pos, tri, [H, W], ranges=ranges)
    assert (rr[0] == full[0]).all()
    sub, _ = oracle.rasterize(pos[None], tri[10:35], [H, W])
    ids = sub[0, :, :, 3]
    exp = np.where(ids > 0, ids + 10, 0)
    assert (rr[1, :, :, 3] == exp).all() and (rr[1, :, :, :3] == sub[0, :, :, :3]).all()


def test_interpolate_matches_barycentric_sum_and_grad(oracle):
    rng = np.random.default_rng(5)
    pos, tri = helpers.random_mesh(rng, 40)
    H, W = 32, 32
    rast, _ = oracle.rasterize(pos[None], tri, [H, W])
    attr = rng.normal(size=(1, pos.shape[0], 3)).astype(np.float32)
    out = oracle.interpolate(attr, rast, tri)
    ids = rast[0, :, :, 3].astype(int)
    for y, x in zip(*np.nonzero(ids)):
        t = ids[y, x] - 1
        b0, b1 = rast[0, y, x, 0].astype(np.float64), rast[0, y, x, 1].astype(np.float64)
        e = b0 * attr[0, tri[t, 0]] + b1 * attr[0, tri[t, 1]] + (1 - b0 - b1) * attr[0, tri[t, 2]]
        assert np.abs(out[0, y, x] - e).max() < 1e-5
    assert (out[0][ids == 0] == 0).all()
    ones = oracle.interpolate(np.ones((1, pos.shape[0], 1), np.float32), rast, tri)
    assert np.abs(ones[0][ids > 0] - 1).max() <= 2e-7
    dy = rng.normal(size=out.shape).astype(np.float32)
    ga, gr = oracle.interpolate_grad(attr, rast, tri, dy)
    # d(sum(out*dy))/d attr by linearity
    e = np.zeros_like(attr)
    for y, x in zip(*np.nonzero(ids)):
        t = ids[y, x] - 1
        b = [rast[0, y, x, 0], rast[0, y, x, 1], 1 - rast[0, y, x, 0] - rast[0, y, x, 1]]
        for k in range(3):
            e[0, tri[t, k]] += b[k] * dy[0, y, x]
    assert np.abs(ga - e).max() < 1e-4


def test_interpolate_pixel_differentials(oracle):
    """interpolate's second output (diff_attrs): d attr / d(X, Y) per pixel.  Known answers: an attribute that equals the
    vertex's own pixel coordinate interpolates, under perspective, to the pixel coordinate itself -- its differential is
    (1, 0) for x and (0, 1) for y at every covered pixel; the neighbouring pixel's interpolated value differs by the
    differential to first order; the backward pass is the transpose of a linear map."""
    H, W = 40, 48
    pos = np.array([[-0.8, -0.7, 0.1, 1.0], [0.9, -0.6, 0.4, 1.6], [0.1, 0.8, -0.3, 0.7], [0.7, 0.9, 0.2, 1.1]], np.float32)
    tri = np.array([[0, 1, 2], [2, 1, 3]], np.int32)
    rast, db = oracle.rasterize(pos[None], tri, [H, W])
    cov = rast[0, :, :, 3] > 0
    assert cov.sum() > 300
    px = (pos[:, 0] / pos[:, 3] * 0.5 + 0.5) * W   # continuous pixel coordinates of the vertices (pixel i covers [i, i+1))
    py = (pos[:, 1] / pos[:, 3] * 0.5 + 0.5) * H
    rng = np.random.default_rng(3)
    attr = np.stack([px, py, rng.normal(size=4), rng.normal(size=4)], axis=1)[None].astype(np.float32)
    # NOTE: screen position is linear in (u, v) only for SCREEN-space barycentrics; nvdiffrast's (u, v) are perspective
    # correct, under which  attr_j = x_j  does not interpolate to x.  Use clip-space w = 1 for the affine known answer.
    pos1 = pos.copy()
    pos1[:, :3] /= pos1[:, 3:4]
    pos1[:, 3] = 1.0
    rast1, db1 = oracle.rasterize(pos1[None], tri, [H, W])
    cov1 = rast1[0, :, :, 3] > 0
    da1 = oracle.interpolate_da(attr, rast1, db1, tri, "all")
    assert da1.shape == (1, H, W, 8)
    assert np.abs(da1[0][cov1][:, 0] - 1).max() < 2e-4 and np.abs(da1[0][cov1][:, 1]).max() < 2e-4   # d x / d(X, Y)
    assert np.abs(da1[0][cov1][:, 2]).max() < 2e-4 and np.abs(da1[0][cov1][:, 3] - 1).max() < 2e-4   # d y / d(X, Y)
    assert (da1[0][~cov1] == 0).all()
    # perspective case: first-order agreement with the neighbouring pixel of the same triangle
    out = oracle.interpolate(attr, rast, tri)
    da = oracle.interpolate_da(attr, rast, db, tri, [2, 3])
    ids = rast[0, :, :, 3]
    n = 0
    for y in range(H - 1):
        for x in range(W - 1):
            if ids[y, x] > 0 and ids[y, x] == ids[y, x + 1] == ids[y + 1, x]:
                for i, j in enumerate((2, 3)):
                    assert abs((out[0, y, x + 1, j] - out[0, y, x, j]) - da[0, y, x, 2 * i]) < 2e-2 * (1 + abs(da[0, y, x, 2 * i]))
                    assert abs((out[0, y + 1, x, j] - out[0, y, x, j]) - da[0, y, x, 2 * i + 1]) < 2e-2 * (1 + abs(da[0, y, x, 2 * i + 1]))
                n += 1
    assert n > 200
    # a list selects and orders channels of 'all'
    full = oracle.interpolate_da(attr, rast, db, tri, "all")
    assert (da == full[..., [4, 5, 6, 7]]).all()
    assert (oracle.interpolate_da(attr, rast, db, tri, [3, 0]) == full[..., [6, 7, 0, 1]]).all()
    # backward = transpose: <dy, J a> == <J^T dy, a> for the map attr -> out_da and rast_db -> out_da
    dy = rng.normal(size=full.shape).astype(np.float32)
    ga, gdb = oracle.interpolate_da_grad(attr, rast, db, tri, dy, "all")
    lhs = float((dy.astype(np.float64) * full).sum())
    assert abs(lhs - float((ga.astype(np.float64) * attr).sum())) < 1e-3 * (1 + abs(lhs))
    assert abs(lhs - float((gdb.astype(np.float64) * db).sum())) < 1e-3 * (1 + abs(lhs))
    assert (gdb[0][~cov] == 0).all()


def test_rasterize_grad_matches_finite_differences(oracle):
    rng = np.random.default_rng(7)
    pos = np.array([[-0.7, -0.6, 0.1, 1.0], [0.8, -0.5, 0.2, 1.3], [0.0, 0.7, -0.1, 0.9]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    H, W = 24, 24
    rast, _ = oracle.rasterize(pos[None], tri, [H, W])
    dy = np.zeros_like(rast)
    dy[..., :2] = rng.normal(size=rast.shape[:3] + (2,))
    dy *= (rast[..., 3:4] > 0)
    g = oracle.rasterize_grad(pos[None], tri, rast, dy)

    def uv(p):
        r, _ = oracle.rasterize(p[None].astype(np.float32), tri, [H, W])
        return r[..., :2].astype(np.float64), r[..., 3] > 0

    for vi in range(3):
        for c in (0, 1, 3):
            e = 2e-3
            pp, pm = pos.astype(np.float64).copy(), pos.astype(np.float64).copy()
            pp[vi, c] += e
            pm[vi, c] -= e
            (up, cp), (um, cm) = uv(pp), uv(pm)
            keep = (cp & cm & (rast[..., 3] > 0))[..., None]  # pixels covered in all three renders
            fd = float(((up - um) * dy[..., :2] * keep).sum()) / (2 * e)
            ga = oracle.rasterize_grad(pos[None], tri, rast, dy * keep)[0, vi, c]
            assert abs(fd - ga) <= 2e-2 * max(1.0, abs(fd)), (vi, c, fd, ga)


def test_rasterize_db_grad_matches_finite_differences(oracle):
    """d(rast_db)/d(pos) (the second half of dr.rasterize's backward; forward-mode arithmetic over the nine vertex inputs in the
    oracle) against central differences of rasterize's own rast_db."""
    rng = np.random.default_rng(11)
    pos = np.array([[-0.7, -0.6, 0.1, 1.0], [0.8, -0.5, 0.2, 1.3], [0.0, 0.7, -0.1, 0.9]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    H, W = 24, 24
    rast, db = oracle.rasterize(pos[None], tri, [H, W])
    ddb = rng.normal(size=db.shape).astype(np.float32) * (rast[..., 3:4] > 0)

    def dbs(p):
        r, d = oracle.rasterize(p[None].astype(np.float32), tri, [H, W])
        return d.astype(np.float64), r[..., 3] > 0

    n = 0
    for vi in range(3):
        for c in (0, 1, 3):
            e = 2e-3
            pp, pm = pos.astype(np.float64).copy(), pos.astype(np.float64).copy()
            pp[vi, c] += e
            pm[vi, c] -= e
            (dp, cp), (dm, cm) = dbs(pp), dbs(pm)
            keep = (cp & cm & (rast[..., 3] > 0))[..., None]
            # (interior pixels only: at the clamp of the barycentrics the forward has a kink the backward ignores)
            inner = ((rast[..., 0] > 0.02) & (rast[..., 1] > 0.02) & (rast[..., 0] + rast[..., 1] < 0.98))[..., None]
            keep = keep & inner
            fd = float(((dp - dm) * ddb * keep).sum()) / (2 * e)
            ga = oracle.rasterize_grad_db(pos[None], tri, rast, (ddb * keep).astype(np.float32))[0, vi, c]
            assert abs(fd - ga) <= 2e-2 * max(1.0, abs(fd)), (vi, c, fd, ga)
            n += 1
    assert n == 9
    assert (oracle.rasterize_grad_db(pos[None], tri, rast, ddb)[0, :, 2] == 0).all()   # z carries no gradient
