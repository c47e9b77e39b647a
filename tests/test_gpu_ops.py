"""GPU parity of the three drop-in ops, called through the C ABI (easyhec_amd.dr -> libehr_hip.so), against the CPU
oracle on the same seeded inputs.  Integer work (coverage, triangle ids, topology) is bit-exact; float outputs of
single-threaded-order kernels are bit-exact too -- including dr.antialias's forward pass, which gathers every pixel's
blends in the serial sweep's order; the atomically accumulated gradients carry the stated tolerance."""
import os

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from easyhec_amd import _lib, dr
    assert os.path.exists(_lib.LIB_PATH)
    return dr, dr.RasterizeCudaContext(), torch.device("cuda:0")


@pytest.fixture(params=["direct", "queued"])
def raster_path(request):
    """Both forms of the drop-in rasterizer (ehr_raster.hip): the direct one (small launches: two kernels, a key image in
    global memory) and the queued one (count / allocate / fill / one workgroup per tile).  Same bits either way."""
    old = os.environ.get("EHR_RASTER_DIRECT_MAX")
    os.environ["EHR_RASTER_DIRECT_MAX"] = "0" if request.param == "queued" else "1000000000"
    yield request.param
    if old is None:
        del os.environ["EHR_RASTER_DIRECT_MAX"]
    else:
        os.environ["EHR_RASTER_DIRECT_MAX"] = old


def t(a, dev, grad=False):
    x = torch.tensor(np.ascontiguousarray(a), device=dev)
    if grad:
        x.requires_grad_(True)
    return x


@pytest.mark.parametrize("H,W,n,shared", [(64, 64, 50, True), (72, 104, 400, True), (250, 333, 3000, True),
                                          (128, 160, 40, False), (8, 8, 5, True), (720, 1280, 20000, True)])
def test_rasterize_interpolate_antialias_parity(env, oracle, raster_path, H, W, n, shared):
    dr, ctx, dev = env
    rng = np.random.default_rng(H * 1000 + n)
    pos, tri = helpers.random_mesh(rng, n, shared=shared, size=0.6 if not shared else 0.25)
    r_ref, db_ref = oracle.rasterize(pos[None], tri, [H, W])
    tp, tt = t(pos[None], dev, True), t(tri, dev)
    r, db = dr.rasterize(ctx, tp, tt, [H, W])
    assert (r.detach().cpu().numpy() == r_ref).all()       # ids, barycentrics, depth: bit-exact
    assert (db.detach().cpu().numpy() == db_ref).all()
    attr = rng.uniform(0, 1, size=(1, pos.shape[0], 3)).astype(np.float32)
    ta = t(attr, dev, True)
    c, da = dr.interpolate(ta, r, tt)
    c_ref = oracle.interpolate(attr, r_ref, tri)
    assert (c.detach().cpu().numpy() == c_ref).all() and da.shape[-1] == 0
    th = dr.antialias_construct_topology_hash(tt)
    assert (th.opp.cpu().numpy() == oracle.topology(tri)).all()
    aa = dr.antialias(c, r, tp, tt, topology_hash=th)
    aa_ref = oracle.antialias(c_ref, r_ref, pos[None], tri)
    assert (aa.detach().cpu().numpy() == aa_ref).all()   # bit-exact: every pixel gathers its blends in the serial sweep's order
    aa2 = dr.antialias(c, r, tp, tt)                      # topology rebuilt per call, as the reference
    assert (aa2.detach().cpu().numpy() == aa_ref).all()
    gy = rng.normal(size=aa_ref.shape).astype(np.float32)
    (aa * t(gy, dev)).sum().backward()
    gc_ref, gp_ref = oracle.antialias_grad(c_ref, r_ref, pos[None], tri, gy)
    ga_ref, gr_ref = oracle.interpolate_grad(attr, r_ref, tri, gc_ref)
    gp_ref = gp_ref + oracle.rasterize_grad(pos[None], tri, r_ref, gr_ref)
    assert np.abs(ta.grad.cpu().numpy() - ga_ref).max() <= 1e-5 * max(1.0, np.abs(ga_ref).max())
    assert np.abs(tp.grad.cpu().numpy() - gp_ref).max() <= 1e-5 * max(1.0, np.abs(gp_ref).max())


def test_ops_golden_fixture(env):
    dr, ctx, dev = env
    g = np.load(os.path.join(GOLD, "ops_random_72x104.npz"))
    H, W = g["rast"].shape[1:3]
    tp, tt, ta = t(g["pos"][None], dev, True), t(g["tri"], dev), t(g["attr"], dev, True)
    r, db = dr.rasterize(ctx, tp, tt, [H, W])
    assert (r.detach().cpu().numpy() == g["rast"]).all() and (db.detach().cpu().numpy() == g["db"]).all()
    c, _ = dr.interpolate(ta, r, tt)
    assert (c.detach().cpu().numpy() == g["col"]).all()
    aa = dr.antialias(c, r, tp, tt)
    assert (aa.detach().cpu().numpy() == g["aa"]).all()
    (aa * t(g["dy"], dev)).sum().backward()
    assert np.abs(ta.grad.cpu().numpy() - g["grad_attr"]).max() <= 1e-5 * np.abs(g["grad_attr"]).max()
    assert np.abs(tp.grad.cpu().numpy()[0] - g["grad_pos"][0]).max() <= 1e-5 * np.abs(g["grad_pos"]).max()


def test_rasterize_survives_a_frame_that_outgrows_the_queue_storage():
    """the drop-in rasterizer skips its size read-back when the previous frames of the same shape
    needed little; a frame that then needs more entries than the storage holds must still come out exact (the tiles whose
    queues did not fit find their triangles themselves), never NaN or incomplete, and the storage grows afterwards."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env2 = dict(os.environ, EHR_RASTER_MIN_ENTRIES="64", EHR_RASTER_DIRECT_MAX="0")  # (the queued form is the one with storage)
    out = subprocess.run([sys.executable, os.path.join(here, "raster_growth_worker.py")], env=env2, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert len(res) == 7
    for r in res:
        assert r["rast_equal"] and r["db_equal"] and not r["nan"], r
    assert res[4]["covered"] > 10 * res[0]["covered"]  # the zoomed frames really are much larger


def test_range_mode_and_batches(env, oracle, raster_path):
    dr, ctx, dev = env
    rng = np.random.default_rng(9)
    pos, tri = helpers.random_mesh(rng, 120)
    H, W = 48, 80
    ranges = np.array([[0, 120], [10, 50], [119, 1], [0, 0]], np.int32)
    ref, _ = oracle.rasterize(pos, tri, [H, W], ranges=ranges)
    r, _ = dr.rasterize(ctx, t(pos, dev), t(tri, dev), [H, W], ranges=torch.tensor(ranges))
    assert (r.cpu().numpy() == ref).all() and (r[3] == 0).all()
    # instance mode with B = 3 different vertex sets
    posb = np.stack([pos, pos * np.array([1, -1, 1, 1], np.float32), pos[::-1].copy()])
    refb, dbb = oracle.rasterize(posb, tri, [H, W])
    rb, db = dr.rasterize(ctx, t(posb, dev), t(tri, dev), [H, W])
    assert (rb.cpu().numpy() == refb).all() and (db.detach().cpu().numpy() == dbb).all()
    attr = rng.uniform(size=(3, pos.shape[0], 2)).astype(np.float32)
    c, _ = dr.interpolate(t(attr, dev), rb, t(tri, dev))
    assert (c.cpu().numpy() == oracle.interpolate(attr, refb, tri)).all()
    aa = dr.antialias(c, rb, t(posb, dev), t(tri, dev))
    assert (aa.cpu().numpy() == oracle.antialias(c.cpu().numpy(), refb, posb, tri)).all()


def test_rasterize_calls_of_changing_shape_on_one_context(env, oracle, raster_path):
    """A drop-in rasterize call starts without a fill kernel: its last kernel leaves the context's queue counters zero
    for the next call.  Calls of different resolution, batch size and mode interleaved on ONE context must each equal the
    oracle -- a counter word left dirty by one layout would corrupt the next (the per-image ranges of a range-mode call
    used to live behind the counters, where a larger layout's counters land)."""
    dr = env[0]
    dev = env[2]
    ctx = dr.RasterizeCudaContext()
    rng = np.random.default_rng(21)
    pos, tri = helpers.random_mesh(rng, 300)
    posb = np.stack([pos, pos * np.array([-1, 1, 1, 1], np.float32)])
    ranges = np.array([[0, 300], [17, 200], [299, 1]], np.int32)
    plan = [("inst", (40, 56)), ("range", (24, 40)), ("inst", (200, 312)), ("range", (96, 96)), ("inst", (8, 8)),
            ("inst", (200, 312)), ("range", (24, 40)), ("inst", (40, 56))]
    for kind, (H, W) in plan * 2:
        if kind == "inst":
            ref, _ = oracle.rasterize(posb, tri, [H, W])
            r, _ = dr.rasterize(ctx, t(posb, dev), t(tri, dev), [H, W])
        else:
            ref, _ = oracle.rasterize(pos, tri, [H, W], ranges=ranges)
            r, _ = dr.rasterize(ctx, t(pos, dev), t(tri, dev), [H, W], ranges=torch.tensor(ranges))
        assert (r.cpu().numpy() == ref).all(), (kind, H, W)


@pytest.mark.parametrize("H,W,seed", [(720, 1280, 0), (333, 517, 1), (64, 2000, 2), (1100, 90, 3)])
def test_rasterizer_forms_on_slivers_big_and_clipped_triangles(env, oracle, H, W, seed):
    """What the direct form's work distribution has to get right: long thin triangles (boxes far larger than their
    coverage), triangles that fill the screen (spread over the bands of rows), triangles crossing the eye plane (clipped
    into two), many micro-triangles -- all in one mesh, several overlapping in depth.  Both forms == the oracle, bit for bit
    (ids, barycentrics, depth, pixel differentials), in instance mode with two images."""
    dr, dev = env[0], env[2]
    ctx = dr.RasterizeCudaContext()
    rng = np.random.default_rng(100 + seed)
    parts = []
    pos_small, tri_small = helpers.random_mesh(rng, 1500)
    parts.append((pos_small, tri_small))
    # slivers: two close points and one far away
    sl = []
    for _ in range(300):
        a = rng.uniform(-1.1, 1.1, size=2)
        b = a + rng.uniform(-0.01, 0.01, size=2)
        c = rng.uniform(-1.3, 1.3, size=2)
        for xy in (a, b, c):
            w = rng.uniform(0.7, 1.5)
            sl.append([xy[0] * w, xy[1] * w, rng.uniform(-0.6, 0.6) * w, w])
    parts.append((np.array(sl, np.float32), np.arange(900, dtype=np.int32).reshape(-1, 3)))
    # screen-filling triangles at several depths, and triangles crossing the eye plane / the near plane
    big = [[-3, -3, 0.7, 1], [3, -3, 0.7, 1], [0, 3, 0.7, 1], [-2.5, 2, 0.3, 1], [2.5, 2, 0.3, 1], [0, -4, 0.9, 1],
           [-0.8, -0.8, 0.1, 1.0], [0.8, -0.7, 0.1, 1.0], [0.1, 2.0, -1.5, -0.4],
           [-1.5, 0.2, -0.9, 0.5], [1.5, 0.3, 0.4, 1.2], [0.0, -1.2, 0.2, 0.9]]
    parts.append((np.array(big, np.float32), np.arange(12, dtype=np.int32).reshape(-1, 3)))
    pos, tri, off = [], [], 0
    for p_, t_ in parts:
        pos.append(p_)
        tri.append(t_ + off)
        off += p_.shape[0]
    pos, tri = np.concatenate(pos), np.concatenate(tri)
    tri = tri[rng.permutation(tri.shape[0])]
    posb = np.stack([pos, pos * np.array([-1, 1, 1, 1], np.float32)])
    ref, dbr = oracle.rasterize(posb, tri, [H, W])
    assert (ref[..., 3] > 0).mean() > 0.5
    old = os.environ.get("EHR_RASTER_DIRECT_MAX")
    try:
        for form in ("1000000000", "0"):
            os.environ["EHR_RASTER_DIRECT_MAX"] = form
            r, db = dr.rasterize(ctx, t(posb, dev), t(tri, dev), [H, W])
            assert (r.cpu().numpy() == ref).all() and (db.detach().cpu().numpy() == dbr).all(), (form, H, W)
    finally:
        if old is None:
            os.environ.pop("EHR_RASTER_DIRECT_MAX", None)
        else:
            os.environ["EHR_RASTER_DIRECT_MAX"] = old


def test_interpolate_pixel_differentials_match_the_oracle(env, oracle):
    """dr.interpolate(attr, rast, tri, rast_db=, diff_attrs=): the attribute pixel differentials (not on EasyHeC's path;
    they complete the op's nvdiffrast signature).  Forward bit-exact against the oracle for 'all' and for an index list,
    gradients w.r.t. the attributes (through BOTH outputs) and rast_db within the atomics' tolerance."""
    dr, ctx, dev = env
    rng = np.random.default_rng(17)
    pos, tri = helpers.random_mesh(rng, 300)
    H, W = 72, 104
    r_ref, db_ref = oracle.rasterize(pos[None], tri, [H, W])
    attr = rng.normal(size=(1, pos.shape[0], 4)).astype(np.float32)
    tp, tt = t(pos[None], dev), t(tri, dev)
    r, db = dr.rasterize(ctx, tp, tt, [H, W])
    for sel in ("all", [2, 0], [1]):
        ta = t(attr, dev, True)
        tdb = db.clone().requires_grad_(True)
        out, da = dr.interpolate(ta, r, tt, rast_db=tdb, diff_attrs=sel)
        da_ref = oracle.interpolate_da(attr, r_ref, db_ref, tri, sel)
        assert da.shape == da_ref.shape and (da.detach().cpu().numpy() == da_ref).all(), sel
        assert (out.detach().cpu().numpy() == oracle.interpolate(attr, r_ref, tri)).all()
        gy, gda = rng.normal(size=out.shape).astype(np.float32), rng.normal(size=da_ref.shape).astype(np.float32)
        ((out * t(gy, dev)).sum() + (da * t(gda, dev)).sum()).backward()
        ga_ref, _ = oracle.interpolate_grad(attr, r_ref, tri, gy)
        ga2_ref, gdb_ref = oracle.interpolate_da_grad(attr, r_ref, db_ref, tri, gda, sel)
        ga_ref = ga_ref + ga2_ref
        assert np.abs(ta.grad.cpu().numpy() - ga_ref).max() <= 1e-5 * max(1.0, np.abs(ga_ref).max()), sel
        assert np.abs(tdb.grad.cpu().numpy() - gdb_ref).max() <= 1e-6 * max(1.0, np.abs(gdb_ref).max()), sel
    with pytest.raises(RuntimeError):
        dr.interpolate(t(attr, dev), r, tt, diff_attrs="all")           # needs rast_db
    with pytest.raises(RuntimeError):
        dr.interpolate(t(attr, dev), r, tt, rast_db=db, diff_attrs=[7])  # index out of range
    _, empty = dr.interpolate(t(attr, dev), r, tt, rast_db=db)           # rast_db alone: no differentials asked for
    assert empty.shape[-1] == 0


def test_rasterize_backward_through_rast_db(env, oracle, raster_path):
    """dr.rasterize's second output is differentiable too: d(rast_db)/d(pos) against the oracle's forward-mode
    restatement (itself pinned by finite differences, tests/test_oracle_raster.py), alone and together with the (u, v) half."""
    dr, ctx, dev = env
    rng = np.random.default_rng(23)
    pos, tri = helpers.random_mesh(rng, 200)
    H, W = 56, 88
    r_ref, db_ref = oracle.rasterize(pos[None], tri, [H, W])
    gy = np.zeros_like(r_ref)
    gy[..., :2] = rng.normal(size=r_ref.shape[:3] + (2,))
    gdb = rng.normal(size=db_ref.shape).astype(np.float32)
    g_uv = oracle.rasterize_grad(pos[None], tri, r_ref, gy)
    g_db = oracle.rasterize_grad_db(pos[None], tri, r_ref, gdb)
    tp = t(pos[None], dev, True)
    r, db = dr.rasterize(ctx, tp, t(tri, dev), [H, W])
    assert db.requires_grad
    (db * t(gdb, dev)).sum().backward(retain_graph=True)
    got = tp.grad.cpu().numpy().copy()
    assert np.abs(got - g_db).max() <= 1e-4 * max(1.0, np.abs(g_db).max())
    assert (got[..., 2] == 0).all()
    tp.grad = None
    ((r * t(gy, dev)).sum() + (db * t(gdb, dev)).sum()).backward()
    both = tp.grad.cpu().numpy()
    assert np.abs(both - (g_uv + g_db)).max() <= 1e-4 * max(1.0, np.abs(g_uv + g_db).max())
    _, db0 = dr.rasterize(ctx, t(pos[None], dev, True), t(tri, dev), [H, W], grad_db=False)
    assert db0.shape[-1] == 0 and not db0.requires_grad


def test_rasterizer_forms_alternate_on_one_context(env, oracle):
    """The two forms keep separate state on a context (queue counters all zero / key image all ones between calls);
    alternating them call by call must not disturb either."""
    dr, dev = env[0], env[2]
    ctx = dr.RasterizeCudaContext()
    rng = np.random.default_rng(5)
    pos, tri = helpers.random_mesh(rng, 500)
    old = os.environ.get("EHR_RASTER_DIRECT_MAX")
    try:
        for i, (H, W) in enumerate([(64, 96), (64, 96), (120, 200), (64, 96), (120, 200), (120, 200)]):
            os.environ["EHR_RASTER_DIRECT_MAX"] = "0" if i % 2 else "1000000000"
            ref, dbr = oracle.rasterize(pos[None], tri, [H, W])
            r, db = dr.rasterize(ctx, t(pos[None], dev), t(tri, dev), [H, W])
            assert (r.cpu().numpy() == ref).all() and (db.detach().cpu().numpy() == dbr).all(), (i, H, W)
    finally:
        if old is None:
            os.environ.pop("EHR_RASTER_DIRECT_MAX", None)
        else:
            os.environ["EHR_RASTER_DIRECT_MAX"] = old


def test_edge_cases(env, oracle, raster_path):
    dr, ctx, dev = env
    H, W = 40, 72
    # clipped / behind-camera / far / degenerate / NaN / out-of-range indices, plus one triangle covering many tiles
    pos = np.array([[-0.5, -0.5, 0.2, 1.0], [0.5, -0.5, 0.2, 1.0], [0.0, 3.0, -2.0, -0.5],   # crosses the eye plane
                    [-3, -3, 0.5, 1], [3, -3, 0.5, 1], [0, 3, 0.5, 1],                          # covers the screen
                    [0, 0, 0, 1], [0.5, 0.5, 0, 1], [1, 1, 0, 1],                               # collinear
                    [np.nan, 0, 0, 1], [0.2, 0.1, 1.5, 1.0], [0.3, 0.4, 1.5, 1.0], [0.1, 0.5, 1.5, 1]], np.float32)
    tri = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8], [0, 1, 9], [10, 11, 12], [0, 1, 99], [0, 0, 1]], np.int32)
    ref, dbr = oracle.rasterize(pos[None], tri, [H, W])
    r, db = dr.rasterize(ctx, t(pos[None], dev), t(tri, dev), [H, W])
    assert (r.cpu().numpy() == ref).all() and (db.detach().cpu().numpy() == dbr).all()
    assert (ref[..., 3] > 0).mean() > 0.9
    col = (ref[..., 3:4] > 0).astype(np.float32)
    aa = dr.antialias(t(col, dev), r, t(pos[None], dev), t(tri, dev))
    assert (aa.cpu().numpy() == oracle.antialias(col, ref, pos[None], tri)).all()
    # empty triangle list -> all-zero image
    r0, _ = dr.rasterize(ctx, t(pos[None], dev), torch.zeros((0, 3), dtype=torch.int32, device=dev), [H, W])
    assert (r0 == 0).all()
    # argument validation raises like nvdiffrast (RuntimeError), CPU tensors are refused
    with pytest.raises(RuntimeError):
        dr.rasterize(ctx, torch.zeros(1, 3, 4), t(tri, dev), [H, W])
    with pytest.raises(RuntimeError):
        dr.rasterize(ctx, t(pos[None], dev), t(tri, dev).long(), [H, W])
    with pytest.raises(RuntimeError):
        dr.rasterize(ctx, t(pos, dev), t(tri, dev), [H, W])  # instance mode needs [B,V,4]


def test_renderer_wrapper_matches_reference_semantics(env, oracle, xarm7):
    """NVDiffrastRenderer.render_mask == oracle pipeline incl. the final flip (nvdiffrast_renderer.py:33-47)."""
    _, _, dev = env
    from easyhec_amd.config import XARM7_K_1280x720
    from easyhec_amd.renderer import NVDiffrastRenderer
    from easyhec_amd.synthetic import camera_Tc_c2b, make_views, scaled_K
    H, W = 240, 320
    K = scaled_K(XARM7_K_1280x720, 0.25, W, H, True)
    _, lp = make_views(xarm7, 1, seed=2)
    Tc = camera_Tc_c2b()
    ren = NVDiffrastRenderer([H, W])
    v, f = xarm7.meshes[3]
    pose = torch.tensor(Tc @ lp[0, 3].astype(np.float64), dtype=torch.float32, device=dev)
    tv, tf = t(v, dev), t(f, dev)
    m = ren.render_mask(tv, tf, torch.tensor(K, dtype=torch.float32, device=dev), pose)
    assert m.shape == (H, W) and m.dtype == torch.float32
    # oracle on the SAME clip positions the wrapper produced
    from easyhec_amd.nvdiffrast_utils import K_to_projection, transform_pos
    pos = transform_pos(K_to_projection(torch.tensor(K, dtype=torch.float32, device=dev), H, W) @
                        (ren.opencv2blender @ pose), tv).cpu().numpy()
    rast, _ = oracle.rasterize(pos, f, [H, W])
    col = oracle.interpolate(np.ones((1, v.shape[0], 3), np.float32), rast, f)
    exp = oracle.antialias(col, rast, pos, f)[0, ::-1, :, 0]
    assert np.abs(m.cpu().numpy() - exp).max() <= 5e-7
    hard = ren.render_mask(tv, tf, torch.tensor(K, dtype=torch.float32, device=dev), pose, anti_aliasing=False)
    assert hard.dtype == torch.bool and (hard.cpu().numpy() == (rast[0, ::-1, :, 2] > 0)).all()


def test_tile_flags_follow_the_rasterizer_and_change_no_result(env, oracle):
    """ABI 6: dr.rasterize attaches one byte per (image, 32 x 8 tile) to the `rast` it returns -- non-zero iff a pixel of the tile
    holds a triangle (direct form; the queued form sets them all) -- and dr.interpolate / dr.antialias / the backward passes skip
    `rast` where it is zero.  The results with the flags equal the results without them (a `rast` whose attribute was removed), bit
    for bit where the kernels gather, within the atomics' tolerance for the position gradient."""
    dr, _, dev = env
    ctx = dr.RasterizeCudaContext()
    H, W = 200, 360
    rng = np.random.default_rng(77)
    pos, tri = helpers.random_mesh(rng, 300, shared=True, size=0.12)   # a small object: most tiles stay empty
    pos[:, 0] = pos[:, 0] * 0.35 + 0.4 * pos[:, 3]
    pos[:, 1] = pos[:, 1] * 0.35 - 0.3 * pos[:, 3]
    tt = t(tri, dev)
    attr = t(rng.uniform(0, 1, size=(1, pos.shape[0], 3)).astype(np.float32), dev)
    outs = []
    for with_flags in (True, False):
        tp = t(pos[None], dev, True)
        r, _ = dr.rasterize(ctx, tp, tt, [H, W])
        flags = getattr(r, "_ehr_tile_flags", None)
        assert flags is not None and flags.dtype == torch.uint8
        if with_flags:
            ntx, nty = (W + 31) // 32, (H + 7) // 8
            f = flags[: ntx * nty].cpu().numpy().reshape(nty, ntx) != 0
            ids = r.detach().cpu().numpy()[0, :, :, 3]
            want = np.zeros((nty, ntx), dtype=bool)
            for ty in range(nty):
                for tx in range(ntx):
                    want[ty, tx] = (ids[ty * 8:(ty + 1) * 8, tx * 32:(tx + 1) * 32] != 0).any()
            assert (f == want).all() and 0 < want.sum() < want.size // 2
            rr = r
        else:
            assert getattr(r.detach(), "_ehr_tile_flags", None) is None   # (a re-wrapped tensor carries none ...)
            delattr(r, "_ehr_tile_flags")                                  # ... and this one shall not either
            rr = r
        c, _ = dr.interpolate(attr, rr, tt)
        a = dr.antialias(c, rr, tp, tt)
        (a * a).sum().backward()
        outs.append((c.detach().clone(), a.detach().clone(), tp.grad.detach().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    g0, g1 = outs[0][2], outs[1][2]
    assert float((g0 - g1).abs().max()) <= 1e-5 * max(1.0, float(g1.abs().max())) and float(g1.abs().max()) > 0
    c_ref = oracle.interpolate(attr.cpu().numpy(), oracle.rasterize(pos[None], tri, [H, W])[0], tri)
    assert (outs[0][0].cpu().numpy() == c_ref).all()


def test_tile_flags_are_dropped_when_the_rasterizer_output_is_edited_or_swapped(env):
    """the flags ride on the `rast` tensor object; they are only believed while that tensor still is the
    rasterizer's output -- same storage, same version.  An in-place edit (here: a triangle id written into an empty tile) or
    flags carried onto another tensor must make interpolate look at every pixel again."""
    dr, _, dev = env
    ctx = dr.RasterizeCudaContext()
    H, W = 64, 128
    rng = np.random.default_rng(3)
    pos, tri = helpers.random_mesh(rng, 60, shared=True, size=0.1)
    pos[:, 0] = pos[:, 0] * 0.3 - 0.5 * pos[:, 3]
    tp, tt = t(pos[None], dev), t(tri, dev)
    attr = torch.ones((1, pos.shape[0], 1), device=dev)
    r, _ = dr.rasterize(ctx, tp, tt, [H, W], grad_db=False)
    assert dr._flags_of(r) is not None and dr._flags_of(dr.carry_tile_flags(r, r.detach())) is not None
    other = r.clone()
    assert dr._flags_of(dr.carry_tile_flags(r, other)) is None           # another storage: not carried
    empty = (r[0, :, :, 3] == 0).nonzero()
    y, x = int(empty[-1, 0]), int(empty[-1, 1])                          # a pixel of a tile nothing was drawn in
    r[0, y, x] = torch.tensor([0.25, 0.25, 0.5, 1.0], device=dev)        # triangle 0, edited in place
    assert dr._flags_of(r) is None
    c, _ = dr.interpolate(attr, r, tt)
    assert float(c[0, y, x, 0]) == 1.0                                   # the edited pixel is shaded, not skipped


def test_a_context_whose_calls_were_captured_refuses_to_move_its_scratch():
    """a torch.cuda.CUDAGraph that recorded dr.rasterize holds the context's scratch pointers.  A later call
    that would have to GROW that scratch must fail loudly instead of freeing what the graph's replays write into."""
    from easyhec_amd import dr
    dev = torch.device("cuda:0")
    ctx = dr.RasterizeCudaContext()
    rng = np.random.default_rng(5)
    pos, tri = helpers.random_mesh(rng, 200, shared=True, size=0.3)
    tp, tt = t(pos[None], dev), t(tri, dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            dr.rasterize(ctx, tp, tt, [64, 96], grad_db=False)   # warm-up: sizes the scratch
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        r, _ = dr.rasterize(ctx, tp, tt, [64, 96], grad_db=False)
    g.replay()
    torch.cuda.synchronize()
    first = r.clone()
    dr.rasterize(ctx, tp, tt, [64, 96], grad_db=False)            # same shape, eager: fine
    with pytest.raises(RuntimeError, match="captured graph"):
        dr.rasterize(ctx, tp, tt, [512, 768], grad_db=False)      # would move the key image
    g.replay()                                                    # ... and the graph still replays into live memory
    torch.cuda.synchronize()
    assert torch.equal(r, first)


@pytest.mark.parametrize("range_mode", [False, True])
def test_antialias_gradient_buffer_cleared_by_the_forward_pass(env, oracle, range_mode):
    """dr.antialias' forward kernel clears the buffer its backward pass accumulates pos's gradient into
    (``ehr_antialias_fwd_zg``: no fill launch per call).  The gradient equals the oracle's; a second backward() through a
    retained graph (whose buffer the first one handed to autograd) gives the same gradient again; under no_grad no buffer is
    made; and a buffer that held garbage before the forward pass is clean after it (pointer-level call)."""
    import ctypes
    from easyhec_amd import _lib
    dr, ctx, dev = env
    H, W = 96, 136
    rng = np.random.default_rng(77)
    pos, tri = helpers.random_mesh(rng, 300, shared=True, size=0.3)
    nimg = 3
    if range_mode:
        V, T = pos.shape[0], tri.shape[0]
        posb = np.concatenate([pos + np.float32(0.01 * i) * np.array([1, 1, 0, 0], np.float32) for i in range(nimg)])
        trib = np.concatenate([tri + i * V for i in range(nimg)]).astype(np.int32)
        ranges = torch.tensor([[i * T, T] for i in range(nimg)], dtype=torch.int32)
        tp, tt = t(posb, dev, True), t(trib, dev)
        r, _ = dr.rasterize(ctx, tp, tt, [H, W], ranges=ranges)
    else:
        posb = np.stack([pos + np.float32(0.01 * i) * np.array([1, 1, 0, 0], np.float32) for i in range(nimg)])
        tp, tt = t(posb, dev, True), t(tri, dev)
        r, _ = dr.rasterize(ctx, tp, tt, [H, W])
    col = t(rng.uniform(0, 1, size=((posb.shape[0], 2) if range_mode else (nimg, posb.shape[1], 2))).astype(np.float32), dev)
    c, _ = dr.interpolate(col, r.detach(), tt)
    th = dr.antialias_construct_topology_hash(tt)
    aa = dr.antialias(c, r.detach(), tp, tt, topology_hash=th)
    gy = t(rng.normal(size=tuple(aa.shape)).astype(np.float32), dev)
    (g1,) = torch.autograd.grad((aa * gy).sum(), tp, retain_graph=True)
    (g2,) = torch.autograd.grad((aa * gy).sum(), tp)
    _, gp_ref = oracle.antialias_grad(c.cpu().numpy(), r.detach().cpu().numpy(), posb, tri if not range_mode else trib, gy.cpu().numpy())
    scale = max(1.0, float(np.abs(gp_ref).max()))
    assert np.abs(g1.cpu().numpy() - gp_ref).max() <= 1e-5 * scale
    assert (g1 - g2).abs().max().item() <= 1e-5 * scale and g1.data_ptr() != g2.data_ptr()
    with torch.no_grad():
        aa_ng = dr.antialias(c, r.detach(), tp, tt, topology_hash=th)
    assert torch.equal(aa_ng, aa.detach())
    # pointer level: garbage in, zeros out, same image
    lib = _lib.lib()
    B = r.shape[0]
    Vn, Tn, C = tp.shape[-2], tt.shape[0], c.shape[-1]
    out = torch.empty_like(c)
    work = torch.empty((lib.ehr_antialias_work_bytes(B, H, W),), dtype=torch.uint8, device=dev)
    dirty = torch.full_like(tp.detach(), float("nan"))
    _lib.check(lib.ehr_antialias_fwd_zg(_lib.ptr(c), _lib.ptr(r.detach()), _lib.ptr(tp.detach()), _lib.ptr(tt), _lib.ptr(th.opp),
                                        int(range_mode), B, Vn, Tn, H, W, C, _lib.ptr(out), _lib.ptr(work), None,
                                        _lib.ptr(dirty), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "aa")
    torch.cuda.synchronize()
    assert torch.equal(out, aa.detach()) and (dirty == 0).all()
