"""CPU checks of the oracle's space-explorer score (oracle.mask_variance) -- the checker the GPU kernel
ehr_mask_variance is compared with: closed-form cases, the torch.var formulation of the reference
(space_explorer.py:163-164), and the committed fixture."""
import os

import numpy as np

import helpers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def quad(x0, y0, x1, y1, H, W, z=0.5):
    """Axis-aligned rectangle covering pixel columns [x0,x1) and image rows [y0,y1) (row 0 = top), as clip-space
    vertices with w = 1 (identity MVP)."""
    def ndc(px, py):
        return [2.0 * px / W - 1.0, 1.0 - 2.0 * py / H, z]
    v = np.array([ndc(x0, y0), ndc(x1, y0), ndc(x1, y1), ndc(x0, y1)], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return v, f


def test_shifted_rectangles_closed_form(oracle):
    H, W = 48, 64
    v, f = quad(10, 8, 30, 28, H, W)          # 20 x 20 pixels
    S = 3
    eye = np.eye(4, dtype=np.float32)
    mvp = np.tile(eye, (1, S, 1, 1, 1))
    for s, dx in enumerate([0, 5, 12]):        # shift right by dx pixels: NDC translation 2 dx / W
        mvp[0, s, 0, 0, 3] = 2.0 * dx / W
    score, counts = oracle.mask_variance(v, f, np.zeros(4, np.int32), mvp, H, W, return_counts=True)
    # columns covered: [10,30), [15,35), [22,42)  ->  per-column count c, 20 rows each
    c = np.zeros(W, np.int64)
    for a, b in [(10, 30), (15, 35), (22, 42)]:
        c[a:b] += 1
    assert int(score[0]) == 20 * int((c * (S - c)).sum())
    assert (counts[0, 8:28] == c.astype(np.uint8)[None]).all() and counts[0, :8].sum() == 0 and counts[0, 28:].sum() == 0
    # z/w <= 0 (closer than ~2 mm in the reference's projection) is "not mask": nvdiffrast_renderer.py:70
    v2, _ = quad(10, 8, 30, 28, H, W, z=-0.25)
    score2, counts2 = oracle.mask_variance(v2, f, np.zeros(4, np.int32), mvp, H, W, return_counts=True)
    assert score2[0] == 0 and counts2.sum() == 0


def test_equals_unbiased_variance_of_the_masks(oracle, xarm7):
    g = np.load(os.path.join(GOLD, "score_xarm7_160x120.npz"))
    H, W = int(g["H"]), int(g["W"])
    mvp = g["mvp"]
    Q, S = mvp.shape[:2]
    verts, tris, _, _ = helpers.scene_arrays(xarm7)
    vl = np.concatenate([np.full(v.shape[0], l, np.int32) for l, (v, _) in enumerate(xarm7.meshes)])
    score, counts = oracle.mask_variance(verts, tris, vl, mvp, H, W, return_counts=True)
    assert (score == g["score"]).all() and (counts == g["counts"]).all()      # committed fixture
    assert (score > 0).all()
    # the reference's formulation on explicit masks: S single-pose calls give the S binary masks
    for q in range(Q):
        masks = np.stack([oracle.mask_variance(verts, tris, vl, mvp[q:q + 1, s:s + 1], H, W, return_counts=True)[1][0]
                          for s in range(S)]).astype(np.float64)
        assert set(np.unique(masks)) <= {0.0, 1.0}
        var_sum = masks.reshape(S, -1).var(axis=0, ddof=1).sum()
        assert abs(var_sum - score[q] / (S * (S - 1.0))) < 1e-9 * var_sum
        assert (masks.sum(0) == counts[q]).all()
