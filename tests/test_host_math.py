"""Host-side mirrors of the reference's Python glue: projection, clip transform, SE(3) maps, Adam step."""
import numpy as np
import torch

from easyhec_amd.nvdiffrast_utils import K_to_projection, opencv2blender, transform_pos
from easyhec_amd.se3 import hat, se3_exp_map, se3_log_map, so3_log_numpy
import helpers


def test_K_to_projection_matches_formula():
    K = torch.tensor([[906.8, 0, 650.2], [0, 906.7, 367.7], [0, 0, 1.0]])
    P = K_to_projection(K, 720, 1280)
    exp = helpers.projection(K.numpy().astype(np.float64), 720, 1280)
    assert P.shape == (4, 4) and P.dtype == torch.float32
    assert np.abs(P.numpy() - exp).max() < 1e-6
    # known answer: camera point (x,y,z) -> NDC x = 2(fu x/z + cu)/W - 1, y = 1 - 2(fv y/z + cv)/H
    pt = np.array([0.1, -0.2, 1.5, 1.0])
    clip = (P.numpy().astype(np.float64) @ np.diag([1.0, -1, -1, 1])) @ pt
    u = 906.8 * 0.1 / 1.5 + 650.2
    v = 906.7 * -0.2 / 1.5 + 367.7
    assert abs(clip[0] / clip[3] - (2 * u / 1280 - 1)) < 1e-6
    assert abs(clip[1] / clip[3] - (1 - 2 * v / 720)) < 1e-6
    n, f = 0.001, 10.0
    assert abs(clip[2] / clip[3] - ((f + n) / (f - n) - 2 * f * n / ((f - n) * 1.5))) < 1e-6


def test_transform_pos_and_flip_matrix():
    M = torch.arange(16, dtype=torch.float32).reshape(4, 4) / 7
    v = torch.tensor([[1.0, 2.0, 3.0], [-1.0, 0.5, 0.25]])
    out = transform_pos(M, v)
    assert out.shape == (1, 2, 4)
    exp = (M.numpy() @ np.array([[1, 2, 3, 1], [-1, 0.5, 0.25, 1]]).T).T
    assert np.abs(out[0].numpy() - exp).max() < 1e-5
    o2b = opencv2blender()
    assert torch.equal(o2b @ o2b, torch.eye(4)) and torch.equal(torch.inverse(o2b), o2b)


def test_se3_exp_matches_matrix_exponential():
    from scipy.linalg import expm
    rng = np.random.default_rng(0)
    for _ in range(10):
        d = rng.normal(size=6) * 0.7
        T = se3_exp_map(torch.tensor(d[None], dtype=torch.float32))[0].numpy()
        X = np.zeros((4, 4))
        X[:3, :3] = hat(torch.tensor(d[None, 3:]))[0].numpy()
        X[:3, 3] = d[:3]
        assert np.abs(T.T - expm(X)).max() < 2e-6  # stored transposed (PyTorch3D convention)


def test_se3_log_inverts_exp_and_small_angle_clamp():
    torch.manual_seed(0)
    d = torch.randn(8, 6) * 0.9
    assert (se3_log_map(se3_exp_map(d)) - d).abs().max() < 2e-5
    # the reference initialises dof with eps=1e-5 (rb_solver.py:32); identity rotation must not blow up
    T = torch.eye(4)[None].clone()
    T[0, 3, :3] = torch.tensor([0.1, -0.2, 0.3])
    d0 = se3_log_map(T, eps=1e-5)
    assert torch.isfinite(d0).all() and (d0[0, 3:].abs() < 1e-6).all() and (d0[0, :3] - T[0, 3, :3]).abs().max() < 1e-5
    # gradient flows through exp at zero rotation thanks to the squared-angle clamp (pytorch3d_se3.py:26)
    z = torch.zeros(1, 6, requires_grad=True)
    se3_exp_map(z).sum().backward()
    assert torch.isfinite(z.grad).all()


def test_so3_log_numpy_against_scipy():
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(1)
    for ang in [1e-7, 1e-3, 0.5, 2.0, 3.0, 3.1415]:
        ax = rng.normal(size=3)
        rv = ax / np.linalg.norm(ax) * ang
        got = so3_log_numpy(R.from_rotvec(rv).as_matrix())
        assert np.abs(R.from_rotvec(got).as_matrix() - R.from_rotvec(rv).as_matrix()).max() < 1e-5


def test_se3_log_self_check_raises_on_garbage():
    import pytest
    bad = torch.eye(4)[None] * 3.0
    with pytest.raises(RuntimeError):
        se3_log_map(bad)


def test_optimizer_is_adam_with_l2_weight_decay():
    """solver/build.py:12-29 + defaults.py:138: Adam(lr, weight_decay=5e-4 as L2 on the gradient)."""
    from easyhec_amd.config import Cfg
    from easyhec_amd.trainer import make_optimizer
    m = torch.nn.Module()
    m.dof = torch.nn.Parameter(torch.tensor([0.1, -0.2, 0.3, 0.01, 0.02, -0.03]))
    opt = make_optimizer(Cfg(), m)
    assert isinstance(opt, torch.optim.Adam)
    g = opt.param_groups[0]
    assert g["lr"] == 0.003 and g["weight_decay"] == 0.0005 and g["betas"] == (0.9, 0.999) and g["eps"] == 1e-8
    p0 = m.dof.detach().clone()
    m.dof.grad = torch.tensor([1.0, -2.0, 0.5, 0.0, 3.0, -1.0])
    opt.step()
    geff = m.dof.grad + 5e-4 * p0
    exp = p0 - 0.003 * geff / (geff.abs() + 1e-8)  # first Adam step = lr * g / (|g| + eps)
    assert (m.dof.detach() - exp).abs().max() < 1e-6


def test_shard_views_partitions_contiguously():
    from easyhec_amd.trainer import shard_views
    for n, w in [(64, 8), (10, 4), (3, 8), (8, 1)]:
        parts = [shard_views(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
    assert [shard_views(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
