"""Host logic of the optimisation loop on CPU (oracle-rendered stand-in model): convergence, checkpoint layout, and
the world_size-2 data-parallel exchange over gloo."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from easyhec_amd.config import XARM7_K_1280x720, Cfg
from easyhec_amd.synthetic import camera_Tc_c2b, make_views, perturb_pose, scaled_K
from easyhec_amd.trainer import RBSolverTrainer, shard_views

H, W = 96, 128


class Boxes:
    def __init__(self, robot):
        self.meshes = [helpers.box_mesh(v) for v, _ in robot.meshes]


def make_problem(robot, n_views, lo=0, hi=None):
    from oracle import oracle
    from oracle_backend import OracleRBSolver
    hi = n_views if hi is None else hi
    K = scaled_K(XARM7_K_1280x720, 0.1, W, H, True)
    _, lp = make_views(robot, n_views, seed=0)
    Tc = camera_Tc_c2b()
    boxes = Boxes(robot)
    verts, tris, toff, voff = helpers.scene_arrays(boxes)
    ref = (oracle.render_mask_loss(verts, tris, toff, voff, helpers.mvp_numpy(K, H, W, Tc, lp),
                                   np.zeros((n_views, H, W), np.float32), want_grad=False)[0] > 0.5).astype(np.float32)
    model = OracleRBSolver(boxes, perturb_pose(Tc), H, W)
    batch = {"mask": torch.from_numpy(ref[lo:hi]), "link_poses": torch.from_numpy(lp[lo:hi]),
             "K": torch.tensor(K, dtype=torch.float32)[None].repeat(hi - lo, 1, 1),
             "Tc_c2b": torch.tensor(Tc, dtype=torch.float32)[None].repeat(hi - lo, 1, 1)}
    return model, batch, Tc


def test_adam_loop_reduces_loss_and_pose_error(xarm7, tmp_path):
    model, batch, Tc = make_problem(xarm7, 2)
    cfg = Cfg()
    tr = RBSolverTrainer(cfg, model, batch)
    l0 = float(tr.step()[1])
    for _ in range(60):
        _, l = tr.step()
    assert float(l) < 0.5 * l0
    T = model.dof.detach()
    from easyhec_amd.se3 import se3_exp_map
    est = se3_exp_map(T[None]).permute(0, 2, 1)[0].numpy()
    err0 = np.linalg.norm(perturb_pose(Tc)[:3, 3] - Tc[:3, 3])
    assert np.linalg.norm(est[:3, 3] - Tc[:3, 3]) < err0
    # checkpoint layout consumed by the reference's tools (tools/validate.py:28, space_explorer.py:35)
    path = str(tmp_path / "model_iteration_000061.pth")
    tr.save(path)
    ck = torch.load(path, weights_only=False)
    assert set(["model", "epoch", "global_steps"]) <= set(ck) and ck["model"]["dof"].shape == (6,)
    assert ck["global_steps"] == 61


def _dp_worker(rank, world, port, n_views, steps, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from easyhec_amd.robot import load_robot
    robot = load_robot("xarm7")
    lo, hi = shard_views(n_views, rank, world)
    model, batch, _ = make_problem(robot, n_views, lo, hi)
    tr = RBSolverTrainer(Cfg(), model, batch)
    assert tr.distributed
    losses = [float(tr.step()[1]) for _ in range(steps)]
    if rank == 0:
        torch.save({"dof": model.dof.detach().clone(), "losses": losses}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gloo_world2_matches_single_process(xarm7, tmp_path):
    n_views, steps = 4, 4
    out = str(tmp_path / "dp.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, n_views, steps, out), nprocs=2, join=True)
    dp = torch.load(out, weights_only=False)
    model, batch, _ = make_problem(xarm7, n_views)
    tr = RBSolverTrainer(Cfg(), model, batch)
    single = [float(tr.step()[1]) for _ in range(steps)]
    # one 8-float all-reduce per step reproduces the single-process mean-loss gradient
    assert np.allclose(dp["losses"], single, rtol=1e-5)
    assert (dp["dof"] - model.dof.detach()).abs().max() < 1e-5


def test_history_cursor_follows_the_buffer_after_load(xarm7):
    """rb_solver.py:50-51: the next pose is recorded at the first all-zero row of ``history_ops``.  The host cursor that
    replaces the reference's per-step scan is re-derived from the buffer after load_state_dict (no GPU needed)."""
    import torch
    from easyhec_amd.config import Cfg
    from easyhec_amd.rb_solver import RBSolver
    from easyhec_amd.synthetic import camera_Tc_c2b
    cfg = Cfg()
    cfg.model.rbsolver.init_Tc_c2b = camera_Tc_c2b().tolist()
    m = RBSolver(cfg, meshes=[xarm7.meshes[0]])
    assert m.history_cursor() == 0
    sd = m.state_dict()
    sd["history_ops"] = sd["history_ops"].clone()
    sd["history_ops"][:37] = torch.arange(1, 37 * 6 + 1, dtype=torch.float32).reshape(37, 6)
    m2 = RBSolver(cfg, meshes=[xarm7.meshes[0]])
    m2._hist_n = 5
    m2.load_state_dict(sd)
    assert m2._hist_n is None and m2.history_cursor() == 37


def test_mask_reader_matches_cv2_imread_flag_2(tmp_path):
    """xarm_real.py:40 reads masks with cv2.imread(path, 2) > 0: grey conversion, alpha ignored, palettes resolved."""
    import numpy as np
    from PIL import Image
    from easyhec_amd.data import _read_mask
    a = np.zeros((4, 5, 4), np.uint8)
    a[..., 3] = 255                       # opaque alpha everywhere must NOT make everything foreground
    a[1, 2, :3] = (0, 200, 0)
    a[2, 3, :3] = (1, 0, 0)               # rounds to grey 0 in OpenCV's BGR2GRAY -> background
    Image.fromarray(a, "RGBA").save(tmp_path / "rgba.png")
    m = _read_mask(str(tmp_path / "rgba.png"))
    assert m.sum() == 1 and m[1, 2]
    p = Image.new("P", (5, 4), 0)
    p.putpalette([0, 0, 0, 255, 255, 255] + [0] * (254 * 3))
    p.putpixel((3, 1), 1)
    p.save(tmp_path / "pal.png")
    m = _read_mask(str(tmp_path / "pal.png"))
    assert m.sum() == 1 and m[1, 3]
    g16 = np.zeros((4, 5), np.uint16)
    g16[0, 0] = 300
    Image.fromarray(g16).save(tmp_path / "g16.png")
    assert _read_mask(str(tmp_path / "g16.png")).sum() == 1


def test_bench_multi_rank_launch_forms_the_group_and_reports_missing_devices():
    """The driver's N > 1 launch line (torch.distributed.run, 127.0.0.1, one process per GPU) on a box without GPUs:
    every rank reaches init_process_group, the group forms, and each rank exits with a clear message -- no hang, no
    traceback from deep inside the library."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without HIP devices")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "the process group formed" in out and "no HIP device for LOCAL_RANK" in out, out[-2000:]
