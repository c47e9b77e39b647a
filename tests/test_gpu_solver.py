"""End-to-end on the GPU: the pose optimisation (BASELINE configs[1] shape: xArm7, 640x480, Adam on Tc_c2b) driven by
the HIP path versus the same optimisation driven by the CPU oracle.

Two facts shape the assertions (measured, tools/traj_check.py): (1) constant-LR Adam (lr 3e-3, the reference's
setting) jitters by ~lr around the optimum and a single view leaves depth/rotation weakly observed, so two runs that
differ by one ulp anywhere drift apart by ~1e-3 in dof after a few dozen steps -- nvdiffrast itself is not
run-to-run deterministic (float atomics); (2) with a handful of views the problem is well conditioned: the estimate
converges to the ground truth and HIP- and oracle-driven runs agree far inside north_star's 1 mm / 0.1 deg bar."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def pose_error(Ta, Tb):
    dt = np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]) * 1000.0
    R = Ta[:3, :3].T @ Tb[:3, :3]
    ang = np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))
    return dt, ang


def setup(xarm7, B, H=480, W=640):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_backend import OracleRBSolver
    from easyhec_amd import fused
    from easyhec_amd.config import XARM7_K_1280x720, Cfg
    from easyhec_amd.rb_solver import RBSolver
    from easyhec_amd.synthetic import camera_Tc_c2b, make_views, perturb_pose, scaled_K
    from easyhec_amd.trainer import RBSolverTrainer
    dev = torch.device("cuda:0")
    K = scaled_K(XARM7_K_1280x720, 0.5, W, H, True)
    _, lp = make_views(xarm7, B, seed=0)
    Tc = camera_Tc_c2b()
    cfg = Cfg()
    cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W
    cfg.model.rbsolver.init_Tc_c2b = perturb_pose(Tc).tolist()
    model = RBSolver(cfg, meshes=xarm7.meshes).to(dev)
    ren, scene = model._ensure_renderer(), model._ensure_scene()
    Kt = torch.tensor(K, dtype=torch.float32, device=dev)
    lpt = torch.tensor(lp, device=dev)
    with torch.no_grad():
        gt, _ = fused.render_mask_loss(ren.glctx, scene, fused.mvp_matrices(
            Kt, H, W, torch.tensor(Tc, dtype=torch.float32, device=dev), lpt), torch.zeros((B, H, W), device=dev))
    ref = (gt > 0.5).float()
    batch = {"mask": ref, "link_poses": lpt, "K": Kt[None].repeat(B, 1, 1),
             "Tc_c2b": torch.tensor(Tc, dtype=torch.float32, device=dev)[None].repeat(B, 1, 1)}
    tr = RBSolverTrainer(cfg, model, batch)
    cpu = OracleRBSolver(xarm7, perturb_pose(Tc), H, W)
    cb = {"mask": ref.cpu(), "link_poses": torch.tensor(lp), "K": torch.tensor(K, dtype=torch.float32)[None].repeat(B, 1, 1)}
    ctr = RBSolverTrainer(cfg, cpu, cb)
    return model, tr, cpu, ctr, Tc


def final_pose(model):
    from easyhec_amd.se3 import se3_exp_map
    return se3_exp_map(model.dof.detach().cpu()[None]).permute(0, 2, 1)[0].numpy().astype(np.float64)


def test_config2_single_view_200_iterations(xarm7, oracle):
    """configs[1]: 1 view, 200 Adam iterations.  Loss falls >5x; the HIP-driven trajectory is the oracle-driven one
    while rounding noise has not yet been amplified (first 10 steps), and stays statistically equivalent after."""
    from easyhec_amd.synthetic import perturb_pose
    model, tr, cpu, ctr, Tc = setup(xarm7, 1)
    gl, cl = [], []
    for it in range(200):
        gl.append(float(tr.step()[1]))
        if it < 10:
            cl.append(float(ctr.step()[1]))
            assert (model.dof.detach().cpu() - cpu.dof.detach()).abs().max() <= 5e-4, it
    assert np.allclose(gl[:3], cl[:3], rtol=1e-3)
    assert gl[-1] < 0.2 * gl[0]
    e0, e1 = pose_error(perturb_pose(Tc), Tc), pose_error(final_pose(model), Tc)
    assert e1[0] < 0.75 * e0[0] and e1[1] < e0[1], (e0, e1)
    assert tr.global_steps == 200 and float(model.history_ops[199].abs().sum()) > 0  # rb_solver.py:50-51 bookkeeping


def test_config2_as_written_three_starts(xarm7):
    """configs[1] exactly as BASELINE.json states it (ONE 640x480 view, 200 Adam iterations, lr 3e-3, wd 5e-4), from
    three different initial errors of ~30 mm / ~4 deg, HIP launch chain.  What a single silhouette pins down, measured
    with tools/config2_study.py (numbers in BASELINE.md): the mean of the last 20 iterates ends 1.1-2.2 mm and
    1.7-2.0 deg from the truth (the oracle-driven run of the same start: 1.4-3.0 mm / 1.8-2.1 deg, i.e. the same
    cloud), different starts end up to 3 mm / 3.8 deg apart -- one rotation about the viewing direction is weakly
    observed, so north_star's 1 mm / 0.1 deg bar is a property of the multi-view problem (next test: 0.3 mm /
    0.04 deg), not of this config.  Asserted here: every start gets within 4 mm / 3 deg and cuts the loss > 5x."""
    from easyhec_amd import fused
    from easyhec_amd.config import XARM7_K_1280x720, Cfg
    from easyhec_amd.rb_solver import RBSolver
    from easyhec_amd.se3 import se3_exp_map
    from easyhec_amd.synthetic import camera_Tc_c2b, make_views, perturb_pose, scaled_K
    from easyhec_amd.trainer import RBSolverTrainer
    dev = torch.device("cuda:0")
    H, W = 480, 640
    K = scaled_K(XARM7_K_1280x720, 0.5, W, H, True)
    _, lp = make_views(xarm7, 1, seed=0)
    Tc = camera_Tc_c2b()
    Kt, lpt = torch.tensor(K, dtype=torch.float32, device=dev), torch.tensor(lp, device=dev)
    starts = [((0.02, -0.015, 0.02), (3.0, -2.0, 2.0)), ((-0.015, 0.02, -0.01), (-2.0, 3.0, -1.5)),
              ((0.01, 0.01, -0.025), (1.5, 2.5, 3.0))]
    for dt, dr in starts:
        init = perturb_pose(Tc, dt=dt, drot_deg=dr)
        cfg = Cfg()
        cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W
        cfg.model.rbsolver.init_Tc_c2b = init.tolist()
        model = RBSolver(cfg, meshes=xarm7.meshes).to(dev)
        with torch.no_grad():
            gt, _ = fused.render_mask_loss(model._ensure_renderer().glctx, model._ensure_scene(), fused.mvp_matrices(
                Kt, H, W, torch.tensor(Tc, dtype=torch.float32, device=dev), lpt), torch.zeros((1, H, W), device=dev))
        tr = RBSolverTrainer(cfg, model, {"mask": (gt > 0.5).float(), "link_poses": lpt, "K": Kt[None]}, fast=True)
        dofs, first = [], None
        for it in range(200):
            loss = float(tr.step()[1])
            first = loss if first is None else first
            dofs.append(model.dof.detach().clone())
        Tm = se3_exp_map(torch.stack(dofs[-20:]).mean(0).cpu()[None]).permute(0, 2, 1)[0].numpy().astype(np.float64)
        e0, e1 = pose_error(init, Tc), pose_error(Tm, Tc)
        assert e0[0] > 20 and e1[0] <= 4.0 and e1[1] <= 3.0, (e0, e1)
        assert loss < 0.2 * first, (first, loss)
        fused.check_status(model._ensure_renderer().glctx)


def test_multi_view_converges_to_ground_truth_and_to_the_oracle_run(xarm7, oracle):
    """4 views: the HIP-driven estimate reaches the ground-truth pose and the oracle-driven estimate within
    1 mm / 0.1 deg (north_star's bar), comparing the mean of the last 20 iterates (Adam's constant-LR jitter)."""
    from easyhec_amd.se3 import se3_exp_map
    model, tr, cpu, ctr, Tc = setup(xarm7, 4)
    G, C = [], []
    for it in range(200):
        tr.step()
        ctr.step()
        G.append(model.dof.detach().cpu().clone())
        C.append(cpu.dof.detach().clone())
    Gm, Cm = torch.stack(G[-20:]).mean(0), torch.stack(C[-20:]).mean(0)
    Tg = se3_exp_map(Gm[None]).permute(0, 2, 1)[0].numpy().astype(np.float64)
    Tcpu = se3_exp_map(Cm[None]).permute(0, 2, 1)[0].numpy().astype(np.float64)
    dmm, ddeg = pose_error(Tg, Tcpu)
    assert dmm <= 1.0 and ddeg <= 0.1, (dmm, ddeg)
    emm, edeg = pose_error(Tg, Tc)
    assert emm <= 1.0 and edeg <= 0.1, (emm, edeg)
    lmm, ldeg = pose_error(final_pose(model), final_pose(cpu))   # even the last raw iterates agree
    assert lmm <= 1.0 and ldeg <= 0.1, (lmm, ldeg)


def test_reference_franka_offline_example_converges(tmp_path):
    """The reference's only real dataset (assets/franka_offline_example.zip, re-packed as a fixture): written back to
    the reference's directory layout, loaded with XarmRealDataset, optimised from the init pose of
    configs/franka/example_franka_offline.yaml.  The reference documents "mask loss converging in less than 1000
    iterations" with a fast drop to a plateau (docs/usage.md:41, docs/optimization_scalar.png)."""
    from PIL import Image
    from easyhec_amd.config import Cfg
    from easyhec_amd.data import XarmRealDataset, collate_all
    from easyhec_amd.rb_solver import RBSolver
    from easyhec_amd.robot import load_robot
    from easyhec_amd.trainer import RBSolverTrainer
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "franka_offline_example.npz"))
    shape = tuple(z["shape"])
    masks = np.unpackbits(z["masks"])[:int(np.prod(shape))].reshape(shape).astype(bool)
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "mask"))
    os.makedirs(os.path.join(d, "qpos"))
    for i in range(shape[0]):
        Image.fromarray((masks[i] * 255).astype(np.uint8)).save(os.path.join(d, "mask", f"{i:06d}.png"))
        np.savetxt(os.path.join(d, "qpos", f"{i:06d}.txt"), z["qpos"][i])
    np.savetxt(os.path.join(d, "K.txt"), z["K"])
    robot = load_robot("franka")
    ds = XarmRealDataset(d, robot)
    assert len(ds) == 10 and ds[0]["mask"].shape == (480, 640) and ds[0]["link_poses"].shape == (9, 4, 4)
    assert torch.equal(ds.Tc_c2b, torch.eye(4))             # no Tc_c2b.txt -> identity, metrics skipped
    batch = collate_all(ds, dev)
    cfg = Cfg()
    cfg.model.rbsolver.H, cfg.model.rbsolver.W = 480, 640
    cfg.model.rbsolver.init_Tc_c2b = z["init_Tc_c2b"].tolist()
    model = RBSolver(cfg, meshes=robot.meshes).to(dev)

    def iou():
        with torch.no_grad():
            out, ld = model(batch)
        r = out["rendered_masks"].cpu().numpy() > 0.5
        assert "metrics" not in out
        return float(ld["mask_loss"]), float(np.mean([(r[i] & masks[i]).sum() / (r[i] | masks[i]).sum() for i in range(10)]))

    l0, i0 = iou()
    tr = RBSolverTrainer(cfg, model, batch, fast=True)
    hist = [float(tr.step()[1]) for _ in range(1000)]
    l1, i1 = iou()
    assert 0.15 < i0 < 0.35 and l0 > 3.5e4
    assert l1 < 0.45 * l0 and i1 > 0.62, (l0, l1, i0, i1)
    assert hist[300] < 0.5 * hist[0]                          # the drop happens early, then a plateau
    assert abs(hist[-1] - hist[600]) < 0.05 * hist[600]
    # the consumer of the result (tools/validate.py:13-48 in the reference): newest checkpoint -> ckpt['model']['dof'] ->
    # render_api overlay on every colour frame
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tr.save(os.path.join(d, "models", "model_iteration_001000.pth"))
    os.makedirs(os.path.join(d, "color"))
    for i in range(shape[0]):
        Image.fromarray(np.full(shape[1:] + (3,), 128, np.uint8)).save(os.path.join(d, "color", f"{i:06d}.png"))
    out = os.path.join(d, "overlay")
    subprocess.run([sys.executable, os.path.join(root, "tools", "validate.py"), "--ckpt_dir", os.path.join(d, "models"),
                    "--data_dir", d, "--robot", "franka", "--out", out], check=True, timeout=600)
    ious = []
    for i in range(shape[0]):
        img = np.asarray(Image.open(os.path.join(out, f"rendered_mask_{i:06d}.png"))).astype(int)
        tinted = (img[..., 0] - img[..., 2]) > 40              # red overlay on the grey frame
        ious.append((tinted & masks[i]).sum() / (tinted | masks[i]).sum())
    assert np.mean(ious) > 0.6, ious


def test_online_loop_solver_plus_explorer(xarm7):
    """The reference's outer loop (trainer/rbsolve_iter.py:157-167) with simulated hardware: every round adds the frame
    the space explorer asked for, re-solves from the configured initial pose over all frames, and scores the next
    candidates from that round's pose history.  More frames must not hurt, and the pose must end within 1 mm / 0.1 deg."""
    from easyhec_amd import render_api
    from easyhec_amd.config import XARM7_K_1280x720
    from easyhec_amd.online import OnlineCalibration
    from easyhec_amd.synthetic import camera_Tc_c2b, perturb_pose
    H, W = 360, 640
    K = np.array(XARM7_K_1280x720, dtype=np.float64)
    K[:2] *= 0.5
    Tc_gt = camera_Tc_c2b()
    init = perturb_pose(Tc_gt, dt=(0.03, -0.02, 0.03), drot_deg=(4.0, -3.0, 3.0))
    rng = np.random.default_rng(1)
    asked = []

    def capture(qpos):
        asked.append(np.array(qpos))
        return render_api.nvdiffrast_render_xarm_api(None, Tc_gt, qpos, H, W, K)

    loop = OnlineCalibration(xarm7, K, H, W, init, capture, lambda r: (xarm7.sample_qpos(200, rng, scale=0.9), None),
                             num_epochs=600, explore_iters=4, sample=8, start=100)
    Tc = loop.fit(np.zeros(7))
    assert [r["frames"] for r in loop.log] == [1, 2, 3, 4] and len(asked) == 4
    assert all(r["variance"] > 0 and r["variance"] >= r["var_mean"] for r in loop.log[:-1])
    assert len({tuple(np.round(q, 6)) for q in asked}) == 4          # the explorer moved the robot every round
    emm, edeg = pose_error(Tc_gt, Tc)
    assert emm <= 1.0 and edeg <= 0.1, (emm, edeg)


def test_graph_replay_of_the_three_op_step_equals_eager(xarm7):
    """the reference-shaped step (three drop-in ops per (view, link) under torch autograd + torch
    Adam) recorded in a torch.cuda.CUDAGraph replays to the same trajectory as the eager step: same op sequence on the same
    inputs; the antialias gradient uses float atomics, so `dof` is compared within float-reassociation noise, the history
    rows and the loss curve likewise."""
    from easyhec_amd.trainer import RBSolverTrainer
    from test_gpu_fast import problem
    cfg, make, batch = problem(xarm7, 2, 120, 160, 0.125)
    cfg.model.rbsolver.use_fused = False
    ma, mb = make(), make()
    ta = RBSolverTrainer(cfg, ma, batch)
    tb = RBSolverTrainer(cfg, mb, batch, graph=True)
    assert tb._cuda_graph is not None
    la, lb = [], []
    for _ in range(8):
        la.append(float(ta.step()[1]))
        lb.append(float(tb.step()[1]))
    torch.cuda.synchronize()
    assert np.allclose(la, lb, rtol=1e-4), (la, lb)
    assert la[-1] < la[0]
    assert (ma.dof.detach() - mb.dof.detach()).abs().max() <= 1e-5
    assert mb.history_cursor() == 8 and ma.history_cursor() == 8
    assert (ma.history_ops[:8] - mb.history_ops[:8]).abs().max() <= 1e-5


@pytest.mark.parametrize("graph", [False, True])
def test_import_swap_only_schedule_equals_the_optimised_mirror(xarm7, graph):
    """``reference_schedule=True`` (``NVDiffrastRenderer(plain=True)`` +
    ``RBSolver._forward_per_call``) issues the call pattern of nvdiffrast_renderer.py:33-47 inside rb_solver.py:58-71: K
    projection, ones[V,3] and transform_pos per call, rast_db written, rast undetached, no topology argument, flip / stack /
    clamp per frame.  It must give what this repo's optimised mirror of the same schedule
    gives -- same masks, same loss curve, same pose trajectory to float-reassociation noise (the gradient through the
    barycentrics of an all-ones colour is exactly zero; the three channels are equal) -- eager and replayed from a graph."""
    from easyhec_amd.renderer import NVDiffrastRenderer
    from easyhec_amd.trainer import RBSolverTrainer
    from test_gpu_fast import problem
    cfg_a, make_a, batch = problem(xarm7, 2, 120, 160, 0.125)
    cfg_a.model.rbsolver.use_fused = False
    cfg_b, make_b, _ = problem(xarm7, 2, 120, 160, 0.125)
    cfg_b.model.rbsolver.use_fused = False
    cfg_b.model.rbsolver.reference_schedule = True
    ma, mb = make_a(), make_b()
    assert type(ma._ensure_renderer()) is NVDiffrastRenderer and not ma._ensure_renderer().plain and mb._ensure_renderer().plain
    with torch.no_grad():
        ra = ma(dict(batch, global_step=0))[0]["rendered_masks"]
        rb = mb(dict(batch, global_step=0))[0]["rendered_masks"]
    assert (ra - rb).abs().max() <= 1e-6 and float(ra.sum()) > 100.0
    for m in (ma, mb):   # (the probing forwards recorded a pose each)
        m.history_ops.zero_()
        m._hist_n = None
    ta = RBSolverTrainer(cfg_a, ma, batch)
    tb = RBSolverTrainer(cfg_b, mb, batch, graph=graph)
    la, lb = [], []
    for _ in range(6):
        la.append(float(ta.step()[1]))
        lb.append(float(tb.step()[1]))
    torch.cuda.synchronize()
    assert np.allclose(la, lb, rtol=1e-4), (la, lb)
    assert la[-1] < la[0]
    assert (ma.dof.detach() - mb.dof.detach()).abs().max() <= 1e-5


@pytest.mark.parametrize("graph", [False, True])
def test_batched_three_ops_equal_the_per_image_schedule(xarm7, graph):
    """``batched_ops=True`` calls dr.rasterize / dr.interpolate / dr.antialias ONCE per step over all (view, link) images
    (range mode: one concatenated vertex / triangle array, a (start, count) range per image) instead of once per image.
    Same ops, same arithmetic per image: masks, loss curve and pose trajectory equal the per-image schedule's to float
    noise (the sum over links is taken by a different torch reduction) -- eager and replayed from a graph."""
    from easyhec_amd.trainer import RBSolverTrainer
    from test_gpu_fast import problem
    cfg_a, make_a, batch = problem(xarm7, 3, 120, 160, 0.125)
    cfg_a.model.rbsolver.use_fused = False
    cfg_b, make_b, _ = problem(xarm7, 3, 120, 160, 0.125)
    cfg_b.model.rbsolver.use_fused = False
    cfg_b.model.rbsolver.batched_ops = True
    ma, mb = make_a(), make_b()
    with torch.no_grad():
        ra = ma(dict(batch, global_step=0))[0]["rendered_masks"]
        rb = mb(dict(batch, global_step=0))[0]["rendered_masks"]
    assert ra.shape == rb.shape and (ra - rb).abs().max() <= 1e-6 and float(ra.sum()) > 100.0
    for m in (ma, mb):
        m.history_ops.zero_()
        m._hist_n = None
    ta = RBSolverTrainer(cfg_a, ma, batch)
    tb = RBSolverTrainer(cfg_b, mb, batch, graph=graph)
    la, lb = [], []
    for _ in range(6):
        la.append(float(ta.step()[1]))
        lb.append(float(tb.step()[1]))
    torch.cuda.synchronize()
    assert np.allclose(la, lb, rtol=1e-4), (la, lb)
    assert la[-1] < la[0]
    assert (ma.dof.detach() - mb.dof.detach()).abs().max() <= 1e-5


@pytest.mark.parametrize("graph", [False, True])
def test_render_lanes_equal_the_one_stream_schedule(xarm7, graph):
    """``render_lanes=2`` issues the frames' render chains alternately on two lanes -- each its own HIP stream and rasterizer
    context -- so that independent chains overlap, forward and backward; ``render_lanes=1`` issues the same calls on the
    step's stream.  Same calls on the same inputs: the masks are the same bits, the loss curve and the pose trajectory equal to
    float noise (dr.antialias' gradient is a sum of float atomics) -- eager and replayed from a graph (parallel branches,
    which is also what the default, render_lanes=-1, gives a graph)."""
    from easyhec_amd.trainer import RBSolverTrainer
    from test_gpu_fast import problem
    cfg_a, make_a, batch = problem(xarm7, 3, 120, 160, 0.125)
    cfg_a.model.rbsolver.use_fused = False
    cfg_a.model.rbsolver.render_lanes = 1
    cfg_b, make_b, _ = problem(xarm7, 3, 120, 160, 0.125)
    cfg_b.model.rbsolver.use_fused = False
    assert cfg_b.model.rbsolver.render_lanes == -1
    if not graph:
        cfg_b.model.rbsolver.render_lanes = 2
    ma, mb = make_a(), make_b()
    with torch.no_grad():
        ra = ma(dict(batch, global_step=0))[0]["rendered_masks"]
        rb = mb(dict(batch, global_step=0))[0]["rendered_masks"]
    torch.cuda.synchronize()
    assert torch.equal(ra, rb) and float(ra.sum()) > 100.0
    for m in (ma, mb):
        m.history_ops.zero_()
        m._hist_n = None
    ta = RBSolverTrainer(cfg_a, ma, batch)
    tb = RBSolverTrainer(cfg_b, mb, batch, graph=graph)
    la, lb = [], []
    for _ in range(8):
        la.append(float(ta.step()[1]))
        lb.append(float(tb.step()[1]))
    torch.cuda.synchronize()
    assert len(mb.renderer._lanes) == 2 and not ma.renderer._lanes
    assert np.allclose(la, lb, rtol=1e-5), (la, lb)
    assert la[-1] < la[0]
    assert (ma.dof.detach() - mb.dof.detach()).abs().max() <= 1e-5
