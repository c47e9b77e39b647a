#!/usr/bin/env python
"""bench.py -- mask-render fwd+bwd frames/s of the EasyHeC pose-optimisation step on MI355X.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): xArm7 (8 links, 41 096 triangles) at 1280x720,
8 synthetic views per GPU, antialias on, multi-view mask loss.  One "step" = one optimisation iteration of
/root/reference/easyhec/trainer/rbsolver.py:29-43 over that batch: dof -> Tc_c2b -> per-(view, link) clip matrices ->
fused render / composite / SSE loss / gradient -> 6-DoF gradient -> [one 8-float all-reduce when N > 1] -> Adam.
A "frame" = one camera view with all its links, forward + backward.  Inputs are resident in HBM before timing.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 200 --warmup 20

Rank 0 prints ONE JSON line (driver contract) with two extra objects: "roofline" (dominant kernel vs HBM peak,
hipEvent-timed on the launch stream) and "cpu_baseline" (the CPU oracle timed on this box's host cores; a reported
baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
WORKLOAD = "xarm7_1280x720_8view"   # BASELINE configs[2]; --workload selects another one for side measurements
VIEWS_PER_GPU = 8


def csrc_sha16():
    """First 16 hex digits of the SHA-256 over the kernel sources (easyhec_amd/csrc/*.hip, *.h, sorted by name): what the
    committed counter files (profiles/counters.json, profiles/traffic.json) are keyed on, so that this script can say when
    they describe other kernels than the ones it is timing (there is no .git on the GPU box)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "easyhec_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes_per_frame(robot, H, W):
    """2*G + 16*P + 128*L  (geometry read fwd+bwd; per pixel: write mask, read ref, read both again)."""
    G = 12 * robot.num_verts + 12 * robot.num_tris
    return 2 * G + 16 * H * W + 128 * robot.num_links


def build_problem(rank, world, dev, eager=False, graph=True, workload=WORKLOAD):
    from easyhec_amd import dr, fused
    from easyhec_amd.config import Cfg
    from easyhec_amd.rb_solver import RBSolver
    from easyhec_amd.robot import load_robot
    from easyhec_amd.synthetic import WORKLOADS, camera_Tc_c2b, make_views, perturb_pose
    from easyhec_amd.trainer import RBSolverTrainer, shard_views

    wl = WORKLOADS[workload]
    robot = load_robot(wl["robot"])
    H, W, K = wl["H"], wl["W"], np.asarray(wl["K"], dtype=np.float64)
    n_views = (VIEWS_PER_GPU if workload == WORKLOAD else wl["views"]) * world
    _, link_poses = make_views(robot, n_views, seed=0)
    lo, hi = shard_views(n_views, rank, world)
    link_poses = link_poses[lo:hi]
    B = hi - lo
    Tc_gt = camera_Tc_c2b(radius=wl["radius"], lift=wl["lift"])
    Tc_init = perturb_pose(Tc_gt)

    cfg = Cfg()
    cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W
    cfg.model.rbsolver.init_Tc_c2b = Tc_init.tolist()
    model = RBSolver(cfg, meshes=robot.meshes).to(dev)
    lp = torch.tensor(link_poses, dtype=torch.float32, device=dev)
    Kt = torch.tensor(K, dtype=torch.float32, device=dev)
    Tgt = torch.tensor(Tc_gt, dtype=torch.float32, device=dev)
    # reference masks = binarised ground-truth silhouettes rendered by the HIP path (no dataset offline)
    renderer = model._ensure_renderer()
    scene = model._ensure_scene()
    with torch.no_grad():  # (on a context of its own, released afterwards: the solve's context holds what the solve needs)
        tmp_ctx = dr.RasterizeCudaContext(dev)
        mvp_gt = fused.mvp_matrices(Kt, H, W, Tgt, lp)
        gt_mask, _ = fused.render_mask_loss(tmp_ctx, scene, mvp_gt, torch.zeros((B, H, W), device=dev))
        ref = (gt_mask > 0.5).float().contiguous()
        torch.cuda.synchronize()
        del tmp_ctx, gt_mask
    batch = {"mask": ref, "link_poses": lp, "K": Kt[None].repeat(B, 1, 1), "Tc_c2b": Tgt[None].repeat(B, 1, 1)}
    trainer = RBSolverTrainer(cfg, model, batch, fast=not eager, graph=graph)
    return dict(robot=robot, H=H, W=W, K=K, B=B, model=model, trainer=trainer, link_poses=link_poses, Tc_gt=Tc_gt,
                Tc_init=Tc_init, ref=ref, glctx=renderer.glctx, n_views=n_views)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(p, budget_s=8.0):
    """The CPU oracle (oracle/, kind "port": the reference has no CPU renderer) on the same 8-view batch
    at the initial pose, fwd+bwd, OpenMP over the (view, link) images, then rows: every hardware thread the
    process may use (`value`: os.sched_getaffinity, i.e. the cgroup / affinity limit of the box, stated as `cores`; at most
    views x links images are in flight), 8 threads (`value_8thread`, one per view: round 5's figure) and one thread
    (`value_1thread`).  Bounded samples: each leg repeats until ~budget_s of wall time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from oracle import oracle
    verts, tris, toff, voff = helpers.scene_arrays(p["robot"])
    mvp = helpers.mvp_numpy(p["K"], p["H"], p["W"], p["Tc_init"], p["link_poses"])
    ref = p["ref"].cpu().numpy()
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    images = mvp.shape[0] * mvp.shape[1]
    nthreads = max(1, min(allowed, images))

    def leg(threads, views):
        oracle.set_num_threads(threads)
        oracle.render_mask_loss(verts, tris, toff, voff, mvp[:views], ref[:views])  # warm-up
        reps, t0 = 0, time.perf_counter()
        while True:
            oracle.render_mask_loss(verts, tris, toff, voff, mvp[:views], ref[:views])
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget_s or reps >= 200:
                break
        return reps * views / el, reps, el

    fps, reps, el = leg(nthreads, mvp.shape[0])
    fps8, reps8, el8 = leg(min(8, nthreads), mvp.shape[0])
    fps1, reps1, el1 = leg(1, 1)
    oracle.set_num_threads(allowed)
    return {"value": round(fps, 3), "unit": "frames/s", "cores": int(nthreads), "kind": "port",
            "value_8thread": round(fps8, 3), "value_1thread": round(fps1, 3), "cpu_model": cpu_model(),
            "host_cores": os.cpu_count(), "allowed_cores": int(allowed),
            "sample": f"{reps} x ({mvp.shape[0]} views 1280x720, fwd+bwd) of the same batch at the initial pose, {el:.1f} s "
                      f"wall, OpenMP over the {images} (view, link) images then rows ({nthreads} threads); 8-thread leg: "
                      f"{reps8} x {mvp.shape[0]} views, {el8:.1f} s; 1-thread leg: {reps1} x 1 view, {el1:.1f} s; "
                      "oracle built with gcc -O2 -mfma -ffp-contract=off (oracle/Makefile; not -O3 -march=native: the "
                      "library is built in the CPU container and must run on the GPU box's host)"}


def measured_copy_bandwidth(dev, nbytes=1 << 30, reps=10):
    """Device-to-device copy rate of this box (read + write bytes per second), the practical HBM ceiling quoted beside
    the 8 TB/s spec."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def timed_blocks(step, steps, barrier, min_ms, world=1, dev=None, max_blocks=1000):
    """Blocks of exactly `steps` steps, each between two barriers (driver contract), repeated until `min_ms` of timed
    work has accumulated; returns (mean seconds per block, blocks).  Every rank runs the same number of blocks."""
    elapsed, blocks = 0.0, 0
    while True:
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        elapsed += time.perf_counter() - t0
        blocks += 1
        again = elapsed * 1e3 < min_ms and blocks < max_blocks
        if world > 1:
            flag = torch.tensor([1 if again else 0], device=dev)
            dist.broadcast(flag, src=0)
            again = bool(int(flag.item()))
        if not again:
            break
    return elapsed / blocks, blocks


def stage_times(p, steps, step_fn=None):
    """hipEvents around each kernel of the chain (eager launches), `steps` more steps of the same optimisation.
    The default chain launches no resolve kernel (the job kernel resolves its own jobs): the event pair that would bracket
    it is EMPTY, and what it measures is the cost of a pair of events on this stack -- reported as ``event_pair_overhead``
    instead of as a stage; every real stage's figure includes about that much."""
    from easyhec_amd import fused
    tr = p["trainer"]
    fused.set_timing(p["glctx"], True)
    if tr.fast is not None:
        tr.fast.release_graph()  # hipEvents between kernels need eager launches
    for _ in range(steps):
        (step_fn or tr.step)()
    stage_ms, ncalls = fused.read_timing(p["glctx"])
    fused.set_timing(p["glctx"], False)
    st = {k: v / max(ncalls, 1) for k, v in stage_ms.items() if not k.startswith("unused")}
    if "resolve" in st and st["resolve"] < 0.5 * min(st.get("vertex", 1.0), st.get("composite", 1.0)):
        st["event_pair_overhead"] = st.pop("resolve")  # (an empty pair; a chain with the general-triangle pass on keeps "resolve")
    return st


def traffic_of(workload):
    """PMC-measured HBM bytes per step of `workload`'s launch chain from profiles/traffic[_<workload>].json (tools/gpu_traffic.sh;
    separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction), or None; stale = collected on other kernel sources."""
    name = "traffic.json" if workload == WORKLOAD else f"traffic_{workload}.json"
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    try:
        tj = json.load(open(path))
        return {"whole_op": tj.get("hbm_bytes_whole_op"), "dominant_kernel": tj.get("hbm_bytes_dominant_kernel"),
                "source": {"file": "profiles/" + name, "commit": tj.get("commit"), "launch_form": tj.get("launch_form"),
                           "csrc_sha16": tj.get("csrc_sha16")},
                "stale": tj.get("csrc_sha16") != csrc_sha16()}
    except Exception:
        return None


FRAC_NOTE = ("equivalent rate: ALGORITHMIC bytes (SURVEY 8d: 2 x geometry + 16 B per pixel + 128 B per link, per frame) / the "
             "dominant kernel's time / 8 TB/s.  The bound-reference, mask = NULL launch form does not move the 16 B per pixel the "
             "count budgets, so this can exceed 1; frac_hbm_actual beside it is the bandwidth the kernel really uses (PMC bytes)")


def side_workload(name, dev, steps, warmup, min_ms=30.0, graph=False):
    """One of the other BASELINE configs on this GPU (configs[1], [3] and [4]'s total on one device), timed exactly like the
    headline -- the same launch form too (``graph``: the headline's --graph flag; until round 6 these were replayed from a
    hipGraph whatever the headline did, which costs the three-launch chain 4 % at Franka 16 x 1080p):
    {ms_per_step, value, frac, frac_step, kernel_ms}.  Bounded: ~`min_ms` of timed work."""
    from easyhec_amd import fused
    p = build_problem(0, 1, dev, graph=graph, workload=name)
    tr = p["trainer"]

    def barrier():
        torch.cuda.synchronize()

    for _ in range(warmup):
        tr.step()
    if tr.fast is not None and not np.isfinite(float(tr.last_loss)) and tr.fast.recover_from_overflow():
        for _ in range(warmup):
            tr.step()
    el, blocks = timed_blocks(tr.step, steps, barrier, min_ms)
    fused.check_status(p["glctx"])
    st = stage_times(p, steps)
    bytes_frame = algorithmic_bytes_per_frame(p["robot"], p["H"], p["W"])
    fps = p["n_views"] * steps / el
    kernel_ms = st[fused.DOMINANT_STAGE]
    out = {"ms_per_step": round(el / steps * 1e3, 4), "value": round(fps, 1), "views": p["n_views"],
           "frac": round(bytes_frame * p["B"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kernel_ms > 0 else None,
           "frac_step": round(fps * bytes_frame / 1e9 / HBM_PEAK_GBS, 4), "kernel_ms": round(kernel_ms, 5),
           "stage_ms": {k: round(v, 5) for k, v in st.items()},
           "timed_blocks": blocks, "final_mask_loss": round(float(tr.last_loss), 3), "frac_is": FRAC_NOTE}
    tq = traffic_of(name)
    if tq and tq["dominant_kernel"] and kernel_ms > 0:
        out["frac_hbm_actual"] = round(tq["dominant_kernel"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        out["hbm_actual_step_gbs"] = round(tq["whole_op"] / (el / steps) / 1e9, 2)
        out["traffic"] = tq["whole_op"]
        out["traffic_dominant_kernel"] = tq["dominant_kernel"]
        out["traffic_source"] = tq["source"]
        out["counters_stale"] = bool(tq["stale"])
    else:
        out["frac_hbm_actual"] = None  # (no PMC pass of this workload under profiles/)
    del tr, p
    torch.cuda.empty_cache()
    return out


def drop_in_step(p, dev, steps=20):
    """The same optimisation step the way a maintainer gets it by only swapping the import:
    RBSolver.forward through dr.rasterize / dr.interpolate / dr.antialias per (view, link) -- or the fused op -- under torch
    autograd with torch.optim.Adam, recorded once in a torch.cuda.CUDAGraph (RBSolverTrainer(graph=True)) and replayed.
    Same views, intrinsics and reference masks as the headline.  Informational: the headline is the launch chain."""
    from easyhec_amd.config import Cfg
    from easyhec_amd.rb_solver import RBSolver
    from easyhec_amd.trainer import RBSolverTrainer
    tr0 = p["trainer"]
    res = {}
    # three_ops: this repo's optimised mirror of the reference's schedule (renderer.NVDiffrastRenderer); import_swap_only: the
    # call pattern an import swap alone gives (renderer.NVDiffrastRenderer(plain=True) + RBSolver._forward_per_call: three
    # colour channels, rast_db written, nothing cached or batched) -- the number INTEGRATION.md section 2 promises
    # three_ops_batched: the same three ops called once per step over all (view, link) images (nvdiffrast's range mode)
    for name, fusedflag, refsched, graphs in (("three_ops", False, False, (True,)), ("three_ops_batched", False, "batched", (True,)),
                                              ("fused_op_autograd", True, False, (True,)),
                                              ("import_swap_only", False, True, (True, False))):
        for graph in graphs:
            cfg = Cfg()
            cfg.model.rbsolver.H, cfg.model.rbsolver.W = p["H"], p["W"]
            cfg.model.rbsolver.init_Tc_c2b = p["Tc_init"].tolist()
            cfg.model.rbsolver.use_fused = fusedflag
            cfg.model.rbsolver.reference_schedule = refsched is True
            cfg.model.rbsolver.batched_ops = refsched == "batched"
            model = RBSolver(cfg, meshes=p["robot"].meshes).to(dev)
            batch = {k: tr0.batch[k] for k in ("mask", "link_poses", "K")}
            tr = RBSolverTrainer(cfg, model, batch, graph=graph)
            n = steps if graph else max(3, steps // 4)
            for _ in range(3):
                tr.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                tr.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            res[f"{name}_{'graph' if graph else 'eager'}_ms_per_step"] = round(ms, 3)
            if graph:  # (3 + `steps` iterations from the same start for every variant: the losses are comparable)
                res[f"{name}_loss"] = round(float(tr.last_loss), 3)
            del tr, model
    # SURVEY 8d's secondary byte count: the REFERENCE's traffic shape through its three ops is 212-228 B per pixel of every
    # (view, link) image (fwd + bwd).  A count for comparison only -- NOT bytes this library moves (its mirror writes no
    # rast_db and one colour channel), so no rate is derived from it
    res["reference_three_op_traffic_shape_bytes_per_step"] = int(220.0 * p["n_views"] * len(p["robot"].meshes) * p["H"] * p["W"])
    return res


def main():
    # stdout carries exactly ONE line, the JSON: libraries that chat on stdout (RCCL prints a version banner when a
    # communicator is created) go to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(json_fd, "w")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--min-ms", type=float, default=50.0, help="repeat the timed block of --steps steps until this much "
                    "timed work has accumulated (0 = exactly one block)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="reference-shaped torch autograd step instead of the HIP launch chain")
    ap.add_argument("--workload", default=WORKLOAD, help="side measurements only; the headline is the default")
    ap.add_argument("--no-with-mask", action="store_true", help="skip the with-mask leg (profiling runs: one launch form per kernel)")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the drop-in (three ops under autograd, graph replay) step, ~20 s")
    ap.add_argument("--no-side", action="store_true", help="skip the side workloads (the other BASELINE configs, ~0.5 s each)")
    ap.add_argument("--graph", action="store_true", help="replay the launch chain as a natively captured hipGraph (saves host time only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # EHR_BENCH_DEVICE / EHR_BENCH_BACKEND: test-only (tests/test_gpu_launch.py runs this script with two ranks on ONE
    # device over gloo, so that the N > 1 reporting code below has executed before the driver's multi-GPU run)
    dev_index = int(os.environ.get("EHR_BENCH_DEVICE", local_rank))
    backend = os.environ.get("EHR_BENCH_BACKEND", "nccl")
    have_gpu = torch.cuda.is_available() and dev_index < torch.cuda.device_count()
    dev = torch.device("cuda", dev_index) if have_gpu else None
    if have_gpu:
        torch.cuda.set_device(dev_index)
    if world > 1:
        # rendezvous first (127.0.0.1, env:// as the driver launches it): one process per GPU, RCCL ("nccl") over xGMI.
        # Without a device the group still forms (gloo) so that a mis-launch is reported by every rank, not as a hang.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if have_gpu and backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        elif have_gpu:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if not have_gpu:
        msg = (f"bench.py rank {rank}/{world}: no HIP device for LOCAL_RANK={local_rank} "
               f"(device index {dev_index}, {torch.cuda.device_count() if torch.cuda.is_available() else 0} visible) -- the process group formed, "
               "but there is no CPU path to time (the render path is HIP only)")
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit(msg)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from easyhec_amd import fused
    p = build_problem(rank, world, dev, eager=args.eager, graph=args.graph, workload=args.workload)
    tr = p["trainer"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = tr.step
    used_graph = tr.fast is not None and bool(tr.fast._graph)
    for _ in range(args.warmup):
        step()
    # A step the launch chain REPORTS (NaN loss: its slot-limited plan overflowed, or the view needs the general-triangle
    # pass the chain does not launch by default) leaves the optimiser untouched; recover once, before anything is timed,
    # exactly as RBSolverTrainer.fit does (never needed on the BASELINE workloads; all ranks decide alike: same loss).
    if tr.fast is not None and not np.isfinite(float(tr.last_loss)):
        recovered = torch.tensor([1 if tr.fast.recover_from_overflow() else 0], dtype=torch.int32, device=dev)
        if world > 1:  # (the cause is local to a rank's views, the extra steps below hold a collective each: decide together)
            dist.all_reduce(recovered, op=dist.ReduceOp.MAX)
        if int(recovered.item()):
            for _ in range(args.warmup):
                step()
    # The timed block is exactly --steps steps between two barriers (driver contract).  One block of a 0.1 ms step is a
    # thin sample (20 steps = 2 ms), so blocks are repeated -- each bracketed the same way, the optimisation simply
    # continues -- until at least --min-ms of timed work has accumulated; the reported time per step is the mean over all
    # timed steps.  Every rank runs the same number of blocks (rank 0's decision is broadcast).
    elapsed, blocks = timed_blocks(step, args.steps, barrier, args.min_ms, world, dev)  # mean duration of one block
    fused.check_status(p["glctx"])
    per_rank_ms, allreduce_us = None, None
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [round(float(x.item()) / args.steps * 1e3, 4) for x in allr]
        elapsed = max(float(x.item()) for x in allr)  # the job is as slow as its slowest rank
        # the step's ONE exchange on its own: 8 floats, latency-bound -- the one the step makes first, then (where
        # they are available) the others beside it: the library's peer-memory exchange (one kernel, Adam not included here),
        # ncclAllReduce on the library's communicator, torch.distributed's all-reduce
        import ctypes
        from easyhec_amd import _lib
        buf = torch.zeros(8, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        zf = ctypes.c_float(0.0)
        kinds = {}
        if tr.fast is not None and tr.fast.p2p:
            kinds["peer_memory"] = lambda: _lib.check(_lib.lib().ehr_comm_p2p_step(
                p["glctx"].handle, _lib.ptr(buf), None, None, None, None, zf, zf, zf, zf, zf, None, None, stream), "ehr_comm_p2p_step")
        if tr.fast is not None and tr.fast.rccl:
            kinds["rccl"] = lambda: _lib.check(_lib.lib().ehr_comm_allreduce(p["glctx"].handle, _lib.ptr(buf), 8, stream), "ehr_comm_allreduce")
        kinds["torch_distributed"] = lambda: dist.all_reduce(buf)
        exchange_us = {}
        for name, one in kinds.items():  # (every rank runs the same sequence: the exchanges are collective)
            for _ in range(20):
                one()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(200):
                one()
            torch.cuda.synchronize()
            exchange_us[name] = round((time.perf_counter() - ta) / 200 * 1e6, 2)
        exchange_used = next(iter(kinds))
        allreduce_us = exchange_us[exchange_used]
    final_loss = float(tr.last_loss)

    # the same step WITH the rendered masks written (rb_solver.py:73-77 materialises `rendered_masks` every step): the
    # chain then streams every tile of every view (the bound reference's cached sums do not apply) and writes mask[B,H,W]
    with_mask_ms, with_mask_stage_ms = None, None
    if tr.fast is not None and world == 1 and not args.no_with_mask:
        mstep = lambda: tr.fast.step(want_mask=True)  # noqa: E731
        for _ in range(max(2, args.warmup // 4)):
            mstep()
        el_m, _ = timed_blocks(mstep, args.steps, barrier, min(args.min_ms, 30.0))
        with_mask_ms = el_m / args.steps * 1e3
        with_mask_stage_ms = stage_times(p, args.steps, mstep)
    # roofline leg: hipEvents around each kernel of the fused op, same K steps again (continuing the optimisation)
    stage_ms = stage_times(p, args.steps)
    side = None
    if world == 1 and rank == 0 and not args.no_side and args.workload == WORKLOAD and not args.eager:
        side = {}
        for name in ("xarm7_640x480_1view", "franka_1920x1080_16view", "xarm7_1280x720_64view"):
            try:
                # (its own step counts: a side measurement is bounded by time -- ~0.1-0.2 s of GPU work each --, not by the driver's --steps)
                side[name] = side_workload(name, dev, 100, 30, min_ms=40.0, graph=args.graph)
            except Exception as e:  # a side measurement must never cost the headline line
                side[name] = {"error": f"{type(e).__name__}: {e}"}

    drop_in = None
    if world == 1 and rank == 0 and not args.no_drop_in and args.workload == WORKLOAD and not args.eager:
        try:
            drop_in = drop_in_step(p, dev)
        except Exception as e:  # informational: must never cost the headline line
            drop_in = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        frames = p["n_views"] * args.steps
        fps = frames / elapsed
        bytes_frame = algorithmic_bytes_per_frame(p["robot"], p["H"], p["W"])
        # the dominant kernel (fused.DOMINANT_KERNEL), bracketed by its own pair of hipEvents on the launch stream
        # (compare with rocprofv3's average for it in profiles/)
        tile_ms = stage_ms[fused.DOMINANT_STAGE]
        bytes_launch = bytes_frame * p["B"]
        achieved = bytes_launch / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
        step_gbs = fps * bytes_frame / 1e9  # SURVEY 8d's own definition: frames/s x bytes_frame (whole step, all kernels)
        copy_gbs = measured_copy_bandwidth(dev)
        # HBM bytes from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections): counters cannot
        # be read inside this run, so the figures come from profiles/traffic.json -- collected on the SAME launch form
        # this script times (ehr_solver_step, bound reference, no mask output; tools/gpu_traffic.sh) at the commit it names
        traffic, traffic_kernel, traffic_src = None, None, None
        sha_now, stale = csrc_sha16(), False
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.workload == WORKLOAD:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_whole_op")        # PMC-measured HBM bytes of ALL kernels of one step
                traffic_kernel = tj.get("hbm_bytes_dominant_kernel")
                traffic_src = {"file": "profiles/traffic.json", "commit": tj.get("commit"), "launch_form": tj.get("launch_form"),
                               "csrc_sha16": tj.get("csrc_sha16")}
                stale = stale or tj.get("csrc_sha16") != sha_now
            except Exception:
                traffic = None
        # what binds the dominant kernel besides bandwidth (it is not HBM-bound at 8 views): VALU issue utilisation from the SQ
        # counters and the compiler's register / spill figures, carried like `traffic` from a committed profile of this
        # command (profiles/counters.json, tools/make_counters.py; names its commit)
        counters = {}
        cpath = os.path.join(ROOT, "profiles", "counters.json")
        if os.path.exists(cpath) and args.workload == WORKLOAD:
            try:
                cj = json.load(open(cpath))
                counters = {"valu_util": cj.get("valu_util"), "wave_active_frac": cj.get("wave_active_frac"),
                            "spilled_sgprs": cj.get("resources", {}).get("spilled_sgprs"),
                            "spilled_vgprs": cj.get("resources", {}).get("spilled_vgprs"),
                            "vgprs": cj.get("resources", {}).get("vgprs"), "waves_per_simd": cj.get("resources", {}).get("waves_per_simd"),
                            "kernel_us_rocprof": cj.get("kernel_us_rocprof"),
                            "counters_source": {"file": "profiles/counters.json", "commit": cj.get("commit"),
                                                "csrc_sha16": cj.get("csrc_sha16")}}
                stale = stale or cj.get("csrc_sha16") != sha_now
            except Exception:
                counters = {}
        out = {
            "metric": "mask-render fwd+bwd frames/sec, xArm7 50k-tri @1280x720x8-view" if args.workload == WORKLOAD else f"mask-render fwd+bwd frames/sec, {args.workload}",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "timed_blocks": blocks,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "robot": f"{p['robot'].name} ({p['robot'].num_tris} tris, {p['robot'].num_verts} verts)",
                       "resolution": [p["H"], p["W"]], "views_per_gpu": p["B"], "global_views": p["n_views"],
                       "links": p["robot"].num_links, "antialias": True, "optimizer": "Adam lr 3e-3 wd 5e-4",
                       # the reference masks are bound once per solve (ehr_fused_bind_ref): the loss of the tiles no link
                       # touches is a cached constant -- an exact algebraic saving (64-bit fixed-point sums, bit-identical
                       # to streaming the whole image every step), not skipped work
                       "ref_sums": "bound once" if tr.fast is not None else "n/a",
                       # the timed launch form writes no mask image (mask = NULL; loss, gradient and Adam bit-identical to the
                       # form that does): `with_mask_ms_per_step` below times the same chain writing rendered_masks every step
                       "mask_output": False if tr.fast is not None else True,
                       "step": "torch autograd" if args.eager else ("HIP launch chain" + (", hipGraph replay" if used_graph else "")),
                       "parallelism": (f"dp{world} over views, one 8-float exchange/step ("
                                       + ("peer-memory mailboxes over xGMI, sums in rank order + Adam in one kernel on the chain's stream"
                                          if tr.fast is not None and tr.fast.p2p else
                                          "ncclAllReduce on the chain's stream, library-owned RCCL communicator"
                                          if tr.fast is not None and tr.fast.rccl else "torch.distributed") + ")")
                       if world > 1 else "single GPU",
                       "final_mask_loss": round(final_loss, 3),
                       # device memory the solve's rasterizer context holds (job slots: one per view tile by default)
                       "context_scratch_mb": round(p["glctx"].scratch_bytes() / 1048576.0, 1)},
            "roofline": {"bound": "hbm", "kernel": fused.DOMINANT_KERNEL, "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         # the same algorithmic bytes over the WHOLE step (driver-timed value x bytes_frame): the honest
                         # figure for "how close is the step to streaming its images once", next to the per-kernel one
                         "achieved_step": round(step_gbs, 2), "frac_step": round(step_gbs / HBM_PEAK_GBS, 5),
                         "peak_measured_copy": round(copy_gbs, 1), "frac_step_vs_measured_copy": round(step_gbs / copy_gbs, 5),
                         "traffic": traffic, "traffic_dominant_kernel": traffic_kernel, "traffic_source": traffic_src,
                         # real HBM rate of the whole step: counter bytes / driver-timed step (next to the algorithmic one)
                         "hbm_actual_step": round(traffic / (elapsed / args.steps) / 1e9, 2) if traffic else None,
                         # ... and of the dominant kernel: counter bytes / its duration in THIS run / peak.  `frac` above is an
                         # equivalent rate (algorithmic bytes the kernel does not move: no mask is written, the bound reference
                         # is not re-read); this is the bandwidth the kernel really uses
                         "frac_hbm_actual": round(traffic_kernel / (tile_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                         if traffic_kernel and tile_ms > 0 else None,
                         # true when profiles/counters.json or traffic.json were collected on other kernel sources than the
                         # ones in this tree (they name the SHA-256 of easyhec_amd/csrc they were collected on)
                         "counters_stale": bool(stale), "csrc_sha16": sha_now,
                         "algorithmic_bytes_per_launch": bytes_launch, "kernel_ms": round(tile_ms, 5),
                         "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()}},
        }
        out["roofline"].update(counters)
        out["roofline"]["frac_is"] = FRAC_NOTE
        if with_mask_ms is not None:
            out["with_mask_ms_per_step"] = round(with_mask_ms, 4)
            out["with_mask_value"] = round(p["n_views"] / (with_mask_ms * 1e-3), 1)
            # the same figures for the launch form that WRITES the rendered masks every step (rb_solver.py:73-77's
            # `rendered_masks`): the step the reference's own forward corresponds to
            km = with_mask_stage_ms[fused.DOMINANT_STAGE]
            fps_m = p["n_views"] / (with_mask_ms * 1e-3)
            out["roofline_with_mask"] = {
                "launch_form": "ehr_solver_step with a mask output: mask[B,H,W] written every step (4 B per pixel: the job tiles by "
                               "their owners, zeros elsewhere), reference read at the job tiles, its sums over the other tiles cached",
                "ms_per_step": round(with_mask_ms, 4), "value": round(fps_m, 1), "kernel": fused.DOMINANT_KERNEL,
                "kernel_ms": round(km, 5), "frac": round(bytes_launch / (km * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if km > 0 else None,
                "frac_step": round(fps_m * bytes_frame / 1e9 / HBM_PEAK_GBS, 5),
                "stage_ms": {k: round(v, 5) for k, v in with_mask_stage_ms.items()}, "frac_is": FRAC_NOTE}
        if side is not None:
            out["side"] = side
        if drop_in is not None:
            out["drop_in"] = drop_in
        if per_rank_ms is not None:
            out["per_rank_ms_per_step"] = per_rank_ms
            out["allreduce_8float_us"] = round(allreduce_us, 2)
            out["exchange"] = {"used_by_the_step": exchange_used, "us_per_exchange_of_8_floats": exchange_us}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p)
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
