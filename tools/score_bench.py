"""Throughput of the space-explorer scoring kernel on the reference's default exploration round
(cfg.model.space_explorer: 1000 candidate joint configurations x 10 sampled camera poses @1280x720, xArm7).
Run on the GPU box:  python tools/score_bench.py [--q 1000] [--s 10] [--chunk 0] [--reps 5]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from easyhec_amd import space_explorer as se  # noqa: E402
from easyhec_amd.config import XARM7_K_1280x720  # noqa: E402
from easyhec_amd.robot import load_robot  # noqa: E402
from easyhec_amd.synthetic import camera_Tc_c2b, perturb_pose  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--q", type=int, default=1000)
    ap.add_argument("--s", type=int, default=10)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-q", type=int, default=0, help="also time the CPU oracle on this many candidates")
    a = ap.parse_args()
    H, W = 720, 1280
    rb = load_robot("xarm7")
    ex = se.SpaceExplorer(rb, XARM7_K_1280x720, H, W, chunk_views=a.chunk)
    rng = np.random.default_rng(0)
    q = rb.sample_qpos(a.q, rng, scale=1.0)
    Tc0 = camera_Tc_c2b()
    Tc = np.stack([perturb_pose(Tc0, dt=rng.normal(0, 0.02, 3), drot_deg=rng.normal(0, 2.0, 3)) for _ in range(a.s)])
    t0 = time.time()
    lp = ex.link_poses(q)
    t_fk = time.time() - t0
    mvp = ex.mvp(Tc, lp)
    torch.cuda.synchronize()
    var, score = se.mask_variance(ex.glctx, ex.scene, mvp, H, W, chunk_views=a.chunk)   # warm-up (allocations)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        t0 = time.time()
        var, score = se.mask_variance(ex.glctx, ex.scene, mvp, H, W, chunk_views=a.chunk)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    t = float(np.median(ts))
    n = a.q * a.s
    out = {"metric": "space-explorer mask renders/sec (non-AA packed robot mask + per-pixel variance over poses)",
           "value": round(n / t, 1), "unit": "renders/s", "seconds_per_round": round(t, 5),
           "us_per_render": round(1e6 * t / n, 2), "dtype": "f32 positions, integer coverage/score",
           "config": {"workload": f"xarm7 {a.q} qpos x {a.s} poses @1280x720", "chunk_views": a.chunk or 512,
                      "robot_tris": rb.num_tris}, "fk_seconds_host": round(t_fk, 3), "best": int(score.argmax()),
           "var_max": float(var.max()), "var_min": float(var.min())}
    if a.cpu_q > 0:
        from oracle import oracle
        verts, tris, _, _ = helpers.scene_arrays(rb)
        vl = np.concatenate([np.full(v.shape[0], l, np.int32) for l, (v, _) in enumerate(rb.meshes)])
        m = mvp[:a.cpu_q].cpu().numpy()
        t0 = time.time()
        s_ref = oracle.mask_variance(verts, tris, vl, m, H, W)
        tc = time.time() - t0
        assert (s_ref == score[:a.cpu_q].cpu().numpy()).all(), "GPU score differs from the oracle"
        out["cpu_baseline"] = {"value": round(a.cpu_q * a.s / tc, 1), "unit": "renders/s", "cores": oracle.num_threads(),
                               "kind": "port", "sample": f"{a.cpu_q} candidates x {a.s} poses, oracle.mask_variance"}
    import json
    print(json.dumps(out))


if __name__ == "__main__":
    main()
