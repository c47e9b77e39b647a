"""Simulated run of the reference's online loop (easyhec/trainer/rbsolve_iter.py:157-167): a hidden ground-truth camera
pose plays the RealSense + segmentation network, uniformly sampled joint vectors play the planner.  Prints per-round
records and the final pose error.  python tools/online_loop.py [--explore-iters 5] [--epochs 1000] [--candidates 1000]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easyhec_amd import render_api  # noqa: E402
from easyhec_amd.config import XARM7_K_1280x720  # noqa: E402
from easyhec_amd.online import OnlineCalibration  # noqa: E402
from easyhec_amd.robot import load_robot  # noqa: E402
from easyhec_amd.synthetic import camera_Tc_c2b, perturb_pose  # noqa: E402


def pose_error(A, B):
    d = np.linalg.inv(A) @ B
    ang = np.degrees(np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)))
    return float(np.linalg.norm(d[:3, 3]) * 1000), float(ang)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--explore-iters", type=int, default=5)
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--candidates", type=int, default=1000)
    ap.add_argument("--sample", type=int, default=10)
    ap.add_argument("--scale", type=float, default=0.5, help="image scale relative to 1280x720")
    a = ap.parse_args()
    rb = load_robot("xarm7")
    W, H = int(1280 * a.scale), int(720 * a.scale)
    K = np.array(XARM7_K_1280x720, dtype=np.float64)
    K[:2] *= a.scale
    Tc_gt = camera_Tc_c2b()
    init = perturb_pose(Tc_gt, dt=(0.04, -0.03, 0.05), drot_deg=(6.0, -5.0, 4.0))
    rng = np.random.default_rng(0)

    def capture(qpos):  # the camera + segmentation: the true silhouette
        return render_api.nvdiffrast_render_xarm_api(None, Tc_gt, qpos, H, W, K)

    def candidates(_round):
        return rb.sample_qpos(a.candidates, rng, scale=0.9), None

    loop = OnlineCalibration(rb, K, H, W, init, capture, candidates, num_epochs=a.epochs,
                             explore_iters=a.explore_iters, sample=a.sample)
    Tc = loop.fit(np.zeros(7))
    for r in loop.log:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
    e0, e1 = pose_error(Tc_gt, init), pose_error(Tc_gt, Tc)
    print(json.dumps({"initial_error_mm_deg": [round(x, 3) for x in e0], "final_error_mm_deg": [round(x, 3) for x in e1]}))


if __name__ == "__main__":
    main()
