"""Step time of the drop-in three-op path (RBSolver(use_fused=False): dr.rasterize -> dr.interpolate -> dr.antialias per
(view, link), torch autograd, torch.optim.Adam) on the headline workload -- what a maintainer gets by only swapping the
import.  python tools/three_op_bench.py [--views 8] [--steps 20]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from easyhec_amd.config import Cfg  # noqa: E402
from easyhec_amd.rb_solver import RBSolver  # noqa: E402
from easyhec_amd.robot import load_robot  # noqa: E402
from easyhec_amd.synthetic import WORKLOADS, camera_Tc_c2b, make_views, perturb_pose  # noqa: E402
from easyhec_amd.trainer import RBSolverTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="record the autograd step in a torch.cuda.CUDAGraph and replay it "
                    "(RBSolverTrainer(graph=True)): the ~1 500 launches of the three-op step at GPU speed")
    ap.add_argument("--lanes", type=int, default=-1, help="render_lanes of the three_ops line (streams the frames' chains go to)")
    ap.add_argument("--only", default="", help="one of three_ops / three_ops_one_stream / three_ops_batched / import_swap_only / fused_autograd (profiling runs)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = WORKLOADS["xarm7_1280x720_8view"]
    rb = load_robot("xarm7")
    H, W, K = wl["H"], wl["W"], wl["K"]
    _, lp = make_views(rb, a.views)
    Tc = camera_Tc_c2b(radius=wl["radius"], lift=wl["lift"])
    out = {}
    for fusedflag, refsched in ((False, False), (False, "one_stream"), (False, "batched"), (False, True), (True, False)):
        name = "fused_autograd" if fusedflag else {True: "import_swap_only", "batched": "three_ops_batched",
                                                   "one_stream": "three_ops_one_stream", False: "three_ops"}[refsched]
        if a.only and a.only != name:
            continue
        cfg = Cfg()
        cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W
        cfg.model.rbsolver.init_Tc_c2b = perturb_pose(Tc).tolist()
        cfg.model.rbsolver.use_fused = fusedflag
        cfg.model.rbsolver.reference_schedule = refsched is True  # True: the reference's own statements, only the import swapped
        cfg.model.rbsolver.batched_ops = refsched == "batched"   # one call per op and step over all (view, link) images
        cfg.model.rbsolver.render_lanes = 1 if refsched == "one_stream" else a.lanes
        model = RBSolver(cfg, meshes=rb.meshes).to(dev)
        batch = {"mask": torch.zeros((a.views, H, W), device=dev), "link_poses": torch.tensor(lp, device=dev),
                 "K": torch.tensor(K, dtype=torch.float32, device=dev)[None].repeat(a.views, 1, 1)}
        tr = RBSolverTrainer(cfg, model, batch, graph=a.graph)
        for _ in range(3):
            tr.step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(a.steps):
            tr.step()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / a.steps
        out[name + ("_graph" if a.graph else "")] = {
            "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(a.views / dt, 1), "loss": round(float(tr.last_loss), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
