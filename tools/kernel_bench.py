"""Stage timings of the fused op on the bench workload (no optimiser): python tools/kernel_bench.py [views]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from easyhec_amd import dr, fused
from easyhec_amd.robot import load_robot
from easyhec_amd.synthetic import WORKLOADS, camera_Tc_c2b, make_views, perturb_pose
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rb = load_robot("xarm7")
wl = WORKLOADS["xarm7_1280x720_8view"]
H, W, K = wl["H"], wl["W"], wl["K"]
_, lp = make_views(rb, B)
Tc = camera_Tc_c2b(radius=wl["radius"], lift=wl["lift"])
mvp = torch.tensor(helpers.mvp_numpy(K, H, W, perturb_pose(Tc), lp), device=dev, requires_grad=True)
ctx = dr.RasterizeCudaContext()
scene = fused.LinkScene([v for v, _ in rb.meshes], [f for _, f in rb.meshes], dev)
ref = torch.zeros((B, H, W), device=dev)
for _ in range(5):
    fused.render_mask_loss(ctx, scene, mvp, ref)
fused.check_status(ctx)
fused.set_timing(ctx, True)
for _ in range(50):
    fused.render_mask_loss(ctx, scene, mvp, ref)
ms, n = fused.read_timing(ctx)
print(os.environ.get("EHR_TILE_GRID_MULT", "-"), {k: round(v / n * 1e3, 1) for k, v in ms.items()}, "us per call", flush=True)
fused.check_status(ctx)
