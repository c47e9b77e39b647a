"""Child process of tests/test_gpu_fused.py::test_franka_64_views_1080p_under_a_2gb_scratch_budget: the scratch budget is
an environment variable the library reads once, so it needs a process of its own.  Franka, 64 views at 1920x1080
(576 (view, link) units: more than one pass of the chain holds, and ~11 GB of scratch in one piece), one call of the fused
op without mask output under the budget the parent sets (EHR_VB_SCRATCH_MB); prints the device memory the context took and every view's loss /
gradient checksum, plus the same views 0-2 from a 3-view call for comparison."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import helpers  # noqa: E402


def main():
    from easyhec_amd import dr, fused
    from easyhec_amd.robot import load_robot
    from easyhec_amd.synthetic import WORKLOADS, camera_Tc_c2b, make_views, perturb_pose
    dev = torch.device("cuda:0")
    fr = load_robot("franka")
    wl = WORKLOADS["franka_1920x1080_16view"]
    H, W, K, B = wl["H"], wl["W"], wl["K"], 64
    _, lp = make_views(fr, B, seed=0)
    mvp = helpers.mvp_numpy(K, H, W, perturb_pose(camera_Tc_c2b(radius=wl["radius"], lift=wl["lift"])), lp)
    scene = fused.LinkScene([v for v, _ in fr.meshes], [f for _, f in fr.meshes], dev)
    mvp_t = torch.tensor(mvp, device=dev)
    ref = torch.zeros((B, H, W), device=dev)
    ref[:, ::7, ::5] = 1.0
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    ctx = dr.RasterizeCudaContext()
    def call(c, m, r):
        m = m.clone().requires_grad_(True)
        _, ls = fused.render_mask_loss(c, scene, m, r, want_mask=False)
        ls.sum().backward()
        torch.cuda.synchronize()
        fused.check_status(c)
        return ls.detach(), m.grad

    loss, grad = call(ctx, mvp_t, ref)
    taken = free0 - torch.cuda.mem_get_info()[0]
    ctx3 = dr.RasterizeCudaContext()
    loss3, grad3 = call(ctx3, mvp_t[:3].contiguous(), ref[:3].contiguous())
    print(json.dumps({"scratch_bytes": int(taken), "finite": bool(torch.isfinite(loss).all() and torch.isfinite(grad).all()),
                      "loss_min": float(loss.min()), "same_loss": bool((loss[:3] == loss3).all()),
                      "same_grad": bool((grad[:3] == grad3).all()), "views": int(loss.numel())}))


if __name__ == "__main__":
    main()
