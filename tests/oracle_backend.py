"""TEST-ONLY stand-in for RBSolver's render step, driven by the CPU oracle, so that host logic (trainer, Adam, data
parallel exchange) can be exercised without a GPU.  Never imported by the product."""
import numpy as np
import torch
import torch.nn as nn

import helpers
from easyhec_amd import fused
from easyhec_amd.se3 import se3_exp_map, se3_log_map


class _OracleRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mvp, ref, scene):
        from oracle import oracle
        verts, tris, toff, voff = scene
        mask, loss, g = oracle.render_mask_loss(verts, tris, toff, voff, mvp.detach().numpy(), ref.numpy())
        ctx.save_for_backward(torch.from_numpy(g))
        return torch.from_numpy(mask), torch.from_numpy(loss)

    @staticmethod
    def backward(ctx, _gm, gl):
        (g,) = ctx.saved_tensors
        return g * gl[:, None, None, None], None, None


class OracleRBSolver(nn.Module):
    """Same parameterisation and forward contract as easyhec_amd.rb_solver.RBSolver, CPU, oracle-rendered."""

    def __init__(self, robot, init_Tc_c2b, H, W):
        super().__init__()
        self.scene = helpers.scene_arrays(robot)
        self.H, self.W = H, W
        init = torch.as_tensor(np.asarray(init_Tc_c2b), dtype=torch.float32)
        self.dof = nn.Parameter(se3_log_map(init[None].permute(0, 2, 1), eps=1e-5)[0])

    def forward(self, dps, with_outputs=True):
        Tc = se3_exp_map(self.dof[None]).permute(0, 2, 1)[0]
        mvp = fused.mvp_matrices(dps["K"][0], self.H, self.W, Tc, dps["link_poses"])
        mask, losses = _OracleRender.apply(mvp, dps["mask"].float(), self.scene)
        out = {"rendered_masks": mask} if with_outputs else {}
        return out, {"mask_loss": losses.mean()}
