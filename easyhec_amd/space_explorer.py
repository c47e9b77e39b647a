"""Next-best-view scoring of the space explorer.

Mirrors the render + variance core of ``SpaceExplorer.forward``
(/root/reference/easyhec/modeling/models/rb_solve/space_explorer.py:87-96,152-185): sample camera poses from the
optimisation history, render the robot mask for every candidate joint configuration under every sampled pose, score a
configuration by the summed per-pixel variance of its masks, return the best one.  The reference spends 10 000
nvdiffrast renders (``render_api.nvdiffrast_parallel_render_xarm_api``) + a torch.var per exploration round on this;
here it is one call of ``ehr_mask_variance`` (include/ehr.h), which returns the exact integer numerator.

Motion planning, self-collision and workspace checks of the reference (pymp / SAPIEN, space_explorer.py:104-150) are
out of scope: pass their outcome as ``valid`` and rejected candidates score 0 exactly as the reference's
``variances.append(0)`` does."""
import ctypes

import numpy as np
import torch

from . import _lib, dr
from .fused import LinkScene, mvp_matrices
from .se3 import se3_exp_map

__all__ = ["mask_variance", "sample_history_poses", "SpaceExplorer"]


def mask_variance(glctx, scene, mvp, H, W, return_counts=False, chunk_views=0):
    """mvp [Q,S,L,4,4] float32 on the HIP device -> (var_sum [Q] float32, score [Q] int64[, counts [Q,H,W] uint8]).

    ``score[q] = sum_px c (S - c)`` is exact; ``var_sum = score / (S (S - 1))`` is what
    ``torch.var(masks.reshape(S, -1).float(), dim=0).sum()`` (space_explorer.py:164) evaluates in floating point."""
    if mvp.dim() != 5 or mvp.shape[-2:] != (4, 4):
        raise ValueError("mvp must be [Q,S,L,4,4]")
    if not mvp.is_cuda:
        raise RuntimeError("mask_variance: tensors must live on the HIP device (there is no CPU path)")
    Q, S, L = mvp.shape[:3]
    if L != scene.num_links:
        raise ValueError(f"mvp has {L} links, the scene {scene.num_links}")
    mvp = mvp.contiguous().float()
    score = torch.empty((Q,), dtype=torch.int64, device=mvp.device)
    counts = torch.empty((Q, H, W), dtype=torch.uint8, device=mvp.device) if return_counts else None
    with torch.cuda.device(mvp.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().ehr_mask_variance(glctx.handle, _lib.ptr(scene.verts), _lib.ptr(scene.tris),
                                                _lib.ptr(scene.vert_link), _lib.ptr(mvp), Q, S, L, scene.num_verts,
                                                scene.num_tris, H, W, _lib.ptr(score), _lib.ptr(counts),
                                                int(chunk_views), stream), "ehr_mask_variance")
    if S > 1:
        var_sum = (score.double() / float(S * (S - 1))).float()
    else:
        var_sum = torch.full((Q,), float("nan"), device=mvp.device)  # torch.var of one sample
    return (var_sum, score, counts) if return_counts else (var_sum, score)


def sample_history_poses(history_dof6, start=200, sample=10, generator=None):
    """space_explorer.py:58-61,87-91: drop the unused (all-zero) rows of ``history_ops``, skip the first ``start``
    iterations, draw ``sample`` poses without replacement, return Tc_c2b [sample,4,4]."""
    h = torch.as_tensor(history_dof6, dtype=torch.float32).cpu()
    h = h[~(h == 0).all(dim=1)]
    h = h[start:]
    if h.shape[0] < sample:
        raise ValueError(f"only {h.shape[0]} history rows after start={start}; need {sample}")
    idx = torch.randperm(h.shape[0], generator=generator)[:sample]
    return se3_exp_map(h[idx]).permute(0, 2, 1).contiguous()


class SpaceExplorer:
    """Scores candidate joint configurations for one robot.

    robot: :class:`easyhec_amd.robot.Robot` (meshes + URDF chain); K [3,3]; (height, width) as
    ``cfg.model.space_explorer.{height,width,K}`` (easyhec/config/defaults.py:88-92)."""

    def __init__(self, robot, K, height, width, device="cuda:0", chunk_views=0):
        self.robot = robot
        self.device = torch.device(device)
        self.H, self.W = int(height), int(width)
        self.K = torch.as_tensor(np.asarray(K), dtype=torch.float32)
        self.glctx = dr.RasterizeCudaContext(device=self.device)
        self.scene = LinkScene([v for v, _ in robot.meshes], [f for _, f in robot.meshes], self.device)
        self.chunk_views = chunk_views

    def link_poses(self, qposes):
        """FK of every candidate: [Q,L,4,4] base<-link (sapien_kin.py:26-30 through the URDF chain)."""
        return torch.as_tensor(self.robot.link_poses_batch(np.asarray(qposes)), dtype=torch.float32)

    def mvp(self, Tc_c2b, link_poses):
        """[Q,S,L,4,4] = proj(K) @ opencv2blender @ Tc_c2b[s] @ link_poses[q,l]."""
        Tc = torch.as_tensor(Tc_c2b, dtype=torch.float32).to(self.device)
        lp = torch.as_tensor(link_poses, dtype=torch.float32).to(self.device)
        S, Q = Tc.shape[0], lp.shape[0]
        out = [mvp_matrices(self.K.to(self.device), self.H, self.W, Tc[s], lp) for s in range(S)]  # each [Q,L,4,4]
        return torch.stack(out, dim=1).contiguous()

    def score(self, qposes, Tc_c2b, valid=None):
        """variances [Q] float32 (0 for candidates rejected by ``valid``), as space_explorer.py:98-166 builds them."""
        lp = self.link_poses(qposes)
        var_sum, _ = mask_variance(self.glctx, self.scene, self.mvp(Tc_c2b, lp), self.H, self.W,
                                   chunk_views=self.chunk_views)
        if valid is not None:
            var_sum = torch.where(torch.as_tensor(valid, device=var_sum.device, dtype=torch.bool), var_sum,
                                  torch.zeros_like(var_sum))
        return var_sum

    def forward(self, qposes, history_dof6, start=200, sample=10, valid=None, generator=None):
        """The outputs dictionary of SpaceExplorer.forward (space_explorer.py:167-200) minus the motion plan."""
        Tc = sample_history_poses(history_dof6, start, sample, generator)
        variances = self.score(qposes, Tc, valid).cpu()
        top_ids = variances.argsort(descending=True)
        tid = top_ids[0]
        if not variances[tid] > 0:
            raise RuntimeError("no valid qpos found! Consider to increase the number of sampled qpos, "
                               "or increase the max_dist.")
        pos = variances[variances > 0]
        return {"qpos": np.asarray(qposes)[tid], "qpos_idx": tid, "variance": variances[tid], "var_max": pos.max(),
                "var_min": pos.min(), "var_mean": pos.mean(), "variances": variances}
