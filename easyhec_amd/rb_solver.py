"""``RBSolver`` -- the rendering-based pose solver, mirroring
/root/reference/easyhec/modeling/models/rb_solve/rb_solver.py:15-96 (constructor, ``forward(dps)`` and its
``(output, loss_dict)`` return, metrics, ``history_ops`` / ``dof`` state-dict names).

forward(dps) consumes the dataset tensors of /root/reference/easyhec/data/datasets/xarm_real.py:69-84
(``mask [B,H,W]``, ``link_poses [B,L,4,4]``, ``K [B,3,3]``, ``Tc_c2b [B,4,4]``) and renders every frame x link.
With ``use_fused`` (default) the whole double loop + loss is one HIP kernel chain; with ``use_fused=False`` it is the
reference's own per-(frame, link) sequence of rasterize / interpolate / antialias calls."""
import contextlib

import numpy as np
import torch
import torch.nn as nn

from . import fused
from .mesh_io import load_mesh
from .renderer import NVDiffrastRenderer
from .se3 import se3_exp_map, se3_log_map

__all__ = ["RBSolver"]


class _LinkComposite(torch.autograd.Function):
    """``flip(stack(silhouettes).sum(0).clamp(max=1), dims=[0])`` (rb_solver.py:66-69 + nvdiffrast_renderer.py:47), the same
    forward ops in the same order; the backward computes the gradient image once -- flip, then clamp's pass-through mask
    (``sum <= 1``, as torch's clamp backward has it) -- and hands that one contiguous tensor to every link, where autograd's
    own chain (sum -> expand -> stack -> unbind) hands out stride-0 views that each consumer has to copy."""

    @staticmethod
    def forward(ctx, *silhouettes):
        total = torch.stack(silhouettes).sum(0)
        ctx.save_for_backward(total)
        ctx.n = len(silhouettes)
        return torch.flip(total.clamp(max=1), dims=[0])

    @staticmethod
    def backward(ctx, dy):
        (total,) = ctx.saved_tensors
        g = torch.where(total <= 1, torch.flip(dy, dims=[0]), 0.0)
        return (g,) * ctx.n


class RBSolver(nn.Module):
    def __init__(self, cfg, meshes=None):
        """cfg: :class:`easyhec_amd.config.Cfg` (fields of defaults.py ``model.rbsolver``).
        meshes: optional list of (vertices [V,3], faces [T,3]) replacing ``cfg.model.rbsolver.mesh_paths``."""
        super().__init__()
        self.total_cfg = cfg
        self.cfg = cfg.model.rbsolver
        self.dbg = cfg.dbg
        if meshes is None:
            meshes = [load_mesh(p) for p in self.cfg.mesh_paths]
        for link_idx, (vertices, faces) in enumerate(meshes):
            vertices = torch.as_tensor(np.asarray(vertices), dtype=torch.float32)
            faces = torch.as_tensor(np.asarray(faces), dtype=torch.int32)
            self.register_buffer(f"vertices_{link_idx}", vertices)
            self.register_buffer(f"faces_{link_idx}", faces)
        self.nlinks = len(meshes)
        # camera parameters (rb_solver.py:30-34)
        init_Tc_c2b = torch.as_tensor(np.asarray(self.cfg.init_Tc_c2b), dtype=torch.float32)
        init_dof = se3_log_map(init_Tc_c2b[None].permute(0, 2, 1), eps=1e-5, backend="opencv")[0]
        self.dof = nn.Parameter(init_dof, requires_grad=True)
        self.H, self.W = self.cfg.H, self.cfg.W
        self.renderer = None  # created on first forward, on the parameters' device (needs a HIP device)
        self._scene = None
        self.register_buffer("history_ops", torch.zeros(10000, 6))
        # next free row of history_ops.  The reference finds it on every forward as the first all-zero row
        # (rb_solver.py:50-51, one .item() sync per step); here it is a host counter that is re-derived from the buffer
        # (one sync) whenever it may be stale: after load_state_dict and after steps of the HIP launch chain, which
        # writes the rows itself (None = unknown).
        self._hist_n = 0
        self.auto_render_lanes = 1  # cfg.render_lanes = -1: what RBSolverTrainer(graph=True) sets to 2 (see _forward_three_ops)
        self._hist_dev = None  # device-side cursor (a [1] int64 tensor) while the step is being replayed from a graph
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, "_hist_n", None))

    def history_cursor(self):
        """First all-zero row of ``history_ops`` (rb_solver.py:50): where the next pose is recorded."""
        if self._hist_n is None:
            zero_rows = (self.history_ops == 0).all(dim=1).nonzero()
            self._hist_n = int(zero_rows[0, 0]) if zero_rows.numel() > 0 else self.history_ops.shape[0]
        return self._hist_n

    # -- device-side lazies -------------------------------------------------------------------------------------
    def _ensure_renderer(self):
        dev = self.dof.device
        if self.renderer is None or self.renderer.device != dev:
            if dev.type != "cuda":
                raise RuntimeError("RBSolver renders on a HIP device only: move the module with .cuda() first "
                                   "(there is no CPU render path)")
            plain = bool(getattr(self.cfg, "reference_schedule", False)) and not self.cfg.use_fused
            self.renderer = NVDiffrastRenderer([self.H, self.W], device=dev, plain=plain)
            self._scene = None
        return self.renderer

    def _ensure_scene(self):
        if self._scene is None:
            vs = [getattr(self, f"vertices_{i}") for i in range(self.nlinks)]
            fs = [getattr(self, f"faces_{i}") for i in range(self.nlinks)]
            self._scene = fused.LinkScene(vs, fs, self.dof.device)
        return self._scene

    def Tc_c2b(self):
        return se3_exp_map(self.dof[None]).permute(0, 2, 1)[0]

    def _forward_three_ops(self, renderer, Tc_c2b, link_poses, K, masks_ref):
        """The reference's own schedule (rb_solver.py:60-72): one rasterize / interpolate / antialias round trip per
        (frame, link) through the drop-in ops, links summed and clamped, SSE per frame, mean over frames."""
        per_frame_loss, per_frame_mask = [], []
        # The reference's schedule -- one rasterize / interpolate / antialias round trip per (frame, link) -- with the
        # small host-side products around it batched (results are the same dot products; the step is bound by its number
        # of launches, ~2 400 of them): rb_solver.py:63's Tc_c2b @ link_poses[bid, link] and nvdiffrast_renderer.py:35-37's
        # proj @ opencv2blender @ pose as two batched products for all (frame, link) pairs, transform_pos as one product per
        # link for all frames, and the vertical flip (nvdiffrast_renderer.py:47: a permutation, it commutes with the sum
        # over links and the clamp) once per frame instead of once per link.
        # Autograd nodes are launches too: unbind (backward = ONE stack) instead of indexing per (frame, link) (backward = a
        # zero fill + a copy + an add each), and _LinkComposite hands all links one contiguous gradient image.
        mvp_all = renderer.clip_matrices(K, Tc_c2b[None, None] @ link_poses)          # [B, L, 4, 4]
        pos_links = [renderer.clip_positions_batched(m, getattr(self, f"vertices_{k}"))     # [B, V_k, 4] per link
                     for k, m in enumerate(mvp_all.unbind(1))]
        pos_all = [p.unbind(0) for p in pos_links]
        # One lane (HIP stream + rasterizer context) per frame (modulo cfg.render_lanes): the renders of different frames do
        # not depend on each other, and each is a chain of small launches that leaves most of the GPU idle.  A frame's chain
        # -- its L renders, the sum over links, its loss -- goes to its lane whole, so a frame costs one fork and one join; the
        # backward of every op runs on the stream its forward ran on (autograd's rule) and overlaps the same way.
        n_lanes = int(getattr(self.cfg, "render_lanes", -1))
        n_lanes = min(self.auto_render_lanes if n_lanes < 0 else n_lanes, masks_ref.shape[0])
        lanes = renderer.link_lanes(n_lanes) if n_lanes > 1 else None
        here = torch.cuda.current_stream()
        if lanes:
            for k in range(self.nlinks):   # constants the renders share are made before the lanes part
                renderer.warm(getattr(self, f"vertices_{k}"), getattr(self, f"faces_{k}"))
            for lane, _ in lanes:
                lane.wait_stream(here)
                for p in pos_links:
                    p.record_stream(lane)   # made on the step's stream, read on the lanes (forward and backward)
        for frame in range(masks_ref.shape[0]):
            lane, glctx = lanes[frame % n_lanes] if lanes else (None, None)
            with (torch.cuda.stream(lane) if lanes else contextlib.nullcontext()):
                silhouettes = [
                    renderer.mask_from_clip(pos_all[k][frame][None], getattr(self, f"vertices_{k}"),
                                            getattr(self, f"faces_{k}"), flip=False, glctx=glctx)
                    for k in range(self.nlinks)
                ]
                composite = _LinkComposite.apply(*silhouettes)
                frame_loss = ((composite - masks_ref[frame].float()) ** 2).sum()
            if lanes:
                composite.record_stream(here)
                frame_loss.record_stream(here)
            per_frame_mask.append(composite)
            per_frame_loss.append(frame_loss)
        if lanes:
            for lane, _ in lanes:
                here.wait_stream(lane)
        return torch.stack(per_frame_mask), torch.stack(per_frame_loss).mean()

    def _batched_topology(self, B, dev):
        """Static part of the batched schedule for B frames: the triangles of every (frame, link) image shifted to that image's
        slice of the concatenated vertex array, the (start, count) range of every image (CPU int32, as dr.rasterize takes
        them), the all-ones colour and the edge topology of the concatenated triangle array.  Cached per (B, device)."""
        key = (B, str(dev))
        ent = getattr(self, "_batched_cache", None)
        if ent is not None and ent[0] == key:
            return ent[1]
        from . import dr
        vs = [getattr(self, f"vertices_{k}") for k in range(self.nlinks)]
        fs = [getattr(self, f"faces_{k}") for k in range(self.nlinks)]
        tris, ranges, voff, toff = [], [], 0, 0
        for _ in range(B):
            for v, f in zip(vs, fs):
                tris.append(f + voff)
                ranges.append((toff, int(f.shape[0])))
                voff += int(v.shape[0])
                toff += int(f.shape[0])
        tri = torch.cat(tris).to(torch.int32).contiguous()
        ent = {"tri": tri, "ranges": torch.tensor(ranges, dtype=torch.int32),
               "ones": torch.ones((voff, 1), dtype=torch.float32, device=dev),
               "topology": dr.antialias_construct_topology_hash(tri)}
        self._batched_cache = (key, ent)
        return ent

    def _forward_three_ops_batched(self, renderer, Tc_c2b, link_poses, K, masks_ref):
        """``batched_ops``: the reference's three ops called once each over all B x L (frame, link) images (range mode), then
        rb_solver.py:66-72's sum over links, clamp, SSE per frame and mean -- on [B, L, H, W] at once."""
        from . import dr
        B = masks_ref.shape[0]
        st = self._batched_topology(B, masks_ref.device)
        mvp_all = renderer.clip_matrices(K, Tc_c2b[None, None] @ link_poses)                     # [B, L, 4, 4]
        per_link = [renderer.clip_positions_batched(m, getattr(self, f"vertices_{k}"))           # L x [B, V_k, 4]
                    for k, m in enumerate(mvp_all.unbind(1))]
        # ([B, V_k, 4] blocks side by side along the vertex axis ARE the image-major concatenation -- one cat of L tensors and a
        #  view, where round 5 concatenated B x L slices: 64 copies forward and a fill + an add per slice backward, a quarter of
        #  this step's launches)
        pos = torch.cat(per_link, dim=1).reshape(-1, 4)                                          # [sum V, 4], image-major
        rast, _ = dr.rasterize(renderer.glctx, pos, st["tri"], [self.H, self.W], ranges=st["ranges"], grad_db=False)
        color, _ = dr.interpolate(st["ones"], dr.carry_tile_flags(rast, rast.detach()), st["tri"])
        aa = dr.antialias(color, rast, pos, st["tri"], topology_hash=st["topology"])                 # [B L, H, W, 1]
        si = aa.view(B, self.nlinks, self.H, self.W)
        masks = torch.flip(si.sum(1).clamp(max=1), dims=[1])                                     # row 0 = top
        loss = ((masks - masks_ref.float()) ** 2).sum(dim=(1, 2)).mean()
        return masks, loss

    def _forward_per_call(self, renderer, Tc_c2b, link_poses, K, masks_ref):
        """``cfg.model.rbsolver.reference_schedule``: the schedule of rb_solver.py:58-71 with no host-side batching -- one
        pose product and one ``render_mask`` call (flip included) per (frame, link), the links stacked, summed and clamped
        per frame, a squared-error sum per frame, their mean."""
        meshes = [(getattr(self, f"vertices_{k}"), getattr(self, f"faces_{k}")) for k in range(self.nlinks)]
        frames, sse = [], []
        for f, target in enumerate(masks_ref):
            layers = torch.stack([renderer.render_mask(v, t, K=K, object_pose=Tc_c2b @ link_poses[f, k])
                                  for k, (v, t) in enumerate(meshes)])
            frames.append(layers.sum(0).clamp(max=1))
            sse.append(((frames[-1] - target.float()) ** 2).sum())
        return torch.stack(frames), torch.stack(sse).mean()

    # -- forward -------------------------------------------------------------------------------------------------
    def forward(self, dps, with_outputs=True):
        assert dps.get("global_step", 0) == 0
        renderer = self._ensure_renderer()
        if self._hist_dev is not None:
            # captured step (RBSolverTrainer(graph=True)): a host counter would be frozen into the graph, so the cursor
            # lives on the device and every replay records its pose in the next row (the last row absorbs an overrun)
            self.history_ops.index_copy_(0, self._hist_dev, self.dof.detach()[None])
            self._hist_dev.add_(1).clamp_(max=self.history_ops.shape[0] - 1)
        else:
            put_id = self.history_cursor()  # rb_solver.py:50-51 without the per-step .item() sync
            if put_id < self.history_ops.shape[0]:
                self.history_ops[put_id] = self.dof.detach()
                self._hist_n = put_id + 1
        Tc_c2b = self.Tc_c2b()
        masks_ref = dps["mask"]
        link_poses = dps["link_poses"]
        K = dps["K"][0]
        batch_size = masks_ref.shape[0]

        if self.cfg.use_fused:
            scene = self._ensure_scene()
            mvp = fused.mvp_matrices(K, self.H, self.W, Tc_c2b, link_poses)
            rendered, losses = fused.render_mask_loss(renderer.glctx, scene, mvp, masks_ref.float(),
                                                      want_mask=with_outputs)
            loss = losses.mean()
            all_frame_all_link_si = rendered if with_outputs else None
        elif renderer.plain:
            rendered, loss = self._forward_per_call(renderer, Tc_c2b, link_poses, K, masks_ref)
            all_frame_all_link_si = rendered
        elif getattr(self.cfg, "batched_ops", False):
            rendered, loss = self._forward_three_ops_batched(renderer, Tc_c2b, link_poses, K, masks_ref)
            all_frame_all_link_si = rendered
        else:
            rendered, loss = self._forward_three_ops(renderer, Tc_c2b, link_poses, K, masks_ref)
            all_frame_all_link_si = rendered

        output = {}
        if with_outputs:
            output = {"rendered_masks": all_frame_all_link_si,
                      "ref_masks": masks_ref,
                      "error_maps": (all_frame_all_link_si.detach() - masks_ref.float()).abs()}
        # metrics (rb_solver.py:79-92): differences of log coordinates, cm and degrees
        gt_dof6 = dps.get("gt_dof6")
        if gt_dof6 is None and "Tc_c2b" in dps and with_outputs:
            gt_Tc_c2b = dps["Tc_c2b"][0]
            if not torch.allclose(gt_Tc_c2b, torch.eye(4, device=gt_Tc_c2b.device)):
                gt_dof6 = se3_log_map(gt_Tc_c2b[None].permute(0, 2, 1), backend="opencv")[0]
        if gt_dof6 is not None:
            trans_err = ((gt_dof6[:3] - self.dof[:3]) * 100).abs()
            rot_err = (gt_dof6[3:] - self.dof[3:]).abs().max() / np.pi * 180
            output["metrics"] = {"err_x": trans_err[0], "err_y": trans_err[1], "err_z": trans_err[2],
                                 "err_trans": trans_err.norm(), "err_rot": rot_err}
        if with_outputs:
            output["tsfm"] = se3_exp_map(self.dof[None].detach().cpu()).permute(0, 2, 1)[0]
        loss_dict = {"mask_loss": loss}
        return output, loss_dict
