"""Optimisation loop of the pose solver, mirroring the step of
/root/reference/easyhec/trainer/rbsolver.py:29-43 (zero_grad -> forward -> backward -> [clip] -> Adam.step) with the
optimiser of /root/reference/easyhec/solver/build.py:12-29 (Adam, lr = max_lr, weight_decay 5e-4, constant LR).

Data parallelism: views are independent given ``dof``; each rank holds a contiguous block of views
and the ranks exchange ONE 8-float all-reduce per step -- [d loss/d dof (6), sum of per-frame losses, frame count] --
instead of the reference's DDP all-reduce plus up to six 1-float ``reduce_loss`` calls (trainer/rbsolver.py:45-49).
The reduced gradient is the gradient of the mean per-frame loss over ALL views, which equals the reference's DDP
average whenever the shards are equal."""
import os
import time

import torch
import torch.distributed as dist

from .se3 import se3_log_map

__all__ = ["RBSolverTrainer", "make_optimizer", "shard_views"]


def make_optimizer(cfg, model, capturable=False):
    """solver/build.py:12-29: one param group per parameter, weight decay on everything not named *bias*."""
    params = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        params += [{"params": [value], "lr": cfg.solver.max_lr, "weight_decay": cfg.solver.weight_decay}]
    if cfg.solver.optimizer == "Adam":
        return torch.optim.Adam(params, cfg.solver.max_lr, capturable=capturable)
    if cfg.solver.optimizer == "SGD":
        return torch.optim.SGD(params, cfg.solver.max_lr, momentum=0.9)
    raise NotImplementedError(cfg.solver.optimizer)


def shard_views(n_views, rank, world_size):
    """Contiguous block of view indices owned by ``rank`` (DistributedSampler(shuffle=False)-like, base.py:354-366,
    but contiguous so a rank's reference masks stay one slab in HBM)."""
    per = (n_views + world_size - 1) // world_size
    lo = min(rank * per, n_views)
    hi = min(lo + per, n_views)
    return lo, hi


class RBSolverTrainer:
    def __init__(self, cfg, model, batch, process_group=None, fast=False, graph=False, rccl=None):
        """batch: dict of this rank's device tensors (mask, link_poses, K, Tc_c2b) -- the single batch the reference
        builds with batch_size=100 >= #frames (configs/xarm7/example.yaml:45).
        fast: run the step as the fixed HIP launch chain of :class:`easyhec_amd.fast.FusedPoseStep` (same arithmetic,
        no autograd / torch glue); graph: additionally replay it as a captured hipGraph (fast) or, without ``fast``, record
        the torch-autograd step -- the three drop-in ops per (view, link), backward, torch Adam -- in a
        ``torch.cuda.CUDAGraph`` and replay that (see :meth:`_capture_autograd_step`)."""
        self.cfg = cfg
        self.model = model
        self.batch = dict(batch)
        self.batch.setdefault("global_step", 0)
        self.optimizer = make_optimizer(cfg, model, capturable=bool(graph and not fast))
        self.global_steps = 0
        self.pg = process_group
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1
        self.last_loss = None
        self._reported = 0
        self.fast = None
        self._cuda_graph = None
        if graph and not fast:
            self._capture_autograd_step()
        if fast:
            from .fast import FusedPoseStep
            if cfg.solver.do_grad_clip or cfg.solver.optimizer != "Adam":
                raise ValueError("fast path implements the reference's default solver only (Adam, no gradient clipping)")
            self.fast = FusedPoseStep(model, self.batch, lr=cfg.solver.max_lr, weight_decay=cfg.solver.weight_decay,
                                      process_group=process_group, rccl=rccl)
            if graph and (not self.distributed or self.fast.rccl):
                self.fast.capture()
        if "Tc_c2b" in self.batch and "gt_dof6" not in self.batch:
            gt = self.batch["Tc_c2b"][0]
            if not torch.allclose(gt.cpu(), torch.eye(4)):  # decided once, not per step (rb_solver.py:80)
                self.batch["gt_dof6"] = se3_log_map(gt[None].permute(0, 2, 1), backend="opencv")[0]

    def _capture_autograd_step(self):
        """graph=True without fast: the reference-shaped step -- RBSolver.forward through the three drop-in ops (or the
        fused op) under torch autograd, loss.backward(), torch.optim.Adam(capturable=True).step() -- recorded once into
        a ``torch.cuda.CUDAGraph`` and replayed: ~1 500 Python-driven launches per step become one graph launch.  The
        ops allocate their outputs from the graph's private pool and never synchronise (the drop-in rasterizer records
        the call with its queue storage as the warm-up steps sized it; a frame that outgrows it is still exact, only
        slower).  Single process only; the pose history cursor moves to the device (RBSolver._hist_dev)."""
        if self.distributed:
            raise RuntimeError("graph capture of the autograd step is single-process only")
        if self.cfg.solver.do_grad_clip:
            # (the eager step clips between backward() and step(); recording that needs the clip inside the capture, which
            #  nobody has asked for: refuse rather than silently drop it -- the fast path refuses the same configuration)
            raise ValueError("graph=True records the reference's default solver only (no gradient clipping)")
        m = self.model
        dev = m.dof.device
        # render_lanes = -1 (automatic): the per-image three-op schedule goes to two streams when it is replayed from a graph
        # (two parallel branches: 4.0 -> 3.1 ms per step at 8 views; three or more are slower than two under ROCm 7.2's graph
        # executor, and issued from Python the extra stream bookkeeping costs more than the overlap returns).  Set before the
        # warm-up steps: the lanes' rasterizer contexts must exist before the capture begins.
        m.auto_render_lanes = 2
        if "gt_dof6" not in self.batch and "Tc_c2b" in self.batch:
            gt = self.batch["Tc_c2b"][0]
            if not torch.allclose(gt.cpu(), torch.eye(4)):
                self.batch["gt_dof6"] = se3_log_map(gt[None].permute(0, 2, 1), backend="opencv")[0]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        self._static_loss = torch.zeros((), device=dev)
        snap_dof = m.dof.detach().clone()
        snap_hist = m.history_ops.clone()
        hist_n = m.history_cursor()
        m._hist_dev = torch.full((1,), hist_n, dtype=torch.int64, device=dev)
        with torch.cuda.stream(side):
            for _ in range(3):  # warm-up: sizes every scratch buffer, builds the caches, initialises the Adam state
                self.optimizer.zero_grad(set_to_none=False)
                _, ld = m(self.batch, with_outputs=False)
                sum(ld.values()).backward()
                self.optimizer.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # the warm-up steps must not count: parameters, Adam moments, step counter and history go back
        with torch.no_grad():
            m.dof.copy_(snap_dof)
            m.history_ops.copy_(snap_hist)
            m._hist_dev.fill_(hist_n)
            for st in self.optimizer.state.values():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.zero_()
        g = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=False)
        with torch.cuda.graph(g):
            self.optimizer.zero_grad(set_to_none=False)
            _, ld = m(self.batch, with_outputs=False)
            loss = sum(ld.values())
            loss.backward()
            self.optimizer.step()
            self._static_loss.copy_(loss.detach())
        # the capture pass itself does not execute: state is still the snapshot's
        self._cuda_graph = g

    # one optimisation step == one "epoch" of the reference (trainer/rbsolver.py:29-43)
    def step(self, with_outputs=False):
        if self._cuda_graph is not None and not with_outputs:
            self._cuda_graph.replay()
            self.model._hist_n = None  # the device cursor moved; the host one is stale
            self.global_steps += 1
            # (a copy: the graph writes the same tensor again at the next replay, and callers collect losses lazily)
            self.last_loss = self._static_loss.clone()
            return {}, self.last_loss
        if self.fast is not None:
            # ONE optimiser state whatever is asked for: the launch chain also produces the rendered masks
            loss_value = self.fast.step(want_mask=with_outputs)[0]
            self.global_steps += 1
            self.last_loss = loss_value
            return (self._fast_outputs() if with_outputs else {}), loss_value
        self.optimizer.zero_grad(set_to_none=False)
        output, loss_dict = self.model(self.batch, with_outputs=with_outputs)
        loss = sum(v for v in loss_dict.values())
        n_local = self.batch["mask"].shape[0]
        if self.distributed:
            # local mean -> local sum so that the reduced quantity is the global mean's gradient
            (loss * n_local).backward()
            dof = self.model.dof
            buf = torch.cat([dof.grad.reshape(-1), (loss.detach() * n_local).reshape(1),
                             torch.full((1,), float(n_local), dtype=dof.dtype, device=dof.device)])
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            dof.grad.copy_((buf[:6] / buf[7]).reshape(dof.grad.shape))
            loss_value = buf[6] / buf[7]
        else:
            loss.backward()
            loss_value = loss.detach()
        if self.cfg.solver.do_grad_clip:
            if self.cfg.solver.grad_clip_type == "norm":
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.cfg.solver.grad_clip)
            else:
                torch.nn.utils.clip_grad_value_(self.model.parameters(), self.cfg.solver.grad_clip)
        self.optimizer.step()
        self.global_steps += 1
        self.last_loss = loss_value
        return output, loss_value

    def _fast_outputs(self):
        """The ``output`` dict of RBSolver.forward (rb_solver.py:73-94) for the step the launch chain just took; the pose
        it refers to is the one the step STARTED from (the row the chain recorded in history_ops), as in the reference,
        where forward() runs before optimizer.step()."""
        f, m = self.fast, self.model
        rendered = f.mask.clone()
        ref = self.batch["mask"]
        out = {"rendered_masks": rendered, "ref_masks": ref, "error_maps": (rendered - ref.float()).abs()}
        row = (f.hist_row.long() - 1).clamp(0, m.history_ops.shape[0] - 1)
        dof_before = m.history_ops[row][0]
        gt_dof6 = self.batch.get("gt_dof6")
        if gt_dof6 is not None:
            import numpy as np
            trans_err = ((gt_dof6[:3] - dof_before[:3]) * 100).abs()
            rot_err = (gt_dof6[3:] - dof_before[3:]).abs().max() / np.pi * 180
            out["metrics"] = {"err_x": trans_err[0], "err_y": trans_err[1], "err_z": trans_err[2],
                              "err_trans": trans_err.norm(), "err_rot": rot_err}
        from .se3 import se3_exp_map
        out["tsfm"] = se3_exp_map(dof_before[None].detach().cpu()).permute(0, 2, 1)[0]
        return out

    def fit(self, num_steps=None, log=None):
        """Runs ``num_epochs`` EFFECTIVE steps (base.py:161 ``fit``); returns the list of logged (step, loss) pairs.
        A step the launch chain reports instead of taking (NaN loss; dof and Adam untouched: a slot-limited plan that
        overflowed, a view that needs the general-triangle pass) does not count: the chain recovers by itself
        (FusedPoseStep._poll, every few steps, or here at a logged step) and the lost iterations are run again, so the
        solve takes exactly the steps it was asked for -- on every rank alike (Adam's step counter is replicated)."""
        n = self.cfg.solver.num_epochs if num_steps is None else num_steps
        history = []
        begin = time.time()
        base_steps = self.global_steps
        start = self.fast.steps_done if self.fast is not None else 0
        remaining, rounds = n, 0
        while remaining > 0:
            for it in range(remaining):
                last = it == remaining - 1
                do_log = (self.global_steps % self.cfg.solver.log_interval == 0) or last
                _, loss = self.step(with_outputs=False)
                if do_log:
                    lv = float(loss)
                    if lv != lv and self.fast is not None:
                        what = self.fast.recover_from_overflow()
                        if what:
                            self.fast.recoveries.append(what)
                        else:
                            from . import fused
                            fused.check_status(self.fast.glctx)  # raises with the context's message
                        continue
                    history.append((self.global_steps, lv))
                    if log is not None:
                        log(f"step {self.global_steps} mask_loss {lv:.4f} elapsed {time.time() - begin:.2f}s")
            if self.fast is None:
                break
            done = self.fast.steps_done - start  # (synchronises: once per fit, twice if a step was reported)
            if log is not None:
                while self._reported < len(self.fast.recoveries):
                    what = self.fast.recoveries[self._reported]
                    self._reported += 1
                    log("job slots overflowed; planned again with a slot per (view, link, tile)" if what == "job slots" else
                        "triangles at the near plane; the general-triangle pass joins the chain")
            self.global_steps = base_steps + done
            remaining = n - done
            if remaining > 0:
                rounds += 1
                if rounds > 4:
                    from . import fused
                    fused.check_status(self.fast.glctx)
                    raise RuntimeError(f"fit: {remaining} of {n} steps keep being reported as not taken (NaN loss)")
                what = self.fast.recover_from_overflow()
                if what:
                    self.fast.recoveries.append(what)
        return history

    # checkpoint in the reference's layout: ckpt['model']['dof'] / ['history_ops'] (trainer/rbsolver.py:95-114)
    def save(self, path):
        """``model`` (dof, history_ops, meshes), counters and the optimiser that actually stepped: the launch chain's
        Adam state in fast mode (exp_avg, exp_avg_sq, step), torch's otherwise -- same layout either way."""
        opt = self.fast.state_dict() if self.fast is not None else self.optimizer.state_dict()
        d = {"model": self.model.state_dict(), "epoch": self.global_steps, "best_val_loss": float("inf"),
             "global_steps": self.global_steps, "optimizer": opt}
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(d, path)

    def resume(self, path):
        """base.py:388-440: restores the model, the optimiser state and the counters.  The next step continues the Adam
        bias correction where it stopped and appends to ``history_ops`` at the first all-zero row (rb_solver.py:50-51)
        instead of overwriting it from row 0."""
        d = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(d["model"])  # also invalidates the model's history cursor
        if self.fast is not None:
            # the history row always comes from the restored model (first all-zero row); the Adam state only from a
            # saved optimiser -- without one the solve continues with a fresh Adam, like the reference's load_model path
            self.fast.hist_row.fill_(int(self.model.history_cursor()))
            self.fast.step_t.zero_()
            self.fast.exp_avg.zero_()
            self.fast.exp_avg_sq.zero_()
            if "optimizer" in d:
                self.fast.load_state_dict(d["optimizer"])
            self.fast.mark_counts()
        elif self._cuda_graph is not None:
            # The captured graph updates the tensors it was recorded with: the restored state goes INTO them (a new state
            # dict would leave the graph stepping the old moments, and save() writing tensors no replay touches), and the
            # device-side history cursor moves to the restored history's first free row.
            m = self.model
            with torch.no_grad():
                m._hist_dev.fill_(int(m.history_cursor()))
                st = self.optimizer.state[m.dof]
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()  # no saved optimiser: a fresh Adam, like the reference's load_model path
                saved = d.get("optimizer", {}).get("state", {})
                if len(saved) > 0:
                    first = saved[sorted(saved.keys())[0]]
                    for k in ("exp_avg", "exp_avg_sq"):
                        st[k].copy_(torch.as_tensor(first[k], dtype=st[k].dtype).reshape(st[k].shape))
                    st["step"].fill_(float(torch.as_tensor(first["step"]).reshape(-1)[0]))
        elif "optimizer" in d and len(d["optimizer"].get("state", {})) > 0:
            sd = d["optimizer"]
            first = sd["state"][sorted(sd["state"].keys())[0]]
            if "param_groups" in sd and all(torch.is_tensor(v) or not isinstance(v, (list, tuple)) for v in first.values()):
                # a torch-shaped state dict (base.py:388-440 does exactly this): moments, step AND param_groups (lr);
                # works for every optimiser make_optimizer builds (Adam's exp_avg / SGD's momentum_buffer)
                sd = {"state": {k: {n: (torch.as_tensor(t).reshape(()).float() if n == "step" else t)
                                    for n, t in st.items()} for k, st in sd["state"].items()},
                      "param_groups": sd["param_groups"]}
                try:
                    self.optimizer.load_state_dict(sd)
                    sd = None
                except (ValueError, KeyError):
                    pass  # saved by a different optimiser type / group layout: fall through to the moment copy
            if sd is not None and {"exp_avg", "exp_avg_sq", "step"} <= set(first.keys()):
                p = self.model.dof  # the launch chain's Adam state into torch.optim.Adam
                self.optimizer.state[p] = {
                    "step": torch.as_tensor(first["step"], dtype=torch.float32).reshape(()).clone(),
                    "exp_avg": torch.as_tensor(first["exp_avg"], dtype=p.dtype).reshape(p.shape).to(p.device).clone(),
                    "exp_avg_sq": torch.as_tensor(first["exp_avg_sq"], dtype=p.dtype).reshape(p.shape).to(p.device).clone()}
        self.global_steps = d.get("global_steps", 0)
