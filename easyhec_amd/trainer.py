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


def make_optimizer(cfg, model):
    """solver/build.py:12-29: one param group per parameter, weight decay on everything not named *bias*."""
    params = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        params += [{"params": [value], "lr": cfg.solver.max_lr, "weight_decay": cfg.solver.weight_decay}]
    if cfg.solver.optimizer == "Adam":
        return torch.optim.Adam(params, cfg.solver.max_lr)
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
    def __init__(self, cfg, model, batch, process_group=None, fast=False, graph=False):
        """batch: dict of this rank's device tensors (mask, link_poses, K, Tc_c2b) -- the single batch the reference
        builds with batch_size=100 >= #frames (configs/xarm7/example.yaml:45).
        fast: run the step as the fixed HIP launch chain of :class:`easyhec_amd.fast.FusedPoseStep` (same arithmetic,
        no autograd / torch glue); graph: additionally replay it as a captured hipGraph."""
        self.cfg = cfg
        self.model = model
        self.batch = dict(batch)
        self.batch.setdefault("global_step", 0)
        self.optimizer = make_optimizer(cfg, model)
        self.global_steps = 0
        self.pg = process_group
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1
        self.last_loss = None
        self.fast = None
        if fast:
            from .fast import FusedPoseStep
            if cfg.solver.do_grad_clip or cfg.solver.optimizer != "Adam":
                raise ValueError("fast path implements the reference's default solver only (Adam, no gradient clipping)")
            self.fast = FusedPoseStep(model, self.batch, lr=cfg.solver.max_lr, weight_decay=cfg.solver.weight_decay,
                                      process_group=process_group)
            if graph and not self.distributed:
                self.fast.capture()
        if "Tc_c2b" in self.batch and "gt_dof6" not in self.batch:
            gt = self.batch["Tc_c2b"][0]
            if not torch.allclose(gt.cpu(), torch.eye(4)):  # decided once, not per step (rb_solver.py:80)
                self.batch["gt_dof6"] = se3_log_map(gt[None].permute(0, 2, 1), backend="opencv")[0]

    # one optimisation step == one "epoch" of the reference (trainer/rbsolver.py:29-43)
    def step(self, with_outputs=False):
        if self.fast is not None and not with_outputs:
            loss_value = self.fast.step()[0]
            self.global_steps += 1
            self.last_loss = loss_value
            return {}, loss_value
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

    def fit(self, num_steps=None, log=None):
        """Runs ``num_epochs`` steps (base.py:161 ``fit``); returns the list of logged (step, loss) pairs."""
        n = self.cfg.solver.num_epochs if num_steps is None else num_steps
        history = []
        begin = time.time()
        for it in range(n):
            do_log = (it % self.cfg.solver.log_interval == 0) or it == n - 1
            _, loss = self.step(with_outputs=False)
            if do_log:
                lv = float(loss)
                history.append((self.global_steps, lv))
                if log is not None:
                    log(f"step {self.global_steps} mask_loss {lv:.4f} elapsed {time.time() - begin:.2f}s")
        return history

    # checkpoint in the reference's layout: ckpt['model']['dof'] / ['history_ops'] (trainer/rbsolver.py:95-114)
    def save(self, path):
        d = {"model": self.model.state_dict(), "epoch": self.global_steps, "best_val_loss": float("inf"),
             "global_steps": self.global_steps, "optimizer": self.optimizer.state_dict()}
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(d, path)

    def resume(self, path):
        d = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(d["model"])
        if "optimizer" in d:
            self.optimizer.load_state_dict(d["optimizer"])
        self.global_steps = d.get("global_steps", 0)
