"""The online calibration loop of /root/reference/easyhec/trainer/rbsolve_iter.py:157-167 (``fit``) with the hardware
replaced by callbacks: per exploration round ``capture_data`` (robot + camera + segmentation -> one more (mask, qpos)
frame), ``rebuild`` (a fresh RBSolver from the configured initial pose over ALL frames so far, rbsolve_iter.py:277-285),
``do_fit`` (``num_epochs`` Adam iterations -- one ``ehr_solver_step`` launch chain each), ``explore_next_state``
(space_explorer.py:87-96,152-185: sample camera poses from this round's optimisation history, score the candidate joint
configurations by summed mask variance -- one ``ehr_mask_variance`` call --, move to the best).

Robot drivers, RealSense capture, PointRend / SAM segmentation and motion planning are the parts of the reference that
stay outside this package; they enter as ``capture`` and ``candidates`` callables."""
import time

import numpy as np
import torch

from .config import Cfg
from .rb_solver import RBSolver
from .space_explorer import SpaceExplorer
from .trainer import RBSolverTrainer

__all__ = ["OnlineCalibration"]


class OnlineCalibration:
    def __init__(self, robot, K, H, W, init_Tc_c2b, capture, candidates, device="cuda:0", num_epochs=1000,
                 explore_iters=5, sample=10, start=200, lr=0.003, seed=0):
        """capture(qpos) -> bool/float mask [H,W] of the robot as the camera sees it at that joint configuration;
        candidates(round) -> (qposes [Q,dof], valid [Q] bool or None): the joint configurations the planner would
        accept this round (space_explorer.py:98-150 decides that with pymp / SAPIEN in the reference)."""
        self.robot, self.K, self.H, self.W = robot, np.asarray(K, dtype=np.float64), int(H), int(W)
        self.init_Tc_c2b = np.asarray(init_Tc_c2b, dtype=np.float64)
        self.capture, self.candidates = capture, candidates
        self.device = torch.device(device)
        self.num_epochs, self.explore_iters, self.sample, self.start, self.lr = num_epochs, explore_iters, sample, start, lr
        self.gen = torch.Generator().manual_seed(seed)
        self.explorer = SpaceExplorer(robot, self.K, self.H, self.W, device=self.device)
        self.qposes, self.masks = [], []
        self.model = None
        self.log = []

    def _batch(self):
        lp = np.stack([self.robot.link_poses(q) for q in self.qposes]).astype(np.float32)
        n = len(self.qposes)
        return {"mask": torch.as_tensor(np.stack(self.masks), dtype=torch.float32, device=self.device),
                "link_poses": torch.as_tensor(lp, device=self.device),
                "K": torch.as_tensor(self.K, dtype=torch.float32, device=self.device)[None].repeat(n, 1, 1)}

    def fit(self, first_qpos):
        """Returns the final Tc_c2b [4,4] (numpy).  ``self.log`` holds one record per exploration round."""
        qpos = np.asarray(first_qpos, dtype=np.float64)
        for it in range(self.explore_iters):
            t0 = time.perf_counter()
            self.qposes.append(qpos)                                      # capture_data (rbsolve_iter.py:169-262)
            self.masks.append(np.asarray(self.capture(qpos), dtype=np.float32))
            cfg = Cfg()                                                   # rebuild (rbsolve_iter.py:277-285)
            cfg.model.rbsolver.H, cfg.model.rbsolver.W = self.H, self.W
            cfg.model.rbsolver.init_Tc_c2b = self.init_Tc_c2b.tolist()
            cfg.solver.max_lr = self.lr
            self.model = RBSolver(cfg, meshes=self.robot.meshes).to(self.device)
            trainer = RBSolverTrainer(cfg, self.model, self._batch(), fast=True)
            # do_fit (rbsolve_iter.py:139-155): exactly num_epochs effective steps -- a step the chain reports instead of
            # taking (a close-up view the slot-limited plan cannot hold) is recovered from and run again by fit()
            trainer.fit(self.num_epochs)
            loss = trainer.last_loss
            torch.cuda.synchronize(self.device)
            if not np.isfinite(float(loss)):
                raise RuntimeError(f"round {it}: the solve ended on a reported step (mask_loss NaN)")
            t1 = time.perf_counter()
            rec = {"round": it, "frames": len(self.qposes), "mask_loss": float(loss), "solve_s": t1 - t0}
            if it + 1 < self.explore_iters:                               # explore_next_state (rbsolve_iter.py:263-275)
                cand, valid = self.candidates(it)
                out = self.explorer.forward(cand, self.model.history_ops.detach().cpu(), start=self.start,
                                            sample=self.sample, valid=valid, generator=self.gen)
                torch.cuda.synchronize(self.device)
                qpos = np.asarray(out["qpos"], dtype=np.float64)
                rec.update(explore_s=time.perf_counter() - t1, variance=float(out["variance"]),
                           var_mean=float(out["var_mean"]), candidates=int(len(cand)))
            self.log.append(rec)
        return self.model.Tc_c2b().detach().cpu().numpy()
