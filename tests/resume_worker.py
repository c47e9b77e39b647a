"""Child process of tests/test_gpu_resume.py: rebuilds the problem, resumes a checkpoint, takes N steps, dumps the state."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def build(fast):
    from easyhec_amd.robot import load_robot
    from easyhec_amd.trainer import RBSolverTrainer
    from test_gpu_fast import problem
    cfg, make, batch = problem(load_robot("xarm7"), 2, 120, 160, 0.125)
    model = make()
    return model, RBSolverTrainer(cfg, model, batch, fast=fast)


def dump(path, model, tr):
    opt = tr.fast.state_dict() if tr.fast is not None else tr.optimizer.state_dict()
    st = opt["state"][sorted(opt["state"].keys())[0]]
    np.savez(path, dof=model.dof.detach().cpu().numpy(), history=model.history_ops[:80].cpu().numpy(),
             exp_avg=torch.as_tensor(st["exp_avg"]).cpu().numpy(), exp_avg_sq=torch.as_tensor(st["exp_avg_sq"]).cpu().numpy(),
             step=float(torch.as_tensor(st["step"]).reshape(-1)[0]), global_steps=tr.global_steps,
             loss=float(tr.last_loss))


if __name__ == "__main__":
    ckpt, out, nsteps, fast = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4] == "fast"
    model, tr = build(fast)
    tr.resume(ckpt)
    for _ in range(nsteps):
        tr.step()
    torch.cuda.synchronize()
    dump(out, model, tr)
