import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from oracle_backend import OracleRBSolver
from easyhec_amd import fused
from easyhec_amd.config import XARM7_K_1280x720, Cfg
from easyhec_amd.rb_solver import RBSolver
from easyhec_amd.robot import load_robot
from easyhec_amd.synthetic import camera_Tc_c2b, make_views, perturb_pose, scaled_K
from easyhec_amd.trainer import RBSolverTrainer
xarm7 = load_robot("xarm7")
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W, iters = 480, 640, 200
K = scaled_K(XARM7_K_1280x720, 0.5, W, H, True)
_, lp = make_views(xarm7, B, seed=0)
Tc = camera_Tc_c2b()
cfg = Cfg(); cfg.model.rbsolver.H, cfg.model.rbsolver.W = H, W
cfg.model.rbsolver.init_Tc_c2b = perturb_pose(Tc).tolist()
model = RBSolver(cfg, meshes=xarm7.meshes).to(dev)
ren, scene = model._ensure_renderer(), model._ensure_scene()
Kt = torch.tensor(K, dtype=torch.float32, device=dev); lpt = torch.tensor(lp, device=dev)
with torch.no_grad():
    gt, _ = fused.render_mask_loss(ren.glctx, scene, fused.mvp_matrices(Kt, H, W, torch.tensor(Tc, dtype=torch.float32, device=dev), lpt), torch.zeros((B, H, W), device=dev))
ref = (gt > 0.5).float()
batch = {"mask": ref, "link_poses": lpt, "K": Kt[None].repeat(B,1,1)}
tr = RBSolverTrainer(cfg, model, batch)
cpu = OracleRBSolver(xarm7, perturb_pose(Tc), H, W)
cb = {"mask": ref.cpu(), "link_poses": torch.tensor(lp), "K": torch.tensor(K, dtype=torch.float32)[None].repeat(B,1,1)}
ctr = RBSolverTrainer(cfg, cpu, cb)
from easyhec_amd.se3 import se3_log_map
gt6 = se3_log_map(torch.tensor(Tc, dtype=torch.float32)[None].permute(0,2,1))[0]
G=[];C=[]
for it in range(iters):
    lg = float(tr.step()[1]); lc = float(ctr.step()[1])
    dg = model.dof.detach().cpu(); dc = cpu.dof.detach()
    G.append(dg.clone()); C.append(dc.clone())
    if it % 10 == 0 or it == iters - 1:
        print(it, f"loss {lg:.2f} {lc:.2f} | dof diff {(dg-dc).abs().max():.2e} | err gpu t {(dg[:3]-gt6[:3]).norm()*1000:.2f}mm r {(dg[3:]-gt6[3:]).abs().max()*57.3:.3f}deg")
G=torch.stack(G); C=torch.stack(C)
for n in (20, 50):
    print("mean last", n, "diff", (G[-n:].mean(0)-C[-n:].mean(0)).abs(), "jitter std", G[-n:].std(0))
