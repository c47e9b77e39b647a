"""Stage timings of the solver step (the chain bench.py times) on a bench workload, without the CPU baseline:
    [EHR_LIB=ab/libehr_x.so] python tools/step_bench.py [workload] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time
import torch
import bench
from easyhec_amd import fused
wl = sys.argv[1] if len(sys.argv) > 1 else bench.WORKLOAD
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
p = bench.build_problem(0, 1, dev, graph=False, workload=wl)
tr = p["trainer"]
for _ in range(20):
    tr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
fused.check_status(p["glctx"])
fused.set_timing(p["glctx"], True)
for _ in range(steps):
    tr.step()
ms, n = fused.read_timing(p["glctx"])
print(os.environ.get("EHR_LIB", "default"), wl, f"{el / steps * 1e6:.1f} us/step {p['n_views'] * steps / el:.0f} frames/s",
      {k: round(v / n * 1e3, 1) for k, v in ms.items() if not k.startswith("unused")}, "loss", float(tr.last_loss), flush=True)
