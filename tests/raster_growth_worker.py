"""Child process of tests/test_gpu_ops.py::test_rasterize_survives_a_frame_that_outgrows_the_queue_storage: the floor of
the drop-in rasterizer's queue storage is an environment variable the library reads once (EHR_RASTER_MIN_ENTRIES), so it
needs a process of its own.  The same mesh is rasterized a few times small on screen (the sync-free size read-back
settles: the next call of this shape does not wait for the size), then, same shape, ten times as large: the queues
need ~10x the entries of the frames before and do not fit.  Every frame is compared with the oracle, bit for bit."""
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
    from easyhec_amd import dr
    from oracle import oracle
    dev = torch.device("cuda:0")
    ctx = dr.RasterizeCudaContext(dev)
    H, W = 720, 1280
    pos, tri = helpers.grid_mesh(20, z=0.2, lo=-0.9, hi=0.9)  # 800 triangles: ~1 tile each when small, ~14 when they fill the screen
    tt = torch.tensor(tri, device=dev)
    res = []
    for scale in [0.1, 0.1, 0.1, 0.1, 1.0, 1.0, 0.1]:
        p = pos.copy()
        p[:, :2] *= scale
        r_ref, db_ref = oracle.rasterize(p[None], tri, [H, W])
        r, db = dr.rasterize(ctx, torch.tensor(p[None], device=dev), tt, [H, W])
        torch.cuda.synchronize()
        rn, dbn = r.cpu().numpy(), db.detach().cpu().numpy()
        res.append({"scale": scale, "rast_equal": bool((rn == r_ref).all()), "db_equal": bool((dbn == db_ref).all()),
                    "nan": bool(np.isnan(rn).any()), "covered": int((rn[..., 3] > 0).sum())})
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
