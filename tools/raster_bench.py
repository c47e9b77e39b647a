"""Time of one drop-in dr.rasterize call per xArm7 link at 1280x720 (the reference's per-(frame, link) call), both forms of
the rasterizer:  python tools/raster_bench.py [--reps 200]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easyhec_amd import _lib, dr  # noqa: E402,F401
from easyhec_amd.renderer import NVDiffrastRenderer  # noqa: E402
from easyhec_amd.robot import load_robot  # noqa: E402
from easyhec_amd.synthetic import WORKLOADS, camera_Tc_c2b, make_views  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--db", type=int, default=1, help="also write rast_db (dr.rasterize's default)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = WORKLOADS["xarm7_1280x720_8view"]
    rb = load_robot("xarm7")
    H, W = wl["H"], wl["W"]
    K = torch.tensor(wl["K"], dtype=torch.float32, device=dev)
    _, lp = make_views(rb, 1)
    Tc = torch.tensor(camera_Tc_c2b(radius=wl["radius"], lift=wl["lift"]), dtype=torch.float32, device=dev)
    lp = torch.tensor(lp, device=dev)[0]
    r = NVDiffrastRenderer([H, W])
    ctx = r.glctx
    rows = []
    for k, (v, f) in enumerate(rb.meshes):
        verts = torch.tensor(v, dtype=torch.float32, device=dev)
        faces = torch.tensor(f, dtype=torch.int32, device=dev)
        mvp = r.clip_matrices(K, (Tc @ lp[k])[None])
        pos = r.clip_positions_batched(mvp, verts).contiguous()
        row = {"link": k, "tris": int(faces.shape[0])}
        for form, env in (("direct", "1000000000"), ("queued", "0")):
            os.environ["EHR_RASTER_DIRECT_MAX"] = env
            # straight through the C ABI into preallocated outputs: the python op costs ~30 us of host time per call, more
            # than the launches it makes
            out = torch.empty((1, H, W, 4), dtype=torch.float32, device=dev)
            db = torch.empty((1, H, W, 4), dtype=torch.float32, device=dev)
            fn, st = _lib.lib().ehr_rasterize_fwd, torch.cuda.current_stream().cuda_stream
            args = (ctx.handle, _lib.ptr(pos), _lib.ptr(faces), None, 1, int(verts.shape[0]), int(faces.shape[0]), H, W,
                    _lib.ptr(out), _lib.ptr(db) if a.db else None, None, ctypes.c_void_p(st))
            for _ in range(10):
                _lib.check(fn(*args), "rasterize")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn(*args)
            e1.record()
            torch.cuda.synchronize()
            row[form + "_us"] = round(e0.elapsed_time(e1) * 1e3 / a.reps, 1)
            row["covered"] = int((out[..., 3] > 0).sum())
        rows.append(row)
        print(row)


if __name__ == "__main__":
    main()
