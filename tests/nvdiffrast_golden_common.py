"""Shared by tests/test_nvdiffrast_golden.py (CPU: the oracle) and tests/test_gpu_nvdiffrast_golden.py (HIP ops, fused
path): reads the files tools/dump_nvdiffrast_golden.py wrote on an NVIDIA box (tests/golden/nvdiffrast_*.npz, or the
directory EHR_NVDIFFRAST_GOLDEN_DIR names) and scores an implementation's outputs against nvdiffrast's."""
import glob
import os

import numpy as np

GOLD = os.environ.get("EHR_NVDIFFRAST_GOLDEN_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PREFIX = os.environ.get("EHR_NVDIFFRAST_GOLDEN_PREFIX", "nvdiffrast")
MASK_TOL = 1e-4   # BASELINE.json north_star: rendered masks match the reference to <= 1e-4 L-infinity
GRAD_RTOL = 1e-3  # gradients: relative to the largest component (float atomics on both sides, different orders)


def files(kind=None):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLD, PREFIX + "_*.npz"))):
        g = np.load(f, allow_pickle=False)
        if kind is None or str(g["kind"]) == kind:
            out.append(f)
    return out


def load_links(robot):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if robot == "zeropos":
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xarm7_zeropos.npz"))
        return [(z["vertices"].astype(np.float32), z["faces"].astype(np.int32))]
    z = np.load(os.path.join(root, "easyhec_amd", "assets", f"{robot}.npz"), allow_pickle=False)
    vo, to = z["vert_offsets"], z["tri_offsets"]
    return [(z["vertices"][vo[i]:vo[i + 1]].astype(np.float32), z["faces"][to[i]:to[i + 1]].astype(np.int32))
            for i in range(len(vo) - 1)]


def score_ops(name, g, rast, aa, grad_pos):
    """-> (report line, ok).  ids must agree wherever nvdiffrast drew; colours within MASK_TOL; gradients within GRAD_RTOL."""
    id_ref, id_out = g["rast"][..., 3], rast[..., 3]
    n_id = int((id_ref != id_out).sum())
    n_cov = int(((id_ref > 0) != (id_out > 0)).sum())
    both = (id_ref == id_out) & (id_ref > 0)
    d_bary = float(np.abs(g["rast"][..., :3] - rast[..., :3])[both].max()) if both.any() else 0.0
    d_aa = np.abs(g["aa"] - aa)
    n_aa = int((d_aa > MASK_TOL).sum())
    gs = max(float(np.abs(g["grad_pos"]).max()), 1e-30)
    d_g = float(np.abs(g["grad_pos"] - grad_pos).max() / gs)
    line = (f"{name}: coverage differs at {n_cov} px, triangle id at {n_id} px (of {int((id_ref > 0).sum())} covered); "
            f"bary/depth Linf {d_bary:.2e} where ids agree; antialiased colour Linf {float(d_aa.max()):.2e}, "
            f"{n_aa} px above {MASK_TOL:g}; grad_pos rel Linf {d_g:.2e}")
    return line, (n_cov == 0 and n_aa == 0 and d_g <= GRAD_RTOL)


def score_fused(name, g, mask, loss, grad_mvp):
    d_m = np.abs(g["mask"] - mask)
    n_m = int((d_m > MASK_TOL).sum())
    d_l = float(np.abs(g["loss"] - loss).max() / max(1.0, float(np.abs(g["loss"]).max())))
    gs = max(float(np.abs(g["grad_mvp"]).max()), 1e-30)
    d_g = float(np.abs(g["grad_mvp"] - grad_mvp).max() / gs)
    line = (f"{name}: mask Linf {float(d_m.max()):.2e}, {n_m} px above {MASK_TOL:g} (of {mask.size}); loss rel {d_l:.2e}; "
            f"grad_mvp rel Linf {d_g:.2e}")
    return line, (n_m == 0 and d_l <= 1e-4 and d_g <= GRAD_RTOL)
