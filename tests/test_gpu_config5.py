"""BASELINE configs[4] on one device: 64 synthetic views @1280x720 (xArm7) in one launch, and the same 64 views as the
8 shards of 8 views ``trainer.shard_views`` hands to 8 ranks.  Views are independent given ``dof`` and every per-view
sum is an integer (fixed-point) sum, so the per-view losses and matrix gradients of a shard are BIT-identical to the
same views inside the 64-view launch, whatever the batch composition; the 8-float exchange vector ``red`` of the whole
job is the sum of the shards' vectors (float sums in a different order: 1e-6).  Properties at full size: the loss is the
SSE of its own mask, masks lie in [0, 1], the reference mask does not influence the render."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_64_views_equal_their_8_shards(xarm7):
    from easyhec_amd import _lib
    from easyhec_amd.fast import FusedPoseStep
    from easyhec_amd.trainer import shard_views
    from test_gpu_fast import problem
    cfg, make, batch = problem(xarm7, 64, 720, 1280, 1.0)
    B = 64
    model = make()
    full = FusedPoseStep(model, batch)
    dof0 = model.dof.detach().clone()
    full.step(want_mask=True)
    torch.cuda.synchronize()
    loss_b, grad_mvp, red = full.loss_b.clone(), full.grad_mvp.clone(), full.red.clone()
    mask = full.mask.clone()
    # properties of the 64-view render
    assert float(mask.min()) >= 0.0 and float(mask.max()) <= 1.0
    sse = ((mask - batch["mask"]) ** 2).sum(dim=(1, 2))
    assert torch.allclose(sse, loss_b, rtol=2e-5)
    assert float(red[7]) == B and abs(float(red[6]) - float(loss_b.double().sum())) <= 1e-5 * float(red[6])
    fg = (mask > 0).float().mean(dim=(1, 2))
    assert float(fg.min()) > 0.01 and float(fg.max()) < 0.5          # every view shows the robot
    del full
    # the 8 ranks' shards, one after the other on this device
    red_sum = torch.zeros(8, dtype=torch.float64)
    for rank in range(8):
        lo, hi = shard_views(B, rank, 8)
        assert hi - lo == 8
        sb = {k: v[lo:hi].contiguous() for k, v in batch.items()}
        m = make()
        m.dof.data.copy_(dof0)
        fs = FusedPoseStep(m, sb)
        fs.distributed = True                      # defer Adam: stop at the exchange vector, as a rank of 8 does
        lib = _lib.lib()
        fs._enqueue = fs._enqueue                  # (no collective is issued below: torch.distributed is not initialised)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = lambda x: ctypes.c_float(float(x))
        sc, hist = fs.scene, m.history_ops
        _lib.check(lib.ehr_solver_step(
            fs.glctx.handle, _lib.ptr(sc.verts), _lib.ptr(sc.tris), _lib.ptr(sc.tri_link), _lib.ptr(sc.vert_link),
            _lib.ptr(sc.opp), _lib.ptr(fs.K), _lib.ptr(fs.link_poses), _lib.ptr(fs.ref), fs.B, fs.L, sc.num_verts,
            sc.num_tris, fs.H, fs.W, f(fs.near), f(fs.far), _lib.ptr(m.dof.data), _lib.ptr(fs.exp_avg),
            _lib.ptr(fs.exp_avg_sq), _lib.ptr(fs.step_t), _lib.ptr(hist), hist.shape[0], _lib.ptr(fs.hist_row), f(fs.lr),
            f(fs.betas[0]),
            f(fs.betas[1]), f(fs.eps), f(fs.wd), _lib.ptr(fs.mvp), _lib.ptr(fs.tc_jac), _lib.ptr(fs.mask),
            _lib.ptr(fs.loss_b), _lib.ptr(fs.grad_mvp), _lib.ptr(fs.red), _lib.ptr(fs.loss), _lib.ptr(fs.grad), 1,
            stream), "ehr_solver_step")
        torch.cuda.synchronize()
        assert torch.equal(fs.loss_b, loss_b[lo:hi]), rank
        assert torch.equal(fs.grad_mvp, grad_mvp[lo:hi]), rank
        assert torch.equal(fs.mask, mask[lo:hi]), rank
        assert torch.equal(m.dof.data, dof0)                             # deferred: the rank has not stepped yet
        red_sum += fs.red.double().cpu()
        del fs, m
    assert float(red_sum[7]) == B
    assert np.abs(red_sum.numpy() - red.double().cpu().numpy()).max() <= 1e-6 * float(red.abs().max())
