"""The data-parallel launch path end to end on whatever world the launcher gives (one rank per GPU, backend nccl = RCCL):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/dp_probe.py

init_process_group("nccl") -> FusedPoseStep picks the library-owned RCCL exchange (ehr_comm_*: rank 0's ncclUniqueId over
the group, ncclCommInitRank, self-check all-reduce) -> the step [ehr_solver_step(defer_adam), ncclAllReduce on the chain's
stream, ehr_pose_adam] captured in a hipGraph and replayed.  Rank 0 prints one JSON line: timings of the data-parallel and
of the plain single-GPU step on this rank's views and, at world size 1, whether the two end bit-identical."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    json_fd = os.dup(1)
    os.dup2(2, 1)  # RCCL's banner goes to stderr: stdout carries the JSON line only
    out = os.fdopen(json_fd, "w")
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    import bench
    from easyhec_amd.fast import FusedPoseStep
    p = bench.build_problem(rank, world, dev, graph=False)
    tr = p["trainer"]
    fast = tr.fast
    if world == 1:  # the trainer only takes the data-parallel form for world > 1: build it explicitly on the same problem
        fast = FusedPoseStep(p["model"], tr.batch, lr=tr.cfg.solver.max_lr, weight_decay=tr.cfg.solver.weight_decay, rccl=True)
    assert fast.rccl, "the library-owned RCCL exchange was not selected"
    from easyhec_amd.fast import ranks_agree
    assert ranks_agree(True, None, dev) and not ranks_agree(False, None, dev)   # (on the nccl backend: a device flag)
    fast.capture()

    def run(step, n):
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    dp_ms = run(fast.step, args.steps)
    dof_dp = p["model"].dof.detach().clone()
    res = {"world": world, "backend": dist.get_backend(), "rccl": bool(fast.rccl), "graph": bool(fast._graph),
           "dp_ms_per_step": round(dp_ms, 4), "views_per_rank": p["B"]}
    if world == 1:
        p2 = bench.build_problem(0, 1, dev, graph=True)
        plain_ms = run(p2["trainer"].step, args.steps)
        res["plain_ms_per_step"] = round(plain_ms, 4)
        res["bit_equal_to_plain_step"] = bool(torch.equal(dof_dp, p2["model"].dof.detach()))
    if rank == 0:
        out.write(json.dumps(res) + "\n")
        out.flush()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
