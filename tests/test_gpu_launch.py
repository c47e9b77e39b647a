"""bench.py as the driver launches it for N > 1, at world size 1: `python -m torch.distributed.run --nproc-per-node 1 ...
bench.py --gpus 1` with RANK / WORLD_SIZE / MASTER_* from the environment.  With WORLD_SIZE=1 bench.py forms no process
group, so the test also runs tools/dp_probe.py under the same launcher: init_process_group(backend="nccl") -> the
library-owned RCCL communicator (rank 0's ncclUniqueId over the group, ncclCommInitRank, the self-check all-reduce) ->
the data-parallel launch sequence [ehr_solver_step(defer_adam), ncclAllReduce, ehr_pose_adam] captured in a hipGraph and
replayed -- end to end, one rank.  N > 1 over xGMI is the driver's to run; this keeps the path it will take exercised."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, out.stdout[-1500:]
    return json.loads(lines[-1])


def test_bench_under_torchrun_matches_the_plain_run():
    port = 29700 + (os.getpid() % 1000)
    common = ["--gpus", "1", "--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--no-side"]
    plain = _run([sys.executable, "bench.py"] + common)
    launched = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                     "127.0.0.1", "--master-port", str(port), "bench.py"] + common)
    assert plain["n_gpus"] == launched["n_gpus"] == 1
    assert plain["config"]["workload"] == launched["config"]["workload"] == "xarm7_1280x720_8view"
    # the same chain, the same box, seconds apart (a fresh process each): within a few per cent
    assert abs(launched["value"] - plain["value"]) <= 0.05 * plain["value"], (plain["value"], launched["value"])
    assert abs(launched["config"]["final_mask_loss"] - plain["config"]["final_mask_loss"]) <= 1e-3 * plain["config"]["final_mask_loss"]


def test_nccl_group_rccl_exchange_and_graph_at_world_size_one():
    port = 30700 + (os.getpid() % 1000)
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
              "127.0.0.1", "--master-port", str(port), "tools/dp_probe.py", "--steps", "100"])
    assert r["backend"] == "nccl" and r["rccl"] and r["graph"] and r["world"] == 1
    assert r["bit_equal_to_plain_step"], r
    # the data-parallel form adds one 8-float all-reduce and one 6-thread Adam launch to the chain
    assert r["dp_ms_per_step"] <= 1.5 * r["plain_ms_per_step"], r


def test_bench_with_two_ranks_on_one_device_and_the_peer_memory_exchange():
    """VERDICT round 5, task 6: ``bench.py --gpus 2`` with the one-shot peer-memory exchange (EHR_COMM=p2p: IPC mailboxes; two
    ranks on ONE device, which RCCL refuses and IPC allows): the step uses it, the line names it and prints its latency next to
    torch.distributed's."""
    port = 31900 + (os.getpid() % 1000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EHR_BENCH_BACKEND="gloo", EHR_BENCH_DEVICE="0", EHR_COMM="p2p")
    args = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "40", "--warmup", "10", "--no-cpu-baseline"]
    out = subprocess.run(args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["exchange"]["used_by_the_step"] == "peer_memory" and "peer-memory" in r["config"]["parallelism"]
    ex = r["exchange"]["us_per_exchange_of_8_floats"]
    assert 0.0 < ex["peer_memory"] < 1.0e5 and 0.0 < ex["torch_distributed"] < 1.0e5
    assert np.isfinite(r["config"]["final_mask_loss"]) and r["value"] > 0


@pytest.mark.parametrize("try_rccl", [False, True])
def test_bench_with_two_ranks_on_one_device_over_gloo(try_rccl):
    """bench.py's N > 1 code -- the rendezvous, the collective warm-up decision, the broadcast in
    timed_blocks, per_rank_ms_per_step, allreduce_8float_us -- has to have run before the driver's multi-GPU bench does.
    No second GPU here: two ranks share cuda:0 and exchange over gloo (EHR_BENCH_BACKEND / EHR_BENCH_DEVICE, test-only).
    try_rccl: the library-owned RCCL exchange is ATTEMPTED (EHR_TRY_RCCL=1); two ranks on one device are refused by RCCL, so
    this walks FusedPoseStep's agreement + fall-back branches (fast.py): both ranks must end up on torch.distributed."""
    port = 31700 + (os.getpid() % 1000) + (50 if try_rccl else 0)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EHR_BENCH_BACKEND="gloo", EHR_BENCH_DEVICE="0")
    if try_rccl:
        env["EHR_TRY_RCCL"] = "1"
    args = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "40", "--warmup", "10", "--no-cpu-baseline"]
    out = subprocess.run(args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]            # exactly one JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_views"] == 16 and r["config"]["views_per_gpu"] == 8
    assert len(r["per_rank_ms_per_step"]) == 2 and all(0.0 < t < 50.0 for t in r["per_rank_ms_per_step"])
    assert 0.0 < r["allreduce_8float_us"] < 1.0e5
    assert r["value"] > 0 and r["scaling"] == "weak" and "torch.distributed" in r["config"]["parallelism"]
    assert r["ms_per_step"] >= max(r["per_rank_ms_per_step"]) - 1e-6     # the job is as slow as its slowest rank
    assert np.isfinite(r["config"]["final_mask_loss"])
    if try_rccl:
        assert "library-owned RCCL exchange unavailable" in out.stderr
