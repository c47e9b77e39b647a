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
