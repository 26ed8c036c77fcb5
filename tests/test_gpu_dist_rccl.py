"""The N > 1 path's collectives over RCCL on the GPU box.  No multi-GPU node is available to this build, so the `nccl` branch
of raytracer_amd/dist.py and bench.py had never executed anywhere (only its gloo twin, tests/test_dist_gloo.py); this runs
it with as many ranks as the box has GPUs -- one -- through the launcher the driver uses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_rccl_takes_the_exchange_calls_of_the_multi_gpu_path():
    import torch
    world = 1                           # one rank per GPU; the box has one (never more ranks than GPUs over RCCL)
    assert torch.cuda.device_count() >= 1
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "tests", "dist_rccl_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["world"] == world and j["backend"] == "nccl"
    assert j["all_ok"] and j["gather_ok"] and j["max_float"] == 1.5 + (world - 1) and j["sum_float"] == 2.0 * world
    for which, target in (("0", 5000), ("1", 300)):
        m = j["maps"][which]
        assert m["n"] == target and m["equals_single_process"] and m["broadcast_ok"] and m["allgather_bytes"] > 0, m


def test_bench_multi_rank_path_over_rccl_with_the_ranks_the_box_has():
    """bench.py's N > 1 code path as the driver launches it (torch.distributed.run, backend nccl), forced on with one rank
    (RT_BENCH_FORCE_MULTI=1): process group, tiles + packed output, the asynchronous gather to rank 0 and the frame's
    re-assembly, the all-reduces over the ranks' times and ray counts, and the lines that only exist for N > 1 (kNN
    aggregate, drt_1m_strong, emission_sharded).  The assembled frame must equal the frame rendered whole, no line may
    fail, exit status 0."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", RT_BENCH_FORCE_MULTI="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29572", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["value"] > 0 and "failed_lines" not in j
    assert j["config"]["assembled_frame_equals_one_gpu_frame"] is True
    sec = j["secondary"]
    assert set(sec) == {"knn", "drt_1m_strong", "emission_sharded"}
    assert sec["drt_1m_strong"]["config"]["assembled_frame_equals_one_gpu_frame"] is True
    assert sec["emission_sharded"]["config"]["heap_equals_one_rank_heap"] and sec["emission_sharded"]["config"]["every_rank_answers_alike"]
