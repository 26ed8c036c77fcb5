"""CPU tests of the multi-rank path (gloo, world_size 2 and 4): tile dealing, per-rank compaction,
gather to rank 0 and re-assembly, exactly the code bench.py runs over RCCL on the 8-GPU node
(--dry-run replaces the GPU render with a rank-coloured fill)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("res,nranks", [((3840, 2160), 8), ((200, 136), 3), ((64, 64), 2), ((65, 1), 4)])
def test_tiles_partition_the_frame(res, nranks):
    parts = [bench.rank_pixel_index(res[0], res[1], r, nranks) for r in range(nranks)]
    allp = np.concatenate(parts)
    assert allp.size == res[0] * res[1] and np.array_equal(np.sort(allp), np.arange(res[0] * res[1]))
    if res == (3840, 2160):
        sizes = [p.size for p in parts]
        assert max(sizes) - min(sizes) <= 64 * 64 * 2           # round-robin dealing balances the load


def test_weak_scaling_frames():
    for n in (1, 2, 4, 8):
        w, h = bench.frame_for(n)
        assert w * h == n * 1920 * 1080          # BASELINE configs[2]: 1080p (x 64 spp) per GPU


@pytest.mark.parametrize("world", [2, 4])
def test_gather_to_rank0_over_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29530 + world), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "3", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["assembled_ok"] and j["n_gpus"] == world and sum(j["pixels_per_rank"]) == j["frame"][0] * j["frame"][1]


def test_a_stalled_rank_ends_the_run_with_a_nonzero_status():
    """bench.py's watchdog: a rank that never reaches a collective must not leave the run reading as a success.  Rank 1 is
    made to stall before the last barrier (RT_BENCH_TEST_STALL_RANK, dry-run only); after RT_BENCH_WATCHDOG_S seconds every
    rank gives up, rank 0 prints its line marked with `error`, and the launcher's exit status is non-zero."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RT_BENCH_TEST_STALL_RANK="1", RT_BENCH_WATCHDOG_S="20")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29539", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0, "a stalled run exited with status 0"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines and "error" in json.loads(lines[-1]), r.stdout[-2000:]
    assert "exit status 3" in r.stderr


def test_failed_line_detection():
    assert bench.line_failed({"error": "x"})
    assert bench.line_failed({"value": 1.0, "parity": {"parity_sample_ok": False}})
    assert bench.line_failed({"value": 1.0, "config": {"assembled_frame_equals_one_gpu_frame": False}})
    assert not bench.line_failed({"value": 1.0, "parity": {"parity_sample_ok": True}, "config": {"heap_equals_one_rank_heap": True}})
    assert not bench.line_failed({"value": 1.0, "parity": None})


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_emission_allgather_over_gloo(world):
    """SURVEY 8e's one exchange step: emission sharded by emission index, counts + payload all-gathered, every rank
    ends with the array a single process emits (raytracer_amd/dist.py, the code the GPU ranks run over RCCL)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29560 + world),
           os.path.join(ROOT, "tests", "dist_emit_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["world"] == world
    for which, target in (("0", 5000), ("1", 300)):
        res = j["results"][which]
        assert res["n"] == target and res["same_everywhere"] and res["equals_single_process"], res
        assert res["heap_shared_equals_own_balance"], res          # one balance per node, broadcast
        assert res["allgather_bytes"] > 0
