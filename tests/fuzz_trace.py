#!/usr/bin/env python3
"""Fuzz of rt_trace_closest / rt_trace_any, not collected by pytest: random soups (with and without the two huge floor
triangles, 1-6 objects, 8 .. 200 000 triangles) and random ray sets of awkward sizes (1, 63, 65, 4097 ...: the per-lane
kernels claim rays from eight ranges), the four traversals against the oracle: primitive ids, len bits, point bits,
occlusion with and without a far limit.  `python tests/fuzz_trace.py [cases] [seed]` on a GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as ol
from raytracer_amd import host
from test_gpu_trace import _soup_scene, _random_rays, _primary_rays, _edge_rays

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    n = int(2 ** rng.uniform(3, 17.6))
    sd = _soup_scene(n, int(rng.integers(0, 1 << 30)), nobj=int(rng.integers(1, 7)), big_floor=bool(rng.random() < 0.6),
                     spread=float(rng.uniform(0.1, 0.6)), tri=float(10 ** rng.uniform(-2.5, -0.7)))
    nr = int(rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 511, 4097, 20000, int(rng.integers(1, 60000))]))
    rays = _random_rays(nr, int(rng.integers(0, 1 << 30)), box=float(rng.uniform(0.3, 1.2)))
    if rng.random() < 0.3:
        rays = np.concatenate([rays, _edge_rays()])
    if rng.random() < 0.3:
        rays = np.concatenate([rays, _primary_rays(sd, 48, 32)])
    if rng.random() < 0.3:                                          # unnormalised / tiny directions: the guarded path
        k = max(1, rays.shape[0] // 5)
        rays[:k, 3:6] *= np.float32(rng.uniform(0.001, 9.0))
    tmax = rng.uniform(0.02, 1.5, rays.shape[0]).astype(np.float32)
    print(f"case {c}: {n} triangles, {rays.shape[0]} rays ...", flush=True)
    g = host.Scene(sd); o = ol.OracleScene(sd)
    want = o.trace_closest(rays); wany = o.trace_any(rays); wlim = o.trace_any(rays, tmax)
    res = []
    for mode in ("bvh", "fast", "lane", "strict"):
        g.set_traversal(mode)
        got = g.trace_closest(rays)
        ok = bool(np.array_equal(got["prim"], want["prim"]) and got["len"].tobytes() == want["len"].tobytes()
                  and got["p"].tobytes() == want["p"].tobytes()
                  and np.array_equal(g.trace_any(rays), wany) and np.array_equal(g.trace_any(rays, tmax), wlim))
        res.append((mode, ok))
    good = all(ok for _, ok in res)
    bad += 0 if good else 1
    print(f"case {c}: hits {int((want['prim'] >= 0).sum())} " + " ".join(f"{m}={'ok' if ok else 'DIFFER'}" for m, ok in res), flush=True)
    g.close(); o.close()
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
