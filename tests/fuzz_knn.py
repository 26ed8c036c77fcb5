#!/usr/bin/env python3
"""Fuzz of the radiance estimate and the kNN lists, not collected by pytest: random maps (sizes, key patterns), k, radius and
queries; rt_photon_knn (LocatePhoton's lists: indices, dist2, found, radius -- bit-exact) and rt_photon_irradiance (the
default wave-shared / per-lane estimate and its strict redo: 1e-5 relative) against the oracle on the same heap.
`python tests/fuzz_knn.py [cases] [seed]` on a GPU box."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as ol
from raytracer_amd import abi, host, scene
from test_gpu_photon import _random_map, _tie_map

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
g = host.Scene(sd)
bad = 0
only = int(os.environ.get("FUZZ_ONLY", "-1"))                          # replay one case of a run (the draws before it are consumed alike)
for c in range(cases):
    kind = ["random", "wall", "grid", "random", "wall"][int(rng.integers(0, 5))]
    n = int(2 ** rng.uniform(0, 19))                                  # 1 .. 520 k
    seed = int(rng.integers(0, 1 << 30))
    src = _random_map(n, seed) if kind == "random" else _tie_map(n, seed, kind)
    src["power"][1:] = rng.uniform(0.0, 1.0, (n, 3)).astype(np.float32)
    src["phi"][1:] = rng.integers(0, 256, n); src["theta"][1:] = rng.integers(0, 256, n)
    if only >= 0 and c != only:
        heap = src                                                    # same shapes for the draws below; no build
    else:
        heap = g.photonmap_build(abi.RT_PMAP_GLOBAL, src, power_scale=1.0)
    k = int(rng.choice([1, 2, 7, 8, 9, 31, 32, 63, 64, 100, 103, 104, 112, int(rng.integers(1, 113))]))
    radius = float(np.float32(10 ** rng.uniform(-2.5, 0.3)))
    nq = 90_000 if c % 10 == 9 else 1500                            # every tenth case: enough chunks for the XCD segment mapping of k_irr_main
    q = rng.uniform(-0.6, 0.6, (nq, 3)).astype(np.float32); q[:, 2] += 0.5
    pos = heap["position"][1:]
    on = rng.integers(0, n, nq // 3)
    q[: nq // 3] = pos[on]                                            # a third of the queries sit on photons
    q[nq // 3: nq // 2, 1] = np.float32(-0.5)                         # some on the floor plane
    nrm = rng.normal(size=(nq, 3)).astype(np.float32); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    if only >= 0 and c != only:
        continue
    oi = np.zeros((nq, k), np.int32); od = np.zeros((nq, k), np.float32); of = np.zeros(nq, np.int32)
    orr = np.zeros(nq, np.float32); vis = C.c_int64(0)
    h = np.ascontiguousarray(heap)
    ol.oracle().orc_pmap_knn(h.ctypes.data, n, q.ctypes.data, nq, k, radius, oi.ctypes.data, od.ctypes.data, of.ctypes.data,
                             orr.ctypes.data, C.byref(vis))
    want = np.zeros((nq, 3), np.float32)
    ol.oracle().orc_pmap_irradiance(h.ctypes.data, n, q.ctypes.data, nrm.ctypes.data, nq, k, radius, want.ctypes.data)
    gi, gd, gf, gr = g.photon_knn(abi.RT_PMAP_GLOBAL, q, k, radius)
    got = g.photon_irradiance(abi.RT_PMAP_GLOBAL, q, nrm, k, radius)
    lists = bool(np.array_equal(gi, oi) and np.array_equal(gf, of) and gd.tobytes() == od.tobytes() and gr.tobytes() == orr.tobytes())
    irr = bool(np.allclose(got, want, rtol=1e-5, atol=1e-12, equal_nan=True))      # r2 = 0 (k photons on the query): (1/pi)/0 times the sum, inf or NaN as in the reference
    bad += 0 if (lists and irr) else 1
    if only >= 0:
        w = np.flatnonzero(~np.isclose(got, want, rtol=1e-5, atol=1e-12, equal_nan=True).all(axis=1))
        print("differing queries:", w[:20], "of", nq)
        for j in w[:6]:
            print(" query", j, q[j], "found", of[j], "r2", orr[j], "want", want[j], "got", got[j], "d2", od[j, :of[j]])
    print(f"case {c}: {kind} n={n} k={k} r={radius:.4g} found_mean={of.mean():.1f} lists={'ok' if lists else 'DIFFER'} "
          f"irradiance={'ok' if irr else 'DIFFER'}", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
