#!/usr/bin/env python3
"""Fuzz of the device photon-map builder (photon_build_dev.hip), not collected by pytest: random sizes and key patterns,
rt_photonmap_build_dev against the host builder (= PhotonMap::balance, photon.cpp:133-254) -- heap order, planes, scaled
powers, device image and box byte for byte.  `python tests/fuzz_photon_build.py [cases] [seed]` on a GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import oracle_lib as ol
from raytracer_amd import abi, host, scene
from test_gpu_photon import _map_image, _random_map, _tie_map

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sd = scene.parse_scene_file(os.path.join(ol.GOLDEN, "final_input.txt"), flavour="final")
a_scene, b_scene = host.Scene(sd), host.Scene(sd)
bad = 0
for c in range(cases):
    kind = ["random", "wall", "grid", "same", "sorted", "reversed", "mixed"][int(rng.integers(0, 7))]
    n = int(2 ** rng.uniform(0, 20.3))                       # 1 .. 1.3 M, log-uniform
    if kind in ("same", "sorted", "reversed"):
        n = min(n, 120_000)                                  # quickselect's worst cases: thousands of passes
    seed = int(rng.integers(0, 1 << 30))
    if kind == "mixed":                                      # part random, part on a wall, part duplicated
        src = _random_map(n, seed)
        p = src["position"][1:]
        k = n // 3
        p[:k, int(rng.integers(0, 3))] = np.float32(rng.uniform(-0.5, 0.5))
        if n > 8:
            p[k:2 * k] = p[rng.integers(0, n, k)]
    elif kind == "random":
        src = _random_map(n, seed)
    else:
        src = _tie_map(n, seed, kind)
    scale = float(np.float32(rng.uniform(0.1, 2.0)))
    os.environ["RT_PHOTON_BUILD"] = "host"
    h_heap = a_scene.photonmap_build(0, src.copy(), scale)
    del os.environ["RT_PHOTON_BUILD"]
    d = torch.from_numpy(src.copy().view(np.uint8)).cuda()
    b_scene.photonmap_build_dev(0, d.data_ptr(), n, scale)
    torch.cuda.synchronize()
    d_heap = d.cpu().numpy().view(abi.photon_dtype)
    na, ia, ba = _map_image(a_scene); nb, ib, bb = _map_image(b_scene)
    ok = (h_heap[1:].tobytes() == d_heap[1:].tobytes() and h_heap["power"][0].tobytes() == d_heap["power"][0].tobytes()
          and na == nb == n and ba.tobytes() == bb.tobytes() and np.array_equal(ia, ib))
    bad += 0 if ok else 1
    print(f"case {c}: {kind} n={n} seed={seed} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
