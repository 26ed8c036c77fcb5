#!/usr/bin/env python3
"""one-off confidence run for the culling slack, not collected by pytest: many batches of near-in-plane rays,
bvh / fast / lane against strict on the GPU, on three scenes; prints the number of rays compared and of mismatches
(expected: 0).  Run as `python tests/grazing_campaign.py [batches]` on a GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import bench
from raytracer_amd import host, synth
from test_gpu_trace import _grazing_rays, _soup_scene
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, sd in (("monkey", bench.bench_scene()), ("soup", _soup_scene(3000, 31)), ("mesh_100k", synth.mesh_scene(320, 160))):
    g = host.Scene(sd)
    total = bad = hits = 0
    for b in range(batches):
        rays = _grazing_rays(sd, 2_000_000, 1000 + b)
        g.set_traversal("strict")
        want = g.trace_closest(rays); wany = g.trace_any(rays)
        hits += int((want["prim"] >= 0).sum())
        for mode in ("bvh", "fast", "lane"):
            g.set_traversal(mode)
            got = g.trace_closest(rays)
            bad += int((got["prim"] != want["prim"]).sum()) + int((got["len"].view(np.uint32) != want["len"].view(np.uint32)).sum())
            bad += int((g.trace_any(rays) != wany).sum())
        total += rays.shape[0]
        print(name, "batch", b, "rays", total, "hits", hits, "mismatches", bad, flush=True)
    g.close()
