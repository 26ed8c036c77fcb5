#!/usr/bin/env python3
"""experiment: how many of the adversarial grazing rays (tests/test_gpu_trace.py) does each traversal get wrong with the
lateral slack scaled by RT_DEBUG_SLACK_SCALE (0 = none, 1 = the product)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from raytracer_amd import host, synth
from test_gpu_trace import _adversarial_rays, _soup_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for name, sd, first in (("monkey", bench.bench_scene(), 2), ("soup", _soup_scene(3000, 31), 2), ("coarse_soup", _soup_scene(600, 9, nobj=2, big_floor=False, spread=2.0, tri=0.3), 0), ("coarse_soup_4k", _soup_scene(4000, 10, nobj=2, big_floor=False, spread=3.0, tri=0.25), 0)):
    rays = _adversarial_rays(sd, n, 5, first_tri=first)
    os.environ.pop("RT_DEBUG_SLACK_SCALE", None)
    g = host.Scene(sd); g.set_traversal("strict")
    want = g.trace_closest(rays); wany = g.trace_any(rays)
    g.close()
    print(name, "rays", rays.shape[0], "strict hits", int((want["prim"] >= 0).sum()), "occluded", int(wany.sum()), flush=True)
    for scale in ("1", "0.1", "0"):
        os.environ["RT_DEBUG_SLACK_SCALE"] = scale
        g = host.Scene(sd)
        for mode in ("bvh", "fast", "lane"):
            g.set_traversal(mode)
            got = g.trace_closest(rays)
            bad = int(((got["prim"] != want["prim"]) | (got["len"].view(np.uint32) != want["len"].view(np.uint32))).sum())
            bada = int((g.trace_any(rays) != wany).sum())
            print(f"  scale {scale} {mode}: closest mismatches {bad}, any-hit mismatches {bada}", flush=True)
        g.close()
