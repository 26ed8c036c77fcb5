#!/usr/bin/env python3
"""work counters of the headline frame and of the 1 M-triangle mesh: lane-level tests vs wave-level record fetches,
in the traversal mode RT_TRAVERSAL selects"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracer_amd import abi, host, synth
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, sd in (("monkey", bench.bench_scene()), ("mesh_1m", synth.mesh_scene(1000, 500))):
    gs = host.Scene(sd)
    cam = host.camera_for(sd, bench.BASE_RES)
    p = host.default_params("hw4", primary_only=1)
    a = bench.algorithmic_bytes_per_ray(gs, cam, p, sp)
    print(name, os.environ.get("RT_TRAVERSAL", "fast"), {k: round(v, 2) if isinstance(v, float) else v for k, v in a.items()})
