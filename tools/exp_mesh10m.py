#!/usr/bin/env python3
"""10 M-triangle mesh at 4K, primary + shadow: wave-synchronous BVH walk against the per-lane kernels, and the work counters"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracer_amd import abi, host, synth
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
sd = synth.mesh_scene(3162, 1581)
gs = host.Scene(sd)
cam = host.camera_for(sd, bench.BASE_RES)
p = host.default_params("hw4", primary_only=1)
frame = torch.zeros(cam.resx * cam.resy * 3, dtype=torch.uint8, device="cuda")
st = abi.rt_stats()
for mode in ("bvh", "lane", "fast"):
    gs.set_traversal(mode)
    ms = []
    for i in range(4):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), sp))
        if i: ms.append((st.kernel_ms, st.trace_closest_ms, st.trace_any_ms))
    print(mode, [round(sum(x) / len(ms), 3) for x in zip(*ms)], flush=True)
gs.set_traversal("bvh")
a = bench.algorithmic_bytes_per_ray(gs, cam, p, sp)
print({k: round(v, 2) if isinstance(v, float) else v for k, v in a.items()})
