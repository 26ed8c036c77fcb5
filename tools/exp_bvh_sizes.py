#!/usr/bin/env python3
"""primary + shadow frame at 4K over meshes of growing size, in the KD ("fast") and BVH traversal modes"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracer_amd import abi, host, synth
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cases = [("monkey", bench.bench_scene)] + [(f"mesh_{2 * nu * nv}", (lambda nu=nu, nv=nv: synth.mesh_scene(nu, nv)))
                                           for nu, nv in ((40, 20), (100, 50), (320, 160), (1000, 500))]
for name, make in cases:
    sd = make()
    gs = host.Scene(sd)
    cam = host.camera_for(sd, bench.BASE_RES)
    p = host.default_params("hw4", primary_only=1)
    frame = torch.zeros(cam.resx * cam.resy * 3, dtype=torch.uint8, device="cuda")
    st = abi.rt_stats()
    row = {}
    for mode in ("fast", "bvh"):
        gs.set_traversal(mode)
        ms = []
        for i in range(6):
            abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), sp))
            if i: ms.append(st.kernel_ms)
        row[mode] = round(sum(ms) / len(ms), 3)
    print(name, gs.info().n_tree_prims if hasattr(gs.info(), "n_tree_prims") else "", row, flush=True)
