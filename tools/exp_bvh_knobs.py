#!/usr/bin/env python3
"""sweep of the BVH build knobs (RT_BVH_*): 4K primary + shadow frame on the monkey and on the 1 M-triangle mesh"""
import ctypes as C, os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracer_amd import abi, host, synth
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
scenes = [("monkey", bench.bench_scene()), ("mesh_100k", synth.mesh_scene(320, 160)), ("mesh_1m", synth.mesh_scene(1000, 500))]
def frame_ms(sd):
    gs = host.Scene(sd)
    cam = host.camera_for(sd, bench.BASE_RES)
    p = host.default_params("hw4", primary_only=1)
    frame = torch.zeros(cam.resx * cam.resy * 3, dtype=torch.uint8, device="cuda")
    st = abi.rt_stats()
    ms = []
    for i in range(5):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), sp))
        if i: ms.append(st.kernel_ms)
    gs.close()
    return round(sum(ms) / len(ms), 3)
# the knobs are read once per process (static initialisers): one child process per setting
if len(sys.argv) > 1:
    print(sys.argv[1], {n: frame_ms(sd) for n, sd in scenes}, flush=True)
else:
    import subprocess
    settings = [{}] + [{"RT_BVH_LEAF": v} for v in (3, 4, 8)] + [{"RT_BVH_CONE_MAX": v} for v in (48, 4096)] + \
               [{"RT_BVH_SIZE_THR": v} for v in (4, 16, 1e9)] + [{"RT_BVH_CONE_COS": v} for v in (0.0, 0.5, 0.7)]
    for s in settings:
        env = dict(os.environ, **{k: str(v) for k, v in s.items()})
        subprocess.run([sys.executable, __file__, str(s)], env=env, check=True, timeout=200)
