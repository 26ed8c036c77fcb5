#!/usr/bin/env python3
"""BASELINE configuration 3's frame alone (bench.py's headline workload), a few frames, for experiments: options from the
environment (RT_PATH_DEPTH, RT_PATH_BATCH, RT_PATH_WAVES, RT_LANE_DEPTH, RT_TRACE_PASSES=1 ...).
    python3 tools/exp_drt.py [frames=4] [share=1: tile_nranks, renders rank 0's share of the frame]"""
import ctypes as C
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch            # noqa: E402
import bench            # noqa: E402
from raytracer_amd import abi, host, synth     # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
share = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sd = synth.mesh_scene(1000, 500, reflect=0.3)
gs = host.Scene(sd)
cam = host.camera_for(sd, bench.HEAD_RES)
kw = dict(tile_size=64, tile_rank=0, tile_nranks=share, packed_output=1) if share > 1 else {}
p = host.default_params("hw4", spp=bench.HEAD_SPP, **kw, **bench.DRT_KW)
frame = torch.zeros(bench.HEAD_RES[0] * bench.HEAD_RES[1] * 3, dtype=torch.uint8, device="cuda")
st = abi.rt_stats()
for _ in range(2):
    abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, None, None))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, None, None))
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / frames * 1e3
abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), None))
rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract
print(json.dumps(dict(ms_per_frame=ms, mrays_per_s=rays / ms / 1e3, share=share, passes=st.passes, stats_frame_ms=st.kernel_ms, path_ms=st.path_ms,
                      path_roots=st.path_roots, path_closest_rays=st.path_closest_rays, path_shadow_rays=st.path_shadow_rays,
                      closest_ms=st.trace_closest_ms, any_ms=st.trace_any_ms, lane_closest_ms=st.lane_closest_ms, lane_any_ms=st.lane_any_ms,
                      rays=dict(primary=st.rays_primary, shadow=st.rays_shadow, reflect=st.rays_reflect, refract=st.rays_refract))))
