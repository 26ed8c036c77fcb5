#!/usr/bin/env python3
"""per-pass traversal work of the drt_1m workload: frames with max_depth = d and d-1, difference of the work counters."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import abi, host, synth
res = (960, 540); spp = 16
sd = synth.mesh_scene(1000, 500, reflect=0.3)
gs = host.Scene(sd)
cam = host.camera_for(sd, res)
frame = torch.zeros(res[0] * res[1] * 3, dtype=torch.uint8, device="cuda")
prev = None
for d in range(0, 7):
    p = host.default_params("hw4", spp=spp, lens_radius=0.01, focal_distance=1.3, light_radius=0.5, glossy=0.05, seed=3,
                            max_depth=d, count_work=1)
    st = abi.rt_stats()
    abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), None))
    cur = (st.rays_primary + st.rays_reflect + st.rays_refract, list(st.closest_work), st.rays_shadow, list(st.any_work))
    if prev is None:
        dn, dw, dsn, dsw = cur[0], cur[1], cur[2], cur[3]
    else:
        dn = cur[0] - prev[0]; dw = [a - b for a, b in zip(cur[1], prev[1])]
        dsn = cur[2] - prev[2]; dsw = [a - b for a, b in zip(cur[3], prev[3])]
    print(f"pass {d}: {dn} rays box/ray {dw[0]/max(dn,1):.1f} tri/ray {dw[1]/max(dn,1):.1f} exact/ray {dw[2]/max(dn,1):.2f} fetch/ray {dw[3]/max(dn,1):.1f} | "
          f"shadow {dsn} box/ray {dsw[0]/max(dsn,1):.1f} tri/ray {dsw[1]/max(dsn,1):.1f} fetch/ray {dsw[3]/max(dsn,1):.1f}")
    prev = cur
