#!/usr/bin/env python3
"""lane utilisation of the path kernel on config 3 (RT_PATH_DIAG): one counted frame with the path kernel forced."""
import ctypes as C, os, sys
os.environ["RT_PATH_DIAG"] = "1"; os.environ.setdefault("RT_PATH_MAX_ROOTS", "100000000")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from raytracer_amd import abi, host, synth
sd = synth.mesh_scene(1000, 500, reflect=0.3)
gs = host.Scene(sd)
cam = host.camera_for(sd, bench.HEAD_RES)
p = host.default_params("hw4", spp=bench.HEAD_SPP, count_work=1, **bench.DRT_KW)
frame = torch.zeros(bench.HEAD_RES[0] * bench.HEAD_RES[1] * 3, dtype=torch.uint8, device="cuda")
st = abi.rt_stats()
abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), None))
print("path_ms", st.path_ms, "roots", st.path_roots, "closest rays", st.path_closest_rays, "shadow rays", st.path_shadow_rays, "work", list(st.path_work))
