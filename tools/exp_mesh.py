"""experiment: 1M-triangle mesh at 4K, primary+shadow, fast vs strict; prints timings and work counters"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import synth, host, abi
nu, nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 500)
res = (3840, 2160)
sd = synth.mesh_scene(nu, nv)
t0 = time.time(); g = host.Scene(sd); print("scene create %.2fs" % (time.time() - t0), flush=True)
inf = g.info(); print("nodes", inf.flat_nodes, "leafrefs", inf.leaf_refs, "device MB", inf.device_bytes / 1e6)
cam = host.camera_for(sd, res)
frame = torch.zeros(res[0] * res[1] * 3, dtype=torch.uint8, device="cuda")
ids = {}
for mode in ("fast", "strict"):
    g.set_traversal(mode)
    p = host.default_params("hw4", primary_only=1)
    st = abi.rt_stats()
    idt = torch.zeros(res[0] * res[1], dtype=torch.int32, device="cuda")
    for it in range(3):
        abi.check(g.L.rt_render_dev(g.h, C.byref(cam), C.byref(p), frame.data_ptr(), idt.data_ptr(), C.byref(st), None))
    ids[mode] = idt.cpu().numpy()
    rays = st.rays_primary + st.rays_shadow
    print(mode, "kernel_ms %.2f closest %.2f any %.2f rays %d => %.1f Mrays/s" % (st.kernel_ms, st.trace_closest_ms, st.trace_any_ms, rays, rays / st.kernel_ms / 1e3), flush=True)
    p.count_work = 1
    abi.check(g.L.rt_render_dev(g.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), None))
    print("   closest work/ray box %.1f tri %.1f exact %.2f fetch %.1f" % tuple(x / st.rays_primary for x in st.closest_work))
    print("   any work/ray     box %.1f tri %.1f exact %.2f fetch %.1f" % tuple(x / max(1, st.rays_shadow) for x in st.any_work), flush=True)
print("fast == strict ids:", np.array_equal(ids["fast"], ids["strict"]), "hits", int((ids["fast"] >= 0).sum()))
