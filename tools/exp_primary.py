"""k_primary_only on the 1 M-triangle mesh at 3840x2160 (bench.py's mesh_1m line) and on the monkey: kernel ms and the work
counters, for experiments with the builder's knobs (RT_BVH_LEAF, RT_BVH_CONE_MAX, ...) set in the environment.
usage: exp_primary.py [frames]"""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from raytracer_amd import abi, host, scene, synth
import bench
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
RES = (3840, 2160)
for name, sd in (("mesh_1m", synth.mesh_scene(1000, 500)), ("monkey", scene.parse_scene_file(os.path.join(ROOT, "tests", "golden", "monkey.txt")))):
    gs = host.Scene(sd)
    cam = host.camera_for(sd, RES); p = host.default_params("hw4", primary_only=1)
    frame = torch.zeros(RES[0] * RES[1] * 3, dtype=torch.uint8, device="cuda"); ids = torch.zeros(RES[0] * RES[1], dtype=torch.int32, device="cuda")
    st = abi.rt_stats(); sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = []
    for i in range(frames + 2):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), ids.data_ptr(), C.byref(st), sp))
        if i >= 2: ms.append(st.trace_closest_ms)
    w = bench.render_work(gs, cam, p, sp, torch)
    print(f"{name}: k_primary_only {min(ms):.4f} ms min, {sum(ms) / len(ms):.4f} ms mean; per primary ray: {w.closest_work[0] / max(1, w.rays_primary):.1f} box tests, "
          f"{w.closest_work[1] / max(1, w.rays_primary):.1f} triangle records; per shadow ray: {w.any_work[0] / max(1, w.rays_shadow):.1f} / {w.any_work[1] / max(1, w.rays_shadow):.1f}; "
          f"ids checksum {int(ids.to(torch.int64).sum())}")
    gs.close()
