#!/usr/bin/env python3
"""One workload of bench.py, alone, for rocprofv3 (so a kernel name maps to one workload):
    python3 tools/prof_target.py monkey_4k | mesh_1m | drt_1m | knn_10m | photon_frame | photon_frame_full | stress_mesh_10m | stress_knn_200m"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch            # noqa: E402
import bench            # noqa: E402
from raytracer_amd import abi, host   # noqa: E402

what = sys.argv[1]
if what == "monkey_4k":
    sd = bench.bench_scene()
    gs = host.Scene(sd)
    cam = host.camera_for(sd, bench.BASE_RES)
    p = host.default_params("hw4", primary_only=1)
    frame = torch.zeros(bench.BASE_RES[0] * bench.BASE_RES[1] * 3, dtype=torch.uint8, device="cuda")
    st = abi.rt_stats()
    for _ in range(5):
        abi.check(gs.L.rt_render_dev(gs.h, C.byref(cam), C.byref(p), frame.data_ptr(), None, C.byref(st), None))
    print("monkey_4k closest ms", st.trace_closest_ms)
elif what == "mesh_1m":
    r = bench.mesh_secondary(torch, 0, steps=3)
    print("mesh_1m", r["value"], r["roofline"]["avg_launch_ms"])
elif what == "knn_10m":
    gs = host.Scene(bench.bench_scene())
    r = bench.knn_secondary(gs, torch, 10_000_000, 1920 * 1080, steps=2, with_cpu=False)
    print("knn_10m", r["value"], r["ms_per_step"])
elif what == "drt_1m":
    # every kernel on one stream, as in the frame bench.py takes the roofline's kernel times from: side by side on two
    # streams (the default, profiles/r02_timeline_drt_1m.txt) kernels time each other
    os.environ.setdefault("RT_RENDER_OVERLAP", "0")
    from raytracer_amd import synth
    sd = synth.mesh_scene(1000, 500, reflect=0.3)
    gs = host.Scene(sd)
    cam = host.camera_for(sd, bench.HEAD_RES)
    p = host.default_params("hw4", spp=bench.HEAD_SPP, **bench.DRT_KW)
    r = bench.drt_analysis(torch, gs, sd, cam, p, None, with_cpu=False)
    print("drt_1m", r["rays"] / r["frame_ms"] / 1e3, r["frame_ms"])
elif what == "photon_frame":
    r = bench.photon_frame_secondary(torch, 0)
    print("photon_frame", r["value"], r["lookups_per_s"])
elif what == "photon_frame_full":
    r = bench.photon_frame_secondary(torch, 0, 10_000_000, 1_000_000, (1920, 1080), tag="photon_frame_full")
    print("photon_frame_full", r["value"], r["lookups_per_s"])
elif what == "stress_mesh_10m":
    r = bench.mesh_secondary(torch, 0, steps=3, nu=3162, nv=1581, tag="stress_mesh_10m")
    print("stress_mesh_10m", r["value"], r["roofline"]["avg_launch_ms"])
elif what == "stress_knn_200m":
    gs = host.Scene(bench.bench_scene())
    r = bench.knn_secondary(gs, torch, 200_000_000, 1920 * 1080, steps=2, with_cpu=False)
    print("stress_knn_200m", r["value"], r["ms_per_step"])
else:
    sys.exit("unknown workload")
