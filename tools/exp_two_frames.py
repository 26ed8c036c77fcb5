#!/usr/bin/env python3
"""Experiment: do the shallow (issue-bound) and the deep (latency-bound) halves of BASELINE configuration 3's frame hide behind
each other?  Two scenes render the same frame from two host threads on two streams, the second started `offset_ms` after
the first; if 2 x K frames take clearly less than 2 x K x (one frame alone), pipelining the frame's chunks would pay.
    python3 tools/exp_two_frames.py [frames=6] [offset_ms=40] [res_scale=1]"""
import ctypes as C
import sys, os, time, threading, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracer_amd import abi, host, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
offset = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
half = len(sys.argv) > 3 and sys.argv[3] == "half"
sd = synth.mesh_scene(1000, 500, reflect=0.3)
res = bench.HEAD_RES
scenes = [host.Scene(sd), host.Scene(sd)]
cam = host.camera_for(sd, res)
# with "half": each scene renders half of the frame's tiles (rank r of 2) -- the same total work as ONE frame
ps = [host.default_params("hw4", spp=bench.HEAD_SPP, **(dict(tile_size=64, tile_rank=r, tile_nranks=2, packed_output=1) if half else {}), **bench.DRT_KW) for r in range(2)]
bufs = [torch.zeros(res[0] * res[1] * 3, dtype=torch.uint8, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def run(k, n):
    for _ in range(n):
        abi.check(scenes[k].L.rt_render_dev(scenes[k].h, C.byref(cam), C.byref(ps[k]), bufs[k].data_ptr(), None, None, C.c_void_p(streams[k].cuda_stream)))
    streams[k].synchronize()

run(0, 2); run(1, 2)
t0 = time.perf_counter(); run(0, frames); alone = (time.perf_counter() - t0) / frames * 1e3
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(0, frames))]
th[0].start(); time.sleep(offset * 1e-3)
th.append(threading.Thread(target=run, args=(1, frames))); th[1].start()
for t in th: t.join()
both = (time.perf_counter() - t0) * 1e3
print(json.dumps(dict(one_scene_ms_per_call=alone, two_scenes_total_ms=both, calls_each=frames, ms_per_pair=(both - offset) / frames, half=half)))
