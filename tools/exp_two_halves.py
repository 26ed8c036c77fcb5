#!/usr/bin/env python3
"""does config 3's frame gain from running the latency-bound deep passes of one part of the frame beside the scalar-bound wide
passes of another?  K scenes (one per share of the tiles), one host thread and one stream each, rendering their shares
concurrently and repeatedly; against one scene rendering the whole frame.   python3 tools/exp_two_halves.py [K=2] [frames=6]"""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from raytracer_amd import abi, host, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sd = synth.mesh_scene(1000, 500, reflect=0.3)
cam = host.camera_for(sd, bench.HEAD_RES)
scenes = [host.Scene(sd) for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
bufs = [torch.zeros(bench.HEAD_RES[0] * bench.HEAD_RES[1] * 3, dtype=torch.uint8, device="cuda") for _ in range(K)]
def params(r, k):
    kw = dict(tile_size=64, tile_rank=r, tile_nranks=k, packed_output=1) if k > 1 else {}
    return host.default_params("hw4", spp=bench.HEAD_SPP, **kw, **bench.DRT_KW)
def run(r, k, n, delay=0.0):
    p = params(r, k)
    if delay: time.sleep(delay)
    for _ in range(n):
        abi.check(scenes[r].L.rt_render_dev(scenes[r].h, C.byref(cam), C.byref(p), bufs[r].data_ptr(), None, None, C.c_void_p(streams[r].cuda_stream)))
    streams[r].synchronize()
# whole frame, one scene
run(0, 1, 2); torch.cuda.synchronize()
t0 = time.perf_counter(); run(0, 1, frames); torch.cuda.synchronize()
print("one scene, whole frame: %.2f ms per frame" % ((time.perf_counter() - t0) / frames * 1e3), flush=True)
# shares one after another on one stream
for r in range(K): run(r, K, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    for r in range(K): run(r, K, 1)
torch.cuda.synchronize()
print("%d shares one after another: %.2f ms per frame" % (K, (time.perf_counter() - t0) / frames * 1e3), flush=True)
for delay_ms in (0.0, 8.0, 16.0):
    th = [threading.Thread(target=run, args=(r, K, frames, r * delay_ms * 1e-3)) for r in range(K)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0 - (K - 1) * delay_ms * 1e-3
    print("%d shares side by side, start delay %.0f ms: %.2f ms per frame" % (K, delay_ms, dt / frames * 1e3), flush=True)
