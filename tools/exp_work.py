#!/usr/bin/env python3
"""work counters of the headline frame: lane-level tests vs wave-level record fetches"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracer_amd import abi, host
sd = bench.bench_scene()
gs = host.Scene(sd)
cam = host.camera_for(sd, bench.BASE_RES)
p = host.default_params("hw4", primary_only=1)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
a = bench.algorithmic_bytes_per_ray(gs, cam, p, sp)
print({k: round(v, 2) if isinstance(v, float) else v for k, v in a.items()})
