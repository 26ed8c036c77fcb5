"""experiment: kNN irradiance throughput on a synthetic 10M-photon map (bench.py's secondary), prints ms + lookups/s"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracer_amd import host, abi
sd = bench.bench_scene()
g = host.Scene(sd)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1920 * 1080
r = bench.knn_secondary(g, torch, n, q)
print({k: r[k] for k in ("value", "ms_per_step", "map_build_s")}, r["roofline"], r["cpu_baseline"]["value"])
