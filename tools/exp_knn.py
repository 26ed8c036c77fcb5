"""experiment: kNN irradiance throughput on a synthetic photon map (bench.py's knn line), prints ms + lookups/s + parity
    python3 tools/exp_knn.py [photons=10e6] [queries=1920*1080]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracer_amd import host
g = host.Scene(bench.bench_scene())
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1920 * 1080
r = bench.knn_secondary(g, torch, n, q, with_cpu=n <= 20_000_000)
rf = r["roofline"]
print(json.dumps(dict(mlookups_per_s=r["value"] / 1e6, ms_per_step=r["ms_per_step"], parity=r["parity"],
                      waves=rf.get("waves"), walk_nodes_per_wave=rf.get("walk_nodes_per_wave"), inserted_per_wave=rf.get("inserted_per_wave"),
                      redo=rf.get("strict_redo_queries"), phase1=rf.get("phase1_fetches_per_lookup"))))
