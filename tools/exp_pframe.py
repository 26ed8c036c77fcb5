#!/usr/bin/env python3
"""photon_frame secondary of bench.py alone (RT_KNN etc. from the environment)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
r = bench.photon_frame_secondary(torch, 0)
print(json.dumps({k: r[k] for k in ("value", "lookups_per_s", "ms_per_frame", "knn_lookups", "knn")}))
