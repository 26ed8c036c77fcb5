#!/usr/bin/env python3
"""photon_frame line of bench.py alone (RT_KNN_LIST etc. from the environment): python3 tools/exp_pframe.py [full]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
full = len(sys.argv) > 1 and sys.argv[1] == "full"
r = bench.photon_frame_secondary(torch, 0, 10_000_000, 1_000_000, (1920, 1080), tag="photon_frame_full") if full else bench.photon_frame_secondary(torch, 0)
def find(o, path=""):
    import numpy as np
    if isinstance(o, dict):
        for k, v in o.items(): find(v, path + "/" + str(k))
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o): find(v, path + f"[{i}]")
    elif isinstance(o, (np.ndarray, np.generic)):
        print("numpy value at", path, type(o), file=sys.stderr)
find(r)
rf = r["roofline"]
print(bench.dumps(dict({k: r[k] for k in ("value", "lookups_per_s", "ms_per_frame", "knn_lookups")}, one_stream_ms=r.get("ms_per_frame_on_one_stream"), knn_ms=rf.get("ms_per_frame_in_the_estimate"),
                       waves=rf.get("waves"), walk_nodes_per_wave=rf.get("walk_nodes_per_wave"), inserted_per_wave=rf.get("inserted_per_wave"), redo=rf.get("strict_redo_queries"), parity=r["parity"]["parity_sample_ok"])))
