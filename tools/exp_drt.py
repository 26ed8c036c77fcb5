#!/usr/bin/env python3
"""drt_1m secondary of bench.py alone (RT_CHUNK_LOG2 etc. from the environment)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
r = bench.drt_secondary(torch, 0)
print(json.dumps({k: r[k] for k in ("value", "ms_per_frame", "passes", "trace_closest_ms", "trace_any_ms")}))
