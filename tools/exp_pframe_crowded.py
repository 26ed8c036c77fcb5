#!/usr/bin/env python3
"""config 4 at full size on a crowded GPU: most of the device memory is taken before the frame (a tensor of `hold` GB), the
renderer must shrink its chunks instead of failing (render_photon.cpp).   python3 tools/exp_pframe_crowded.py [hold_gb=230]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
hold = float(sys.argv[1]) if len(sys.argv) > 1 else 230.0
free, total = torch.cuda.mem_get_info()
print("free %.1f GB of %.1f" % (free / 1e9, total / 1e9), flush=True)
blk = torch.empty(int(hold * 1e9), dtype=torch.uint8, device="cuda")
free2, _ = torch.cuda.mem_get_info()
print("holding %.0f GB, free now %.1f GB" % (hold, free2 / 1e9), flush=True)
r = bench.photon_frame_secondary(torch, 0, 10_000_000, 1_000_000, (1920, 1080), tag="photon_frame_full")
free3, _ = torch.cuda.mem_get_info()
print(bench.dumps(dict(ms_per_frame=r["ms_per_frame"], lookups_per_s=r["lookups_per_s"], launches=r["roofline"].get("launches_per_frame"),
                       parity=r["parity"]["parity_sample_ok"], free_after_gb=free3 / 1e9)))
