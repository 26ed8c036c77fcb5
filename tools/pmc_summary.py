#!/usr/bin/env python3
"""Summarise gpurun_out/pmc/<workload>/<pass>/*.csv (tools/prof_pmc.sh) into profiles/:
   profiles/<tag>_kernel_stats_<workload>.txt   rocprofv3 --kernel-trace --stats (per-kernel durations)
   profiles/traffic_<tag>.json                  per workload and kernel: HBM bytes per launch from the PMC passes
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE tallies
128-B read requests at 64 B (MI355X_MICROARCH.md, HBM).  That calibration was made on wide coalesced streams; the
traversal kernels read records on the scalar path and the kNN kernel reads scattered 16-B records, so treat the
absolute as approximate (the guide: "other access widths are uncalibrated").  Infinity-Cache hits are counted too."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
base = os.path.join(ROOT, "gpurun_out", "pmc")
out = {}
for wdir in sorted(glob.glob(os.path.join(base, "*"))):
    w = os.path.basename(wdir)
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(wdir, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    ks = os.path.join(wdir, "stats", "r_kernel_stats.csv")
    dur = {}
    if os.path.exists(ks):
        rows = list(csv.DictReader(open(ks)))
        with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{w}.txt"), "w") as fo:
            fo.write(f"# rocprofv3 --kernel-trace --stats -- python3 tools/prof_target.py {w}\n")
            fo.write("# kernel, calls, total_us, average_us, percent\n")
            for r in rows:
                fo.write(f"{r['Name'][:90]}, {r['Calls']}, {float(r['TotalDurationNs']) / 1e3:.0f}, {float(r['AverageNs']) / 1e3:.1f}, {r['Percentage']}\n")
                dur[r["Name"]] = float(r["AverageNs"]) / 1e3
    wout = {}
    import re
    for k, c in per.items():
        m = re.search(r"\b(k_[a-z0-9_]+)", k)
        if not m:
            continue
        name = m.group(1)
        mean = lambda v: sum(v) / len(v) if v else None
        fs, ws = mean(c.get("FETCH_SIZE", [])), mean(c.get("WRITE_SIZE", []))
        hit, miss = sum(c.get("TCC_HIT_sum", [])), sum(c.get("TCC_MISS_sum", []))
        e = dict(launches=len(c.get("FETCH_SIZE", [])), fetch_size_kib=fs, write_size_kib=ws,
                 hbm_bytes_per_launch=(2 * (fs or 0) + (ws or 0)) * 1024 if fs is not None else None,
                 l2_hit_rate=hit / (hit + miss) if hit + miss > 0 else None, avg_us=dur.get(k))
        if name in wout and (wout[name].get("avg_us") or 0) > (e.get("avg_us") or 0):
            continue                      # several instantiations of one kernel: keep the one that takes the time
        if e["hbm_bytes_per_launch"] and e["avg_us"]:
            e["hbm_gb_per_s"] = e["hbm_bytes_per_launch"] / (e["avg_us"] * 1e-6) / 1e9
        wout[name] = e
    out[w] = wout
json.dump(out, open(os.path.join(ROOT, "profiles", f"traffic_{tag}.json"), "w"), indent=1)
for w, ks in out.items():
    for k, e in ks.items():
        if e["avg_us"] and e["avg_us"] > 50:
            print(f"{w:13s} {k:22s} avg {e['avg_us']:10.1f} us  hbm {((e['hbm_bytes_per_launch'] or 0) / 1e6):10.1f} MB/launch "
                  f"{e.get('hbm_gb_per_s', 0):8.1f} GB/s  L2 hit {e['l2_hit_rate'] if e['l2_hit_rate'] is None else round(e['l2_hit_rate'], 3)}")
