#!/usr/bin/env python3
"""Summarise gpurun_out/pmc/<workload>/<pass>/*.csv (tools/prof_all.sh) into profiles/:
   profiles/<tag>_kernel_stats_<workload>.txt   rocprofv3 --kernel-trace --stats (per-kernel durations)
   profiles/<tag>_counters.json                 per workload and kernel, PER LAUNCH: HBM bytes from the PMC passes, L2 hit
                                                rate, SQ instruction / cycle counters, and the busy fractions derived from them
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE tallies
128-B read requests at 64 B (MI355X_MICROARCH.md, HBM).  That calibration was made on wide coalesced streams; scattered
16-B reads and scalar-path reads are "uncalibrated" per the guide, so the absolute is approximate; Infinity-Cache hits
are counted too.
Issue model (MI355X_MICROARCH.md:54,485): a wave64 vector instruction occupies its SIMD-32 for 2 cycles; 1 024 SIMDs at
2.4 GHz.  Scalar unit: one SALU / SMEM / branch instruction per cycle and CU, 256 CUs.  SQ_WAVE_CYCLES / SQ_WAIT_* count
quad-cycles per wave; their ratios are what is used."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "r03"
base = os.path.join(ROOT, "gpurun_out", "pmc")
outp = os.path.join(ROOT, "profiles", f"{tag}_counters.json")
out = json.load(open(outp)) if os.path.exists(outp) else {}
out.pop("_source", None)
meta = out.setdefault("_meta", {})       # per workload: the commit and kernel-source hash its counters were collected at
import subprocess
import time
try:
    head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    if subprocess.run(["git", "status", "--porcelain", "raytracer_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout.strip():
        head += "+uncommitted"
except Exception:
    head = None
mean = lambda v: sum(v) / len(v) if v else None


def reduce_csvs(wdir):
    """the CSVs of one workload -> {"per": {kernel: {counter: [mean, sum, launches]}}, "stats": [[name, calls, total_ns, avg_ns, pct]]}
    (what the rest of this script needs; small enough to travel back from the GPU box)"""
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(wdir, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    red = {k: {cn: [mean(v), sum(v), len(v)] for cn, v in c.items()} for k, c in per.items()}
    stats = []
    ks = os.path.join(wdir, "stats", "r_kernel_stats.csv")
    if os.path.exists(ks):
        for r in csv.DictReader(open(ks)):
            stats.append([r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), r["Percentage"]])
    return dict(per=red, stats=stats)


if len(sys.argv) > 2 and sys.argv[1] == "--box":
    # on the GPU box, after a workload's passes (tools/prof_all.sh): keep the summary, drop the CSVs (tens of MB)
    wdir = os.path.join(base, sys.argv[2])
    json.dump(reduce_csvs(wdir), open(os.path.join(wdir, "summary.json"), "w"))
    for f in glob.glob(os.path.join(wdir, "*", "*.csv")):
        os.remove(f)
    sys.exit(0)

for wdir in sorted(glob.glob(os.path.join(base, "*"))):
    if not os.path.isdir(wdir):
        continue
    w = os.path.basename(wdir)
    shaf = os.path.join(wdir, "kernel_source_sha.txt")
    meta[w] = dict(commit=head, kernel_source_sha=open(shaf).read().strip() if os.path.exists(shaf) else None,
                   reduced=time.strftime("%Y-%m-%d %H:%M"))
    sm = os.path.join(wdir, "summary.json")
    red = json.load(open(sm)) if os.path.exists(sm) else reduce_csvs(wdir)
    per = red["per"]
    dur = {}
    if red["stats"]:
        with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{w}.txt"), "w") as fo:
            fo.write(f"# rocprofv3 --kernel-trace --stats -- python3 tools/prof_target.py {w}\n")
            fo.write("# kernel, calls, total_us, average_us, percent\n")
            for name, calls, tot, avg, pct in red["stats"]:
                fo.write(f"{name[:90]}, {calls}, {tot / 1e3:.0f}, {avg / 1e3:.1f}, {pct}\n")
                dur[name] = (avg / 1e3, calls)
    wout = {}
    for k, c in per.items():
        m = re.search(r"\b(k_[a-z0-9_]+(?:<[^>]*>)?)", k)
        if not m:
            continue
        name = m.group(1).replace(" ", "")
        e = {cn: v[0] for cn, v in c.items()}
        fs, ws = e.get("FETCH_SIZE"), e.get("WRITE_SIZE")
        hit, miss = c.get("TCC_HIT_sum", [0, 0, 0])[1], c.get("TCC_MISS_sum", [0, 0, 0])[1]
        e["launches"] = max(v[2] for v in c.values())
        e["hbm_bytes_per_launch"] = (2 * (fs or 0) + (ws or 0)) * 1024 if fs is not None else None
        e["l2_hit_rate"] = hit / (hit + miss) if hit + miss > 0 else None
        if e.get("SQC_DCACHE_REQ"):
            e["scalar_cache_hit_rate"] = (e.get("SQC_DCACHE_HITS") or 0.0) / e["SQC_DCACHE_REQ"]
        if k in dur:
            e["avg_us"], e["calls_in_stats_run"] = dur[k]
        t = (e.get("avg_us") or 0) * 1e-6
        if t > 0:
            d = {}
            if e["hbm_bytes_per_launch"]:
                d["hbm_gb_per_s"] = e["hbm_bytes_per_launch"] / t / 1e9
                d["hbm_traffic_fraction_of_8TBs"] = d["hbm_gb_per_s"] / 8000.0
            if e.get("SQ_INSTS_VALU"):
                d["valu_issue_fraction"] = e["SQ_INSTS_VALU"] * 2 / (1024 * 2.4e9) / t
                d["valu_instructions_per_wave"] = e["SQ_INSTS_VALU"] / max(1.0, e.get("SQ_WAVES") or 1.0)
                sc = (e.get("SQ_INSTS_SALU") or 0) + (e.get("SQ_INSTS_SMEM") or 0) + (e.get("SQ_INSTS_BRANCH") or 0)
                d["scalar_issue_fraction"] = sc / (256 * 2.4e9) / t
                d["scalar_instructions_per_wave"] = sc / max(1.0, e.get("SQ_WAVES") or 1.0)
            if e.get("SQ_WAVE_CYCLES"):
                d["wave_cycles_waiting_to_issue"] = (e.get("SQ_WAIT_INST_ANY") or 0) / e["SQ_WAVE_CYCLES"]
                d["wave_cycles_waiting_on_memory"] = (e.get("SQ_WAIT_ANY") or 0) / e["SQ_WAVE_CYCLES"]
            e["derived"] = d
        if name in wout and (wout[name].get("avg_us") or 0) * (wout[name].get("calls_in_stats_run") or 1) > (e.get("avg_us") or 0) * (e.get("calls_in_stats_run") or 1):
            continue
        wout[name] = e
    out[w] = wout
json.dump(out, open(outp, "w"), indent=1)
for w, ks in out.items():
    if w.startswith("_"):
        continue
    for k, e in ks.items():
        if e.get("avg_us") and e["avg_us"] * e.get("calls_in_stats_run", 1) > 500:
            d = e.get("derived", {})
            print(f"{w:13s} {k[:34]:34s} avg {e['avg_us']:9.1f} us x{e.get('calls_in_stats_run', 0):4d}  hbm {((e['hbm_bytes_per_launch'] or 0) / 1e6):9.1f} MB "
                  f"({d.get('hbm_traffic_fraction_of_8TBs', 0):.3f})  L2 {e['l2_hit_rate'] if e['l2_hit_rate'] is None else round(e['l2_hit_rate'], 3)}  "
                  f"valu {d.get('valu_issue_fraction', 0):.2f} scalar {d.get('scalar_issue_fraction', 0):.2f} "
                  f"wait-issue {d.get('wave_cycles_waiting_to_issue', 0):.2f} wait-mem {d.get('wave_cycles_waiting_on_memory', 0):.2f}")
