#!/usr/bin/env python3
"""Summarise gpurun_out/sq/monkey*/ (tools/prof_sq_monkey.sh) into profiles/<tag>_sq_monkey_4k.json: the SQ counters of
the headline kernel per launch, and the time its vector instructions alone need to issue."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kernel_ms = float(sys.argv[2]) if len(sys.argv) > 2 else None
per = collections.defaultdict(list)
for d in ("monkey", "monkey2"):
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "sq", d, "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if "k_primary_only" in row["Kernel_Name"]:
                per[row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in sorted(per.items())}
out["launches_averaged"] = min(len(v) for v in per.values())
valu_ms = out["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9) * 1e3
out["derived"] = {
    "valu_issue_ms": valu_ms,
    "valu_instructions_per_wave": out["SQ_INSTS_VALU"] / out["SQ_WAVES"],
    "kernel_ms": kernel_ms,
    "valu_issue_fraction": (valu_ms / kernel_ms) if kernel_ms else None,
    "note": "per launch of k_primary_only (monkey 3840x2160); 1024 SIMDs at 2.4 GHz, 4 cycles per wave64 vector instruction; "
            "kernel_ms = the same launch's duration in the rocprofv3 --kernel-trace run (profiles/<tag>_kernel_stats_monkey_4k.txt)",
}
json.dump({"k_primary_only": out}, open(os.path.join(ROOT, "profiles", f"{tag}_sq_monkey_4k.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
