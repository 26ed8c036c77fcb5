#!/bin/bash
# diagnostic counters of the per-lane walk kernels on config 3 (one-stream frames): L1 / texture-addresser, SQ issue, wait, lane activity
# usage: [env RT_LANE_STEP=.. ...] tools/prof_step_diag.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-a}
i=0
for pass in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU" \
            "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); out=gpurun_out/stepdiag/$tag/p$i; rm -rf "$out"; mkdir -p "$out"
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$out" -o r -- python3 tools/prof_target.py drt_1m > "$out/stdout.txt" 2> "$out/stderr.txt"
  rc=$?; echo "$tag p$i rc=$rc $(tail -n 1 $out/stdout.txt)"; if [ $rc -ne 0 ]; then tail -n 5 "$out/stderr.txt"; exit $rc; fi
done
python3 - "$tag" <<'PY'
import csv, glob, collections, sys, json
tag = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/stepdiag/{tag}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bvh_step" in k or "bvh_lanes" in k or "k_paths" in k:
            per[k[:32]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in per.items():
    out[k] = {n: sum(v) / len(v) for n, v in sorted(c.items())}
    out[k]["launches"] = max(len(v) for v in c.values())
    print(tag, k, json.dumps(out[k]))
json.dump(out, open(f"gpurun_out/stepdiag/{tag}.json", "w"), indent=1)
PY
rm -rf gpurun_out/stepdiag/$tag
