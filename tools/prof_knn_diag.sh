#!/bin/bash
# diagnostic counters of the radiance-estimate kernels: texture-addresser / L1 busy, L1->L2 requests, SQ issue and wait
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for w in ${1:-knn_10m photon_frame}; do
  i=0
  for pass in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
              "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS"; do
    i=$((i+1)); out=gpurun_out/diag/$w/p$i; rm -rf "$out"; mkdir -p "$out"
    timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$out" -o r -- python3 tools/prof_target.py "$w" > "$out/stdout.txt" 2> "$out/stderr.txt"
    echo "$w p$i rc=$? $(tail -n 1 $out/stdout.txt)"; tail -n 2 "$out/stderr.txt"
  done
done
python3 - <<'PY'
import csv, glob, collections
for w in ("knn_10m", "photon_frame"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/diag/{w}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_irr_main" in r["Kernel_Name"] or "k_photon_knn" in r["Kernel_Name"]:
                per[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in per.items():
        print(w, k, {n: sum(v) / len(v) for n, v in sorted(c.items())}, "launches", max(len(v) for v in c.values()))
PY
