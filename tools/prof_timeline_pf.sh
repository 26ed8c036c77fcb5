#!/bin/bash
# kernel timeline of the timed (two-stream) frame of config 4 at full size (tools/exp_pframe.py full): start, end, duration, queue, kernel
# usage: tools/prof_timeline_pf.sh [min_us=2000]   -> gpurun_out/timeline_pframe_full.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/tl_pf; rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d "$out" -o r -- python3 tools/exp_pframe.py full > "$out/stdout.txt" 2> "$out/stderr.txt"
rc=$?; tail -n 1 "$out/stdout.txt" | cut -c1-200; if [ $rc -ne 0 ]; then tail -n 5 "$out/stderr.txt"; exit $rc; fi
python3 - "$out" "${1:-2000}" <<'PY'
import csv, glob, sys
out, min_us = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
# frames start with k_pm_primary / the first generate kernel of a frame: find the frame boundaries by the eye-ray generation kernel
gens = [i for i, r in enumerate(rows) if "k_pm_generate" in r[3]]
print("generate launches:", len(gens))
a, b = gens[1], gens[2] if len(gens) > 2 else len(rows)      # the second frame = the timed one (the first is the instrumented frame)
t0 = rows[a][0]
qs = {}
with open("gpurun_out/timeline_pframe_full.txt", "w") as fo:
    fo.write("# start_ms end_ms duration_ms queue kernel\n")
    for s, e, q, k in rows[a:b]:
        if e - s < min_us * 1000: continue
        qn = qs.setdefault(q, "q%d" % (len(qs) + 1))
        fo.write(f"{(s - t0) / 1e6:9.2f} {(e - t0) / 1e6:9.2f} {(e - s) / 1e6:8.2f} {qn} {k[:60]}\n")
print(open("gpurun_out/timeline_pframe_full.txt").read())
PY
rm -rf "$out"
