#!/bin/bash
# kernel timeline of one two-stream frame of config 3 (the last timed frame of tools/exp_drt.py): start, end, duration, queue, kernel
# usage: tools/prof_timeline.sh <tag>   -> gpurun_out/timeline_<tag>.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-a}; out=gpurun_out/tl_$tag; rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d "$out" -o r -- python3 tools/exp_drt.py 2 ${2:-1} > "$out/stdout.txt" 2> "$out/stderr.txt"
rc=$?; tail -n 1 "$out/stdout.txt" | cut -c1-150; if [ $rc -ne 0 ]; then tail -n 5 "$out/stderr.txt"; exit $rc; fi
python3 - "$out" "$tag" "${3:-100}" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
gens = [i for i, r in enumerate(rows) if "k_generate_drt" in r[3]]
a = gens[-2]; b = gens[-1]                      # the last timed frame (the stats frame follows it)
t0 = rows[a][0]
qs = {}
with open(f"gpurun_out/timeline_{tag}.txt", "w") as fo:
    fo.write("# start_ms end_ms duration_ms queue kernel\n")
    for s, e, q, k in rows[a:b]:
        if e - s < int(sys.argv[3]) * 1000: continue
        qn = qs.setdefault(q, "q%d" % (len(qs) + 1))
        fo.write(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f} {qn} {k[:64]}\n")
print(open(f"gpurun_out/timeline_{tag}.txt").read())
PY
rm -rf "$out"
