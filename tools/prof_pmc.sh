#!/bin/bash
# rocprofv3 on each workload of bench.py: one --kernel-trace --stats run, then the PMC passes (each in a run of its
# own, as the pool requires): FETCH_SIZE, WRITE_SIZE, TCC hit/miss.  Output under gpurun_out/pmc/<workload>/<pass>/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
W="${1:-monkey_4k mesh_1m drt_1m knn_10m photon_frame}"
for w in $W; do
  for pass in stats FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo "$pass" | tr ' ' '_')
    out=gpurun_out/pmc/$w/$tag
    rm -rf "$out"; mkdir -p "$out"
    if [ "$pass" = stats ]; then
      timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o r -- python3 tools/prof_target.py "$w" > "$out/stdout.txt" 2> "$out/stderr.txt"
    else
      timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$out" -o r -- python3 tools/prof_target.py "$w" > "$out/stdout.txt" 2> "$out/stderr.txt"
    fi
    rc=$?
    echo "$w $tag rc=$rc"; tail -n 1 "$out/stdout.txt"
    if [ $rc -ne 0 ]; then tail -n 5 "$out/stderr.txt"; exit $rc; fi
  done
done
find gpurun_out/pmc -name "*.csv" | head -40
