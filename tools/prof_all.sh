#!/bin/bash
# rocprofv3 on each workload of bench.py (tools/prof_target.py runs ONE workload, so a kernel name maps to one workload):
# one --kernel-trace --stats run, then the PMC passes, each in a run of its own as the pool requires: FETCH_SIZE,
# WRITE_SIZE, TCC hit/miss, two SQ passes (<= 8 SQ counters each) and the scalar data cache's.  Output: gpurun_out/pmc/<workload>/<pass>/.
# tools/prof_counters.py turns that into profiles/<tag>_counters.json and profiles/<tag>_kernel_stats_<workload>.txt.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
W="${1:-monkey_4k mesh_1m drt_1m knn_10m photon_frame photon_frame_full stress_mesh_10m stress_knn_200m}"
LIMIT="${2:-300}"
mkdir -p gpurun_out/pmc
# what the counters belong to: the hash of the kernel sources on this box (bench.py flags counters whose hash is not the live one)
python3 -c "import bench; print(bench.kernel_source_sha())" > gpurun_out/pmc/kernel_source_sha.txt || exit 1
for w in $W; do
  cp gpurun_out/pmc/kernel_source_sha.txt /tmp/sha_$w.txt
  for pass in stats FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" \
              "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
              "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
              "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE"; do
    tag=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
    out=gpurun_out/pmc/$w/$tag
    rm -rf "$out"; mkdir -p "$out"
    if [ "$pass" = stats ]; then
      timeout -s KILL "$LIMIT" rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o r -- python3 tools/prof_target.py "$w" > "$out/stdout.txt" 2> "$out/stderr.txt"
    else
      timeout -s KILL "$LIMIT" rocprofv3 --pmc $pass --output-format csv -d "$out" -o r -- python3 tools/prof_target.py "$w" > "$out/stdout.txt" 2> "$out/stderr.txt"
    fi
    rc=$?
    echo "$w $tag rc=$rc $(tail -n 1 "$out/stdout.txt")"
    if [ $rc -ne 0 ]; then tail -n 5 "$out/stderr.txt"; exit $rc; fi
  done
  cp /tmp/sha_$w.txt gpurun_out/pmc/$w/kernel_source_sha.txt
  python3 tools/prof_counters.py --box $w || exit 1      # summary.json instead of tens of MB of CSV (gpurun_out travels back)
done
