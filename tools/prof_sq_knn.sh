#!/bin/bash
# SQ counters of the kNN workload with the wave-shared irradiance kernel (RT_KNN=fast)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/sq/knn_wave; rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out" -o r -- python3 tools/prof_target.py knn_10m > "$out/stdout.txt" 2> "$out/stderr.txt"
echo "rc=$?"; tail -n 1 "$out/stdout.txt"
out=gpurun_out/sq/knn_wave_stats; rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o r -- python3 tools/prof_target.py knn_10m > "$out/stdout.txt" 2> "$out/stderr.txt"
echo "rc=$?"; tail -n 1 "$out/stdout.txt"
