#!/bin/bash
# SQ counters of one workload of tools/prof_target.py (default mesh_1m): how busy the vector / scalar units are
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
w="${1:-mesh_1m}"
out=gpurun_out/sq/$w; rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$out" -o r -- python3 tools/prof_target.py "$w" > "$out/stdout.txt" 2> "$out/stderr.txt"
echo "rc=$?"; tail -n 1 "$out/stdout.txt"
