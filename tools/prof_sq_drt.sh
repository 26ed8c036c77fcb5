#!/bin/bash
# SQ counter pass on the distributed-ray-tracing workload: where do the wave cycles of the per-lane kernels go
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/sq/drt; rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$out" -o r -- python3 tools/prof_target.py drt_1m > "$out/stdout.txt" 2> "$out/stderr.txt"
echo "rc=$?"; tail -n 1 "$out/stdout.txt"
