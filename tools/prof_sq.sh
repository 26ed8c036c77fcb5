#!/bin/bash
# SQ counter pass on the kNN workload, strict vs fast kernel: where do the wave cycles go
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for m in strict fast; do
  out=gpurun_out/sq/$m; rm -rf "$out"; mkdir -p "$out"
  RT_KNN=$m timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out" -o r -- python3 tools/prof_target.py knn_10m > "$out/stdout.txt" 2> "$out/stderr.txt"
  echo "$m rc=$?"; tail -n 1 "$out/stdout.txt"
done
