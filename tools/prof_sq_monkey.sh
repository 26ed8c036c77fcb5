#!/bin/bash
# SQ counters of the headline workload (monkey 4K, primary + shadow): how busy the vector / scalar units are
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/sq/monkey; rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$out" -o r -- python3 tools/prof_target.py monkey_4k > "$out/stdout.txt" 2> "$out/stderr.txt"
echo "rc=$?"; tail -n 1 "$out/stdout.txt"
out=gpurun_out/sq/monkey2; rm -rf "$out"; mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --output-format csv -d "$out" -o r -- python3 tools/prof_target.py monkey_4k > "$out/stdout.txt" 2> "$out/stderr.txt"
echo "rc=$?"; tail -n 1 "$out/stdout.txt"
