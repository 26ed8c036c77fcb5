#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace.  Each GPU step has its own
# time limit; after a crash / time-out (anything but pytest's "tests failed" code 1) nothing else runs.
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
step() { echo "=== $* ===" ; }

step pytest
timeout -k 10 600 python -u -m pytest tests -m gpu -v --timeout 180 --timeout-method thread > gpurun_out/pytest.log 2>&1
rc=$?; tail -n 40 gpurun_out/pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi

step smoke
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -n 5 gpurun_out/smoke.log; echo "smoke rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi

step bench
timeout -k 10 500 python -u bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err
rc=$?; tail -n 5 gpurun_out/bench.err; cat gpurun_out/bench.json; echo "bench rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi

step rocprof
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ${PROF_ARGS:-} > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
rc=$?; echo "rocprof rc=$rc"; tail -n 3 gpurun_out/prof.err
find gpurun_out/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 20 "$f"
exit $rc
