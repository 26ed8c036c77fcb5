#!/bin/bash
# One GPU-box session: parity tests, smoke, bench.  Each GPU step has its own time limit; after a crash / time-out
# (anything but pytest's "tests failed" code 1 or bench.py's "a line failed" code 4) nothing else runs.
set -u
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
step() { echo "=== $* ===" ; }

step pytest
timeout -k 10 ${PYTEST_LIMIT:-900} python -u -m pytest tests -m gpu -v -s --durations=15 --timeout 300 --timeout-method thread ${PYTEST_ARGS:-} > gpurun_out/pytest.log 2>&1
rc=$?; grep -E "PASSED|FAILED|ERROR|passed|failed|error" gpurun_out/pytest.log | grep -v PASSED | tail -n 30; tail -n 22 gpurun_out/pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi

step smoke
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -n 5 gpurun_out/smoke.log; echo "smoke rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi

if [ -n "${SKIP_BENCH:-}" ]; then exit 0; fi
step bench
timeout -k 10 ${BENCH_LIMIT:-600} python -u bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err
rc=$?; tail -n 25 gpurun_out/bench.err; python3 tools/show_bench.py gpurun_out/bench.json; echo "bench rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 4 ]; then exit $rc; fi
exit 0
