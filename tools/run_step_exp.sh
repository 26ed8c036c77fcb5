set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_render.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/step_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/step_tests.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/prof_timeline.sh spawn
