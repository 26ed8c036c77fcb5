set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_trace.py tests/test_gpu_render.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/step_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/step_tests.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for v in "RT_LANE_STEP=1" "RT_LANE_REFILL_BATCH=16" "RT_LANE_REFILL_BATCH=4" "RT_LANE_REFILL_BATCH=2 RT_LANE_ACC_BATCH=2" "RT_LANE_CHUNK=64" "RT_LANE_CHUNK=512" "RT_LANE_WAVES=6144" "RT_LANE_WAVES=8192" "RT_PATH_MAX_ROOTS=0"; do
  echo "== $v"; env $v timeout -k 10 120 python -u tools/exp_drt.py 4 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('ms_per_frame','mrays_per_s','lane_closest_ms','lane_any_ms','closest_ms','any_ms','path_ms')})" || exit $?
done
