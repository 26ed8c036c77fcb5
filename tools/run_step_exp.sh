set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_render.py -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -3 || exit $?
for share in 1 8 1; do
  echo "== share $share"; RT_TRACE_PASSES=0 timeout -k 10 120 python -u tools/exp_drt.py 4 $share | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('ms_per_frame','mrays_per_s','lane_closest_ms','lane_any_ms','closest_ms','any_ms','path_ms','path_roots')})" || exit $?
done
bash tools/prof_timeline.sh spawn 1 300
