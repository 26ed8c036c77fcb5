set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "RT_PATH_BATCH=16" "RT_PATH_BATCH=8" "RT_PATH_BATCH=24" "RT_PATH_BATCH=32" "RT_PATH_BATCH=12" "RT_LANE_ACC_BATCH=2" "RT_LANE_ACC_BATCH=8"; do
  echo "== $v"; env $v timeout -k 10 120 python -u tools/exp_drt.py 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('ms_per_frame','path_ms','lane_closest_ms','lane_any_ms')})" || exit $?
done
