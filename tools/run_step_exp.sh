set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "RT_PATH_DEPTH=3 RT_LANE_DEPTH=1" "RT_PATH_DEPTH=2 RT_LANE_DEPTH=1" "RT_PATH_DEPTH=4" "RT_PATH_DEPTH=3 RT_PATH_WAVES=6144" "RT_PATH_DEPTH=3 RT_LANE_WAVES=2048" "RT_PATH_DEPTH=3 RT_LANE_WAVES=4096"; do
  echo "== $v"; env RT_PATH_MAX_ROOTS=100000000 $v timeout -k 10 120 python -u tools/exp_drt.py 4 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('ms_per_frame','mrays_per_s','lane_closest_ms','lane_any_ms','closest_ms','any_ms','path_ms','path_roots')})" || exit $?
done
for v in "RT_PATH_DEPTH=2" "RT_PATH_DEPTH=3" "RT_PATH_DEPTH=1" "RT_PATH_DEPTH=2 RT_PATH_WAVES=2048" "RT_PATH_DEPTH=2 RT_LANE_STEP=0"; do
  echo "== share8 $v"; env RT_PATH_MAX_ROOTS=100000000 $v timeout -k 10 120 python -u tools/exp_drt.py 8 8 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('ms_per_frame','mrays_per_s','lane_closest_ms','lane_any_ms','closest_ms','any_ms','path_ms','path_roots')})" || exit $?
done
