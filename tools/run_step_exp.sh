set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "RT_PATH_WAVES=3072" "RT_PATH_WAVES=4096"; do
  for share in 1 8; do
  echo "== $v share $share"; env $v timeout -k 10 120 python -u tools/exp_drt.py 4 $share | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('ms_per_frame','mrays_per_s','lane_closest_ms','lane_any_ms','closest_ms','any_ms','path_ms','path_roots')})" || exit $?
  done
done
