set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "RT_KNN_REDO_WALK=1" "RT_KNN_REDO_WALK=2" "RT_KNN_REDO_WALK=1" "RT_KNN_REDO_WALK=2"; do
  echo "== $v 10M / 200M / frame"
  env $v timeout -k 10 200 python -u tools/exp_knn.py 2>/dev/null | tail -1 | cut -c1-100 || exit $?
  env $v timeout -k 10 300 python -u tools/exp_knn.py 2e8 2>/dev/null | tail -1 | cut -c1-100 || exit $?
  env $v timeout -k 10 300 python -u tools/exp_pframe.py full 2>/dev/null | tail -1 | cut -c1-130 || exit $?
done
