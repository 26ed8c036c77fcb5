set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_photon.py tests/test_gpu_photon_render.py tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread -m gpu 2>&1 | tail -3; echo "pytest rc=$?"
timeout -k 10 900 python -u tools/exp_redo.py 1e6 20000 2>&1 | grep -v "^kNN\|amdgpu" | tail -10 | cut -c1-250
for v in "RT_KNN_REDO_WALK=1" "RT_KNN_REDO_WALK=2" "RT_KNN_REDO_SWITCH=10000"; do
  echo "== $v"
  env $v timeout -k 10 200 python -u tools/exp_knn.py 2>/dev/null | tail -1 | cut -c1-100 || exit $?
  env $v timeout -k 10 300 python -u tools/exp_pframe.py full 2>/dev/null | tail -1 | cut -c1-200 || exit $?
done
