set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_photon.py tests/test_gpu_photon_render.py tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread -m gpu 2>&1 | tail -4; echo "pytest rc=$?"
for v in "RT_KNN_SPLIT=1" "RT_KNN_SPLIT=0"; do
  echo "== $v"
  env $v timeout -k 10 200 python -u tools/exp_knn.py 2>/dev/null | tail -1 | cut -c1-400 || exit $?
  env $v timeout -k 10 200 python -u tools/exp_pframe.py 2>/dev/null | tail -1 || exit $?
  env $v timeout -k 10 300 python -u tools/exp_pframe.py full 2>/dev/null | tail -1 || exit $?
done
