set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_photon.py tests/test_gpu_photon_render.py tests/test_zz_no_alloc.py tests/test_gpu_config4_size.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pm_tests.log 2>&1
rc=$?; tail -n 6 gpurun_out/pm_tests.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for v in "RT_KNN_REDO_WALK=1" "RT_KNN_REDO_WALK=0"; do
  echo "== $v"; env $v timeout -k 10 300 python -u tools/exp_pframe.py full 2>/dev/null | tail -n 1 | cut -c1-160 || exit $?
  env $v timeout -k 10 300 python -u tools/exp_knn.py 2>/dev/null | tail -n 1 | cut -c1-100 || exit $?
done
RT_KNN_TRACE=1 timeout -k 10 400 python -u tools/exp_pframe.py full 2> gpurun_out/knn_trace.txt | tail -n 1 | cut -c1-100; grep "kNN call: 1000000" gpurun_out/knn_trace.txt | sort -t, -k5 | tail -n 12
