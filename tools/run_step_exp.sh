set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_dist_rccl.py -x -q --timeout 300 --timeout-method thread -m gpu 2>&1 | tail -30
