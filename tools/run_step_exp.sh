set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_photon.py -x -q --timeout 200 --timeout-method thread -m gpu -k "redo or lattice" 2>&1 | tail -12
