set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
FUZZ_ONLY=0 timeout -k 10 300 python -u tests/fuzz_render.py 100 7 2>&1 | tail -25
FUZZ_ONLY=30 timeout -k 10 300 python -u tests/fuzz_render.py 100 7 2>&1 | grep -E "mode|oracle frame|case 30" | tail -12
