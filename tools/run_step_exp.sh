set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "RT_PM_CHUNK_LOG2=21" "RT_PM_CHUNK_LOG2=22"; do
  echo "== $v"
  env $v timeout -k 10 300 python -u tools/exp_pframe.py full 2>/dev/null | tail -1 | cut -c1-330 || exit $?
done
