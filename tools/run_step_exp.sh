set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "RT_PRIMARY_XCD_BANDS=0" "RT_PRIMARY_XCD_BANDS=4" "RT_PRIMARY_XCD_BANDS=15" "RT_PRIMARY_XCD_BANDS=60" "RT_PRIMARY_XCD_BANDS=240"; do
  echo "== $v"; env $v timeout -k 10 200 python -u tools/exp_primary.py 10 2>/dev/null | cut -c1-90 || exit $?
done
