for v in "RT_BVH_LEAF=6" "RT_BVH_LEAF=7" "RT_BVH_LEAF=8" "RT_BVH_LEAF=12"; do
  echo "== $v"; env $v timeout -k 10 120 python tools/exp_primary.py 10 2>&1 | grep -v amdgpu.ids || exit 1
done
