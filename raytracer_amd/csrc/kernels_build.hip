// kernels_build.hip -- the device half of rt_scene_create: the octant layouts of both hierarchies.
//
// The host builds each hierarchy ONCE, in one pre-order: the reference KD-trees in the reference's own order (layout 7:
// the right child first everywhere, hw4.cpp:831-836) and the build's BVH in layout 0 (left child first).  Layout o of
// either is the same tree with, at a node split on axis a, the child on the side rays of octant o come from first
// (rt_internal.h: DevSceneView::nodes_oct).  A node's place in layout o follows from its path from the root alone: the
// sizes of the sibling subtrees it passes are differences of skip pointers, so one thread per node walks that path
// once and accumulates all seven places -- no host pass per layout, and 1/8 of the upload (10 M triangles: 0.7 GB
// instead of 5.6 GB).  The arrays are byte-identical to those the host emits (RT_HOST_LAYOUTS=1;
// tests/test_gpu_build.py compares them).  HBM-bound: 48 / 64 B read and 7 x 48 / 64 B written per node, the walk itself
// stays in L2.
#include <hip/hip_runtime.h>
#include "rt_internal.h"

namespace {

struct KdOut { DevNode* p[8]; };
struct BvhOut { DevBvhNode* p[8]; };

// roots[r] = index of the first node of tree r (ascending), roots[nroots] = n.  Source: layout 7.
__global__ __launch_bounds__(256) void k_kd_layouts(const DevNode* __restrict__ src, int32_t n, const int32_t* __restrict__ roots,
                                                    int32_t nroots, KdOut out) {
    const int32_t i = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n) return;
    int lo = 0, hi = nroots - 1;                       // the last root <= i
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (roots[mid] <= i) lo = mid; else hi = mid - 1; }
    int32_t a = roots[lo];
    int32_t pos[7];
#pragma unroll
    for (int o = 0; o < 7; ++o) pos[o] = a;
    while (a != i) {
        const int32_t end = src[a].skip, axis = (src[a].meta >> 12) & 3;
        const int32_t c1 = a + 1, c2 = src[c1].skip;            // right child, left child (a leaf's skip is its index + 1)
        const bool in_right = i < c2;
        const int32_t other = in_right ? end - c2 : c2 - c1;    // size of the sibling's subtree
#pragma unroll
        for (int o = 0; o < 7; ++o) {
            const bool right_first = (o >> axis) & 1;
            pos[o] += 1 + ((right_first != in_right) ? other : 0);
        }
        a = in_right ? c1 : c2;
    }
    DevNode nd = src[i];
    const int32_t size = nd.skip - i;
#pragma unroll
    for (int o = 0; o < 7; ++o) { nd.skip = pos[o] + size; out.p[o][pos[o]] = nd; }
}

// Source: layout 0.  A leaf's `skip` is its first triangle record (kept), its subtree is itself.
__global__ __launch_bounds__(256) void k_bvh_layouts(const DevBvhNode* __restrict__ src, int32_t n, BvhOut out) {
    const int32_t i = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n) return;
    int32_t a = 0;
    int32_t pos[8];
#pragma unroll
    for (int o = 1; o < 8; ++o) pos[o] = 0;
    while (a != i) {
        const int32_t end = src[a].skip, axis = (src[a].meta >> 12) & 3;
        const int32_t c1 = a + 1;                                // left child, then the right one
        const int32_t c2 = (src[c1].meta & 0x100) ? c1 + 1 : src[c1].skip;
        const bool in_left = i < c2;
        const int32_t other = in_left ? end - c2 : c2 - c1;
#pragma unroll
        for (int o = 1; o < 8; ++o) {
            const bool right_first = (o >> axis) & 1;
            pos[o] += 1 + ((right_first == in_left) ? other : 0);
        }
        a = in_left ? c1 : c2;
    }
    const DevBvhNode nd = src[i];
    const bool leaf = (nd.meta & 0x100) != 0;
    const int32_t size = leaf ? 1 : nd.skip - i;
#pragma unroll
    for (int o = 1; o < 8; ++o) {
        DevBvhNode w = nd;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if ((o >> k) & 1) { w.b[2 * k] = -nd.b[2 * k + 1]; w.b[2 * k + 1] = -nd.b[2 * k]; }
        if (!leaf) w.skip = pos[o] + size;
        out.p[o][pos[o]] = w;
    }
}

// reachable_rank's first probe per BVH triangle in one 32-byte record (rt_internal.h: DevTriReach)
__global__ __launch_bounds__(256) void k_tri_reach(const int32_t* __restrict__ tri_ref, const DevRefLeaf* __restrict__ ref_leaf,
                                                   const float* __restrict__ chain_box, int64_t nt, DevTriReach* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nt) return;
    DevTriReach o;
    o.lo[0] = o.lo[1] = o.lo[2] = o.hi[0] = o.hi[1] = o.hi[2] = __builtin_nanf("");
    o.rank = 0x7fffffff; o.pad = 0;
    const int32_t first = tri_ref[2 * t], cnt = tri_ref[2 * t + 1];
    if (cnt > 0) {
        const DevRefLeaf rl = ref_leaf[first];
        const float* cb = chain_box + 6 * (int64_t)rl.leaf;
#pragma unroll
        for (int k = 0; k < 3; ++k) { o.lo[k] = cb[k]; o.hi[k] = cb[3 + k]; }
        o.rank = rl.rank;
    }
    out[t] = o;
}

}  // namespace

hipError_t rtk_tri_reach(const int32_t* tri_ref, const DevRefLeaf* ref_leaf, const float* chain_box, int64_t nt, DevTriReach* out, hipStream_t st) {
    if (nt <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tri_reach, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, tri_ref, ref_leaf, chain_box, nt, out);
    return hipGetLastError();
}

hipError_t rtk_kd_layouts(const DevNode* src, int32_t n, const int32_t* d_roots, int32_t nroots, DevNode* const dst[7], hipStream_t st) {
    if (n <= 0) return hipSuccess;
    KdOut out{};
    for (int o = 0; o < 7; ++o) out.p[o] = dst[o];
    hipLaunchKernelGGL(k_kd_layouts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, n, d_roots, nroots, out);
    return hipGetLastError();
}

hipError_t rtk_bvh_layouts(const DevBvhNode* src, int32_t n, DevBvhNode* const dst[8], hipStream_t st) {
    if (n <= 0) return hipSuccess;
    BvhOut out{};
    for (int o = 1; o < 8; ++o) out.p[o] = dst[o];
    hipLaunchKernelGGL(k_bvh_layouts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, n, out);
    return hipGetLastError();
}
