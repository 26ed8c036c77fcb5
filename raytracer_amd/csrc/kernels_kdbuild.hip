// kernels_kdbuild.hip -- KDNode::BuildKD (HW4/hw4/hw4.cpp:393-490, with CalcBound :330-389) on the device, level by level.
//
// The reference builds one tree per object by recursion: bounds of the node's triangle list, the longest axis, the split
// value = element size/2 of the SORTED vertex coordinates on that axis (three per triangle), every triangle into the right
// list if a vertex is >= the split and into the left list if a vertex is < it (both: a straddler), and the node is
// subdivided iff the straddlers are fewer than left.size / modelScale and right.size / modelScale (float division);
// otherwise it is a leaf that keeps its own list.  Lists keep file order.  scene_build.cpp restates that on the host
// (threads; exact order statistic); this file does the same for ALL nodes of one tree level at once:
//   * a node is a segment of the level's triangle-reference array (A / B, ping-pong);
//   * bounds: per-reference triangle boxes, reduced per wave where the wave lies in one segment, then min / max atomics on
//     order-preserving integer keys (min and max do not depend on the order: the float the host finds);
//   * split value: the (segment, coordinate) pairs of the level as 64-bit keys through ONE radix sort (hipCUB), the
//     element at 3 off + (3 cnt) / 2 is the order statistic of the segment -- what std::sort's element size/2 is;
//   * lists: two flag arrays, two exclusive scans, a scatter that keeps the order (position = scan value - the segment's);
//   * the subdivision rule in float, with the correctly rounded division the library is built with.
// Output: the tree breadth-first (children of a node adjacent in the next level) and every leaf's list; scene_build.cpp
// turns that into the structure its flattener reads, so every array that goes to the kernels is the host build's, byte
// for byte (tests/test_gpu_build.py; RT_KD_BUILD=host|device, RT_CHECK_KD=1 builds both and compares).
// Not reproduced: the sign of a zero in a box when +0 and -0 both occur among a node's coordinates (the host keeps the
// first, the atomics the smaller key) -- no comparison the traversal makes can tell them apart.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "rt_internal.h"

namespace {

struct KdSeg {                 // one node of the level being built
    uint32_t mnk[3], mxk[3];   // bounds as ordered keys (kd_key) while they are reduced
    int32_t off, cnt;          // its references: [off, off + cnt) of the level's array
    float center[3], extend[3];
    int32_t axis;
    float split, model_scale;
    int32_t lcnt, rcnt, sub;   // sizes of the two lists; subdivided?
    int32_t child;             // index (within the next level) of its left child; the right one follows
    int32_t leaf_off;          // a leaf's list in leaf_ids
};

__device__ __forceinline__ uint32_t kd_key(float f) { const uint32_t u = (uint32_t)__float_as_int(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float kd_unkey(uint32_t k) { return __int_as_float((int)((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k)); }

__global__ __launch_bounds__(256) void k_kd_level_init(KdSeg* __restrict__ seg, int32_t m, int32_t* __restrict__ seg_off, int32_t nref) {
    const int32_t s = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (s > m) return;
    if (s == m) { seg_off[m] = nref; return; }
    seg_off[s] = seg[s].off;
    const uint32_t hi = kd_key(1e9f), lo = kd_key(-1e9f);          // CalcBound's start values (hw4.cpp:334-335)
#pragma unroll
    for (int k = 0; k < 3; ++k) { seg[s].mnk[k] = hi; seg[s].mxk[k] = lo; }
}

// the segment of every reference: the last segment whose offset is <= i
__global__ __launch_bounds__(256) void k_kd_seg_of(const int32_t* __restrict__ seg_off, int32_t m, int32_t nref, int32_t* __restrict__ segof) {
    const int32_t i = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (i >= nref) return;
    int lo = 0, hi = m - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid - 1; }
    segof[i] = lo;
}

__global__ __launch_bounds__(256) void k_kd_bounds(const rt_prim* __restrict__ prims, int32_t id0, const int32_t* __restrict__ ids,
                                                   const int32_t* __restrict__ segof, int32_t nref, KdSeg* __restrict__ seg) {
    const int32_t i = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    const bool on = i < nref;
    uint32_t mn[3], mx[3];
    int32_t s = -1;
    if (on) {
        s = segof[i];
        const float* f = prims[ids[i] - id0].f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t a = kd_key(f[k]), b = kd_key(f[3 + k]), c = kd_key(f[6 + k]);
            mn[k] = min(a, min(b, c)); mx[k] = max(a, max(b, c));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = 0xffffffffu; mx[k] = 0u; }
    }
    // a wave that lies in one segment (every wave of a large one) reduces first: one atomic per bound and wave
    const int32_t s0 = __shfl(s, 0, 64);
    const bool uniform = __ballot(on && s != s0) == 0ull && s0 >= 0;
    if (uniform) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { mn[k] = min(mn[k], (uint32_t)__shfl_xor((int)mn[k], o, 64)); mx[k] = max(mx[k], (uint32_t)__shfl_xor((int)mx[k], o, 64)); }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { atomicMin(&seg[s0].mnk[k], mn[k]); atomicMax(&seg[s0].mxk[k], mx[k]); }
        }
    } else if (on) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { atomicMin(&seg[s].mnk[k], mn[k]); atomicMax(&seg[s].mxk[k], mx[k]); }
    }
}

// CalcBound's last lines (hw4.cpp:380-389): centre, extent, longest axis (the first of equals)
__global__ __launch_bounds__(256) void k_kd_node_box(KdSeg* __restrict__ seg, int32_t m) {
    const int32_t s = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (s >= m) return;
    KdSeg& n = seg[s];
    float best = -1e9f; int axis = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float mn = kd_unkey(n.mnk[k]), mx = kd_unkey(n.mxk[k]);
        n.center[k] = (mn + mx) * 0.5f;             // operator/(vec3, 2.0f): times (float)(1.0 / 2.0)
        n.extend[k] = mx - mn;
        if (n.extend[k] > best) { best = n.extend[k]; axis = k; }
    }
    n.axis = axis;
}

// the three vertex coordinates of a reference on its node's axis, keyed by (segment, coordinate)
__global__ __launch_bounds__(256) void k_kd_keys(const rt_prim* __restrict__ prims, int32_t id0, const int32_t* __restrict__ ids,
                                                 const int32_t* __restrict__ segof, int32_t nref, const KdSeg* __restrict__ seg,
                                                 unsigned long long* __restrict__ keys) {
    const int32_t i = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (i >= nref) return;
    const int32_t s = segof[i];
    const int ax = seg[s].axis;
    const float* f = prims[ids[i] - id0].f;
    const unsigned long long hi = (unsigned long long)(uint32_t)s << 32;
    keys[3 * (size_t)i] = hi | kd_key(f[ax]);
    keys[3 * (size_t)i + 1] = hi | kd_key(f[3 + ax]);
    keys[3 * (size_t)i + 2] = hi | kd_key(f[6 + ax]);
}

// element size/2 of the node's sorted coordinates (hw4.cpp:405-419)
__global__ __launch_bounds__(256) void k_kd_split(KdSeg* __restrict__ seg, int32_t m, const unsigned long long* __restrict__ sorted) {
    const int32_t s = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (s >= m) return;
    const size_t at = 3 * (size_t)seg[s].off + (3 * (size_t)seg[s].cnt) / 2;
    seg[s].split = kd_unkey((uint32_t)(sorted[at] & 0xffffffffull));
}

// hw4.cpp:427-463: right if any vertex >= split, left if any vertex < split
__global__ __launch_bounds__(256) void k_kd_classify(const rt_prim* __restrict__ prims, int32_t id0, const int32_t* __restrict__ ids,
                                                     const int32_t* __restrict__ segof, int32_t nref, const KdSeg* __restrict__ seg,
                                                     int32_t* __restrict__ lf, int32_t* __restrict__ rf) {
    const int32_t i = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (i > nref) return;
    if (i == nref) { lf[i] = 0; rf[i] = 0; return; }
    const KdSeg& n = seg[segof[i]];
    const int ax = n.axis;
    const float split = n.split;
    const float* f = prims[ids[i] - id0].f;
    const float a = f[ax], b = f[3 + ax], c = f[6 + ax];
    rf[i] = (a >= split || b >= split || c >= split) ? 1 : 0;
    lf[i] = (a < split || b < split || c < split) ? 1 : 0;
}

// hw4.cpp:469-487: subdivide iff the straddlers are fewer than both lists divided by modelScale
__global__ __launch_bounds__(256) void k_kd_decide(KdSeg* __restrict__ seg, int32_t m, const int32_t* __restrict__ lpos, const int32_t* __restrict__ rpos,
                                                   int32_t* __restrict__ sz, int32_t* __restrict__ subflag, int32_t* __restrict__ leafsz) {
    const int32_t s = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (s > m) return;
    if (s == m) { sz[2 * m] = 0; subflag[m] = 0; leafsz[m] = 0; return; }
    KdSeg& n = seg[s];
    const int32_t lc = lpos[n.off + n.cnt] - lpos[n.off], rc = rpos[n.off + n.cnt] - rpos[n.off];
    const int32_t both = lc + rc - n.cnt;            // every reference is in one list at least: a coordinate is < split or >= it
    const bool sub = (float)both < (float)lc / n.model_scale && (float)both < (float)rc / n.model_scale;
    n.lcnt = lc; n.rcnt = rc; n.sub = sub ? 1 : 0;
    sz[2 * s] = sub ? lc : 0; sz[2 * s + 1] = sub ? rc : 0;
    subflag[s] = sub ? 1 : 0;
    leafsz[s] = sub ? 0 : n.cnt;
}

__global__ __launch_bounds__(256) void k_kd_scatter(const int32_t* __restrict__ ids, const int32_t* __restrict__ segof, int32_t nref,
                                                    const KdSeg* __restrict__ seg, const int32_t* __restrict__ lf, const int32_t* __restrict__ rf,
                                                    const int32_t* __restrict__ lpos, const int32_t* __restrict__ rpos,
                                                    const int32_t* __restrict__ noff, const int32_t* __restrict__ loff, int32_t leaf_base,
                                                    int32_t* __restrict__ next_ids, int32_t* __restrict__ leaf_ids) {
    const int32_t i = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (i >= nref) return;
    const int32_t s = segof[i];
    const KdSeg& n = seg[s];
    const int32_t id = ids[i];
    if (n.sub) {
        if (lf[i]) next_ids[noff[2 * s] + (lpos[i] - lpos[n.off])] = id;
        if (rf[i]) next_ids[noff[2 * s + 1] + (rpos[i] - rpos[n.off])] = id;
    } else {
        leaf_ids[(size_t)leaf_base + (size_t)loff[s] + (size_t)(i - n.off)] = id;
    }
}

// tot = {subdivided nodes, references of the next level, leaf references of this level}: read by the host BEFORE anything
// is written that they bound
__global__ void k_kd_totals(int32_t m, const int32_t* __restrict__ noff, const int32_t* __restrict__ cidx, const int32_t* __restrict__ loff, int32_t* __restrict__ tot) {
    tot[0] = cidx[m]; tot[1] = noff[2 * m]; tot[2] = loff[m];
}

// the next level's nodes
__global__ __launch_bounds__(256) void k_kd_next(KdSeg* __restrict__ seg, int32_t m, const int32_t* __restrict__ noff, const int32_t* __restrict__ cidx,
                                                 const int32_t* __restrict__ loff, int32_t leaf_base, KdSeg* __restrict__ next) {
    const int32_t s = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (s >= m) return;
    KdSeg& n = seg[s];
    if (n.sub) {
        const int32_t c = 2 * cidx[s];
        n.child = c; n.leaf_off = -1;
        KdSeg l{}, r{};
        l.off = noff[2 * s]; l.cnt = n.lcnt; l.model_scale = 2.0f;       // KDNode's constructor default (hw4.h:203)
        r.off = noff[2 * s + 1]; r.cnt = n.rcnt; r.model_scale = 2.0f;
        next[c] = l; next[c + 1] = r;
    } else {
        n.child = -1; n.leaf_off = leaf_base + loff[s];
    }
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T* as() const { return (T*)p; }
};

inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// Builds the tree of the triangles ids[0 .. nt) (file order; all within [id0, id0 + nprims)) of `prims` (host pointers).
// Returns 0 and the tree breadth-first -- nodes[i].child = index of the left child in `nodes` (the right one follows), or
// -1 and leaf_off / cnt = the leaf's list in leaf_ids -- or 1 when the device build does not apply (a capacity of this
// implementation is exceeded: the caller builds on the host), or a negative RT_E_ code with *err set.
int rtk_kd_build_device(int device, const rt_prim* prims, int32_t id0, int64_t nprims, const int32_t* ids, int64_t nt, float model_scale,
                        std::vector<RtKdBfsNode>* nodes, std::vector<int32_t>* leaf_ids, int32_t* max_depth, std::string* err) {
    nodes->clear(); leaf_ids->clear(); *max_depth = 0;
    if (nt <= 0) return 1;
    if (nt > (int64_t)100 << 20) return 1;                          // 32-bit offsets of 3 keys per reference
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed"; return RT_E_NODEVICE; }
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    const bool timing = getenv("RT_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    const int64_t cap_refs = 4 * nt + 4096, cap_leaf = 8 * nt + 4096, cap_nodes = 2 * nt + 4096;
    DevBuf d_prims, A, B, segof, keys0, keys1, lf, rf, lpos, rpos, leafb, segs, seg_off, sz, noff, subflag, cidx, leafsz, loff, tot, temp;
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)(3 * cap_refs), 0, 64, (hipStream_t)nullptr);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int)(cap_refs + 1), (hipStream_t)nullptr);
    const size_t temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    hipError_t e = hipSuccess;
    auto need = [&e](DevBuf& b, size_t bytes) { if (e == hipSuccess) e = b.alloc(bytes); };
    need(d_prims, (size_t)nprims * sizeof(rt_prim)); need(A, (size_t)cap_refs * 4); need(B, (size_t)cap_refs * 4); need(segof, (size_t)cap_refs * 4);
    need(keys0, (size_t)cap_refs * 24); need(keys1, (size_t)cap_refs * 24);
    need(lf, (size_t)(cap_refs + 1) * 4); need(rf, (size_t)(cap_refs + 1) * 4); need(lpos, (size_t)(cap_refs + 1) * 4); need(rpos, (size_t)(cap_refs + 1) * 4);
    need(leafb, (size_t)cap_leaf * 4); need(segs, (size_t)cap_nodes * sizeof(KdSeg));
    need(seg_off, (size_t)(cap_nodes + 1) * 4); need(sz, (size_t)(2 * cap_nodes + 1) * 4); need(noff, (size_t)(2 * cap_nodes + 1) * 4);
    need(subflag, (size_t)(cap_nodes + 1) * 4); need(cidx, (size_t)(cap_nodes + 1) * 4); need(leafsz, (size_t)(cap_nodes + 1) * 4); need(loff, (size_t)(cap_nodes + 1) * 4);
    need(tot, 16); need(temp, temp_bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return 1; }      // not enough device memory for the build's workspace: the host builds
    const double t1 = now();
    auto fail = [err](hipError_t x, const char* what) { *err = std::string(what) + ": " + hipGetErrorString(x); return RT_E_NODEVICE; };
    if ((e = hipMemcpy(d_prims.p, prims, (size_t)nprims * sizeof(rt_prim), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "KD build: upload");
    if ((e = hipMemcpy(A.p, ids, (size_t)nt * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "KD build: upload");
    KdSeg root{};
    root.off = 0; root.cnt = (int32_t)nt; root.model_scale = model_scale;
    if ((e = hipMemcpy(segs.p, &root, sizeof root, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "KD build: upload");

    const double t2 = now();
    hipStream_t st = nullptr;
    const rt_prim* P = d_prims.as<rt_prim>();
    int64_t base = 0, leaf_total = 0;
    int32_t m = 1, nref = (int32_t)nt;
    int32_t *cur = A.as<int32_t>(), *nxt = B.as<int32_t>();
    std::vector<int32_t> level_first;                                // first node of every level in segs
    int level = 0;
    while (m > 0) {
        if (level > 250) { *err = "KD-tree deeper than 250 levels"; return RT_E_LIMIT; }
        level_first.push_back((int32_t)base);
        KdSeg* S = segs.as<KdSeg>() + base;
        hipLaunchKernelGGL(k_kd_level_init, dim3(blocks(m + 1)), dim3(256), 0, st, S, m, seg_off.as<int32_t>(), nref);
        hipLaunchKernelGGL(k_kd_seg_of, dim3(blocks(nref)), dim3(256), 0, st, seg_off.as<int32_t>(), m, nref, segof.as<int32_t>());
        hipLaunchKernelGGL(k_kd_bounds, dim3(blocks(nref)), dim3(256), 0, st, P, id0, cur, segof.as<int32_t>(), nref, S);
        hipLaunchKernelGGL(k_kd_node_box, dim3(blocks(m)), dim3(256), 0, st, S, m);
        hipLaunchKernelGGL(k_kd_keys, dim3(blocks(nref)), dim3(256), 0, st, P, id0, cur, segof.as<int32_t>(), nref, S, keys0.as<unsigned long long>());
        int seg_bits = 1;
        while (((int64_t)1 << seg_bits) < m) ++seg_bits;
        size_t tb = temp_bytes;
        if ((e = hipcub::DeviceRadixSort::SortKeys(temp.p, tb, keys0.as<unsigned long long>(), keys1.as<unsigned long long>(), (int)(3 * (int64_t)nref), 0, 32 + seg_bits, st)) != hipSuccess)
            return fail(e, "KD build: sort");
        hipLaunchKernelGGL(k_kd_split, dim3(blocks(m)), dim3(256), 0, st, S, m, keys1.as<unsigned long long>());
        hipLaunchKernelGGL(k_kd_classify, dim3(blocks((int64_t)nref + 1)), dim3(256), 0, st, P, id0, cur, segof.as<int32_t>(), nref, S, lf.as<int32_t>(), rf.as<int32_t>());
        tb = temp_bytes;
        if ((e = hipcub::DeviceScan::ExclusiveSum(temp.p, tb, lf.as<int32_t>(), lpos.as<int32_t>(), nref + 1, st)) != hipSuccess) return fail(e, "KD build: scan");
        tb = temp_bytes;
        if ((e = hipcub::DeviceScan::ExclusiveSum(temp.p, tb, rf.as<int32_t>(), rpos.as<int32_t>(), nref + 1, st)) != hipSuccess) return fail(e, "KD build: scan");
        hipLaunchKernelGGL(k_kd_decide, dim3(blocks(m + 1)), dim3(256), 0, st, S, m, lpos.as<int32_t>(), rpos.as<int32_t>(), sz.as<int32_t>(), subflag.as<int32_t>(), leafsz.as<int32_t>());
        tb = temp_bytes;
        if ((e = hipcub::DeviceScan::ExclusiveSum(temp.p, tb, sz.as<int32_t>(), noff.as<int32_t>(), 2 * m + 1, st)) != hipSuccess) return fail(e, "KD build: scan");
        tb = temp_bytes;
        if ((e = hipcub::DeviceScan::ExclusiveSum(temp.p, tb, subflag.as<int32_t>(), cidx.as<int32_t>(), m + 1, st)) != hipSuccess) return fail(e, "KD build: scan");
        tb = temp_bytes;
        if ((e = hipcub::DeviceScan::ExclusiveSum(temp.p, tb, leafsz.as<int32_t>(), loff.as<int32_t>(), m + 1, st)) != hipSuccess) return fail(e, "KD build: scan");
        hipLaunchKernelGGL(k_kd_totals, dim3(1), dim3(1), 0, st, m, noff.as<int32_t>(), cidx.as<int32_t>(), loff.as<int32_t>(), tot.as<int32_t>());
        int32_t h[3] = {0, 0, 0};
        if ((e = hipMemcpy(h, tot.p, sizeof h, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "KD build: totals");
        const int64_t m_next = 2 * (int64_t)h[0];
        if (h[1] > cap_refs || leaf_total + h[2] > cap_leaf || base + m + m_next > cap_nodes) return 1;      // more duplication than the workspace holds: the host builds
        hipLaunchKernelGGL(k_kd_next, dim3(blocks(m)), dim3(256), 0, st, S, m, noff.as<int32_t>(), cidx.as<int32_t>(), loff.as<int32_t>(), (int32_t)leaf_total, S + m);
        hipLaunchKernelGGL(k_kd_scatter, dim3(blocks(nref)), dim3(256), 0, st, cur, segof.as<int32_t>(), nref, S, lf.as<int32_t>(), rf.as<int32_t>(), lpos.as<int32_t>(),
                           rpos.as<int32_t>(), noff.as<int32_t>(), loff.as<int32_t>(), (int32_t)leaf_total, nxt, leafb.as<int32_t>());
        leaf_total += h[2];
        base += m; m = (int32_t)m_next; nref = h[1];
        int32_t* t = cur; cur = nxt; nxt = t;
        ++level;
    }
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "KD build");
    *max_depth = level - 1;
    const double t3 = now();
    std::vector<KdSeg> hs((size_t)base);
    if ((e = hipMemcpy(hs.data(), segs.p, (size_t)base * sizeof(KdSeg), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "KD build: download");
    leaf_ids->resize((size_t)leaf_total);
    if (leaf_total && (e = hipMemcpy(leaf_ids->data(), leafb.p, (size_t)leaf_total * 4, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "KD build: download");
    nodes->resize((size_t)base);
    level_first.push_back((int32_t)base);
    for (size_t lv = 0; lv + 1 < level_first.size(); ++lv) {
        const int32_t a = level_first[lv], b = level_first[lv + 1];
        for (int32_t i = a; i < b; ++i) {
            const KdSeg& s = hs[(size_t)i];
            RtKdBfsNode& o = (*nodes)[(size_t)i];
            for (int k = 0; k < 3; ++k) { o.center[k] = s.center[k]; o.extend[k] = s.extend[k]; }
            o.axis = s.axis; o.cnt = s.cnt;
            o.child = s.sub ? b + s.child : -1;
            o.leaf_off = s.sub ? -1 : s.leaf_off;
        }
    }
    if (timing) fprintf(stderr, "KD build on the device: workspace %.3f s, upload %.3f s, %d levels %.3f s, download + nodes %.3f s\n", t1 - t0, t2 - t1, level, t3 - t2, now() - t3);
    return 0;
}
