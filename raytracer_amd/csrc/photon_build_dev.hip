// photon_build_dev.hip -- the photon-map build ON THE DEVICE: PhotonMap::balance / balance_segment /
// median_split (Final/photon/photon.cpp:133-254) replayed exactly, and the device image of a map (photon_api.cpp:
// upload_heap) assembled from it without a trip through the host.
//
// Why a replay and not a sort: median_split is a Hoare quickselect, and WHICH of several photons with the same coordinate
// on the split axis (photons on a wall, on the caustic floor) ends up at the median depends on the order the segment is
// in, i.e. on every swap made before.  Heap slots are compared index for index with the reference (kNN index sets), so the
// swaps are made as the reference makes them.
//
// What is parallel:
//  * segments of one level are independent (photon_build.cpp runs them on threads): one workgroup per segment, level by
//    level, while a segment is larger than PB_SUB; a segment of <= PB_SUB photons is loaded into LDS once and its whole
//    subtree finished there, one lane per sub-segment (k_pb_small);
//  * inside a large segment, one partition pass pairs the k-th element from the left that is not < pivot with the k-th from
//    the right that is not > pivot, for k = 1, 2, ... until the two scans meet.  While the two scans are in DISJOINT chunks
//    the pairs of a chunk against a chunk are known without looking at anything else, and are swapped by 256 lanes at once
//    (the order of the pairs is the serial one: both lists are consumed front to back).  Where the scans meet (<= 3074
//    elements) the zone is copied to LDS and finished by one more parallel step (pb_zone_pass: the same pairing over the
//    whole zone, plus the rule for where the left scan ends).  A range of <= PB_ZONE elements stays in LDS for all its
//    remaining passes.
#include "rt_internal.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int PB_T = 256;                  // lanes per workgroup
constexpr int PB_EBIG = 16;                // elements per lane in a big chunk
constexpr int PB_CBIG = PB_T * PB_EBIG;    // 4096
constexpr int PB_EMID = 4;
constexpr int PB_CMID = PB_T * PB_EMID;    // 1024
constexpr int PB_SUB = 1024;               // a segment this short: its whole subtree in LDS

struct PbSeg {                 // balance_segment's arguments, and how far the wide passes (below) have taken its median_split
    uint32_t index; int32_t start, end; float lo[3], hi[3];
    int32_t left, right;       // median_split's current range: start, end when nothing has run yet
    int32_t have_box;          // tight: lo / hi already hold the segment's own bounding box
};

struct PbCtx {
    float4* org;           // porg with the position beside the index: {x, y, z, bits of idx}, 1-based
    int32_t* bal;          // pbal: heap slot -> source index
    int8_t* plane;         // by source index, -1 = never a median of a segment
    int tight;             // second heap: split on the axis of the segment's own largest extent
    PbSeg* next; uint32_t* nnext; uint32_t cap_next;
    PbSeg* small; uint32_t* nsmall; uint32_t cap_small;
    uint32_t* overflow;
    uint32_t* dbg;         // RT_PB_DEBUG: [0] most passes of a segment, [1] most serial steps, [2] most chunk rounds (per launch)
};

__device__ __forceinline__ float pb_comp(const float4& r, int axis) { return axis == 0 ? r.x : (axis == 1 ? r.y : r.z); }
__device__ __forceinline__ float pb_key(const float4* org, int p, int axis) { return reinterpret_cast<const float*>(org + p)[axis]; }

// balance_segment's median index, photon.cpp:173-183
__device__ __forceinline__ int pb_median(int start, int end) {
    const long long count = (long long)end - start + 1;
    long long median = 1;
    while (4 * median <= count) median += median;
    if (3 * median <= count) { median += median; median += start - 1; }
    else median = end - median + 1;
    return (int)median;
}
__device__ __forceinline__ int pb_axis(const float* lo, const float* hi) {          // photon.cpp:187-193
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    int axis = 2;
    if (ex > ey && ex > ez) axis = 0;
    else if (ey > ez) axis = 1;
    return axis;
}

// median_split (photon.cpp:224-254), literally, on a window: w[p - base] is org[p]
__device__ void pb_split_window(float4* w, int base, int start, int end, int median, int axis) {
    int left = start, right = end;
    while (right > left) {
        const float v = pb_comp(w[right - base], axis);
        int i = left - 1, j = right;
        for (;;) {
            while (pb_comp(w[++i - base], axis) < v) {}
            while (pb_comp(w[--j - base], axis) > v && j > left) {}
            if (i >= j) break;
            const float4 t = w[i - base]; w[i - base] = w[j - base]; w[j - base] = t;
        }
        const float4 t = w[i - base]; w[i - base] = w[right - base]; w[right - base] = t;
        if (i >= median) right = i - 1;
        if (i <= median) left = i + 1;
    }
}

// One chunk of one scan: the positions base, base +- 1, ... (E per lane, lane-blocked) whose key stops the scan -- the left
// scan stops at !(key < v), the right one at !(key > v) -- written to `list` in scan order.  Returns how many.
template <int E, bool LEFT>
__device__ int pb_scan_chunk(const float4* org, int base, int axis, float v, int* list, int* s_wsum) {
    const int t = threadIdx.x;
    unsigned flags = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int p = LEFT ? base + (t * E + e) : base - (t * E + e);
        const float key = pb_key(org, p, axis);
        const bool stop = LEFT ? !(key < v) : !(key > v);
        flags |= (stop ? 1u : 0u) << e;
    }
    const int c = __popc(flags);
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if ((t & 63) >= d) incl += o;
    }
    if ((t & 63) == 63) s_wsum[t >> 6] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < PB_T / 64; ++k) { const int s = s_wsum[k]; if (k < (t >> 6)) woff += s; total += s; }
    int off = woff + incl - c;
#pragma unroll
    for (int e = 0; e < E; ++e)
        if ((flags >> e) & 1u) list[off++] = LEFT ? base + (t * E + e) : base - (t * E + e);
    __syncthreads();
    return total;
}

__device__ void pb_push(PbSeg* q, uint32_t* n, uint32_t cap, uint32_t* overflow, const PbSeg& s) {
    const uint32_t k = atomicAdd(n, 1u);
    if (k < cap) q[k] = s; else atomicExch(overflow, 1u);
}

// after the split: the node, and what balance_segment does with the two halves (photon.cpp:197-221)
__device__ void pb_finish_segment(const PbCtx& g, uint32_t index, int start, int end, int median, int axis, const float* lo, const float* hi) {
    const float4 node = g.org[median];
    const int nidx = __float_as_int(node.w);
    g.bal[index] = nidx;
    g.plane[nidx] = (int8_t)axis;
    const float cut = pb_comp(node, axis);
    if (median > start) {
        if (start < median - 1) {
            PbSeg s; s.index = 2u * index; s.start = s.left = start; s.end = s.right = median - 1; s.have_box = 0;
            for (int k = 0; k < 3; ++k) { s.lo[k] = lo[k]; s.hi[k] = hi[k]; }
            s.hi[axis] = cut;
            if (s.end - s.start + 1 > PB_SUB) pb_push(g.next, g.nnext, g.cap_next, g.overflow, s);
            else pb_push(g.small, g.nsmall, g.cap_small, g.overflow, s);
        } else g.bal[2u * index] = __float_as_int(g.org[start].w);
    }
    if (median < end) {
        if (median + 1 < end) {
            PbSeg s; s.index = 2u * index + 1u; s.start = s.left = median + 1; s.end = s.right = end; s.have_box = 0;
            for (int k = 0; k < 3; ++k) { s.lo[k] = lo[k]; s.hi[k] = hi[k]; }
            s.lo[axis] = cut;
            if (s.end - s.start + 1 > PB_SUB) pb_push(g.next, g.nnext, g.cap_next, g.overflow, s);
            else pb_push(g.small, g.nsmall, g.cap_small, g.overflow, s);
        } else g.bal[2u * index + 1u] = __float_as_int(g.org[end].w);
    }
}

// One whole partition pass of median_split over positions (i0, j0) of an LDS window (win[p - wbase] is org[p]), by all lanes.
// State on entry, as the reference's loop would have it: i stands at i0 (everything up to i0 is behind the left scan), j
// at j0, and position j0 is known to stop the left scan (the pivot itself, a swapped-in element, or an element > v).
//   A = the positions in (i0, j0) that stop the left scan, ascending, then j0;  B = those that stop the right scan
//   (`left` always does, photon.cpp:240), descending.
// The reference swaps A[k] with B[k] for k = 0, 1, ... while A[k] < B[k]: between two swaps its scans only cross
// positions no swap has touched yet -- the swapped ones are behind the scan that made them and ahead of the other.  At the
// first k with A[k] >= B[k] (or B exhausted: the right scan runs into a swapped-in element or i0) the left scan stops at
// A[k], or earlier at B[k-1], which now holds an element from the left: i = min(A[k], B[k-1]).  Returns i.
constexpr int PB_ZONE = 4096;
constexpr int PB_ZROWS = PB_ZONE / PB_T;
struct PbZoneShared { int cnt[2][PB_ZROWS * (PB_T / 64)]; int pre[2][PB_ZROWS * (PB_T / 64)]; int tot[2]; int npairs; };
__device__ int pb_zone_pass(float4* win, int wbase, int i0, int j0, int left, int axis, float v, int* listA, int* listB, PbZoneShared& zs) {
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int zlo = i0 + 1, zhi = j0 - 1;
    const int rows = zhi >= zlo ? (zhi - zlo + PB_T) / PB_T : 0;
    constexpr int NW = PB_T / 64;
    unsigned fa = 0, fb = 0;
    for (int r = 0; r < rows; ++r) {
        const int p = zlo + r * PB_T + t;
        bool a = false, b = false;
        if (p <= zhi) {
            const float key = pb_comp(win[p - wbase], axis);
            a = !(key < v);
            b = !(key > v) || p == left;
        }
        fa |= (a ? 1u : 0u) << r; fb |= (b ? 1u : 0u) << r;
        const unsigned long long ba = __ballot(a), bb = __ballot(b);
        if (lane == 0) { zs.cnt[0][r * NW + w] = __popcll(ba); zs.cnt[1][r * NW + w] = __popcll(bb); }
    }
    if (t == 0) zs.npairs = 0;
    __syncthreads();
    if (w < 2) {                                        // exclusive prefix of the (row, wave) counts: wave 0 for A, wave 1 for B
        const int x = lane < rows * NW ? zs.cnt[w][lane] : 0;
        int incl = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        zs.pre[w][lane] = incl - x;
        if (lane == 63) zs.tot[w] = incl;
    }
    __syncthreads();
    const int nA = zs.tot[0] + 1, nB = zs.tot[1];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int r = 0; r < rows; ++r) {
        const bool a = (fa >> r) & 1u, b = (fb >> r) & 1u;
        const unsigned long long ba = __ballot(a), bb = __ballot(b);
        const int p = zlo + r * PB_T + t;
        if (a) listA[zs.pre[0][r * NW + w] + __popcll(ba & below)] = p;
        if (b) listB[nB - 1 - (zs.pre[1][r * NW + w] + __popcll(bb & below))] = p;
    }
    if (t == 0) listA[nA - 1] = j0;
    __syncthreads();
    const int nmin = min(nA, nB);
    int mine = 0;
    for (int k = t; k < nmin; k += PB_T) mine += listA[k] < listB[k] ? 1 : 0;     // a prefix: A ascends, B descends
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane == 0 && mine) atomicAdd(&zs.npairs, mine);
    __syncthreads();
    const int np = zs.npairs;
    for (int k = t; k < np; k += PB_T) {
        const int pa = listA[k] - wbase, pb = listB[k] - wbase;
        const float4 x = win[pa]; win[pa] = win[pb]; win[pb] = x;
    }
    const int ia = listA[np];                          // np < nA: the terminator j0 is never below an element of B
    const int i = np >= 1 ? min(ia, listB[np - 1]) : ia;
    __syncthreads();
    return i;
}

// ---- one workgroup per segment of more than PB_SUB photons ---------------------------------------------------------------
__global__ __launch_bounds__(PB_T) void k_pb_big(PbCtx g, const PbSeg* __restrict__ segs, uint32_t nsegs) {
    __shared__ float4 win[PB_ZONE + 8];
    __shared__ int listL[PB_CBIG + 8], listR[PB_CBIG + 8];
    __shared__ int s_wsum[PB_T / 64];
    __shared__ float s_red[6][PB_T / 64];
    __shared__ PbZoneShared zs;
    const int t = threadIdx.x;
    if (blockIdx.x >= nsegs) return;
    const PbSeg sg = segs[blockIdx.x];
    const int start = sg.start, end = sg.end;
    const int median = pb_median(start, end);
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = sg.lo[k]; hi[k] = sg.hi[k]; }
    if (g.tight && !sg.have_box) {       // the segment's own bounding box (photon_build.cpp: Balancer::segment, tight)
        float l[3] = {1e30f, 1e30f, 1e30f}, h[3] = {-1e30f, -1e30f, -1e30f};
        for (int p = start + t; p <= end; p += PB_T) {
            const float4 r = g.org[p];
            const float c[3] = {r.x, r.y, r.z};
            for (int k = 0; k < 3; ++k) { if (c[k] < l[k]) l[k] = c[k]; if (c[k] > h[k]) h[k] = c[k]; }
        }
        for (int k = 0; k < 3; ++k) {
            for (int d = 32; d >= 1; d >>= 1) {
                const float ol = __shfl_xor(l[k], d, 64), oh = __shfl_xor(h[k], d, 64);
                if (ol < l[k]) l[k] = ol;
                if (oh > h[k]) h[k] = oh;
            }
            if ((t & 63) == 0) { s_red[k][t >> 6] = l[k]; s_red[3 + k][t >> 6] = h[k]; }
        }
        __syncthreads();
        for (int k = 0; k < 3; ++k) {
            float a = s_red[k][0], b = s_red[3 + k][0];
            for (int q = 1; q < PB_T / 64; ++q) { if (s_red[k][q] < a) a = s_red[k][q]; if (s_red[3 + k][q] > b) b = s_red[3 + k][q]; }
            lo[k] = a; hi[k] = b;
        }
        __syncthreads();
    }
    const int axis = pb_axis(lo, hi);

    int left = sg.left, right = sg.right;                // median_split (from where the wide passes left it); uniform
    uint32_t d_pass = 0, d_rounds = 0;
    while (right > left) {
        const int len = right - left + 1;
        if (len <= PB_ZONE) {                            // the range fits the window: every remaining pass in LDS
            const int wbase = left;
            for (int p = t; p < len; p += PB_T) win[p] = g.org[wbase + p];
            __syncthreads();
            int l = left, r = right;
            while (r > l) {
                ++d_pass;
                const float v = pb_comp(win[r - wbase], axis);
                const int i = pb_zone_pass(win, wbase, l - 1, r, l, axis, v, listL, listR, zs);
                if (t == 0) { const float4 x = win[i - wbase]; win[i - wbase] = win[r - wbase]; win[r - wbase] = x; }
                __syncthreads();
                if (i >= median) r = i - 1;
                if (i <= median) l = i + 1;
            }
            for (int p = t; p < len; p += PB_T) g.org[wbase + p] = win[p];
            __syncthreads();
            break;
        }
        ++d_pass;
        const float v = pb_key(g.org, right, axis);
        int lo_p = left, hi_p = right - 1;               // next unscanned position of the left / right scan
        int nL = 0, offL = 0, nR = 0, offR = 0;
        bool big = true;
        for (;;) {
            int gap = hi_p - lo_p + 1;
            if (big && gap < 2 * PB_CBIG) {              // near the meeting point: smaller chunks; what a big chunk left
                big = false;                             // pending is handed back to be scanned again
                if (nL > offL) lo_p = listL[offL];
                if (nR > offR) hi_p = listR[offR];
                nL = offL = nR = offR = 0;
                gap = hi_p - lo_p + 1;
                __syncthreads();
            }
            const int ch = big ? PB_CBIG : PB_CMID;
            if (gap < 2 * ch) break;
            if (nL == offL) {
                nL = big ? pb_scan_chunk<PB_EBIG, true>(g.org, lo_p, axis, v, listL, s_wsum) : pb_scan_chunk<PB_EMID, true>(g.org, lo_p, axis, v, listL, s_wsum);
                offL = 0; lo_p += ch;
            }
            if (nR == offR) {
                nR = big ? pb_scan_chunk<PB_EBIG, false>(g.org, hi_p, axis, v, listR, s_wsum) : pb_scan_chunk<PB_EMID, false>(g.org, hi_p, axis, v, listR, s_wsum);
                offR = 0; hi_p -= ch;
            }
            ++d_rounds;
            const int m = min(nL - offL, nR - offR);
            for (int q = t; q < m; q += PB_T) {
                const int pa = listL[offL + q], pb = listR[offR + q];
                const float4 a = g.org[pa], b = g.org[pb];
                g.org[pa] = b; g.org[pb] = a;
            }
            offL += m; offR += m;
            __syncthreads();
        }
        // the meeting zone, in LDS: the reference's loop from where the two scans stand.  i: the next position it tests is
        // the first pending left stop (everything before it was scanned and is < v), else the first unscanned one; j alike
        // (position j0 holds an element that stops the left scan: swapped in, scanned as > v, or the pivot).
        const int i0 = (nL > offL ? listL[offL] : lo_p) - 1;
        const int j0 = (nR > offR ? listR[offR] : hi_p) + 1;
        const int wlo = max(i0, left), whi = j0;                       // <= gap + one mid chunk + 2 positions
        const int wn = whi - wlo + 1;
        __syncthreads();
        for (int p = t; p < wn; p += PB_T) win[p] = g.org[wlo + p];
        __syncthreads();
        const int i = pb_zone_pass(win, wlo, i0, j0, left, axis, v, listL, listR, zs);
        if (right == whi && t == 0) { const float4 x = win[i - wlo]; win[i - wlo] = win[right - wlo]; win[right - wlo] = x; }
        __syncthreads();
        for (int p = t; p < wn; p += PB_T) g.org[wlo + p] = win[p];
        __syncthreads();
        if (right > whi && t == 0) { const float4 a = g.org[i], b = g.org[right]; g.org[i] = b; g.org[right] = a; }
        __syncthreads();
        if (i >= median) right = i - 1;
        if (i <= median) left = i + 1;
    }
    if (t == 0) {
        pb_finish_segment(g, sg.index, start, end, median, axis, lo, hi);
        if (g.dbg) { atomicMax(g.dbg, d_pass); atomicMax(g.dbg + 2, d_rounds); }
    }
}

// ---- one workgroup per segment of <= PB_SUB photons: the whole subtree in LDS, one lane per sub-segment -----------------
struct PbSub { uint32_t index; uint16_t start, end; float lo[3], hi[3]; };      // start / end relative to the segment
constexpr int PB_LANE = 128;               // a sub-segment this short: one lane
// the node of a sub-segment after its split, and its halves (as pb_finish_segment, inside the LDS window)
__device__ void pb_small_children(const PbCtx& g, const float4* w, const PbSub& s, int start, int end, int median, int axis,
                                  const float* lo, const float* hi, PbSub* out, int* nout) {
    const float4 node = w[median];
    const int nidx = __float_as_int(node.w);
    g.bal[s.index] = nidx;
    g.plane[nidx] = (int8_t)axis;
    const float cut = pb_comp(node, axis);
    if (median > start) {
        if (start < median - 1) {
            PbSub c; c.index = 2u * s.index; c.start = (uint16_t)start; c.end = (uint16_t)(median - 1);
            for (int k = 0; k < 3; ++k) { c.lo[k] = lo[k]; c.hi[k] = hi[k]; }
            c.hi[axis] = cut;
            out[atomicAdd(nout, 1)] = c;
        } else g.bal[2u * s.index] = __float_as_int(w[start].w);
    }
    if (median < end) {
        if (median + 1 < end) {
            PbSub c; c.index = 2u * s.index + 1u; c.start = (uint16_t)(median + 1); c.end = (uint16_t)end;
            for (int k = 0; k < 3; ++k) { c.lo[k] = lo[k]; c.hi[k] = hi[k]; }
            c.lo[axis] = cut;
            out[atomicAdd(nout, 1)] = c;
        } else g.bal[2u * s.index + 1u] = __float_as_int(w[end].w);
    }
}
__global__ __launch_bounds__(PB_T) void k_pb_small(PbCtx g, const PbSeg* __restrict__ segs, uint32_t nsegs) {
    __shared__ float4 w[PB_SUB];
    __shared__ PbSub lists[2][PB_SUB / 2 + 1];
    __shared__ int s_n[2];
    __shared__ int listA[PB_SUB + 8], listB[PB_SUB + 8];
    __shared__ float s_red[6][PB_T / 64];
    __shared__ PbZoneShared zs;
    const int t = threadIdx.x;
    if (blockIdx.x >= nsegs) return;
    const PbSeg sg = segs[blockIdx.x];
    const int base = sg.start, n = sg.end - sg.start + 1;
    for (int p = t; p < n; p += PB_T) w[p] = g.org[base + p];
    if (t == 0) {
        PbSub s; s.index = sg.index; s.start = 0; s.end = (uint16_t)(n - 1);
        for (int k = 0; k < 3; ++k) { s.lo[k] = sg.lo[k]; s.hi[k] = sg.hi[k]; }
        lists[0][0] = s; s_n[0] = 1; s_n[1] = 0;
    }
    __syncthreads();
    int cur = 0;
    for (;;) {
        const int ns = s_n[cur];
        if (ns == 0) break;
        // sub-segments of more than PB_LANE photons (the top levels of the subtree, a handful): one after the other, every
        // pass by the whole workgroup (pb_zone_pass); the rest: one lane each, the reference's loop as it stands
        for (int q = 0; q < ns; ++q) {
            const PbSub s = lists[cur][q];
            const int start = s.start, end = s.end;
            if (end - start + 1 <= PB_LANE) continue;
            const int median = pb_median(start, end);
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k) { lo[k] = s.lo[k]; hi[k] = s.hi[k]; }
            if (g.tight) {
                float l[3] = {1e30f, 1e30f, 1e30f}, h[3] = {-1e30f, -1e30f, -1e30f};
                for (int p = start + t; p <= end; p += PB_T) {
                    const float4 r = w[p];
                    const float c[3] = {r.x, r.y, r.z};
                    for (int k = 0; k < 3; ++k) { if (c[k] < l[k]) l[k] = c[k]; if (c[k] > h[k]) h[k] = c[k]; }
                }
                for (int k = 0; k < 3; ++k) {
                    for (int d = 32; d >= 1; d >>= 1) {
                        const float ol = __shfl_xor(l[k], d, 64), oh = __shfl_xor(h[k], d, 64);
                        if (ol < l[k]) l[k] = ol;
                        if (oh > h[k]) h[k] = oh;
                    }
                    if ((t & 63) == 0) { s_red[k][t >> 6] = l[k]; s_red[3 + k][t >> 6] = h[k]; }
                }
                __syncthreads();
                for (int k = 0; k < 3; ++k) {
                    float a = s_red[k][0], b = s_red[3 + k][0];
                    for (int z = 1; z < PB_T / 64; ++z) { if (s_red[k][z] < a) a = s_red[k][z]; if (s_red[3 + k][z] > b) b = s_red[3 + k][z]; }
                    lo[k] = a; hi[k] = b;
                }
                __syncthreads();
            }
            const int axis = pb_axis(lo, hi);
            int l = start, r = end;
            while (r > l) {
                const float v = pb_comp(w[r], axis);
                const int i = pb_zone_pass(w, 0, l - 1, r, l, axis, v, listA, listB, zs);
                if (t == 0) { const float4 x = w[i]; w[i] = w[r]; w[r] = x; }
                __syncthreads();
                if (i >= median) r = i - 1;
                if (i <= median) l = i + 1;
            }
            if (t == 0) pb_small_children(g, w, s, start, end, median, axis, lo, hi, lists[cur ^ 1], &s_n[cur ^ 1]);
        }
        __syncthreads();
        for (int q = t; q < ns; q += PB_T) {
            const PbSub s = lists[cur][q];
            const int start = s.start, end = s.end;
            if (end - start + 1 > PB_LANE) continue;
            const int median = pb_median(start, end);
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k) { lo[k] = s.lo[k]; hi[k] = s.hi[k]; }
            if (g.tight) {
                for (int k = 0; k < 3; ++k) { lo[k] = 1e30f; hi[k] = -1e30f; }
                for (int p = start; p <= end; ++p) {
                    const float4 r = w[p];
                    const float c[3] = {r.x, r.y, r.z};
                    for (int k = 0; k < 3; ++k) { if (c[k] < lo[k]) lo[k] = c[k]; if (c[k] > hi[k]) hi[k] = c[k]; }
                }
            }
            const int axis = pb_axis(lo, hi);
            pb_split_window(w, 0, start, end, median, axis);
            pb_small_children(g, w, s, start, end, median, axis, lo, hi, lists[cur ^ 1], &s_n[cur ^ 1]);
        }
        __syncthreads();
        if (t == 0) s_n[cur] = 0;
        cur ^= 1;
        __syncthreads();
    }
}

// ---- wide passes: the partition passes of LARGE ranges, every segment of the level at once, the whole GPU on each ---------
// The pairing of pb_zone_pass holds for a zone of any size, so a pass over a range of millions of photons is: flag the
// positions that stop either scan (tiles of PBW_TILE positions), scan the tile counts, scatter the two lists, find how
// many pairs cross (A ascends, B descends: a binary search), swap them, move the pivot.  Seven small launches per pass
// for all the level's segments together, instead of one workgroup streaming a range alone; a segment leaves the wide
// passes when its range is down to PBW_KEEP photons (k_pb_big continues from PbSeg::left / right).
constexpr int PBW_E = 8;
constexpr int PBW_TILE = PB_T * PBW_E;     // 2048
constexpr int PBW_MIN = 131072;            // segments larger than this start with wide passes
constexpr int PBW_KEEP = 32768;            // ... and stay while the range is larger than this
struct PbWide { int active, left, right, median, axis, nA, nB, np, i, tile0, ntiles; float v; };

__global__ void k_pbw_init(const PbSeg* __restrict__ segs, uint32_t nsegs, int tight, PbWide* wd, uint32_t* d_nwide) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsegs) return;
    const PbSeg sg = segs[s];
    PbWide w;
    w.active = sg.end - sg.start + 1 > PBW_MIN ? 1 : 0;
    w.left = sg.start; w.right = sg.end; w.median = pb_median(sg.start, sg.end);
    w.axis = tight ? 0 : pb_axis(sg.lo, sg.hi);
    w.nA = w.nB = w.np = w.i = 0; w.v = 0.f;
    w.tile0 = 0; w.ntiles = w.active ? (sg.end - sg.start + 1 + PBW_TILE - 1) / PBW_TILE : 0;      // for the box (tight)
    wd[s] = w;
    if (w.active) atomicAdd(d_nwide, 1u);
}
// exclusive scan of ntiles over the segments -> tile0, total -> *d_total (one workgroup)
__global__ __launch_bounds__(PB_T) void k_pbw_tiles(PbWide* wd, uint32_t nsegs, uint32_t* d_total) {
    __shared__ int s_w[PB_T / 64];
    __shared__ int s_carry;
    const int t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nsegs; base += PB_T) {
        const uint32_t s = base + t;
        const int x = s < nsegs ? wd[s].ntiles : 0;
        int incl = x;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if ((t & 63) >= d) incl += o; }
        if ((t & 63) == 63) s_w[t >> 6] = incl;
        __syncthreads();
        int woff = 0, total = 0;
        for (int k = 0; k < PB_T / 64; ++k) { if (k < (t >> 6)) woff += s_w[k]; total += s_w[k]; }
        const int carry = s_carry;
        if (s < nsegs) wd[s].tile0 = carry + woff + incl - x;
        __syncthreads();
        if (t == 0) s_carry = carry + total;
        __syncthreads();
    }
    if (t == 0) *d_total = (uint32_t)s_carry;
}
// the segment a tile belongs to: the last one whose tile0 <= tile (segments without tiles share the tile0 of their successor)
__device__ __forceinline__ uint32_t pbw_find(const PbWide* wd, uint32_t nsegs, int tile) {
    uint32_t lo = 0, hi = nsegs - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (wd[mid].tile0 <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// tight: the box of a wide segment, tile partials then one workgroup per segment
__global__ __launch_bounds__(PB_T) void k_pbw_box_tiles(const float4* __restrict__ org, const PbSeg* __restrict__ segs, const PbWide* __restrict__ wd,
                                                        uint32_t nsegs, float* partial) {
    __shared__ float s_red[6][PB_T / 64];
    const int t = threadIdx.x;
    const uint32_t s = pbw_find(wd, nsegs, (int)blockIdx.x);
    const int tile = (int)blockIdx.x - wd[s].tile0;
    const int first = segs[s].start + tile * PBW_TILE, last = min(first + PBW_TILE - 1, segs[s].end);
    float l[3] = {1e30f, 1e30f, 1e30f}, h[3] = {-1e30f, -1e30f, -1e30f};
    for (int p = first + t; p <= last; p += PB_T) {
        const float4 r = org[p];
        const float c[3] = {r.x, r.y, r.z};
        for (int k = 0; k < 3; ++k) { if (c[k] < l[k]) l[k] = c[k]; if (c[k] > h[k]) h[k] = c[k]; }
    }
    for (int k = 0; k < 3; ++k) {
        for (int d = 32; d >= 1; d >>= 1) {
            const float ol = __shfl_xor(l[k], d, 64), oh = __shfl_xor(h[k], d, 64);
            if (ol < l[k]) l[k] = ol;
            if (oh > h[k]) h[k] = oh;
        }
        if ((t & 63) == 0) { s_red[k][t >> 6] = l[k]; s_red[3 + k][t >> 6] = h[k]; }
    }
    __syncthreads();
    if (t < 3) {
        float a = s_red[t][0], b = s_red[3 + t][0];
        for (int q = 1; q < PB_T / 64; ++q) { if (s_red[t][q] < a) a = s_red[t][q]; if (s_red[3 + t][q] > b) b = s_red[3 + t][q]; }
        partial[(size_t)blockIdx.x * 6 + t] = a; partial[(size_t)blockIdx.x * 6 + 3 + t] = b;
    }
}
__global__ __launch_bounds__(PB_T) void k_pbw_box_reduce(PbSeg* segs, PbWide* wd, uint32_t nsegs, const float* __restrict__ partial) {
    __shared__ float s_red[6][PB_T / 64];
    const uint32_t s = blockIdx.x;
    const int t = threadIdx.x;
    if (s >= nsegs || !wd[s].active) return;
    const int t0 = wd[s].tile0, nt = wd[s].ntiles;
    float l[3] = {1e30f, 1e30f, 1e30f}, h[3] = {-1e30f, -1e30f, -1e30f};
    for (int q = t; q < nt; q += PB_T)
        for (int k = 0; k < 3; ++k) {
            const float a = partial[(size_t)(t0 + q) * 6 + k], b = partial[(size_t)(t0 + q) * 6 + 3 + k];
            if (a < l[k]) l[k] = a;
            if (b > h[k]) h[k] = b;
        }
    for (int k = 0; k < 3; ++k) {
        for (int d = 32; d >= 1; d >>= 1) {
            const float ol = __shfl_xor(l[k], d, 64), oh = __shfl_xor(h[k], d, 64);
            if (ol < l[k]) l[k] = ol;
            if (oh > h[k]) h[k] = oh;
        }
        if ((t & 63) == 0) { s_red[k][t >> 6] = l[k]; s_red[3 + k][t >> 6] = h[k]; }
    }
    __syncthreads();
    if (t == 0) {
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            float a = s_red[k][0], b = s_red[3 + k][0];
            for (int q = 1; q < PB_T / 64; ++q) { if (s_red[k][q] < a) a = s_red[k][q]; if (s_red[3 + k][q] > b) b = s_red[3 + k][q]; }
            lo[k] = a; hi[k] = b; segs[s].lo[k] = a; segs[s].hi[k] = b;
        }
        segs[s].have_box = 1;
        wd[s].axis = pb_axis(lo, hi);
    }
}
// per pass: which segments still take a wide pass, their pivot and tiles (ntiles; k_pbw_tiles scans them)
__global__ void k_pbw_plan(const float4* __restrict__ org, PbWide* wd, uint32_t nsegs) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsegs) return;
    PbWide w = wd[s];
    if (w.active && !(w.right > w.left && w.right - w.left + 1 > PBW_KEEP)) w.active = 0;
    w.ntiles = 0;
    if (w.active) {
        w.v = pb_key(org, w.right, w.axis);
        w.ntiles = (w.right - w.left + PBW_TILE - 1) / PBW_TILE;                 // positions left .. right - 1
    }
    wd[s] = w;
}
// flags of one tile: bit e of fa / fb = position first + t * PBW_E + e stops the left / right scan
__device__ __forceinline__ void pbw_flags(const float4* org, const PbWide& w, int first, int t, unsigned& fa, unsigned& fb) {
    fa = fb = 0;
#pragma unroll
    for (int e = 0; e < PBW_E; ++e) {
        const int p = first + t * PBW_E + e;
        if (p < w.right) {
            const float key = pb_key(org, p, w.axis);
            fa |= (!(key < w.v) ? 1u : 0u) << e;
            fb |= ((!(key > w.v) || p == w.left) ? 1u : 0u) << e;
        }
    }
}
// block-wide exclusive scan of two per-lane counts
__device__ __forceinline__ void pbw_scan2(int ca, int cb, int& offa, int& offb, int& tota, int& totb, int (*s_w)[PB_T / 64]) {
    const int t = threadIdx.x;
    int ia = ca, ib = cb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int oa = __shfl_up(ia, d, 64), ob = __shfl_up(ib, d, 64);
        if ((t & 63) >= d) { ia += oa; ib += ob; }
    }
    if ((t & 63) == 63) { s_w[0][t >> 6] = ia; s_w[1][t >> 6] = ib; }
    __syncthreads();
    int wa = 0, wb = 0; tota = totb = 0;
#pragma unroll
    for (int k = 0; k < PB_T / 64; ++k) { if (k < (t >> 6)) { wa += s_w[0][k]; wb += s_w[1][k]; } tota += s_w[0][k]; totb += s_w[1][k]; }
    offa = wa + ia - ca; offb = wb + ib - cb;
    __syncthreads();
}
__global__ __launch_bounds__(PB_T) void k_pbw_count(const float4* __restrict__ org, const PbWide* __restrict__ wd, uint32_t nsegs, int2* tcnt) {
    __shared__ int s_w[2][PB_T / 64];
    const uint32_t s = pbw_find(wd, nsegs, (int)blockIdx.x);
    const PbWide w = wd[s];
    const int first = w.left + ((int)blockIdx.x - w.tile0) * PBW_TILE;
    unsigned fa, fb;
    pbw_flags(org, w, first, threadIdx.x, fa, fb);
    int oa, ob, ta, tb;
    pbw_scan2(__popc(fa), __popc(fb), oa, ob, ta, tb, s_w);
    if (threadIdx.x == 0) tcnt[blockIdx.x] = make_int2(ta, tb);
}
__global__ __launch_bounds__(PB_T) void k_pbw_scan(PbWide* wd, uint32_t nsegs, const int2* __restrict__ tcnt, int2* toff) {
    __shared__ int s_w[2][PB_T / 64];
    __shared__ int s_carry[2];
    const uint32_t s = blockIdx.x;
    const int t = threadIdx.x;
    if (s >= nsegs || !wd[s].active) return;
    const int t0 = wd[s].tile0, nt = wd[s].ntiles;
    if (t == 0) { s_carry[0] = 0; s_carry[1] = 0; }
    __syncthreads();
    for (int base = 0; base < nt; base += PB_T) {
        const int q = base + t;
        const int2 c = q < nt ? tcnt[t0 + q] : make_int2(0, 0);
        int oa, ob, ta, tb;
        pbw_scan2(c.x, c.y, oa, ob, ta, tb, s_w);
        const int ca = s_carry[0], cb = s_carry[1];
        if (q < nt) toff[t0 + q] = make_int2(ca + oa, cb + ob);
        __syncthreads();
        if (t == 0) { s_carry[0] = ca + ta; s_carry[1] = cb + tb; }
        __syncthreads();
    }
    if (t == 0) { wd[s].nA = s_carry[0] + 1; wd[s].nB = s_carry[1]; }
}
__global__ __launch_bounds__(PB_T) void k_pbw_scatter(const float4* __restrict__ org, const PbWide* __restrict__ wd, uint32_t nsegs,
                                                      const int2* __restrict__ toff, int* gA, int* gB) {
    __shared__ int s_w[2][PB_T / 64];
    const uint32_t s = pbw_find(wd, nsegs, (int)blockIdx.x);
    const PbWide w = wd[s];
    const int tile = (int)blockIdx.x - w.tile0;
    const int first = w.left + tile * PBW_TILE;
    unsigned fa, fb;
    pbw_flags(org, w, first, threadIdx.x, fa, fb);
    int oa, ob, ta, tb;
    pbw_scan2(__popc(fa), __popc(fb), oa, ob, ta, tb, s_w);
    const int2 off = toff[blockIdx.x];
    int ka = off.x + oa, kb = off.y + ob;
#pragma unroll
    for (int e = 0; e < PBW_E; ++e) {
        const int p = first + threadIdx.x * PBW_E + e;
        if ((fa >> e) & 1u) gA[w.left + ka++] = p;
        if ((fb >> e) & 1u) gB[w.left + (w.nB - 1 - kb++)] = p;
    }
    if (tile == 0 && threadIdx.x == 0) gA[w.left + w.nA - 1] = w.right;            // the terminator: the pivot stops the left scan
}
__global__ void k_pbw_pairs(PbWide* wd, uint32_t nsegs, const int* __restrict__ gA, const int* __restrict__ gB) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsegs || !wd[s].active) return;
    const PbWide w = wd[s];
    int lo = 0, hi = min(w.nA, w.nB);                    // pairs k < np cross: gA[k] < gB[k]; a prefix
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (gA[w.left + mid] < gB[w.left + mid]) lo = mid + 1; else hi = mid;
    }
    const int np = lo;
    const int ia = gA[w.left + np];
    wd[s].np = np;
    wd[s].i = np >= 1 ? min(ia, gB[w.left + np - 1]) : ia;
}
__global__ __launch_bounds__(PB_T) void k_pbw_swap(float4* org, const PbWide* __restrict__ wd, uint32_t nsegs, const int* __restrict__ gA,
                                                   const int* __restrict__ gB) {
    const uint32_t s = pbw_find(wd, nsegs, (int)blockIdx.x);
    const PbWide w = wd[s];
    const int k0 = ((int)blockIdx.x - w.tile0) * PBW_TILE;
    for (int k = k0 + threadIdx.x; k < min(k0 + PBW_TILE, w.np); k += PB_T) {
        const int pa = gA[w.left + k], pb = gB[w.left + k];
        const float4 a = org[pa], b = org[pb];
        org[pa] = b; org[pb] = a;
    }
}
__global__ void k_pbw_finish(float4* org, PbWide* wd, PbSeg* segs, uint32_t nsegs) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsegs || !wd[s].active) return;
    PbWide w = wd[s];
    const float4 a = org[w.i], b = org[w.right];
    org[w.i] = b; org[w.right] = a;
    if (w.i >= w.median) w.right = w.i - 1;
    if (w.i <= w.median) w.left = w.i + 1;
    wd[s].left = w.left; wd[s].right = w.right;
    segs[s].left = w.left; segs[s].right = w.right;
}

// ---- the arrays around the balance -------------------------------------------------------------------------------------
__global__ void k_pb_org_from_photons(const rt_photon* __restrict__ ph, int64_t n, float4* org) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    org[i] = make_float4(ph[i].position[0], ph[i].position[1], ph[i].position[2], __int_as_float((int)i));
}
// photon.cpp:145-165 permutes in place; the net effect is new[j] = old[pbal[j]], with the plane balance_segment chose
__global__ void k_pb_permute(const rt_photon* __restrict__ old, const int32_t* __restrict__ bal, const int8_t* __restrict__ plane,
                             int64_t n, rt_photon* out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (j > n) return;
    const int32_t src = bal[j];
    rt_photon p = old[src];
    if (plane[src] >= 0) p.plane = (int16_t)plane[src];
    out[j] = p;
}
__global__ void k_pb_scale(rt_photon* ph, int64_t n, float scale) {        // ScalePhotonPower(scale, 0, n): slots 0 .. n-1
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 3; ++k) ph[i].power[k] = ph[i].power[k] * scale;
}

// image, part 1: the reference-order records, and the box of the finite positions (block partials)
__global__ void k_pb_image_ref(const rt_photon* __restrict__ heap, int64_t n, DevPhotonPos* dev, DevPhotonPow* pw, float* partial) {
    __shared__ float s[6][PB_T / 64];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    if (i <= n) {
        DevPhotonPos d; DevPhotonPow q;
        if (i == 0) { d.pos[0] = d.pos[1] = d.pos[2] = 0.f; d.packed = 0; q.power[0] = q.power[1] = q.power[2] = 0.f; q.pad = 0; }
        else {
            const rt_photon p = heap[i];
            for (int a = 0; a < 3; ++a) {
                d.pos[a] = p.position[a]; q.power[a] = p.power[a];
                if (p.position[a] > -1e30f && p.position[a] < 1e30f) { lo[a] = p.position[a]; hi[a] = p.position[a]; }
            }
            d.packed = ((uint32_t)p.plane & 3u) | ((uint32_t)p.theta << 8) | ((uint32_t)p.phi << 16);
            q.pad = 0;
        }
        dev[i] = d; pw[i] = q;
    }
    for (int a = 0; a < 3; ++a) {
        for (int d = 32; d >= 1; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64)); }
        if ((threadIdx.x & 63) == 0) { s[a][threadIdx.x >> 6] = lo[a]; s[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = s[threadIdx.x][0];
        for (int q = 1; q < PB_T / 64; ++q) r = threadIdx.x < 3 ? fminf(r, s[threadIdx.x][q]) : fmaxf(r, s[threadIdx.x][q]);
        partial[(size_t)blockIdx.x * 6 + threadIdx.x] = r;
    }
}
__global__ void k_pb_box_reduce(const float* partial, int64_t nblocks, float* box6) {
    __shared__ float s[6][PB_T];
    float r[6] = {1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f};
    for (int64_t b = threadIdx.x; b < nblocks; b += PB_T)
        for (int a = 0; a < 6; ++a) r[a] = a < 3 ? fminf(r[a], partial[b * 6 + a]) : fmaxf(r[a], partial[b * 6 + a]);
    for (int a = 0; a < 6; ++a) s[a][threadIdx.x] = r[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        float x = s[threadIdx.x][0];
        for (int q = 1; q < PB_T; ++q) x = threadIdx.x < 3 ? fminf(x, s[threadIdx.x][q]) : fmaxf(x, s[threadIdx.x][q]);
        box6[threadIdx.x] = x;
    }
}
// image, part 2: the second heap's records (photon_api.cpp: upload_heap)
__global__ void k_pb_image_second(const rt_photon* __restrict__ heap, const int32_t* __restrict__ bal2, const int8_t* __restrict__ plane2,
                                  int64_t n, DevPhotonPos* pp, DevPhotonPow* pq) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n + 1) return;
    DevPhotonPos d; DevPhotonPow q;
    if (j == 0 || j == n + 1) {
        d.pos[0] = d.pos[1] = d.pos[2] = 0.f; d.packed = 0;
        pp[j] = d;
        if (j == 0) { q.power[0] = q.power[1] = q.power[2] = 0.f; q.pad = 0; pq[0] = q; }
        return;
    }
    const int64_t half = n / 2 - 1;
    const int64_t r = n == 1 ? 1 : bal2[j];
    const rt_photon p = heap[r];
    int8_t pl = n == 1 ? 0 : plane2[r];
    if (pl < 0) pl = 0;
    const bool reachable = r == 1 || (r >> 1) < half;                // photon.cpp:58: children only below half_stored_photons
    for (int a = 0; a < 3; ++a) { d.pos[a] = p.position[a]; q.power[a] = p.power[a]; }
    d.packed = ((uint32_t)pl & 3u) | (reachable ? 0u : RT_PH_UNREACHABLE) | ((uint32_t)p.theta << 8) | ((uint32_t)p.phi << 16);
    q.pad = (uint32_t)r;
    pp[j] = d; pq[j] = q;
}
// boxes of the sub-heaps, one heap level per launch, bottom up: box[i] = photon i + box[2i] + box[2i+1]
__global__ void k_pb_boxes_level(const DevPhotonPos* __restrict__ pp, int64_t n, int64_t nb, int64_t first, int64_t last, DevPhotonPos* bx) {
    const int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > last) return;
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = pp[i].pos[a];
    for (int64_t c = 2 * i; c <= 2 * i + 1 && c <= n; ++c)
        for (int a = 0; a < 3; ++a) {
            const float cl = c <= nb ? bx[2 * c].pos[a] : pp[c].pos[a], ch = c <= nb ? bx[2 * c + 1].pos[a] : pp[c].pos[a];
            lo[a] = cl < lo[a] ? cl : lo[a]; hi[a] = hi[a] < ch ? ch : hi[a];          // std::min / std::max, as upload_heap
        }
    DevPhotonPos l, h;
    for (int a = 0; a < 3; ++a) { l.pos[a] = lo[a]; h.pos[a] = hi[a]; }
    l.packed = 0; h.packed = 0;
    bx[2 * i] = l; bx[2 * i + 1] = h;
}
// the per-lane walk's 64-byte node records
__global__ void k_pb_nodes64(const DevPhotonPos* __restrict__ pp, const DevPhotonPos* __restrict__ bx, int64_t n, int64_t nb, float* nd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    float o[16];
    if (i == 0) { for (int k = 0; k < 16; ++k) o[k] = 0.f; }
    else {
        for (int a = 0; a < 3; ++a) o[a] = pp[i].pos[a];
        uint32_t pk = pp[i].packed & ~(8u | 16u);
        for (int side = 0; side < 2; ++side) {
            const int64_t c = 2 * i + side;
            float* b = o + 4 + 6 * side;
            if (c > n) { for (int a = 0; a < 6; ++a) b[a] = 0.f; continue; }
            if (c <= nb) { for (int a = 0; a < 3; ++a) { b[a] = bx[2 * c].pos[a]; b[3 + a] = bx[2 * c + 1].pos[a]; } }
            else {
                for (int a = 0; a < 3; ++a) { b[a] = pp[c].pos[a]; b[3 + a] = pp[c].pos[a]; }
                if (pp[c].packed & RT_PH_UNREACHABLE) pk |= side ? 16u : 8u;
            }
        }
        o[3] = __uint_as_float(pk);
    }
    float4* dst = reinterpret_cast<float4*>(nd + 16 * (size_t)i);
    for (int k = 0; k < 4; ++k) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}
__global__ void k_pb_org_second(const rt_photon* __restrict__ heap, int64_t n, float4* org) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    org[i] = make_float4(heap[i].position[0], heap[i].position[1], heap[i].position[2], __int_as_float((int)i));
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + PB_T - 1) / PB_T); }

#define PBCHK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return e__; } while (0)

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T* as() const { return (T*)p; }
};

// one balance: d_org[0..n] prepared; on return d_bal[1..n], d_plane[0..n] as photon_build.cpp's Balancer leaves them
hipError_t pb_balance(float4* d_org, int64_t n, int tight, const float lo0[3], const float hi0[3], int32_t* d_bal, int8_t* d_plane,
                      hipStream_t st, int* overflow_out) {
    *overflow_out = 0;
    PBCHK(hipMemsetAsync(d_bal, 0, ((size_t)2 * n + 4) * 4, st));
    PBCHK(hipMemsetAsync(d_plane, 0xff, (size_t)n + 1, st));
    if (n < 2) return hipSuccess;
    // a segment handled by a workgroup holds > PB_SUB photons, and pushes at most two children
    // (its children hold at least a third of it each: a left-balanced tree), so the small segments, disjoint, are < 3n / PB_SUB
    const uint32_t cap_big = (uint32_t)(n / PB_SUB + 2), cap_small = (uint32_t)(4 * (n / PB_SUB + 2) + 2);
    DevBuf qa, qb, qs, cnt;
    PBCHK(qa.alloc((size_t)cap_big * sizeof(PbSeg))); PBCHK(qb.alloc((size_t)cap_big * sizeof(PbSeg)));
    PBCHK(qs.alloc((size_t)cap_small * sizeof(PbSeg))); PBCHK(cnt.alloc(64));
    PBCHK(hipMemsetAsync(cnt.p, 0, 64, st));
    const bool debug = getenv("RT_PB_DEBUG") != nullptr;
    uint32_t* d_cnt = cnt.as<uint32_t>();               // [0] next big, [1] small, [2] overflow
    PbSeg root; root.index = 1; root.start = root.left = 1; root.end = root.right = (int32_t)n; root.have_box = 0;
    for (int k = 0; k < 3; ++k) { root.lo[k] = lo0[k]; root.hi[k] = hi0[k]; }
    PbCtx g;
    g.org = d_org; g.bal = d_bal; g.plane = d_plane; g.tight = tight;
    g.nnext = d_cnt; g.cap_next = cap_big; g.small = qs.as<PbSeg>(); g.nsmall = d_cnt + 1; g.cap_small = cap_small; g.overflow = d_cnt + 2;
    g.dbg = debug ? d_cnt + 4 : nullptr;
    uint32_t nbig = 0, h[3] = {0, 0, 0};
    PbSeg* cur = qa.as<PbSeg>(); PbSeg* nxt = qb.as<PbSeg>();
    if (n > PB_SUB) { PBCHK(hipMemcpyAsync(cur, &root, sizeof root, hipMemcpyHostToDevice, st)); nbig = 1; }
    else { PBCHK(hipMemcpyAsync(qs.p, &root, sizeof root, hipMemcpyHostToDevice, st)); h[1] = 1; PBCHK(hipMemcpyAsync(d_cnt + 1, &h[1], 4, hipMemcpyHostToDevice, st)); }
    PBCHK(hipStreamSynchronize(st));                    // (root lives on the stack)
    // wide passes' arrays: the two lists (a segment's at its own positions), tile counts / offsets, box partials
    DevBuf wdesc, gA, gB, tcnt, toff, wpart;
    bool wide = n > PBW_MIN && !getenv("RT_PB_NO_WIDE");
    const size_t max_tiles = (size_t)n / PBW_TILE + cap_big + 2;
    if (wide) {
        PBCHK(wdesc.alloc((size_t)cap_big * sizeof(PbWide))); PBCHK(gA.alloc(((size_t)n + 2) * 4)); PBCHK(gB.alloc(((size_t)n + 2) * 4));
        PBCHK(tcnt.alloc(max_tiles * sizeof(int2))); PBCHK(toff.alloc(max_tiles * sizeof(int2)));
        if (tight) PBCHK(wpart.alloc(max_tiles * 24));
    }
    while (nbig > 0) {                                  // level by level while segments are larger than PB_SUB
        if (wide) {
            PbWide* wd = wdesc.as<PbWide>();
            uint32_t hw[2] = {0, 0};
            PBCHK(hipMemsetAsync(d_cnt + 3, 0, 4, st));
            k_pbw_init<<<(nbig + PB_T - 1) / PB_T, PB_T, 0, st>>>(cur, nbig, tight, wd, d_cnt + 3);
            PBCHK(hipGetLastError());
            PBCHK(hipMemcpyAsync(hw, d_cnt + 3, 4, hipMemcpyDeviceToHost, st));
            PBCHK(hipStreamSynchronize(st));
            if (hw[0] == 0) wide = false;                // segments only get smaller
            else {
                if (tight) {
                    k_pbw_tiles<<<1, PB_T, 0, st>>>(wd, nbig, d_cnt + 7);
                    PBCHK(hipMemcpyAsync(hw + 1, d_cnt + 7, 4, hipMemcpyDeviceToHost, st));
                    PBCHK(hipStreamSynchronize(st));
                    k_pbw_box_tiles<<<hw[1], PB_T, 0, st>>>(d_org, cur, wd, nbig, wpart.as<float>());
                    k_pbw_box_reduce<<<nbig, PB_T, 0, st>>>(cur, wd, nbig, wpart.as<float>());
                    PBCHK(hipGetLastError());
                }
                uint32_t iters = 0;
                for (;;) {
                    k_pbw_plan<<<(nbig + PB_T - 1) / PB_T, PB_T, 0, st>>>(d_org, wd, nbig);
                    k_pbw_tiles<<<1, PB_T, 0, st>>>(wd, nbig, d_cnt + 7);
                    PBCHK(hipMemcpyAsync(hw + 1, d_cnt + 7, 4, hipMemcpyDeviceToHost, st));
                    PBCHK(hipStreamSynchronize(st));
                    const uint32_t nt = hw[1];
                    if (nt == 0) break;
                    ++iters;
                    k_pbw_count<<<nt, PB_T, 0, st>>>(d_org, wd, nbig, tcnt.as<int2>());
                    k_pbw_scan<<<nbig, PB_T, 0, st>>>(wd, nbig, tcnt.as<int2>(), toff.as<int2>());
                    k_pbw_scatter<<<nt, PB_T, 0, st>>>(d_org, wd, nbig, toff.as<int2>(), gA.as<int>(), gB.as<int>());
                    k_pbw_pairs<<<(nbig + PB_T - 1) / PB_T, PB_T, 0, st>>>(wd, nbig, gA.as<int>(), gB.as<int>());
                    k_pbw_swap<<<nt, PB_T, 0, st>>>(d_org, wd, nbig, gA.as<int>(), gB.as<int>());
                    k_pbw_finish<<<(nbig + PB_T - 1) / PB_T, PB_T, 0, st>>>(d_org, wd, cur, nbig);
                    PBCHK(hipGetLastError());
                }
                if (debug) fprintf(stderr, "[pb] tight=%d segments=%u: %u wide, %u wide passes\n", tight, nbig, hw[0], iters);
            }
        }
        g.next = nxt;
        k_pb_big<<<nbig, PB_T, 0, st>>>(g, cur, nbig);
        PBCHK(hipGetLastError());
        PBCHK(hipMemcpyAsync(h, d_cnt, 12, hipMemcpyDeviceToHost, st));
        PBCHK(hipStreamSynchronize(st));
        if (h[2]) { *overflow_out = 1; return hipSuccess; }
        if (debug) {
            uint32_t d[3];
            PBCHK(hipMemcpy(d, d_cnt + 4, 12, hipMemcpyDeviceToHost));
            PBCHK(hipMemset(d_cnt + 4, 0, 12));
            fprintf(stderr, "[pb] tight=%d segments=%u: most passes %u, serial steps %u, chunk rounds %u\n", tight, nbig, d[0], d[1], d[2]);
        }
        nbig = h[0];
        PBCHK(hipMemsetAsync(d_cnt, 0, 4, st));
        PbSeg* x = cur; cur = nxt; nxt = x;
    }
    if (h[1] > 0) {
        g.next = nxt;
        k_pb_small<<<h[1], PB_T, 0, st>>>(g, qs.as<PbSeg>(), h[1]);
        PBCHK(hipGetLastError());
    }
    PBCHK(hipStreamSynchronize(st));
    return hipSuccess;
}

}  // namespace

// d_photons[0..n] in storage order (slot 0 unused) -> heap order in place (rt_photon_balance_host's contract), scaled like
// rt_photonmap_build, and the device image of the map (rt_photon_image_records(n) 16-byte records at *d_image, allocated
// here; the caller owns it) with its box.  *status: 0 ok, 1 = a work queue overflowed (cannot happen by the bounds above;
// the caller falls back to nothing -- it is reported as an error).
hipError_t rtk_photon_build_device(rt_photon* d_photons, int64_t n, float power_scale, DevPhotonPos** d_image, float box6[6],
                                   hipStream_t st, int* status) {
    *status = 0; *d_image = nullptr;
    for (int a = 0; a < 6; ++a) box6[a] = 0.f;
    DevBuf org, bal, plane, old;
    PBCHK(org.alloc(((size_t)n + 1) * 16)); PBCHK(bal.alloc(((size_t)2 * n + 4) * 4)); PBCHK(plane.alloc((size_t)n + 1));
    // ---- the reference's heap ----
    if (n > 1) {                                                                   // photon.cpp:135
        PBCHK(old.alloc(((size_t)n + 1) * sizeof(rt_photon)));
        PBCHK(hipMemcpyAsync(old.p, d_photons, ((size_t)n + 1) * sizeof(rt_photon), hipMemcpyDeviceToDevice, st));
        k_pb_org_from_photons<<<nblk(n + 1), PB_T, 0, st>>>(d_photons, n, org.as<float4>());
        PBCHK(hipGetLastError());
        const float lo0[3] = {-1e9f, -1e9f, -1e9f}, hi0[3] = {1e9f, 1e9f, 1e9f};
        PBCHK(pb_balance(org.as<float4>(), n, 0, lo0, hi0, bal.as<int32_t>(), plane.as<int8_t>(), st, status));
        if (*status) return hipSuccess;
        k_pb_permute<<<nblk(n), PB_T, 0, st>>>(old.as<rt_photon>(), bal.as<int32_t>(), plane.as<int8_t>(), n, d_photons);
        PBCHK(hipGetLastError());
        PBCHK(hipStreamSynchronize(st));
        (void)hipFree(old.p); old.p = nullptr;
    }
    if (power_scale != 1.0f && n > 0) { k_pb_scale<<<nblk(n), PB_T, 0, st>>>(d_photons, n, power_scale); PBCHK(hipGetLastError()); }
    // ---- the image ----
    DevBuf img;
    const size_t nrec = rt_photon_image_records(n);
    PBCHK(img.alloc(nrec * sizeof(DevPhotonPos)));
    PBCHK(hipMemsetAsync(img.p, 0, nrec * sizeof(DevPhotonPos), st));
    DevPhotonPos* dev = img.as<DevPhotonPos>();
    DevPhotonPow* pw = reinterpret_cast<DevPhotonPow*>(dev + (size_t)n + 1);
    DevPhotonPos* pp = dev + 2 * ((size_t)n + 1);
    DevPhotonPow* pq = reinterpret_cast<DevPhotonPow*>(pp + (size_t)n + 2);
    DevPhotonPos* bx = reinterpret_cast<DevPhotonPos*>(pq + (size_t)n + 1);
    float* nd = reinterpret_cast<float*>(dev + rt_photon_nodes64_offset(n));
    const int64_t nb = n / 2;
    {
        DevBuf partial, dbox;
        const unsigned blocks = nblk(n + 1);
        PBCHK(partial.alloc((size_t)blocks * 24)); PBCHK(dbox.alloc(24));
        k_pb_image_ref<<<blocks, PB_T, 0, st>>>(d_photons, n, dev, pw, partial.as<float>());
        PBCHK(hipGetLastError());
        k_pb_box_reduce<<<1, PB_T, 0, st>>>(partial.as<float>(), blocks, dbox.as<float>());
        PBCHK(hipGetLastError());
        float b[6];
        PBCHK(hipMemcpyAsync(b, dbox.p, 24, hipMemcpyDeviceToHost, st));
        PBCHK(hipStreamSynchronize(st));
        for (int a = 0; a < 3; ++a) { const bool ok = b[a] <= b[3 + a]; box6[a] = ok ? b[a] : 0.f; box6[3 + a] = ok ? b[3 + a] : 0.f; }
    }
    // ---- the second heap: same photons, every segment split on the axis of its own largest extent ----
    if (n > 1) {
        k_pb_org_second<<<nblk(n + 1), PB_T, 0, st>>>(d_photons, n, org.as<float4>());
        PBCHK(hipGetLastError());
        const float z[3] = {0.f, 0.f, 0.f};
        PBCHK(pb_balance(org.as<float4>(), n, 1, z, z, bal.as<int32_t>(), plane.as<int8_t>(), st, status));
        if (*status) return hipSuccess;
    }
    if (n >= 1) {
        k_pb_image_second<<<nblk(n + 2), PB_T, 0, st>>>(d_photons, bal.as<int32_t>(), plane.as<int8_t>(), n, pp, pq);
        PBCHK(hipGetLastError());
        int64_t top = 1;
        while (top * 2 <= nb) top *= 2;                                           // first node of the deepest level with children
        for (int64_t first = top; first >= 1; first /= 2) {
            const int64_t last = std::min<int64_t>(2 * first - 1, nb);
            if (last >= first) {
                k_pb_boxes_level<<<nblk(last - first + 1), PB_T, 0, st>>>(pp, n, nb, first, last, bx);
                PBCHK(hipGetLastError());
            }
        }
        k_pb_nodes64<<<nblk(n + 1), PB_T, 0, st>>>(pp, bx, n, nb, nd);
        PBCHK(hipGetLastError());
    }
    PBCHK(hipStreamSynchronize(st));
    *d_image = dev; img.p = nullptr;
    return hipSuccess;
}
