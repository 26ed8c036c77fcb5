// photon_build.cpp -- host-side photon-map build: the left-balanced kd-heap of
// PhotonMap::balance / balance_segment / median_split
// (Final/photon/photon.cpp:133-254), producing the same heap order and the same
// `plane` fields as the reference, so that kNN index sets can be compared
// index for index.  Runs once per map on the host; the heap is then uploaded.
#include "rt_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

// How it is run here (the result is the reference's, element for element):
//  * the permutation `porg` carries each photon's position along with its index (16-byte records), so
//    median_split's scans stream through memory instead of chasing 40-byte photons; the comparisons, and so
//    the swaps, are the reference's;
//  * after a split the two halves touch disjoint ranges of porg, disjoint heap slots and their own copy of
//    the bounding box (the only state the recursion carries), so the large ones run on their own threads.

int rt_host_threads() {
    static const int n = [] {
        if (const char* e = getenv("RT_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) return v > 64 ? 64 : v; }
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::min(16u, std::max(1u, hc));
    }();
    return n;
}

namespace {

struct Rec { float p[3]; int32_t idx; };
struct Box { float lo[3], hi[3]; };             // bbMin / bbMax; the reference never updates them from the
                                                // photons (photon.cpp:263-264 start inverted, :856-859 cannot fire)

struct Balancer {
    Rec* org;                       // porg, 1-based
    int32_t* bal;                   // pbal: heap slot -> source index
    int8_t* plane;                  // by source index
    int64_t par_min;                // segments at least this long get a thread for one half
    bool tight;                     // second heap: split on the axis of the segment's ACTUAL largest extent

    // median_split, photon.cpp:224-254
    void split(int64_t start, int64_t end, int64_t median, int axis) const {
        int64_t left = start, right = end;
        while (right > left) {
            const float v = org[right].p[axis];
            int64_t i = left - 1, j = right;
            for (;;) {
                while (org[++i].p[axis] < v) {}
                while (org[--j].p[axis] > v && j > left) {}
                if (i >= j) break;
                std::swap(org[i], org[j]);
            }
            std::swap(org[i], org[right]);
            if (i >= median) right = i - 1;
            if (i <= median) left = i + 1;
        }
    }

    // balance_segment, photon.cpp:171-222
    void segment(int64_t index, int64_t start, int64_t end, Box bb) const {
        const int64_t count = end - start + 1;
        int64_t median = 1;
        while (4 * median <= count) median += median;
        if (3 * median <= count) { median += median; median += start - 1; }
        else median = end - median + 1;

        if (tight) {
            for (int k = 0; k < 3; ++k) { bb.lo[k] = 1e30f; bb.hi[k] = -1e30f; }
            for (int64_t i = start; i <= end; ++i)
                for (int k = 0; k < 3; ++k) {
                    const float v = org[i].p[k];
                    if (v < bb.lo[k]) bb.lo[k] = v;
                    if (v > bb.hi[k]) bb.hi[k] = v;
                }
        }
        int axis = 2;
        const float ex = bb.hi[0] - bb.lo[0], ey = bb.hi[1] - bb.lo[1], ez = bb.hi[2] - bb.lo[2];
        if (ex > ey && ex > ez) axis = 0;
        else if (ey > ez) axis = 1;

        split(start, end, median, axis);
        bal[index] = org[median].idx;
        plane[org[median].idx] = (int8_t)axis;
        const float cut = org[median].p[axis];

        std::thread other;
        if (median > start) {
            if (start < median - 1) {
                Box l = bb; l.hi[axis] = cut;
                if (count >= par_min) other = std::thread([this, index, start, median, l] { segment(2 * index, start, median - 1, l); });
                else segment(2 * index, start, median - 1, l);
            } else {
                bal[2 * index] = org[start].idx;
            }
        }
        if (median < end) {
            if (median + 1 < end) {
                Box r = bb; r.lo[axis] = cut;
                segment(2 * index + 1, median + 1, end, r);
            } else {
                bal[2 * index + 1] = org[end].idx;
            }
        }
        if (other.joinable()) other.join();
    }
};

template <class F>
void parallel_ranges(int64_t n, int nthreads, F f) {
    if (nthreads <= 1 || n < (1 << 16)) { f(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
        const int64_t a = n * t / nthreads, b = n * (t + 1) / nthreads;
        if (a < b) th.emplace_back([=] { f(a, b); });
    }
    for (std::thread& t : th) t.join();
}

}  // namespace

// The order of a SECOND heap over the same photons, for the wave-shared radiance estimate: the same
// balancing algorithm, but every segment is split on the axis of its photons' real largest extent (the reference takes
// the axis from a box that photon.cpp:856-859 was meant to maintain and never does).
// With the box the reference actually uses (never filled in, Q7) the split axis stops following the data next to the
// scene's minimum faces, and coplanar photons (a wall, the caustic floor) end up in sub-heaps without spatial order,
// which every query there has to scan; this order has none.  src[j] = position in `heap` of the photon at slot j,
// plane[j] = its split axis; heap[1..n] is not modified.
void rt_photon_product_order(const rt_photon* heap, int64_t n, std::vector<int32_t>& src, std::vector<int8_t>& plane_out) {
    src.assign((size_t)n + 2, 0);
    plane_out.assign((size_t)n + 2, 0);
    if (n < 1) return;
    if (n == 1) { src[1] = 1; return; }
    const int nt = rt_host_threads();
    std::vector<Rec> org((size_t)n + 1);
    std::vector<int32_t> bal((size_t)2 * n + 4, 0);
    std::vector<int8_t> plane((size_t)n + 1, -1);
    Box bb;
    for (int k = 0; k < 3; ++k) { bb.lo[k] = 1e30f; bb.hi[k] = -1e30f; }
    for (int64_t i = 0; i <= n; ++i) {
        Rec& r = org[(size_t)i];
        for (int k = 0; k < 3; ++k) {
            r.p[k] = heap[i].position[k];
            if (i >= 1 && r.p[k] > -1e30f && r.p[k] < 1e30f) { bb.lo[k] = std::min(bb.lo[k], r.p[k]); bb.hi[k] = std::max(bb.hi[k], r.p[k]); }
        }
        r.idx = (int32_t)i;
    }
    for (int k = 0; k < 3; ++k) if (!(bb.lo[k] <= bb.hi[k])) { bb.lo[k] = 0.f; bb.hi[k] = 0.f; }
    Balancer b{org.data(), bal.data(), plane.data(), nt > 1 ? std::max<int64_t>(1 << 15, n / (4 * nt)) : INT64_MAX, true};
    b.segment(1, 1, n, bb);
    for (int64_t j = 1; j <= n; ++j) {
        src[(size_t)j] = bal[(size_t)j];
        const int8_t pl = plane[(size_t)bal[(size_t)j]];
        plane_out[(size_t)j] = pl >= 0 ? pl : 0;
    }
}

// photons[0] unused, photons[1..n] in storage order; on return photons[1..n] in heap order.
void rt_photon_balance_host(rt_photon* photons, int64_t n) {
    if (n <= 1) return;                                   // photon.cpp:135
    const int nt = rt_host_threads();
    std::vector<Rec> org((size_t)n + 1);
    std::vector<int32_t> bal((size_t)2 * n + 4, 0);
    std::vector<int8_t> plane((size_t)n + 1, -1);
    parallel_ranges(n + 1, nt, [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            Rec& r = org[(size_t)i];
            r.p[0] = photons[i].position[0]; r.p[1] = photons[i].position[1]; r.p[2] = photons[i].position[2];
            r.idx = (int32_t)i;
        }
    });
    Balancer b{org.data(), bal.data(), plane.data(), nt > 1 ? std::max<int64_t>(1 << 15, n / (4 * nt)) : INT64_MAX, false};
    Box bb;
    for (int k = 0; k < 3; ++k) { bb.lo[k] = -1e9f; bb.hi[k] = 1e9f; }
    b.segment(1, 1, n, bb);
    std::vector<Rec>().swap(org);
    // photon.cpp:145-165 permutes in place; the net effect is new[j] = old[pa1[j]]
    std::vector<rt_photon> old(photons, photons + n + 1);
    parallel_ranges(n, nt, [&](int64_t a, int64_t c) {
        for (int64_t j = a + 1; j <= c; ++j) {
            const int32_t src = bal[(size_t)j];
            photons[j] = old[(size_t)src];
            if (plane[(size_t)src] >= 0) photons[j].plane = (int16_t)plane[(size_t)src];
        }
    });
}
