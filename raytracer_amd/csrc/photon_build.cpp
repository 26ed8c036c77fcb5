// photon_build.cpp -- host-side photon-map build: the left-balanced kd-heap of
// PhotonMap::balance / balance_segment / median_split
// (Final/photon/photon.cpp:133-254), producing the same heap order and the same
// `plane` fields as the reference, so that kNN index sets can be compared
// index for index.  Runs once per map on the host; the heap is then uploaded.
#include "rt_internal.h"
#include <cstring>
#include <vector>

namespace {

struct Balancer {
    rt_photon* ph;                  // 1-based
    std::vector<int64_t> org;       // porg: permutation being partitioned
    std::vector<int64_t> bal;       // pbal: heap slot -> source index
    float lo[3], hi[3];             // bbMin / bbMax; the reference never updates them from the
                                    // photons (photon.cpp:263-264 start inverted, :856-859 cannot fire)

    // median_split, photon.cpp:224-254
    void split(int64_t start, int64_t end, int64_t median, int axis) {
        int64_t left = start, right = end;
        while (right > left) {
            const float v = ph[org[right]].position[axis];
            int64_t i = left - 1, j = right;
            for (;;) {
                while (ph[org[++i]].position[axis] < v) {}
                while (ph[org[--j]].position[axis] > v && j > left) {}
                if (i >= j) break;
                std::swap(org[i], org[j]);
            }
            std::swap(org[i], org[right]);
            if (i >= median) right = i - 1;
            if (i <= median) left = i + 1;
        }
    }

    // balance_segment, photon.cpp:171-222
    void segment(int64_t index, int64_t start, int64_t end) {
        const int64_t count = end - start + 1;
        int64_t median = 1;
        while (4 * median <= count) median += median;
        if (3 * median <= count) { median += median; median += start - 1; }
        else median = end - median + 1;

        int axis = 2;
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        if (ex > ey && ex > ez) axis = 0;
        else if (ey > ez) axis = 1;

        split(start, end, median, axis);
        bal[index] = org[median];
        ph[bal[index]].plane = (int16_t)axis;
        const float cut = ph[bal[index]].position[axis];

        if (median > start) {
            if (start < median - 1) {
                const float keep = hi[axis];
                hi[axis] = cut;
                segment(2 * index, start, median - 1);
                hi[axis] = keep;
            } else {
                bal[2 * index] = org[start];
            }
        }
        if (median < end) {
            if (median + 1 < end) {
                const float keep = lo[axis];
                lo[axis] = cut;
                segment(2 * index + 1, median + 1, end);
                lo[axis] = keep;
            } else {
                bal[2 * index + 1] = org[end];
            }
        }
    }
};

}  // namespace

// photons[0] unused, photons[1..n] in storage order; on return photons[1..n] in heap order.
void rt_photon_balance_host(rt_photon* photons, int64_t n) {
    if (n <= 1) return;                                   // photon.cpp:135
    Balancer b;
    b.ph = photons;
    b.org.resize((size_t)n + 1);
    b.bal.assign((size_t)2 * n + 4, 0);
    for (int64_t i = 0; i <= n; ++i) b.org[(size_t)i] = i;
    for (int k = 0; k < 3; ++k) { b.lo[k] = -1e9f; b.hi[k] = 1e9f; }
    b.segment(1, 1, n);
    // photon.cpp:145-165 permutes in place; the net effect is new[j] = old[pa1[j]]
    std::vector<rt_photon> old(photons, photons + n + 1);
    for (int64_t j = 1; j <= n; ++j) photons[j] = old[(size_t)b.bal[(size_t)j]];
}
