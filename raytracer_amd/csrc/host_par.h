// host_par.h -- the little parallel machinery of the host-side builders (scene_build.cpp, bvh_build.cpp): a chunked
// parallel-for, a task pool, an exact parallel order statistic of floats (radix select) and a stable two-way
// partition.  Every routine gives the result of its serial counterpart whatever the thread count (min / max and counts
// are merged per chunk in chunk order; the partition keeps the input order on both sides).
#pragma once
#include "rt_internal.h"
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

// fn(chunk, a, b) over [0, n) cut into `nt` equal chunks, chunk c on its own thread (chunk 0 on the caller's)
template <class F>
static inline void rt_parallel_chunks(int64_t n, int nt, F fn) {
    if (nt <= 1 || n < 2) { fn(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)nt - 1);
    for (int c = 1; c < nt; ++c) th.emplace_back([&fn, c, n, nt] { fn(c, n * c / nt, n * (c + 1) / nt); });
    fn(0, (int64_t)0, n / nt);
    for (std::thread& t : th) t.join();
}

// fn(i) for i in [0, ntask), handed out one at a time to `nt` threads
template <class F>
static inline void rt_parallel_tasks(int64_t ntask, int nt, F fn) {
    if (nt <= 1 || ntask < 2) { for (int64_t i = 0; i < ntask; ++i) fn(i); return; }
    std::atomic<int64_t> next{0};
    auto work = [&next, &fn, ntask] { for (int64_t i; (i = next.fetch_add(1)) < ntask;) fn(i); };
    std::vector<std::thread> th;
    const int n = (int)std::min<int64_t>(nt, ntask);
    th.reserve((size_t)n - 1);
    for (int c = 1; c < n; ++c) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
}

// floats as unsigned keys in the order of operator< (-0 just below +0, which compare equal: either is the same answer to
// every comparison made with the result)
static inline uint32_t rt_float_key(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static inline float rt_key_float(uint32_t k) { const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &u, 4); return f; }

// The k-th smallest (0-based) of the values each(i, emit) emits for i in [0, n): what std::nth_element leaves at position
// k.  Three passes of a radix select (11 + 11 + 10 bits), each a parallel histogram.  each(i, emit) calls emit(float) any
// fixed number of times.
template <class E>
static inline float rt_parallel_select(int64_t n, int nt, int64_t k, E each) {
    uint32_t prefix = 0, mask = 0;
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = shifts[pass], nb = 1 << bits[pass];
        std::vector<std::vector<int64_t>> hist((size_t)std::max(1, nt), std::vector<int64_t>((size_t)nb, 0));
        rt_parallel_chunks(n, nt, [&](int c, int64_t a, int64_t b) {
            int64_t* h = hist[(size_t)c].data();
            for (int64_t i = a; i < b; ++i)
                each(i, [&](float f) { const uint32_t key = rt_float_key(f); if ((key & mask) == prefix) h[(key >> sh) & (uint32_t)(nb - 1)]++; });
        });
        int64_t seen = 0; int bin = nb - 1;
        for (int q = 0; q < nb; ++q) {
            int64_t cnt = 0;
            for (const auto& h : hist) cnt += h[(size_t)q];
            if (k < seen + cnt) { bin = q; break; }
            seen += cnt;
        }
        k -= seen;
        prefix |= (uint32_t)bin << sh;
        mask |= (uint32_t)(nb - 1) << sh;
    }
    return rt_key_float(prefix);
}

// stable partition of src[0, n) by pred: the elements with pred first, in their order, then the others in theirs, written
// to dst[0, n).  Returns the number with pred.
template <class T, class P>
static inline int64_t rt_parallel_partition(const T* src, int64_t n, T* dst, int nt, P pred) {
    nt = std::max(1, nt);
    std::vector<int64_t> yes((size_t)nt + 1, 0);
    rt_parallel_chunks(n, nt, [&](int c, int64_t a, int64_t b) { int64_t y = 0; for (int64_t i = a; i < b; ++i) y += pred(src[i]) ? 1 : 0; yes[(size_t)c + 1] = y; });
    for (int c = 0; c < nt; ++c) yes[(size_t)c + 1] += yes[(size_t)c];
    const int64_t ny = yes[(size_t)nt];
    rt_parallel_chunks(n, nt, [&](int c, int64_t a, int64_t b) {
        int64_t y = yes[(size_t)c], no = ny + (a - yes[(size_t)c]);
        for (int64_t i = a; i < b; ++i) { if (pred(src[i])) dst[y++] = src[i]; else dst[no++] = src[i]; }
    });
    return ny;
}

// The k-th smallest (0-based) of v[0, n), one thread: what std::nth_element leaves at position k, found in about one pass.
// A sorted sample brackets the answer; one pass counts what lies below the bracket and gathers what lies inside, the
// order statistic is then taken among the gathered few.  A bracket that misses (rare) falls back to nth_element.
// v is left in an unspecified order; `buf` is scratch.
static inline float rt_select_serial(float* v, int64_t n, int64_t k, std::vector<float>& buf) {
    if (n >= 4096) {
        constexpr int S = 512;
        float smp[S];
        for (int i = 0; i < S; ++i) smp[i] = v[(size_t)((n * (2 * (int64_t)i + 1)) / (2 * S))];
        std::sort(smp, smp + S);
        const double q = (double)k / (double)n * S;
        const int lo_i = (int)q - 40, hi_i = (int)q + 40;
        if (lo_i >= 0 && hi_i < S) {
            const float lo = smp[lo_i], hi = smp[hi_i];
            int64_t below = 0;
            buf.clear();
            for (int64_t i = 0; i < n; ++i) { const float x = v[(size_t)i]; below += x < lo ? 1 : 0; if (x >= lo && x <= hi) buf.push_back(x); }
            if (k >= below && k < below + (int64_t)buf.size()) {
                std::nth_element(buf.begin(), buf.begin() + (ptrdiff_t)(k - below), buf.end());
                return buf[(size_t)(k - below)];
            }
        }
    }
    std::nth_element(v, v + k, v + n);
    return v[(size_t)k];
}
