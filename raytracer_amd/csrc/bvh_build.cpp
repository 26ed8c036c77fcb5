// bvh_build.cpp -- host side of the build's OWN hierarchy over a scene's tree triangles.
//
// The reference's KD-tree (scene_build.cpp) decides WHICH triangles a ray may hit: only those in a leaf whose chain of
// boxes the ray passes (KDNode::HitTest, hw4.cpp:281-328).  It is a poor index, though (monkey: leaves of ~30
// overlapping triangles, every triangle in 2.2 leaves, 42 box + 66 triangle tests per primary ray).  This file builds
// a second index over the same triangles -- a binary BVH, binned-SAH splits (triangles much larger than their node's
// median first go to a child of their own), <= RT_GROUP triangles per leaf, every triangle once -- that the default
// traversal walks instead; a hit found through it counts only if the
// reference could have found it, which the device checks per accepted hit against the reference's own boxes
// (trace_device.h: reachable_rank).  What is kept per triangle for that check: the reference leaves that hold it, in
// the reference's visit order.
#include "rt_internal.h"
#include "host_par.h"
#include <emmintrin.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

namespace {

struct BNode {
    int64_t a = 0, b = 0;               // range of `idx`
    int axis = 0;
    int32_t size = 1;                   // nodes of the subtree
    bool top = false;                   // split by all threads together; its children were finished later (Builder::fold)
    BNode *l = nullptr, *r = nullptr;   // owned by the builder's blocks
    float lo[3], hi[3];
    float gm = 0.f, amin = 0.f, emax = 0.f;   // max |e1||e2|, min |e1 x e2|, max of |e1|, |e2|
    float cone[3] = {0, 0, 1}, cos_t = 0.f, sin_t = 1.f;
    bool use_cone = false;
};

static int64_t env_i64(const char* name, int64_t dflt) { const char* e = getenv(name); return e && *e ? atoll(e) : dflt; }
static double env_f(const char* name, double dflt) { const char* e = getenv(name); return e && *e ? atof(e) : dflt; }
// tuning knobs (measured defaults)
static const int64_t CONE_MAX = env_i64("RT_BVH_CONE_MAX", 256);   // subtrees up to this many triangles get a normal cone of their own
static const int64_t LEAF_MAX = std::min<int64_t>(env_i64("RT_BVH_LEAF", RT_GROUP), 255);
static const double SIZE_THR = env_f("RT_BVH_SIZE_THR", 8.0);
static const double CONE_COS = env_f("RT_BVH_CONE_COS", 0.2);

// The tree does not depend on the schedule: nodes of at least `par_step` triangles are split one after another by all
// threads together (per-chunk bins and bounds merged in chunk order, stable partitions), the subtrees below them are
// independent tasks.
struct BvhBuilder {
    const rt_scene_desc* d;
    double dbg_slack = 1.0;             // RT_DEBUG_SLACK_SCALE (tests only), read per build
    std::vector<int32_t> idx;           // triangle (scene prim) ids, partitioned in place
    std::vector<int32_t> tmp;           // the parallel partitions' other buffer
    rt_uvec<float> gm_of, emax_of;      // per scene prim: |e1| |e2|, max(|e1|, |e2|)
    rt_uvec<double> nrm_of;             // per scene prim: unit normal (3), |e1 x e2|
    // nodes come in blocks of 4096, a thread takes a block at a time
    std::vector<std::unique_ptr<BNode[]>> blocks;
    std::mutex blocks_mu;
    static uint64_t next_serial() { static std::atomic<uint64_t> n{0}; return ++n; }
    const uint64_t serial = next_serial();      // (not `this`: the next builder may live at the same address)
    BNode* new_node() {
        struct Mine { uint64_t owner; BNode* cur; int left; };
        static thread_local Mine m{0, nullptr, 0};
        if (m.owner != serial || m.left == 0) {
            std::unique_ptr<BNode[]> blk(new BNode[4096]);
            m = Mine{serial, blk.get(), 4096};
            std::lock_guard<std::mutex> g(blocks_mu);
            blocks.push_back(std::move(blk));
        }
        --m.left;
        return m.cur++;
    }
    int nt = 1;
    size_t par_step = SIZE_MAX;
    struct Task { BNode* n; int depth; };
    std::vector<Task> tasks;

    int threads_for(int64_t cnt) const { return (int)std::max<int64_t>(1, std::min<int64_t>(nt, cnt >> 16)); }
    float centroid(int32_t p, int ax) const { const float* f = d->prims[p].f; return (f[ax] + f[3 + ax] + f[6 + ax]) * (1.0f / 3.0f); }

    // the normal cone of idx[a, b) into n (use_cone false: none narrow enough)
    void cone_of(int64_t a, int64_t b, BNode& n) const {
        static thread_local std::vector<std::array<double, 3>> nrm;
        nrm.clear();
        double ax[3] = {0, 0, 0};
        bool degenerate = false;
        n.use_cone = false;
        for (int64_t j = a; j < b; ++j) {
            const double* q = &nrm_of[(size_t)idx[(size_t)j] * 4];
            if (!(q[3] > 0)) { degenerate = true; break; }
            std::array<double, 3> u = {q[0], q[1], q[2]};
            if (!nrm.empty() && u[0] * nrm[0][0] + u[1] * nrm[0][1] + u[2] * nrm[0][2] < 0) { u[0] = -u[0]; u[1] = -u[1]; u[2] = -u[2]; }
            nrm.push_back(u); ax[0] += u[0]; ax[1] += u[1]; ax[2] += u[2];
        }
        if (degenerate || nrm.empty()) return;
        const double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        if (!(al > 1e-12)) return;
        double cmin = 1.0;
        for (const auto& u : nrm) cmin = std::min(cmin, (u[0] * ax[0] + u[1] * ax[1] + u[2] * ax[2]) / al);
        cmin = std::max(-1.0, cmin - 1e-4);                  // round the cone outwards
        if (cmin > CONE_COS) {                                     // wider cones bound nothing useful
            for (int k = 0; k < 3; ++k) n.cone[k] = (float)(ax[k] / al);
            n.cos_t = (float)cmin; n.sin_t = (float)std::min(1.0, std::sqrt(std::max(0.0, 1.0 - cmin * cmin)) + 1e-4);
            n.use_cone = true;
        }
    }

    void leaf_stats(BNode& n) const {
        float gm = 0.f, emax = 0.f; double amin = 1e300;
        for (int k = 0; k < 3; ++k) { n.lo[k] = 1e30f; n.hi[k] = -1e30f; }
        for (int64_t j = n.a; j < n.b; ++j) {
            const int32_t p = idx[(size_t)j];
            const float* f = d->prims[p].f;
            for (int k = 0; k < 3; ++k)
                for (int v = 0; v < 3; ++v) { n.lo[k] = std::min(n.lo[k], f[3 * v + k]); n.hi[k] = std::max(n.hi[k], f[3 * v + k]); }
            gm = std::max(gm, gm_of[(size_t)p]); emax = std::max(emax, emax_of[(size_t)p]);
            const double ar = nrm_of[(size_t)p * 4 + 3];
            amin = ar > 0 ? std::min(amin, ar) : 0.0;
        }
        n.gm = gm; n.emax = emax;
        n.amin = amin >= 1e299 ? 0.f : (float)(amin * (1.0 - 1e-5));
        n.use_cone = false;
        if (n.b - n.a <= CONE_MAX) cone_of(n.a, n.b, n);
    }

    // stable partition of idx[a, b) by pred with `th` threads; returns the partition point
    template <class P>
    int64_t partition(int64_t a, int64_t b, int th, P pred) {
        if (th <= 1) {
            static thread_local std::vector<int32_t> rest;
            rest.clear();
            int64_t w = a;
            for (int64_t j = a; j < b; ++j) { const int32_t p = idx[(size_t)j]; if (pred(p)) idx[(size_t)w++] = p; else rest.push_back(p); }
            if (!rest.empty()) memcpy(idx.data() + w, rest.data(), rest.size() * 4);
            return w;
        }
        const int64_t ny = rt_parallel_partition(idx.data() + a, b - a, tmp.data() + a, th, pred);
        rt_parallel_chunks(b - a, th, [this, a](int, int64_t x, int64_t y) { if (y > x) memcpy(idx.data() + a + x, tmp.data() + a + x, (size_t)(y - x) * 4); });
        return a + ny;
    }

    // a triangle's three vertices as vectors (lane 3: the next float of the record, never used), its centroid with the
    // float operations of centroid() lane by lane, its bounds.  min / max keep std::min / std::max's choice of operand:
    // _mm_min_ps(v, acc) = v < acc ? v : acc.
    struct Tri4 { __m128 c, lo, hi; };
    Tri4 tri4(int32_t p) const {
        const float* f = d->prims[p].f;
        const __m128 v0 = _mm_loadu_ps(f), v1 = _mm_loadu_ps(f + 3), v2 = _mm_loadu_ps(f + 6);
        Tri4 t;
        t.c = _mm_mul_ps(_mm_add_ps(_mm_add_ps(v0, v1), v2), _mm_set1_ps(1.0f / 3.0f));
        t.lo = _mm_min_ps(v0, _mm_min_ps(v1, v2));
        t.hi = _mm_max_ps(v0, _mm_max_ps(v1, v2));
        return t;
    }

    // bounds of the centroids of idx[a, b), and the range of |e1||e2| there
    void centroid_bounds(int64_t a, int64_t b, int th, float* clo, float* chi, float* grange) const {
        struct MM { __m128 lo, hi; float g0, g1; };
        MM one; std::vector<MM> many;
        MM* part = &one;
        if (th > 1) { many.resize((size_t)th); part = many.data(); }
        rt_parallel_chunks(b - a, th, [&](int c, int64_t x, int64_t y) {
            __m128 lo = _mm_set1_ps(1e30f), hi = _mm_set1_ps(-1e30f);
            float g0 = INFINITY, g1 = -INFINITY;
            for (int64_t j = a + x; j < a + y; ++j) {
                const int32_t p = idx[(size_t)j];
                const float* f = d->prims[p].f;
                const __m128 v = _mm_mul_ps(_mm_add_ps(_mm_add_ps(_mm_loadu_ps(f), _mm_loadu_ps(f + 3)), _mm_loadu_ps(f + 6)), _mm_set1_ps(1.0f / 3.0f));
                lo = _mm_min_ps(v, lo); hi = _mm_max_ps(v, hi);
                const float g = gm_of[(size_t)p];
                if (g < g0) g0 = g;
                if (g > g1) g1 = g;
            }
            part[c] = MM{lo, hi, g0, g1};
        });
        __m128 lo = _mm_set1_ps(1e30f), hi = _mm_set1_ps(-1e30f);
        grange[0] = INFINITY; grange[1] = -INFINITY;
        for (int c = 0; c < std::max(1, th); ++c) {
            const MM& m = part[c];
            lo = _mm_min_ps(m.lo, lo); hi = _mm_max_ps(m.hi, hi);
            if (m.g0 < grange[0]) grange[0] = m.g0;
            if (m.g1 > grange[1]) grange[1] = m.g1;
        }
        float l4[4], h4[4];
        _mm_storeu_ps(l4, lo); _mm_storeu_ps(h4, hi);
        for (int k = 0; k < 3; ++k) { clo[k] = l4[k]; chi[k] = h4[k]; }
    }

    // Binned surface-area heuristic: 16 bins of the centroids per axis, the cheapest of the 45 cuts by
    // area(left) * count(left) + area(right) * count(right).  Returns the partition point, or -1 (no cut separates the
    // centroids).  Sets n.axis.
    int64_t sah_split(BNode& n, int th, const float* clo, const float* chi) {
        constexpr int NB = 16;
        struct Bins { __m128 lo[3][NB], hi[3][NB]; int64_t cnt[3][NB]; };
        float scale[3];
        for (int ax = 0; ax < 3; ++ax) scale[ax] = chi[ax] - clo[ax] > 0.f ? (float)NB / (chi[ax] - clo[ax]) : 0.f;
        const __m128 base4 = _mm_setr_ps(clo[0], clo[1], clo[2], 0.f), scale4 = _mm_setr_ps(scale[0], scale[1], scale[2], 0.f);
        Bins one; std::vector<Bins> many;
        Bins* part = &one;
        if (th > 1) { many.resize((size_t)th); part = many.data(); }
        rt_parallel_chunks(n.b - n.a, th, [&](int c, int64_t x, int64_t y) {
            Bins& B = part[c];
            for (int ax = 0; ax < 3; ++ax) for (int q = 0; q < NB; ++q) { B.lo[ax][q] = _mm_set1_ps(1e30f); B.hi[ax][q] = _mm_set1_ps(-1e30f); B.cnt[ax][q] = 0; }
            for (int64_t j = n.a + x; j < n.a + y; ++j) {
                const Tri4 t = tri4(idx[(size_t)j]);
                // (int)((centroid - clo) * scale) per axis, as the scalar conversion does it (truncation)
                alignas(16) int32_t q4[4];
                _mm_store_si128((__m128i*)q4, _mm_cvttps_epi32(_mm_mul_ps(_mm_sub_ps(t.c, base4), scale4)));
                for (int ax = 0; ax < 3; ++ax) {
                    if (!(scale[ax] > 0.f)) continue;
                    const int q = std::min(NB - 1, std::max(0, q4[ax]));
                    B.cnt[ax][q]++;
                    B.lo[ax][q] = _mm_min_ps(t.lo, B.lo[ax][q]); B.hi[ax][q] = _mm_max_ps(t.hi, B.hi[ax][q]);
                }
            }
        });
        Bins& all = part[0];
        for (int c = 1; c < th; ++c)
            for (int ax = 0; ax < 3; ++ax)
                for (int q = 0; q < NB; ++q) {
                    all.cnt[ax][q] += part[c].cnt[ax][q];
                    all.lo[ax][q] = _mm_min_ps(part[c].lo[ax][q], all.lo[ax][q]); all.hi[ax][q] = _mm_max_ps(part[c].hi[ax][q], all.hi[ax][q]);
                }
        auto area = [](__m128 lo4, __m128 hi4) {
            float lo[4], hi[4];
            _mm_storeu_ps(lo, lo4); _mm_storeu_ps(hi, hi4);
            const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
            return x * y + y * z + z * x;
        };
        double best = 1e300; int best_ax = -1, best_cut = 0;
        for (int ax = 0; ax < 3; ++ax) {
            if (!(scale[ax] > 0.f)) continue;
            const int64_t* cnt = all.cnt[ax];
            double right_cost[NB + 1];
            right_cost[NB] = 0.0;
            {
                __m128 lo = _mm_set1_ps(1e30f), hi = _mm_set1_ps(-1e30f); int64_t c = 0;
                for (int b = NB - 1; b >= 1; --b) {
                    if (!cnt[b]) { right_cost[b] = right_cost[b + 1]; continue; }     // the same set as one bin further right
                    lo = _mm_min_ps(all.lo[ax][b], lo); hi = _mm_max_ps(all.hi[ax][b], hi);
                    c += cnt[b];
                    right_cost[b] = area(lo, hi) * (double)c;
                }
            }
            __m128 lo = _mm_set1_ps(1e30f), hi = _mm_set1_ps(-1e30f); int64_t c = 0;
            for (int cut = 1; cut < NB; ++cut) {                      // bins [0, cut) left, [cut, NB) right
                if (!cnt[cut - 1]) continue;                          // the same two sets as the cut before: never cheaper
                lo = _mm_min_ps(all.lo[ax][cut - 1], lo); hi = _mm_max_ps(all.hi[ax][cut - 1], hi);
                c += cnt[cut - 1];
                if (c == n.b - n.a) continue;
                const double cost = area(lo, hi) * (double)c + right_cost[cut];
                if (cost < best) { best = cost; best_ax = ax; best_cut = cut; }
            }
        }
        if (best_ax < 0) return -1;
        const int ax = best_ax;
        const float sc = scale[ax], base = clo[ax];
        const int cut = best_cut;
        n.axis = ax;
        return partition(n.a, n.b, th, [this, ax, sc, base, cut](int32_t p) { return std::min(NB - 1, std::max(0, (int)((centroid(p, ax) - base) * sc))) < cut; });
    }

    // cuts n in two (children made, not built); false: n is a leaf (its statistics are set)
    bool split(BNode& n, int depth, int th) {
        const int64_t cnt = n.b - n.a;
        if (cnt <= LEAF_MAX) { leaf_stats(n); return false; }
        // A node's box is inflated in proportion to the LARGEST |e1| |e2| below it (trace_device.h: lateral_slack): a
        // few big triangles among small ones (a floor under a mesh) would loosen every box above them.  They are
        // split off by size first: triangles above 8 x the median product go to a child of their own.
        int64_t mid = -1;
        float clo[3], chi[3], grange[2];
        centroid_bounds(n.a, n.b, th, clo, chi, grange);
        // (no triangle is above 8 x the median when the largest product is within 8 x the smallest: the usual case)
        bool by_size = grange[1] > (float)SIZE_THR * grange[0];
        if (by_size) {
            // ... nor when at least half of the products are at least x, the smallest float with 8 x >= the largest:
            // the median is then at least x
            float x = grange[1] / (float)SIZE_THR;
            while ((float)SIZE_THR * x < grange[1]) x = std::nextafter(x, INFINITY);
            int64_t one = 0; std::vector<int64_t> many;
            int64_t* part = &one;
            if (th > 1) { many.assign((size_t)th, 0); part = many.data(); }
            rt_parallel_chunks(cnt, th, [&](int c, int64_t x0, int64_t y0) {
                int64_t k = 0; for (int64_t j = n.a + x0; j < n.a + y0; ++j) k += gm_of[(size_t)idx[(size_t)j]] >= x ? 1 : 0;
                part[c] = k;
            });
            int64_t nge = 0; for (int c = 0; c < std::max(1, th); ++c) nge += part[c];
            if (nge >= cnt - cnt / 2) by_size = false;
        }
        if (by_size) {
            float med;
            if (th <= 1) {
                static thread_local std::vector<float> g, buf;
                g.resize((size_t)cnt);
                for (int64_t j = 0; j < cnt; ++j) g[(size_t)j] = gm_of[(size_t)idx[(size_t)(n.a + j)]];
                med = rt_select_serial(g.data(), cnt, cnt / 2, buf);
            } else {
                med = rt_parallel_select(cnt, th, cnt / 2, [this, &n](int64_t j, auto emit) { emit(gm_of[(size_t)idx[(size_t)(n.a + j)]]); });
            }
            const float thr = (float)SIZE_THR * med;
            auto small_first = [this, thr](int32_t p) { return !(gm_of[(size_t)p] > thr); };
            int64_t one = 0; std::vector<int64_t> many;
            int64_t* part = &one;
            if (th > 1) { many.assign((size_t)th, 0); part = many.data(); }
            rt_parallel_chunks(cnt, th, [&](int c, int64_t x, int64_t y) {
                int64_t k = 0; for (int64_t j = n.a + x; j < n.a + y; ++j) k += small_first(idx[(size_t)j]) ? 1 : 0;
                part[c] = k;
            });
            int64_t nsmall = 0; for (int c = 0; c < std::max(1, th); ++c) nsmall += part[c];
            if (nsmall < cnt && cnt - nsmall <= cnt / 4) {
                mid = partition(n.a, n.b, th, small_first);
                n.axis = 0;
            }
        }
        if (mid < 0 && depth < 64) mid = sah_split(n, th, clo, chi);
        if (mid < 0) {                                                // median of the centroids on their widest axis
            int ax = 0;
            if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
            if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
            n.axis = ax;
            mid = n.a + cnt / 2;
            std::nth_element(idx.begin() + n.a, idx.begin() + mid, idx.begin() + n.b, [this, ax](int32_t p, int32_t q) {
                const float cp = centroid(p, ax), cq = centroid(q, ax);
                return cp < cq || (cp == cq && p < q);
            });
        }
        n.l = new_node(); n.r = new_node();
        n.l->a = n.a; n.l->b = mid; n.r->a = mid; n.r->b = n.b;
        return true;
    }

    // an inner node's statistics from its finished children
    void join(BNode& n) const {
        for (int k = 0; k < 3; ++k) { n.lo[k] = std::min(n.l->lo[k], n.r->lo[k]); n.hi[k] = std::max(n.l->hi[k], n.r->hi[k]); }
        n.gm = std::max(n.l->gm, n.r->gm); n.emax = std::max(n.l->emax, n.r->emax);
        n.amin = std::min(n.l->amin, n.r->amin);
        n.size = 1 + n.l->size + n.r->size;
        if (n.b - n.a <= CONE_MAX) cone_of(n.a, n.b, n);   // a cone of its own where it can be narrow
    }

    void build(BNode& n, int depth) {                                 // one thread, the whole subtree
        if (!split(n, depth, 1)) return;
        build(*n.l, depth + 1); build(*n.r, depth + 1);
        join(n);
    }

    void build_top(BNode& n, int depth) {
        if ((size_t)(n.b - n.a) < par_step) { tasks.push_back(Task{&n, depth}); return; }
        if (!split(n, depth, threads_for(n.b - n.a))) return;
        n.top = true;
        build_top(*n.l, depth + 1); build_top(*n.r, depth + 1);
    }

    void fold(BNode& n) const { if (n.top) { fold(*n.l); fold(*n.r); join(n); } }

    void run(BNode& root) {
        if (par_step != SIZE_MAX) tmp.resize(idx.size());
        const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        build_top(root, 0);
        if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "BVH top: %.2f s, %zu tasks\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0, tasks.size());
        std::stable_sort(tasks.begin(), tasks.end(), [](const Task& x, const Task& y) { return x.n->b - x.n->a > y.n->b - y.n->a; });
        rt_parallel_tasks((int64_t)tasks.size(), nt, [this](int64_t i) { build(*tasks[(size_t)i].n, tasks[(size_t)i].depth); });
        fold(root);
        std::vector<int32_t>().swap(tmp);
    }

    void fill_node(const BNode& n, int depth, int o, int32_t pos, const double* croot, DevBvhNode& r) const {
        const bool leaf = !n.l;
        for (int k = 0; k < 3; ++k) {                                 // mirrored frame of layout o (rt_internal.h: DevBvhNode::b)
            const bool neg = (o >> k) & 1;
            r.b[2 * k] = neg ? -n.hi[k] : n.lo[k];
            r.b[2 * k + 1] = neg ? -n.lo[k] : n.hi[k];
        }
        {   // lateral_slack (trace_device.h) for |dir| in [0.999, 1.001], the rays traverse_bvh takes; 15 eps = 8.9e-7 is
            // applied as 1.0e-6
            const double g = 1.001 * (double)n.gm, d0 = 1e-6 - 4.0e-7 * g;
            // dbg_slack: 1 in the product; RT_DEBUG_SLACK_SCALE shrinks the LATERAL part only (the slab-rounding term stays), so
            // that tests can show which rays need it (tests/test_gpu_trace.py: adversarial grazing rays)
            r.slack = d0 > 1e-12 ? (float)std::min(1e30, dbg_slack * (1.0e-6 * g / d0) * (1.0 + 1e-6) + 2e-5) : 1e30f;
            r.cone_k = n.amin > 0.f ? (float)std::min(1e30, dbg_slack * (1.0e-6 * g / (0.999 * (double)n.amin)) * (1.0 + 1e-6)) : 1e30f;
            double dc = 0.0, rad = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double c = 0.5 * ((double)n.lo[k] + n.hi[k]) - croot[k], h = 0.5 * ((double)n.hi[k] - n.lo[k]);
                dc += c * c; rad += h * h;
            }
            r.rho = (float)((std::sqrt(dc) + std::sqrt(rad) + 0.88 * (double)n.emax) * (1.0 + 1e-5) + 1e-30);
        }
        memcpy(r.axis, n.cone, 12); r.cos_t = n.cos_t; r.sin_t = n.sin_t;
        r.skip = leaf ? (int32_t)n.a : pos + n.size;
        // bits 12..13: the split axis of an inner node (what orders its children in the eight layouts)
        r.meta = (depth & 0xff) | (leaf ? 0x100 : 0) | (n.use_cone ? 0x200 : 0) | (leaf ? (int32_t)(n.b - n.a) << 16 : (n.axis & 3) << 12);
    }

    // Layout o in pre-order at out[pos ...]: a node, the subtree of the child on the side rays of octant o come from, the
    // other child's.  The subtree sizes give every position beforehand, so subtrees are emitted independently.
    void emit(const BNode& n, int depth, int o, int32_t pos, const double* croot, DevBvhNode* out) const {
        fill_node(n, depth, o, pos, croot, out[pos]);
        if (!n.l) return;
        const bool right_first = (o >> n.axis) & 1;
        const BNode& first = right_first ? *n.r : *n.l; const BNode& second = right_first ? *n.l : *n.r;
        emit(first, depth + 1, o, pos + 1, croot, out);
        emit(second, depth + 1, o, pos + 1 + first.size, croot, out);
    }
    struct EmitTask { const BNode* n; int depth, o; int32_t pos; };
    void emit_top(const BNode& n, int depth, int o, int32_t pos, const double* croot, DevBvhNode* out, int32_t chunk, std::vector<EmitTask>& todo) const {
        if (n.size <= chunk) { todo.push_back(EmitTask{&n, depth, o, pos}); return; }
        fill_node(n, depth, o, pos, croot, out[pos]);
        const bool right_first = (o >> n.axis) & 1;
        const BNode& first = right_first ? *n.r : *n.l; const BNode& second = right_first ? *n.l : *n.r;
        emit_top(first, depth + 1, o, pos + 1, croot, out, chunk, todo);
        emit_top(second, depth + 1, o, pos + 1 + first.size, croot, out, chunk, todo);
    }
};

int depth_of(const BNode& n) { return n.l ? 1 + std::max(depth_of(*n.l), depth_of(*n.r)) : 0; }

}  // namespace

// `flat` already holds the reference trees (rt_build_flat_scene).  Fills flat->bvh_* .
int rt_build_bvh(const rt_scene_desc* d, HostFlatScene* flat, std::string* err) {
    const rt_uvec<DevNode>& ref = flat->nodes[7];
    const size_t nn = ref.size();
    const int nth = rt_host_threads();
    const bool timing = getenv("RT_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // parents of the reference's nodes (pre-order + skip pointers -> a stack walk)
    flat->ref_parent.resize(nn);
    {
        std::vector<int32_t> stack;
        for (size_t i = 0; i < nn; ++i) {
            while (!stack.empty() && ref[(size_t)stack.back()].skip <= (int32_t)i) stack.pop_back();
            flat->ref_parent[i] = stack.empty() ? -1 : stack.back();
            if (!(ref[i].meta & 0x100)) stack.push_back((int32_t)i);
        }
    }
    // per reference node: the intersection of the boxes on its chain (its own and its ancestors').  The boxes are
    // nested only up to the rounding of center -+ extend / 2, so this is the one box inside all of them; a leaf
    // whose intersection is inverted on some axis (flat boxes) gets NaN = "walk the chain".
    flat->ref_chain_box.resize(nn * 6);
    for (size_t i = 0; i < nn; ++i) {
        float* c = &flat->ref_chain_box[i * 6];
        const int32_t p = flat->ref_parent[i];
        for (int k = 0; k < 3; ++k) {
            c[k] = p < 0 ? ref[i].lo[k] : std::max(ref[i].lo[k], flat->ref_chain_box[(size_t)p * 6 + k]);
            c[3 + k] = p < 0 ? ref[i].hi[k] : std::min(ref[i].hi[k], flat->ref_chain_box[(size_t)p * 6 + 3 + k]);
        }
    }
    for (size_t i = 0; i < nn; ++i) {
        float* c = &flat->ref_chain_box[i * 6];
        if ((ref[i].meta & 0x100) && (c[0] > c[3] || c[1] > c[4] || c[2] > c[5])) c[0] = std::nanf("");
    }
    // Per triangle: the reference leaves that hold it, by rising visit rank.  The leaves lie in the reference's order
    // and the ranks rise along them (rank = leaf_first + position in the leaf's list), so a walk over the leaves meets
    // every triangle's records in rank order: thread t walks all of them and keeps those of its own range of triangles
    // -- a count, a prefix sum, the same walk again to place them.
    const int64_t np = d->nprim;
    const size_t nref = flat->rank_prim.size();
    std::vector<int32_t> first_of((size_t)np + 1, 0);
    flat->bvh_ref_leaf.resize(nref);
    const int tref = np < (1 << 16) ? 1 : nth;
    rt_parallel_chunks(np, tref, [&](int, int64_t p0, int64_t p1) {
        for (size_t r = 0; r < nref; ++r) { const int32_t p = flat->rank_prim[r]; if (p >= p0 && p < p1) first_of[(size_t)p + 1]++; }
    });
    for (int64_t p = 0; p < np; ++p) first_of[(size_t)p + 1] += first_of[(size_t)p];
    {
        std::vector<int32_t> cur(first_of.begin(), first_of.end() - 1);
        rt_parallel_chunks(np, tref, [&](int, int64_t p0, int64_t p1) {
            for (size_t i = 0; i < nn; ++i) {
                if (!(ref[i].meta & 0x100)) continue;
                const int32_t f = ref[i].leaf_first;
                for (int32_t k = 0; k < ref[i].leaf_count; ++k) {
                    const int32_t p = flat->rank_prim[(size_t)(f + k)];
                    if (p >= p0 && p < p1) flat->bvh_ref_leaf[(size_t)cur[(size_t)p]++] = DevRefLeaf{(int32_t)i, f + k};
                }
            }
        });
    }
    const double t1 = now();
    BvhBuilder B;
    B.dbg_slack = std::max(0.0, env_f("RT_DEBUG_SLACK_SCALE", 1.0));
    B.d = d;
    B.nt = nth;
    for (int64_t p = 0; p < np; ++p) if (first_of[(size_t)p + 1] > first_of[(size_t)p]) B.idx.push_back((int32_t)p);
    const int64_t nt = (int64_t)B.idx.size();
    B.gm_of.resize((size_t)np); B.emax_of.resize((size_t)np); B.nrm_of.resize((size_t)np * 4);
    rt_parallel_chunks(np, tref, [&](int, int64_t p0, int64_t p1) {
        for (int64_t p = p0; p < p1; ++p) {
            if (first_of[(size_t)p + 1] == first_of[(size_t)p]) { B.gm_of[(size_t)p] = 0.f; continue; }
            const float* f = d->prims[p].f;
            float e1[3], e2[3];
            for (int k = 0; k < 3; ++k) { e1[k] = f[3 + k] - f[k]; e2[k] = f[6 + k] - f[k]; }
            const float l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
            B.gm_of[(size_t)p] = l1 * l2; B.emax_of[(size_t)p] = std::max(l1, l2);
            const double cx = (double)e1[1] * e2[2] - (double)e1[2] * e2[1], cy = (double)e1[2] * e2[0] - (double)e1[0] * e2[2],
                         cz = (double)e1[0] * e2[1] - (double)e1[1] * e2[0];
            const double ar = std::sqrt(cx * cx + cy * cy + cz * cz);
            double* q = &B.nrm_of[(size_t)p * 4];
            q[3] = ar;
            if (ar > 0) { q[0] = cx / ar; q[1] = cy / ar; q[2] = cz / ar; } else { q[0] = q[1] = q[2] = 0.0; }
        }
    });
    for (int o = 0; o < 8; ++o) flat->bvh_nodes[o].clear();
    flat->bvh_tris.clear(); flat->bvh_tri_ref.clear();
    if (nt == 0) return RT_OK;
    B.par_step = nth > 1 ? std::max<size_t>((size_t)1 << 16, (size_t)nt / (size_t)(4 * nth)) : SIZE_MAX;
    BNode root; root.a = 0; root.b = nt;
    const double t2 = now();
    B.run(root);
    const double t3 = now();
    if (depth_of(root) + 1 > 250) { *err = "BVH deeper than 250 levels"; return RT_E_LIMIT; }
    if (root.size > 1 << 25) { *err = "BVH exceeds 2^25 nodes"; return RT_E_LIMIT; }   // 32-bit byte offsets on the device
    // triangle records in the order of idx (a leaf is a contiguous run)
    flat->bvh_tris.resize((size_t)nt); flat->bvh_tri_ref.resize((size_t)nt * 2);
    const int tpar = nt < (1 << 14) ? 1 : nth;
    rt_parallel_chunks(nt, tpar, [&](int, int64_t j0, int64_t j1) {
        for (int64_t j = j0; j < j1; ++j) {
            const int32_t p = B.idx[(size_t)j];
            const float* f = d->prims[p].f;
            DevLeafTri t;
            for (int k = 0; k < 3; ++k) { t.v0[k] = f[k]; t.e1[k] = f[3 + k] - f[k]; t.e2[k] = f[6 + k] - f[k]; }
            t.prim = p;
            t.flags = ((double)d->materials[d->prim_material[p]].refract > 0.1) ? RT_PF_SKIP_SHADOW : 0;
            t.rank = (int32_t)j;                                      // index into bvh_tri_ref
            flat->bvh_tris[(size_t)j] = t;
            flat->bvh_tri_ref[(size_t)j * 2] = first_of[(size_t)p]; flat->bvh_tri_ref[(size_t)j * 2 + 1] = first_of[(size_t)p + 1] - first_of[(size_t)p];
        }
    });
    double croot[3];
    for (int k = 0; k < 3; ++k) croot[k] = 0.5 * ((double)root.lo[k] + root.hi[k]);
    {
        std::vector<BvhBuilder::EmitTask> todo;
        const int32_t chunk = tpar > 1 ? std::max(1024, root.size / (4 * tpar)) : INT32_MAX;
        for (int o = 0; o < (flat->host_layouts ? 8 : 1); ++o) {     // 1..7: derived on the device (kernels_build.hip)
            flat->bvh_nodes[o].resize((size_t)root.size);
            B.emit_top(root, 0, o, 0, croot, flat->bvh_nodes[o].data(), chunk, todo);
        }
        rt_parallel_tasks((int64_t)todo.size(), tpar, [&](int64_t i) {
            const BvhBuilder::EmitTask& t = todo[(size_t)i];
            B.emit(*t.n, t.depth, t.o, t.pos, croot, flat->bvh_nodes[t.o].data());
        });
    }
    // compact nodes for the per-lane walks (rt_internal.h: DevBvhNode32)
    flat->bvh_c32.clear();
    if (LEAF_MAX <= 7 && !getenv("RT_BVH_NO_C32")) {
        auto up14 = [](double x) -> uint32_t {                          // smallest 14-bit float (8 exponent, 6 mantissa) >= x >= 0
            float f = (float)std::min(x, 1e30);
            if ((double)f < x) f = std::nextafter(f, INFINITY);
            uint32_t bits; memcpy(&bits, &f, 4);
            return (bits >> 17) + ((bits & 0x1ffffu) ? 1u : 0u);
        };
        const rt_uvec<DevBvhNode>& src = flat->bvh_nodes[0];
        flat->bvh_c32.resize(src.size());
        rt_parallel_chunks((int64_t)src.size(), tpar, [&](int, int64_t i0, int64_t i1) {
            for (int64_t i = i0; i < i1; ++i) {
                const DevBvhNode& n = src[(size_t)i];
                DevBvhNode32& c = flat->bvh_c32[(size_t)i];
                memcpy(c.b, n.b, sizeof c.b);
                const bool leaf = (n.meta & 0x100) != 0;
                c.link = leaf ? (0x80000000u | (uint32_t)n.skip) : (uint32_t)n.skip;
                const uint32_t s14 = up14((double)n.slack);
                uint32_t sbits = s14 << 17; float sdec; memcpy(&sdec, &sbits, 4);      // the slack the device will use
                const uint32_t r14 = up14((double)sdec * (double)n.rho * (1.0 + 1e-6));
                c.sq = (s14 << 18) | (r14 << 4) | ((leaf ? (uint32_t)(n.meta >> 16) & 7u : 0u) << 1);
            }
        });
    }
    for (int k = 0; k < 3; ++k) flat->bvh_root_centre[k] = (float)croot[k];
    if (timing) fprintf(stderr, "rt_build_bvh: reference chains + leaf refs %.2f s, tables %.2f s, SAH build %.2f s, records + 8 layouts %.2f s\n",
                        t1 - t0, t2 - t1, t3 - t2, now() - t3);
    return RT_OK;
}
