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
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

namespace {

struct BNode {
    int64_t a = 0, b = 0;               // range of `idx`
    int axis = 0;
    std::unique_ptr<BNode> l, r;
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

struct BvhBuilder {
    const rt_scene_desc* d;
    double dbg_slack = 1.0;             // RT_DEBUG_SLACK_SCALE (tests only), read per build
    std::vector<int32_t> idx;           // triangle (scene prim) ids, partitioned in place
    std::vector<float> gm_of;           // per scene prim: |e1| |e2|
    size_t par_min;

    float centroid(int32_t p, int ax) const { const float* f = d->prims[p].f; return (f[ax] + f[3 + ax] + f[6 + ax]) * (1.0f / 3.0f); }

    void leaf_stats(BNode& n) const {
        float gm = 0.f, emax = 0.f; double amin = 1e300, ax[3] = {0, 0, 0};
        std::vector<std::array<double, 3>> nrm;
        const bool want_cone = n.b - n.a <= CONE_MAX;
        for (int k = 0; k < 3; ++k) { n.lo[k] = 1e30f; n.hi[k] = -1e30f; }
        for (int64_t j = n.a; j < n.b; ++j) {
            const float* f = d->prims[idx[(size_t)j]].f;
            float e1[3], e2[3];
            for (int k = 0; k < 3; ++k) {
                e1[k] = f[3 + k] - f[k]; e2[k] = f[6 + k] - f[k];
                for (int v = 0; v < 3; ++v) { n.lo[k] = std::min(n.lo[k], f[3 * v + k]); n.hi[k] = std::max(n.hi[k], f[3 * v + k]); }
            }
            const float l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
            gm = std::max(gm, l1 * l2); emax = std::max(emax, std::max(l1, l2));
            const double cx = (double)e1[1] * e2[2] - (double)e1[2] * e2[1], cy = (double)e1[2] * e2[0] - (double)e1[0] * e2[2],
                         cz = (double)e1[0] * e2[1] - (double)e1[1] * e2[0];
            const double ar = std::sqrt(cx * cx + cy * cy + cz * cz);
            if (ar > 0) {
                amin = std::min(amin, ar);
                if (want_cone) {
                    std::array<double, 3> u = {cx / ar, cy / ar, cz / ar};
                    if (!nrm.empty() && u[0] * nrm[0][0] + u[1] * nrm[0][1] + u[2] * nrm[0][2] < 0) { u[0] = -u[0]; u[1] = -u[1]; u[2] = -u[2]; }
                    nrm.push_back(u); ax[0] += u[0]; ax[1] += u[1]; ax[2] += u[2];
                }
            } else amin = 0.0;
        }
        n.gm = gm; n.emax = emax;
        n.amin = amin >= 1e299 ? 0.f : (float)(amin * (1.0 - 1e-5));
        n.use_cone = false;
        if (want_cone && !nrm.empty() && amin > 0) {
            const double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            if (al > 1e-12) {
                double cmin = 1.0;
                for (const auto& u : nrm) cmin = std::min(cmin, (u[0] * ax[0] + u[1] * ax[1] + u[2] * ax[2]) / al);
                cmin = std::max(-1.0, cmin - 1e-4);                  // round the cone outwards
                if (cmin > CONE_COS) {                                     // wider cones bound nothing useful
                    for (int k = 0; k < 3; ++k) n.cone[k] = (float)(ax[k] / al);
                    n.cos_t = (float)cmin; n.sin_t = (float)std::min(1.0, std::sqrt(std::max(0.0, 1.0 - cmin * cmin)) + 1e-4);
                    n.use_cone = true;
                }
            }
        }
    }

    // Binned surface-area heuristic: 16 bins of the centroids per axis, the cheapest of the 45 cuts by
    // area(left) * count(left) + area(right) * count(right).  Returns the partition point, or -1 (no cut separates the
    // centroids).  Sets n.axis.
    int64_t sah_split(BNode& n) {
        constexpr int NB = 16;
        struct Bin { float lo[3], hi[3]; int64_t cnt; };
        float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
        for (int64_t j = n.a; j < n.b; ++j)
            for (int k = 0; k < 3; ++k) { const float c = centroid(idx[(size_t)j], k); clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c); }
        auto area = [](const float* lo, const float* hi) {
            const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
            return x * y + y * z + z * x;
        };
        double best = 1e300; int best_ax = -1, best_cut = 0;
        for (int ax = 0; ax < 3; ++ax) {
            const float ext = chi[ax] - clo[ax];
            if (!(ext > 0.f)) continue;
            const float scale = (float)NB / ext;
            Bin bins[NB];
            for (Bin& b : bins) { for (int k = 0; k < 3; ++k) { b.lo[k] = 1e30f; b.hi[k] = -1e30f; } b.cnt = 0; }
            for (int64_t j = n.a; j < n.b; ++j) {
                const int32_t p = idx[(size_t)j];
                const float* f = d->prims[p].f;
                Bin& b = bins[std::min(NB - 1, std::max(0, (int)((centroid(p, ax) - clo[ax]) * scale)))];
                b.cnt++;
                for (int v = 0; v < 3; ++v)
                    for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], f[3 * v + k]); b.hi[k] = std::max(b.hi[k], f[3 * v + k]); }
            }
            double right_cost[NB];
            {
                float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f}; int64_t c = 0;
                for (int b = NB - 1; b >= 1; --b) {
                    if (bins[b].cnt) for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], bins[b].lo[k]); hi[k] = std::max(hi[k], bins[b].hi[k]); }
                    c += bins[b].cnt;
                    right_cost[b] = c ? area(lo, hi) * (double)c : 0.0;
                }
            }
            float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f}; int64_t c = 0;
            for (int cut = 1; cut < NB; ++cut) {                      // bins [0, cut) left, [cut, NB) right
                const Bin& b = bins[cut - 1];
                if (b.cnt) for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
                c += b.cnt;
                if (c == 0 || c == n.b - n.a) continue;
                const double cost = area(lo, hi) * (double)c + right_cost[cut];
                if (cost < best) { best = cost; best_ax = ax; best_cut = cut; }
            }
        }
        if (best_ax < 0) return -1;
        const int ax = best_ax;
        const float scale = (float)NB / (chi[ax] - clo[ax]), base = clo[ax];
        const int cut = best_cut;
        auto left = [this, ax, scale, base, cut](int32_t p) { return std::min(NB - 1, std::max(0, (int)((centroid(p, ax) - base) * scale))) < cut; };
        const auto m = std::stable_partition(idx.begin() + n.a, idx.begin() + n.b, left);
        n.axis = ax;
        return (int64_t)(m - idx.begin());
    }

    void build(BNode& n, int depth = 0) {
        const int64_t cnt = n.b - n.a;
        if (cnt <= LEAF_MAX) { leaf_stats(n); return; }
        // A node's box is inflated in proportion to the LARGEST |e1| |e2| below it (trace_device.h: lateral_slack): a
        // few big triangles among small ones (a floor under a mesh) would loosen every box above them.  They are
        // split off by size first: triangles above 8 x the median product go to a child of their own.
        int64_t mid = -1;
        {
            std::vector<float> g((size_t)cnt);
            for (int64_t j = 0; j < cnt; ++j) g[(size_t)j] = gm_of[(size_t)idx[(size_t)(n.a + j)]];
            std::nth_element(g.begin(), g.begin() + cnt / 2, g.end());
            const float thr = (float)SIZE_THR * g[(size_t)(cnt / 2)];
            auto small_first = [this, thr](int32_t p) { return !(gm_of[(size_t)p] > thr); };
            const int64_t nsmall = std::count_if(idx.begin() + n.a, idx.begin() + n.b, small_first);
            if (nsmall < cnt && cnt - nsmall <= cnt / 4) {
                std::stable_partition(idx.begin() + n.a, idx.begin() + n.b, small_first);
                mid = n.a + nsmall;
                n.axis = 0;
            }
        }
        if (mid < 0 && depth < 64) mid = sah_split(n);
        if (mid < 0) {                                                // median of the centroids on their widest axis
            float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
            for (int64_t j = n.a; j < n.b; ++j)
                for (int k = 0; k < 3; ++k) { const float c = centroid(idx[(size_t)j], k); clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c); }
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
        n.l.reset(new BNode()); n.r.reset(new BNode());
        n.l->a = n.a; n.l->b = mid; n.r->a = mid; n.r->b = n.b;
        if ((size_t)cnt >= par_min) {
            std::thread other([this, &n, depth] { build(*n.l, depth + 1); });
            build(*n.r, depth + 1);
            other.join();
        } else { build(*n.l, depth + 1); build(*n.r, depth + 1); }
        for (int k = 0; k < 3; ++k) { n.lo[k] = std::min(n.l->lo[k], n.r->lo[k]); n.hi[k] = std::max(n.l->hi[k], n.r->hi[k]); }
        n.gm = std::max(n.l->gm, n.r->gm); n.emax = std::max(n.l->emax, n.r->emax);
        n.amin = std::min(n.l->amin, n.r->amin);
        if (cnt <= CONE_MAX) {                                        // a cone of its own where it can be narrow
            BNode tmp; tmp.a = n.a; tmp.b = n.b;
            leaf_stats(tmp);
            n.use_cone = tmp.use_cone; memcpy(n.cone, tmp.cone, sizeof n.cone); n.cos_t = tmp.cos_t; n.sin_t = tmp.sin_t;
        }
    }

    void emit(const BNode& n, int depth, int o, const double* croot, std::vector<DevBvhNode>& out) const {
        const size_t me = out.size();
        out.emplace_back();
        const bool leaf = !n.l;
        if (!leaf) {
            const bool right_first = (o >> n.axis) & 1;
            emit(right_first ? *n.r : *n.l, depth + 1, o, croot, out);
            emit(right_first ? *n.l : *n.r, depth + 1, o, croot, out);
        }
        DevBvhNode& r = out[me];
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
        r.skip = leaf ? (int32_t)n.a : (int32_t)out.size();
        r.meta = (depth & 0xff) | (leaf ? 0x100 : 0) | (n.use_cone ? 0x200 : 0) | (leaf ? (int32_t)(n.b - n.a) << 16 : 0);
    }
};

int depth_of(const BNode& n) { return n.l ? 1 + std::max(depth_of(*n.l), depth_of(*n.r)) : 0; }

}  // namespace

// `flat` already holds the reference trees (rt_build_flat_scene).  Fills flat->bvh_* .
int rt_build_bvh(const rt_scene_desc* d, HostFlatScene* flat, std::string* err) {
    const std::vector<DevNode>& ref = flat->nodes[7];
    const size_t nn = ref.size();
    const bool timing = getenv("RT_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // parents of the reference's nodes (pre-order + skip pointers -> a stack walk)
    flat->ref_parent.assign(nn, -1);
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
    flat->ref_chain_box.assign(nn * 6, 0.f);
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
    // per triangle: the reference leaves that hold it, in the reference's visit order (rank)
    struct Ref { int32_t prim, rank, leaf; };
    std::vector<Ref> refs;
    refs.reserve(flat->leaf_tris.size());
    for (size_t i = 0; i < nn; ++i)
        if (ref[i].meta & 0x100)
            for (int32_t k = 0; k < ref[i].leaf_count; ++k) {
                const DevLeafTri& t = flat->leaf_tris[(size_t)(ref[i].leaf_first + k)];
                refs.push_back(Ref{t.prim, t.rank, (int32_t)i});
            }
    {   // order by (prim, rank): a counting sort by prim (the keys are dense), then each triangle's few records by rank
        std::vector<int64_t> at((size_t)d->nprim + 1, 0);
        for (const Ref& r : refs) at[(size_t)r.prim + 1]++;
        for (int64_t p = 0; p < d->nprim; ++p) at[(size_t)p + 1] += at[(size_t)p];
        std::vector<Ref> sorted(refs.size());
        {
            std::vector<int64_t> cur(at.begin(), at.end() - 1);
            for (const Ref& r : refs) sorted[(size_t)cur[(size_t)r.prim]++] = r;
        }
        for (int64_t p = 0; p < d->nprim; ++p)
            if (at[(size_t)p + 1] - at[(size_t)p] > 1)
                std::sort(sorted.begin() + at[(size_t)p], sorted.begin() + at[(size_t)p + 1], [](const Ref& x, const Ref& y) { return x.rank < y.rank; });
        refs.swap(sorted);
    }
    const double t1 = now();
    BvhBuilder B;
    B.dbg_slack = std::max(0.0, env_f("RT_DEBUG_SLACK_SCALE", 1.0));
    B.d = d;
    std::vector<int32_t> first_of((size_t)d->nprim, -1), count_of((size_t)d->nprim, 0);
    flat->bvh_ref_leaf.resize(refs.size());
    for (size_t i = 0; i < refs.size(); ++i) {
        flat->bvh_ref_leaf[i] = DevRefLeaf{refs[i].leaf, refs[i].rank};
        if (first_of[(size_t)refs[i].prim] < 0) { first_of[(size_t)refs[i].prim] = (int32_t)i; B.idx.push_back(refs[i].prim); }
        count_of[(size_t)refs[i].prim]++;
    }
    const int64_t nt = (int64_t)B.idx.size();
    B.gm_of.assign((size_t)d->nprim, 0.f);
    for (int32_t p : B.idx) {
        const float* f = d->prims[p].f;
        float l1 = 0.f, l2 = 0.f;
        for (int k = 0; k < 3; ++k) { const float a = f[3 + k] - f[k], b = f[6 + k] - f[k]; l1 += a * a; l2 += b * b; }
        B.gm_of[(size_t)p] = std::sqrt(l1) * std::sqrt(l2);
    }
    for (int o = 0; o < 8; ++o) flat->bvh_nodes[o].clear();
    flat->bvh_tris.clear(); flat->bvh_tri_ref.clear();
    if (nt == 0) return RT_OK;
    const int nth = rt_host_threads();
    B.par_min = nth > 1 ? std::max<size_t>((size_t)1 << 15, (size_t)nt / (size_t)(4 * nth)) : SIZE_MAX;
    BNode root; root.a = 0; root.b = nt;
    const double t2 = now();
    B.build(root);
    const double t3 = now();
    if (depth_of(root) + 1 > 250) { *err = "BVH deeper than 250 levels"; return RT_E_LIMIT; }
    // triangle records in the order of idx (a leaf is a contiguous run)
    flat->bvh_tris.resize((size_t)nt); flat->bvh_tri_ref.resize((size_t)nt * 2);
    for (int64_t j = 0; j < nt; ++j) {
        const int32_t p = B.idx[(size_t)j];
        const float* f = d->prims[p].f;
        DevLeafTri t{};
        for (int k = 0; k < 3; ++k) { t.v0[k] = f[k]; t.e1[k] = f[3 + k] - f[k]; t.e2[k] = f[6 + k] - f[k]; }
        t.prim = p;
        t.flags = ((double)d->materials[d->prim_material[p]].refract > 0.1) ? RT_PF_SKIP_SHADOW : 0;
        t.rank = (int32_t)j;                                      // index into bvh_tri_ref
        flat->bvh_tris[(size_t)j] = t;
        flat->bvh_tri_ref[(size_t)j * 2] = first_of[(size_t)p]; flat->bvh_tri_ref[(size_t)j * 2 + 1] = count_of[(size_t)p];
    }
    double croot[3];
    for (int k = 0; k < 3; ++k) croot[k] = 0.5 * ((double)root.lo[k] + root.hi[k]);
    if (nth > 1 && nt > 4096) {
        std::vector<std::thread> th;
        for (int o = 0; o < 8; ++o) th.emplace_back([&B, &root, flat, o, &croot] { B.emit(root, 0, o, croot, flat->bvh_nodes[o]); });
        for (std::thread& t : th) t.join();
    } else {
        for (int o = 0; o < 8; ++o) B.emit(root, 0, o, croot, flat->bvh_nodes[o]);
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
        const std::vector<DevBvhNode>& src = flat->bvh_nodes[0];
        flat->bvh_c32.resize(src.size());
        for (size_t i = 0; i < src.size(); ++i) {
            const DevBvhNode& n = src[i];
            DevBvhNode32& c = flat->bvh_c32[i];
            memcpy(c.b, n.b, sizeof c.b);
            const bool leaf = (n.meta & 0x100) != 0;
            c.link = leaf ? (0x80000000u | (uint32_t)n.skip) : (uint32_t)n.skip;
            const uint32_t s14 = up14((double)n.slack);
            uint32_t sbits = s14 << 17; float sdec; memcpy(&sdec, &sbits, 4);      // the slack the device will use
            const uint32_t r14 = up14((double)sdec * (double)n.rho * (1.0 + 1e-6));
            c.sq = (s14 << 18) | (r14 << 4) | ((leaf ? (uint32_t)(n.meta >> 16) & 7u : 0u) << 1);
        }
    }
    for (int k = 0; k < 3; ++k) flat->bvh_root_centre[k] = (float)croot[k];
    if (timing) fprintf(stderr, "rt_build_bvh: reference chains + leaf refs %.2f s, tables %.2f s, SAH build %.2f s, records + 8 layouts %.2f s\n",
                        t1 - t0, t2 - t1, t3 - t2, now() - t3);
    if (flat->bvh_nodes[0].size() > (size_t)1 << 25) { *err = "BVH exceeds 2^25 nodes"; return RT_E_LIMIT; }   // 32-bit byte offsets on the device
    return RT_OK;
}
