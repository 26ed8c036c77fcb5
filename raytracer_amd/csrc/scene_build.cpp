// scene_build.cpp -- host side of rt_scene_create: builds the per-object
// KD-trees by the reference rule (KDNode::CalcBound / BuildKD,
// HW4/hw4/hw4.cpp:330-490) and flattens them into the stackless device layout
// described in rt_internal.h.  Runs once per scene on the host; the result is
// uploaded to HBM by api.cpp.
#include "rt_internal.h"
#include "host_par.h"
#include <algorithm>
#include <array>
#include <cmath>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

static float dbg_slack_scale() { const char* e = getenv("RT_DEBUG_SLACK_SCALE"); const double v = e && *e ? atof(e) : 1.0; return (float)(v < 0.0 ? 0.0 : v); }

namespace {

// algebra3 semantics needed on the host (algebra3.cpp:280-282, 295-302, 345-347)
inline float inv_via_double(float d) { return (float)(1.0 / (double)d); }
inline float length3(const float* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
inline void normalize3(float* v) {
    float di = inv_via_double(length3(v));
    v[0] *= di; v[1] *= di; v[2] *= di;
}

struct BuildNode {
    std::vector<int32_t> prims;      // kdMeshes (triangle ids, file order preserved)
    std::unique_ptr<BuildNode> left, right;
    float center[3] = {0, 0, 0}, extend[3] = {0, 0, 0};
    int axis = 0;
    bool leaf = false;
    float model_scale = 2.0f;        // KDNode ctor default (hw4.h:203); only roots get another value
    // filled while flattening in reference order, reused by the other seven layouts
    int32_t leaf_first = 0, group_first = 0;
    float margin = 0.f;
};

// The tree does not depend on the schedule.  Nodes of at least `par_step` triangles -- the top of the tree, where there
// are fewer nodes than threads -- are split one after another by all threads together (bounds, the split value as an
// exact order statistic, the two lists); the subtrees below them are independent tasks, each built by one thread.
struct Builder {
    const rt_prim* prims;
    int nt;
    size_t par_step;
    std::atomic<int64_t> built{0};
    std::atomic<int32_t> max_depth{0};
    struct Task { BuildNode* n; int depth; };
    std::vector<Task> tasks;

    int threads_for(size_t cnt) const { return (int)std::max<size_t>(1, std::min<size_t>((size_t)nt, cnt >> 16)); }

    // CalcBound, hw4.cpp:330-389
    void bound(BuildNode& n, int th) const {
        struct MM { float mn[3], mx[3]; };
        std::vector<MM> part((size_t)th);
        rt_parallel_chunks((int64_t)n.prims.size(), th, [&](int c, int64_t a, int64_t b) {
            float mn[3] = {1e9f, 1e9f, 1e9f}, mx[3] = {-1e9f, -1e9f, -1e9f};
            for (int64_t i = a; i < b; ++i) {
                const float* f = prims[n.prims[(size_t)i]].f;
                for (int j = 0; j < 3; ++j)
                    for (int k = 0; k < 3; ++k) {
                        float v = f[3 * j + k];
                        if (v < mn[k]) mn[k] = v;
                        if (v > mx[k]) mx[k] = v;
                    }
            }
            memcpy(part[(size_t)c].mn, mn, 12); memcpy(part[(size_t)c].mx, mx, 12);
        });
        float mn[3] = {1e9f, 1e9f, 1e9f}, mx[3] = {-1e9f, -1e9f, -1e9f};
        for (const MM& p : part)
            for (int k = 0; k < 3; ++k) { if (p.mn[k] < mn[k]) mn[k] = p.mn[k]; if (p.mx[k] > mx[k]) mx[k] = p.mx[k]; }
        const float half = inv_via_double(2.0f);           // operator/(vec3, 2.0f)
        for (int k = 0; k < 3; ++k) {
            n.center[k] = (mn[k] + mx[k]) * half;
            n.extend[k] = mx[k] - mn[k];
        }
        float best = -1e9f;
        for (int k = 0; k < 3; ++k)
            if (n.extend[k] > best) { best = n.extend[k]; n.axis = k; }
    }

    // One step of BuildKD, hw4.cpp:393-490, with `th` threads.  The reference sorts every vertex coordinate on the split
    // axis and takes element size/2: the order statistic, which nth_element (one thread) and the radix select (many)
    // both return.  True: the node was subdivided.
    bool split(BuildNode& n, int depth, int th) {
        for (int32_t cur = max_depth.load(); depth > cur && !max_depth.compare_exchange_weak(cur, depth);) {}
        built.fetch_add(1);
        bound(n, th);
        const int ax = n.axis;
        const int64_t cnt = (int64_t)n.prims.size();
        float split;
        if (th <= 1) {
            static thread_local std::vector<float> coord, buf;
            coord.resize((size_t)cnt * 3);
            for (int64_t i = 0; i < cnt; ++i) {
                const float* f = prims[n.prims[(size_t)i]].f;
                coord[(size_t)i * 3] = f[ax]; coord[(size_t)i * 3 + 1] = f[3 + ax]; coord[(size_t)i * 3 + 2] = f[6 + ax];
            }
            split = rt_select_serial(coord.data(), cnt * 3, (cnt * 3) / 2, buf);
        } else {
            split = rt_parallel_select(cnt, th, (cnt * 3) / 2, [this, &n, ax](int64_t i, auto emit) {
                const float* f = prims[n.prims[(size_t)i]].f;
                emit(f[ax]); emit(f[3 + ax]); emit(f[6 + ax]);
            });
        }
        n.left.reset(new BuildNode());
        n.right.reset(new BuildNode());
        auto sides = [this, ax, split](int32_t id, bool& l, bool& r) {
            const float* f = prims[id].f;
            r = f[ax] >= split || f[3 + ax] >= split || f[6 + ax] >= split;
            l = f[ax] < split || f[3 + ax] < split || f[6 + ax] < split;
        };
        size_t both = 0;
        if (th <= 1) {
            static thread_local std::vector<int32_t> ll, rr;
            ll.clear(); rr.clear();
            for (int32_t id : n.prims) {
                bool l, r; sides(id, l, r);
                if (r) rr.push_back(id);
                if (l) ll.push_back(id);
                both += (l && r);
            }
            n.left->prims.assign(ll.begin(), ll.end()); n.right->prims.assign(rr.begin(), rr.end());
        } else {
            struct Cnt { int64_t l, r, both; };
            std::vector<Cnt> part((size_t)th + 1, Cnt{0, 0, 0});
            rt_parallel_chunks(cnt, th, [&](int c, int64_t a, int64_t b) {
                Cnt k{0, 0, 0};
                for (int64_t i = a; i < b; ++i) { bool l, r; sides(n.prims[(size_t)i], l, r); k.l += l; k.r += r; k.both += (l && r); }
                part[(size_t)c + 1] = k;
            });
            for (int c = 0; c < th; ++c) { part[(size_t)c + 1].l += part[(size_t)c].l; part[(size_t)c + 1].r += part[(size_t)c].r; part[(size_t)c + 1].both += part[(size_t)c].both; }
            n.left->prims.resize((size_t)part[(size_t)th].l); n.right->prims.resize((size_t)part[(size_t)th].r);
            both = (size_t)part[(size_t)th].both;
            rt_parallel_chunks(cnt, th, [&](int c, int64_t a, int64_t b) {
                int64_t pl = part[(size_t)c].l, pr = part[(size_t)c].r;
                for (int64_t i = a; i < b; ++i) {
                    const int32_t id = n.prims[(size_t)i];
                    bool l, r; sides(id, l, r);
                    if (r) n.right->prims[(size_t)pr++] = id;
                    if (l) n.left->prims[(size_t)pl++] = id;
                }
            });
        }
        const bool subdivide = (float)both < (float)n.left->prims.size() / n.model_scale &&
                               (float)both < (float)n.right->prims.size() / n.model_scale;
        if (subdivide) {
            std::vector<int32_t>().swap(n.prims);       // interior lists are never read by traversal
        } else {
            n.leaf = true;                              // hw4.cpp:484: this node tests its own list
            n.left.reset(); n.right.reset();            // its children are never visited (hw4.cpp:827)
        }
        return subdivide;
    }

    void build(BuildNode& n, int depth) {               // one thread, the whole subtree
        if (!split(n, depth, 1)) return;
        build(*n.left, depth + 1);
        build(*n.right, depth + 1);
    }

    void build_top(BuildNode& n, int depth) {
        if (n.prims.size() < par_step) { tasks.push_back(Task{&n, depth}); return; }
        if (!split(n, depth, threads_for(n.prims.size()))) return;
        build_top(*n.left, depth + 1);
        build_top(*n.right, depth + 1);
    }

    void run(BuildNode& root) {
        const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        build_top(root, 0);
        if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "KD top: %.2f s, %zu tasks\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0, tasks.size());
        std::stable_sort(tasks.begin(), tasks.end(), [](const Task& a, const Task& b) { return a.n->prims.size() > b.n->prims.size(); });
        rt_parallel_tasks((int64_t)tasks.size(), nt, [this](int64_t i) { build(*tasks[(size_t)i].n, tasks[(size_t)i].depth); });
    }
};

// the device build's breadth-first nodes (kernels_kdbuild.hip) as the tree the flattener reads
void from_bfs(BuildNode& n, const std::vector<RtKdBfsNode>& v, int32_t i, const std::vector<int32_t>& leaf_ids) {
    const RtKdBfsNode& s = v[(size_t)i];
    for (int k = 0; k < 3; ++k) { n.center[k] = s.center[k]; n.extend[k] = s.extend[k]; }
    n.axis = s.axis;
    if (s.child < 0) {
        n.leaf = true;
        n.prims.assign(leaf_ids.begin() + s.leaf_off, leaf_ids.begin() + s.leaf_off + s.cnt);
        return;
    }
    n.left.reset(new BuildNode()); n.right.reset(new BuildNode());
    from_bfs(*n.left, v, s.child, leaf_ids);
    from_bfs(*n.right, v, s.child + 1, leaf_ids);
}
// ... with the subtrees below the first levels made by all threads
void from_bfs_parallel(BuildNode& root, const std::vector<RtKdBfsNode>& v, const std::vector<int32_t>& leaf_ids, int nt) {
    struct Job { BuildNode* n; int32_t i; };
    std::vector<Job> level{{&root, 0}}, jobs;
    for (int depth = 0; depth < 8 && !level.empty(); ++depth) {
        std::vector<Job> next;
        for (const Job& j : level) {
            const RtKdBfsNode& s = v[(size_t)j.i];
            if (s.child < 0) { jobs.push_back(j); continue; }
            for (int k = 0; k < 3; ++k) { j.n->center[k] = s.center[k]; j.n->extend[k] = s.extend[k]; }
            j.n->axis = s.axis;
            j.n->left.reset(new BuildNode()); j.n->right.reset(new BuildNode());
            next.push_back(Job{j.n->left.get(), s.child}); next.push_back(Job{j.n->right.get(), s.child + 1});
        }
        level.swap(next);
    }
    jobs.insert(jobs.end(), level.begin(), level.end());
    rt_parallel_tasks((int64_t)jobs.size(), nt, [&](int64_t k) { from_bfs(*jobs[(size_t)k].n, v, jobs[(size_t)k].i, leaf_ids); });
}
bool same_tree(const BuildNode& a, const BuildNode& b) {
    if (a.leaf != b.leaf || memcmp(a.center, b.center, 12) != 0 || memcmp(a.extend, b.extend, 12) != 0 || a.axis != b.axis) return false;
    if (a.leaf) return a.prims == b.prims;
    return same_tree(*a.left, *b.left) && same_tree(*a.right, *b.right);
}

void fill_tri(float* a, const float* f) {
    for (int k = 0; k < 3; ++k) {
        a[k] = f[k];
        a[3 + k] = f[3 + k] - f[k];       // edge1 = vertices[1]-vertices[0]
        a[6 + k] = f[6 + k] - f[k];       // edge2 = vertices[2]-vertices[0]
    }
}

struct Flattener {
    const rt_scene_desc* d;
    HostFlatScene* out;
    float dbg_slack;                 // RT_DEBUG_SLACK_SCALE, read once per build

    // The records and groups of one leaf, written in place at the leaf's offsets (leaf_first = the reference's visit rank
    // of its first record): the reference's list order gives the rank, a Morton curve the storage order.
    void make_leaf(BuildNode& n, std::vector<uint64_t>& refs) const {
        const size_t cnt = n.prims.size();
        refs.resize(cnt);
        for (size_t k = 0; k < cnt; ++k) {
            const float* f = d->prims[n.prims[k]].f;
            uint32_t code = 0;
            for (int ax = 0; ax < 3; ++ax) {
                const float c = (f[ax] + f[3 + ax] + f[6 + ax]) * (1.0f / 3.0f);
                const float lo = n.center[ax] - n.extend[ax] * 0.5f, ext = n.extend[ax] > 0.f ? n.extend[ax] : 1.f;
                float u = (c - lo) / ext; u = u < 0.f ? 0.f : (u > 1.f ? 1.f : u);
                uint32_t q = (uint32_t)(u * 1023.0f);
                q = (q | (q << 16)) & 0x030000FF; q = (q | (q << 8)) & 0x0300F00F;
                q = (q | (q << 4)) & 0x030C30C3; q = (q | (q << 2)) & 0x09249249;
                code |= q << ax;
            }
            refs[k] = (uint64_t)code << 32 | (uint64_t)k;          // by code, ties in list order
            out->rank_prim[(size_t)n.leaf_first + k] = n.prims[k];
        }
        std::sort(refs.begin(), refs.end());
        float margin = 0.f;
        DevLeafTri* tris = &out->leaf_tris[(size_t)n.leaf_first];
        DevLeafGroup* groups = out->leaf_groups.data() + (size_t)n.group_first;
        const float slack_scale = dbg_slack;
        for (size_t g0 = 0; g0 < cnt; g0 += RT_GROUP) {
            DevLeafGroup g{};
            for (int k = 0; k < 3; ++k) { g.lo[k] = 1e30f; g.hi[k] = -1e30f; }
            float gm = 0.f;
            double amin = 1e300, ax[3] = {0, 0, 0};
            std::array<std::array<double, 3>, RT_GROUP> nrm;
            int nn = 0;
            for (size_t j = g0; j < std::min(cnt, g0 + RT_GROUP); ++j) {
                const int32_t k0 = (int32_t)(refs[j] & 0xffffffffu);
                const int32_t id = n.prims[(size_t)k0];
                const float* f = d->prims[id].f;
                DevLeafTri t;
                float a[9]; fill_tri(a, f);
                memcpy(t.v0, a, 12); memcpy(t.e1, a + 3, 12); memcpy(t.e2, a + 6, 12);
                t.prim = id;
                t.flags = ((double)d->materials[d->prim_material[id]].refract > 0.1) ? RT_PF_SKIP_SHADOW : 0;
                t.rank = n.leaf_first + k0;
                tris[j] = t;
                for (int v = 0; v < 3; ++v)
                    for (int k = 0; k < 3; ++k) { g.lo[k] = std::min(g.lo[k], f[3 * v + k]); g.hi[k] = std::max(g.hi[k], f[3 * v + k]); }
                // |edge1|*|edge2| bounds the cancellation in det and in the t / u / v numerators
                const float l1 = std::sqrt(a[3] * a[3] + a[4] * a[4] + a[5] * a[5]);
                const float l2 = std::sqrt(a[6] * a[6] + a[7] * a[7] + a[8] * a[8]);
                gm = std::max(gm, l1 * l2);
                // normal cone: det = dir . (edge2 x edge1), so |det| >= |edge1 x edge2| * |dir . n|
                const double cx = (double)a[4] * a[8] - (double)a[5] * a[7], cy = (double)a[5] * a[6] - (double)a[3] * a[8],
                             cz = (double)a[3] * a[7] - (double)a[4] * a[6];
                const double ar = std::sqrt(cx * cx + cy * cy + cz * cz);
                if (ar > 0) {
                    amin = std::min(amin, ar);
                    std::array<double, 3> u = {cx / ar, cy / ar, cz / ar};
                    // normals are taken up to sign: align with the first one
                    if (nn > 0 && u[0] * nrm[0][0] + u[1] * nrm[0][1] + u[2] * nrm[0][2] < 0) { u[0] = -u[0]; u[1] = -u[1]; u[2] = -u[2]; }
                    nrm[(size_t)nn++] = u; ax[0] += u[0]; ax[1] += u[1]; ax[2] += u[2];
                }
            }
            // RT_DEBUG_SLACK_SCALE (tests only; 1 in the product) shrinks the lateral inflation of the group boxes, which
            // lateral_slack (trace_device.h) takes in proportion to gm; the distance-culling margin below keeps the real gm
            g.gm = gm * slack_scale;
            const double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            double cmin = 0.0;
            if (al > 1e-12 && nn > 0) {
                cmin = 1.0;
                for (int q = 0; q < nn; ++q) cmin = std::min(cmin, (nrm[(size_t)q][0] * ax[0] + nrm[(size_t)q][1] * ax[1] + nrm[(size_t)q][2] * ax[2]) / al);
                for (int k = 0; k < 3; ++k) g.axis[k] = (float)(ax[k] / al);
            }
            // round the cone outwards: a slightly smaller cosine, a slightly larger sine
            cmin = std::max(0.0, cmin - 1e-4);
            g.cos_t = (float)cmin; g.sin_t = (float)std::min(1.0, std::sqrt(std::max(0.0, 1.0 - cmin * cmin)) + 1e-4);
            g.area2_min = nn == 0 ? 0.f : (float)(amin * (1.0 - 1e-5));
            // the cone costs ~12 vector ops per test: only worth it where the worst-case slack (|det| = 1e-6) is not
            // already small against the box (typical reach of a ray through the scene: a few units)
            float ext = 0.f;
            for (int k = 0; k < 3; ++k) ext = std::max(ext, g.hi[k] - g.lo[k]);
            g.pad[0] = (2.0f * gm * 4.0f > 0.1f * ext) ? 1.0f : 0.0f;
            // what the group adds to a ray's reach (trace_device.h: lateral_slack): its triangles may stick out of the
            // leaf's box by the group's L1 diagonal, plus 0.44 (|e1| + |e2|) <= 0.88 of that diagonal
            g.pad[1] = 1.9f * ((g.hi[0] - g.lo[0]) + (g.hi[1] - g.lo[1]) + (g.hi[2] - g.lo[2]));
            margin = std::max(margin, gm);
            groups[g0 / RT_GROUP] = g;
        }
        n.margin = margin;
    }

    static void collect_leaves(BuildNode& n, std::vector<BuildNode*>& leaves) {
        if (n.leaf) { leaves.push_back(&n); return; }
        collect_leaves(*n.right, leaves);
        collect_leaves(*n.left, leaves);
    }

    // every leaf's offsets in the reference's leaf order, then the records of blocks of leaves in parallel
    void make_leaves(BuildNode& root) const {
        std::vector<BuildNode*> leaves;
        collect_leaves(root, leaves);
        size_t refs = out->leaf_tris.size(), groups = out->leaf_groups.size();
        for (BuildNode* l : leaves) {
            l->leaf_first = (int32_t)refs; l->group_first = (int32_t)groups;
            refs += l->prims.size(); groups += (l->prims.size() + RT_GROUP - 1) / RT_GROUP;
        }
        out->leaf_tris.resize(refs);
        out->leaf_groups.resize(groups);
        out->rank_prim.resize(refs);
        const int nt = leaves.size() < 256 ? 1 : rt_host_threads();
        constexpr int64_t BLOCK = 128;
        const int64_t nblock = ((int64_t)leaves.size() + BLOCK - 1) / BLOCK;
        rt_parallel_tasks(nblock, nt, [this, &leaves](int64_t blk) {
            std::vector<uint64_t> scratch;
            for (size_t i = (size_t)(blk * BLOCK); i < std::min(leaves.size(), (size_t)((blk + 1) * BLOCK)); ++i) make_leaf(*leaves[i], scratch);
        });
    }

    // Reference order (layout 7): pre-order with the RIGHT child first, the order in which the reference
    // pops nodes (hw4.cpp:831-836); folds the culling slack of every subtree.
    int32_t emit_reference(BuildNode& n, int depth) {
        rt_uvec<DevNode>& nodes = out->nodes[7];
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        if (!n.leaf) {
            emit_reference(*n.right, depth + 1);
            emit_reference(*n.left, depth + 1);
            n.margin = std::max(n.left->margin, n.right->margin);
        }
        fill(nodes[(size_t)me], n, depth);
        nodes[(size_t)me].skip = (int32_t)nodes.size();
        return me;
    }

    // Layout o: the child on the side the ray comes from goes first (bit a of o: direction negative on axis a,
    // so the larger-coordinate child, `right`, is nearer).
    void emit_octant(const BuildNode& n, int depth, int o) {
        rt_uvec<DevNode>& nodes = out->nodes[o];
        const size_t me = nodes.size();
        nodes.emplace_back();
        if (!n.leaf) {
            const bool right_first = (o >> n.axis) & 1;
            emit_octant(right_first ? *n.right : *n.left, depth + 1, o);
            emit_octant(right_first ? *n.left : *n.right, depth + 1, o);
        }
        fill(nodes[me], n, depth);
        nodes[me].skip = (int32_t)nodes.size();
    }

    static void fill(DevNode& r, const BuildNode& n, int depth) {
        for (int k = 0; k < 3; ++k) {
            r.lo[k] = n.center[k] - n.extend[k] / 2.0f;
            r.hi[k] = n.center[k] + n.extend[k] / 2.0f;
        }
        r.meta = depth | (n.leaf ? 0x100 : (n.axis & 3) << 12);
        // 4x safety on the bound, plus a floor that covers the rounding of the slab arithmetic itself
        r.margin = 4.0f * n.margin + 1e-4f;
        r.leaf_first = n.leaf ? n.leaf_first : 0;
        r.leaf_count = n.leaf ? (int32_t)n.prims.size() : 0;
        r.group_first = n.leaf ? n.group_first : 0;
    }
};

}  // namespace

float rt_float_above(double t) {
    float f = (float)t;
    while ((double)f > t) f = std::nextafterf(f, -INFINITY);
    while (!((double)f > t)) f = std::nextafterf(f, INFINITY);
    return f;
}

int rt_build_flat_scene(const rt_scene_desc* d, HostFlatScene* out, std::string* err) {
    const int64_t np = d->nprim;
    for (int64_t i = 0; i < np; ++i) {
        if (d->prim_material[i] < 0 || d->prim_material[i] >= d->nmaterial) {
            *err = "prim_material index out of range"; return RT_E_INVALID;
        }
        if (d->prim_type[i] == RT_PRIM_SPHERE) out->nsphere++; else if (d->prim_type[i] == RT_PRIM_TRIANGLE) out->ntri++;
    }
    std::vector<uint8_t> in_tree((size_t)np, 0);
    for (int32_t o = 0; o < d->nobject; ++o) {
        const int64_t first = d->obj_first[o], cnt = d->obj_count[o];
        if (first < 0 || cnt < 0 || first + cnt > np) { *err = "object range out of bounds"; return RT_E_INVALID; }
        BuildNode root;
        for (int64_t i = first; i < first + cnt; ++i)
            if (d->prim_type[i] == RT_PRIM_TRIANGLE) {
                if (in_tree[(size_t)i]) { *err = "object ranges overlap"; return RT_E_INVALID; }
                root.prims.push_back((int32_t)i); in_tree[(size_t)i] = 1;
            }
        if (root.prims.empty()) continue;                       // hw4.cpp:731
        // hw4.cpp:736-738: modelScale = max(1.5, log10(count) - 1), root only
        const float tempF = (float)(std::log10((double)(int)cnt) - 1.0f);
        root.model_scale = tempF < 1.5f ? 1.5f : tempF;
        const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        const int nt = rt_host_threads();
        // Where: on the device (kernels_kdbuild.hip: every node of a level at once) from RT_KD_DEVICE_MIN triangles up, on
        // the host's threads below that and whenever the device build declines (no device, workspace exceeded).
        // RT_KD_BUILD=host|device forces either; RT_CHECK_KD=1 builds both and fails unless the trees are the same.
        const int kd_mode = [] { const char* e = getenv("RT_KD_BUILD"); return !e ? 0 : (strcmp(e, "host") == 0 ? 1 : (strcmp(e, "device") == 0 ? 2 : 0)); }();
        const int64_t kd_device_min = [] { const char* e = getenv("RT_KD_DEVICE_MIN"); return e && atoll(e) > 0 ? atoll(e) : (int64_t)500000; }();
        const bool kd_check = [] { const char* e = getenv("RT_CHECK_KD"); return e && atoi(e) != 0; }();
        int64_t built_nodes = 0; int32_t tree_depth = 0;
        bool on_device = false;
        if (out->device >= 0 && kd_mode != 1 && (kd_mode == 2 || kd_check || (int64_t)root.prims.size() >= kd_device_min)) {
            std::vector<RtKdBfsNode> bfs; std::vector<int32_t> leaf_ids;
            const int rcd = rtk_kd_build_device(out->device, d->prims + first, (int32_t)first, cnt, root.prims.data(), (int64_t)root.prims.size(), root.model_scale,
                                                &bfs, &leaf_ids, &tree_depth, err);
            if (rcd < 0) return rcd;
            if (rcd == 0) {
                on_device = true;
                built_nodes = (int64_t)bfs.size();
                std::vector<int32_t> all; all.swap(root.prims);
                const double tt = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
                from_bfs_parallel(root, bfs, leaf_ids, nt);
                if (getenv("RT_BUILD_TIMING")) fprintf(stderr, "KD build: tree from the device's nodes %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - tt);
                if (kd_check) {
                    BuildNode ref; ref.prims.swap(all); ref.model_scale = root.model_scale;
                    Builder bh;
                    bh.prims = d->prims; bh.nt = nt;
                    bh.par_step = nt > 1 ? std::max<size_t>((size_t)1 << 16, ref.prims.size() / (size_t)(4 * nt)) : SIZE_MAX;
                    bh.run(ref);
                    if (bh.built.load() != built_nodes || bh.max_depth.load() != tree_depth || !same_tree(root, ref)) {
                        *err = "RT_CHECK_KD: the device-built KD-tree differs from the host-built one"; return RT_E_INVALID;
                    }
                }
            } else if (kd_mode == 2 && getenv("RT_BUILD_TIMING")) fprintf(stderr, "KD build: the device build declined, building on the host\n");
        }
        if (!on_device) {
            Builder b;
            b.prims = d->prims; b.nt = nt;
            b.par_step = nt > 1 ? std::max<size_t>((size_t)1 << 16, root.prims.size() / (size_t)(4 * nt)) : SIZE_MAX;
            b.run(root);
            built_nodes = b.built.load(); tree_depth = b.max_depth.load();
        }
        out->kd_nodes_built += built_nodes;
        out->kd_max_depth = std::max(out->kd_max_depth, tree_depth);
        if (tree_depth + 1 > 250) { *err = "KD-tree deeper than 250 levels"; return RT_E_LIMIT; }
        const bool timing = getenv("RT_BUILD_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t1 = now();
        Flattener f{d, out, dbg_slack_scale()};
        for (int o = out->host_layouts ? 0 : 7; o < 8; ++o) out->nodes[o].reserve(out->nodes[o].size() + (size_t)built_nodes);
        f.make_leaves(root);
        const double t2 = now();
        out->kd_roots.push_back((int32_t)out->nodes[7].size());
        f.emit_reference(root, 0);
        const double t3 = now();
        if (!out->host_layouts) {                                  // the device derives them (kernels_build.hip)
        } else if (nt > 1 && out->nodes[7].size() > 4096) {        // the other seven layouts: one thread each
            std::vector<std::thread> th;
            for (int o = 0; o < 7; ++o) th.emplace_back([&f, &root, o] { f.emit_octant(root, 0, o); });
            for (std::thread& t : th) t.join();
        } else {
            for (int o = 0; o < 7; ++o) f.emit_octant(root, 0, o);
        }
        if (timing)
            fprintf(stderr, "object %d: %zu nodes: build %.2f s, leaf records %.2f s, reference layout %.2f s, 7 layouts %.2f s\n",
                    (int)o, out->nodes[7].size(), t1 - t0, t2 - t1, t3 - t2, now() - t3);
        if (out->nodes[7].size() > (size_t)INT32_MAX / 2 || out->leaf_tris.size() > (size_t)INT32_MAX / 2) {
            *err = "scene exceeds 2^30 nodes or leaf references"; return RT_E_LIMIT;
        }
    }
    out->prim_normal.resize((size_t)np * 3);
    out->prim_geom.resize((size_t)np * 4);
    {
        auto tables = [d, out](int64_t a, int64_t b) {
            if (b > a) {
                memset(&out->prim_normal[(size_t)a * 3], 0, (size_t)(b - a) * 12);
                memset(&out->prim_geom[(size_t)a * 4], 0, (size_t)(b - a) * 16);
            }
            for (int64_t i = a; i < b; ++i) {
                const float* f = d->prims[i].f;
                if (d->prim_type[i] == RT_PRIM_TRIANGLE) {
                    // Triangle::GetNormal (hw4.cpp:122-137) re-normalises normal[0] in place on
                    // every call; the device uses the value that sequence converges to.
                    float nrm[3] = {f[9], f[10], f[11]};
                    for (int it = 0; it < 16; ++it) {
                        float prev[3] = {nrm[0], nrm[1], nrm[2]};
                        normalize3(nrm);
                        if (prev[0] == nrm[0] && prev[1] == nrm[1] && prev[2] == nrm[2]) break;
                    }
                    memcpy(&out->prim_normal[(size_t)i * 3], nrm, 12);
                } else if (d->prim_type[i] == RT_PRIM_SPHERE) {
                    memcpy(&out->prim_geom[(size_t)i * 4], f, 16);
                }
            }
        };
        const int nt = np < (1 << 16) ? 1 : rt_host_threads();
        if (nt <= 1) tables(0, np);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back(tables, np * t / nt, np * (t + 1) / nt);
            for (std::thread& t : th) t.join();
        }
    }
    for (int64_t i = 0; i < np; ++i) {
        const float* f = d->prims[i].f;
        if (in_tree[(size_t)i]) continue;
        if (d->prim_type[i] != RT_PRIM_TRIANGLE && d->prim_type[i] != RT_PRIM_SPHERE) continue;  // disabled
        DevLoose l{};
        l.prim = (int32_t)i;
        l.flags = ((double)d->materials[d->prim_material[i]].refract > 0.1) ? RT_PF_SKIP_SHADOW : 0;
        if (d->prim_type[i] == RT_PRIM_SPHERE) { l.flags |= RT_PF_SPHERE; memcpy(l.a, f, 16); }
        else fill_tri(l.a, f);
        out->loose.push_back(l);
    }
    return RT_OK;
}
