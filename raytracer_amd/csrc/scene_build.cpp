// scene_build.cpp -- host side of rt_scene_create: builds the per-object
// KD-trees by the reference rule (KDNode::CalcBound / BuildKD,
// HW4/hw4/hw4.cpp:330-490) and flattens them into the stackless device layout
// described in rt_internal.h.  Runs once per scene on the host; the result is
// uploaded to HBM by api.cpp.
#include "rt_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

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
    int32_t leaf_first = 0;
    float margin = 0.f;
};

struct Builder {
    const rt_prim* prims;
    int64_t built = 0;
    int32_t max_depth = 0;

    // CalcBound, hw4.cpp:330-389
    void bound(BuildNode& n) const {
        float mn[3] = {1e9f, 1e9f, 1e9f}, mx[3] = {-1e9f, -1e9f, -1e9f};
        for (int32_t id : n.prims) {
            const float* f = prims[id].f;
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    float v = f[3 * j + k];
                    if (v < mn[k]) mn[k] = v;
                    if (v > mx[k]) mx[k] = v;
                }
        }
        const float half = inv_via_double(2.0f);           // operator/(vec3, 2.0f)
        for (int k = 0; k < 3; ++k) {
            n.center[k] = (mn[k] + mx[k]) * half;
            n.extend[k] = mx[k] - mn[k];
        }
        float best = -1e9f;
        for (int k = 0; k < 3; ++k)
            if (n.extend[k] > best) { best = n.extend[k]; n.axis = k; }
    }

    // BuildKD, hw4.cpp:393-490.  The reference sorts every vertex coordinate on
    // the split axis and takes element size/2; nth_element yields the same value.
    void build(BuildNode& n, int depth) {
        if (depth > max_depth) max_depth = depth;
        ++built;
        bound(n);
        const int ax = n.axis;
        std::vector<float> coord;
        coord.reserve(n.prims.size() * 3);
        for (int32_t id : n.prims) {
            const float* f = prims[id].f;
            coord.push_back(f[ax]); coord.push_back(f[3 + ax]); coord.push_back(f[6 + ax]);
        }
        const size_t mid = coord.size() / 2;
        std::nth_element(coord.begin(), coord.begin() + (ptrdiff_t)mid, coord.end());
        const float split = coord[mid];
        std::vector<float>().swap(coord);

        n.left.reset(new BuildNode());
        n.right.reset(new BuildNode());
        size_t both = 0;
        for (int32_t id : n.prims) {
            const float* f = prims[id].f;
            const bool r = f[ax] >= split || f[3 + ax] >= split || f[6 + ax] >= split;
            const bool l = f[ax] < split || f[3 + ax] < split || f[6 + ax] < split;
            if (r) n.right->prims.push_back(id);
            if (l) n.left->prims.push_back(id);
            both += (l && r);
        }
        const bool subdivide = (float)both < (float)n.left->prims.size() / n.model_scale &&
                               (float)both < (float)n.right->prims.size() / n.model_scale;
        if (subdivide) {
            std::vector<int32_t>().swap(n.prims);       // interior lists are never read by traversal
            build(*n.left, depth + 1);
            build(*n.right, depth + 1);
        } else {
            n.leaf = true;                              // hw4.cpp:484: this node tests its own list
            n.left.reset(); n.right.reset();            // its children are never visited (hw4.cpp:827)
        }
    }
};

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

    // Reference order (layout 7): pre-order with the RIGHT child first, the order in which the reference
    // pops nodes (hw4.cpp:831-836).  Emits the leaf records (their index is the reference's visit rank)
    // and computes the culling slack of every subtree.
    int32_t emit_reference(BuildNode& n, int depth) {
        std::vector<DevNode>& nodes = out->nodes[7];
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        float margin = 0.f;
        if (n.leaf) {
            n.leaf_first = (int32_t)out->leaf_tris.size();
            for (int32_t id : n.prims) {
                DevLeafTri t{};
                float a[9]; fill_tri(a, d->prims[id].f);
                memcpy(t.v0, a, 12); memcpy(t.e1, a + 3, 12); memcpy(t.e2, a + 6, 12);
                t.prim = id;
                t.flags = ((double)d->materials[d->prim_material[id]].refract > 0.1) ? RT_PF_SKIP_SHADOW : 0;
                out->leaf_tris.push_back(t);
                // |edge1|*|edge2| bounds the cancellation in det and in the t numerator
                const float l1 = std::sqrt(a[3] * a[3] + a[4] * a[4] + a[5] * a[5]);
                const float l2 = std::sqrt(a[6] * a[6] + a[7] * a[7] + a[8] * a[8]);
                margin = std::max(margin, l1 * l2);
            }
        } else {
            emit_reference(*n.right, depth + 1);
            emit_reference(*n.left, depth + 1);
            margin = std::max(n.left->margin, n.right->margin);
        }
        n.margin = margin;
        fill(nodes[(size_t)me], n, depth);
        nodes[(size_t)me].skip = (int32_t)nodes.size();
        return me;
    }

    // Layout o: the child on the side the ray comes from goes first (bit a of o: direction negative on axis a,
    // so the larger-coordinate child, `right`, is nearer).
    void emit_octant(const BuildNode& n, int depth, int o) {
        std::vector<DevNode>& nodes = out->nodes[o];
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
        r.meta = depth | (n.leaf ? 0x100 : 0);
        // 4x safety on the bound, plus a floor that covers the rounding of the slab arithmetic itself
        r.margin = 4.0f * n.margin + 1e-4f;
        r.leaf_first = n.leaf ? n.leaf_first : 0;
        r.leaf_count = n.leaf ? (int32_t)n.prims.size() : 0;
        r.pad = 0;
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
        Builder b{d->prims};
        b.build(root, 0);
        out->kd_nodes_built += b.built;
        out->kd_max_depth = std::max(out->kd_max_depth, b.max_depth);
        if (b.max_depth + 1 > 250) { *err = "KD-tree deeper than 250 levels"; return RT_E_LIMIT; }
        Flattener f{d, out};
        f.emit_reference(root, 0);
        for (int o = 0; o < 7; ++o) f.emit_octant(root, 0, o);
        if (out->nodes[7].size() > (size_t)INT32_MAX / 2 || out->leaf_tris.size() > (size_t)INT32_MAX / 2) {
            *err = "scene exceeds 2^30 nodes or leaf references"; return RT_E_LIMIT;
        }
    }
    out->prim_normal.assign((size_t)np * 3, 0.f);
    out->prim_geom.assign((size_t)np * 4, 0.f);
    for (int64_t i = 0; i < np; ++i) {
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
