// trace_device.h -- device-side traversal / intersection routines shared by the gfx950 kernels of
// librt_hip.so (kernels_trace.hip, kernels_emit.hip, kernels_pm.hip).  Everything here is wave-synchronous:
// the traversal functions use wave-wide ballots, so all 64 lanes of a wave must call them together
// (lanes without a ray pass valid = false).
//
// Replaces (paths under the reference root):
//   KDNode::HitTest                 HW4/hw4/hw4.cpp:281-328
//   Triangle::IntersectionTest      HW4/hw4/hw4.cpp:171-231 (non-culling branch)
//   Sphere::IntersectionTest        HW4/hw4/hw4.cpp:73-116
//   closest-hit loop                HW4/hw4/hw4.cpp:808-871, Final/photon/final.cpp:802-816
//   any-hit loop (IsInShadow)       HW4/hw4/hw4.cpp:987-1046, Final/photon/final.cpp:997-1012
#pragma once
#include <hip/hip_runtime.h>
#include "rt_internal.h"

#define RT_ALIVE 0x7fffffff
#define RT_DEAD  (-1)

namespace {

struct Rayf { float ox, oy, oz, dx, dy, dz; };

__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }

// wave-wide vote on a bool: the lane mask of the comparison itself (HIP's __ballot takes an int, so the bool is first
// materialised as 0 / 1 in a vector register and compared again)
__device__ __forceinline__ unsigned long long rt_ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// KDNode::HitTest, hw4.cpp:281-328.  lo/hi are center -/+ extend/2.0f.
__device__ __forceinline__ bool box_test_b(const float* __restrict__ lo, const float* __restrict__ hi, const Rayf& r) {
    float tmin = (lo[0] - r.ox) / r.dx;
    float tmax = (hi[0] - r.ox) / r.dx;
    if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
    float tymin = (lo[1] - r.oy) / r.dy;
    float tymax = (hi[1] - r.oy) / r.dy;
    if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (lo[2] - r.oz) / r.dz;
    float tzmax = (hi[2] - r.oz) / r.dz;
    if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return true;
}
__device__ __forceinline__ bool box_test(const DevNode& n, const Rayf& r) { return box_test_b(n.lo, n.hi, r); }

// Triangle::IntersectionTest, hw4.cpp:171-231.  a = {v0, edge1, edge2}.
// eps_f = smallest float above the reference's double `epsilon` 1e-6, so
// (det > -epsilon && det < epsilon) in double  <=>  |det| < eps_f in float.
__device__ __forceinline__ bool tri_test(const float* __restrict__ a, const Rayf& r, float eps_f,
                                         float& px, float& py, float& pz) {
    const float v0x = a[0], v0y = a[1], v0z = a[2];
    const float e1x = a[3], e1y = a[4], e1z = a[5];
    const float e2x = a[6], e2y = a[7], e2z = a[8];
    // pvec = rayDir ^ edge2
    const float pvx = r.dy * e2z - r.dz * e2y;
    const float pvy = r.dz * e2x - r.dx * e2z;
    const float pvz = r.dx * e2y - r.dy * e2x;
    const float det = e1x * pvx + e1y * pvy + e1z * pvz;
    if (__builtin_fabsf(det) < eps_f) return false;
    const float detInv = 1.0f / det;
    const float tx = r.ox - v0x, ty = r.oy - v0y, tz = r.oz - v0z;
    const float u = (tx * pvx + ty * pvy + tz * pvz) * detInv;
    if (u < 0 || u > 1.0f) return false;
    // qvec = tvec ^ edge1
    const float qx = ty * e1z - tz * e1y;
    const float qy = tz * e1x - tx * e1z;
    const float qz = tx * e1y - ty * e1x;
    const float v = (r.dx * qx + r.dy * qy + r.dz * qz) * detInv;
    if (v < 0 || u + v > 1.0f) return false;
    const float t = (e2x * qx + e2y * qy + e2z * qz) * detInv;
    if (t < 0) return false;
    px = r.ox + t * r.dx; py = r.oy + t * r.dy; pz = r.oz + t * r.dz;
    return true;
}

// Sphere::IntersectionTest, hw4.cpp:73-116.  a = {center, radius}.
__device__ __forceinline__ bool sphere_test(const float* __restrict__ a, const Rayf& r,
                                            float& px, float& py, float& pz) {
    const float hx = a[0] - r.ox, hy = a[1] - r.oy, hz = a[2] - r.oz;
    const float mu = hx * r.dx + hy * r.dy + hz * r.dz;
    const float delta = (mu * mu) - (hx * hx + hy * hy + hz * hz) + (a[3] * a[3]);
    if (delta < 0) return false;
    const float sq = sqrt_rn(delta);
    const float t1 = mu + sq, t2 = mu - sq;
    if (t1 <= 0 && t2 <= 0) return false;
    float t;
    if (t1 >= 0 && t2 < 0) t = t1;
    else if (t1 < 0 && t2 >= 0) t = t2;
    else t = t1 < t2 ? t1 : t2;
    px = r.ox + t * r.dx; py = r.oy + t * r.dy; pz = r.oz + t * r.dz;
    return true;
}

// len = (ori - v).length(), hw4.cpp:851
__device__ __forceinline__ float dist_to(const Rayf& r, float px, float py, float pz) {
    const float ax = r.ox - px, ay = r.oy - py, az = r.oz - pz;
    return sqrt_rn(ax * ax + ay * ay + az * az);
}

// vec3::normalize, algebra3.cpp:280-282,301: d_inv = 1./length (double divide)
__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
    const float len = sqrt_rn(x * x + y * y + z * z);
    const float di = (float)(1.0 / (double)len);
    x *= di; y *= di; z *= di;
}

// culling term of hw4.cpp:853 / final.cpp:808: GetNormal(v) * dir < 0
__device__ __forceinline__ bool facing(const DevSceneView& sc, int prim, bool sphere, const Rayf& r,
                                       float px, float py, float pz) {
    float nx, ny, nz;
    if (sphere) {
        const float* g = sc.prim_geom + 4 * (size_t)prim;
        nx = px - g[0]; ny = py - g[1]; nz = pz - g[2];
        normalize3(nx, ny, nz);
    } else {
        const float* n = sc.prim_normal + 3 * (size_t)prim;
        nx = n[0]; ny = n[1]; nz = n[2];
    }
    return (nx * r.dx + ny * r.dy + nz * r.dz) < 0;
}

struct Closest { float len; int prim; float px, py, pz; int rank; };

// per-lane work counters of the measurement-only kernel k_trace_work (bench.py's algorithmic bytes)
struct Work { unsigned box, tri, exact, fetch; };


// ---------------------------------------------------------------------------
// STRICT traversal: the reference's visit order and nothing else (no culling).
// One wave = 64 rays.  ANY: stop at the first accepted hit per lane.
template <bool ANY, bool COUNT = false>
__device__ __forceinline__ void traverse(const DevSceneView& sc, const Rayf& r, bool valid,
                                         float eps_f, float tmax, bool use_tmax,
                                         Closest& best, bool& occluded, Work* w = nullptr) {
    int fail = valid ? RT_ALIVE : RT_DEAD;
    const int nn = sc.nnodes;
    int i = 0;
    while (i < nn) {
        if (ANY) { if (rt_ballot(fail != RT_DEAD) == 0ull) break; }
        const DevNode nd = sc.nodes[i];                 // wave-uniform address
        const int depth = nd.meta & 0xff;
        if (fail >= depth) fail = RT_ALIVE;             // left the subtree this lane had failed in
        bool hit = false;
        if (COUNT) w->fetch += 1;
        if (fail == RT_ALIVE) {
            if (COUNT) w->box += 1;
            hit = box_test(nd, r);
            if (!hit) fail = depth;
        }
        if (rt_ballot(hit) == 0ull) { i = nd.skip; continue; }
        if (nd.meta & 0x100) {
            for (int k = 0; k < nd.leaf_count; ++k) {
                const DevLeafTri* __restrict__ t = sc.leaf_tris + (nd.leaf_first + k);   // wave-uniform
                if (ANY && use_tmax && (t->flags & RT_PF_SKIP_SHADOW)) continue;
                if (COUNT) w->fetch += 1;
                if (hit) {
                    float px, py, pz;
                    if (COUNT) { w->tri += 1; w->exact += 1; }
                    if (tri_test(t->v0, r, eps_f, px, py, pz)) {
                        const float len = dist_to(r, px, py, pz);
                        if (ANY) {
                            if (len < (use_tmax ? tmax : 1e9f) && len >= sc.tol_f) { occluded = true; fail = RT_DEAD; hit = false; }
                        } else {
                            // the records of a leaf are stored in Morton order, not in the reference's list order:
                            // an equal len goes to the lower visit rank, which is what the reference's strict < keeps
                            if ((len < best.len || (len == best.len && t->rank < best.rank)) && len >= sc.tol_f &&
                                (!sc.culling || facing(sc, t->prim, false, r, px, py, pz))) {
                                best.len = len; best.prim = t->prim; best.px = px; best.py = py; best.pz = pz;
                                best.rank = t->rank;
                            }
                        }
                    }
                }
            }
        }
        i = i + 1;
    }
}

// ---------------------------------------------------------------------------
// FAST traversal: same accepted hits, fewer tests.
//  * the wave walks the layout ordered for its rays' octant (near child first);
//  * a lane skips a subtree that lies wholly behind its origin, or wholly beyond its current best hit by
//    more than the subtree's slack: no hit the reference accepts there can win or tie;
//  * ties (equal len) are resolved by the reference's visit rank (= leaf record index), so the winner is
//    the one the reference's fixed order would have met first;
//  * box booleans come from reciprocal-multiply slabs when the decision is clear of rounding doubt, and
//    from the reference's divisions otherwise; triangle candidates are screened with an approximate
//    reciprocal and confirmed with the reference's exact sequence.
struct RayAux {
    float rx, ry, rz;      // RN(1/d) per axis
    float dlen;            // |dir|
    float rlen;            // 1 / |dir|
    bool plain;            // all direction components finite and non-zero, |dir| in [0.5, 2]: fast paths allowed
};

__device__ __forceinline__ RayAux ray_aux(const Rayf& r) {
    RayAux a;
    a.rx = 1.0f / r.dx; a.ry = 1.0f / r.dy; a.rz = 1.0f / r.dz;
    a.dlen = sqrt_rn(r.dx * r.dx + r.dy * r.dy + r.dz * r.dz);
    a.rlen = 1.0f / a.dlen;
    const float big = 1e18f, small = 1e-18f;
    const float ax = __builtin_fabsf(r.dx), ay = __builtin_fabsf(r.dy), az = __builtin_fabsf(r.dz);
    a.plain = (ax > small && ax < big) && (ay > small && ay < big) && (az > small && az < big) &&
              (a.dlen > 0.5f && a.dlen < 2.0f) &&
              (__builtin_fabsf(r.ox) < big) && (__builtin_fabsf(r.oy) < big) && (__builtin_fabsf(r.oz) < big);
    return a;
}

// How far outside a triangle's true extent the (exact) ray may pass its plane and still be accepted by the
// reference's rounded u / v tests, per unit of `reach` >= |orig - v0| + 0.44 (|e1| + |e2|)  (DESIGN 4.1b):
//   |u_true - u| <= (E_du + u E_det) / |det_true|,  E_du <= 7.5 eps |tvec| |dir| |e2|,  E_det <= 6.6 eps |dir| |e1| |e2|
// (same for v), so the plane point is within 15 eps |dir| gm reach / |det_true| of the triangle, gm = |e1| |e2|.
// |det_true| >= the reference's cut-off (hw4.cpp:209) less E_det, and >= |e1 x e2| |dir . n| bounded through the
// normal cone.  15 eps = 8.9e-7 is applied as 1.2e-6; no bound (a huge value) when neither floor is positive.
__device__ __forceinline__ float lateral_slack(float gm, float area2_min, bool cone, const float* __restrict__ axis,
                                               float cos_t, float sin_t, const Rayf& r, const RayAux& aux, float eps_f) {
    const float dg = aux.dlen * gm;
    float D = eps_f - 4.0e-7f * dg;
    if (cone) {
        const float c = __builtin_fabsf(r.dx * axis[0] + r.dy * axis[1] + r.dz * axis[2]) * aux.rlen;
        const float cc = __builtin_fminf(c, 1.0f);
        // cos(angle + cone) <= |dir . n| / |dir|; the hardware square root is good to 1 ulp of a value <= 1, which the
        // 2e-7 covers
        const float lb = (cc * cos_t - 2e-7f) - __builtin_amdgcn_sqrtf(1.0f - cc * cc) * sin_t;
        D = __builtin_fmaxf(D, area2_min * lb * aux.dlen * 0.999f);
    }
    return D > 1e-12f ? 1.2e-6f * dg * __builtin_amdgcn_rcpf(D) : 1e30f;      // 1.2e-6 leaves a third of room over 15 eps
}

// returns the reference's HitTest boolean; enter/exit are slab parameters good to ~1e-6 relative
__device__ __forceinline__ bool box_fast(const DevNode& n, const Rayf& r, const RayAux& a, float& enter, float& exit) {
    const float x0 = (n.lo[0] - r.ox) * a.rx, x1 = (n.hi[0] - r.ox) * a.rx;
    const float y0 = (n.lo[1] - r.oy) * a.ry, y1 = (n.hi[1] - r.oy) * a.ry;
    const float z0 = (n.lo[2] - r.oz) * a.rz, z1 = (n.hi[2] - r.oz) * a.rz;
    enter = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
    exit  = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
    const float gap = exit - enter;
    // each slab value is within 2^-22 relative of the reference's rounded quotient; 2^-19 leaves 8x room
    const float doubt = 1.9073486e-6f * (__builtin_fabsf(enter) + __builtin_fabsf(exit)) + 1e-30f;
    if (a.plain && __builtin_fabsf(gap) > doubt) return gap > 0.0f;
    return box_test(n, r);
}

template <bool ANY, bool COUNT = false>
__device__ __forceinline__ void traverse_fast(const DevSceneView& sc, const Rayf& r, bool valid,
                                              float eps_f, float tmax, bool use_tmax,
                                              Closest& best, bool& occluded, Work* w = nullptr) {
    int fail = valid ? RT_ALIVE : RT_DEAD;
    const RayAux aux = ray_aux(r);
    // the wave's layout: octant of its first valid ray (any choice is correct, this one is fastest)
    const unsigned long long vm = rt_ballot(valid);
    if (vm == 0ull) return;
    const int oct_l = (r.dx < 0.f ? 1 : 0) | (r.dy < 0.f ? 2 : 0) | (r.dz < 0.f ? 4 : 0);
    const int oct = __builtin_amdgcn_readlane(oct_l, __builtin_ctzll(vm));
    const DevNode* __restrict__ nodes = sc.nodes_oct[oct];
    const float far_limit = ANY ? (use_tmax ? tmax : 1e9f) : 0.f;
    const int nn = sc.nnodes;
    int i = 0;
    while (i < nn) {
        if (ANY) { if (rt_ballot(fail != RT_DEAD) == 0ull) break; }
        const DevNode nd = nodes[i];                    // wave-uniform address
        const int depth = nd.meta & 0xff;
        if (fail >= depth) fail = RT_ALIVE;
        bool hit = false;
        float leaf_el = 0.f, leaf_xl = 0.f;
        if (COUNT) w->fetch += 1;
        if (fail == RT_ALIVE) {
            float enter, exit;
            if (COUNT) w->box += 1;
            hit = box_fast(nd, r, aux, enter, exit);
            if (hit && aux.plain) {
                // lengths along the ray of the slab interval, shrunk/grown by the subtree's slack
                const float el = enter * aux.dlen, xl = exit * aux.dlen;
                leaf_el = __builtin_fabsf(el); leaf_xl = __builtin_fabsf(xl);
                const float slack = nd.margin * (__builtin_fabsf(el) + __builtin_fabsf(xl)) + 1e-6f;
                const float limit = ANY ? far_limit : best.len;
                if (xl + slack < 0.0f || el - slack > limit) hit = false;   // behind the origin / beyond the best hit
            }
            if (!hit) fail = depth;
        }
        if (rt_ballot(hit) == 0ull) { i = nd.skip; continue; }
        if (nd.meta & 0x100) {
            // lateral slack of this leaf for the lane: how far outside a triangle's true extent a ray may pass and
            // still be accepted by the reference's rounded u / v tests, per unit of group factor
            float reach = 0.f;
            if (hit && aux.plain) {
                const float diag = (nd.hi[0] - nd.lo[0]) + (nd.hi[1] - nd.lo[1]) + (nd.hi[2] - nd.lo[2]);
                reach = leaf_el + leaf_xl + 2.0f * diag;
            }
            const int ngroups = (nd.leaf_count + RT_GROUP - 1) / RT_GROUP;
            for (int g = 0; g < ngroups; ++g) {
                const DevLeafGroup* __restrict__ gp = sc.leaf_groups + (nd.group_first + g);   // wave-uniform
                bool ghit = hit;
                if (COUNT) { w->fetch += 1; if (hit) w->box += 1; }
                if (hit && aux.plain) {
                    // lateral slack (lateral_slack above); the group's own extent joins the reach: its triangles may
                    // stick out of the leaf's box
                    const float lat = lateral_slack(gp->gm, gp->area2_min, gp->pad[0] != 0.0f, gp->axis, gp->cos_t, gp->sin_t, r, aux, eps_f);
                    const float sl = (lat + 2e-5f) * (reach + gp->pad[1]) + 1e-6f;
                    const float x0 = (gp->lo[0] - sl - r.ox) * aux.rx, x1 = (gp->hi[0] + sl - r.ox) * aux.rx;
                    const float y0 = (gp->lo[1] - sl - r.oy) * aux.ry, y1 = (gp->hi[1] + sl - r.oy) * aux.ry;
                    const float z0 = (gp->lo[2] - sl - r.oz) * aux.rz, z1 = (gp->hi[2] + sl - r.oz) * aux.rz;
                    const float en = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
                    const float ex = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
                    const float limit = ANY ? far_limit : best.len;
                    // the box is already inflated by sl on every side: the interval test needs no further slack
                    if (ex < en || ex * aux.dlen < 0.0f || en * aux.dlen > limit) ghit = false;
                }
                if (rt_ballot(ghit) == 0ull) continue;
                const int t0 = nd.leaf_first + g * RT_GROUP;
                const int t1 = (t0 + RT_GROUP < nd.leaf_first + nd.leaf_count) ? t0 + RT_GROUP : nd.leaf_first + nd.leaf_count;
                for (int ti = t0; ti < t1; ++ti) {
                    const DevLeafTri* __restrict__ t = sc.leaf_tris + ti;               // wave-uniform
                    if (ANY && use_tmax && (t->flags & RT_PF_SKIP_SHADOW)) continue;
                    // --- screening with an approximate reciprocal (same dot products as the exact test) ---
                    const float e1x = t->e1[0], e1y = t->e1[1], e1z = t->e1[2];
                    const float e2x = t->e2[0], e2y = t->e2[1], e2z = t->e2[2];
                    const float pvx = r.dy * e2z - r.dz * e2y;
                    const float pvy = r.dz * e2x - r.dx * e2z;
                    const float pvz = r.dx * e2y - r.dy * e2x;
                    const float det = e1x * pvx + e1y * pvy + e1z * pvz;
                    if (COUNT) { w->fetch += 1; if (ghit) w->tri += 1; }
                    bool cand = ghit && !(__builtin_fabsf(det) < eps_f);
                    const float ri = __builtin_amdgcn_rcpf(det);
                    const float tx = r.ox - t->v0[0], ty = r.oy - t->v0[1], tz = r.oz - t->v0[2];
                    const float du = tx * pvx + ty * pvy + tz * pvz;
                    const float ua = du * ri;
                    cand = cand && !(ua < -1e-30f) && !(ua > 1.00001f);
                    if (rt_ballot(cand) == 0ull) continue;
                    const float qx = ty * e1z - tz * e1y;
                    const float qy = tz * e1x - tx * e1z;
                    const float qz = tx * e1y - ty * e1x;
                    const float dv = r.dx * qx + r.dy * qy + r.dz * qz;
                    const float dt = e2x * qx + e2y * qy + e2z * qz;
                    const float va = dv * ri, ta = dt * ri;
                    cand = cand && !(va < -1e-30f) && !(ua + va > 1.00002f) && !(ta < -1e-30f);
                    if (rt_ballot(cand) == 0ull) continue;
                    // --- the reference's exact sequence for the survivors (hw4.cpp:211-230) ---
                    if (cand) {
                        if (COUNT) w->exact += 1;
                        const float detInv = 1.0f / det;
                        const float u = du * detInv;
                        if (!(u < 0 || u > 1.0f)) {
                            const float v = dv * detInv;
                            if (!(v < 0 || u + v > 1.0f)) {
                                const float tt = dt * detInv;
                                if (!(tt < 0)) {
                                    const float px = r.ox + tt * r.dx, py = r.oy + tt * r.dy, pz = r.oz + tt * r.dz;
                                    const float len = dist_to(r, px, py, pz);
                                    if (ANY) {
                                        if (len < far_limit && len >= sc.tol_f) { occluded = true; fail = RT_DEAD; hit = false; ghit = false; }
                                    } else {
                                        if ((len < best.len || (len == best.len && t->rank < best.rank)) && len >= sc.tol_f &&
                                            (!sc.culling || facing(sc, t->prim, false, r, px, py, pz))) {
                                            best.len = len; best.prim = t->prim; best.px = px; best.py = py; best.pz = pz;
                                            best.rank = t->rank;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        i = i + 1;
    }
}

// ---------------------------------------------------------------------------
// FAST traversal, PER-LANE form, for rays that are not coherent across a wave (reflection / glossy bounces,
// shadow rays of scattered hit points).  traverse_fast makes the wave walk the union of its 64 rays' paths, which
// is the right trade while the paths overlap; when they do not, the union is ~64 paths long.  Here every lane
// walks its own path through the layout of its own octant: the skip pointers make that a plain loop (no stack,
// no re-activation bookkeeping), records are fetched with per-lane vector loads.  The accept rule, the culling
// inequalities, the group test and the tie-break by visit rank are those of traverse_fast, so a ray gets the
// same answer from either (tests/test_gpu_trace.py runs both against the oracle).
// Loop shape ("while-while"): each lane advances to its next leaf group that survives the box tests, then all
// lanes that have one test its <= RT_GROUP triangles; the wave leaves when no lane has work left.
template <bool ANY>
__device__ __forceinline__ void traverse_lane(const DevSceneView& sc, const Rayf& r, bool valid,
                                              float eps_f, float tmax, bool use_tmax,
                                              Closest& best, bool& occluded) {
    const RayAux aux = ray_aux(r);
    const int oct = (r.dx < 0.f ? 1 : 0) | (r.dy < 0.f ? 2 : 0) | (r.dz < 0.f ? 4 : 0);
    const DevNode* __restrict__ nodes = sc.nodes_oct[0];
#pragma unroll
    for (int o = 1; o < 8; ++o) if (oct == o) nodes = sc.nodes_oct[o];
    const float far_limit = ANY ? (use_tmax ? tmax : 1e9f) : 0.f;
    const int nn = sc.nnodes;
    int i = valid ? 0 : nn;            // node cursor
    int g = 0, gend = 0;               // group cursor inside the current leaf
    int gbase = 0, lf = 0, lend = 0;   // the current leaf's first group / first record / end of records
    float reach = 0.f;
    for (;;) {
        int t0 = 0, t1 = 0;
        while (t0 == t1 && (g < gend || i < nn)) {
            if (g < gend) {
                const DevLeafGroup gp = sc.leaf_groups[g];
                bool ghit = true;
                if (aux.plain) {
                    const float lat = lateral_slack(gp.gm, gp.area2_min, gp.pad[0] != 0.0f, gp.axis, gp.cos_t, gp.sin_t, r, aux, eps_f);
                    const float sl = (lat + 2e-5f) * (reach + gp.pad[1]) + 1e-6f;
                    const float x0 = (gp.lo[0] - sl - r.ox) * aux.rx, x1 = (gp.hi[0] + sl - r.ox) * aux.rx;
                    const float y0 = (gp.lo[1] - sl - r.oy) * aux.ry, y1 = (gp.hi[1] + sl - r.oy) * aux.ry;
                    const float z0 = (gp.lo[2] - sl - r.oz) * aux.rz, z1 = (gp.hi[2] + sl - r.oz) * aux.rz;
                    const float en = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
                    const float ex = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
                    const float limit = ANY ? far_limit : best.len;
                    if (ex < en || ex * aux.dlen < 0.0f || en * aux.dlen > limit) ghit = false;
                }
                if (ghit) {
                    t0 = lf + (g - gbase) * RT_GROUP;
                    t1 = (t0 + RT_GROUP < lend) ? t0 + RT_GROUP : lend;
                }
                g = g + 1;
            } else {
                const DevNode nd = nodes[i];
                float enter, exit;
                bool hit = box_fast(nd, r, aux, enter, exit);
                float lel = 0.f, lxl = 0.f;
                if (hit && aux.plain) {
                    const float el = enter * aux.dlen, xl = exit * aux.dlen;
                    lel = __builtin_fabsf(el); lxl = __builtin_fabsf(xl);
                    const float slack = nd.margin * (lel + lxl) + 1e-6f;
                    const float limit = ANY ? far_limit : best.len;
                    if (xl + slack < 0.0f || el - slack > limit) hit = false;
                }
                if (!hit) { i = nd.skip; }
                else {
                    i = i + 1;
                    if (nd.meta & 0x100) {
                        gbase = nd.group_first; g = gbase; gend = gbase + (nd.leaf_count + RT_GROUP - 1) / RT_GROUP;
                        lf = nd.leaf_first; lend = lf + nd.leaf_count;
                        reach = aux.plain ? lel + lxl + 2.0f * ((nd.hi[0] - nd.lo[0]) + (nd.hi[1] - nd.lo[1]) + (nd.hi[2] - nd.lo[2])) : 0.f;
                    }
                }
            }
        }
        if (rt_ballot(t0 < t1) == 0ull) break;
        for (int ti = t0; ti < t1; ++ti) {
            const DevLeafTri* __restrict__ t = sc.leaf_tris + ti;
            if (ANY && use_tmax && (t->flags & RT_PF_SKIP_SHADOW)) continue;
            const float e1x = t->e1[0], e1y = t->e1[1], e1z = t->e1[2];
            const float e2x = t->e2[0], e2y = t->e2[1], e2z = t->e2[2];
            const float pvx = r.dy * e2z - r.dz * e2y;
            const float pvy = r.dz * e2x - r.dx * e2z;
            const float pvz = r.dx * e2y - r.dy * e2x;
            const float det = e1x * pvx + e1y * pvy + e1z * pvz;
            if (__builtin_fabsf(det) < eps_f) continue;
            const float ri = __builtin_amdgcn_rcpf(det);
            const float tx = r.ox - t->v0[0], ty = r.oy - t->v0[1], tz = r.oz - t->v0[2];
            const float du = tx * pvx + ty * pvy + tz * pvz;
            const float ua = du * ri;
            if ((ua < -1e-30f) || (ua > 1.00001f)) continue;
            const float qx = ty * e1z - tz * e1y;
            const float qy = tz * e1x - tx * e1z;
            const float qz = tx * e1y - ty * e1x;
            const float dv = r.dx * qx + r.dy * qy + r.dz * qz;
            const float dt = e2x * qx + e2y * qy + e2z * qz;
            const float va = dv * ri, ta = dt * ri;
            if ((va < -1e-30f) || (ua + va > 1.00002f) || (ta < -1e-30f)) continue;
            // the reference's exact sequence for the survivors (hw4.cpp:211-230)
            const float detInv = 1.0f / det;
            const float u = du * detInv;
            if (u < 0 || u > 1.0f) continue;
            const float v = dv * detInv;
            if (v < 0 || u + v > 1.0f) continue;
            const float tt = dt * detInv;
            if (tt < 0) continue;
            const float px = r.ox + tt * r.dx, py = r.oy + tt * r.dy, pz = r.oz + tt * r.dz;
            const float len = dist_to(r, px, py, pz);
            if (ANY) {
                if (len < far_limit && len >= sc.tol_f) { occluded = true; i = nn; g = gend; break; }
            } else {
                if ((len < best.len || (len == best.len && t->rank < best.rank)) && len >= sc.tol_f &&
                    (!sc.culling || facing(sc, t->prim, false, r, px, py, pz))) {
                    best.len = len; best.prim = t->prim; best.px = px; best.py = py; best.pz = pz;
                    best.rank = t->rank;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// BVH traversal: the walk of traverse_fast over the build's OWN hierarchy (bvh_build.cpp: every tree
// triangle once, <= RT_GROUP per leaf, boxes that follow the geometry) instead of the reference's KD-tree.
//  * every node is tested like a leaf group of 4.1b: its box inflated by the rounding slack of the triangles below
//    it, skipped when missed, behind the origin or beyond the best hit;
//  * a triangle is tested with the reference's exact sequence (same screening as traverse_fast), so it is accepted
//    or not, and gets its len, exactly as the reference would -- IF the reference tests it for this ray, i.e. if
//    it sits in a reference leaf whose whole chain of boxes passes KDNode::HitTest.  reachable_rank decides that per
//    ACCEPTED hit (one or two per ray), over the reference leaves that hold the triangle, in the reference's visit
//    order.  HitTest's quotients (bound - orig) / dir are monotone in the bound, so a ray with no zero direction
//    component (no NaN quotient) that passes HitTest on a box INSIDE every box of the chain passes every test of the
//    chain: bvh_build.cpp keeps that box per leaf (the chain's boxes are nested up to rounding, so it is the leaf's
//    box give or take an ulp).  The rare ray that misses it walks the chain (traverse_bvh takes no ray with a zero
//    component).  Returned: the visit rank of the first reachable record, which
//    is also the tie-break key among equal len.
// Rays with a zero or non-finite component, or not of unit length (|dir| within 1e-3 of 1), take traverse_fast with
// their whole wave.
__device__ __forceinline__ int reachable_rank(const DevSceneView& sc, int tref, const Rayf& r) {
    const int first = sc.bvh_tri_ref[2 * tref], cnt = sc.bvh_tri_ref[2 * tref + 1];
    for (int j = 0; j < cnt; ++j) {
        const DevRefLeaf rl = sc.bvh_ref_leaf[first + j];
        const float* __restrict__ cb = sc.ref_chain_box + 6 * (int64_t)rl.leaf;
        if (cb[0] == cb[0] && box_test_b(cb, cb + 3, r)) return rl.rank;
        bool ok = true;
        for (int k = rl.leaf; k >= 0 && ok; k = sc.ref_parent[k]) ok = box_test(sc.nodes[k], r);
        if (ok) return rl.rank;
    }
    return 0x7fffffff;
}

// per-ray constants of the BVH walk
typedef float rt_v2f __attribute__((ext_vector_type(2)));
struct BvhConsts {
    float rx, ry, rz;      // 1/d per axis in the MIRRORED frame of the layout walked (axis negated where the layout's bit is set)
    float dlen, rlen;
    float px, py, pz;      // -orig * (1/d) per axis (the same in either frame)
    float d0;              // reach of a node = d0 + its rho >= |orig - v0| + 0.44 (|e1| + |e2|) for every triangle below it
    float sl0;             // the constant part of the inflation
};
__device__ __forceinline__ BvhConsts bvh_consts(const DevSceneView& sc, const Rayf& r, const RayAux& aux, float root_rho, int oct) {
    BvhConsts b;
    b.rx = (oct & 1) ? -aux.rx : aux.rx; b.ry = (oct & 2) ? -aux.ry : aux.ry; b.rz = (oct & 4) ? -aux.rz : aux.rz;
    b.dlen = aux.dlen; b.rlen = aux.rlen;
    const float oabs = __builtin_fabsf(r.ox) + __builtin_fabsf(r.oy) + __builtin_fabsf(r.oz);
    const float cx = sc.bvh_root_centre[0] - r.ox, cy = sc.bvh_root_centre[1] - r.oy, cz = sc.bvh_root_centre[2] - r.oz;
    b.d0 = 1.001f * sqrt_rn(cx * cx + cy * cy + cz * cz) + 1e-6f * oabs + 1e-6f;
    // slabs as one fused multiply-add per plane, (bound -+ sl) * (1/d) - orig * (1/d): against the subtraction first, the
    // plane moves by <= 3 eps (|orig| + |bound|), which the constant part of the inflation covers
    b.px = -(r.ox * aux.rx); b.py = -(r.oy * aux.ry); b.pz = -(r.oz * aux.rz);
    b.sl0 = 1e-6f + 2e-6f * oabs + 1e-6f * (b.d0 + root_rho);
    return b;
}
// does the ray pass the node's box, inflated by the slack of the triangles below it, in front of the origin and not
// beyond `limit`?  The box is stored in the mirrored frame of its layout (rt_internal.h: DevBvhNode::b) and the ray's
// reciprocals are mirrored alike (bvh_consts), so both near planes, both far planes are one packed instruction each.
// UNI: every direction component is positive in that frame (the ray's octant is the layout's): entry = the largest
// near parameter, exit = the smallest far one.  Otherwise the per-axis min / max sort them, as the reference's
// HitTest does.  Same values either way: (-hi - sl) * (-1/d) is (hi + sl) * (1/d) exactly.
template <bool UNI>
__device__ __forceinline__ void bvh_box_interval(const DevBvhNode& nd, const Rayf& r, const BvhConsts& b, float& en, float& ex) {
    float lat = nd.slack;
    if (nd.meta & 0x200) {
        const float c = __builtin_fabsf(r.dx * nd.axis[0] + r.dy * nd.axis[1] + r.dz * nd.axis[2]) * b.rlen;
        const float cc = __builtin_fminf(c, 1.0f);
        const float lb = (cc * nd.cos_t - 2e-7f) - __builtin_amdgcn_sqrtf(1.0f - cc * cc) * nd.sin_t;   // cos(angle + cone) <= |dir . n| / |dir|
        const float tight = __builtin_fminf(lat, nd.cone_k * __builtin_amdgcn_rcpf(lb) + 2e-5f);
        lat = lb > 0.0f ? tight : lat;                                      // a select: no exec masking inside the box test
    }
    const float sl = __builtin_fmaf(lat, b.d0 + nd.rho, b.sl0);
    const rt_v2f s2 = {-sl, sl};
    rt_v2f x = {nd.b[0], nd.b[1]}, y = {nd.b[2], nd.b[3]}, z = {nd.b[4], nd.b[5]};
    x = x + s2; y = y + s2; z = z + s2;
    // (the six multiply-adds stay single: packed, each of rx .. pz would occupy a register PAIR, which spills)
    x.x = __builtin_fmaf(x.x, b.rx, b.px); x.y = __builtin_fmaf(x.y, b.rx, b.px);
    y.x = __builtin_fmaf(y.x, b.ry, b.py); y.y = __builtin_fmaf(y.y, b.ry, b.py);
    z.x = __builtin_fmaf(z.x, b.rz, b.pz); z.y = __builtin_fmaf(z.y, b.rz, b.pz);
    if (UNI) {
        en = __builtin_fmaxf(__builtin_fmaxf(x.x, y.x), z.x);
        ex = __builtin_fminf(__builtin_fminf(x.y, y.y), z.y);
    } else {
        en = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x.x, x.y), __builtin_fminf(y.x, y.y)), __builtin_fminf(z.x, z.y));
        ex = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x.x, x.y), __builtin_fmaxf(y.x, y.y)), __builtin_fmaxf(z.x, z.y));
    }
}
template <bool UNI>
__device__ __forceinline__ bool bvh_box_hit(const DevBvhNode& nd, const Rayf& r, const BvhConsts& b, float limit) {
    float en, ex;
    bvh_box_interval<UNI>(nd, r, b, en, ex);
    return !(ex < en || ex < 0.0f || en * b.dlen > limit);
}
// The same decision as a WAVE MASK (bit = lane), for the wave-synchronous walk: compare results stay in scalar registers,
// where "no lane" is one scalar compare and "these lanes AND those" one scalar and -- no lane predicate is turned into
// a register of 0 / 1 and compared again, no exec masking around the test.  Evaluated for every lane of the wave; the
// caller masks.  Predicates are LLVM's: the U.. ones hold for unordered operands, as the negated tests above do.
#define RT_FCMP_UGE 11
#define RT_FCMP_ULE 13
#define RT_ICMP_NE 33
#define RT_ICMP_SGE 39
template <bool UNI>
__device__ __forceinline__ unsigned long long bvh_box_mask(const DevBvhNode& nd, const Rayf& r, const BvhConsts& b, float limit) {
    float en, ex;
    bvh_box_interval<UNI>(nd, r, b, en, ex);
    // !(ex < en || ex < 0) = !(ex < max(en, 0)) (a NaN en leaves 0, as both tests on it would be false)
    return __builtin_amdgcn_fcmpf(ex, __builtin_fmaxf(en, 0.0f), RT_FCMP_UGE) & __builtin_amdgcn_fcmpf(en * b.dlen, limit, RT_FCMP_ULE);
}
__device__ __forceinline__ bool bvh_fit(const RayAux& aux) { return aux.plain && aux.dlen > 0.999f && aux.dlen < 1.001f; }

template <bool ANY, bool COUNT, bool UNI>
__device__ __forceinline__ void traverse_bvh_walk(const DevSceneView& sc, const Rayf& r, bool valid, const RayAux& aux, int oct,
                                                  float eps_f, float tmax, bool use_tmax,
                                                  Closest& best, bool& occluded, Work* w) {
    int fail = valid ? RT_ALIVE : RT_DEAD;
    const DevBvhNode* __restrict__ nodes = sc.bvh_oct[oct];
    const float far_limit = ANY ? (use_tmax ? tmax : 1e9f) : 0.f;
    const int nn = sc.nbvh;
    const BvhConsts bc = bvh_consts(sc, r, aux, sc.bvh_root_rho, oct);
    int i = 0;
    // Lane sets are 64-bit wave masks in scalar registers (bvh_box_mask); inverse_ballot turns one back into a lane
    // predicate where lanes diverge (the exact test of the few candidates).  All compares that build masks sit where the
    // whole wave is active.
    while (i < nn) {
        if (ANY) { if (__builtin_amdgcn_sicmp(fail, RT_DEAD, RT_ICMP_NE) == 0ull) break; }
        // wave-uniform address; 32-bit byte offsets (the build keeps both tables below 2^32 bytes): one scalar shift or
        // multiply per record instead of a 64-bit multiply-add
        // Fetched as ONE 16-dword vector: field by field, the compiler sinks the cone's loads into the cone branch and splits
        // the rest into five (seven scalar loads and a second wait per node)
        typedef int rt_v16i __attribute__((ext_vector_type(16)));
        const rt_v16i raw = *(const rt_v16i*)((const char*)nodes + ((unsigned)i << 6));
        DevBvhNode nd;
        __builtin_memcpy(&nd, &raw, sizeof nd);
        const int depth = nd.meta & 0xff;
        // the lanes that are at this node: alive, or waiting since a node at least this deep (RT_ALIVE is the largest int)
        const unsigned long long in = __builtin_amdgcn_sicmp(fail, depth, RT_ICMP_SGE);
        unsigned long long hm = in & bvh_box_mask<UNI>(nd, r, bc, ANY ? far_limit : best.len);
        // the lanes at this node hold fail >= depth, the others less: min() parks exactly the former at this depth
        fail = __builtin_amdgcn_inverse_ballot_w64(hm) ? RT_ALIVE : (fail < depth ? fail : depth);
        if (COUNT) { w->fetch += 1; if (__builtin_amdgcn_inverse_ballot_w64(in)) w->box += 1; }
        const bool leaf = (nd.meta & 0x100) != 0;
        if (hm != 0ull && leaf) {
            const int t1 = nd.skip + (nd.meta >> 16);                // a leaf holds at least one triangle
            int ti = nd.skip;
            do {
                const DevLeafTri* __restrict__ t = (const DevLeafTri*)((const char*)sc.bvh_tris + (unsigned)ti * (unsigned)sizeof(DevLeafTri));
                if (ANY && use_tmax && (t->flags & RT_PF_SKIP_SHADOW)) continue;
                const float e1x = t->e1[0], e1y = t->e1[1], e1z = t->e1[2];
                const float e2x = t->e2[0], e2y = t->e2[1], e2z = t->e2[2];
                const float pvx = r.dy * e2z - r.dz * e2y;
                const float pvy = r.dz * e2x - r.dx * e2z;
                const float pvz = r.dx * e2y - r.dy * e2x;
                const float det = e1x * pvx + e1y * pvy + e1z * pvz;
                if (COUNT) { w->fetch += 1; if (__builtin_amdgcn_inverse_ballot_w64(hm)) w->tri += 1; }
                const float ri = __builtin_amdgcn_rcpf(det);
                const float tx = r.ox - t->v0[0], ty = r.oy - t->v0[1], tz = r.oz - t->v0[2];
                const float du = tx * pvx + ty * pvy + tz * pvz;
                const float ua = du * ri;
                // screen with the reciprocal: !(|det| < eps) && !(ua < -1e-30) && !(ua > 1.00001)
                unsigned long long cm = hm & __builtin_amdgcn_fcmpf(__builtin_fabsf(det), eps_f, RT_FCMP_UGE) &
                                        __builtin_amdgcn_fcmpf(ua, -1e-30f, RT_FCMP_UGE) & __builtin_amdgcn_fcmpf(ua, 1.00001f, RT_FCMP_ULE);
                if (cm == 0ull) continue;
                const float qx = ty * e1z - tz * e1y;
                const float qy = tz * e1x - tx * e1z;
                const float qz = tx * e1y - ty * e1x;
                const float dv = r.dx * qx + r.dy * qy + r.dz * qz;
                const float dt = e2x * qx + e2y * qy + e2z * qz;
                const float va = dv * ri, ta = dt * ri;
                cm &= __builtin_amdgcn_fcmpf(va, -1e-30f, RT_FCMP_UGE) & __builtin_amdgcn_fcmpf(ua + va, 1.00002f, RT_FCMP_ULE) &
                      __builtin_amdgcn_fcmpf(ta, -1e-30f, RT_FCMP_UGE);
                if (cm == 0ull) continue;
                if (__builtin_amdgcn_inverse_ballot_w64(cm)) {                       // hw4.cpp:211-230: the same values, tested together
                    if (COUNT) w->exact += 1;
                    const float detInv = 1.0f / det;
                    const float u = du * detInv, v = dv * detInv, tt = dt * detInv;
                    if (!(u < 0 || u > 1.0f) && !(v < 0 || u + v > 1.0f) && !(tt < 0)) {
                        const float px = r.ox + tt * r.dx, py = r.oy + tt * r.dy, pz = r.oz + tt * r.dz;
                        const float len = dist_to(r, px, py, pz);
                        if (ANY) {
                            if (len < far_limit && len >= sc.tol_f && reachable_rank(sc, t->rank, r) != 0x7fffffff) {
                                occluded = true; fail = RT_DEAD;
                            }
                        } else if (len <= best.len && len >= sc.tol_f &&
                                   (!sc.culling || facing(sc, t->prim, false, r, px, py, pz))) {
                            const int rr = reachable_rank(sc, t->rank, r);
                            if (rr != 0x7fffffff && (len < best.len || rr < best.rank)) {
                                best.len = len; best.prim = t->prim; best.px = px; best.py = py; best.pz = pz;
                                best.rank = rr;
                            }
                        }
                    }
                }
                if (ANY) {                                                            // occluded lanes test no more triangles
                    hm &= __builtin_amdgcn_sicmp(fail, RT_DEAD, RT_ICMP_NE);
                    if (hm == 0ull) break;
                }
            } while (++ti < t1);
        }
        i = (hm != 0ull || leaf) ? i + 1 : nd.skip;                  // (any-hit: hm may have emptied in the leaf, where it is i + 1 anyway)
    }
}

template <bool ANY, bool COUNT = false>
__device__ __forceinline__ void traverse_bvh(const DevSceneView& sc, const Rayf& r, bool valid,
                                             float eps_f, float tmax, bool use_tmax,
                                             Closest& best, bool& occluded, Work* w = nullptr) {
    const RayAux aux = ray_aux(r);
    if (rt_ballot(valid && !bvh_fit(aux)) != 0ull || sc.nbvh == 0) { traverse_fast<ANY, COUNT>(sc, r, valid, eps_f, tmax, use_tmax, best, occluded, w); return; }
    const unsigned long long vm = rt_ballot(valid);
    if (vm == 0ull) return;
    // the wave's layout: octant of its first valid ray (any choice is correct, this one is fastest); when every valid ray
    // is of that octant -- a tile of primary rays, shadow rays towards one light -- the box test needs no min / max
    const int oct_l = (r.dx < 0.f ? 1 : 0) | (r.dy < 0.f ? 2 : 0) | (r.dz < 0.f ? 4 : 0);
    const int oct = __builtin_amdgcn_readlane(oct_l, __builtin_ctzll(vm));
    if (rt_ballot(valid && oct_l != oct) == 0ull) traverse_bvh_walk<ANY, COUNT, true>(sc, r, valid, aux, oct, eps_f, tmax, use_tmax, best, occluded, w);
    else traverse_bvh_walk<ANY, COUNT, false>(sc, r, valid, aux, oct, eps_f, tmax, use_tmax, best, occluded, w);
}

// ---------------------------------------------------------------------------
// BVH walk PER LANE, for rays that are not coherent across a wave: the state of one ray's walk, advanced in
// the "while-while" shape of traverse_lane -- to the next leaf whose box the ray passes, then through its triangles.
// The kernels (k_trace_*_bvh_lanes) keep a wave's 64 lanes busy by handing a lane the next ray of the launch as soon as
// its own is finished, so a launch costs the SUM of the rays' walks, not the sum over waves of the longest walk.
// Same tests and the same accept rule as traverse_bvh.
struct BvhLaneWalk {
    Rayf r; BvhConsts bc; const DevBvhNode* nodes;
    int i;                 // node cursor; >= nbvh: the walk is over
    float far_limit;       // any-hit: the distance to stay under
    // single-step walk only: the cursor walks the pre-order range [.., seg_end) and then goes on at node `nxt`
    // (bvh_lane_start_from).  A walk from the root is the one range [0, nbvh) with nxt = nbvh.
    int seg_end, nxt;      // seg_end < 0: the range is the subtree of node i (its end is in the node's record, nxt in sc.bvh_up)
};
__device__ __forceinline__ void bvh_lane_start(const DevSceneView& sc, const Rayf& r, const RayAux& aux, float far_limit, BvhLaneWalk& wk) {
    const int oct = (r.dx < 0.f ? 1 : 0) | (r.dy < 0.f ? 2 : 0) | (r.dz < 0.f ? 4 : 0);
    const DevBvhNode* nodes = sc.bvh_oct[0];
#pragma unroll
    for (int o = 1; o < 8; ++o) if ((oct & sc.lane_layout_mask) == o) nodes = sc.bvh_oct[o];
    wk.r = r; wk.nodes = nodes; wk.i = 0; wk.far_limit = far_limit;
    wk.seg_end = sc.nbvh; wk.nxt = sc.nbvh;
    wk.bc = bvh_consts(sc, r, aux, sc.bvh_root_rho, oct & sc.lane_layout_mask);
}
// The same walk for a ray that STARTS ON triangle `prim` (a shadow ray, a reflected or refracted child): the boxes of the
// ~20 ancestors of that triangle's leaf contain the origin, so a walk from the root spends a step and a 32-byte fetch on
// each of them only to learn that the ray enters.  This one starts at the leaf and climbs: the leaf, its sibling's subtree,
// then the subtree of the sibling of every node on the way up, each walked in pre-order like any range.  Where to go on
// after the subtree of a sibling X -- the sibling of X's parent -- is sc.bvh_up[X], fetched WITH X's record by the load that otherwise
// carries the third 16 bytes of a triangle: a load instruction of its own per climb costs more than the ancestors' steps
// save (measured: 13 % fewer wave steps, 7 % more time).  The nodes visited are those of the walk from the root minus
// the ancestors; the tests and the accept rule are order-independent (the (len, rank) rule), so
// the hit is the same -- and nearer geometry comes first.
// prim < 0, or a primitive outside the hierarchy: the walk from the root.
__device__ __forceinline__ void bvh_lane_start_from(const DevSceneView& sc, const Rayf& r, const RayAux& aux, float far_limit, BvhLaneWalk& wk, int prim) {
    bvh_lane_start(sc, r, aux, far_limit, wk);
    if (sc.prim_start != nullptr && prim >= 0) {
        const int4 at = reinterpret_cast<const int4*>(sc.prim_start)[prim];
        if (at.x >= 0) { wk.i = at.x; wk.seg_end = at.y; wk.nxt = at.z; }
    }
}
// advances the lane (if `on`) to its next leaf; returns its triangle range.  `limit`: the distance beyond which a box is
// of no interest (closest hit: the best len so far; any hit: the far limit); nbox (optional): counts the box tests
__device__ __forceinline__ void bvh_lane_next_leaf_lim(const DevSceneView& sc, BvhLaneWalk& wk, bool on, float limit, int& t0, int& t1,
                                                       unsigned* nbox = nullptr) {
    t0 = t1 = 0;
    const int nn = sc.nbvh;
    while (on && t0 == t1 && wk.i < nn) {
        const DevBvhNode nd = wk.nodes[wk.i];
        const bool leaf = (nd.meta & 0x100) != 0;
        const bool hit = bvh_box_hit<false>(nd, wk.r, wk.bc, limit);
        if (nbox) *nbox += 1;
        if (!hit) wk.i = leaf ? wk.i + 1 : nd.skip;
        else {
            wk.i = wk.i + 1;
            if (leaf) { t0 = nd.skip; t1 = nd.skip + (nd.meta >> 16); }
        }
    }
}
template <bool ANY>
__device__ __forceinline__ void bvh_lane_next_leaf(const DevSceneView& sc, BvhLaneWalk& wk, bool on, const Closest& best, int& t0, int& t1) {
    bvh_lane_next_leaf_lim(sc, wk, on, ANY ? wk.far_limit : best.len, t0, t1);
}
// the same step over the compact nodes (rt_internal.h: DevBvhNode32; needs sc.bvh_c32): two 16-byte loads per node
__device__ __forceinline__ void bvh_lane_next_leaf_c32_lim(const DevSceneView& sc, BvhLaneWalk& wk, bool on, float limit, int& t0, int& t1,
                                                           unsigned* nbox = nullptr) {
    t0 = t1 = 0;
    const int nn = sc.nbvh;
    const BvhConsts& b = wk.bc;
    while (on && t0 == t1 && wk.i < nn) {
        const float4* __restrict__ np = reinterpret_cast<const float4*>(sc.bvh_c32) + 2 * (size_t)wk.i;
        const float4 q0 = np[0], q1 = np[1];
        const unsigned link = (unsigned)__float_as_int(q1.z), sq = (unsigned)__float_as_int(q1.w);
        const float slack = __int_as_float((int)((sq >> 18) << 17)), srho = __int_as_float((int)(((sq >> 4) & 0x3fffu) << 17));
        const float sl = __builtin_fmaf(slack, b.d0, srho) + b.sl0;
        const rt_v2f s2 = {-sl, sl};
        rt_v2f x = {q0.x, q0.y}, y = {q0.z, q0.w}, z = {q1.x, q1.y};
        x = x + s2; y = y + s2; z = z + s2;
        x.x = __builtin_fmaf(x.x, b.rx, b.px); x.y = __builtin_fmaf(x.y, b.rx, b.px);
        y.x = __builtin_fmaf(y.x, b.ry, b.py); y.y = __builtin_fmaf(y.y, b.ry, b.py);
        z.x = __builtin_fmaf(z.x, b.rz, b.pz); z.y = __builtin_fmaf(z.y, b.rz, b.pz);
        const float en = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x.x, x.y), __builtin_fminf(y.x, y.y)), __builtin_fminf(z.x, z.y));
        const float ex = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x.x, x.y), __builtin_fmaxf(y.x, y.y)), __builtin_fmaxf(z.x, z.y));
        const bool hit = !(ex < en || ex < 0.0f || en * b.dlen > limit);
        const bool leaf = (link >> 31) != 0;
        if (nbox) *nbox += 1;
        if (!hit) wk.i = leaf ? wk.i + 1 : (int)link;
        else {
            wk.i = wk.i + 1;
            if (leaf) { t0 = (int)(link & 0x7fffffffu); t1 = t0 + (int)((sq >> 1) & 7u); }
        }
    }
}
template <bool ANY>
__device__ __forceinline__ void bvh_lane_next_leaf_c32(const DevSceneView& sc, BvhLaneWalk& wk, bool on, const Closest& best, int& t0, int& t1) {
    bvh_lane_next_leaf_c32_lim(sc, wk, on, ANY ? wk.far_limit : best.len, t0, t1);
}
// the triangles [t0, t1) of the lane's leaf; any-hit: returns true (and ends the walk) when one occludes
// (`any` is a compile-time constant in bvh_lane_leaf<ANY>, the lane's mode in the path kernel of kernels_paths.hip;
// ntri (optional): counts the triangle records read)
__device__ __forceinline__ bool bvh_lane_leaf_any(const DevSceneView& sc, BvhLaneWalk& wk, int t0, int t1, float eps_f, bool use_tmax,
                                                  const bool ANY, Closest& best, unsigned* ntri = nullptr) {
    const Rayf& r = wk.r;
    for (int ti = t0; ti < t1; ++ti) {
        const DevLeafTri* __restrict__ t = sc.bvh_tris + ti;
        if (ANY && use_tmax && (t->flags & RT_PF_SKIP_SHADOW)) continue;
        if (ntri) *ntri += 1;
        const float e1x = t->e1[0], e1y = t->e1[1], e1z = t->e1[2];
        const float e2x = t->e2[0], e2y = t->e2[1], e2z = t->e2[2];
        const float pvx = r.dy * e2z - r.dz * e2y;
        const float pvy = r.dz * e2x - r.dx * e2z;
        const float pvz = r.dx * e2y - r.dy * e2x;
        const float det = e1x * pvx + e1y * pvy + e1z * pvz;
        if (__builtin_fabsf(det) < eps_f) continue;
        const float ri = __builtin_amdgcn_rcpf(det);
        const float tx = r.ox - t->v0[0], ty = r.oy - t->v0[1], tz = r.oz - t->v0[2];
        const float du = tx * pvx + ty * pvy + tz * pvz;
        const float ua = du * ri;
        if ((ua < -1e-30f) || (ua > 1.00001f)) continue;
        const float qx = ty * e1z - tz * e1y;
        const float qy = tz * e1x - tx * e1z;
        const float qz = tx * e1y - ty * e1x;
        const float dv = r.dx * qx + r.dy * qy + r.dz * qz;
        const float dt = e2x * qx + e2y * qy + e2z * qz;
        const float va = dv * ri, ta = dt * ri;
        if ((va < -1e-30f) || (ua + va > 1.00002f) || (ta < -1e-30f)) continue;
        const float detInv = 1.0f / det;                                     // hw4.cpp:211-230, exactly
        const float u = du * detInv;
        if (u < 0 || u > 1.0f) continue;
        const float v = dv * detInv;
        if (v < 0 || u + v > 1.0f) continue;
        const float tt = dt * detInv;
        if (tt < 0) continue;
        const float px = r.ox + tt * r.dx, py = r.oy + tt * r.dy, pz = r.oz + tt * r.dz;
        const float len = dist_to(r, px, py, pz);
        if (ANY) {
            if (len < wk.far_limit && len >= sc.tol_f && reachable_rank(sc, t->rank, r) != 0x7fffffff) { wk.i = sc.nbvh; return true; }
        } else if (len <= best.len && len >= sc.tol_f && (!sc.culling || facing(sc, t->prim, false, r, px, py, pz))) {
            const int rr = reachable_rank(sc, t->rank, r);
            if (rr != 0x7fffffff && (len < best.len || rr < best.rank)) {
                best.len = len; best.prim = t->prim; best.px = px; best.py = py; best.pz = pz;
                best.rank = rr;
            }
        }
    }
    return false;
}
template <bool ANY>
__device__ __forceinline__ bool bvh_lane_leaf(const DevSceneView& sc, BvhLaneWalk& wk, int t0, int t1, float eps_f, bool use_tmax, Closest& best) {
    return bvh_lane_leaf_any(sc, wk, t0, t1, eps_f, use_tmax, ANY, best);
}

// ---------------------------------------------------------------------------
// The per-lane walk ONE RECORD AT A TIME (round 3; DESIGN 4.1c "single-step walk").  The while-while shape above makes a
// wave wait for its slowest lane twice over: a lane that has found its next leaf idles until the last lane of the wave
// has found one too (the mean lane needs 24 node steps to its next leaf, the slowest of 64 about 100: measured 13 of 64
// lanes active per vector instruction), and every such wave step is one dependent fetch -- the kernel is bound by the
// latency of a fetch that only a fifth of its lanes have issued.  Here every lane with a ray fetches exactly ONE record
// per wave step, whichever its own walk needs next -- a node, a triangle of its current leaf, or the reachability
// record of a candidate hit -- through one address select and one set of loads, so a wave step carries up to 64 fetches
// and a ray advances at every step.  The records a ray sees, their order, every test and the accept rule are those of
// bvh_lane_next_leaf_c32 + bvh_lane_leaf_any: the same hit, bit for bit.
//   A candidate (a triangle that passed the exact test) waits in the cursor until the caller lets the pending lanes
// run the reachability test together (`do_acc`): reachable_rank's first probe is one 32-byte record (DevTriReach, made
// on the device by k_tri_reach from the tables reachable_rank reads); the rare ray that misses it takes the original
// function.  best / the far limit do not change while a candidate waits, so deferring it changes nothing.
struct LaneCursor {
    int ti, t1;            // the triangles of the current leaf still to test: [ti, t1)
    int acc;               // >= 0: a candidate hit waits for its reachability test (the triangle's index into bvh_tri_reach)
    float clen, cpx, cpy, cpz; int cprim;
};
__device__ __forceinline__ void lane_cursor_reset(LaneCursor& cu) { cu.ti = cu.t1 = 0; cu.acc = -1; cu.clen = cu.cpx = cu.cpy = cu.cpz = 0.f; cu.cprim = -1; }
__device__ __forceinline__ bool lane_walk_busy(const BvhLaneWalk& wk, const LaneCursor& cu, int nn) { return wk.i < nn || cu.ti < cu.t1 || cu.acc >= 0; }
__device__ __forceinline__ void lane_walk_end(BvhLaneWalk& wk, LaneCursor& cu, int nn) { wk.i = nn; cu.ti = cu.t1 = 0; cu.acc = -1; }

// one step for every lane that is `on` (a lane whose walk is over must pass on = false).  any: the lane's walk is an
// any-hit walk (returns with occ = true and the walk ended when a reachable hit below wk.far_limit is found).
__device__ __forceinline__ void bvh_lane_step(const DevSceneView& sc, BvhLaneWalk& wk, LaneCursor& cu, bool on, const bool any, float eps_f,
                                              bool use_tmax, bool do_acc, Closest& best, bool& occ,
                                              unsigned* nbox = nullptr, unsigned* ntri = nullptr) {
    const int nn = sc.nbvh;
    const bool k_acc = on && cu.acc >= 0;
    const bool k_tri = on && cu.acc < 0 && cu.ti < cu.t1;
    const bool k_node = on && cu.acc < 0 && cu.ti >= cu.t1 && wk.i < nn;
    // one address, one set of loads: 32 bytes of a node or a reachability record, 48 of a triangle
    const char* p = (const char*)sc.bvh_c32 + (k_node ? ((size_t)(unsigned)wk.i << 5) : (size_t)0);
    if (k_tri) p = (const char*)sc.bvh_tris + (size_t)(unsigned)cu.ti * sizeof(DevLeafTri);
    if (k_acc) p = (const char*)sc.bvh_tri_reach + ((size_t)(unsigned)cu.acc << 5);
    const bool k_accgo = k_acc && do_acc;
    float4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f}, q2 = {0.f, 0.f, 0.f, 0.f};
    if (k_node || k_tri || k_accgo) { q0 = reinterpret_cast<const float4*>(p)[0]; q1 = reinterpret_cast<const float4*>(p)[1]; }   // (a lane without a fetch asks the texture path for nothing)
    const bool k_root = k_node && wk.seg_end < 0;                            // the first node of a range: where to go on after it
    if (k_tri || k_root) {
        const char* p2 = k_root ? (const char*)(sc.bvh_up + wk.i) : p + 32;
        q2 = *reinterpret_cast<const float4*>(p2);                           // (bvh_up: 4-byte aligned, three entries too many)
    }
#ifdef RT_EXP_EXTRA_LOAD
    if (k_node) { const float4 qx = reinterpret_cast<const float4*>(p)[1]; asm volatile("" :: "v"(qx.x), "v"(qx.y), "v"(qx.z), "v"(qx.w)); }
#endif
    const Rayf& r = wk.r;
    if (k_node) {                                                           // bvh_lane_next_leaf_c32_lim, one trip
        const BvhConsts& b = wk.bc;
        const float limit = any ? wk.far_limit : best.len;
        const unsigned link = (unsigned)__float_as_int(q1.z), sq = (unsigned)__float_as_int(q1.w);
        const float slack = __int_as_float((int)((sq >> 18) << 17)), srho = __int_as_float((int)(((sq >> 4) & 0x3fffu) << 17));
        const float sl = __builtin_fmaf(slack, b.d0, srho) + b.sl0;
        const rt_v2f s2 = {-sl, sl};
        rt_v2f x = {q0.x, q0.y}, y = {q0.z, q0.w}, z = {q1.x, q1.y};
        x = x + s2; y = y + s2; z = z + s2;
        x.x = __builtin_fmaf(x.x, b.rx, b.px); x.y = __builtin_fmaf(x.y, b.rx, b.px);
        y.x = __builtin_fmaf(y.x, b.ry, b.py); y.y = __builtin_fmaf(y.y, b.ry, b.py);
        z.x = __builtin_fmaf(z.x, b.rz, b.pz); z.y = __builtin_fmaf(z.y, b.rz, b.pz);
        const float en = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x.x, x.y), __builtin_fminf(y.x, y.y)), __builtin_fminf(z.x, z.y));
        const float ex = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x.x, x.y), __builtin_fmaxf(y.x, y.y)), __builtin_fmaxf(z.x, z.y));
        const bool hit = !(ex < en || ex < 0.0f || en * b.dlen > limit);
        const bool leaf = (link >> 31) != 0;
        if (nbox) *nbox += 1;
        const int se = k_root ? (leaf ? wk.i + 1 : (int)link) : wk.seg_end;
        if (k_root) wk.nxt = __float_as_int(q2.x);
        int ni = (hit || leaf) ? wk.i + 1 : (int)link;
        const bool climb = ni >= se;                                        // the range is done: on to the sibling of its root's parent
        wk.i = climb ? wk.nxt : ni; wk.seg_end = climb ? -1 : se;
        if (hit && leaf) { cu.ti = (int)(link & 0x7fffffffu); cu.t1 = cu.ti + (int)((sq >> 1) & 7u); }
    }
    if (k_tri) {                                                            // bvh_lane_leaf_any, one triangle
        cu.ti += 1;
        const int flags = __float_as_int(q2.z);
        do {
            if (any && use_tmax && (flags & RT_PF_SKIP_SHADOW)) break;
            if (ntri) *ntri += 1;
            const float e1x = q0.w, e1y = q1.x, e1z = q1.y;
            const float e2x = q1.z, e2y = q1.w, e2z = q2.x;
            const float pvx = r.dy * e2z - r.dz * e2y;
            const float pvy = r.dz * e2x - r.dx * e2z;
            const float pvz = r.dx * e2y - r.dy * e2x;
            const float det = e1x * pvx + e1y * pvy + e1z * pvz;
            if (__builtin_fabsf(det) < eps_f) break;
            const float ri = __builtin_amdgcn_rcpf(det);
            const float tx = r.ox - q0.x, ty = r.oy - q0.y, tz = r.oz - q0.z;
            const float du = tx * pvx + ty * pvy + tz * pvz;
            const float ua = du * ri;
            if ((ua < -1e-30f) || (ua > 1.00001f)) break;
            const float qx = ty * e1z - tz * e1y;
            const float qy = tz * e1x - tx * e1z;
            const float qz = tx * e1y - ty * e1x;
            const float dv = r.dx * qx + r.dy * qy + r.dz * qz;
            const float dt = e2x * qx + e2y * qy + e2z * qz;
            const float va = dv * ri, ta = dt * ri;
            if ((va < -1e-30f) || (ua + va > 1.00002f) || (ta < -1e-30f)) break;
            const float detInv = 1.0f / det;                                     // hw4.cpp:211-230, exactly
            const float u = du * detInv;
            if (u < 0 || u > 1.0f) break;
            const float v = dv * detInv;
            if (v < 0 || u + v > 1.0f) break;
            const float tt = dt * detInv;
            if (tt < 0) break;
            const float px = r.ox + tt * r.dx, py = r.oy + tt * r.dy, pz = r.oz + tt * r.dz;
            const float len = dist_to(r, px, py, pz);
            const int prim = __float_as_int(q2.y);
            if (any) { if (!(len < wk.far_limit && len >= sc.tol_f)) break; }
            else if (!(len <= best.len && len >= sc.tol_f && (!sc.culling || facing(sc, prim, false, r, px, py, pz)))) break;
            cu.acc = __float_as_int(q2.w); cu.clen = len; cu.cpx = px; cu.cpy = py; cu.cpz = pz; cu.cprim = prim;
        } while (false);
    }
    if (k_accgo) {                                                          // reachable_rank, first probe from one record
        const float lo[3] = {q0.x, q0.y, q0.z}, hi[3] = {q0.w, q1.x, q1.y};
        int rr = __float_as_int(q1.z);
        if (!(lo[0] == lo[0] && box_test_b(lo, hi, r))) rr = reachable_rank(sc, cu.acc, r);
        cu.acc = -1;
        if (rr != 0x7fffffff) {
            if (any) { occ = true; lane_walk_end(wk, cu, nn); }
            else if (cu.clen < best.len || rr < best.rank) {
                best.len = cu.clen; best.prim = cu.cprim; best.px = cu.cpx; best.py = cu.cpy; best.pz = cu.cpz; best.rank = rr;
            }
        }
    }
}

// A wave's pool of claimed rays [next, end) for the refilling per-lane kernels: ONE atomic per chunk, not per refill -- a
// device-scope atomic on a word that thousands of waves share is the slowest thing such a kernel does (measured: with one
// atomic per refill of 8 lanes the single-step kernels lost 20 % to it).  The launch's rays are eight contiguous ranges
// with a counter each (sc.ray_counter); a wave starts in the range of its workgroup's XCD label and moves on when its
// own is used up, so the rays one L2 serves at a time come from one part of the queue.  Chunks shrink as a range runs out
// (guided: about half a fair share of what was left at the wave's last claim, between 16 and sc.lane_chunk), so no wave
// sits on a long private queue while the others are done.  All members are wave-uniform.
struct RayPool { int64_t next, end, left_seen; int cur, spent; };
__device__ __forceinline__ void ray_pool_init(RayPool& p, int64_t n) {
    p.next = p.end = 0; p.left_seen = n >> 3; p.cur = (int)(blockIdx.x & 7u); p.spent = 0;
}
// makes the pool non-empty; false: every ray of the launch has been claimed
__device__ __forceinline__ bool ray_pool_fill(const DevSceneView& sc, RayPool& p, int64_t n, int lane) {
    while (p.next >= p.end) {
        if (p.spent >= 8) return false;
        const int64_t lo = (n * p.cur) >> 3, hi = (n * (p.cur + 1)) >> 3;
        const int64_t share = (int64_t)(gridDim.x >> 2 ? gridDim.x >> 2 : 1);       // 1/8 of the waves start in a range: half a fair share each
        int64_t want = p.left_seen / share;
        want = want < 16 ? 16 : (want > sc.lane_chunk ? sc.lane_chunk : want);
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(sc.ray_counter + p.cur, (unsigned long long)want);
        base = __shfl(base, 0, 64);
        if ((int64_t)base >= hi - lo) { p.cur = (p.cur + 1) & 7; ++p.spent; p.left_seen = n >> 3; }
        else {
            p.next = lo + (int64_t)base;
            p.end = p.next + want < hi ? p.next + want : hi;
            p.left_seen = hi - p.end;
        }
    }
    return true;
}
// the lanes of `need` take the pool's next rays in lane order; returns this lane's ray or -1 (not asking, or the pool ran out)
__device__ __forceinline__ int64_t ray_pool_take(RayPool& p, unsigned long long need, int lane) {
    const int64_t claim = p.next + __builtin_popcountll(need & ((1ull << lane) - 1ull));
    const bool mine = ((need >> lane) & 1ull) != 0ull && claim < p.end;
    const int64_t took = p.next + __builtin_popcountll(need);
    p.next = took < p.end ? took : p.end;
    return mine ? claim : (int64_t)-1;
}

// primitives outside every tree, in index order (the whole scene for final.cpp)
template <bool ANY, bool COUNT = false>
__device__ __forceinline__ void loose_prims(const DevSceneView& sc, const Rayf& r, bool valid,
                                            float eps_f, float tmax, bool use_tmax,
                                            Closest& best, bool& occluded, Work* w = nullptr) {
    bool live = valid && !(ANY && occluded);
    for (int k = 0; k < sc.nloose; ++k) {
        if (ANY) { if (rt_ballot(live) == 0ull) break; }
        const DevLoose* __restrict__ l = sc.loose + k;                              // wave-uniform
        const int flags = l->flags;
        if (ANY && use_tmax && (flags & RT_PF_SKIP_SHADOW)) continue;
        if (COUNT) w->fetch += 1;
        if (!live) continue;
        if (COUNT) { w->tri += 1; w->exact += 1; }
        float px, py, pz;
        const bool sph = (flags & RT_PF_SPHERE) != 0;
        const bool b = sph ? sphere_test(l->a, r, px, py, pz) : tri_test(l->a, r, eps_f, px, py, pz);
        if (!b) continue;
        const float len = dist_to(r, px, py, pz);
        if (ANY) {
            if (len < (use_tmax ? tmax : 1e9f) && len >= sc.tol_f) { occluded = true; live = false; }
        } else {
            // loose primitives come after every tree in the reference's order: strict < keeps an equal-len tree hit
            if (len < best.len && len >= sc.tol_f && (!sc.culling || facing(sc, l->prim, sph, r, px, py, pz))) {
                best.len = len; best.prim = l->prim; best.px = px; best.py = py; best.pz = pz;
            }
        }
    }
}

__device__ __forceinline__ Rayf load_ray(const rt_ray* __restrict__ rays, int64_t i) {
    const float* p = reinterpret_cast<const float*>(rays + i);
    Rayf r; r.ox = p[0]; r.oy = p[1]; r.oz = p[2]; r.dx = p[3]; r.dy = p[4]; r.dz = p[5];
    return r;
}

// Camera::GetPixelCenter + normalize, hw4.cpp:268-276, 1053-1056.
__device__ __forceinline__ void primary_ray(const rt_camera& cam, float fx, float fy, float* o6) {
    const float wx = cam.corners[1][0] - cam.corners[0][0], wy = cam.corners[1][1] - cam.corners[0][1],
                wz = cam.corners[1][2] - cam.corners[0][2];
    const float hx = cam.corners[0][0] - cam.corners[2][0], hy = cam.corners[0][1] - cam.corners[2][1],
                hz = cam.corners[0][2] - cam.corners[2][2];
    const float pixelWidth = sqrt_rn(wx * wx + wy * wy + wz * wz) / ((float)cam.resx);
    const float pixelHeight = sqrt_rn(hx * hx + hy * hy + hz * hz) / ((float)cam.resy);
    const float a = 0.5f * pixelWidth, b = 0.5f * pixelHeight, c = fx * pixelWidth, d = fy * pixelHeight;
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        p[k] = (((cam.corners[0][k] + a * cam.xaxis[k]) + b * cam.yaxis[k]) + c * cam.xaxis[k]) + d * cam.yaxis[k];
    float dx = p[0] - cam.center[0], dy = p[1] - cam.center[1], dz = p[2] - cam.center[2];
    normalize3(dx, dy, dz);
    o6[0] = cam.center[0]; o6[1] = cam.center[1]; o6[2] = cam.center[2];
    o6[3] = dx; o6[4] = dy; o6[5] = dz;
}

}  // namespace
