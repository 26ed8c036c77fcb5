// kernels_trace.hip -- gfx950 traversal / intersection kernels of librt_hip.so.
//
// Replaces (paths under the reference root):
//   KDNode::HitTest                 HW4/hw4/hw4.cpp:281-328
//   Triangle::IntersectionTest      HW4/hw4/hw4.cpp:171-231 (non-culling branch)
//   Sphere::IntersectionTest        HW4/hw4/hw4.cpp:73-116
//   closest-hit loop                HW4/hw4/hw4.cpp:808-871, Final/photon/final.cpp:802-816
//   any-hit loop (IsInShadow)       HW4/hw4/hw4.cpp:987-1046, Final/photon/final.cpp:997-1012
//   GetPixelCenter + normalize      HW4/hw4/hw4.cpp:268-276,1053-1056
//
// Design (MI355X): the reference traversal is EXHAUSTIVE and its visit order
// does not depend on the ray (pre-order, right child first, objects in file
// order; a ray only filters that order through its box tests).  So one
// 64-lane wavefront walks the node array front to back ONCE for its 64 rays:
// the node / triangle records are wave-uniform (scalar-path loads, one fetch
// per wave instead of 64), a wave-wide ballot decides whether a subtree is
// entered, and each lane keeps a single integer (the depth at which it left
// the current path) instead of a stack.  Per-lane results are bit-identical to
// the reference because every lane performs exactly the reference's float
// operations in the reference's order on the nodes the reference would visit.
//
// Compile with -ffp-contract=off and correctly rounded f32 divide / sqrt.
#include <atomic>
#include <cstdlib>
#include <hipcub/hipcub.hpp>
#include "trace_device.h"
#include "rng_device.h"

// ---------------------------------------------------------------------------
// active: optional per-ray int32 (ray skipped when < 0), used by the renderer
extern "C" __global__ void __launch_bounds__(256)
k_trace_closest(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                int64_t n, float eps_f, int fast, rt_hit* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = i < n;
    const bool valid = inb && (active == nullptr || active[i] >= 0);
    Rayf r = {0, 0, 0, 0, 0, 1};
    if (valid) r = load_ray(rays, i);
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool occ = false;
    if (fast) traverse_fast<false>(sc, r, valid, eps_f, 0.f, false, best, occ);
    else traverse<false>(sc, r, valid, eps_f, 0.f, false, best, occ);
    loose_prims<false>(sc, r, valid, eps_f, 0.f, false, best, occ);
    if (inb) {
        float* o = reinterpret_cast<float*>(out + i);
        reinterpret_cast<int32_t*>(o)[0] = best.prim;
        o[1] = best.len; o[2] = best.px; o[3] = best.py; o[4] = best.pz;
    }
}

extern "C" __global__ void __launch_bounds__(256)
k_trace_any(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
            const float* __restrict__ tmax, int64_t n, float eps_f, int fast, uint8_t* __restrict__ occluded) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = i < n;
    const bool valid = inb && (active == nullptr || active[i] >= 0);
    Rayf r = {0, 0, 0, 0, 0, 1};
    float tm = 0.f;
    if (valid) { r = load_ray(rays, i); if (tmax) tm = tmax[i]; }
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool occ = false;
    if (fast) traverse_fast<true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ);
    else traverse<true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ);
    loose_prims<true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ);
    if (inb) occluded[i] = occ ? 1 : 0;
}

// per-lane traversal (trace_device.h: traverse_lane) for incoherent rays; one wave per block so a finished wave frees
// its slot at once
extern "C" __global__ void __launch_bounds__(64)
k_trace_closest_lane(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                     int64_t n, float eps_f, int unfit_only, rt_hit* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool inb = i < n;
    bool valid = inb && (active == nullptr || active[i] >= 0);
    Rayf r = {0, 0, 0, 0, 0, 1};
    if (valid) r = load_ray(rays, i);
    if (unfit_only) { valid = valid && !bvh_fit(ray_aux(r)); inb = valid; }    // the rest was k_trace_closest_bvh_lanes' 
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool occ = false;
    traverse_lane<false>(sc, r, valid, eps_f, 0.f, false, best, occ);
    loose_prims<false>(sc, r, valid, eps_f, 0.f, false, best, occ);
    if (inb) {
        float* o = reinterpret_cast<float*>(out + i);
        reinterpret_cast<int32_t*>(o)[0] = best.prim;
        o[1] = best.len; o[2] = best.px; o[3] = best.py; o[4] = best.pz;
    }
}

extern "C" __global__ void __launch_bounds__(64)
k_trace_any_lane(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                 const float* __restrict__ tmax, int64_t n, float eps_f, int unfit_only, uint8_t* __restrict__ occluded) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool inb = i < n;
    bool valid = inb && (active == nullptr || active[i] >= 0);
    Rayf r = {0, 0, 0, 0, 0, 1};
    float tm = 0.f;
    if (valid) { r = load_ray(rays, i); if (tmax) tm = tmax[i]; }
    if (unfit_only) { valid = valid && !bvh_fit(ray_aux(r)); inb = valid; }    // the rest was k_trace_any_bvh_lanes' 
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool occ = false;
    traverse_lane<true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ);
    loose_prims<true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ);
    if (inb) occluded[i] = occ ? 1 : 0;
}

// Per-lane walk of the build's own hierarchy with lanes refilled from the launch's rays (trace_device.h: BvhLaneWalk):
// a fixed number of waves; a lane whose ray is finished writes its result and takes the next unclaimed ray.  Rays that
// traverse_bvh does not take (zero / non-finite components, not unit length) are left for k_trace_*_lane; inactive
// rays get the "no hit" result here.  Needs nbvh > 0 (the launchers check).
template <bool ANY>
__device__ __forceinline__ void bvh_lanes(const DevSceneView& sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                                          const float* __restrict__ tmax, int64_t n, float eps_f,
                                          rt_hit* __restrict__ out, uint8_t* __restrict__ occluded, const int32_t* __restrict__ order) {
    const int lane = threadIdx.x & 63;
    const bool use_tmax = tmax != nullptr;
    bool have = false, more = true, occ = false, fresh = false;
    int64_t idx = 0;
    BvhLaneWalk wk; wk.i = 0; wk.far_limit = 0.f; wk.nodes = sc.bvh_oct[0];
    wk.r = Rayf{0, 0, 0, 0, 0, 1};
    wk.bc = BvhConsts{1.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    // Claims: the launch's rays are eight contiguous ranges with a counter each; a wave starts in the range of its
    // workgroup's label (blockIdx & 7: workgroups with one label share an XCD, hence an L2) and moves on to the next range
    // when its own is used up, so the rays an L2 serves at one time come from one part of the queue (= of the image) while
    // no wave runs out of work before the launch does.  A label, not a promise: any placement gives the same results.
    int cur = (int)(blockIdx.x & 7u), spent = 0;
    for (;;) {
        const unsigned long long need = rt_ballot(!have && more);
        if (need != 0ull) {
            const int first = __builtin_ctzll(need);
            const int64_t lo = (n * cur) >> 3, hi = (n * (cur + 1)) >> 3;
            unsigned long long base = 0;
            if (lane == first) base = atomicAdd(sc.ray_counter + cur, (unsigned long long)__builtin_popcountll(need));
            base = __shfl(base, first, 64);
            if ((int64_t)base >= hi - lo) {                         // this range is used up (wave-uniform)
                cur = (cur + 1) & 7;
                if (++spent == 8) more = false;
            }
            if (!have && more) {
                // `order` (rtk_ray_order): the k-th claim takes ray order[k], so that the rays a wave holds at one time start
                // in one region with one octant; results still go to the ray's own slot
                const int64_t claim = lo + (int64_t)base + __builtin_popcountll(need & ((1ull << lane) - 1ull));
                const int64_t my = (claim < hi && order != nullptr) ? (int64_t)order[claim] : claim;
                if (claim >= hi) { }                                // beyond the range: claim again next time round
                else if (active != nullptr && active[my] < 0) {
                    if (ANY) occluded[my] = 0;
                    else {
                        float* o = reinterpret_cast<float*>(out + my);
                        reinterpret_cast<int32_t*>(o)[0] = -1; o[1] = 1e9f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f;
                    }
                } else {
                    const Rayf r = load_ray(rays, my);
                    const RayAux aux = ray_aux(r);
                    if (bvh_fit(aux)) {
                        idx = my; have = true; occ = false; fresh = true;
                        best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
                        bvh_lane_start(sc, r, aux, ANY ? (use_tmax ? tmax[my] : 1e9f) : 0.f, wk);
                    }
                }
            }
            // primitives outside the trees (spheres), for the rays just taken.  The reference tests them after the trees
            // and keeps an equal-len tree hit; testing them first gives the same answer because a tree hit of equal len
            // still replaces them (their rank stays 0x7fffffff), and their len already bounds the walk.
            if (sc.nloose > 0 && rt_ballot(fresh) != 0ull) {
                loose_prims<ANY>(sc, wk.r, fresh, eps_f, wk.far_limit, use_tmax, best, occ);
                if (ANY && fresh && occ) wk.i = sc.nbvh;             // occluded already: nothing to walk
                fresh = false;
            }
        }
        if (rt_ballot(have) == 0ull) { if (rt_ballot(more) == 0ull) break; continue; }
        int t0, t1;
        if (sc.bvh_c32) bvh_lane_next_leaf_c32<ANY>(sc, wk, have, best, t0, t1);
        else bvh_lane_next_leaf<ANY>(sc, wk, have, best, t0, t1);
        if (have && t0 < t1) { if (bvh_lane_leaf<ANY>(sc, wk, t0, t1, eps_f, use_tmax, best)) occ = true; }
        if (have && wk.i >= sc.nbvh) {
            if (ANY) occluded[idx] = occ ? 1 : 0;
            else {
                float* o = reinterpret_cast<float*>(out + idx);
                reinterpret_cast<int32_t*>(o)[0] = best.prim;
                o[1] = best.len; o[2] = best.px; o[3] = best.py; o[4] = best.pz;
            }
            have = false;
        }
    }
}

extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_trace_closest_bvh_lanes(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                          int64_t n, float eps_f, rt_hit* __restrict__ out, const int32_t* __restrict__ order) {
    bvh_lanes<false>(sc, rays, active, nullptr, n, eps_f, out, nullptr, order);
}

extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_trace_any_bvh_lanes(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                      const float* __restrict__ tmax, int64_t n, float eps_f, uint8_t* __restrict__ occluded, const int32_t* __restrict__ order) {
    bvh_lanes<true>(sc, rays, active, tmax, n, eps_f, nullptr, occluded, order);
}

// The same launch shape with the walk taken ONE RECORD PER WAVE STEP (trace_device.h: bvh_lane_step): every lane that
// holds a ray fetches the record its own walk needs next, so no lane waits for another lane's way to its next leaf.
// A finished lane writes its result at once; idle lanes claim together when sc.lane_refill_batch of them wait (a claim
// is an atomic, a ray fetch and the per-ray constants: worth sharing), pending candidates take their reachability test
// when sc.lane_acc_batch wait -- or, both, when nothing else can move.  Same rays, same results as bvh_lanes.
template <bool ANY>
__device__ __forceinline__ void bvh_lanes_step(const DevSceneView& sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                                               const float* __restrict__ tmax, int64_t n, float eps_f,
                                               rt_hit* __restrict__ out, uint8_t* __restrict__ occluded, const int32_t* __restrict__ order,
                                               const RaySrc src) {
    const int lane = threadIdx.x & 63;
    const int nn = sc.nbvh;
    const bool use_tmax = tmax != nullptr;
    bool have = false, more = true, occ = false, fresh = false;
    int64_t idx = 0;
    BvhLaneWalk wk; wk.i = nn; wk.far_limit = 0.f; wk.nodes = sc.bvh_oct[0];
    wk.r = Rayf{0, 0, 0, 0, 0, 1};
    wk.bc = BvhConsts{1.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    LaneCursor cu; lane_cursor_reset(cu);
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    RayPool pool; ray_pool_init(pool, n);
    for (;;) {
        if (have && !lane_walk_busy(wk, cu, nn)) {                  // this lane's walk is over: its result, and the lane is free
            if (ANY) occluded[idx] = occ ? 1 : 0;
            else {
                float* o = reinterpret_cast<float*>(out + idx);
                reinterpret_cast<int32_t*>(o)[0] = best.prim;
                o[1] = best.len; o[2] = best.px; o[3] = best.py; o[4] = best.pz;
            }
            have = false;
        }
        const unsigned long long need = rt_ballot(!have && more);
        unsigned long long busy = rt_ballot(have);
        if (need != 0ull && (busy == 0ull || __builtin_popcountll(need) >= sc.lane_refill_batch)) {
            if (!ray_pool_fill(sc, pool, n, lane)) more = false;   // (trace_device.h: RayPool -- one atomic per chunk of rays)
            if (more) {
                const int64_t claim = ray_pool_take(pool, need, lane);
                const int64_t my = (claim >= 0 && order != nullptr) ? (int64_t)order[claim] : claim;
                if (claim < 0) { }                                  // not asking, or the pool ran out: claim again next time round
                else if (active != nullptr && active[my] < 0) {
                    if (ANY) occluded[my] = 0;
                    else {
                        float* o = reinterpret_cast<float*>(out + my);
                        reinterpret_cast<int32_t*>(o)[0] = -1; o[1] = 1e9f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f;
                    }
                } else {
                    const Rayf r = load_ray(rays, my);
                    const RayAux aux = ray_aux(r);
                    if (bvh_fit(aux)) {
                        idx = my; have = true; occ = false; fresh = true;
                        best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
                        const int from = src.base ? *reinterpret_cast<const int32_t*>(src.base + (my / src.div) * (int64_t)src.stride) : -1;
                        bvh_lane_start_from(sc, r, aux, ANY ? (use_tmax ? tmax[my] : 1e9f) : 0.f, wk, from);
                        lane_cursor_reset(cu);
                    }
                }
            }
            if (sc.nloose > 0 && rt_ballot(fresh) != 0ull) {        // primitives outside the trees, for the rays just taken (bvh_lanes says why first)
                loose_prims<ANY>(sc, wk.r, fresh, eps_f, wk.far_limit, use_tmax, best, occ);
                if (ANY && fresh && occ) lane_walk_end(wk, cu, nn);
            }
            fresh = false;
            busy = rt_ballot(have);
        }
        if (busy == 0ull) { if (rt_ballot(more) == 0ull) break; continue; }
        const unsigned long long pend = rt_ballot(have && cu.acc >= 0);
        const unsigned long long stepping = rt_ballot(have && cu.acc < 0 && lane_walk_busy(wk, cu, nn));
        const bool do_acc = pend != 0ull && (stepping == 0ull || __builtin_popcountll(pend) >= sc.lane_acc_batch);
        bvh_lane_step(sc, wk, cu, have && lane_walk_busy(wk, cu, nn), ANY, eps_f, use_tmax, do_acc, best, occ);
    }
}

extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))
k_trace_closest_bvh_step(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                         int64_t n, float eps_f, rt_hit* __restrict__ out, const int32_t* __restrict__ order, RaySrc src) {
    bvh_lanes_step<false>(sc, rays, active, nullptr, n, eps_f, out, nullptr, order, src);
}

extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))
k_trace_any_bvh_step(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                     const float* __restrict__ tmax, int64_t n, float eps_f, uint8_t* __restrict__ occluded, const int32_t* __restrict__ order,
                     RaySrc src) {
    bvh_lanes_step<true>(sc, rays, active, tmax, n, eps_f, nullptr, occluded, order, src);
}

// the build's own hierarchy instead of the reference's tree (trace_device.h: traverse_bvh)
extern "C" __global__ void __launch_bounds__(256)
k_trace_closest_bvh(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                    int64_t n, float eps_f, rt_hit* __restrict__ out, const int32_t* __restrict__ order) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = i0 < n;
    const int64_t i = (inb && order) ? (int64_t)order[i0] : i0;      // `order` (rtk_ray_order): the wave's rays by direction octant
    const bool valid = inb && (active == nullptr || active[i] >= 0);
    Rayf r = {0, 0, 0, 0, 0, 1};
    if (valid) r = load_ray(rays, i);
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool occ = false;
    traverse_bvh<false>(sc, r, valid, eps_f, 0.f, false, best, occ);
    loose_prims<false>(sc, r, valid, eps_f, 0.f, false, best, occ);
    if (inb) {
        float* o = reinterpret_cast<float*>(out + i);
        reinterpret_cast<int32_t*>(o)[0] = best.prim;
        o[1] = best.len; o[2] = best.px; o[3] = best.py; o[4] = best.pz;
    }
}

extern "C" __global__ void __launch_bounds__(256)
k_trace_any_bvh(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
                const float* __restrict__ tmax, int64_t n, float eps_f, uint8_t* __restrict__ occluded, const int32_t* __restrict__ order) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = i0 < n;
    const int64_t i = (inb && order) ? (int64_t)order[i0] : i0;
    const bool valid = inb && (active == nullptr || active[i] >= 0);
    Rayf r = {0, 0, 0, 0, 0, 1};
    float tm = 0.f;
    if (valid) { r = load_ray(rays, i); if (tmax) tm = tmax[i]; }
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool occ = false;
    traverse_bvh<true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ);
    loose_prims<true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ);
    if (inb) occluded[i] = occ ? 1 : 0;
}

// Measurement only (bench.py, tests): the same traversal with per-lane work counters.  out4[0] box tests,
// [1] triangle records read by a lane that was inside the leaf (screening or exact), [2] exact
// reference-sequence triangle tests, [3] node / triangle record fetches summed over lanes (a wave-uniform
// fetch counts 64).  One vector atomic per wave and counter at the end; no results are written.
extern "C" __global__ void __launch_bounds__(256)
k_trace_work(DevSceneView sc, const rt_ray* __restrict__ rays, const int32_t* __restrict__ active,
             const float* __restrict__ tmax, int64_t n, float eps_f, int fast, int any,
             unsigned long long* __restrict__ out4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = i < n;
    const bool valid = inb && (active == nullptr || active[i] >= 0);
    Rayf r = {0, 0, 0, 0, 0, 1};
    float tm = 0.f;
    if (valid) { r = load_ray(rays, i); if (tmax) tm = tmax[i]; }
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool occ = false;
    Work w = {0, 0, 0, 0};
    if (any) {
        if (fast == 3) traverse_bvh<true, true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ, &w);
        else if (fast) traverse_fast<true, true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ, &w);
        else traverse<true, true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ, &w);
        loose_prims<true, true>(sc, r, valid, eps_f, tm, tmax != nullptr, best, occ, &w);
    } else {
        if (fast == 3) traverse_bvh<false, true>(sc, r, valid, eps_f, 0.f, false, best, occ, &w);
        else if (fast) traverse_fast<false, true>(sc, r, valid, eps_f, 0.f, false, best, occ, &w);
        else traverse<false, true>(sc, r, valid, eps_f, 0.f, false, best, occ, &w);
        loose_prims<false, true>(sc, r, valid, eps_f, 0.f, false, best, occ, &w);
    }
    unsigned v[4] = {w.box, w.tri, w.exact, valid ? w.fetch : 0u};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        unsigned long long x = v[c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(out4 + c, x);
    }
}

// rays in the reference's loop order: index = x*resy + y (hw4.cpp:1096-1098)
extern "C" __global__ void __launch_bounds__(256)
k_generate_linear(rt_camera cam, int64_t first, int64_t n, float ox, float oy, rt_ray* __restrict__ rays) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t g = first + i;
    const int x = (int)(g / cam.resy), y = (int)(g % cam.resy);
    primary_ray(cam, (float)x + ox, (float)y + oy, reinterpret_cast<float*>(rays + i));
}

// rays in wave-tile order: ray i belongs to micro-tile tiles[i/64] (8x8 pixels,
// micro-tiles numbered row-major over ceil(resx/8) x ceil(resy/8)), lane i%64 is
// pixel (lane%8, lane/8) of it.  pix[i] = x + y*resx, or -1 outside the frame.
extern "C" __global__ void __launch_bounds__(256)
k_generate_tiled(rt_camera cam, const int32_t* __restrict__ tiles, int64_t ntiles, float ox, float oy,
                 rt_ray* __restrict__ rays, int32_t* __restrict__ pix) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntiles * 64) return;
    const int lane = (int)(i & 63);
    const int t = tiles ? tiles[i >> 6] : (int)(i >> 6);
    const int tw = (cam.resx + 7) >> 3;
    const int x = (t % tw) * 8 + (lane & 7), y = (t / tw) * 8 + (lane >> 3);
    float* o = reinterpret_cast<float*>(rays + i);
    if (x >= cam.resx || y >= cam.resy) {
        pix[i] = -1; o[0] = o[1] = o[2] = o[3] = o[4] = 0.f; o[5] = 1.f;
        return;
    }
    pix[i] = x + y * cam.resx;
    primary_ray(cam, (float)x + ox, (float)y + oy, o);
}

// Distributed ray tracing (build-defined): spp jittered samples per pixel, thin lens.  Ray index
// i = (micro-tile slot * 64 + lane) * spp + s, so the samples of a pixel are contiguous (one run for k_accumulate)
// and a wave still covers a compact patch of pixels.  pixlist[slot*64+lane] = the pixel (or -1), for zero/finalize.
extern "C" __global__ void __launch_bounds__(256)
k_generate_drt(rt_camera cam, const int32_t* __restrict__ tiles, int64_t ntiles, int spp, uint64_t seed,
               float lens_radius, float focal_distance, rt_ray* __restrict__ rays, int32_t* __restrict__ pix,
               float* __restrict__ w, uint32_t* __restrict__ code, int32_t* __restrict__ pixlist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntiles * 64 * spp) return;
    const int64_t slot = i / spp;
    const int sidx = (int)(i - slot * spp);
    const int lane = (int)(slot & 63);
    const int t = tiles ? tiles[slot >> 6] : (int)(slot >> 6);
    const int tw = (cam.resx + 7) >> 3;
    const int x = (t % tw) * 8 + (lane & 7), y = (t / tw) * 8 + (lane >> 3);
    float* o = reinterpret_cast<float*>(rays + i);
    w[i] = (float)(1.0 / (double)spp);
    code[i] = ((uint32_t)sidx << 16) | 1u;
    if (x >= cam.resx || y >= cam.resy) {
        pix[i] = -1; if (sidx == 0) pixlist[slot] = -1;
        o[0] = o[1] = o[2] = o[3] = o[4] = 0.f; o[5] = 1.f;
        return;
    }
    const int p = x + y * cam.resx;
    pix[i] = p; if (sidx == 0) pixlist[slot] = p;
    const uint64_t key = ((uint64_t)p << 16) | (uint64_t)sidx;
    float jx = 0.f, jy = 0.f;
    if (spp > 1) { jx = rt_rng(seed, key, 0, 0) - 0.5f; jy = rt_rng(seed, key, 0, 1) - 0.5f; }
    primary_ray(cam, (float)x + jx, (float)y + jy, o);
    if (lens_radius > 0.f) {
        const float lr = lens_radius * sqrt_rn(rt_rng(seed, key, 0, 2));
        const double phi = RT_TWO_PI_D * (double)rt_rng(seed, key, 0, 3);
        const float cx = lr * (float)cos(phi), cy = lr * (float)sin(phi);
        const float fx = o[0] + focal_distance * o[3], fy = o[1] + focal_distance * o[4], fz = o[2] + focal_distance * o[5];
        o[0] = (o[0] + cx * cam.xaxis[0]) + cy * cam.yaxis[0];
        o[1] = (o[1] + cx * cam.xaxis[1]) + cy * cam.yaxis[1];
        o[2] = (o[2] + cx * cam.xaxis[2]) + cy * cam.yaxis[2];
        float dx = fx - o[0], dy = fy - o[1], dz = fz - o[2];
        normalize3(dx, dy, dz);
        o[3] = dx; o[4] = dy; o[5] = dz;
    }
}

// ---------------------------------------------------------------------------
// Ray reordering for the per-lane kernels (north_star: "ray compaction and sorting to limit divergence").  Rays of the
// deeper recursion levels arrive in pixel order, but their origins are scattered over the scene and their directions over
// all octants.  A 30-bit key per ray -- the 32^3 cell of its origin, then the octant of its direction, then 12 more bits of
// the origin's Morton code -- and one radix sort of (key, ray index) pairs give the order in which the refilling kernels
// hand rays to lanes; nothing is moved, every result is written to the ray's own slot, so the ordered per-pixel sums
// of k_accumulate (and with them the byte-identical frames under tiling and chunking) are untouched.
__device__ __forceinline__ unsigned ro_spread(unsigned v) {      // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}
struct RoBox { float lo[3], inv[3]; };
extern "C" __global__ void __launch_bounds__(256)
k_ray_keys(const rt_ray* __restrict__ rays, const int32_t* __restrict__ active, int64_t n, RoBox box, int octant_only,
           unsigned* __restrict__ keys, int32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned key = 0x3fffffffu;                                   // inactive slots last
    if (active == nullptr || active[i] >= 0) {
        const float* r = reinterpret_cast<const float*>(rays + i);
        unsigned m = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float u = (r[a] - box.lo[a]) * box.inv[a];
            u = u > 0.f ? (u < 1.f ? u : 1.f) : 0.f;              // also sends NaN to 0
            m |= ro_spread((unsigned)(u * 511.0f)) << a;          // 27-bit Morton code, 9 bits per axis
        }
        const unsigned oct = (r[3] < 0.f ? 1u : 0u) | (r[4] < 0.f ? 2u : 0u) | (r[5] < 0.f ? 4u : 0u);
        key = ((m >> 12) << 15) | (oct << 12) | (m & 0xfffu);
        if (octant_only) key = oct;
    } else if (octant_only) key = 8u;
    keys[i] = key; vals[i] = (int32_t)i;
}

namespace {
struct RoWork { unsigned *keys_in, *keys_out; int32_t *vals_in, *vals_out; void* temp; size_t temp_bytes, total; };
RoWork ro_carve(void* base, int64_t n) {
    RoWork W;
    size_t t1 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (const unsigned*)nullptr, (unsigned*)nullptr, (const int32_t*)nullptr,
                                             (int32_t*)nullptr, (int)n, 0, 30, (hipStream_t)nullptr);
    char* w = (char*)base;
    auto take = [&w](size_t bytes) { char* p = w; w += (bytes + 255) & ~(size_t)255; return (void*)p; };
    W.keys_in = (unsigned*)take((size_t)n * 4); W.keys_out = (unsigned*)take((size_t)n * 4);
    W.vals_in = (int32_t*)take((size_t)n * 4); W.vals_out = (int32_t*)take((size_t)n * 4);
    W.temp_bytes = t1; W.temp = take(t1);
    W.total = (size_t)(w - (char*)base) + 256;
    return W;
}
}  // namespace

size_t rtk_ray_order_workspace(int64_t n) { return n > 0 && n <= 0x7fffffff ? ro_carve(nullptr, n).total : 0; }

// *order_out = n ray indices in traversal order (inside d_work), or NULL when n is too small to be worth a sort.
// what = RT_ORDER_OCTANT: a stable partition by direction octant alone (one 4-bit pass; inactive slots last), RT_ORDER_CELL:
// by the 32^3 cell of the origin alone (the key's top 15 bits) -- for the wave-synchronous kernels (render.cpp)
hipError_t rtk_ray_order(const rt_ray* d_rays, const int32_t* d_active, int64_t n, const float lo[3], const float hi[3],
                         void* d_work, const int32_t** order_out, hipStream_t st, int what) {
    *order_out = nullptr;
    if (n < 4096 || n > 0x7fffffff || !d_work) return hipSuccess;
    const RoWork W = ro_carve(d_work, n);
    RoBox box;
    for (int a = 0; a < 3; ++a) { box.lo[a] = lo[a]; box.inv[a] = hi[a] > lo[a] ? 1.0f / (hi[a] - lo[a]) : 0.f; }
    hipLaunchKernelGGL(k_ray_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_rays, d_active, n, box, what == RT_ORDER_OCTANT ? 1 : 0, W.keys_in, W.vals_in);
    size_t temp = W.temp_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(W.temp, temp, W.keys_in, W.keys_out, W.vals_in, W.vals_out, (int)n, what == RT_ORDER_CELL ? 15 : 0, what == RT_ORDER_OCTANT ? 4 : 30, st);
    if (e != hipSuccess) return e;
    *order_out = W.vals_out;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// host launchers (declared in rt_internal.h)
static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }
// the refilling per-lane BVH kernels: RT_LANE_BVH=0 keeps the reference-tree walk for every ray
static inline int bvh_lanes_usable(const DevSceneView& sc) {
    static const int on = [] { const char* e = getenv("RT_LANE_BVH"); return e ? atoi(e) : 1; }();
    return (on && sc.nbvh > 0 && sc.ray_counter != nullptr) ? 1 : 0;
}
// waves of a refilling kernel: enough to fill the chip several times over, never more than the rays need
// the launch's own claim counter: the next word of the scene's ring, zeroed on the launch's stream
hipError_t rtk_next_ray_counter(DevSceneView& sc, hipStream_t st) {
    static std::atomic<unsigned> seq{0};
    sc.ray_counter = sc.ray_counter + 8 * (size_t)(seq.fetch_add(1) % RT_RAY_COUNTERS);
    return hipMemsetAsync(sc.ray_counter, 0, 8 * sizeof(unsigned long long), st);       // the eight range counters of bvh_lanes
}
static inline hipError_t next_ray_counter(DevSceneView& sc, hipStream_t st) { return rtk_next_ray_counter(sc, st); }
// waves of a refilling kernel: never more than the rays need.  Fewer resident walks than the chip can hold are faster --
// every walk in flight is a stream of scattered line requests and a share of L2 (1 M-triangle mesh, 1080p x 64 spp:
// 8 192 waves 116.2 ms per frame, 4 096 waves 113.0); a frame that runs two such kernels side by side (render.cpp) asks
// for 3 072 each through sc.lane_waves.  RT_LANE_WAVES overrides both.
static inline unsigned lanes_grid(const DevSceneView& sc, int64_t n) {
    static const int64_t forced = [] { const char* e = getenv("RT_LANE_WAVES"); return e && atoll(e) > 0 ? atoll(e) : (int64_t)0; }();
    const int64_t cap = forced ? forced : sc.lane_waves > 0 ? (int64_t)sc.lane_waves : (int64_t)4096;
    const int64_t w = (n + 63) / 64;
    return (unsigned)(w < cap ? w : cap);
}

hipError_t rtk_trace_closest(const DevSceneView& sc, const rt_ray* d_rays, const int32_t* d_active, int64_t n,
                             float eps_f, int fast, rt_hit* d_out, hipStream_t st, const int32_t* d_order, const RaySrc* src) {
    if (n <= 0) return hipSuccess;
    const RaySrc from = (src && src->base && src->div > 0) ? *src : RaySrc{nullptr, 0, 1};
    if (fast == 3) {
        hipLaunchKernelGGL(k_trace_closest_bvh, dim3(grid_for(n, 256)), dim3(256), 0, st, sc, d_rays, d_active, n, eps_f, d_out, d_order);
        return hipGetLastError();
    }
    if (fast == 2) {
        const int lanes = bvh_lanes_usable(sc);
        if (lanes) {
            DevSceneView scl = sc;
            hipError_t e = next_ray_counter(scl, st);
            if (e != hipSuccess) return e;
            if (sc.lane_step) hipLaunchKernelGGL(k_trace_closest_bvh_step, dim3(lanes_grid(sc, n)), dim3(64), 0, st, scl, d_rays, d_active, n, eps_f, d_out, d_order, from);
            else hipLaunchKernelGGL(k_trace_closest_bvh_lanes, dim3(lanes_grid(sc, n)), dim3(64), 0, st, scl, d_rays, d_active, n, eps_f, d_out, d_order);
        }
        hipLaunchKernelGGL(k_trace_closest_lane, dim3(grid_for(n, 64)), dim3(64), 0, st, sc, d_rays, d_active, n, eps_f, lanes, d_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_trace_closest, dim3(grid_for(n, 256)), dim3(256), 0, st, sc, d_rays, d_active, n, eps_f, fast, d_out);
    return hipGetLastError();
}

hipError_t rtk_trace_any(const DevSceneView& sc, const rt_ray* d_rays, const int32_t* d_active, const float* d_tmax,
                         int64_t n, float eps_f, int fast, uint8_t* d_occ, hipStream_t st, const int32_t* d_order, const RaySrc* src) {
    if (n <= 0) return hipSuccess;
    const RaySrc from = (src && src->base && src->div > 0) ? *src : RaySrc{nullptr, 0, 1};
    if (fast == 3) {
        hipLaunchKernelGGL(k_trace_any_bvh, dim3(grid_for(n, 256)), dim3(256), 0, st, sc, d_rays, d_active, d_tmax, n, eps_f, d_occ, d_order);
        return hipGetLastError();
    }
    if (fast == 2) {
        const int lanes = bvh_lanes_usable(sc);
        if (lanes) {
            DevSceneView scl = sc;
            hipError_t e = next_ray_counter(scl, st);
            if (e != hipSuccess) return e;
            if (sc.lane_step) hipLaunchKernelGGL(k_trace_any_bvh_step, dim3(lanes_grid(sc, n)), dim3(64), 0, st, scl, d_rays, d_active, d_tmax, n, eps_f, d_occ, d_order, from);
            else hipLaunchKernelGGL(k_trace_any_bvh_lanes, dim3(lanes_grid(sc, n)), dim3(64), 0, st, scl, d_rays, d_active, d_tmax, n, eps_f, d_occ, d_order);
        }
        hipLaunchKernelGGL(k_trace_any_lane, dim3(grid_for(n, 64)), dim3(64), 0, st, sc, d_rays, d_active, d_tmax, n, eps_f, lanes, d_occ);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_trace_any, dim3(grid_for(n, 256)), dim3(256), 0, st, sc, d_rays, d_active, d_tmax, n, eps_f, fast, d_occ);
    return hipGetLastError();
}

hipError_t rtk_trace_work(const DevSceneView& sc, const rt_ray* d_rays, const int32_t* d_active, const float* d_tmax,
                          int64_t n, float eps_f, int fast, int any, unsigned long long* d_out4, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_trace_work, dim3(grid_for(n, 256)), dim3(256), 0, st, sc, d_rays, d_active, d_tmax, n, eps_f, fast,
                       any, d_out4);
    return hipGetLastError();
}

hipError_t rtk_generate_linear(const rt_camera& cam, int64_t first, int64_t n, float ox, float oy, rt_ray* d_rays,
                               hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_generate_linear, dim3(grid_for(n, 256)), dim3(256), 0, st, cam, first, n, ox, oy, d_rays);
    return hipGetLastError();
}

hipError_t rtk_generate_tiled(const rt_camera& cam, const int32_t* d_tiles, int64_t ntiles, float ox, float oy,
                              rt_ray* d_rays, int32_t* d_pix, hipStream_t st) {
    if (ntiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_generate_tiled, dim3(grid_for(ntiles * 64, 256)), dim3(256), 0, st, cam, d_tiles, ntiles, ox,
                       oy, d_rays, d_pix);
    return hipGetLastError();
}

hipError_t rtk_generate_drt(const rt_camera& cam, const int32_t* d_tiles, int64_t ntiles, int spp, uint64_t seed,
                            float lens_radius, float focal_distance, rt_ray* d_rays, int32_t* d_pix, float* d_w,
                            uint32_t* d_code, int32_t* d_pixlist, hipStream_t st) {
    if (ntiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_generate_drt, dim3(grid_for(ntiles * 64 * spp, 256)), dim3(256), 0, st, cam, d_tiles, ntiles, spp,
                       seed, lens_radius, focal_distance, d_rays, d_pix, d_w, d_code, d_pixlist);
    return hipGetLastError();
}
