// kernels_paths.hip -- the deep end of the Whitted / distributed ray tracer as ONE persistent kernel.
//
// Replaces, for the rays of recursion depth >= RT_OPT_PATH_DEPTH, the per-depth passes of render.cpp
// (trace_closest -> k_shade_flags -> trace_any -> k_shade_lights -> k_accumulate -> scan + k_spawn, once per depth of
// SimpleMeshRayTrace, HW4/hw4/hw4.cpp:794-955): a lane takes a ray of the depth-P queue and follows the WHOLE subtree
// of the reference's recursion below it -- closest hit (hw4.cpp:808-871), one shadow ray per light (IsInShadow,
// :957-1051), the local shading (:926-948), then the reflected child (:897-904) and the refracted one (:907-924) --
// depth first, the second child of a hit waiting on a small per-lane stack in device memory.  When a lane's subtree
// is finished it writes ONE contribution for its queue entry (the subtree's weighted sum, added in depth-first order)
// and claims the next entry, so the launch costs the sum of the walks and there is no launch, no compaction and no
// host round trip per depth.  k_accumulate then adds the entries per pixel in queue order, as for any pass: the sums
// depend on the queue's order only, which the stable compaction fixes -- frames stay byte-identical under tiling and
// chunking.  Against the per-depth passes a pixel's contributions are added in another association (subtree by subtree
// instead of depth by depth): float round-off, no more.
//
// Why: on BASELINE configuration 3 (1 M triangles, 1080p x 64 spp) the passes of depth >= 2 hold 7 % of the rays and
// took 40 of the frame's 82 ms -- nine launches per kernel, each ending in the tail of its longest walks, with the
// shading / compaction kernels and a read-back of the child count between them (profiles/r02_timeline_drt_1m.txt).
//
// Shape: every lane is a small state machine.  The walk step (the per-lane BVH walk of trace_device.h: next leaf whose
// inflated box the ray passes, then its <= 6 triangles) is what all walking lanes execute together, closest-hit and
// shadow walks side by side (the mode is a lane value, the accept rule of bvh_lane_leaf_any follows it).  Everything
// else -- shading, spawning, popping, claiming -- happens in `events`, run for all waiting lanes at once when at
// least `batch` lanes wait or none walks: a divergent event per lane and walk end would cost more instructions than
// the walks themselves.
// Same device functions as the pass kernels (trace_device.h, shade_device.h), same tests, same accept rule: every ray
// gets the hit, the occlusion bit and the children the passes give it.
#include <cstdlib>
#include "shade_device.h"

#define PATH_MAX_LIGHTS 32

struct PathStackEntry { float o[6]; float w, nr; uint32_t code; int32_t depth; int32_t src; };     // 44 B: a pending child (src: the triangle it starts on)
static_assert(sizeof(PathStackEntry) == 44, "PathStackEntry must be 44 bytes");

struct PathArgs {
    const rt_ray* rays; const int32_t* pix; const float* w; const float* nr; const uint32_t* code;   // the depth-P queue
    int64_t n;
    int32_t depth0, max_depth;
    float eps_f, ambient, eye[3], start_nr;
    uint64_t seed; float light_radius, glossy;
    float* contrib;                      // [n][3]: one weighted sum per queue entry
    PathStackEntry* stack; int32_t stack_depth;      // [wave][stack_depth][64 lanes]
    unsigned long long* counters;        // the frame's counters (render.cpp): [0] rays traced, [1] hits, [3] reflect, [4] refract
                                         // children; [5] closest-hit rays, [6] shadow rays of THIS kernel
    unsigned long long* work;            // COUNT: [0] box tests, [1] triangle records of the closest walks; [4], [5] of the shadow walks
    int32_t batch;                       // events run when this many lanes wait (or no lane walks)
    const int32_t* src;                  // NULL, or per queue entry the triangle its ray starts on (-1: unknown)
    const int32_t* order;                // NULL, or the order in which queue entries are handed to lanes (rtk_ray_order); sums go to the entries' own slots
};

enum PathMode { M_IDLE = 0, M_CLOSEST, M_SHADOW, M_SHADOW_SETUP, M_SHADE, M_NEXT, M_START };

#ifndef RT_PATH_WPE
#define RT_PATH_WPE 4
#endif
template <bool COUNT, bool STEP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_PATH_WPE, 8)))
k_paths_bvh_lanes(DevSceneView sc, PathArgs a) {
    const int lane = threadIdx.x & 63;
    const int nn = sc.nbvh;
    const int L = sc.nlights;
    const bool drt = a.code != nullptr;
    int mode = M_IDLE;
    bool more = true;
    // the queue entry this lane works for, and the sum of its subtree so far
    int64_t idx = 0; int px = -1;
    float ar = 0.f, ag = 0.f, ab = 0.f;
    // the ray at hand
    float w = 0.f, nr = 1.f; uint32_t code = 0u; int depth = 0;
    int src = -1;                                 // the triangle the ray at hand starts on (-1: unknown): its walk starts at that leaf
    BvhLaneWalk wk; wk.i = nn; wk.far_limit = 0.f; wk.nodes = sc.bvh_oct[0];
    wk.r = Rayf{0, 0, 0, 0, 0, 1};
    wk.bc = BvhConsts{1.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool occ = false;
    LaneCursor cu; lane_cursor_reset(cu);         // the single-step walk's cursor (trace_device.h: bvh_lane_step); idle in the while-while walk
    // the hit being shaded (alive from the end of the closest walk to the children)
    HitCtx c; c.ox = c.oy = c.oz = 0.f; c.dx = c.dy = 0.f; c.dz = 1.f; c.hx = c.hy = c.hz = 0.f; c.nx = c.ny = 0.f; c.nz = 1.f; c.prim = -1; c.mat = 0;
    int light = 0; unsigned occbits = 0u;
    int sp = 0;
    PathStackEntry* const stk = a.stack + (size_t)blockIdx.x * (size_t)a.stack_depth * 64 + lane;     // entry s at stk[64 s]
    unsigned n_rays = 0, n_hits = 0, n_refl = 0, n_refr = 0, n_shadow = 0;
    unsigned wb_c = 0, wt_c = 0, wb_s = 0, wt_s = 0;
    unsigned d_iter = 0, d_on = 0, d_ev = 0, d_wait = 0;       // COUNT: wave steps, lanes that stepped, events, lanes waiting at events (lane 0 / per lane)
    // claims: eight contiguous ranges of the queue with a counter each, a wave starts in the range of its workgroup's XCD
    // label and moves on when its own is used up; one atomic per chunk of entries (trace_device.h: RayPool)
    RayPool pool; ray_pool_init(pool, a.n);

    for (;;) {
        const bool in_walk = mode == M_CLOSEST || mode == M_SHADOW;
        const bool walk_on = lane_walk_busy(wk, cu, nn);
        const bool walking = in_walk && walk_on;
        const bool waiting = (in_walk && !walk_on) || (mode == M_IDLE && more);
        const unsigned long long wm = rt_ballot(waiting), km = rt_ballot(walking);
        if ((wm | km) == 0ull) break;
        if (wm != 0ull && (km == 0ull || __builtin_popcountll(wm) >= a.batch)) {
            if (COUNT) { d_ev += 1; d_wait += waiting ? 1u : 0u; }
            // ---- events, in the order a ray moves through them ----------------------------------------------------
            // A: a closest-hit walk has ended (hw4.cpp:873-895)
            if (mode == M_CLOSEST && !walk_on) {
                if (best.prim < 0) mode = M_NEXT;
                else {
                    ++n_hits;
                    c.ox = wk.r.ox; c.oy = wk.r.oy; c.oz = wk.r.oz; c.dx = wk.r.dx; c.dy = wk.r.dy; c.dz = wk.r.dz;
                    c.hx = best.px; c.hy = best.py; c.hz = best.pz; c.prim = best.prim;
                    hit_context(sc, c);
                    light = 0; occbits = 0u;
                    mode = L > 0 ? M_SHADOW_SETUP : M_SHADE;
                }
            } else if (mode == M_SHADOW && !walk_on) {
                // B: a shadow walk has ended (hw4.cpp:929-931)
                occbits |= (occ ? 1u : 0u) << light;
                ++light;
                mode = light < L ? M_SHADOW_SETUP : M_SHADE;
            }
            // C: every light is answered: local shading (hw4.cpp:926-948), then the children (:897-924)
            if (mode == M_SHADE) {
                float cr, cg, cb;
                shade_local(sc, c, a.ambient, a.eye[0], a.eye[1], a.eye[2], [&](int l) { return ((occbits >> l) & 1u) != 0u; }, cr, cg, cb);
                cr *= w; cg *= w; cb *= w;
                ar = ar + cr; ag = ag + cg; ab = ab + cb;
                mode = M_NEXT;
                if (depth < a.max_depth) {
                    const DevMaterial m = sc.materials[c.mat];
                    const bool f0 = m.reflect > 0.0001f;                 // hw4.cpp:897
                    const bool f1 = (double)m.refract > 0.0001;          // hw4.cpp:907 (double literal)
                    float o6[6];
                    if (f1) {
                        refract_child(sc, c, nr, m.Nr, o6);
                        ++n_refr;
                        if (f0) {                                        // the reflected child goes first: this one waits
                            PathStackEntry* e = stk + 64 * (size_t)sp;
                            e->o[0] = o6[0]; e->o[1] = o6[1]; e->o[2] = o6[2]; e->o[3] = o6[3]; e->o[4] = o6[4]; e->o[5] = o6[5];
                            e->w = w * m.refract; e->nr = m.Nr; e->code = child_code(code, 1u); e->depth = depth + 1; e->src = c.prim;
                            ++sp;
                        } else {
                            wk.r = Rayf{o6[0], o6[1], o6[2], o6[3], o6[4], o6[5]};
                            w = w * m.refract; nr = m.Nr; code = child_code(code, 1u); depth = depth + 1; src = c.prim;
                            mode = M_START;
                        }
                    }
                    if (f0) {
                        reflect_child(c, a.glossy, drt, a.seed, px, code, o6);
                        ++n_refl;
                        wk.r = Rayf{o6[0], o6[1], o6[2], o6[3], o6[4], o6[5]};
                        w = w * m.reflect; code = child_code(code, 0u); depth = depth + 1; src = c.prim;      // nr stays (hw4.cpp:903)
                        mode = M_START;
                    }
                }
            }
            // D: this ray is done: a waiting child, or the subtree is finished
            if (mode == M_NEXT) {
                if (sp > 0) {
                    --sp;
                    const PathStackEntry* e = stk + 64 * (size_t)sp;
                    wk.r = Rayf{e->o[0], e->o[1], e->o[2], e->o[3], e->o[4], e->o[5]};
                    w = e->w; nr = e->nr; code = e->code; depth = e->depth; src = e->src;
                    mode = M_START;
                } else {
                    a.contrib[3 * idx] = ar; a.contrib[3 * idx + 1] = ag; a.contrib[3 * idx + 2] = ab;
                    mode = M_IDLE;
                }
            }
            // E: idle lanes claim queue entries
            {
                const unsigned long long need = rt_ballot(mode == M_IDLE && more);
                if (need != 0ull) {
                    if (!ray_pool_fill(sc, pool, a.n, lane)) more = false;
                    else {
                        const int64_t claim = ray_pool_take(pool, need, lane);  // (-1: the pool ran out -- claims again at the next events)
                        const int64_t my = (claim >= 0 && a.order != nullptr) ? (int64_t)a.order[claim] : claim;
                        if (my >= 0) {
                            const int p = a.pix[my];
                            if (p < 0) { a.contrib[3 * my] = 0.f; a.contrib[3 * my + 1] = 0.f; a.contrib[3 * my + 2] = 0.f; }
                            else {
                                idx = my; px = p; ar = ag = ab = 0.f; sp = 0;
                                wk.r = load_ray(a.rays, my);
                                w = a.w ? a.w[my] : 1.0f; nr = a.nr ? a.nr[my] : a.start_nr; code = a.code ? a.code[my] : 0u;
                                depth = a.depth0; src = a.src ? a.src[my] : -1;
                                mode = M_START;
                            }
                        }
                    }
                }
            }
            // F, G: set a walk up -- the ray at hand (closest hit), or the shadow ray towards light `light`
            bool fresh = false, unfit = false;
            if (mode == M_SHADOW_SETUP) {
                float lx, ly, lz, s6[6];
                light_point(sc, light, drt, a.light_radius, a.seed, px, code, lx, ly, lz);
                shadow_ray(sc, c, lx, ly, lz, s6);
                wk.r = Rayf{s6[0], s6[1], s6[2], s6[3], s6[4], s6[5]};
                ++n_shadow;
            }
            if (mode == M_START || mode == M_SHADOW_SETUP) {
                const bool shadow = mode == M_SHADOW_SETUP;
                if (!shadow) ++n_rays;
                const RayAux aux = ray_aux(wk.r);
                fresh = true; occ = false;
                best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
                lane_cursor_reset(cu);
                if (bvh_fit(aux)) {                                                            // HW4's IsInShadow has no far limit (Q4)
                    if (STEP) bvh_lane_start_from(sc, wk.r, aux, shadow ? 1e9f : 0.f, wk, shadow ? c.prim : src);
                    else bvh_lane_start(sc, wk.r, aux, shadow ? 1e9f : 0.f, wk);
                }
                else { unfit = true; wk.i = nn; wk.far_limit = shadow ? 1e9f : 0.f; }
                mode = shadow ? M_SHADOW : M_CLOSEST;
            }
            // rays the hierarchy does not take (a zero or non-finite component, not unit length): the reference tree, whole
            if (rt_ballot(unfit) != 0ull) {
                bool o2 = false;
                traverse_lane<false>(sc, wk.r, unfit && mode == M_CLOSEST, a.eps_f, 0.f, false, best, o2);
                Closest dummy; dummy.len = 1e9f; dummy.prim = -1; dummy.px = dummy.py = dummy.pz = 0.f; dummy.rank = 0x7fffffff;
                traverse_lane<true>(sc, wk.r, unfit && mode == M_SHADOW, a.eps_f, 0.f, false, dummy, occ);
            }
            // primitives outside the trees (spheres), for the rays just set up (kernels_trace.hip: bvh_lanes says why first)
            if (sc.nloose > 0 && rt_ballot(fresh) != 0ull) {
                bool o2 = false;
                loose_prims<false>(sc, wk.r, fresh && mode == M_CLOSEST, a.eps_f, 0.f, false, best, o2);
                Closest dummy; dummy.len = 1e9f; dummy.prim = -1; dummy.px = dummy.py = dummy.pz = 0.f; dummy.rank = 0x7fffffff;
                loose_prims<true>(sc, wk.r, fresh && mode == M_SHADOW, a.eps_f, 0.f, false, dummy, occ);
                if (mode == M_SHADOW && occ) lane_walk_end(wk, cu, nn); // occluded already: nothing to walk
            }
        }
        // ---- one walk step for every lane inside a walk -----------------------------------------------------------
        {
            const bool any = mode == M_SHADOW;
            const bool on = (mode == M_CLOSEST || any) && lane_walk_busy(wk, cu, nn);
            if (STEP) {
                // one record for every walking lane; candidates take their reachability test when enough of them wait or
                // no lane can step (kernels_trace.hip: bvh_lanes_step)
                const unsigned long long pend = rt_ballot(on && cu.acc >= 0), stepping = rt_ballot(on && cu.acc < 0);
                if ((pend | stepping) != 0ull) {
                    const bool do_acc = pend != 0ull && (stepping == 0ull || __builtin_popcountll(pend) >= sc.lane_acc_batch);
                    unsigned nb = 0, nt = 0;
                    if (COUNT) { d_iter += 1; d_on += (on && (cu.acc < 0 || do_acc)) ? 1u : 0u; }
                    bvh_lane_step(sc, wk, cu, on, any, a.eps_f, false, do_acc, best, occ, COUNT ? &nb : nullptr, COUNT ? &nt : nullptr);
                    if (COUNT) { if (any) { wb_s += nb; wt_s += nt; } else { wb_c += nb; wt_c += nt; } }
                }
            } else if (rt_ballot(on) != 0ull) {
                const float limit = any ? wk.far_limit : best.len;
                int t0, t1;
                unsigned nb = 0, nt = 0;
                if (sc.bvh_c32) bvh_lane_next_leaf_c32_lim(sc, wk, on, limit, t0, t1, COUNT ? &nb : nullptr);
                else bvh_lane_next_leaf_lim(sc, wk, on, limit, t0, t1, COUNT ? &nb : nullptr);
                if (on && t0 < t1) { if (bvh_lane_leaf_any(sc, wk, t0, t1, a.eps_f, false, any, best, COUNT ? &nt : nullptr)) occ = true; }
                if (COUNT) { if (any) { wb_s += nb; wt_s += nt; } else { wb_c += nb; wt_c += nt; } }
            }
        }
    }
    // the wave's counts: one atomic per counter and wave
    {
        unsigned v[5] = {n_rays, n_hits, n_refl, n_refr, n_shadow};
        const int slot[5] = {0, 1, 3, 4, 6};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            unsigned long long x = v[k];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
            if (lane == 0 && x) { atomicAdd(a.counters + slot[k], x); if (k == 0) atomicAdd(a.counters + 5, x); }
        }
        if (COUNT) {
            unsigned q[8] = {wb_c, wt_c, wb_s, wt_s, lane == 0 ? d_iter : 0u, d_on, lane == 0 ? d_ev : 0u, d_wait};
            const int ws[8] = {0, 1, 4, 5, 2, 3, 6, 7};     // [2], [3], [6], [7]: diagnostics (render.cpp: RT_PATH_DIAG)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                unsigned long long x = q[k];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
                if (lane == 0 && x) atomicAdd(a.work + ws[k], x);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
int rtk_paths_max_lights() { return PATH_MAX_LIGHTS; }

// waves of the launch: as many as the chip holds at the kernel's occupancy, never more than the queue needs
static int64_t paths_waves(int64_t n) {
    static const int64_t forced = [] { const char* e = getenv("RT_PATH_WAVES"); return e && atoll(e) > 0 ? atoll(e) : (int64_t)0; }();
    const int64_t cap = forced ? forced : 4096;
    const int64_t wv = (n + 63) / 64;
    return wv < cap ? wv : cap;
}
size_t rtk_paths_stack_bytes(int64_t n, int depth0, int max_depth) {
    const int sd = max_depth - depth0 + 1;
    return (size_t)paths_waves(n) * 64 * (size_t)(sd > 1 ? sd : 1) * sizeof(PathStackEntry);
}

// counters: the frame's (render.cpp), 8 words; work: 8 words or NULL (then nothing is counted)
hipError_t rtk_paths(const DevSceneView& sc0, const rt_ray* rays, const int32_t* pix, const float* w, const float* nr,
                     const uint32_t* code, int64_t n, int depth0, int max_depth, float eps_f, float ambient, const float eye[3],
                     float start_nr, uint64_t seed, float light_radius, float glossy, float* contrib, void* stack,
                     unsigned long long* counters, unsigned long long* work, hipStream_t st, const int32_t* order, const int32_t* src) {
    if (n <= 0) return hipSuccess;
    DevSceneView sc = sc0;
    hipError_t e = rtk_next_ray_counter(sc, st);         // this launch's eight range counters, zeroed on `st`
    if (e != hipSuccess) return e;
    static const int batch = [] { const char* v = getenv("RT_PATH_BATCH"); const int b = v ? atoi(v) : 16; return b < 1 ? 1 : (b > 64 ? 64 : b); }();
    PathArgs a;
    a.rays = rays; a.pix = pix; a.w = w; a.nr = nr; a.code = code; a.n = n; a.depth0 = depth0; a.max_depth = max_depth;
    a.eps_f = eps_f; a.ambient = ambient; a.eye[0] = eye[0]; a.eye[1] = eye[1]; a.eye[2] = eye[2]; a.start_nr = start_nr;
    a.seed = seed; a.light_radius = light_radius; a.glossy = glossy; a.contrib = contrib;
    a.stack = (PathStackEntry*)stack; a.stack_depth = (max_depth - depth0 + 1) > 1 ? (max_depth - depth0 + 1) : 1;
    a.counters = counters; a.work = work; a.batch = batch; a.order = order; a.src = src;
    const unsigned grid = (unsigned)paths_waves(n);
    if (sc.lane_step) {
        if (work) hipLaunchKernelGGL((k_paths_bvh_lanes<true, true>), dim3(grid), dim3(64), 0, st, sc, a);
        else hipLaunchKernelGGL((k_paths_bvh_lanes<false, true>), dim3(grid), dim3(64), 0, st, sc, a);
    } else {
        if (work) hipLaunchKernelGGL((k_paths_bvh_lanes<true, false>), dim3(grid), dim3(64), 0, st, sc, a);
        else hipLaunchKernelGGL((k_paths_bvh_lanes<false, false>), dim3(grid), dim3(64), 0, st, sc, a);
    }
    return hipGetLastError();
}
