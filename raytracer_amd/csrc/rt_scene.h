// rt_scene.h -- the opaque rt_scene of include/rt_hip.h (host side only).
#pragma once
#include "rt_internal.h"
#include <hip/hip_runtime_api.h>

#define RT_NSCRATCH 56

struct DeviceGuard {
    int prev, cur = -1; bool ok;
    explicit DeviceGuard(int dev);
    ~DeviceGuard();
};

struct PhotonMapDev {
    DevPhoton* d_photons = nullptr;   // [n+1], heap order, slot 0 unused
    int64_t n = 0;
    int64_t half = 0;                 // half_stored_photons = n/2 - 1 (photon.cpp:168)
    float box[6] = {0, 0, 0, 0, 0, 0};// lo[3], hi[3] of the photon positions (the query sort's grid)
};

struct rt_scene {
    int device = 0;
    int64_t nprim = 0;
    DevSceneView view;
    float eps_f = 0.f;
    unsigned long long* d_knn_stats = nullptr;   // 16 work counters of the wave-shared kNN kernels, NULL = not counting (RT_OPT_KNN_STATS)
    unsigned long long* d_knn_stats_buf = nullptr;
    int knn_fast = 1;                 // radiance estimate: 1 = wave-shared kernel (default), 0 = strict list replay
    int fused_primary = 1;            // rt_render with primary_only: one kernel for the whole frame (RT_OPT_FUSED_PRIMARY)
    int lane_depth = 2;               // rt_render: passes of this depth and deeper trace per lane (RT_OPT_LANE_DEPTH)
    int path_depth = 3;               // rt_render: rays of this depth and deeper are traced as whole subtrees of the recursion by ONE
                                      // persistent kernel (kernels_paths.hip, RT_OPT_PATH_DEPTH); a value above max_depth: never
    int ray_sort = 0;                 // rt_render: per-lane passes hand rays to lanes in (cell, octant, Morton) order (RT_OPT_RAY_SORT;
                                      // measured slower on BASELINE configuration 3, DESIGN 4.1c, so off by default)
    int overlap_shadow = 1;           // rt_render: the shadow half of a pass on a second stream beside the next pass (RT_RENDER_OVERLAP=0: off)
    hipStream_t aux_stream = nullptr; // that stream, created by the first frame that uses it
    float bvh_lo[3] = {0, 0, 0}, bvh_hi[3] = {0, 0, 0};   // box of the tree triangles (the grid of the ray-order keys)
    int fast = 3;                     // traversal: 3 = the build's own BVH (default), 1 = the reference's tree, ordered + culled
                                      // (4.1b), 2 = the same per lane, 0 = strict reference order
    rt_scene_info info;
    std::vector<rt_light> host_lights;
    std::vector<void*> allocs;
    int64_t device_bytes = 0;
    void* scratch_ptr[RT_NSCRATCH] = {};
    size_t scratch_cap[RT_NSCRATCH] = {};
    PhotonMapDev pmap[2];
    float* d_radiance_out = nullptr;  // set for the duration of rt_render_radiance: [resx*resy][3] un-quantised colour
    float* d_dir_tables = nullptr;    // cosTheta, sinTheta, cosPhi, sinPhi [4][256] (photon.cpp:24-27)
    // the micro-tile list of the last rt_render (a frame loop asks for the same one every time): key, host copy, device copy
    int tiles_key[5] = {-1, -1, -1, -1, -1};
    std::vector<int32_t> tiles_host;
    int32_t* d_tiles_cache = nullptr;

    int dev_alloc(void** p, size_t bytes);
    int scratch(int slot, size_t bytes, void** p);
};

// HIP events of one call, destroyed on every way out of it
struct EventBag {
    std::vector<hipEvent_t> ev;
    ~EventBag() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
    // a new event recorded on `st`; nullptr (and *err set) on failure
    hipEvent_t mark(hipStream_t st, hipError_t* err) {
        hipEvent_t e = nullptr;
        *err = hipEventCreate(&e);
        if (*err != hipSuccess) return nullptr;
        ev.push_back(e);
        *err = hipEventRecord(e, st);
        return *err == hipSuccess ? e : nullptr;
    }
};

int rt_fail(int code, const char* fmt, ...);
int rt_hipfail(hipError_t e, const char* what);
int rt_noalloc_setup(int dev);      // noalloc.hip: resolves the fixed workspace of rt_photon_irradiance_noalloc_dev on device `dev`
