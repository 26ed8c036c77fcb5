// kernels_primary.hip -- the whole "primary + shadow rays only" frame (BASELINE configuration 2) in one kernel.
//
// Replaces, for rt_params.primary_only: GenerateColorImage's pixel loop (HW4/hw4/hw4.cpp:1096-1124) down to
// SimpleMeshRayTrace's closest-hit loop (:808-871), IsInShadow (:957-1051) and the local shading (:881-951), with no
// reflection / refraction children.  The wavefront renderer (render.cpp) runs this as six kernels that hand rays, hits,
// shadow rays, occlusion bytes and contributions to each other through HBM (~0.9 GB per 4K frame); here a lane keeps
// its pixel in registers from the camera ray to the quantised colour: generate -> closest hit (wave-synchronous fast
// traversal) -> one shadow ray per light (any-hit traversal, the lanes that hit nothing ride along) -> shade -> write.
// Every step calls the device functions the wavefront kernels call (trace_device.h, shade_device.h), so the bytes are
// the same (tests/test_gpu_render.py::test_fused_primary_frame_equals_the_wavefront_passes).
#include <cstdlib>
#include "shade_device.h"

#define PRIMARY_MAX_LIGHTS 32

template <bool BVH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8)))
k_primary_only(DevSceneView sc, rt_camera cam, const int32_t* __restrict__ tiles, int64_t ntiles, float eps_f,
               float ambient_intensity, float bgr, float bgg, float bgb, uint8_t* __restrict__ rgb8,
               int32_t* __restrict__ hit_ids, float* __restrict__ radiance_out, unsigned int* __restrict__ wave_pairs,
               int64_t packed_base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(i & 63);
    const bool slot = i < ntiles * 64;
    if (rt_ballot(slot) == 0ull) return;                          // a wave beyond the last micro-tile
    const int t = slot ? (tiles ? tiles[i >> 6] : (int)(i >> 6)) : 0;
    const int tw = (cam.resx + 7) >> 3;
    const int x = (t % tw) * 8 + (lane & 7), y = (t / tw) * 8 + (lane >> 3);
    const bool inside = slot && x < cam.resx && y < cam.resy;
    float o6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 1.f};
    if (inside) primary_ray(cam, (float)x, (float)y, o6);
    const Rayf r = {o6[0], o6[1], o6[2], o6[3], o6[4], o6[5]};

    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool occ = false;
    if (BVH) traverse_bvh<false>(sc, r, inside, eps_f, 0.f, false, best, occ);
    else traverse_fast<false>(sc, r, inside, eps_f, 0.f, false, best, occ);
    loose_prims<false>(sc, r, inside, eps_f, 0.f, false, best, occ);
    const bool hit = inside && best.prim >= 0;

    // per-wave counts of rays and hits (rt_stats): no barrier, the waves of a block finish at different times
    {
        const unsigned long long m0 = rt_ballot(inside), m1 = rt_ballot(hit);
        if (lane == 0 && wave_pairs != nullptr) { wave_pairs[2 * (i >> 6)] = (unsigned)__builtin_popcountll(m0); wave_pairs[2 * (i >> 6) + 1] = (unsigned)__builtin_popcountll(m1); }
    }

    HitCtx c;
    c.ox = r.ox; c.oy = r.oy; c.oz = r.oz; c.dx = r.dx; c.dy = r.dy; c.dz = r.dz;
    c.hx = best.px; c.hy = best.py; c.hz = best.pz; c.prim = best.prim; c.mat = 0;
    c.nx = 0.f; c.ny = 0.f; c.nz = 1.f;
    if (hit) hit_context(sc, c);

    // IsInShadow per light (hw4.cpp:929-931; no far limit, Q4): the wave traverses together, valid = the lane has a hit
    unsigned occluded = 0;
    const int L = sc.nlights;
    for (int l = 0; l < L; ++l) {
        float s6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 1.f};
        if (hit) { const DevLight lt = sc.lights[l]; shadow_ray(sc, c, lt.pos[0], lt.pos[1], lt.pos[2], s6); }
        const Rayf sr = {s6[0], s6[1], s6[2], s6[3], s6[4], s6[5]};
        Closest dummy; dummy.len = 1e9f; dummy.prim = -1; dummy.px = dummy.py = dummy.pz = 0.f; dummy.rank = 0x7fffffff;
        bool o = false;
        if (BVH) traverse_bvh<true>(sc, sr, hit, eps_f, 0.f, false, dummy, o);
        else traverse_fast<true>(sc, sr, hit, eps_f, 0.f, false, dummy, o);
        loose_prims<true>(sc, sr, hit, eps_f, 0.f, false, dummy, o);
        occluded |= (o ? 1u : 0u) << l;
    }
    if (!inside) return;
    const int p = x + y * cam.resx;
    if (hit_ids) hit_ids[p] = hit ? c.prim : -1;
    float cr = 0.f, cg = 0.f, cb = 0.f;
    if (hit) {
        shade_local(sc, c, ambient_intensity, cam.center[0], cam.center[1], cam.center[2],
                    [&](int l) { return ((occluded >> l) & 1u) != 0u; }, cr, cg, cb);
        cr *= 1.0f; cg *= 1.0f; cb *= 1.0f;                    // the ray's weight (k_shade_lights)
        cr = 0.f + cr; cg = 0.f + cg; cb = 0.f + cb;           // k_accumulate onto the zeroed pixel
    }
    // rt_params.packed_output: the pixel goes to its slot in the call's pixel list (coalesced; a rank's share of a
    // tiled frame is then one contiguous buffer)
    uint8_t* out = rgb8 + 3 * (size_t)(packed_base >= 0 ? packed_base + i : (int64_t)p);
    if (hit) { out[0] = quantize(cr); out[1] = quantize(cg); out[2] = quantize(cb); }
    else { out[0] = (uint8_t)bgr; out[1] = (uint8_t)bgg; out[2] = (uint8_t)bgb; }
    if (radiance_out) { radiance_out[3 * (size_t)p] = cr; radiance_out[3 * (size_t)p + 1] = cg; radiance_out[3 * (size_t)p + 2] = cb; }
}

int rtk_primary_only_max_lights() { return PRIMARY_MAX_LIGHTS; }

// d_wave_pairs: 2 unsigned per micro-tile (= per wave), or NULL when nobody asks for the ray counts
hipError_t rtk_primary_only(const DevSceneView& sc, const rt_camera& cam, const int32_t* d_tiles, int64_t ntiles, float eps_f,
                            float ambient, const float bg[3], uint8_t* d_rgb8, int32_t* d_hit_ids, float* d_radiance,
                            unsigned int* d_wave_pairs, int bvh, int64_t packed_base, hipStream_t st) {
    if (ntiles <= 0) return hipSuccess;
    static const int block = [] { const char* e = getenv("RT_PRIMARY_BLOCK"); const int v = e ? atoi(e) : 256; return (v == 64 || v == 128 || v == 256) ? v : 256; }();
    const unsigned grid = (unsigned)((ntiles * 64 + block - 1) / block);
    if (bvh)
        hipLaunchKernelGGL(k_primary_only<true>, dim3(grid), dim3(block), 0, st, sc, cam, d_tiles, ntiles, eps_f, ambient, bg[0], bg[1],
                           bg[2], d_rgb8, d_hit_ids, d_radiance, d_wave_pairs, packed_base);
    else
        hipLaunchKernelGGL(k_primary_only<false>, dim3(grid), dim3(block), 0, st, sc, cam, d_tiles, ntiles, eps_f, ambient, bg[0], bg[1],
                           bg[2], d_rgb8, d_hit_ids, d_radiance, d_wave_pairs, packed_base);
    return hipGetLastError();
}
