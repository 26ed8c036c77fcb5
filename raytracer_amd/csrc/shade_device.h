// shade_device.h -- device-side shading routines shared by the wavefront Whitted kernels (kernels_shade.hip) and the
// fused primary + shadow kernel (kernels_primary.hip): one definition, so both paths compute the same bytes.
//
// Replaces (paths under the reference root): the local shading of SimpleMeshRayTrace, HW4/hw4/hw4.cpp:881-951;
// IsInShadow's ray, :961-963; GenerateColorImage's pixel write, :1108-1122.
#pragma once
#include <hip/hip_runtime.h>
#include "rt_internal.h"
#include "trace_device.h"
#include "rng_device.h"

namespace {

__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }   // std::max

struct HitCtx {
    float ox, oy, oz, dx, dy, dz;      // the ray
    float hx, hy, hz;                  // hitPoint
    float nx, ny, nz;                  // normal, flipped against the ray (hw4.cpp:888-891)
    int prim, mat;
};

// c.prim, the ray and the hit point are set: material and normal
__device__ __forceinline__ void hit_context(const DevSceneView& sc, HitCtx& c) {
    c.mat = sc.prim_material[c.prim];
    if (sc.prim_type[c.prim] == RT_PRIM_SPHERE) {                 // Sphere::GetNormal, hw4.cpp:118-120
        const float* g = sc.prim_geom + 4 * (size_t)c.prim;
        c.nx = c.hx - g[0]; c.ny = c.hy - g[1]; c.nz = c.hz - g[2];
        normalize3(c.nx, c.ny, c.nz);
    } else {                                                      // Triangle::GetNormal, hw4.cpp:122-137
        const float* n = sc.prim_normal + 3 * (size_t)c.prim;
        c.nx = n[0]; c.ny = n[1]; c.nz = n[2];
    }
    if (c.nx * c.dx + c.ny * c.dy + c.nz * c.dz > 0) { c.nx = -1.0f * c.nx; c.ny = -1.0f * c.ny; c.nz = -1.0f * c.nz; }
}

__device__ __forceinline__ bool load_hit(const DevSceneView& sc, const rt_ray* __restrict__ rays,
                                         const rt_hit* __restrict__ hits, int64_t i, HitCtx& c) {
    const float* h = reinterpret_cast<const float*>(hits + i);
    c.prim = reinterpret_cast<const int32_t*>(h)[0];
    if (c.prim < 0) return false;
    c.hx = h[2]; c.hy = h[3]; c.hz = h[4];
    const float* r = reinterpret_cast<const float*>(rays + i);
    c.ox = r[0]; c.oy = r[1]; c.oz = r[2]; c.dx = r[3]; c.dy = r[4]; c.dz = r[5];
    hit_context(sc, c);
    return true;
}

// IsInShadow's ray towards the light position (lx, ly, lz), hw4.cpp:961-963
__device__ __forceinline__ void shadow_ray(const DevSceneView& sc, const HitCtx& c, float lx, float ly, float lz, float* o6) {
    const float bx = c.hx + sc.shadow_bias * c.nx, by = c.hy + sc.shadow_bias * c.ny, bz = c.hz + sc.shadow_bias * c.nz;
    float dx = lx - bx, dy = ly - by, dz = lz - bz;
    normalize3(dx, dy, dz);
    o6[0] = bx; o6[1] = by; o6[2] = bz; o6[3] = dx; o6[4] = dy; o6[5] = dz;
}

// The point of light l a shadow ray aims at: the light's position, or -- soft shadows of the distributed ray tracer
// (build-defined) -- one uniform point of the ball of radius light_radius around it, drawn from the ray's
// stream (key = pixel << 16 | sample, position in the recursion, dimension 16 + 8 l)
__device__ __forceinline__ void light_point(const DevSceneView& sc, int l, bool drt, float light_radius, uint64_t seed, int px,
                                            uint32_t code, float& lx, float& ly, float& lz) {
    const DevLight lt = sc.lights[l];
    lx = lt.pos[0]; ly = lt.pos[1]; lz = lt.pos[2];
    if (light_radius > 0.f && drt) {
        float b3[3];
        rt_ball(seed, ((uint64_t)px << 16) | (code >> 16), code & 0xffffu, 16 + 8 * l, b3);
        lx = lx + light_radius * b3[0]; ly = ly + light_radius * b3[1]; lz = lz + light_radius * b3[2];
    }
}

// The reflected ray of a hit, hw4.cpp:897-904: origin = the hit point, direction = dir - 2 (dir . n) n; with the glossy lobe
// of the distributed ray tracer the direction + glossy * (a uniform point of the unit ball), normalised
__device__ __forceinline__ void reflect_child(const HitCtx& c, float glossy, bool drt, uint64_t seed, int px, uint32_t code, float* o6) {
    const float dot = c.dx * c.nx + c.dy * c.ny + c.dz * c.nz;
    const float k2 = 2 * dot;                                  // reflected = dir - 2*dot*normal
    o6[0] = c.hx; o6[1] = c.hy; o6[2] = c.hz;
    float rx = c.dx - k2 * c.nx, ry = c.dy - k2 * c.ny, rz = c.dz - k2 * c.nz;
    if (glossy > 0.f && drt) {
        float b3[3];
        rt_ball(seed, ((uint64_t)px << 16) | (code >> 16), code & 0xffffu, 8, b3);
        rx = rx + glossy * b3[0]; ry = ry + glossy * b3[1]; rz = rz + glossy * b3[2];
        normalize3(rx, ry, rz);
    }
    o6[3] = rx; o6[4] = ry; o6[5] = rz;
}

// The refracted ray of a hit, hw4.cpp:907-924: Snell with n1 (the ray's current index) -> n2 (the material's), origin
// displaced by refract_bias against the normal
__device__ __forceinline__ void refract_child(const DevSceneView& sc, const HitCtx& c, float n1, float n2, float* o6) {
    const float cos1 = c.dx * c.nx + c.dy * c.ny + c.dz * c.nz;
    const float ratio = n1 / n2;
    const float sin2 = ratio * sqrt_rn(1 - (cos1 * cos1));
    const float kk = cos1 * ratio - sqrt_rn(1 - (sin2 * sin2));
    float tx = ratio * c.dx + kk * c.nx, ty = ratio * c.dy + kk * c.ny, tz = ratio * c.dz + kk * c.nz;
    normalize3(tx, ty, tz);
    o6[0] = c.hx - sc.refract_bias * c.nx; o6[1] = c.hy - sc.refract_bias * c.ny; o6[2] = c.hz - sc.refract_bias * c.nz;
    o6[3] = tx; o6[4] = ty; o6[5] = tz;
}

// position of a child in the recursion tree (rng_device.h: reflect 2c, refract 2c + 1), sample index kept in the top half
__device__ __forceinline__ uint32_t child_code(uint32_t code, unsigned slot) {
    return (code & 0xffff0000u) | ((2u * (code & 0xffffu) + slot) & 0xffffu);
}

// ambient + per-light Blinn-Phong, hw4.cpp:926-948, including the reference's quirk that `diffuse` / `specular` keep
// the previous light's values when a light is shadowed (:876-877, :947).  occluded(l): light l is in shadow.
template <class Occ>
__device__ __forceinline__ void shade_local(const DevSceneView& sc, const HitCtx& c, float ambient_intensity, float eyex, float eyey,
                                            float eyez, Occ occluded, float& cr, float& cg, float& cb) {
    const DevMaterial m = sc.materials[c.mat];
    // ambient = (surfColor * ambientIntensity) * Ka
    cr = (ambient_intensity * m.surf[0]) * m.Ka; cg = (ambient_intensity * m.surf[1]) * m.Ka; cb = (ambient_intensity * m.surf[2]) * m.Ka;
    float dr = 0, dg = 0, db = 0, sr = 0, sg = 0, sb = 0;
    const int L = sc.nlights;
    for (int l = 0; l < L; ++l) {
        if (!occluded(l)) {
            const DevLight lt = sc.lights[l];
            float tx = lt.pos[0] - c.hx, ty = lt.pos[1] - c.hy, tz = lt.pos[2] - c.hz;
            normalize3(tx, ty, tz);
            float dot = max_std(tx * c.nx + ty * c.ny + tz * c.nz, 0.0f);
            const float ir = lt.intensity * m.surf[0], ig = lt.intensity * m.surf[1], ib = lt.intensity * m.surf[2];
            const float kd = m.Kd * dot;
            dr = kd * ir; dg = kd * ig; db = kd * ib;
            float ex = eyex - c.hx, ey = eyey - c.hy, ez = eyez - c.hz;
            normalize3(ex, ey, ez);
            float hx = ex + tx, hy = ey + ty, hz = ez + tz;
            normalize3(hx, hy, hz);
            dot = max_std(hx * c.nx + hy * c.ny + hz * c.nz, 0.0f);
            const float pw = powf(dot, m.specularity);
            sr = pw * (m.Ks * ir); sg = pw * (m.Ks * ig); sb = pw * (m.Ks * ib);
        }
        cr = (cr + dr) + sr; cg = (cg + dg) + sg; cb = (cb + db) + sb;
    }
}

// GenerateColorImage's pixel write, hw4.cpp:1108-1122
__device__ __forceinline__ uint8_t quantize(float c) {
    if (c > 1.0f) return 255;
    const float v = c * 255.0f;
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return 0;     // cvttss2si's 0x80000000 -> low byte 0
    return (uint8_t)(int32_t)v;
}

}  // namespace
