// kernels_pm.hip -- wavefront kernels of the photon-mapping renderer (RT_RENDER_PHOTON) on gfx950.
//
// Replaces, under Final/photon/ of the reference:
//   pixel loop + AA taps + TestVisualizeLight   final.cpp:1146-1202, :1021-1043
//   SimpleMeshRayTrace(pmap, cpmap, ...)        final.cpp:785-983   (shading, children, gather sampling)
//   IsInShadow over the rect-light grid         final.cpp:944-975, :985-1018, photon.h:90-111
//   TraceIrradiance                             photon.cpp:714-820
//
// The reference recurses; here the frame is a set of queues.  A *sample* is one AA tap of one pixel
// (sid = pixel slot * ntap + tap); every ray carries its sample id and a scalar weight (the product of the
// reflect / refract / Fresnel factors above it).  Queues stay ordered by sample id through stable
// compaction, so the per-sample sums are evaluated by one thread in a fixed order (k_accumulate) and the
// frame does not depend on tiling or chunking.  Eye path pass (depth 0..max_depth):
//   trace_closest -> k_pm_shade -> k_pm_shadow (fused ray generation + any-hit over the light grid)
//   -> kNN on the caustic map -> k_pm_direct -> k_accumulate
//   -> k_pm_gather_spawn -> [trace_closest -> k_gi_shade -> kNN on the global map -> k_gi_contrib
//                            -> k_accumulate -> k_gi_spawn]* (TraceIrradiance, depth <= 3)
//   -> k_pm_spawn (children of the eye path)
#include "trace_device.h"
#include "rng_device.h"

#define PM_PI 3.141592        // parameter.h:6

namespace {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 muls(V3 a, float d) { return mk(d * a.x, d * a.y, d * a.z); }      // vec3 * float
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 norm(V3 a) { normalize3(a.x, a.y, a.z); return a; }
__device__ __forceinline__ float len(V3 a) { return sqrt_rn(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }

__constant__ float AA_FILTER[6][3] = {      // final.cpp:1033-1043
    {-0.52f, 0.38f, 0.128f}, {0.41f, 0.56f, 0.119f}, {0.27f, 0.08f, 0.294f},
    {-0.17f, -0.29f, 0.249f}, {0.58f, -0.55f, 0.104f}, {-0.31f, -0.71f, 0.106f}};

// GetNormal (final.cpp:94-113): un-flipped `normal`
__device__ __forceinline__ V3 prim_normal(const DevSceneView& sc, int prim, V3 hit) {
    if (sc.prim_type[prim] == RT_PRIM_SPHERE) {
        const float* g = sc.prim_geom + 4 * (size_t)prim;
        return norm(mk(hit.x - g[0], hit.y - g[1], hit.z - g[2]));
    }
    return ld3(sc.prim_normal + 3 * (size_t)prim);
}

// final.cpp:871-886 / photon.cpp:796-807
__device__ __forceinline__ float refract_terms(V3 dir, V3 normal, V3 normalL, float currentNr, V3& reflRay, V3& tdir,
                                               float& Re, float& Tr) {
    reflRay = sub(dir, muls(dir, dot(muls(normal, 2), normal)));         // dir - normal*2*normal*dir
    const bool into = dot(normal, normalL) > 0;
    const float nc = 1, nt = currentNr, nnt = into ? nc / nt : nt / nc, ddn = dot(dir, normalL);
    const float cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
    V3 t = sub(muls(dir, nnt), muls(normal, (into ? 1 : -1) * (ddn * nnt + sqrt_rn(cos2t))));
    t = norm(t);
    const float a = nt - nc, b = nt + nc, R0 = a * a / (b * b), c = 1 - (into ? -ddn : dot(t, normal));
    Re = R0 + (1 - R0) * c * c * c * c * c; Tr = 1 - Re;
    tdir = t;
    return cos2t;
}

struct HitInfo { int prim; V3 o, d, hit, normal, normalL; };

__device__ __forceinline__ bool load_hit(const DevSceneView& sc, const rt_ray* __restrict__ rays,
                                         const rt_hit* __restrict__ hits, int64_t i, HitInfo& h) {
    const float* hh = reinterpret_cast<const float*>(hits + i);
    h.prim = reinterpret_cast<const int32_t*>(hh)[0];
    if (h.prim < 0) return false;
    h.hit = mk(hh[2], hh[3], hh[4]);
    const float* r = reinterpret_cast<const float*>(rays + i);
    h.o = mk(r[0], r[1], r[2]); h.d = mk(r[3], r[4], r[5]);
    h.normal = prim_normal(sc, h.prim, h.hit);
    h.normalL = h.normal;
    if (dot(h.normal, h.d) > 0) h.normalL = muls(h.normal, -1.0f);
    return true;
}

// one vector atomic per block and counter (every thread of the block must call it: it synchronises).  Atomics of many
// waves on one address serialise, so this stays out of kernels with a thread per shadow ray.
__device__ __forceinline__ void block_add(unsigned long long* ctr, unsigned v) {
    __shared__ unsigned part[16];
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (unsigned w2 = 0; w2 < (blockDim.x + 63) / 64; ++w2) t += part[w2];
        if (t) atomicAdd(ctr, (unsigned long long)t);
    }
    __syncthreads();
}

}  // namespace

// counters: [0] primary rays traced, [1] eye-path hits, [2] gather rays traced, [3] reflect children, [4] refract
// children, [5] kNN lookups (global), [6] kNN lookups (caustic), [7] shadow rays
#define PMC_PRIMARY 0
#define PMC_HITS 1
#define PMC_GATHER 2
#define PMC_REFLECT 3
#define PMC_REFRACT 4
#define PMC_KNN_G 5
#define PMC_KNN_C 6
#define PMC_SHADOW 7

// Primary rays of tiles [tile0, tile0+nt) of `tiles` (8x8-pixel micro-tiles), ntap taps each; ray index
// i = (t*ntap + tap)*64 + lane, sample id = (t*64 + lane)*ntap + tap.  A tap whose ray meets a light quad marks
// the pixel white (final.cpp:1167-1174) and is not traced.
extern "C" __global__ void __launch_bounds__(256)
k_pm_generate(DevSceneView sc, rt_camera cam, const int32_t* __restrict__ tiles, int64_t tile0, int64_t nt, int ntap,
              int use_filter, float eps_f, rt_ray* __restrict__ rays, int32_t* __restrict__ sid, float* __restrict__ w,
              uint32_t* __restrict__ code, int32_t* __restrict__ pixmap, uint8_t* __restrict__ white,
              uint8_t* __restrict__ hit_any, int32_t* __restrict__ hit_ids, unsigned long long* __restrict__ ctr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned traced = 0;
    if (i < nt * ntap * 64) {
        const int lane = (int)(i & 63);
        const int64_t tt = (i >> 6) / ntap;
        const int tap = (int)((i >> 6) % ntap);
        const int t = tiles ? tiles[tile0 + tt] : (int)(tile0 + tt);
        const int tw = (cam.resx + 7) >> 3;
        const int x = (t % tw) * 8 + (lane & 7), y = (t / tw) * 8 + (lane >> 3);
        const int64_t ps = tt * 64 + lane;
        float* o = reinterpret_cast<float*>(rays + i);
        w[i] = 1.0f; code[i] = 1u;
        if (x >= cam.resx || y >= cam.resy) {
            if (tap == 0) { pixmap[ps] = -1; white[ps] = 0; hit_any[ps] = 0; }
            sid[i] = -1; o[0] = o[1] = o[2] = o[3] = o[4] = 0.f; o[5] = 1.f;
        } else {
            if (tap == 0) { pixmap[ps] = x + y * cam.resx; if (hit_ids) hit_ids[x + y * cam.resx] = -1; }
            const float fx = use_filter ? x + AA_FILTER[tap][0] : (float)x;
            const float fy = use_filter ? y + AA_FILTER[tap][1] : (float)y;
            primary_ray(cam, fx, fy, o);
            const Rayf r = {o[0], o[1], o[2], o[3], o[4], o[5]};
            bool sees = false;                                     // TestVisualizeLight, final.cpp:1021-1031
            for (int k = 0; k < sc.nlight_tris && !sees; ++k) {
                float px, py, pz;
                sees = tri_test(sc.light_tris + 9 * k, r, eps_f, px, py, pz);
            }
            if (sees) { white[ps] = 1; sid[i] = -1; }
            else { sid[i] = (int32_t)(ps * ntap + tap); traced = 1; }
        }
    }
    block_add(ctr + PMC_PRIMARY, traced);
}

extern "C" __global__ void __launch_bounds__(256)
k_pm_clear(int64_t nps, int ntap, uint8_t* __restrict__ white, uint8_t* __restrict__ hit_any, float* __restrict__ accum) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nps) { white[i] = 0; hit_any[i] = 0; }
    if (i < nps * ntap) { accum[3 * i] = 0.f; accum[3 * i + 1] = 0.f; accum[3 * i + 2] = 0.f; }
}

// Per eye-path ray: hit bookkeeping, un-flipped normal (kNN argument, final.cpp:909) and flipped normalL,
// child flags (3 per ray: reflect :843, refraction's reflRay :877/:894, refraction's tdir :895).
extern "C" __global__ void __launch_bounds__(256)
k_pm_shade(DevSceneView sc, const rt_ray* __restrict__ rays, const rt_hit* __restrict__ hits,
           const int32_t* __restrict__ sid, int64_t n, int ntap, int depth0, int spawn, float start_nr,
           uint8_t* __restrict__ flags, float* __restrict__ nrm, float* __restrict__ nrmL, float* __restrict__ qpos,
           uint8_t* __restrict__ hit_any, int32_t* __restrict__ hit_ids, const int32_t* __restrict__ pixmap,
           unsigned long long* __restrict__ ctr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned nh = 0;
    if (i < n) {
        const int s = sid[i];
        HitInfo h;
        const bool hit = s >= 0 && load_hit(sc, rays, hits, i, h);
        uint8_t f0 = 0, f1 = 0, f2 = 0;
        if (depth0 && s >= 0) {
            const int ps = s / ntap;
            if (hit) hit_any[ps] = 1;                              // rayHitAnything, final.cpp:829
            if (hit_ids && (s % ntap) == 0) hit_ids[pixmap[ps]] = hit ? h.prim : -1;
        }
        if (hit) {
            nh = 1;
            const DevMaterial m = sc.materials[sc.prim_material[h.prim]];
            if (spawn) {
                f0 = m.reflect > 0.0001f;
                if ((double)m.refract > 0.0001) {
                    V3 rr, td; float Re, Tr;
                    const float cos2t = refract_terms(h.d, h.normal, h.normalL, start_nr, rr, td, Re, Tr);
                    f1 = 1; f2 = !(cos2t < 0);
                }
            }
            st3(nrm + 3 * i, h.normal); st3(nrmL + 3 * i, h.normalL); st3(qpos + 3 * i, h.hit);
        } else {
            st3(nrm + 3 * i, mk(0, 0, 1)); st3(nrmL + 3 * i, mk(0, 0, 1)); st3(qpos + 3 * i, mk(3e18f, 3e18f, 3e18f));
        }
        if (flags) { flags[3 * i] = f0; flags[3 * i + 1] = f1; flags[3 * i + 2] = f2; }
    }
    block_add(ctr + PMC_HITS, nh);
}

// IsInShadow (final.cpp:985-1018) for every (hit, light sample): thread t -> hit t / S, sample t % S.  The 64 rays
// of a wave leave (nearly) one point towards neighbouring light samples, so the wave-synchronous any-hit walk is
// coherent; nothing but the occlusion byte is written.
extern "C" __global__ void __launch_bounds__(256)
k_pm_shadow(DevSceneView sc, const rt_hit* __restrict__ hits, const int32_t* __restrict__ sid,
            const float* __restrict__ nrmL, int64_t n, int S, float eps_f, int fast, uint8_t* __restrict__ occ,
            unsigned long long* __restrict__ ctr) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t / S;
    const int j = (int)(t - i * S);
    bool valid = false;
    Rayf r = {0, 0, 0, 0, 0, 1};
    float tLight = 0.f;
    if (i < n && sid[i] >= 0) {
        const float* hh = reinterpret_cast<const float*>(hits + i);
        if (reinterpret_cast<const int32_t*>(hh)[0] >= 0) {
            valid = true;
            const V3 nl = ld3(nrmL + 3 * i);
            const V3 point = add(mk(hh[2], hh[3], hh[4]), muls(nl, sc.shadow_bias));
            const V3 lp = ld3(sc.light_samples + 3 * (size_t)j);
            const V3 dl = sub(lp, point);
            const V3 dir = norm(dl);
            tLight = len(dl);
            r.ox = point.x; r.oy = point.y; r.oz = point.z; r.dx = dir.x; r.dy = dir.y; r.dz = dir.z;
        }
    }
    Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
    bool o = false;
    if (fast) traverse_fast<true>(sc, r, valid, eps_f, tLight, true, best, o);
    else traverse<true>(sc, r, valid, eps_f, tLight, true, best, o);
    loose_prims<true>(sc, r, valid, eps_f, tLight, true, best, o);
    if (i < n) occ[t] = o ? 1 : 0;
    // no counter here: one atomic per wave on one address (180 000 waves per launch) cost ten times the kernel's work;
    // the shadow rays are S per counted hit (PMC_HITS, same predicate), which the host multiplies out
    (void)ctr;
}

// caustic term (final.cpp:909-910) + direct illumination over every light's sample grid with the reference's
// function-scope diffuse / specular that a shadowed sample re-adds (final.cpp:822-823, :944-975), times the weight.
extern "C" __global__ void __launch_bounds__(256)
k_pm_direct(DevSceneView sc, const rt_hit* __restrict__ hits, const int32_t* __restrict__ sid,
            const float* __restrict__ w, const float* __restrict__ nrmL, int64_t n, int S,
            const uint8_t* __restrict__ occ, const float* __restrict__ caustic, float eyex, float eyey, float eyez,
            float* __restrict__ contrib) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float cr = 0.f, cg = 0.f, cb = 0.f;
    const float* hh = reinterpret_cast<const float*>(hits + i);
    const int prim = reinterpret_cast<const int32_t*>(hh)[0];
    if (sid[i] >= 0 && prim >= 0) {
        const DevMaterial m = sc.materials[sc.prim_material[prim]];
        const V3 hit = mk(hh[2], hh[3], hh[4]), nl = ld3(nrmL + 3 * i);
        cr = caustic[3 * i]; cg = caustic[3 * i + 1]; cb = caustic[3 * i + 2];
        V3 diffuse = mk(0, 0, 0), specular = mk(0, 0, 0);
        const V3 eyeVec = norm(sub(mk(eyex, eyey, eyez), hit));
        for (int l = 0; l < sc.nlights; ++l) {
            const int s0 = sc.light_sample_first[l], s1 = sc.light_sample_first[l + 1];
            const int ns = s1 - s0;
            const float factor = (float)(1.0 / (double)(ns >= 1 ? ns : 1));
            const V3 isurf = muls(ld3(m.surf), sc.lights[l].intensity);
            for (int j = s0; j < s1; ++j) {
                if (!occ[i * (int64_t)S + j]) {
                    const V3 toLight = norm(sub(ld3(sc.light_samples + 3 * (size_t)j), hit));
                    float d = max_std(dot(toLight, nl), 0.0f);
                    diffuse = muls(isurf, m.Kd * d);
                    const V3 halfVec = norm(add(eyeVec, toLight));
                    d = max_std(dot(halfVec, nl), 0.0f);
                    specular = muls(muls(isurf, m.Ks), powf(d, m.specularity));
                }
                const V3 a = muls(add(diffuse, specular), factor);
                cr += a.x; cg += a.y; cb += a.z;
            }
        }
        const float ww = w[i];
        cr *= ww; cg *= ww; cb *= ww;
    }
    contrib[3 * i] = cr; contrib[3 * i + 1] = cg; contrib[3 * i + 2] = cb;
}

// 50 cosine-weighted gather rays per eye-path hit, final.cpp:916-927; gather ray (i, g) -> slot i*G + g
extern "C" __global__ void __launch_bounds__(256)
k_pm_gather_spawn(const rt_hit* __restrict__ hits, const int32_t* __restrict__ sid, const float* __restrict__ w,
                  const uint32_t* __restrict__ code, const float* __restrict__ nrmL, const int32_t* __restrict__ pixmap,
                  int64_t n, int G, int ntap, uint64_t seed, rt_ray* __restrict__ grays, int32_t* __restrict__ gsid,
                  float* __restrict__ gw) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * G) return;
    const int64_t i = t / G;
    const int g = (int)(t - i * G);
    const float* hh = reinterpret_cast<const float*>(hits + i);
    const int s = sid[i];
    float* o = reinterpret_cast<float*>(grays + t);
    if (s < 0 || reinterpret_cast<const int32_t*>(hh)[0] < 0) {
        gsid[t] = -1; gw[t] = 0.f; o[0] = o[1] = o[2] = o[3] = o[4] = 0.f; o[5] = 1.f;
        return;
    }
    const uint64_t key = (uint64_t)pixmap[s / ntap] * 8ull + (uint64_t)(s % ntap);
    const double r1 = 2.0 * PM_PI * rt_rng(seed, key, code[i], 2ull * g);
    const double r2 = rt_rng(seed, key, code[i], 2ull * g + 1), r2s = sqrt(r2);
    const V3 wv = ld3(nrmL + 3 * i);
    const V3 u = norm(cross((double)__builtin_fabsf(wv.x) > 0.1 ? mk(0, 1.0f, 0) : mk(1.0f, 0, 0), wv));
    const V3 v = cross(wv, u);
    V3 d = add(add(muls(muls(u, (float)cos(r1)), (float)r2s), muls(muls(v, (float)sin(r1)), (float)r2s)),
               muls(wv, (float)sqrt(1 - r2)));
    d = norm(d);
    o[0] = hh[2]; o[1] = hh[3]; o[2] = hh[4]; o[3] = d.x; o[4] = d.y; o[5] = d.z;
    gsid[t] = s; gw[t] = w[i];
}

// TraceIrradiance's body for one traced gather ray, photon.cpp:756-815: kNN query (pos, un-flipped normal, Kd) for a
// diffuse hit, child flags (2 per ray).  A refracting hit returns only its two children (:808) -- its own terms and
// the reflect child are dropped, so kd = 0 and flags = {reflRay, tdir}; otherwise flags = {reflect child, 0}.
// `last` = children would exceed IRRADIANCE_MAX_DEPTH and return black: nothing is spawned.
extern "C" __global__ void __launch_bounds__(256)
k_gi_shade(DevSceneView sc, const rt_ray* __restrict__ rays, const rt_hit* __restrict__ hits,
           const int32_t* __restrict__ sid, int64_t n, int last, float start_nr, uint8_t* __restrict__ flags,
           float* __restrict__ qpos, float* __restrict__ qnrm, float* __restrict__ kd, unsigned long long* __restrict__ ctr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned traced = 0, look = 0;
    if (i < n) {
        uint8_t f0 = 0, f1 = 0;
        float k = 0.f;
        V3 qp = mk(3e18f, 3e18f, 3e18f), qn = mk(0, 0, 1);
        HitInfo h;
        if (sid[i] >= 0) {
            traced = 1;
            if (load_hit(sc, rays, hits, i, h)) {
                const DevMaterial m = sc.materials[sc.prim_material[h.prim]];
                if (m.refract > 0.01f) {
                    f0 = 1; f1 = 1;
                    if (last) {       // both children return black; the reference still multiplies by Re, Tr (:808)
                        V3 rr, td; float Re, Tr;
                        refract_terms(h.d, h.normal, h.normalL, start_nr, rr, td, Re, Tr);
                        k = 0.f * Re + 0.f * Tr;                   // 0, or NaN under total internal reflection
                    }
                } else {
                    if (m.Kd > 0) { k = m.Kd; qp = h.hit; qn = h.normal; look = 1; }
                    f0 = m.reflect > 0.01f;
                }
                if (last) { f0 = 0; f1 = 0; }
            }
        }
        st3(qpos + 3 * i, qp); st3(qnrm + 3 * i, qn); kd[i] = k;
        flags[2 * i] = f0; flags[2 * i + 1] = f1;
    }
    block_add(ctr + PMC_GATHER, traced);
    block_add(ctr + PMC_KNN_G, look);
}

// contribution of a gather ray's own hit: weight * (Kd * irradiance) / G   (photon.cpp:780-782, final.cpp:928).
// Always multiplied through, so a NaN weight (total internal reflection, photon.cpp:799-807) stays NaN as in the
// reference, also for a ray that hit nothing or a refracting surface whose children are cut by the depth limit.
extern "C" __global__ void __launch_bounds__(256)
k_gi_contrib(const int32_t* __restrict__ sid, const float* __restrict__ w, const float* __restrict__ kd,
             const float* __restrict__ irr, int64_t n, float inv_g, float* __restrict__ contrib) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (sid[i] < 0) { contrib[3 * i] = contrib[3 * i + 1] = contrib[3 * i + 2] = 0.f; return; }
    const float ww = w[i] * inv_g, k = kd[i];
    contrib[3 * i] = ww * (k * irr[3 * i]); contrib[3 * i + 1] = ww * (k * irr[3 * i + 1]); contrib[3 * i + 2] = ww * (k * irr[3 * i + 2]);
}

// ---- stable compaction of children (same tiling as k_scan_count: 8 flags per thread, 256 threads) ---------------
#define SP_ITEMS 8
#define SP_BLOCK 256
#define SP_TILE (SP_ITEMS * SP_BLOCK)

__device__ __forceinline__ uint32_t block_positions(const uint8_t* __restrict__ flags, int64_t nflags, int64_t base,
                                                    const uint32_t* __restrict__ block_offsets, uint8_t* f) {
    __shared__ uint32_t wsum[SP_BLOCK / 64];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SP_ITEMS; ++k) { f[k] = (base + k < nflags) ? flags[base + k] : 0; s += f[k]; }
    uint32_t incl = s;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    return block_offsets[blockIdx.x] + woff + incl - s;
}

// children of eye-path ray i: slot 3i reflect (final.cpp:843-851), 3i+1 refraction's reflRay (:877 / :894),
// 3i+2 refraction's tdir (:895).  code' = 4*code + 1 / 2 / 3.
extern "C" __global__ void __launch_bounds__(SP_BLOCK)
k_pm_spawn(DevSceneView sc, const rt_ray* __restrict__ rays, const rt_hit* __restrict__ hits,
           const int32_t* __restrict__ sid, const float* __restrict__ w, const uint32_t* __restrict__ code,
           const uint8_t* __restrict__ flags, int64_t nflags, const uint32_t* __restrict__ block_offsets, float start_nr,
           rt_ray* __restrict__ orays, int32_t* __restrict__ osid, float* __restrict__ ow, uint32_t* __restrict__ ocode,
           unsigned long long* __restrict__ ctr) {
    const int64_t base = (int64_t)blockIdx.x * SP_TILE + (int64_t)threadIdx.x * SP_ITEMS;
    uint8_t f[SP_ITEMS];
    uint32_t pos = block_positions(flags, nflags, base, block_offsets, f);
    unsigned nrefl = 0, nrefr = 0;
#pragma unroll
    for (int k = 0; k < SP_ITEMS; ++k) {
        if (!f[k]) continue;
        const int64_t slot = base + k, i = slot / 3;
        const int kind = (int)(slot - 3 * i);
        HitInfo h;
        load_hit(sc, rays, hits, i, h);
        const DevMaterial m = sc.materials[sc.prim_material[h.prim]];
        float* o = reinterpret_cast<float*>(orays + pos);
        V3 d; float wt;
        if (kind == 0) {
            const float dt = dot(h.d, h.normalL);
            d = sub(h.d, muls(h.normalL, 2 * dt));
            wt = w[i] * m.reflect; ++nrefl;
        } else {
            V3 rr, td; float Re, Tr;
            const float cos2t = refract_terms(h.d, h.normal, h.normalL, start_nr, rr, td, Re, Tr);
            if (kind == 1) { d = rr; wt = cos2t < 0 ? w[i] * m.refract : (w[i] * m.refract) * Re; }
            else { d = td; wt = (w[i] * m.refract) * Tr; }
            ++nrefr;
        }
        o[0] = h.hit.x; o[1] = h.hit.y; o[2] = h.hit.z; o[3] = d.x; o[4] = d.y; o[5] = d.z;
        osid[pos] = sid[i]; ow[pos] = wt; ocode[pos] = 4u * code[i] + 1u + (unsigned)kind;
        ++pos;
    }
    block_add(ctr + PMC_REFLECT, nrefl);
    block_add(ctr + PMC_REFRACT, nrefr);
}

// children of gather ray i (photon.cpp:785-809): refracting hit -> slot 2i reflRay * Re, 2i+1 tdir * Tr (both times
// `refract`); otherwise slot 2i = mirror reflection about the UN-flipped normal (:787-788) times `reflect`.
extern "C" __global__ void __launch_bounds__(SP_BLOCK)
k_gi_spawn(DevSceneView sc, const rt_ray* __restrict__ rays, const rt_hit* __restrict__ hits,
           const int32_t* __restrict__ sid, const float* __restrict__ w, const uint8_t* __restrict__ flags,
           int64_t nflags, const uint32_t* __restrict__ block_offsets, float start_nr, rt_ray* __restrict__ orays,
           int32_t* __restrict__ osid, float* __restrict__ ow) {
    const int64_t base = (int64_t)blockIdx.x * SP_TILE + (int64_t)threadIdx.x * SP_ITEMS;
    uint8_t f[SP_ITEMS];
    uint32_t pos = block_positions(flags, nflags, base, block_offsets, f);
#pragma unroll
    for (int k = 0; k < SP_ITEMS; ++k) {
        if (!f[k]) continue;
        const int64_t slot = base + k, i = slot >> 1;
        HitInfo h;
        load_hit(sc, rays, hits, i, h);
        const DevMaterial m = sc.materials[sc.prim_material[h.prim]];
        float* o = reinterpret_cast<float*>(orays + pos);
        V3 d; float wt;
        if (m.refract > 0.01f) {
            V3 rr, td; float Re, Tr;
            refract_terms(h.d, h.normal, h.normalL, start_nr, rr, td, Re, Tr);
            if ((slot & 1) == 0) { d = rr; wt = (w[i] * m.refract) * Re; }
            else { d = td; wt = (w[i] * m.refract) * Tr; }
        } else {
            const float dt = dot(h.d, h.normal);
            d = sub(h.d, muls(h.normal, 2 * dt));
            wt = w[i] * m.reflect;
        }
        o[0] = h.hit.x; o[1] = h.hit.y; o[2] = h.hit.z; o[3] = d.x; o[4] = d.y; o[5] = d.z;
        osid[pos] = sid[i]; ow[pos] = wt;
        ++pos;
    }
}

// pixel write of PhotonMappingImageSynthesis, final.cpp:1159-1198: taps combined in order with their filter
// weights; white when a tap saw the light; background unless a depth-0 ray hit.
__device__ __forceinline__ uint8_t quantize(float c) {
    if (c > 1.0f) return 255;
    const float v = c * 255.0f;
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return 0;     // cvttss2si's 0x80000000 -> low byte 0
    return (uint8_t)(int32_t)v;
}

extern "C" __global__ void __launch_bounds__(256)
k_pm_finalize(const int32_t* __restrict__ pixmap, int64_t nps, int ntap, int use_filter, const float* __restrict__ accum,
              const uint8_t* __restrict__ white, const uint8_t* __restrict__ hit_any, float bgr, float bgg, float bgb,
              uint8_t* __restrict__ rgb8, float* __restrict__ radiance) {
    const int64_t ps = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ps >= nps) return;
    const int p = pixmap[ps];
    if (p < 0) return;
    uint8_t* o = rgb8 + 3 * (size_t)p;
    float r = 0.f, g = 0.f, b = 0.f;
    if (white[ps]) { r = g = b = 1.0f; }
    else {
        for (int t = 0; t < ntap; ++t) {
            const float wt = use_filter ? AA_FILTER[t][2] : 1.0f;
            const float* a = accum + 3 * (size_t)(ps * ntap + t);
            r = r + wt * a[0]; g = g + wt * a[1]; b = b + wt * a[2];
        }
    }
    if (radiance) { radiance[3 * (size_t)p] = r; radiance[3 * (size_t)p + 1] = g; radiance[3 * (size_t)p + 2] = b; }
    if (white[ps] || hit_any[ps]) { o[0] = quantize(r); o[1] = quantize(g); o[2] = quantize(b); }
    else { o[0] = (uint8_t)bgr; o[1] = (uint8_t)bgg; o[2] = (uint8_t)bgb; }
}

// ---------------------------------------------------------------------------
static inline unsigned gridf(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

hipError_t rtk_pm_generate(const DevSceneView& sc, const rt_camera& cam, const int32_t* d_tiles, int64_t tile0, int64_t nt,
                           int ntap, int use_filter, float eps_f, rt_ray* rays, int32_t* sid, float* w, uint32_t* code,
                           int32_t* pixmap, uint8_t* white, uint8_t* hit_any, int32_t* hit_ids, float* accum,
                           unsigned long long* ctr, hipStream_t st) {
    if (nt <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pm_clear, dim3(gridf(nt * 64 * ntap, 256)), dim3(256), 0, st, nt * 64, ntap, white, hit_any, accum);
    hipLaunchKernelGGL(k_pm_generate, dim3(gridf(nt * ntap * 64, 256)), dim3(256), 0, st, sc, cam, d_tiles, tile0, nt, ntap,
                       use_filter, eps_f, rays, sid, w, code, pixmap, white, hit_any, hit_ids, ctr);
    return hipGetLastError();
}

hipError_t rtk_pm_shade(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* sid, int64_t n,
                        int ntap, int depth0, int spawn, float start_nr, uint8_t* flags, float* nrm, float* nrmL,
                        float* qpos, uint8_t* hit_any, int32_t* hit_ids, const int32_t* pixmap, unsigned long long* ctr,
                        hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pm_shade, dim3(gridf(n, 256)), dim3(256), 0, st, sc, rays, hits, sid, n, ntap, depth0, spawn,
                       start_nr, flags, nrm, nrmL, qpos, hit_any, hit_ids, pixmap, ctr);
    return hipGetLastError();
}

hipError_t rtk_pm_shadow(const DevSceneView& sc, const rt_hit* hits, const int32_t* sid, const float* nrmL, int64_t n,
                         int S, float eps_f, int fast, uint8_t* occ, unsigned long long* ctr, hipStream_t st) {
    if (n <= 0 || S <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pm_shadow, dim3(gridf(n * S, 256)), dim3(256), 0, st, sc, hits, sid, nrmL, n, S, eps_f, fast, occ, ctr);
    return hipGetLastError();
}

hipError_t rtk_pm_direct(const DevSceneView& sc, const rt_hit* hits, const int32_t* sid, const float* w, const float* nrmL,
                         int64_t n, int S, const uint8_t* occ, const float* caustic, const float eye[3], float* contrib,
                         hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pm_direct, dim3(gridf(n, 256)), dim3(256), 0, st, sc, hits, sid, w, nrmL, n, S, occ, caustic,
                       eye[0], eye[1], eye[2], contrib);
    return hipGetLastError();
}

hipError_t rtk_pm_gather_spawn(const rt_hit* hits, const int32_t* sid, const float* w, const uint32_t* code,
                               const float* nrmL, const int32_t* pixmap, int64_t n, int G, int ntap, uint64_t seed,
                               rt_ray* grays, int32_t* gsid, float* gw, hipStream_t st) {
    if (n <= 0 || G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pm_gather_spawn, dim3(gridf(n * G, 256)), dim3(256), 0, st, hits, sid, w, code, nrmL, pixmap, n, G,
                       ntap, seed, grays, gsid, gw);
    return hipGetLastError();
}

hipError_t rtk_gi_shade(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* sid, int64_t n,
                        int last, float start_nr, uint8_t* flags, float* qpos, float* qnrm, float* kd,
                        unsigned long long* ctr, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gi_shade, dim3(gridf(n, 256)), dim3(256), 0, st, sc, rays, hits, sid, n, last, start_nr, flags, qpos,
                       qnrm, kd, ctr);
    return hipGetLastError();
}

hipError_t rtk_gi_contrib(const int32_t* sid, const float* w, const float* kd, const float* irr, int64_t n, float inv_g,
                          float* contrib, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gi_contrib, dim3(gridf(n, 256)), dim3(256), 0, st, sid, w, kd, irr, n, inv_g, contrib);
    return hipGetLastError();
}

hipError_t rtk_pm_spawn(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* sid, const float* w,
                        const uint32_t* code, const uint8_t* flags, int64_t nflags, uint32_t* block_sums, uint32_t* d_total,
                        float start_nr, rt_ray* orays, int32_t* osid, float* ow, uint32_t* ocode, unsigned long long* ctr,
                        hipStream_t st) {
    if (nflags <= 0) return hipSuccess;
    hipError_t e = rtk_scan_flags(flags, nflags, block_sums, d_total, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pm_spawn, dim3((unsigned)rtk_scan_blocks_needed(nflags)), dim3(SP_BLOCK), 0, st, sc, rays, hits, sid,
                       w, code, flags, nflags, block_sums, start_nr, orays, osid, ow, ocode, ctr);
    return hipGetLastError();
}

hipError_t rtk_gi_spawn(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* sid, const float* w,
                        const uint8_t* flags, int64_t nflags, uint32_t* block_sums, uint32_t* d_total, float start_nr,
                        rt_ray* orays, int32_t* osid, float* ow, hipStream_t st) {
    // the caller has run rtk_scan_flags(flags, nflags, block_sums, d_total) and checked the total against its queues
    if (nflags <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gi_spawn, dim3((unsigned)rtk_scan_blocks_needed(nflags)), dim3(SP_BLOCK), 0, st, sc, rays, hits, sid, w,
                       flags, nflags, block_sums, start_nr, orays, osid, ow);
    return hipGetLastError();
}

hipError_t rtk_pm_finalize(const int32_t* pixmap, int64_t nps, int ntap, int use_filter, const float* accum,
                           const uint8_t* white, const uint8_t* hit_any, const float bg[3], uint8_t* rgb8, float* radiance,
                           hipStream_t st) {
    if (nps <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pm_finalize, dim3(gridf(nps, 256)), dim3(256), 0, st, pixmap, nps, ntap, use_filter, accum, white,
                       hit_any, bg[0], bg[1], bg[2], rgb8, radiance);
    return hipGetLastError();
}
