// kernels_emit.hip -- photon emission on gfx950.
//
// Replaces the emission loops of PhotonMappingImageSynthesis (Final/photon/final.cpp:1057-1119),
// PhotonTrace (photon.cpp:445-568), PhotonTraceCaustics (:570-710) and AddPhoton (:835-885).
//
// One emission index per lane.  The reference recurses; a photon walk is a chain except for the total
// internal reflection case of PhotonTrace, which continues twice (:524-541), so every lane keeps a small
// LIFO of pending continuations (LDS, [slot][lane]) and pops until it is empty.  The closest-hit search is
// the wave-synchronous traversal of trace_device.h: all 64 lanes step together, lanes whose walk has ended
// ride along with valid = false.  Random numbers come from the counter-based generator of rng_device.h:
// emission index k owns the stream (which << 60 | k), so the photons of k do not depend on which lane,
// wave, launch or GPU traced it.
//
// Photons are returned in emission order: pass 0 counts the photons each emission stores, a scan turns the
// counts into offsets, pass 1 walks again and writes.
#include "trace_device.h"
#include "rng_device.h"

#define EMIT_STACK 8          // photon.cpp: depth < 6, a level pushes at most 2 and pops 1
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
__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// the float overloads the reference calls (std::cos(float) ...): evaluated in double and rounded, which is the
// correctly rounded float result in all but ~1e-8 of the cases
__device__ __forceinline__ float cosf_cr(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float sinf_cr(float x) { return (float)sin((double)x); }

struct Pending { V3 o, d, col; int depth; };

__device__ __forceinline__ void push(float* stk, int& sp, int lane, const Pending& p) {
    float* s = stk + (size_t)sp * 10 * 64 + lane;
    s[0 * 64] = p.o.x; s[1 * 64] = p.o.y; s[2 * 64] = p.o.z;
    s[3 * 64] = p.d.x; s[4 * 64] = p.d.y; s[5 * 64] = p.d.z;
    s[6 * 64] = p.col.x; s[7 * 64] = p.col.y; s[8 * 64] = p.col.z;
    s[9 * 64] = __int_as_float(p.depth);
    ++sp;
}
__device__ __forceinline__ Pending pop(const float* stk, int& sp, int lane) {
    --sp;
    const float* s = stk + (size_t)sp * 10 * 64 + lane;
    Pending p;
    p.o = mk(s[0 * 64], s[1 * 64], s[2 * 64]); p.d = mk(s[3 * 64], s[4 * 64], s[5 * 64]);
    p.col = mk(s[6 * 64], s[7 * 64], s[8 * 64]); p.depth = __float_as_int(s[9 * 64]);
    return p;
}

// the refraction block shared by photon.cpp:521-534 / :674-687: reflRay, tdir, Re, Tr (NaN-laden when cos2t < 0)
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

// AddPhoton's record, photon.cpp:846-882 (std::atan2(float,float) / std::acos(float))
__device__ __forceinline__ void store_photon(rt_photon* out, V3 pos, V3 inDir, V3 power) {
    inDir = norm(inDir);
    int phi = (int)((double)(float)atan2((double)inDir.y, (double)inDir.x) * (256.0 / (2.0 * 3.14159265358979323846)));
    int theta = (int)((double)(float)acos((double)inDir.z) * (256.0 / PM_PI));
    if (phi > 255) phi = 255;
    if (phi < 0) phi = (unsigned char)(phi + 256);
    if (theta > 255) theta = 255;
    float* f = reinterpret_cast<float*>(out);
    f[0] = pos.x; f[1] = pos.y; f[2] = pos.z; f[3] = inDir.x; f[4] = inDir.y; f[5] = inDir.z;
    f[6] = power.x; f[7] = power.y; f[8] = power.z;
    reinterpret_cast<uint32_t*>(out)[9] = (uint32_t)(phi & 255) | ((uint32_t)(theta & 255) << 8);   // phi, theta, plane = 0
}

}  // namespace

// write == 0: counts[k] = photons emission k stores.   write == 1: offsets[k] = where they go in `out`.
extern "C" __global__ void __launch_bounds__(64)
k_emit_photons(DevSceneView sc, rt_emit_params e, int light, int64_t first, int64_t count, float eps_f, int fast,
               int write, uint8_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
               rt_photon* __restrict__ out, int64_t cap) {
    __shared__ float stk[EMIT_STACK * 10 * 64];
    const int lane = threadIdx.x;
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    const bool mine = k < count;
    const uint64_t key = ((uint64_t)e.which << 60) | (uint64_t)(first + k);
    uint64_t draw = 0;
    int sp = 0;
    unsigned stored = 0;
    const int64_t base = (write && mine) ? (int64_t)offsets[k] : 0;
    const bool caustic = e.which != RT_PMAP_GLOBAL;
    int flag = 0;                                        // PhotonTraceCaustics' `flag`: a refracting surface was met
    if (mine) {
        // final.cpp:1060-1076 (= :1094-1110)
        const DevLight lt = sc.lights[light];
        const float offset_x = rt_rng(e.seed, key, draw++, 0), offset_y = rt_rng(e.seed, key, draw++, 0);
        const V3 d1 = muls(ld3(lt.dir1), lt.size1), d2 = muls(ld3(lt.dir2), lt.size2);
        const V3 pos = add(sub(add(sub(ld3(lt.pos), muls(d1, 0.5f)), muls(d1, offset_x)), muls(d2, 0.5f)), muls(d2, offset_y));
        const float r1 = (float)(2 * PM_PI * rt_rng(e.seed, key, draw++, 0)), r2 = rt_rng(e.seed, key, draw++, 0);
        const float r2s = sqrt_rn(r2);
        const V3 w = mk(0, -1.0f, 0), u = norm(cross(mk(1.0f, 0, 0), w)), v = cross(w, u);
        V3 d = add(add(muls(muls(u, cosf_cr(r1)), r2s), muls(muls(v, sinf_cr(r1)), r2s)), muls(w, sqrt_rn(1 - r2)));
        d = norm(d);
        Pending p; p.o = pos; p.d = d; p.col = mk(e.power[0], e.power[1], e.power[2]); p.depth = 0;
        push(stk, sp, lane, p);
    }
    // every lane pops one pending continuation per round; the wave leaves when all stacks are empty
    while (rt_ballot(sp > 0) != 0ull) {
        bool active = sp > 0;
        Pending p;
        if (active) p = pop(stk, sp, lane);
        else { p.o = mk(0, 0, 0); p.d = mk(0, 0, 1); p.col = mk(0, 0, 0); p.depth = 0; }
        if (active && !caustic && p.depth >= e.max_depth) active = false;                    // photon.cpp:447
        if (active && caustic) {
            if (p.d.y > 0) active = false;                                                   // :573
            if (active && !flag) {                                                           // :576-588
                bool bb = false;
                const Rayf rr = {p.o.x, p.o.y, p.o.z, p.d.x, p.d.y, p.d.z};
                for (int i = 0; i < e.ncaustic_sphere && !bb; ++i) {
                    float px, py, pz;
                    bb = sphere_test(e.caustic_sphere[i], rr, px, py, pz);
                }
                if (!bb) active = false;
            }
            if (active && p.depth >= 2) active = false;                                      // :594
        }
        const Rayf r = {p.o.x, p.o.y, p.o.z, p.d.x, p.d.y, p.d.z};
        Closest best; best.len = 1e9f; best.prim = -1; best.px = best.py = best.pz = 0.f; best.rank = 0x7fffffff;
        bool occ = false;
        if (fast) traverse_fast<false>(sc, r, active, eps_f, 0.f, false, best, occ);
        else traverse<false>(sc, r, active, eps_f, 0.f, false, best, occ);
        loose_prims<false>(sc, r, active, eps_f, 0.f, false, best, occ);
        if (!active || best.prim < 0) continue;
        const DevMaterial m = sc.materials[sc.prim_material[best.prim]];
        const V3 hit = mk(best.px, best.py, best.pz);
        V3 normal;
        if (sc.prim_type[best.prim] == RT_PRIM_SPHERE) {
            const float* g = sc.prim_geom + 4 * (size_t)best.prim;
            normal = norm(mk(hit.x - g[0], hit.y - g[1], hit.z - g[2]));
        } else {
            normal = ld3(sc.prim_normal + 3 * (size_t)best.prim);
        }
        V3 normalL = normal;
        if (dot(normal, p.d) > 0) normalL = muls(normalL, -1.0f);
        if (!caustic) {
            // PhotonTrace, photon.cpp:500-560
            const float REFLECT_PROB = m.reflect, REFRACT_PROB = m.refract;
            const double ap = 1.0 - (double)REFLECT_PROB - (double)REFRACT_PROB;
            const float ABSORP_PROB = (float)(ap > 0 ? ap : 0);
            const float rr = rt_rng(e.seed, key, draw++, 0);
            if (rr < REFLECT_PROB) {
                const float dt = dot(p.d, normal);
                Pending q; q.o = hit; q.d = sub(p.d, muls(normal, 2 * dt)); q.col = p.col; q.depth = p.depth + 1;
                draw++;                                        // russianRoulette = mRandom()*(wd+ws): drawn, unused (:513)
                push(stk, sp, lane, q);
            } else if (rr < REFLECT_PROB + REFRACT_PROB) {
                V3 reflRay, tdir; float Re, Tr;
                const float cos2t = refract_terms(p.d, normal, normalL, e.start_nr, reflRay, tdir, Re, Tr);
                const float P = (float)(.25 + .5 * Re), RP = Re / P, TP = Tr / (1 - P);
                // LIFO: the continuation the reference runs second is pushed first
                Pending q2; q2.o = hit; q2.depth = p.depth + 1;
                if (rr > REFLECT_PROB) { q2.d = reflRay; q2.col = muls(p.col, RP); }
                else { q2.d = tdir; q2.col = muls(p.col, TP); }
                push(stk, sp, lane, q2);
                if (cos2t < 0) {                               // :524-527, no return: both continuations run
                    Pending q1; q1.o = hit; q1.d = reflRay; q1.col = p.col; q1.depth = p.depth + 1;
                    push(stk, sp, lane, q1);
                }
            } else {
                const V3 power = mk(m.surf[0] * p.col.x / ABSORP_PROB, m.surf[1] * p.col.y / ABSORP_PROB, m.surf[2] * p.col.z / ABSORP_PROB);
                if (write && base + stored < cap) store_photon(out + base + stored, hit, p.d, power);
                ++stored;
            }
        } else {
            // PhotonTraceCaustics, photon.cpp:635-702
            if (!flag && (double)m.refract < 0.01) continue;
            flag = 1;
            const float rr = rt_rng(e.seed, key, draw++, 0);
            if (rr < m.refract) {
                V3 reflRay, tdir; float Re, Tr;
                refract_terms(p.d, normal, normalL, e.start_nr, reflRay, tdir, Re, Tr);
                Pending q; q.o = hit; q.d = tdir; q.col = p.col; q.depth = p.depth + 1;
                push(stk, sp, lane, q);
            } else {
                const V3 power = mk(m.surf[0] * p.col.x, m.surf[1] * p.col.y, m.surf[2] * p.col.z);
                if (!(hit.y > e.caustic_floor_y)) {
                    if (write && base + stored < cap) store_photon(out + base + stored, hit, p.d, power);
                    ++stored;
                }
            }
        }
    }
    if (mine && !write) counts[k] = (uint8_t)(stored > 255u ? 255u : stored);
}

// exclusive prefix of the per-emission counts: offsets[i] = block_offsets[block] + prefix inside the block
extern "C" __global__ void __launch_bounds__(256)
k_count_offsets(const uint8_t* __restrict__ counts, int64_t n, const uint32_t* __restrict__ block_offsets,
                uint32_t* __restrict__ offsets) {
    __shared__ uint32_t wsum[4];
    const int items = 8;
    const int64_t base = (int64_t)blockIdx.x * (256 * items) + (int64_t)threadIdx.x * items;
    uint32_t c[items], s = 0;
#pragma unroll
    for (int k = 0; k < items; ++k) { c[k] = (base + k < n) ? counts[base + k] : 0; s += c[k]; }
    uint32_t incl = s;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    uint32_t pos = block_offsets[blockIdx.x] + woff + incl - s;
#pragma unroll
    for (int k = 0; k < items; ++k) { if (base + k < n) offsets[base + k] = pos; pos += c[k]; }
}

hipError_t rtk_emit_photons(const DevSceneView& sc, const rt_emit_params& e, int64_t first, int64_t count, float eps_f,
                            int fast, int write, uint8_t* d_counts, const uint32_t* d_offsets, rt_photon* d_out, int64_t cap,
                            hipStream_t st) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_emit_photons, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st, sc, e, e.light, first, count,
                       eps_f, fast, write, d_counts, d_offsets, d_out, cap);
    return hipGetLastError();
}

hipError_t rtk_count_offsets(const uint8_t* d_counts, int64_t n, const uint32_t* d_block_offsets, uint32_t* d_offsets,
                             hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_count_offsets, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, st, d_counts, n, d_block_offsets,
                       d_offsets);
    return hipGetLastError();
}
