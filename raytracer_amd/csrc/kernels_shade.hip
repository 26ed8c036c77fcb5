// kernels_shade.hip -- wavefront shading for the Whitted renderer.
//
// Replaces the shading block of SimpleMeshRayTrace (HW4/hw4/hw4.cpp:881-951)
// and the per-pixel quantisation of GenerateColorImage (hw4.cpp:1108-1122).
// The reference recurses (reflect, then refract, then local terms); here every
// ray carries a scalar weight (the product of the reflect / refract factors
// above it) and each pass of the wavefront handles one recursion depth:
//
//   trace_closest -> k_shade_flags  (normal, which children exist, shadow rays)
//                 -> trace_any      (one shadow ray per hit and light)
//                 -> k_shade_lights (ambient + Blinn-Phong, weight applied)
//                 -> k_accumulate   (ordered per-pixel sum: deterministic)
//                 -> scan + k_spawn (stable compaction of the children)
//
// Rays of one pixel stay contiguous and ordered through the stable compaction,
// so the per-pixel sums are evaluated in a fixed order: the image does not
// depend on how the frame is tiled over GPUs.
#include <hip/hip_runtime.h>
#include "rt_internal.h"
#include "shade_device.h"
#include "rng_device.h"


// counters[0]=valid rays in, [1]=hits, [3]=reflect children, [4]=refract children.
// No global atomics: all waves of a frame adding to one address serialise (2 atomics per wave
// cost 1.5 ms on a 4K frame).  Each block writes its pair of sums; k_reduce_pairs adds them up.
__device__ __forceinline__ void block_count2(unsigned long long* ctr /* per-block pairs */, bool p0, bool p1) {
    __shared__ unsigned int part[2];
    if (threadIdx.x < 2) part[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
    if ((threadIdx.x & 63) == 0) {
        if (m0) atomicAdd(&part[0], (unsigned)__builtin_popcountll(m0));
        if (m1) atomicAdd(&part[1], (unsigned)__builtin_popcountll(m1));
    }
    __syncthreads();
    if (threadIdx.x < 2) reinterpret_cast<unsigned int*>(ctr)[2 * blockIdx.x + threadIdx.x] = part[threadIdx.x];
}

// sums the per-block pairs written by block_count2 into counters[0], counters[1]
extern "C" __global__ void __launch_bounds__(256)
k_reduce_pairs(const unsigned int* __restrict__ pairs, int64_t nblocks, unsigned long long* __restrict__ counters) {
    __shared__ unsigned long long red[2][4];                           // 256 lanes: four waves find a place on a busy chip, sixteen wait for one
    unsigned long long a = 0, b = 0;
    for (int64_t i = threadIdx.x; i < nblocks; i += 256) { const uint2 v = reinterpret_cast<const uint2*>(pairs)[i]; a += v.x; b += v.y; }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned long long t = 0;
        for (int w = 0; w < 4; ++w) t += red[threadIdx.x][w];
        counters[threadIdx.x] += t;
    }
}

// Per traced ray: child flags (2 per ray), one biased shadow ray per light, and for the
// primary pass the hit id / "rayHitAnything" image (hw4.cpp:883-884).
extern "C" __global__ void __launch_bounds__(256)
k_shade_flags(DevSceneView sc, const rt_ray* __restrict__ rays, const rt_hit* __restrict__ hits,
              const int32_t* __restrict__ pix, int64_t n, int spawn_children, int primary_pass,
              uint8_t* __restrict__ child_flags, rt_ray* __restrict__ srays, int32_t* __restrict__ sactive,
              int32_t* __restrict__ hit_ids, uint8_t* __restrict__ hit_any, unsigned long long* __restrict__ block_pairs,
              const uint32_t* __restrict__ code, uint64_t seed, float light_radius) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = i < n;
    const int px = inb ? pix[i] : -1;
    HitCtx c;
    const bool hit = (px >= 0) && load_hit(sc, rays, hits, i, c);
    block_count2(block_pairs, px >= 0, hit);
    if (!inb) return;
    const int L = sc.nlights;
    if (primary_pass && px >= 0) {
        // with several samples per pixel the id image shows sample 0, and the pixel is "hit" if any sample hit
        // (k_zero_pixels cleared the flag)
        if (hit_ids && (!code || (code[i] >> 16) == 0)) hit_ids[px] = hit ? c.prim : -1;
        if (hit) hit_any[px] = 1;
    }
    uint8_t f0 = 0, f1 = 0;
    if (hit) {
        const DevMaterial m = sc.materials[c.mat];
        if (spawn_children) {
            f0 = m.reflect > 0.0001f;                         // hw4.cpp:897
            f1 = (double)m.refract > 0.0001;                  // hw4.cpp:907 (double literal)
        }
        for (int l = 0; l < L; ++l) {
            float lx, ly, lz;
            light_point(sc, l, code != nullptr, light_radius, seed, px, code ? code[i] : 0u, lx, ly, lz);
            shadow_ray(sc, c, lx, ly, lz, reinterpret_cast<float*>(srays + (i * L + l)));
            sactive[i * L + l] = 0;
        }
    } else {
        for (int l = 0; l < L; ++l) sactive[i * L + l] = -1;
    }
    if (child_flags) { child_flags[2 * i] = f0; child_flags[2 * i + 1] = f1; }
}

// ambient + per-light Blinn-Phong, hw4.cpp:926-948, including the reference's quirk that
// `diffuse`/`specular` keep the previous light's values when a light is shadowed (:876-877,:947).
extern "C" __global__ void __launch_bounds__(256)
k_shade_lights(DevSceneView sc, const rt_ray* __restrict__ rays, const rt_hit* __restrict__ hits,
               const int32_t* __restrict__ pix, const float* __restrict__ weight, int64_t n,
               const uint8_t* __restrict__ occluded, float ambient_intensity, float eyex, float eyey, float eyez,
               float* __restrict__ contrib, unsigned long long* __restrict__ counters) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool inb = i < n;
    float cr = 0.f, cg = 0.f, cb = 0.f;
    HitCtx c;
    const bool hit = inb && pix[i] >= 0 && load_hit(sc, rays, hits, i, c);
    if (!inb) return;
    if (hit) {
        const int L = sc.nlights;
        shade_local(sc, c, ambient_intensity, eyex, eyey, eyez, [&](int l) { return occluded[i * L + l] != 0; }, cr, cg, cb);
        const float w = weight ? weight[i] : 1.0f;
        cr *= w; cg *= w; cb *= w;
    }
    contrib[3 * i] = cr; contrib[3 * i + 1] = cg; contrib[3 * i + 2] = cb;
}

// Ordered per-pixel accumulation: the first ray of each run of equal pixel ids adds the run, in index order (the same
// additions in the same order whatever the launch shape, so frames are byte-identical under tiling and chunking).  The
// block's 256 contributions and pixel ids are staged in LDS by coalesced loads, so a run leader's serial sum reads LDS,
// eight contributions at a time, not one dependent global round trip per sample; the part of a run that lies beyond the
// block is read from global memory as before.
extern "C" __global__ void __launch_bounds__(256)
k_accumulate(const int32_t* __restrict__ pix, const float* __restrict__ contrib, int64_t n, float* __restrict__ accum) {
    __shared__ float s_c[3 * 256];
    __shared__ int32_t s_p[256];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * 256;
    const int64_t i = base + tid;
    const int p = i < n ? pix[i] : -1;
    s_p[tid] = p;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int64_t k = 3 * base + t * 256 + tid;
        if (k < 3 * n) s_c[t * 256 + tid] = contrib[k];
    }
    __syncthreads();
    if (p < 0) return;
    if (tid > 0 ? s_p[tid - 1] == p : (i > 0 && pix[i - 1] == p)) return;
    float r = accum[3 * (size_t)p], g = accum[3 * (size_t)p + 1], b = accum[3 * (size_t)p + 2];
    int t = tid;
    while (t + 7 < 256 && s_p[t + 7] == p) {       // equal pixel ids are contiguous: s_p[t + 7] == p covers the seven before it
        float c[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) c[u] = s_c[3 * t + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) { r += c[3 * u]; g += c[3 * u + 1]; b += c[3 * u + 2]; }
        t += 8;
    }
    for (; t < 256 && s_p[t] == p; ++t) { r += s_c[3 * t]; g += s_c[3 * t + 1]; b += s_c[3 * t + 2]; }
    if (t == 256) {                                // the run goes on in the next block(s)
        int64_t j = base + 256;
        while (j + 7 < n && pix[j + 7] == p) {
            float c[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) c[u] = contrib[3 * j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { r += c[3 * u]; g += c[3 * u + 1]; b += c[3 * u + 2]; }
            j += 8;
        }
        for (; j < n && pix[j] == p; ++j) { r += contrib[3 * j]; g += contrib[3 * j + 1]; b += contrib[3 * j + 2]; }
    }
    accum[3 * (size_t)p] = r; accum[3 * (size_t)p + 1] = g; accum[3 * (size_t)p + 2] = b;
}

// ---- stable compaction of the children ------------------------------------
#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

extern "C" __global__ void __launch_bounds__(SCAN_BLOCK)
k_scan_count(const uint8_t* __restrict__ flags, int64_t n, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t red[SCAN_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) s += flags[base + k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += red[w]; block_sums[blockIdx.x] = t; }
}

// single block: exclusive scan of block_sums in place, total -> *total.  256 lanes x 16 consecutive entries per trip, wave
// scans by shuffles: a block of four waves finds a place on a chip that a traversal kernel of the other stream has filled
// (the 1024-lane block this replaces waited 0.2 - 0.5 ms for sixteen free wave slots on one CU, three times per frame)
#define SCANB_ITEMS 16
extern "C" __global__ void __launch_bounds__(256)
k_scan_blocks(uint32_t* __restrict__ block_sums, int64_t nblocks, uint32_t* __restrict__ total) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t carry = 0;                                                    // (every lane keeps the same running total)
    for (int64_t base = 0; base < nblocks; base += 256 * SCANB_ITEMS) {
        const int64_t i0 = base + (int64_t)threadIdx.x * SCANB_ITEMS;
        uint32_t v[SCANB_ITEMS];
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < SCANB_ITEMS; ++k) { v[k] = i0 + k < nblocks ? block_sums[i0 + k] : 0u; s += v[k]; }
        uint32_t incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t before = carry, all = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint32_t t = wsum[w]; if (w < wv) before += t; all += t; }
        uint32_t run = before + incl - s;
#pragma unroll
        for (int k = 0; k < SCANB_ITEMS; ++k) { if (i0 + k < nblocks) block_sums[i0 + k] = run; run += v[k]; }
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// Children of ray i: slot 2i = reflection (hw4.cpp:897-904), slot 2i+1 = refraction (:907-924).
extern "C" __global__ void __launch_bounds__(SCAN_BLOCK)
k_spawn(DevSceneView sc, const rt_ray* __restrict__ rays, const rt_hit* __restrict__ hits,
        const int32_t* __restrict__ pix, const float* __restrict__ weight, const float* __restrict__ nr_in,
        const uint8_t* __restrict__ flags, int64_t nflags, const uint32_t* __restrict__ block_offsets,
        rt_ray* __restrict__ orays, int32_t* __restrict__ opix, float* __restrict__ oweight, float* __restrict__ onr,
        unsigned int* __restrict__ kind_pairs, const uint32_t* __restrict__ code, uint32_t* __restrict__ ocode,
        uint64_t seed, float glossy, uint8_t* __restrict__ ooct, int32_t* __restrict__ osrc) {
    __shared__ uint32_t wsum[SCAN_BLOCK / 64];
    __shared__ unsigned int kinds[2];
    __shared__ uint16_t list[SCAN_TILE];                                 // the tile's flagged slots (tile-local index), in slot order
    if (threadIdx.x < 2) kinds[threadIdx.x] = 0;
    const int64_t tile0 = (int64_t)blockIdx.x * SCAN_TILE;
    const int64_t base = tile0 + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint8_t f[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { f[k] = (base + k < nflags) ? flags[base + k] : 0; s += f[k]; }
    // exclusive scan of s over the block
    uint32_t incl = s;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t woff = 0, tile_total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) { const uint32_t t = wsum[w]; if (w < (int)(threadIdx.x >> 6)) woff += t; tile_total += t; }
    // Children are made from the COMPACTED list, one per lane: about a fifth of the slots of config 3's first pass carry a child,
    // and made where they stand -- eight slots per lane, one after the other -- every wave ran the child's arithmetic (glossy lobe
    // in double, hit context, eight stores) eight times with a fifth of its lanes.  Same children at the same positions: entry j
    // of the tile's list goes to block_offsets[tile] + j, which is where its slot's rank put it before.
    {
        uint32_t lp = woff + incl - s;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) if (f[k]) list[lp++] = (uint16_t)(threadIdx.x * SCAN_ITEMS + k);
    }
    __syncthreads();
    const uint32_t pos0 = block_offsets[blockIdx.x];
    int nrefl = 0, nrefr = 0;
    for (uint32_t j = threadIdx.x; j < tile_total; j += SCAN_BLOCK) {
        const int64_t slot = tile0 + (int64_t)list[j], i = slot >> 1;
        const uint32_t pos = pos0 + j;
        HitCtx c;
        load_hit(sc, rays, hits, i, c);
        const DevMaterial m = sc.materials[c.mat];
        const float w = weight ? weight[i] : 1.0f;
        float* o = reinterpret_cast<float*>(orays + pos);
        if ((slot & 1) == 0) {
            reflect_child(c, glossy, code != nullptr, seed, pix[i], code ? code[i] : 0u, o);
            oweight[pos] = w * m.reflect; onr[pos] = nr_in ? nr_in[i] : 1.0f;
            ++nrefl;
        } else {
            const float n1 = nr_in ? nr_in[i] : 1.0f, n2 = m.Nr;
            refract_child(sc, c, n1, n2, o);
            oweight[pos] = w * m.refract; onr[pos] = n2;
            ++nrefr;
        }
        opix[pos] = pix[i];
        if (ooct) ooct[pos] = (uint8_t)((o[3] < 0.f ? 1u : 0u) | (o[4] < 0.f ? 2u : 0u) | (o[5] < 0.f ? 4u : 0u));   // k_ray_keys' octant
        if (ocode) ocode[pos] = child_code(code[i], (unsigned)(slot & 1));
        if (osrc) osrc[pos] = c.prim;                                       // the triangle the child starts on (RaySrc)
    }
    if (nrefl) atomicAdd(&kinds[0], (unsigned)nrefl);
    if (nrefr) atomicAdd(&kinds[1], (unsigned)nrefr);
    __syncthreads();
    // the block's two counts go to its own pair; k_reduce_pairs adds them up (one global atomic per block on one word --
    // 130 000 of them for the 132.7 M rays of config 3's first pass -- cost more than the compaction itself)
    if (threadIdx.x < 2) kind_pairs[2 * blockIdx.x + threadIdx.x] = kinds[threadIdx.x];
}

// ---- stable partition of a queue by direction octant (the order in which a wave-synchronous pass takes its rays) -------
// k_spawn leaves one octant byte per child; the order is made from those bytes alone -- count per tile and octant, one
// scan over (octant-major) tile counts, scatter -- 50 MB read twice and 200 MB written for 49.5 M rays, where keys +
// radix sort of (key, index) pairs moved 3 GB and took 10 ms beside the shadow kernel (profiles/r02_timeline_drt_1m.txt).
// Packed counters: four 16-bit fields per 64-bit word (a tile has 2 048 entries), octants 0-3 in `a`, 4-7 in `b`.
__device__ __forceinline__ void oct_tile_counts(const uint8_t* __restrict__ oct, int64_t n, int64_t base, uint8_t o8[SCAN_ITEMS],
                                                unsigned long long& a, unsigned long long& b) {
    a = 0ull; b = 0ull;
    unsigned long long v = 0x0808080808080808ull;                  // 8 = no entry
    if (base + SCAN_ITEMS <= n) v = *reinterpret_cast<const unsigned long long*>(oct + base);
    else for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) v = (v & ~(0xffull << (8 * k))) | ((unsigned long long)oct[base + k] << (8 * k));
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const unsigned o = (unsigned)(v >> (8 * k)) & 0xffu;
        o8[k] = (uint8_t)o;
        if (o < 4u) a += 1ull << (16 * o); else if (o < 8u) b += 1ull << (16 * (o - 4u));
    }
}

extern "C" __global__ void __launch_bounds__(SCAN_BLOCK)
k_oct_count(const uint8_t* __restrict__ oct, int64_t n, int64_t nblocks, uint32_t* __restrict__ hist /* [8][nblocks] */) {
    __shared__ unsigned long long red[2][SCAN_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint8_t o8[SCAN_ITEMS]; unsigned long long a, b;
    oct_tile_counts(oct, n, base, o8, a, b);
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x < 8) {
        unsigned long long t = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += red[threadIdx.x >> 2][w];
        hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = (uint32_t)(t >> (16 * (threadIdx.x & 3))) & 0xffffu;
    }
}

extern "C" __global__ void __launch_bounds__(SCAN_BLOCK)
k_oct_scatter(const uint8_t* __restrict__ oct, int64_t n, int64_t nblocks, const uint32_t* __restrict__ offs /* scanned hist */,
              int32_t* __restrict__ order) {
    __shared__ unsigned long long wsum[2][SCAN_BLOCK / 64];
    __shared__ uint32_t tile_off[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    if (threadIdx.x < 8) tile_off[threadIdx.x] = offs[(int64_t)threadIdx.x * nblocks + blockIdx.x];
    uint8_t o8[SCAN_ITEMS]; unsigned long long a, b;
    oct_tile_counts(oct, n, base, o8, a, b);
    // exclusive scan of the packed counts over the block: within the wave, then across the four waves
    unsigned long long ia = a, ib = b;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long ta = __shfl_up(ia, off), tb = __shfl_up(ib, off);
        if (lane >= off) { ia += ta; ib += tb; }
    }
    if (lane == 63) { wsum[0][threadIdx.x >> 6] = ia; wsum[1][threadIdx.x >> 6] = ib; }
    __syncthreads();
    unsigned long long ea = ia - a, eb = ib - b;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) { ea += wsum[0][w]; eb += wsum[1][w]; }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const unsigned o = o8[k];
        if (o >= 8u) continue;
        unsigned r;
        if (o < 4u) { r = (unsigned)(ea >> (16 * o)) & 0xffffu; ea += 1ull << (16 * o); }
        else { r = (unsigned)(eb >> (16 * (o - 4u))) & 0xffffu; eb += 1ull << (16 * (o - 4u)); }
        order[tile_off[o] + r] = (int32_t)(base + k);
    }
}

// GenerateColorImage's pixel write, hw4.cpp:1108-1122.  One thread per primary ray of this rank.
extern "C" __global__ void __launch_bounds__(256)
k_finalize(const int32_t* __restrict__ pix0, int64_t n, const float* __restrict__ accum,
           const uint8_t* __restrict__ hit_any, float bgr, float bgg, float bgb, uint8_t* __restrict__ rgb8, int64_t packed_base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pix0[i];
    if (p < 0) return;
    uint8_t* o = rgb8 + 3 * (size_t)(packed_base >= 0 ? packed_base + i : (int64_t)p);   // rt_params.packed_output
    if (hit_any[p]) {
        o[0] = quantize(accum[3 * (size_t)p]); o[1] = quantize(accum[3 * (size_t)p + 1]); o[2] = quantize(accum[3 * (size_t)p + 2]);
    } else {
        o[0] = (uint8_t)bgr; o[1] = (uint8_t)bgg; o[2] = (uint8_t)bgb;
    }
}

extern "C" __global__ void __launch_bounds__(256)
k_copy_radiance(const int32_t* __restrict__ pix0, int64_t n, const float* __restrict__ accum, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pix0[i];
    if (p < 0) return;
    out[3 * (size_t)p] = accum[3 * (size_t)p]; out[3 * (size_t)p + 1] = accum[3 * (size_t)p + 1]; out[3 * (size_t)p + 2] = accum[3 * (size_t)p + 2];
}

extern "C" __global__ void __launch_bounds__(256)
k_zero_pixels(const int32_t* __restrict__ pix0, int64_t n, float* __restrict__ accum, uint8_t* __restrict__ hit_any) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pix0[i];
    if (p < 0) return;
    hit_any[p] = 0;
    accum[3 * (size_t)p] = 0.f; accum[3 * (size_t)p + 1] = 0.f; accum[3 * (size_t)p + 2] = 0.f;
}

// ---------------------------------------------------------------------------
static inline unsigned gridf(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

hipError_t rtk_shade_flags(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* pix, int64_t n,
                           int spawn_children, int primary_pass, uint8_t* child_flags, rt_ray* srays, int32_t* sactive,
                           int32_t* hit_ids, uint8_t* hit_any, unsigned long long* counters, unsigned int* block_pairs,
                           const uint32_t* code, uint64_t seed, float light_radius, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const unsigned nb = gridf(n, 256);
    hipLaunchKernelGGL(k_shade_flags, dim3(nb), dim3(256), 0, st, sc, rays, hits, pix, n, spawn_children,
                       primary_pass, child_flags, srays, sactive, hit_ids, hit_any,
                       reinterpret_cast<unsigned long long*>(block_pairs), code, seed, light_radius);
    hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(256), 0, st, block_pairs, (int64_t)nb, counters);
    return hipGetLastError();
}

hipError_t rtk_reduce_pairs(const unsigned int* block_pairs, int64_t nblocks, unsigned long long* counters, hipStream_t st) {
    if (nblocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(256), 0, st, block_pairs, nblocks, counters);
    return hipGetLastError();
}

hipError_t rtk_shade_lights(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* pix,
                            const float* weight, int64_t n, const uint8_t* occluded, float ambient, const float eye[3],
                            float* contrib, unsigned long long* counters, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_shade_lights, dim3(gridf(n, 256)), dim3(256), 0, st, sc, rays, hits, pix, weight, n, occluded,
                       ambient, eye[0], eye[1], eye[2], contrib, counters);
    return hipGetLastError();
}

hipError_t rtk_accumulate(const int32_t* pix, const float* contrib, int64_t n, float* accum, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_accumulate, dim3(gridf(n, 256)), dim3(256), 0, st, pix, contrib, n, accum);
    return hipGetLastError();
}

int64_t rtk_scan_blocks_needed(int64_t nflags) { return (nflags + SCAN_TILE - 1) / SCAN_TILE; }

// byte flags -> per-2048-tile exclusive offsets in block_sums, total in *d_total
hipError_t rtk_scan_flags(const uint8_t* flags, int64_t n, uint32_t* block_sums, uint32_t* d_total, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t nb = rtk_scan_blocks_needed(n);
    hipLaunchKernelGGL(k_scan_count, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, flags, n, block_sums);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, st, block_sums, nb, d_total);
    return hipGetLastError();
}

hipError_t rtk_compact_children(const DevSceneView& sc, const rt_ray* rays, const rt_hit* hits, const int32_t* pix,
                                const float* weight, const float* nr_in, const uint8_t* flags, int64_t nflags,
                                uint32_t* block_sums, uint32_t* d_total, rt_ray* orays, int32_t* opix, float* oweight,
                                float* onr, unsigned long long* counters, const uint32_t* code, uint32_t* ocode,
                                uint64_t seed, float glossy, hipStream_t st, uint8_t* ooct, int32_t* osrc) {
    if (nflags <= 0) return hipSuccess;
    const int64_t nb = rtk_scan_blocks_needed(nflags);
    hipLaunchKernelGGL(k_scan_count, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, flags, nflags, block_sums);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, st, block_sums, nb, d_total);
    // block_sums: [nb] offsets, then (256-byte aligned) [nb][2] per-block child counts by kind
    unsigned int* kind_pairs = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(block_sums) + (((size_t)nb * 4 + 16 + 255) & ~(size_t)255));
    hipLaunchKernelGGL(k_spawn, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, sc, rays, hits, pix, weight, nr_in, flags,
                       nflags, block_sums, orays, opix, oweight, onr, kind_pairs, code, ocode, seed, glossy, ooct, osrc);
    hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(256), 0, st, kind_pairs, nb, counters + 3);      // [3] reflect, [4] refract children
    return hipGetLastError();
}

// order of work for a wave-synchronous pass: the queue's entries by direction octant, queue order kept inside an octant.
// d_oct: k_spawn's octant bytes; d_work: rtk_octant_order_workspace(n) bytes; *order_out points into it.
size_t rtk_octant_order_workspace(int64_t n) {
    const int64_t nb = rtk_scan_blocks_needed(n);
    return (size_t)n * 4 + 256 + (size_t)(8 * nb + 8) * 4 + 256;
}
hipError_t rtk_octant_order(const uint8_t* d_oct, int64_t n, void* d_work, const int32_t** order_out, hipStream_t st) {
    *order_out = nullptr;
    if (n < 4096 || n > 0x7fffffff || !d_work || !d_oct) return hipSuccess;
    const int64_t nb = rtk_scan_blocks_needed(n);
    int32_t* order = (int32_t*)d_work;
    uint32_t* hist = (uint32_t*)((char*)d_work + (((size_t)n * 4 + 255) & ~(size_t)255));
    hipLaunchKernelGGL(k_oct_count, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, d_oct, n, nb, hist);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, st, hist, 8 * nb, hist + 8 * nb);      // octant-major: a stable partition
    hipLaunchKernelGGL(k_oct_scatter, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, d_oct, n, nb, hist, order);
    *order_out = order;
    return hipGetLastError();
}

hipError_t rtk_finalize(const int32_t* pix0, int64_t n, const float* accum, const uint8_t* hit_any, const float bg[3],
                        uint8_t* rgb8, int64_t packed_base, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_finalize, dim3(gridf(n, 256)), dim3(256), 0, st, pix0, n, accum, hit_any, bg[0], bg[1], bg[2], rgb8, packed_base);
    return hipGetLastError();
}

hipError_t rtk_copy_radiance(const int32_t* pix0, int64_t n, const float* accum, float* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_copy_radiance, dim3(gridf(n, 256)), dim3(256), 0, st, pix0, n, accum, out);
    return hipGetLastError();
}

hipError_t rtk_zero_pixels(const int32_t* pix0, int64_t n, float* accum, uint8_t* hit_any, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_zero_pixels, dim3(gridf(n, 256)), dim3(256), 0, st, pix0, n, accum, hit_any);
    return hipGetLastError();
}
