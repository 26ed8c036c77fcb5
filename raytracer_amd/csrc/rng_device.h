// rng_device.h -- the build-defined counter-based random number generator and the few
// algebra3 helpers the photon kernels share.  The reference draws from one serial
// std::default_random_engine seeded with time() (Final/photon/photon.h:16-18), which a parallel
// program cannot replay; here every consumer owns a stream keyed by (seed, a, b, c):
//   emission:  a = which << 60 | emission index,  b = draw number,  c = 0
//   gather:    a = sample id (pixel * 8 + AA tap),  b = position in the recursion tree,  c = 2*i + {0,1}
// 24 random bits -> [0,1), the value grid generate_canonical<float,24> delivers.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

__host__ __device__ inline uint64_t rt_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}

__host__ __device__ inline float rt_rng(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
    uint64_t x = rt_mix64(seed + 0x9E3779B97F4A7C15ull * (a + 1));
    x = rt_mix64(x ^ (b * 0xC2B2AE3D27D4EB4Full));
    x = rt_mix64(x ^ (c * 0x165667B19E3779F9ull));
    return (float)(x >> 40) * (1.0f / 16777216.0f);
}

// ---- distributed-ray-tracing samplers (build-defined; restated in oracle/rt_oracle.c) ----
// key a = pixel << 16 | sample;  b = position in the recursion (root 1, reflect 2c, refract 2c+1; 0 = camera);
// c = dimension: camera 0,1 pixel jitter, 2,3 lens;  hit 8..12 glossy ball, 16+8l..20+8l ball of light l.
#define RT_TWO_PI_D 6.283185307179586

// uniform point in the unit ball from 5 draws: direction from (u, v), radius = max of three uniforms (cdf r^3)
__device__ inline void rt_ball(uint64_t seed, uint64_t a, uint64_t b, uint64_t c0, float* out3) {
    const float u = rt_rng(seed, a, b, c0), v = rt_rng(seed, a, b, c0 + 1);
    float w = rt_rng(seed, a, b, c0 + 2);
    const float w2 = rt_rng(seed, a, b, c0 + 3), w3 = rt_rng(seed, a, b, c0 + 4);
    w = w < w2 ? w2 : w; w = w < w3 ? w3 : w;
    const float z = 1.0f - 2.0f * u;
    const float rr = 1.0f - z * z;
    const float r = __builtin_sqrtf(rr < 0.0f ? 0.0f : rr);
    const double phi = RT_TWO_PI_D * (double)v;
    out3[0] = w * (r * (float)cos(phi)); out3[1] = w * (r * (float)sin(phi)); out3[2] = w * z;
}
