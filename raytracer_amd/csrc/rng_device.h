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
