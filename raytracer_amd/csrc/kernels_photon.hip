// kernels_photon.hip -- photon-map k-nearest-neighbour gather on gfx950.
//
// Replaces PhotonMap::LocatePhoton (Final/photon/photon.cpp:53-131) and the
// BALANCE_TREE branch of PhotonMap::CalcIrradiance (photon.cpp:391-441, with
// GetPhotonDirection :29-36).
//
// One query per lane, one 64-lane wave per workgroup.  The candidate list of
// the reference (np.dist2[0..k], np.index[1..k]: plain array until k photons
// are found, then a max-heap) lives in LDS as [slot][lane] rows, so every
// lane's accesses fall in its own bank whatever slot it touches.  The
// recursion of LocatePhoton is unrolled into a descent/ascent over the
// implicit heap numbering (children 2i, 2i+1; parent i>>1): the plane distance
// and the photon distance of every node on the current root path are kept in
// LDS per level, so a photon record is fetched from HBM exactly once per
// visit (16 B: position + packed plane/theta/phi).  Insertions replay the
// reference's array/heap code statement by statement, hence np.index comes
// out in the reference's order and the irradiance sum adds in its order.
#include <hip/hip_runtime.h>
#include "rt_internal.h"

#define KNN_LEVELS 32

namespace {

struct KnnLds {
    float* hd;      // [(k+1)][64]  np.dist2
    int*   hi;      // [(k+1)][64]  np.index
    float* s1;      // [KNN_LEVELS][64] plane distance per level
    float* s2;      // [KNN_LEVELS][64] photon distance^2 per level
};

// the "we found a photon" block of LocatePhoton, photon.cpp:79-128
__device__ __forceinline__ void consider(const KnnLds& L, int t, int k, int& found, int& got_heap, int index, float dist2) {
#define HD(j) L.hd[(j) * 64 + t]
#define HI(j) L.hi[(j) * 64 + t]
    if (!(dist2 < HD(0))) return;
    if (found < k) {
        ++found;
        HD(found) = dist2; HI(found) = index;
        return;
    }
    int j, parent;
    if (got_heap == 0) {
        const int half_found = found >> 1;
        for (int kk = half_found; kk >= 1; --kk) {
            parent = kk;
            const int phot = HI(kk);
            const float dst2 = HD(kk);
            while (parent <= half_found) {
                j = parent + parent;
                if (j < found && HD(j) < HD(j + 1)) ++j;
                if (dst2 >= HD(j)) break;
                HD(parent) = HD(j); HI(parent) = HI(j);
                parent = j;
            }
            HD(parent) = dst2; HI(parent) = phot;
        }
        got_heap = 1;
    }
    parent = 1; j = 2;
    while (j <= found) {
        if (j < found && HD(j) < HD(j + 1)) ++j;
        if (dist2 > HD(j)) break;
        HD(parent) = HD(j); HI(parent) = HI(j);
        parent = j; j += j;
    }
    HI(parent) = index; HD(parent) = dist2;
    HD(0) = HD(1);
#undef HD
#undef HI
}

__device__ __forceinline__ void locate(const DevPhoton* __restrict__ ph, int64_t n, int64_t half, const KnnLds& L,
                                       int t, int k, float qx, float qy, float qz, int& found) {
    int got_heap = 0;
    found = 0;
    if (n < 1) return;
    int64_t idx = 1;
    for (;;) {
        // ---- descend from idx to a node without children, stashing the path ----
        float dist2;
        for (;;) {
            const float4 rec = *reinterpret_cast<const float4*>(ph + idx);       // pos.xyz + packed
            float d1 = rec.x - qx;
            dist2 = d1 * d1;
            d1 = rec.y - qy; dist2 += d1 * d1;
            d1 = rec.z - qz; dist2 += d1 * d1;
            if (idx < half) {
                const int plane = __float_as_int(rec.w) & 3;
                const float pp = plane == 0 ? rec.x : (plane == 1 ? rec.y : rec.z);
                const float qp = plane == 0 ? qx : (plane == 1 ? qy : qz);
                const float dist1 = qp - pp;
                const int lvl = 63 - __builtin_clzll((unsigned long long)idx);
                L.s1[lvl * 64 + t] = dist1; L.s2[lvl * 64 + t] = dist2;
                idx = (dist1 > 0.0f) ? 2 * idx + 1 : 2 * idx;
            } else break;
        }
        consider(L, t, k, found, got_heap, (int)idx, dist2);
        // ---- ascend until a far subtree has to be entered ----
        bool descend = false;
        while (idx > 1) {
            const int64_t child = idx;
            idx >>= 1;
            const int lvl = 63 - __builtin_clzll((unsigned long long)idx);
            const float dist1 = L.s1[lvl * 64 + t];
            const int64_t near = (dist1 > 0.0f) ? 2 * idx + 1 : 2 * idx;
            if (child == near && dist1 * dist1 < L.hd[t]) { idx = near ^ 1; descend = true; break; }
            consider(L, t, k, found, got_heap, (int)idx, L.s2[lvl * 64 + t]);
        }
        if (!descend) return;
    }
}

}  // namespace

// mode 0: write idx/dist2/found/radius2.   mode 1: irradiance rgb.
extern "C" __global__ void __launch_bounds__(64)
k_photon_knn(const DevPhoton* __restrict__ ph, int64_t n, int64_t half, const float* __restrict__ pos3,
             const float* __restrict__ nrm3, int64_t q, int k, float maxDist, int mode,
             int32_t* __restrict__ out_idx, float* __restrict__ out_d2, int32_t* __restrict__ out_found,
             float* __restrict__ out_r2, float* __restrict__ out_rgb, const float* __restrict__ tables) {
    extern __shared__ float lds[];
    KnnLds L;
    L.hd = lds;
    L.hi = reinterpret_cast<int*>(lds + (k + 1) * 64);
    L.s1 = lds + 2 * (k + 1) * 64;
    L.s2 = L.s1 + KNN_LEVELS * 64;
    const int t = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 64 + t;
    if (i >= q) return;
    const float qx = pos3[3 * i], qy = pos3[3 * i + 1], qz = pos3[3 * i + 2];
    L.hd[t] = maxDist * maxDist;                                   // np.dist2[0], photon.cpp:405
    int found;
    locate(ph, n, half, L, t, k, qx, qy, qz, found);
    if (mode == 0) {
        for (int j = 0; j < k; ++j) {
            out_idx[i * k + j] = j < found ? L.hi[(j + 1) * 64 + t] : -1;
            out_d2[i * k + j] = j < found ? L.hd[(j + 1) * 64 + t] : 0.0f;
        }
        out_found[i] = found;
        if (out_r2) out_r2[i] = L.hd[t];
    } else {
        float r = 0.f, g = 0.f, b = 0.f;
        if (found >= 8) {                                          // literal 8, photon.cpp:411
            const float nx = nrm3[3 * i], ny = nrm3[3 * i + 1], nz = nrm3[3 * i + 2];
            const float* cosT = tables; const float* sinT = tables + 256;
            const float* cosP = tables + 512; const float* sinP = tables + 768;
            for (int j = 1; j <= found; ++j) {
                const DevPhoton* p = ph + L.hi[j * 64 + t];
                const unsigned pk = p->packed;
                const int theta = (pk >> 8) & 255, phi = (pk >> 16) & 255;
                const float dx = sinT[theta] * cosP[phi], dy = sinT[theta] * sinP[phi], dz = cosT[theta];
                if ((dx * nx + dy * ny + dz * nz) < 0.0f) { r += p->power[0]; g += p->power[1]; b += p->power[2]; }
            }
            const float tmp = (float)(((double)1.0f / 3.141592) / (double)L.hd[t]);   // (1.0f/PI)/(np.dist2[0])
            r *= tmp; g *= tmp; b *= tmp;
        }
        out_rgb[3 * i] = r; out_rgb[3 * i + 1] = g; out_rgb[3 * i + 2] = b;
    }
}

size_t rtk_knn_lds_bytes(int k) { return (size_t)(2 * (k + 1) + 2 * KNN_LEVELS) * 64 * 4; }

hipError_t rtk_photon_knn(const DevPhoton* ph, int64_t n, int64_t half, const float* d_pos3, const float* d_nrm3,
                          int64_t q, int k, float maxDist, int mode, int32_t* d_idx, float* d_d2, int32_t* d_found,
                          float* d_r2, float* d_rgb, const float* d_tables, hipStream_t st) {
    if (q <= 0) return hipSuccess;
    const size_t lds = rtk_knn_lds_bytes(k);
    if (lds > 64 * 1024) {   // per current device; cheap enough to repeat
        hipError_t e = hipFuncSetAttribute((const void*)k_photon_knn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_photon_knn, dim3((unsigned)((q + 63) / 64)), dim3(64), lds, st, ph, n, half, d_pos3, d_nrm3, q, k,
                       maxDist, mode, d_idx, d_d2, d_found, d_r2, d_rgb, d_tables);
    return hipGetLastError();
}
