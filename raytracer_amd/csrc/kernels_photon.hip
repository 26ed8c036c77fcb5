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
#include <cstdlib>
#include "rt_internal.h"

#define KNN_LEVELS 32

// Queries per 64-lane wave.  The candidate heaps (808 B per query at k = 100) are what limits residency: with 64
// queries a wave needs 62 KB of LDS and a CU holds two waves (two of its four SIMDs idle, nothing to hide LDS or
// memory latency behind).  Fewer queries per wave = more resident waves for the same number of queries in flight,
// and fewer lanes waiting for the one lane that is re-heapifying.  Lanes >= KNN_QL exit at once.
// KNN_QL is a template parameter of the kernel; rtk_photon_knn picks the instantiation (RT_KNN_QL overrides).

namespace {

struct KnnLds {
    float* hd;      // [(k+1)][64]  np.dist2
    int*   hi;      // [(k+1)][64]  np.index
    float* s1;      // [KNN_LEVELS][64] plane distance per level
    float* s2;      // [KNN_LEVELS][64] photon distance^2 per level
};

// the "we found a photon" block of LocatePhoton, photon.cpp:79-128.  r2 mirrors np.dist2[0] in a register.
template <int KNN_QL>
__device__ __forceinline__ void consider(const KnnLds& L, int t, int k, int& found, int& got_heap, float& r2, int index,
                                         float dist2) {
#define HD(j) L.hd[(j) * KNN_QL + t]
#define HI(j) L.hi[(j) * KNN_QL + t]
    if (!(dist2 < r2)) return;
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
        float dj = HD(j);
        if (j < found) { const float dj1 = HD(j + 1); if (dj < dj1) { ++j; dj = dj1; } }
        if (dist2 > dj) break;
        HD(parent) = dj; HI(parent) = HI(j);
        parent = j; j += j;
    }
    HI(parent) = index; HD(parent) = dist2;
    r2 = HD(1);
#undef HD
#undef HI
}

// LocatePhoton's recursion (photon.cpp:53-131) unrolled over the implicit heap numbering.  Every trip of the outer
// loop fetches exactly one photon record per lane (all lanes keep a load in flight); a lane that reached a node
// without children then climbs -- candidates come from the per-level stash in LDS, no memory traffic -- until a far
// subtree has to be entered or the root is done.  The candidates reach `consider` in the reference's order
// (near subtree, far subtree, the node itself).
template <int KNN_QL>
__device__ __forceinline__ void locate(const DevPhoton* __restrict__ ph, int64_t n, int64_t half, const KnnLds& L,
                                       int t, int k, float qx, float qy, float qz, float& r2, int& found) {
    int got_heap = 0;
    found = 0;
    bool done = n < 1;
    unsigned idx = 1;
    const unsigned uhalf = half > 0 ? (unsigned)half : 0u;
    while (__ballot(!done) != 0ull) {
        if (done) continue;
        const float4 rec = *reinterpret_cast<const float4*>(ph + idx);           // pos.xyz + packed (16 B record)
        float d1 = rec.x - qx;
        float dist2 = d1 * d1;
        d1 = rec.y - qy; dist2 += d1 * d1;
        d1 = rec.z - qz; dist2 += d1 * d1;
        if (idx < uhalf) {
            const int plane = __float_as_int(rec.w) & 3;
            const float pp = plane == 0 ? rec.x : (plane == 1 ? rec.y : rec.z);
            const float qp = plane == 0 ? qx : (plane == 1 ? qy : qz);
            const float dist1 = qp - pp;
            const int lvl = 31 - __builtin_clz(idx);
            L.s1[lvl * KNN_QL + t] = dist1; L.s2[lvl * KNN_QL + t] = dist2;
            idx = (dist1 > 0.0f) ? 2 * idx + 1 : 2 * idx;
            continue;
        }
        int cidx = (int)idx;
        float cd2 = dist2;
        for (;;) {
            consider<KNN_QL>(L, t, k, found, got_heap, r2, cidx, cd2);
            if (idx == 1) { done = true; break; }
            const unsigned child = idx;
            idx >>= 1;
            const int lvl = 31 - __builtin_clz(idx);
            const float dist1 = L.s1[lvl * KNN_QL + t];
            const unsigned near = (dist1 > 0.0f) ? 2 * idx + 1 : 2 * idx;
            if (child == near && dist1 * dist1 < r2) { idx = near ^ 1u; break; }
            cidx = (int)idx; cd2 = L.s2[lvl * KNN_QL + t];
        }
    }
}

}  // namespace

// mode 0: write idx/dist2/found/radius2.   mode 1: irradiance rgb.
template <int KNN_QL>
__global__ void __launch_bounds__(64)
k_photon_knn(const DevPhoton* __restrict__ ph, int64_t n, int64_t half, const float* __restrict__ pos3,
             const float* __restrict__ nrm3, int64_t q, int k, float maxDist, int mode,
             int32_t* __restrict__ out_idx, float* __restrict__ out_d2, int32_t* __restrict__ out_found,
             float* __restrict__ out_r2, float* __restrict__ out_rgb, const float* __restrict__ tables, int xcd_map,
             const uint8_t* __restrict__ only, const int32_t* __restrict__ list, const unsigned* __restrict__ nlist) {
    extern __shared__ float lds[];
    KnnLds L;
    L.hd = lds;
    L.hi = reinterpret_cast<int*>(lds + (k + 1) * KNN_QL);
    L.s1 = lds + 2 * (k + 1) * KNN_QL;
    L.s2 = L.s1 + KNN_LEVELS * KNN_QL;
    const int t = threadIdx.x;
    // xcd_map: give every XCD (blocks are dealt round-robin to the 8 XCDs) one contiguous range of the queries.
    // Measured SLOWER (287 ms vs 225 ms): the cost of a query varies over the scene and the ranges finish at
    // different times; the interleaved default balances the XCDs and the shared nodes sit in the Infinity Cache.
    // (the grid is a multiple of 8 blocks: block b -> logical block (b % 8) * (grid / 8) + b / 8, a bijection)
    const unsigned per = gridDim.x >> 3;
    const unsigned lb = xcd_map ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    if (t >= KNN_QL) return;
    // `list` (redo pass of the fast kernels): the queries to do are list[0 .. *nlist), ONE per wave (they are few, and
    // a wave is as slow as its slowest query), the blocks stride over them
    if (list && t > 0) return;
    const int64_t stride = list ? (int64_t)gridDim.x : 0;
    for (int64_t j = list ? (int64_t)lb : (int64_t)lb * KNN_QL + t;; j += stride) {
        int64_t i = j;
        if (list) { if (j >= (int64_t)*nlist) break; i = list[j]; }
        else if (i >= q || (only && !only[i])) break;                  // `only`: the queries the cooperative kernel flagged
        const float qx = pos3[3 * i], qy = pos3[3 * i + 1], qz = pos3[3 * i + 2];
        float r2 = maxDist * maxDist;                                  // np.dist2[0], photon.cpp:405
        int found;
        locate<KNN_QL>(ph, n, half, L, t, k, qx, qy, qz, r2, found);
        L.hd[t] = r2;
        if (mode == 0) {
            for (int jj = 0; jj < k; ++jj) {
                out_idx[i * k + jj] = jj < found ? L.hi[(jj + 1) * KNN_QL + t] : -1;
                out_d2[i * k + jj] = jj < found ? L.hd[(jj + 1) * KNN_QL + t] : 0.0f;
            }
            out_found[i] = found;
            if (out_r2) out_r2[i] = L.hd[t];
        } else {
            float r = 0.f, g = 0.f, b = 0.f;
            if (found >= 8) {                                          // literal 8, photon.cpp:411
                const float nx = nrm3[3 * i], ny = nrm3[3 * i + 1], nz = nrm3[3 * i + 2];
                const float* cosT = tables; const float* sinT = tables + 256;
                const float* cosP = tables + 512; const float* sinP = tables + 768;
                const DevPhotonPow* pw = reinterpret_cast<const DevPhotonPow*>(ph + (n + 1));
                for (int jj = 1; jj <= found; ++jj) {
                    const int pi = L.hi[jj * KNN_QL + t];
                    const unsigned pk = ph[pi].packed;
                    const int theta = (pk >> 8) & 255, phi = (pk >> 16) & 255;
                    const float dx = sinT[theta] * cosP[phi], dy = sinT[theta] * sinP[phi], dz = cosT[theta];
                    if ((dx * nx + dy * ny + dz * nz) < 0.0f) {
                        const float4 w4 = *reinterpret_cast<const float4*>(pw + pi);
                        r += w4.x; g += w4.y; b += w4.z;
                    }
                }
                const float tmp = (float)(((double)1.0f / 3.141592) / (double)L.hd[t]);   // (1.0f/PI)/(np.dist2[0])
                r *= tmp; g *= tmp; b *= tmp;
            }
            out_rgb[3 * i] = r; out_rgb[3 * i + 1] = g; out_rgb[3 * i + 2] = b;
        }
        if (!list) break;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// FAST kNN: 16 lanes per query, 4 queries per wave, no LDS.
//
// The strict kernel above replays the reference's array/heap code statement by statement, which pins the ORDER of
// np.index; its price is a per-query heap in LDS (808 B at k = 100), latency-bound sift loops and 4 useful lanes per
// wave.  What the reference's result depends on is only the SET: LocatePhoton visits the nodes in an order the
// candidate list does not influence except through its current maximum, accepts a candidate iff dist2 < that
// maximum, and evicts the maximum.  So here the k candidates of a query live in the registers of its 16 lanes
// (slot s -> lane s % 16, register s / 16), the maximum is a 16-lane reduction, and the walk (same visits, same
// order, same accept test, incl. the reference's quirk that the first candidate after the list is full evicts the
// farthest even when it is farther) leaves the same set -- unless two candidates tie for the maximum when one must
// go: which one the reference's heap drops depends on its history, so such a query is flagged and redone by the
// strict kernel.  The irradiance sum adds the same photons in slot order instead of heap order (float association).
#define COOP_L 16

template <int S>
struct CoopSet { float d[S]; int ix[S]; };

__device__ __forceinline__ float group_max16(float v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off, COOP_L));
    return v;
}
__device__ __forceinline__ float group_sum16(float v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, COOP_L);
    return v;
}
__device__ __forceinline__ int group_sum16i(int v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, COOP_L);
    return v;
}

template <int S>
__device__ __forceinline__ void coop_consider(CoopSet<S>& c, int l16, int k, int& found, float& r2, bool& ambiguous,
                                              int index, float dist2) {
    if (!(dist2 < r2)) return;
    if (found < k) {
        const int slot = found;
        if (l16 == (slot & 15)) {
#pragma unroll
            for (int r = 0; r < S; ++r) if (r == (slot >> 4)) { c.d[r] = dist2; c.ix[r] = index; }
        }
        ++found;
        return;
    }
    // evict the maximum (slots < k are all valid here)
    float lm = -1.0f; int lr = 0, neq = 0;
#pragma unroll
    for (int r = 0; r < S; ++r) if (r * 16 + l16 < k && c.d[r] > lm) { lm = c.d[r]; lr = r; }
    const float M = group_max16(lm);
#pragma unroll
    for (int r = 0; r < S; ++r) neq += (r * 16 + l16 < k && c.d[r] == M) ? 1 : 0;
    if (group_sum16i(neq) > 1) ambiguous = true;
    const unsigned long long owners = __ballot(lm == M) >> (threadIdx.x & 48);       // this group's 16 bits
    const int owner = __builtin_ctz((unsigned)(owners & 0xffffu));
    if (l16 == owner) {
#pragma unroll
        for (int r = 0; r < S; ++r) if (r == lr) { c.d[r] = dist2; c.ix[r] = index; }
    }
    lm = -1.0f;
#pragma unroll
    for (int r = 0; r < S; ++r) if (r * 16 + l16 < k && c.d[r] > lm) lm = c.d[r];
    r2 = group_max16(lm);                                                           // np.dist2[0] = np.dist2[1]
}

// mode 1: irradiance rgb.   mode 2: the candidate SET (idx/dist2 in slot order, found, radius2).
template <int S>
__global__ void __launch_bounds__(64)
k_photon_knn_coop(const DevPhoton* __restrict__ ph, int64_t n, int64_t half, const float* __restrict__ pos3,
                  const float* __restrict__ nrm3, int64_t q, int k, float maxDist, int mode,
                  int32_t* __restrict__ out_idx, float* __restrict__ out_d2, int32_t* __restrict__ out_found,
                  float* __restrict__ out_r2, float* __restrict__ out_rgb, const float* __restrict__ tables,
                  uint8_t* __restrict__ flagged) {
    const int l16 = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 4);
    if (i >= q) return;                                   // whole 16-lane groups leave together
    const float qx = pos3[3 * i], qy = pos3[3 * i + 1], qz = pos3[3 * i + 2];
    CoopSet<S> c;
#pragma unroll
    for (int r = 0; r < S; ++r) { c.d[r] = 0.f; c.ix[r] = 0; }
    float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};       // per-level stash: level v -> lane v % 16, register v / 16
    float r2 = maxDist * maxDist;
    int found = 0;
    bool ambiguous = false, done = n < 1;
    unsigned idx = 1;
    const unsigned uhalf = half > 0 ? (unsigned)half : 0u;
    while (!done) {                                       // uniform within a 16-lane group
        const float4 rec = *reinterpret_cast<const float4*>(ph + idx);
        float d1 = rec.x - qx;
        float dist2 = d1 * d1;
        d1 = rec.y - qy; dist2 += d1 * d1;
        d1 = rec.z - qz; dist2 += d1 * d1;
        if (idx < uhalf) {
            const int plane = __float_as_int(rec.w) & 3;
            const float pp = plane == 0 ? rec.x : (plane == 1 ? rec.y : rec.z);
            const float qp = plane == 0 ? qx : (plane == 1 ? qy : qz);
            const float dist1 = qp - pp;
            const int lvl = 31 - __builtin_clz(idx);
            if (l16 == (lvl & 15)) {
                if (lvl >> 4) { st1[1] = dist1; st2[1] = dist2; } else { st1[0] = dist1; st2[0] = dist2; }
            }
            idx = (dist1 > 0.0f) ? 2 * idx + 1 : 2 * idx;
            continue;
        }
        int cidx = (int)idx;
        float cd2 = dist2;
        for (;;) {
            coop_consider<S>(c, l16, k, found, r2, ambiguous, cidx, cd2);
            if (idx == 1) { done = true; break; }
            const unsigned child = idx;
            idx >>= 1;
            const int lvl = 31 - __builtin_clz(idx);
            const float dist1 = __shfl((lvl >> 4) ? st1[1] : st1[0], lvl & 15, COOP_L);
            const unsigned near = (dist1 > 0.0f) ? 2 * idx + 1 : 2 * idx;
            if (child == near && dist1 * dist1 < r2) { idx = near ^ 1u; break; }
            cidx = (int)idx; cd2 = __shfl((lvl >> 4) ? st2[1] : st2[0], lvl & 15, COOP_L);
        }
    }
    if (flagged && l16 == 0) flagged[i] = ambiguous ? 1 : 0;
    if (mode == 2) {
#pragma unroll
        for (int r = 0; r < S; ++r) {
            const int slot = r * 16 + l16;
            if (slot < k) {
                out_idx[i * k + slot] = slot < found ? c.ix[r] : -1;
                out_d2[i * k + slot] = slot < found ? c.d[r] : 0.0f;
            }
        }
        if (l16 == 0) { out_found[i] = found; if (out_r2) out_r2[i] = r2; }
        return;
    }
    float r = 0.f, g = 0.f, b = 0.f;
    if (found >= 8) {                                              // literal 8, photon.cpp:411
        const float nx = nrm3[3 * i], ny = nrm3[3 * i + 1], nz = nrm3[3 * i + 2];
        const float* cosT = tables; const float* sinT = tables + 256;
        const float* cosP = tables + 512; const float* sinP = tables + 768;
        const DevPhotonPow* pw = reinterpret_cast<const DevPhotonPow*>(ph + (n + 1));
#pragma unroll
        for (int rr = 0; rr < S; ++rr) {
            if (rr * 16 + l16 < found) {
                const int pi = c.ix[rr];
                const unsigned pk = ph[pi].packed;
                const int theta = (pk >> 8) & 255, phi = (pk >> 16) & 255;
                const float dx = sinT[theta] * cosP[phi], dy = sinT[theta] * sinP[phi], dz = cosT[theta];
                if ((dx * nx + dy * ny + dz * nz) < 0.0f) {
                    const float4 w4 = *reinterpret_cast<const float4*>(pw + pi);
                    r += w4.x; g += w4.y; b += w4.z;
                }
            }
        }
        r = group_sum16(r); g = group_sum16(g); b = group_sum16(b);
        const float tmp = (float)(((double)1.0f / 3.141592) / (double)r2);           // (1.0f/PI)/(np.dist2[0])
        r *= tmp; g *= tmp; b *= tmp;
    }
    if (l16 == 0) { out_rgb[3 * i] = r; out_rgb[3 * i + 1] = g; out_rgb[3 * i + 2] = b; }
}

static int knn_ql() {
    static int ql = 0;
    if (!ql) {
        ql = 4;      // measured on MI355X, 10 M photons, k = 100: 64 -> 509 ms, 32 -> 432, 16 -> 314, 8 -> 241, 4 -> 226, 2 -> 296
        if (const char* e = getenv("RT_KNN_QL")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) ql = v; }
    }
    return ql;
}

size_t rtk_knn_lds_bytes(int k) { return (size_t)(2 * (k + 1) + 2 * KNN_LEVELS) * knn_ql() * 4; }

template <int QL>
static hipError_t launch_knn(const DevPhoton* ph, int64_t n, int64_t half, const float* d_pos3, const float* d_nrm3,
                             int64_t q, int k, float maxDist, int mode, int32_t* d_idx, float* d_d2, int32_t* d_found,
                             float* d_r2, float* d_rgb, const float* d_tables, const uint8_t* only, hipStream_t st,
                             const int32_t* list = nullptr, const unsigned* nlist = nullptr) {
    const size_t lds = (size_t)(2 * (k + 1) + 2 * KNN_LEVELS) * QL * 4;
    if (lds > 64 * 1024) {   // per current device; cheap enough to repeat
        hipError_t e = hipFuncSetAttribute((const void*)k_photon_knn<QL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // redo pass: a fixed grid strides over the list (its length is only known on the device)
    const unsigned grid = list ? 2048u : (unsigned)((((q + QL - 1) / QL) + 7) / 8) * 8;
    hipLaunchKernelGGL(k_photon_knn<QL>, dim3(grid), dim3(64), lds, st, ph, n, half, d_pos3, d_nrm3, q,
                       k, maxDist, mode, d_idx, d_d2, d_found, d_r2, d_rgb, d_tables, list ? 0 : (getenv("RT_KNN_XCD") ? atoi(getenv("RT_KNN_XCD")) : 0), only,
                       list, nlist);
    return hipGetLastError();
}

hipError_t rtk_photon_knn(const DevPhoton* ph, int64_t n, int64_t half, const float* d_pos3, const float* d_nrm3,
                          int64_t q, int k, float maxDist, int mode, int32_t* d_idx, float* d_d2, int32_t* d_found,
                          float* d_r2, float* d_rgb, const float* d_tables, hipStream_t st) {
    if (q <= 0) return hipSuccess;
#define KNN_ARGS ph, n, half, d_pos3, d_nrm3, q, k, maxDist, mode, d_idx, d_d2, d_found, d_r2, d_rgb, d_tables, nullptr, st
    switch (knn_ql()) {
        case 2: return launch_knn<2>(KNN_ARGS);
        case 4: return launch_knn<4>(KNN_ARGS);
        case 8: return launch_knn<8>(KNN_ARGS);
        case 32: return launch_knn<32>(KNN_ARGS);
        case 64: return launch_knn<64>(KNN_ARGS);
        case 16: return launch_knn<16>(KNN_ARGS);
        default: return launch_knn<4>(KNN_ARGS);
    }
#undef KNN_ARGS
}

// ---------------------------------------------------------------------------------------------------------------
// The redo of the few queries the wave-shared kernel flags (exact ties): ONE query per wave, everything wave-uniform.
// A launch of k_photon_knn over them is as slow as one query, and what one query costs there is not its ~1 400 record
// fetches but one lane's sift loops over the candidate heap in LDS (~1 000 accepted candidates at r = 1.0, ~100 cycles
// per dependent LDS access).  Here the reference's arrays np.dist2 / np.index live in REGISTERS across the wave (slot s in
// lane s % 64, register s / 64), read with v_readlane and written by the owning lane -- a few cycles per access -- and
// the per-level stash likewise (level l in lane l).  All 64 lanes fetch: when the walk needs a node outside the window,
// lane L loads the L-th node of the six heap levels below it (addresses need no fetch), so the walker descends up to six
// levels per memory latency.  Same visits, same order, same statements as `consider` / `locate`: the lists, radius and
// irradiance are theirs bit for bit (tests/test_gpu_photon.py: ties, lattices, the reference's fixture).
#define KNN_WIN_LEVELS 6
#define REDO_S 2                      // slots 0 .. 127: k <= 126
namespace {
__device__ __forceinline__ float rl_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float rd_f(const float (&a)[REDO_S], int j) { return rl_f((j >> 6) ? a[1] : a[0], j & 63); }
__device__ __forceinline__ int rd_i(const int (&a)[REDO_S], int j) { return __builtin_amdgcn_readlane((j >> 6) ? a[1] : a[0], j & 63); }
__device__ __forceinline__ void wr_f(float (&a)[REDO_S], int j, float v, int lane) {
    const bool mine = lane == (j & 63);
    if (j >> 6) a[1] = mine ? v : a[1]; else a[0] = mine ? v : a[0];
}
__device__ __forceinline__ void wr_i(int (&a)[REDO_S], int j, int v, int lane) {
    const bool mine = lane == (j & 63);
    if (j >> 6) a[1] = mine ? v : a[1]; else a[0] = mine ? v : a[0];
}
// photon.cpp:79-128 on the register arrays; every argument is wave-uniform
__device__ __forceinline__ void consider_uniform(float (&hd)[REDO_S], int (&hi)[REDO_S], int lane, int k, int& found, int& got_heap,
                                                 float& r2, int index, float dist2) {
    if (!(dist2 < r2)) return;
    if (found < k) {
        ++found;
        wr_f(hd, found, dist2, lane); wr_i(hi, found, index, lane);
        return;
    }
    int j, parent;
    if (got_heap == 0) {
        const int half_found = found >> 1;
        for (int kk = half_found; kk >= 1; --kk) {
            parent = kk;
            const int phot = rd_i(hi, kk);
            const float dst2 = rd_f(hd, kk);
            while (parent <= half_found) {
                j = parent + parent;
                if (j < found && rd_f(hd, j) < rd_f(hd, j + 1)) ++j;
                const float dj = rd_f(hd, j);
                if (dst2 >= dj) break;
                wr_f(hd, parent, dj, lane); wr_i(hi, parent, rd_i(hi, j), lane);
                parent = j;
            }
            wr_f(hd, parent, dst2, lane); wr_i(hi, parent, phot, lane);
        }
        got_heap = 1;
    }
    parent = 1; j = 2;
    while (j <= found) {
        float dj = rd_f(hd, j);
        if (j < found) { const float dj1 = rd_f(hd, j + 1); if (dj < dj1) { ++j; dj = dj1; } }
        if (dist2 > dj) break;
        wr_f(hd, parent, dj, lane); wr_i(hi, parent, rd_i(hi, j), lane);
        parent = j; j += j;
    }
    wr_i(hi, parent, index, lane); wr_f(hd, parent, dist2, lane);
    r2 = rd_f(hd, 1);
}
// the key of slot sidx's turn in the reference's visiting order for the query (qx, qy, qz): per ancestor two bits, 0 = the
// query's near side, 1 = the far side; 2 at the slot's own depth = its own turn, after both subtrees (photon.cpp:58-78)
__device__ __forceinline__ unsigned long long redo_visit_key(const DevPhoton* __restrict__ ph, unsigned sidx, float qx, float qy, float qz) {
    const int d = 31 - __builtin_clz(sidx);
    unsigned long long key = 2ull << (62 - 2 * d);
    for (int j = 0; j < d; ++j) {
        const unsigned a = sidx >> (d - j);                         // the ancestor at depth j
        const float4 rec = *reinterpret_cast<const float4*>(ph + a);
        const int plane = __float_as_int(rec.w) & 3;
        const float pp = plane == 0 ? rec.x : (plane == 1 ? rec.y : rec.z);
        const float qp = plane == 0 ? qx : (plane == 1 ? qy : qz);
        const float dist1 = qp - pp;                                // photon.cpp:61
        const unsigned near_right = dist1 > 0.0f ? 1u : 0u;
        const unsigned side = (sidx >> (d - j - 1)) & 1u;
        key |= (unsigned long long)(side == near_right ? 0u : 1u) << (62 - 2 * j);
    }
    return key;
}
}  // namespace

// WALK 1 (default): LocatePhoton's recursion on an explicit stack, and whole six-level windows of the heap skipped at once.
// The reference visits: the near child's subtree, the far child's if dist1^2 < np.dist2[0] AT THAT TIME, then the node
// itself (photon.cpp:53-131).  Two facts make the walk cheap without changing what it accepts or in which order:
//  (1) a photon below a far step with dist1^2 = m has dist2 >= m (the heap's median split; float subtraction, product and
//      sum are monotone), and np.dist2[0] only shrinks -- so a subtree whose path from the root holds a far step with
//      m >= np.dist2[0] NOW cannot contain a photon that will ever be accepted: it is skipped, whether or not the reference
//      would still be inside it; visiting a subtree the reference prunes is harmless for the same reason;
//  (2) a node whose dist2 is not below np.dist2[0] now never will be.
// So when the 63 nodes of the window below a node (one fetch by 63 lanes) hold no candidate, the walk does not step
// through them: every lane derives for its node, level by level, whether the walk reaches it and the largest m on its
// path, the 64 subtrees below the window are put on the stack in the reference's visiting order (near before far at every
// level) with their m, and the window's own nodes are dropped.  Where the caustic map's heap has no spatial order --
// coplanar photons: every third level splits on the axis all photons share, dist1 = 0, both sides always visited -- the
// reference's walk is hundreds of thousands of nodes long (200 ms for ONE query, the whole of config 4's strict-redo cost);
// it becomes a few thousand windows.  A window with a candidate is walked node by node from the stack.
// WALK 0: the stackless walk of rounds 1-2 (RT_KNN_REDO_WALK=0).
// WALK 2 (default since round 3's third session): WALK 1 -- and, for a walk that is still going REDO_SWITCH stack items after
// the reference's heap was built (a query of ~1 000 accepted photons in a spatially ordered heap is done by then), NO walk of
// the reference's heap for the rest.  Once its heap exists the reference accepts a photon only if its dist2 is below the
// radius of that moment, which never grows: every photon that can still be accepted lies within the radius R of NOW.  Those are found in the SECOND heap (spatially ordered,
// a box per sub-heap: a wave-wide frontier walk, ~20 rounds), each gets the key of its turn in the reference's visiting
// order -- per ancestor in the reference's heap one digit, 0 = the query's near side, 1 = the far side, and 2 for the node's
// own turn after both subtrees (photon.cpp:58-78) -- the keys are sorted, and the reference's `consider` runs over them in
// that order: the same photons are accepted in the same order into the same arrays, so np.index, np.dist2 and the sum are
// the reference's bit for bit.  (A far subtree the reference does not enter holds only photons with dist2 >= dist1^2 >= the
// radius of that moment: `consider` rejects them here as the reference would have, had it looked.)  What this replaces is
// the walk, not the list: in a heap without spatial order (the caustic map's coplanar photons, every third level splits on
// y with dist1 = 0) one query's walk took up to 175 ms; the keys cost ~25 fetches per candidate, all independent.  Run from
// the start it loses to WALK 1 on ordinary queries (3.7 against 1.0 ms per query: the radius is still wide when the heap is
// built), hence the switch.  More than REDO_CAND candidates or a frontier wider than REDO_FRONT: WALK 1 simply goes on.
#define REDO_CAND 1024
#define REDO_FRONT 1024
#define REDO_SWITCH 3000
#define REDO_GIVEUP 16000
template <int WALK>
__global__ void __launch_bounds__(64)
k_photon_knn_redo(const DevPhoton* __restrict__ ph, int64_t n, int64_t half, const float* __restrict__ pos3,
                  const float* __restrict__ nrm3, int k, float maxDist, float* __restrict__ out_rgb,
                  const float* __restrict__ tables, const int32_t* __restrict__ list, const unsigned* __restrict__ nlist,
                  unsigned redo_switch, int give_up_after, int32_t* __restrict__ list2, unsigned* __restrict__ nlist2) {
    __shared__ float4 win[64];
    __shared__ unsigned long long c_key[WALK == 2 ? REDO_CAND : 1];  // WALK 2: candidates: visiting-order key,
    __shared__ unsigned c_slot[WALK == 2 ? REDO_CAND : 1];          // ... slot in the reference's heap
    __shared__ unsigned fq[WALK == 2 ? 2 * REDO_FRONT : 1];         // ... the frontier of the second heap's walk, two buffers
    __shared__ unsigned cnts[4];                                    // ... candidates, next frontier
    __shared__ unsigned stk_i[WALK ? 512 : 1];                      // WALK 1: pending items: node | kind << 30 (0 subtree, 2 the node itself)
    __shared__ float stk_m[WALK ? 512 : 1];                         // ... and the largest far-step dist1^2 on the way to it (-1: none)
    __shared__ unsigned ord_i[WALK ? 64 : 1];                       // a window's exits by visiting order
    __shared__ float ord_m[WALK ? 64 : 1];
    __shared__ int ord_v[WALK ? 64 : 1];
    const int lane = threadIdx.x;
    const int wd = 31 - __builtin_clz((unsigned)lane + 1u);        // this lane's node of a window: level wd, position wj
    const unsigned wj = (unsigned)lane + 1u - (1u << wd);
    const unsigned uhalf = half > 0 ? (unsigned)half : 0u;
    const unsigned nl = *nlist;
    for (unsigned jq = blockIdx.x; jq < nl; jq += gridDim.x) {
        const int64_t i = list[jq];
        const float qx = pos3[3 * i], qy = pos3[3 * i + 1], qz = pos3[3 * i + 2];
        float hd[REDO_S] = {0.f, 0.f}; int hi[REDO_S] = {0, 0};
        float r2 = maxDist * maxDist;                               // np.dist2[0], photon.cpp:405
        int found = 0, got_heap = 0;
        bool deferred = false;
        if (WALK >= 1) {
            bool tried = false;                                     // WALK 2: the candidates were tried once (too many: the walk goes on)
            unsigned last_idx = 0;                                  // the photon `consider` saw last
            int sp = 0;
            if (n >= 1) { if (lane == 0) { stk_i[0] = 1u; stk_m[0] = -1.0f; } sp = 1; }
            __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
            unsigned pops = 0;
            bool gave_up = false;
            for (;;) {                                              // (twice at most: the walk, the finish by keys, and the walk again if that gave up)
            bool want_keys = false;
            while (sp > 0) {
                // WALK 1 with a limit: a walk that is still going `give_up_after` items after the heap was built is left to the
                // launch that follows (WALK 2 over list2), which can finish it by keys
                if (WALK == 1 && give_up_after >= 0 && got_heap && (int)++pops > give_up_after) { gave_up = true; break; }
                // WALK 2: a walk that is still going REDO_SWITCH items after the heap was built is in a heap without spatial order:
                // what is left of it is done by keys (everything up to last_idx's turn is behind the walk; what it skipped since
                // cannot be accepted any more and is rejected again)
                if (WALK == 2 && got_heap && !tried && ++pops > redo_switch) { tried = true; want_keys = true; break; }
                --sp;
                const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)stk_i[sp]);
                const float m0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(stk_m[sp])));
                const unsigned idx = item & 0x3fffffffu;
                if ((item >> 30) == 2u) {                           // the node itself, after both subtrees
                    const float4 rec = *reinterpret_cast<const float4*>(ph + idx);
                    float d1 = rec.x - qx; float dist2 = d1 * d1;
                    d1 = rec.y - qy; dist2 += d1 * d1;
                    d1 = rec.z - qz; dist2 += d1 * d1;
                    consider_uniform(hd, hi, lane, k, found, got_heap, r2, (int)idx, rl_f(dist2, 0));
                    last_idx = idx;
                    continue;
                }
                if (!(m0 < r2)) continue;                           // (1): nothing below can be accepted any more
                if (idx >= uhalf) {                                 // no recursion below this index (photon.cpp:58): the node alone
                    const float4 rec = *reinterpret_cast<const float4*>(ph + idx);
                    float d1 = rec.x - qx; float dist2 = d1 * d1;
                    d1 = rec.y - qy; dist2 += d1 * d1;
                    d1 = rec.z - qz; dist2 += d1 * d1;
                    consider_uniform(hd, hi, lane, k, found, got_heap, r2, (int)idx, rl_f(dist2, 0));
                    last_idx = idx;
                    continue;
                }
                // the window below idx: six levels, or what the heap still has (a node from `uhalf` on does not recurse, one beyond
                // n does not exist); the stack must hold its 64 exits
                const bool fits = idx < (1u << 23) && sp + 64 + 100 <= 512;
                bool expanded = false;
                if (fits) {
                    const unsigned node = (idx << wd) + wj;
                    const bool exists = lane < 63 && (unsigned long long)node <= (unsigned long long)n;
                    const int recurses = (exists && node < uhalf) ? 1 : 0;
                    float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (exists) rec = *reinterpret_cast<const float4*>(ph + node);
                    float d1 = rec.x - qx; float dist2 = d1 * d1;
                    d1 = rec.y - qy; dist2 += d1 * d1;
                    d1 = rec.z - qz; dist2 += d1 * d1;
                    if (__ballot(exists && dist2 < r2) == 0ull) {
                        const int plane = __float_as_int(rec.w) & 3;
                        const float pp = plane == 0 ? rec.x : (plane == 1 ? rec.y : rec.z);
                        const float qp = plane == 0 ? qx : (plane == 1 ? qy : qz);
                        const float dist1 = qp - pp;                // photon.cpp:61
                        const float d1sq = dist1 * dist1;
                        const int near_right = dist1 > 0.0f ? 1 : 0;
                        // reached? and the largest far-step dist1^2 on the path, root of the window first
                        int vis = lane == 0 ? 1 : 0;
                        float req = m0;
                        unsigned key = 0u;
                        const int plane_of_parent = (lane + 1) / 2 - 1;      // the parent's lane (window node w = lane + 1, parent w / 2)
                        const int is_right = (lane + 1) & 1;
#pragma unroll
                        for (int lv = 1; lv <= 5; ++lv) {
                            const int pv = __shfl(vis, plane_of_parent, 64) & __shfl(recurses, plane_of_parent, 64), pn = __shfl(near_right, plane_of_parent, 64);
                            const float pr = __shfl(req, plane_of_parent, 64), pd = __shfl(d1sq, plane_of_parent, 64);
                            const unsigned pk = (unsigned)__shfl((int)key, plane_of_parent, 64);
                            if (wd == lv && lane < 63) {
                                const bool near = is_right == pn;
                                req = near ? pr : __builtin_fmaxf(pr, pd);
                                vis = (pv && exists && req < r2) ? 1 : 0;
                                key = (pk << 1) | (near ? 0u : 1u);
                            }
                        }
                        // the 64 subtrees below the window: exit e = child (e & 1) of the level-5 node e >> 1 (lane 31 + (e >> 1))
                        {
                            const int pl = 31 + (lane >> 1);
                            const int pv = __shfl(vis, pl, 64) & __shfl(recurses, pl, 64), pn = __shfl(near_right, pl, 64);
                            const float pr = __shfl(req, pl, 64), pd = __shfl(d1sq, pl, 64);
                            const unsigned pk = (unsigned)__shfl((int)key, pl, 64);
                            const bool near = (lane & 1) == pn;
                            const float ereq = near ? pr : __builtin_fmaxf(pr, pd);
                            const int evis = (pv && ereq < r2) ? 1 : 0;
                            const unsigned ekey = (pk << 1) | (near ? 0u : 1u);          // 6 bits: the visiting order
                            ord_i[ekey] = (idx << 6) + (unsigned)lane; ord_m[ekey] = ereq; ord_v[ekey] = evis;
                        }
                        __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
                        const int v = ord_v[lane];
                        const unsigned long long mask = __ballot(v != 0);
                        if (v) {                                     // the first to be visited on top
                            const int above = lane == 63 ? 0 : __builtin_popcountll(mask >> (lane + 1));
                            stk_i[sp + above] = ord_i[lane]; stk_m[sp + above] = ord_m[lane];
                        }
                        sp += __builtin_popcountll(mask);
                        __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
                        expanded = true;
                    }
                }
                if (expanded) continue;
                // one node: its own turn after both subtrees, the far subtree after the near one
                {
                    const float4 rec = *reinterpret_cast<const float4*>(ph + idx);
                    const int plane = __builtin_amdgcn_readfirstlane(__float_as_int(rec.w)) & 3;
                    const float pp = rl_f(plane == 0 ? rec.x : (plane == 1 ? rec.y : rec.z), 0);
                    const float qp = plane == 0 ? qx : (plane == 1 ? qy : qz);
                    const float dist1 = qp - pp;
                    const unsigned near = (dist1 > 0.0f) ? 2u * idx + 1u : 2u * idx;
                    if (lane == 0) {
                        stk_i[sp] = idx | (2u << 30); stk_m[sp] = -1.0f;
                        stk_i[sp + 1] = near ^ 1u; stk_m[sp + 1] = __builtin_fmaxf(m0, dist1 * dist1);
                        stk_i[sp + 2] = near; stk_m[sp + 2] = m0;
                    }
                    sp += 3;
                    __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
                }
            }
            if (WALK == 1 && gave_up) {
                if (lane == 0) list2[atomicAdd(nlist2, 1u)] = (int32_t)i;
                deferred = true;
                break;
            }
            if (!want_keys) break;
            // every photon that can still be accepted, in the reference's order; ok = false: too many, the walk goes on
                    bool ok = false;
                    {
                const DevPhotonPos* __restrict__ pp2 = ph + 2 * (n + 1);
                const DevPhotonPow* __restrict__ pq2 = reinterpret_cast<const DevPhotonPow*>(pp2 + (n + 2));
                const float4* __restrict__ bx = reinterpret_cast<const float4*>(pq2 + (n + 1));
                const unsigned un = (unsigned)n, nb = (unsigned)(n / 2);
                const float R = r2;
                if (lane == 0) { cnts[0] = 0u; cnts[1] = 0u; fq[0] = 1u; }
                __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
                unsigned ncur = 1u, cur = 0u;
                bool over = false;
                while (ncur > 0u && !over) {
                    for (unsigned base = 0; base < ncur; base += 64u) {
                        const unsigned node = base + lane < ncur ? fq[cur * REDO_FRONT + base + lane] : 0u;
                        if (node != 0u) {
                            const float4 rec = *reinterpret_cast<const float4*>(pp2 + node);
                            float d1 = rec.x - qx; float dist2 = d1 * d1;
                            d1 = rec.y - qy; dist2 += d1 * d1;
                            d1 = rec.z - qz; dist2 += d1 * d1;
                            if (dist2 < R && !((unsigned)__float_as_int(rec.w) & RT_PH_UNREACHABLE)) {
                                const unsigned at = atomicAdd(&cnts[0], 1u);
                                if (at < REDO_CAND) c_slot[at] = pq2[node].pad;
                            }
                            const unsigned c0 = 2u * node;
                            for (unsigned c = c0; c <= c0 + 1u && c <= un; ++c) {
                                bool want = true;
                                if (c <= nb) {
                                    const float4 lo = bx[2 * c], hi4 = bx[2 * c + 1];
                                    const float dx = __builtin_fmaxf(__builtin_fmaxf(lo.x - qx, qx - hi4.x), 0.0f);
                                    const float dy = __builtin_fmaxf(__builtin_fmaxf(lo.y - qy, qy - hi4.y), 0.0f);
                                    const float dz = __builtin_fmaxf(__builtin_fmaxf(lo.z - qz, qz - hi4.z), 0.0f);
                                    float bd = dx * dx; bd += dy * dy; bd += dz * dz;    // <= the dist2 of every photon inside (w_boxdist)
                                    want = bd < R;
                                }
                                if (want) {
                                    const unsigned at = atomicAdd(&cnts[1], 1u);
                                    if (at < REDO_FRONT) fq[(cur ^ 1u) * REDO_FRONT + at] = c;
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
                    const unsigned nnext = (unsigned)__builtin_amdgcn_readfirstlane((int)cnts[1]);
                    const unsigned ncand_now = (unsigned)__builtin_amdgcn_readfirstlane((int)cnts[0]);
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) cnts[1] = 0u;
                    __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
                    over = nnext > REDO_FRONT || ncand_now > REDO_CAND;
                    ncur = nnext; cur ^= 1u;
                }
                if (!over) {
                const unsigned nc = (unsigned)__builtin_amdgcn_readfirstlane((int)cnts[0]);
                // keys; what the walk has seen already (everything up to last_idx's turn) gets the largest key and is skipped below
                const unsigned long long seen = redo_visit_key(ph, last_idx, qx, qy, qz);
                unsigned np2 = 64u; while (np2 < nc) np2 <<= 1;
                for (unsigned t = lane; t < np2; t += 64u) {
                    unsigned long long key = ~0ull;
                    if (t < nc) { key = redo_visit_key(ph, c_slot[t], qx, qy, qz); if (key <= seen) key = ~0ull; }
                    c_key[t] = key;
                }
                __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
                // bitonic sort of (key, slot), ascending
                for (unsigned kk = 2u; kk <= np2; kk <<= 1) {
                    for (unsigned jj = kk >> 1; jj > 0u; jj >>= 1) {
                        for (unsigned t = lane; t < np2; t += 64u) {
                            const unsigned u = t ^ jj;
                            if (u > t) {
                                const unsigned long long a = c_key[t], b = c_key[u];
                                const bool up = (t & kk) == 0u;
                                if ((a > b) == up) {
                                    c_key[t] = b; c_key[u] = a;
                                    const unsigned sa = c_slot[t]; c_slot[t] = c_slot[u]; c_slot[u] = sa;
                                }
                            }
                        }
                        __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
                    }
                }
                // the reference's `consider`, photon by photon in its order (dist2 from the reference's own record, photon.cpp:73-77)
                for (unsigned t = 0; t < nc; ++t) {
                    const unsigned long long key = c_key[t];
                    if (key == ~0ull) break;
                    const unsigned sl = (unsigned)__builtin_amdgcn_readfirstlane((int)c_slot[t]);
                    const float4 rec = *reinterpret_cast<const float4*>(ph + sl);
                    float d1 = rec.x - qx; float dist2 = d1 * d1;
                    d1 = rec.y - qy; dist2 += d1 * d1;
                    d1 = rec.z - qz; dist2 += d1 * d1;
                    consider_uniform(hd, hi, lane, k, found, got_heap, r2, (int)sl, rl_f(dist2, 0));
                }
                ok = true;
                }
                    }
            if (ok) break;
            }
        } else {
        float st1 = 0.f, st2 = 0.f;                                 // LocatePhoton's dist1 / this node's dist2 per level, level l in lane l
        bool done = n < 1;
        unsigned idx = 1, wroot = 0;
        while (!done) {
            int d = -1;
            if (wroot != 0) {
                d = (31 - __builtin_clz(idx)) - (31 - __builtin_clz(wroot));
                if (d < 0 || d >= KNN_WIN_LEVELS || (idx >> d) != wroot) d = -1;
            }
            if (d < 0) {                                            // re-root the window at the node needed
                wroot = idx; d = 0;
                const unsigned long long node = ((unsigned long long)wroot << wd) + wj;
                __builtin_amdgcn_wave_barrier();
                if (lane < 63 && node <= (unsigned long long)n) win[lane] = *reinterpret_cast<const float4*>(ph + node);
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
            }
            const float4 rec4 = win[(1u << d) - 1u + (idx - (wroot << d))];
            const float rx = rl_f(rec4.x, 0), ry = rl_f(rec4.y, 0), rz = rl_f(rec4.z, 0);
            const int rw = __builtin_amdgcn_readfirstlane(__float_as_int(rec4.w));
            float d1 = rx - qx;
            float dist2 = d1 * d1;
            d1 = ry - qy; dist2 += d1 * d1;
            d1 = rz - qz; dist2 += d1 * d1;
            if (idx < uhalf) {
                const int plane = rw & 3;
                const float pp = plane == 0 ? rx : (plane == 1 ? ry : rz);
                const float qp = plane == 0 ? qx : (plane == 1 ? qy : qz);
                const float dist1 = qp - pp;
                const int lvl = 31 - __builtin_clz(idx);
                st1 = lane == lvl ? dist1 : st1; st2 = lane == lvl ? dist2 : st2;
                idx = (dist1 > 0.0f) ? 2 * idx + 1 : 2 * idx;
                continue;
            }
            int cidx = (int)idx;
            float cd2 = dist2;
            for (;;) {
                consider_uniform(hd, hi, lane, k, found, got_heap, r2, cidx, cd2);
                if (idx == 1) { done = true; break; }
                const unsigned child = idx;
                idx >>= 1;
                const int lvl = 31 - __builtin_clz(idx);
                const float dist1 = rl_f(st1, lvl);
                const unsigned near = (dist1 > 0.0f) ? 2 * idx + 1 : 2 * idx;
                if (child == near && dist1 * dist1 < r2) { idx = near ^ 1u; break; }
                cidx = (int)idx; cd2 = rl_f(st2, lvl);
            }
        }
        }
        if (deferred) continue;                                     // (the next launch does this query)
        // CalcIrradiance, photon.cpp:405-441: the sum over np.index[1 .. found] in that order
        float r = 0.f, g = 0.f, b = 0.f;
        if (found >= 8) {                                           // literal 8, photon.cpp:411
            const float nx = nrm3[3 * i], ny = nrm3[3 * i + 1], nz = nrm3[3 * i + 2];
            const float* cosT = tables; const float* sinT = tables + 256;
            const float* cosP = tables + 512; const float* sinP = tables + 768;
            const DevPhotonPow* pw = reinterpret_cast<const DevPhotonPow*>(ph + (n + 1));
            float wx[REDO_S], wy[REDO_S], wz[REDO_S]; int facing[REDO_S];
#pragma unroll
            for (int rr = 0; rr < REDO_S; ++rr) {                   // every lane fetches its own slots' photons ...
                const int slot = rr * 64 + lane;
                wx[rr] = wy[rr] = wz[rr] = 0.f; facing[rr] = 0;
                if (slot >= 1 && slot <= found) {
                    const int pi = hi[rr];
                    const unsigned pk = ph[pi].packed;
                    const int theta = (pk >> 8) & 255, phi = (pk >> 16) & 255;
                    const float dx = sinT[theta] * cosP[phi], dy = sinT[theta] * sinP[phi], dz = cosT[theta];
                    if ((dx * nx + dy * ny + dz * nz) < 0.0f) {
                        const float4 w4 = *reinterpret_cast<const float4*>(pw + pi);
                        wx[rr] = w4.x; wy[rr] = w4.y; wz[rr] = w4.z; facing[rr] = 1;
                    }
                }
            }
            for (int jj = 1; jj <= found; ++jj) {                   // ... and the additions run in the reference's order
                if (rd_i(facing, jj)) { r += rd_f(wx, jj); g += rd_f(wy, jj); b += rd_f(wz, jj); }
            }
            const float tmp = (float)(((double)1.0f / 3.141592) / (double)r2);   // (1.0f/PI)/(np.dist2[0])
            r *= tmp; g *= tmp; b *= tmp;
        }
        if (lane == 0) { out_rgb[3 * i] = r; out_rgb[3 * i + 1] = g; out_rgb[3 * i + 2] = b; }
    }
}

size_t rtk_knn_fast_scratch_bytes(int64_t q) {
    return (((size_t)q + 255) & ~(size_t)255) + 256 + 2 * 4 * (((size_t)q + 63) & ~(size_t)63) + rtk_photon_irr_wave_workspace(q);      // flags, two counters, two lists, the sort
}

// Fast path: wave-shared kernel (mode 1) or cooperative kernel (mode 2), then the strict kernel redoes the flagged
// queries.  d_flagged: rtk_knn_fast_scratch_bytes(q) bytes (one flag per query, then the sort's workspace).
hipError_t rtk_photon_knn_fast(const DevPhoton* ph, int64_t n, int64_t half, const float* box6, unsigned long long* d_stats,
                               const float* d_pos3, const float* d_nrm3,
                               int64_t q, int k, float maxDist, int mode, int32_t* d_idx, float* d_d2, int32_t* d_found,
                               float* d_r2, float* d_rgb, const float* d_tables, uint8_t* d_flagged, hipStream_t st, hipStream_t redo_st) {
    if (q <= 0) return hipSuccess;
    if (k > 112 || !d_flagged) return hipErrorInvalidValue;
    static const bool use_coop = getenv("RT_KNN_COOP") != nullptr;       // the 16-lane kernel for mode 1 too (experiments)
    if (mode == 1 && !use_coop) {
        // irradiance: the wave-shared kernel (kernels_photon_wave.hip), then the strict kernel redoes the flagged queries
        // the wave-shared kernel lists the queries it leaves to the strict one: [count][indices] at the head of the workspace
        unsigned* nlist = (unsigned*)(d_flagged + (((size_t)q + 255) & ~(size_t)255));
        int32_t* list = (int32_t*)(nlist + 64);
        int32_t* list2 = list + (((size_t)q + 63) & ~(size_t)63);      // the flagged queries whose walk the first strict launch gives up (nlist[1] of them)
        void* work = (void*)(list2 + (((size_t)q + 63) & ~(size_t)63));
        hipError_t e = hipMemsetAsync(nlist, 0, 8, st);
        if (e != hipSuccess) return e;
        static const bool trace = getenv("RT_KNN_TRACE") != nullptr;      // per call: queries, flagged queries, ms of the main kernel and of the redo (synchronises)
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        if (trace) { for (auto& x : ev) (void)hipEventCreate(&x); (void)hipEventRecord(ev[0], st); }
        e = rtk_photon_irr_wave(ph, n, half, box6, d_pos3, d_nrm3, q, k, maxDist, d_rgb, d_tables, list, nlist, work, d_stats, st);
        if (e != hipSuccess) return e;
        if (trace) (void)hipEventRecord(ev[1], st);
        if (k + 1 <= 64 * REDO_S - 1 && !getenv("RT_KNN_REDO_LDS")) {
            // redo_st (mode 1 only): the strict redo of the few flagged queries runs on that stream, after the main kernel --
            // a handful of waves whose walks can take 100 ms where the reference's heap has no spatial order (coplanar
            // photons), beside whatever the caller queues on `st` next.  The caller joins redo_st before it reads d_rgb.
            hipStream_t rs = st;
            if (redo_st && redo_st != st && !trace) {
                hipEvent_t done = nullptr;
                if ((e = hipEventCreateWithFlags(&done, hipEventDisableTiming)) != hipSuccess) return e;
                e = hipEventRecord(done, st);
                if (e == hipSuccess) e = hipStreamWaitEvent(redo_st, done, 0);
                (void)hipEventDestroy(done);                        // released once it has completed
                if (e != hipSuccess) return e;
                rs = redo_st;
            }
            // RT_KNN_REDO_WALK: 2 (default) = the stack walk (k_photon_knn_redo<1>), which gives up a walk that is still going
            // RT_KNN_REDO_GIVEUP (16 000, + 2 000 per doubling of the map beyond 2^28 photons: an ordinary query's walk has ~6 000 in a
            // 10 M-photon map and between 9 000 and 12 000 in 200 M -- at 9 000 that map's line loses 30 % to walks handed on for
            // nothing, at 12 000 / 16 000 / 25 000 it is unchanged and config 4's frame takes 2 272 / 2 279 / 2 305 ms) stack items after the reference's heap was built, then k_photon_knn_redo<2> over the
            // queries given up: the walk again, finished by keys after RT_KNN_REDO_SWITCH (3 000) items.  Two launches because
            // the kernel that holds the finish is a fifth slower per step of the walk (106 scalar registers), and ordinary
            // queries -- all of the kNN line's -- never need it.  1 = the stack walk alone, 0 = the stackless walk.
            const int walk_env = [] { const char* v = getenv("RT_KNN_REDO_WALK"); return v ? atoi(v) : 2; }();
            const unsigned redo_switch = [] { const char* v = getenv("RT_KNN_REDO_SWITCH"); return v ? (unsigned)atoi(v) : (unsigned)REDO_SWITCH; }();
            int give_up = [] { const char* v = getenv("RT_KNN_REDO_GIVEUP"); return v ? atoi(v) : -2; }();
            if (give_up == -2) { give_up = REDO_GIVEUP; for (int64_t m = (int64_t)1 << 28; m < n; m <<= 1) give_up += 2000; }
            const bool old_walk = walk_env == 0 || n >= ((int64_t)1 << 30);
            if (old_walk)
                hipLaunchKernelGGL(k_photon_knn_redo<0>, dim3(2048), dim3(64), 0, rs, ph, n, half, d_pos3, d_nrm3, k, maxDist, d_rgb, d_tables,
                                   (const int32_t*)list, (const unsigned*)nlist, 0u, -1, list2, nlist + 1);
            else if (walk_env == 1)
                hipLaunchKernelGGL(k_photon_knn_redo<1>, dim3(2048), dim3(64), 0, rs, ph, n, half, d_pos3, d_nrm3, k, maxDist, d_rgb, d_tables,
                                   (const int32_t*)list, (const unsigned*)nlist, 0u, -1, list2, nlist + 1);
            else {
                hipLaunchKernelGGL(k_photon_knn_redo<1>, dim3(2048), dim3(64), 0, rs, ph, n, half, d_pos3, d_nrm3, k, maxDist, d_rgb, d_tables,
                                   (const int32_t*)list, (const unsigned*)nlist, 0u, give_up, list2, nlist + 1);
                hipLaunchKernelGGL(k_photon_knn_redo<2>, dim3(512), dim3(64), 0, rs, ph, n, half, d_pos3, d_nrm3, k, maxDist, d_rgb, d_tables,
                                   (const int32_t*)list2, (const unsigned*)(nlist + 1), redo_switch, -1, nullptr, nullptr);
            }
            if (trace) {
                (void)hipEventRecord(ev[2], st); (void)hipEventSynchronize(ev[2]);
                unsigned nl = 0; (void)hipMemcpy(&nl, nlist, 4, hipMemcpyDeviceToHost);
                float a = 0.f, b = 0.f; (void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]);
                fprintf(stderr, "kNN call: %lld photons, %lld queries, r %.3g: main %.3f ms, %u flagged, redo %.3f ms\n", (long long)n, (long long)q, (double)maxDist, a, nl, b);
                for (auto& x : ev) (void)hipEventDestroy(x);
            }
            return hipGetLastError();
        }
        return launch_knn<4>(ph, n, half, d_pos3, d_nrm3, q, k, maxDist, 1, d_idx, d_d2, d_found, d_r2, d_rgb, d_tables, nullptr, st, list, nlist);
    }
    const unsigned grid = (unsigned)((q + 3) / 4);
    if (k <= 48)
        hipLaunchKernelGGL(k_photon_knn_coop<3>, dim3(grid), dim3(64), 0, st, ph, n, half, d_pos3, d_nrm3, q, k, maxDist, mode, d_idx,
                           d_d2, d_found, d_r2, d_rgb, d_tables, d_flagged);
    else
        hipLaunchKernelGGL(k_photon_knn_coop<7>, dim3(grid), dim3(64), 0, st, ph, n, half, d_pos3, d_nrm3, q, k, maxDist, mode, d_idx,
                           d_d2, d_found, d_r2, d_rgb, d_tables, d_flagged);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // redo of the (rare) flagged queries with the reference's exact list code; mode 2 -> mode 0 writes the ordered list
    return launch_knn<4>(ph, n, half, d_pos3, d_nrm3, q, k, maxDist, mode == 2 ? 0 : 1, d_idx, d_d2, d_found, d_r2, d_rgb, d_tables,
                         d_flagged, st);
}
