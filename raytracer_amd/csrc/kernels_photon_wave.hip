// kernels_photon_wave.hip -- wave-shared photon-map radiance estimate on gfx950 (the RT_KNN_FAST irradiance path).
//
// Replaces PhotonMap::CalcIrradiance over PhotonMap::LocatePhoton (Final/photon/photon.cpp:391-441, 53-131) for
// 64 queries at a time.  The strict kernel (kernels_photon.hip) replays the reference's list code per lane and ends
// up bound by vector-instruction issue with 4 useful lanes per wave.  Here the 64 queries of a wave -- neighbouring
// hit points, whose photon neighbourhoods overlap -- walk the photon kd-heap TOGETHER: records are wave-uniform
// (scalar-path fetches), the walk is the union of the lanes' walks, and all 64 lanes do useful arithmetic on every
// fetched photon.
//
// What the reference's result depends on: let S = the photons with dist2 < maxDist^2.  If |S| <= k the
// result is S and the radius stays maxDist^2.  Otherwise, when the (k+1)-th photon is accepted the reference builds
// its heap and drops the farthest of the first k accepted photons, E, even if the newcomer is farther
// (photon.cpp:84-128); from then on it keeps the k nearest seen so far.  The result is therefore the k nearest of
// S \ {E}, and the radius their largest dist2 -- given only WHICH photon E is, the visit order no longer matters.
//   phase 1  (k_irr_phase1) per lane, in the reference's visit order with the reference's (still fixed) radius: the
//            first k+1 accepted photons -> E, and tau = max(second largest of the first k, dist2 of the (k+1)-th)
//            = the reference's radius right after its heap exists, an upper bound of the final one;
//   phase 2  (k_irr_main) the wave walks the heap once: a subtree is entered if some lane's plane distance allows it
//            (delta^2 <= tau of that lane); a photon within tau of some lane goes into every such lane's sorted list
//            of the k smallest dist2, held in registers, and tau shrinks to the list's k-th entry;
//   phase 3  the wave walks again with each lane's final radius T: members (dist2 <= T, not E) are counted and
//            their power summed where the photon direction faces the surface (photon.cpp:418-429).
// Waves are started heaviest first (largest tau of phase 1): the launch ends when its longest wave does.
// Ties make the reference's set depend on its heap's history: a tie for the farthest of the first k, or more than
// k photons within T, flags the query, and the strict kernel redoes it.  The sum adds the same photons in another
// order (float association only).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "rt_internal.h"

// photons per query above which the per-lane walk is launched (RT_KNN_WALK overrides).  Since the per-lane walk reads one
// 64-byte line per step it wins at every density measured (1 M photons under 11 M queries: 72.9 against 80.6 ms per
// photon frame; 10 M under 6.7 M per launch: 5.23 against 5.52 s; 10 M under 2 M: 9.4 against 17.8 ms), so it is the default.
#define RT_KNN_LANE_RATIO 0.0
#define W_ALIVE 0x7fffffff
#define W_DEAD  (-1)

namespace {

__device__ __forceinline__ float w_dist2(const float4& rec, float qx, float qy, float qz) {
    float d1 = rec.x - qx;                      // photon.cpp:73-77: the reference's operation order
    float dist2 = d1 * d1;
    d1 = rec.y - qy; dist2 += d1 * d1;
    d1 = rec.z - qz; dist2 += d1 * d1;
    return dist2;
}

__device__ __forceinline__ float w_delta(const float4& rec, float qx, float qy, float qz) {
    const int plane = __float_as_int(rec.w) & 3;
    const float pp = plane == 0 ? rec.x : (plane == 1 ? rec.y : rec.z);
    const float qp = plane == 0 ? qx : (plane == 1 ? qy : qz);
    return qp - pp;                             // photon.cpp:61
}

// squared distance from the query to a box, in the operation order of w_dist2: for a photon inside the box every
// term is <= the photon's (float subtraction, multiplication and addition are monotone), so the result is <= its dist2
__device__ __forceinline__ float w_boxdist(const float4& lo, const float4& hi, float qx, float qy, float qz) {
    const float dx = __builtin_fmaxf(__builtin_fmaxf(lo.x - qx, qx - hi.x), 0.0f);
    const float dy = __builtin_fmaxf(__builtin_fmaxf(lo.y - qy, qy - hi.y), 0.0f);
    const float dz = __builtin_fmaxf(__builtin_fmaxf(lo.z - qz, qz - hi.z), 0.0f);
    float d = dx * dx;
    d += dy * dy;
    d += dz * dz;
    return d;
}

// sorted insertion of x into the ascending list: a chain of compare-exchanges, all in registers.  The entries are the
// BIT PATTERNS of dist2 (>= 0, or the -inf / +inf sentinels): for those, integer order is float order, and
// v_min_i32 / v_max_i32 need no NaN canonicalisation.  Blocks whose last entry no lane undercuts are skipped.
template <int KMAX>
__device__ __forceinline__ void w_insert(int (&reg)[KMAX], int x) {
    constexpr int NB = 8, KB = KMAX / NB;
    if (__ballot(x < reg[KMAX - 1]) == 0ull) return;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b + 1 < NB && __ballot(x < reg[(b + 1) * KB - 1]) == 0ull) continue;
        // reg[s] = min(reg[s], x) IN PLACE, x = max(old reg[s], x), entry after entry.  Written as C the compiler computes
        // every block into fresh registers and copies them back at the block's end (one v_mov per entry: half as much
        // again); four entries per asm statement, so that no hazard padding separates them
        int s = b * KB;
#pragma unroll
        for (; s + 4 <= (b + 1) * KB; s += 4) {
            int t;
            asm("v_max_i32_e32 %4, %0, %5\n\tv_min_i32_e32 %0, %0, %5\n\t"
                "v_max_i32_e32 %5, %1, %4\n\tv_min_i32_e32 %1, %1, %4\n\t"
                "v_max_i32_e32 %4, %2, %5\n\tv_min_i32_e32 %2, %2, %5\n\t"
                "v_max_i32_e32 %5, %3, %4\n\tv_min_i32_e32 %3, %3, %4"
                : "+v"(reg[s]), "+v"(reg[s + 1]), "+v"(reg[s + 2]), "+v"(reg[s + 3]), "=&v"(t), "+v"(x));
        }
#pragma unroll
        for (; s < (b + 1) * KB; ++s) {
            int carry;
            asm("v_max_i32_e32 %0, %1, %2\n\tv_min_i32_e32 %1, %1, %2" : "=&v"(carry), "+v"(reg[s]) : "v"(x));
            x = carry;
        }
    }
}

}  // namespace

// ---- spatial order of the queries ---------------------------------------------------------------------------------
// The 64 queries of a wave should be neighbours: the walk costs the union of their neighbourhoods.  Queries are
// sorted along a 30-bit Morton curve over the photon map's bounding box grown by the search radius (a query outside
// it finds nothing; a key per query, one radix sort of (key, index) pairs); the kernel reads its query through the sorted index and writes the result back to the query's own slot,
// so callers see no reordering.
namespace {

__device__ __forceinline__ unsigned w_spread10(unsigned v) {
    v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}

struct WBox { float lo[3], inv[3]; };

__global__ void __launch_bounds__(256) k_query_keys(const float* __restrict__ pos3, int64_t q, WBox box,
                                                    unsigned* __restrict__ keys, int32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q) return;
    unsigned key = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float u = (pos3[3 * i + a] - box.lo[a]) * box.inv[a];
        u = u > 0.f ? (u < 1.f ? u : 1.f) : 0.f;                           // also sends NaN to 0; queries outside the grid to its rim
        key |= w_spread10((unsigned)(u * 1023.0f)) << a;
    }
    keys[i] = key; vals[i] = (int32_t)i;
}

size_t w_align(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace


// ---- the kernel ------------------------------------------------------------------------------------------------------
// The walk, shared by phases 2 and 3, runs on the SECOND heap (rt_internal.h: same photons, spatially ordered
// everywhere, with the bounding box of every sub-heap): node i at `depth`, `pending` bit d = the sibling of the path's
// node at depth d is still to be visited.  Per lane: `fail` = W_ALIVE, or the depth at which the lane left the path
// (it did not want that subtree: its distance to the sub-heap's box exceeds its radius -- a far tighter test than the
// reference's single split plane, and still exact, see w_boxdist).  Children without children are tested at their
// parent; the records and boxes of both children are fetched together (contiguous), so a descent waits for one fetch.
// KMAX >= k + 1.
#ifndef IRR_SEG
#define IRR_SEG 64u
#endif
template <int KMAX, bool LANE>
__global__ void __launch_bounds__(64)
k_irr_main(const DevPhoton* __restrict__ ph, int64_t n, int64_t half, const float* __restrict__ pos3,
           const float* __restrict__ nrm3, int64_t q, int k, float maxDist, float* __restrict__ out_rgb,
           const float* __restrict__ tables, int32_t* __restrict__ redo_list, unsigned* __restrict__ redo_count,
           const int32_t* __restrict__ perm, unsigned budget, int group, unsigned long long* __restrict__ dbg) {
    const int lane = threadIdx.x;
    const long long t_begin = dbg ? wall_clock64() : 0;
    // `group` queries per wave (64; fewer in a short launch, whose waves would otherwise be few and long: lanes >= group idle)
    // Workgroups are dealt to the eight XCDs in turn (blockIdx & 7 labels the workgroups that share one, and its L2).  The
    // sorted queries are cut into segments of IRR_SEG consecutive chunks and the segments dealt to the labels, so the waves
    // resident on one XCD work on ONE neighbourhood of the map (one L2 = one neighbourhood, not an eighth of eight), while
    // all XCDs advance through the queue together.  Measured (10 M-photon kNN bench, M lookups/s / photon frame, ms):
    // plain blockIdx 194 / 66.7; eight contiguous eighths 205 / 69.5 (uneven queries: imbalance); segments of 2048: 196 /
    // 69.1, 512: 204 / 66.8, 128: 213 / 65.5, 64: 211 / 64.2, 32: 209 / 64.3, 16: 204 / 64.4.  A bijection of the workgroup
    // ids, whatever the placement really is.
    const unsigned full = (gridDim.x / (8u * IRR_SEG)) * (8u * IRR_SEG), wj = blockIdx.x >> 3;
    const unsigned chunk = blockIdx.x < full ? ((wj / IRR_SEG) * 8u + (blockIdx.x & 7u)) * IRR_SEG + (wj % IRR_SEG) : blockIdx.x;
    const int64_t qs = (int64_t)chunk * group + lane;                      // position in the spatially sorted order
    const bool valid = lane < group && qs < q;
    const int64_t qi = valid ? (perm ? (int64_t)perm[qs] : qs) : 0;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (valid) { qx = pos3[3 * qi]; qy = pos3[3 * qi + 1]; qz = pos3[3 * qi + 2]; }
    const DevPhotonPos* __restrict__ pp = ph + 2 * (n + 1);                // second heap: positions [n+2], powers [n+1]
    const DevPhotonPow* __restrict__ pq = reinterpret_cast<const DevPhotonPow*>(pp + (n + 2));
    const float4* __restrict__ bx = reinterpret_cast<const float4*>(pq + (n + 1));   // corners 2i, 2i+1 of node i's sub-heap box
    // per-lane walk: node i as ONE 64-byte line {pos, packed, box of the left child's sub-heap, box of the right child's}
    const float4* __restrict__ nodes64 = reinterpret_cast<const float4*>(ph) + rt_photon_nodes64_offset(n);
    const unsigned un = (unsigned)n, nb = (unsigned)(n / 2);               // nodes 1..nb have a child
    const float r0 = maxDist * maxDist;                                    // np.dist2[0], photon.cpp:405
    const int INF_BITS = 0x7f800000;
    const bool live = valid && n >= 1;
    bool ambiguous = false;
    unsigned d_p1 = 0, d_v2 = 0, d_c2 = 0, d_v3 = 0, d_m3 = 0, d_lanes = 0;  // measurement (dbg != NULL)

    // LANE: every lane walks the second heap for its own query (a stack of the far children in LDS) instead of the wave
    // walking the union of its 64 queries' neighbourhoods: the same pruning rule per lane, so the same photons are
    // seen.  For queries whose neighbourhoods do not overlap (a dense map under sparse queries) the union is 64 walks
    // long, and every photon on it costs the whole wave a list insertion; per lane, one insertion step serves 64
    // different photons.  The price: per-lane record fetches instead of wave-uniform ones.
    // A step of a lane = one line: the node's photon and the boxes of both children's sub-heaps come together, so the
    // children are accepted or pruned without touching their own lines; a child that has no children is a photon -- its
    // "box" is its position and the box distance its dist2, bit for bit -- and is visited right there.  The deferred
    // (farther) child of a node needs no stack entry of its own: every pending sibling lies on the current path, so a bit
    // per depth and the sibling's box distance per depth (LDS) are enough; the sibling at depth d is (i >> (depth(i) - d)) ^ 1.
    __shared__ float lane_stack[LANE ? 32 * 64 : 1];
    auto boxdist6 = [&](float lx, float ly, float lz, float hx, float hy, float hz) -> float {      // w_boxdist on loose floats
        const float dx = __builtin_fmaxf(__builtin_fmaxf(lx - qx, qx - hx), 0.0f);
        const float dy = __builtin_fmaxf(__builtin_fmaxf(ly - qy, qy - hy), 0.0f);
        const float dz = __builtin_fmaxf(__builtin_fmaxf(lz - qz, qz - hz), 0.0f);
        float d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        return d;
    };
    // visit(d2, idx, packed, own, on): one photon per lane; packed carries RT_PH_UNREACHABLE, and theta / phi when `own`
    // The next record is chosen with the bound as it is BEFORE this node's photons are visited and fetched at once, so its
    // latency runs beside the list insertions of the visits.  A bound that the visits would have tightened can only let a
    // lane enter a sub-heap it would have skipped: more photons seen, the same k + 1 smallest kept.
    auto lane_walk = [&](bool start, auto&& bound, auto&& visit, unsigned& steps) {
        unsigned i = start ? 1u : 0u, pending = 0u;
        // (a lane that is done reads record 0, which exists and is all zeros: no masked load, no registers to clear)
        const float4* __restrict__ np = nodes64 + 4 * (size_t)i;
        float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
        while (__ballot(i != 0u) != 0ull) {
            ++steps;
            const bool on = i != 0u;
            const unsigned pk = (unsigned)__float_as_int(q0.w);
            const float d_own = w_dist2(q0, qx, qy, qz);
            const unsigned c0 = 2u * i;
            const bool has_l = on && c0 <= un, has_r = on && c0 + 1u <= un;
            const float dl = boxdist6(q1.x, q1.y, q1.z, q1.w, q2.x, q2.y), dr = boxdist6(q2.z, q2.w, q3.x, q3.y, q3.z, q3.w);
            const bool leaf_l = has_l && c0 > nb, leaf_r = has_r && c0 + 1u > nb;
            unsigned next = 0u;
            if (on) {
                const float bd = bound();
                const bool wl = has_l && !leaf_l && dl <= bd, wr = has_r && !leaf_r && dr <= bd;
                const int di = 31 - __builtin_clz(i);                      // depth of i (root 0)
                if (wl && wr) {
                    const bool left_first = dl <= dr;
                    lane_stack[(di + 1) * 64 + lane] = left_first ? dr : dl;
                    pending |= 1u << (di + 1);
                    next = left_first ? c0 : c0 + 1u;
                } else if (wl) next = c0;
                else if (wr) next = c0 + 1u;
                while (next == 0u && pending != 0u) {                      // the deepest deferred sibling the lane still wants
                    const int d = 31 - __builtin_clz(pending);
                    pending &= ~(1u << d);
                    if (lane_stack[d * 64 + lane] <= bd) next = (i >> (di - d)) ^ 1u;
                }
            }
            const float4* __restrict__ nn = nodes64 + 4 * (size_t)next;
            const float4 n0 = nn[0], n1 = nn[1], n2 = nn[2], n3 = nn[3];
            visit(d_own, i, pk, true, on);
            if (dbg) d_lanes += (unsigned)(__popcll(__ballot(on)) + __popcll(__ballot(leaf_l)) + __popcll(__ballot(leaf_r)));   // photons examined
            if (__ballot(leaf_l) != 0ull) visit(dl, c0, (pk & 8u) ? RT_PH_UNREACHABLE : 0u, false, leaf_l);
            if (__ballot(leaf_r) != 0ull) visit(dr, c0 + 1u, (pk & 16u) ? RT_PH_UNREACHABLE : 0u, false, leaf_r);
            i = next; q0 = n0; q1 = n1; q2 = n2; q3 = n3;
        }
    };

    // ---- phase 2: the k+1 smallest dist2 of S per lane ---------------------------------------------------------
    int reg[KMAX];                                                         // bit patterns of dist2, ascending
#pragma unroll
    for (int s = 0; s < KMAX; ++s) reg[s] = (s < KMAX - (k + 1)) ? (int)0xff800000 : INF_BITS;   // (k+1)-th smallest: reg[KMAX-1]
    int cnt = 0;
    float tau = r0;
    // a first radius per lane, so that no lane asks for the whole heap while its list is still empty: descend to the
    // query's leaf, take the smallest sub-heap around it with at least k+1 photons the reference can reach (at most 4
    // cannot, Q8) and the largest dist2 in it -- the (k+1)-th nearest photon is no farther
    if (live) {
        unsigned leaf = 1;
        while (2 * leaf <= un) {
            const float4 rec = *reinterpret_cast<const float4*>(pp + leaf);
            leaf = 2 * leaf + ((w_delta(rec, qx, qy, qz) > 0.0f && 2 * leaf + 1 <= un) ? 1u : 0u);
        }
        unsigned root = leaf, levels = 1, have = 1;
        while (have < (unsigned)k + 5u && root > 1) {                      // nodes of the sub-heap of `root`, `levels` deep, that exist
            root >>= 1; ++levels;
            have = 0;
            for (unsigned j = 0; j < levels; ++j) {
                const unsigned first = root << j, last = first + (1u << j) - 1u;
                if (first <= un) have += (last <= un ? last : un) - first + 1u;
            }
        }
        if (have >= (unsigned)k + 5u) {
            float far = 0.f;
            for (unsigned j = 0; j < levels; ++j) {
                const unsigned first = root << j;
                for (unsigned t = 0; t < (1u << j) && first + t <= un; ++t) {
                    const float d2 = w_dist2(*reinterpret_cast<const float4*>(pp + first + t), qx, qy, qz);
                    far = d2 > far ? d2 : far;                             // NaN never wins: far stays a bound of the others
                }
            }
            if (far == far) tau = far < tau ? far : tau;
        }
    }
    bool bailed = false;
    if (__ballot(live) != 0ull) {
        auto visit = [&](const float4& r4, bool al) {                      // one photon, every lane that is in its subtree
            const float d2 = w_dist2(r4, qx, qy, qz);
            const bool cand = al && d2 <= tau && d2 < r0 && !(__float_as_int(r4.w) & (int)RT_PH_UNREACHABLE);
            if (__ballot(cand) != 0ull) {
                ++d_c2;
                cnt += cand ? 1 : 0;
                w_insert<KMAX>(reg, cand ? __float_as_int(d2) : INF_BITS);
                const float far = __int_as_float(reg[KMAX - 1]);
                tau = tau < far ? tau : far;
            }
        };
        if (LANE) {
            lane_walk(live, [&]() { return tau; },
                      [&](float d2, unsigned, unsigned pk, bool, bool al) {
                          const bool cand = al && d2 <= tau && d2 < r0 && !(pk & RT_PH_UNREACHABLE);
                          if (__ballot(cand) != 0ull) {
                              ++d_c2;
                              cnt += cand ? 1 : 0;
                              w_insert<KMAX>(reg, cand ? __float_as_int(d2) : INF_BITS);
                              const float far = __int_as_float(reg[KMAX - 1]);
                              tau = tau < far ? tau : far;
                          }
                      }, d_v2);
        } else {
        int fail = live ? W_ALIVE : W_DEAD;
        unsigned i = 1, pending = 0;
        int depth = 0;
        float4 rec = *reinterpret_cast<const float4*>(pp + 1);
        for (;;) {
            const bool al = fail == W_ALIVE;
            if (++d_v2 > budget) { bailed = true; break; }
            if (dbg) d_lanes += (unsigned)__popcll(__ballot(al));
            visit(rec, al);
            bool descended = false;
            const unsigned c0 = 2 * i;
            if (c0 <= un) {
                const bool has_r = c0 + 1 <= un;
                const float4 cl = *reinterpret_cast<const float4*>(pp + c0), cr = *reinterpret_cast<const float4*>(pp + c0 + 1);
                if (c0 > nb) {                                             // the children have none: test them here
                    visit(cl, al);
                    if (has_r) visit(cr, al);
                } else {
                    const float4 l0 = bx[2 * c0], h0 = bx[2 * c0 + 1];
                    const bool w0 = al && w_boxdist(l0, h0, qx, qy, qz) <= tau;
                    bool w1 = false;
                    if (c0 + 1 <= nb) { const float4 l1 = bx[2 * c0 + 2], h1 = bx[2 * c0 + 3]; w1 = al && w_boxdist(l1, h1, qx, qy, qz) <= tau; }
                    else if (has_r) w1 = al && w_dist2(cr, qx, qy, qz) <= tau;
                    const unsigned long long wantL = __ballot(w0), wantR = __ballot(w1);
                    if ((wantL | wantR) != 0ull) {
                        const bool both = wantL != 0ull && wantR != 0ull;
                        // the side most lanes are on first: the lists fill with close photons early and tau shrinks
                        const bool right_first = both ? 2 * __popcll(__ballot(al && w_delta(rec, qx, qy, qz) > 0.0f)) > __popcll(__ballot(al))
                                                      : wantR != 0ull;
                        ++depth;
                        if (both) pending |= 1u << depth;
                        if (al && !(right_first ? w1 : w0)) fail = depth;
                        rec = right_first ? cr : cl; i = c0 + (right_first ? 1u : 0u);
                        descended = true;
                    }
                }
            }
            if (descended) continue;
            bool done = false;
            for (;;) {                                                     // up to the next pending sibling some lane wants
                while (depth > 0 && !((pending >> depth) & 1u)) { i >>= 1; --depth; }
                if (depth == 0) { done = true; break; }
                pending &= ~(1u << depth);
                i ^= 1u;
                if (fail >= depth) fail = W_ALIVE;
                rec = *reinterpret_cast<const float4*>(pp + i);
                // does the lane want this sibling?  decided with the radius it has NOW (smaller than at the parent's visit)
                const bool want = (i <= nb ? w_boxdist(bx[2 * i], bx[2 * i + 1], qx, qy, qz) : w_dist2(rec, qx, qy, qz)) <= tau;
                if (fail == W_ALIVE && !want) fail = depth;
                if (__ballot(fail == W_ALIVE) != 0ull) break;
            }
            if (done) break;
        }
        }
    }
    // a wave whose walk does not end (64 queries that are not neighbours after all) gives up and leaves its queries to
    // the strict kernel: bounds the longest wave, which is what the launch waits for
    if (bailed && live) ambiguous = true;
#ifdef RT_KNN_TRACE_BAIL
    if (dbg && bailed) {
        unsigned long long first = 0;
        if (lane == 0) first = atomicAdd(dbg + 13, 1ull);
        first = __shfl(first, 0, 64);
        if ((first == 5 || first == 200) && live)
            printf("bail lane %d q=(%.5f %.5f %.5f) tau=%.3g cnt=%d r0=%.3g k1th=%.3g n=%lld\n", lane, qx, qy, qz, tau, cnt, r0,
                   __int_as_float(reg[KMAX - 1]), (long long)n);
    }
#endif
    const float kth = __int_as_float(reg[KMAX - 2]), k1th = __int_as_float(reg[KMAX - 1]);
    const bool evict = cnt > k;                                            // |S| >= k+1: the reference's heap exists

    // ---- phase 1: which photon the reference evicts.  Its own heap, its own order, its (still fixed) radius, until it
    //      has accepted k+1 photons; only where |S| > k, so it ends after ~k accepts -------------------------------------
    int E = -1;
    float dE = -1.f;
    if (live && evict && !bailed) {
        const unsigned uhalf = half > 0 ? (unsigned)half : 0u;
        int acc = 0;
        float m2 = -1.f;
        unsigned i = 1, child = 0;
        bool up = false, beyond = false;
        for (;;) {
            const float4 rec = *reinterpret_cast<const float4*>(ph + i);
            ++d_p1;
            if (i < uhalf) {
                const float delta = w_delta(rec, qx, qy, qz);
                const unsigned near = (delta > 0.0f) ? 2 * i + 1 : 2 * i;
                if (!up) { i = near; continue; }
                if (child == near && delta * delta < r0) { i = near ^ 1u; up = false; continue; }
            }
            const float d2 = w_dist2(rec, qx, qy, qz);                     // both subtrees done: the node itself
            if (d2 < r0) {
                if (++acc > k) break;
                // one of the first k accepted lies beyond the k-th nearest: then so does E (the farthest of them), T is
                // the k-th nearest's dist2 and E is no member -- which photon E is no longer matters
                if (d2 > kth) { beyond = true; break; }
                if (d2 > dE) { m2 = dE; dE = d2; E = (int)i; }
                else if (d2 > m2) m2 = d2;
            }
            if (i == 1) break;
            child = i; i >>= 1; up = true;
        }
        if (beyond) { E = -1; dE = 3.0e38f; }
        else if (acc <= k || dE == m2) ambiguous = true;                   // (cannot happen) / two candidates for E
    }
    // S \ {E}: if E is one of the k nearest, the (k+1)-th nearest takes its place
    const int found = evict ? k : cnt;
    const float T = evict ? (dE <= kth ? k1th : kth) : r0;                 // the reference's final np.dist2[0]

    // ---- phase 3: members within T, their power where the photon faces the surface -----------------------------
    float r = 0.f, g = 0.f, b = 0.f;
    int members = 0;
    const bool summing = live && !ambiguous && found >= 8;                 // literal 8, photon.cpp:411
    if (__ballot(summing) != 0ull) {
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (summing) { nx = nrm3[3 * qi]; ny = nrm3[3 * qi + 1]; nz = nrm3[3 * qi + 2]; }
        const float* cosT = tables; const float* sinT = tables + 256;
        const float* cosP = tables + 512; const float* sinP = tables + 768;
        auto visit = [&](const float4& r4, unsigned idx, bool al) {
            const float d2 = w_dist2(r4, qx, qy, qz);
            const unsigned pk = (unsigned)__float_as_int(r4.w);
            bool member = al && d2 <= T && d2 < r0 && !(pk & RT_PH_UNREACHABLE);
            if (__ballot(member) != 0ull) {
                ++d_m3;
                const float4 w4 = *reinterpret_cast<const float4*>(pq + idx);          // power + the reference's slot
                member = member && __float_as_int(w4.w) != E;
                members += member ? 1 : 0;
                const int theta = (pk >> 8) & 255, phi = (pk >> 16) & 255;
                const float dx = sinT[theta] * cosP[phi], dy = sinT[theta] * sinP[phi], dz = cosT[theta];
                if (member && (dx * nx + dy * ny + dz * nz) < 0.0f) { r += w4.x; g += w4.y; b += w4.z; }
            }
        };
        if (LANE) {
            lane_walk(summing, [&]() { return T; },
                      [&](float d2, unsigned idx, unsigned pk, bool own, bool al) {
                          bool member = al && d2 <= T && d2 < r0 && !(pk & RT_PH_UNREACHABLE);
                          if (__ballot(member) != 0ull) {
                              ++d_m3;
                              float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
                              if (member) {
                                  w4 = *reinterpret_cast<const float4*>(pq + idx);      // power + the reference's slot
                                  if (!own) pk = pp[idx].packed;                         // a child visited at its parent: theta / phi
                              }
                              member = member && __float_as_int(w4.w) != E;
                              members += member ? 1 : 0;
                              const int theta = (pk >> 8) & 255, phi = (pk >> 16) & 255;
                              const float dx = sinT[theta] * cosP[phi], dy = sinT[theta] * sinP[phi], dz = cosT[theta];
                              if (member && (dx * nx + dy * ny + dz * nz) < 0.0f) { r += w4.x; g += w4.y; b += w4.z; }
                          }
                      }, d_v3);
        } else {
        int fail = summing ? W_ALIVE : W_DEAD;
        unsigned i = 1, pending = 0;
        int depth = 0;
        float4 rec = *reinterpret_cast<const float4*>(pp + 1);
        for (;;) {
            const bool al = fail == W_ALIVE;
            ++d_v3;
            if (dbg) d_lanes += (unsigned)__popcll(__ballot(al));
            visit(rec, i, al);
            bool descended = false;
            const unsigned c0 = 2 * i;
            if (c0 <= un) {
                const bool has_r = c0 + 1 <= un;
                const float4 cl = *reinterpret_cast<const float4*>(pp + c0), cr = *reinterpret_cast<const float4*>(pp + c0 + 1);
                if (c0 > nb) {
                    visit(cl, c0, al);
                    if (has_r) visit(cr, c0 + 1, al);
                } else {
                    const float4 l0 = bx[2 * c0], h0 = bx[2 * c0 + 1];
                    const bool w0 = al && w_boxdist(l0, h0, qx, qy, qz) <= T;
                    bool w1 = false;
                    if (c0 + 1 <= nb) { const float4 l1 = bx[2 * c0 + 2], h1 = bx[2 * c0 + 3]; w1 = al && w_boxdist(l1, h1, qx, qy, qz) <= T; }
                    else if (has_r) w1 = al && w_dist2(cr, qx, qy, qz) <= T;
                    const unsigned long long wantL = __ballot(w0), wantR = __ballot(w1);
                    if ((wantL | wantR) != 0ull) {
                        const bool both = wantL != 0ull && wantR != 0ull;
                        const bool right_first = both ? false : wantR != 0ull;
                        ++depth;
                        if (both) pending |= 1u << depth;
                        if (al && !(right_first ? w1 : w0)) fail = depth;
                        rec = right_first ? cr : cl; i = c0 + (right_first ? 1u : 0u);
                        descended = true;
                    }
                }
            }
            if (descended) continue;
            bool done = false;
            for (;;) {
                while (depth > 0 && !((pending >> depth) & 1u)) { i >>= 1; --depth; }
                if (depth == 0) { done = true; break; }
                pending &= ~(1u << depth);
                i ^= 1u;
                if (fail >= depth) fail = W_ALIVE;
                rec = *reinterpret_cast<const float4*>(pp + i);
                const bool want = (i <= nb ? w_boxdist(bx[2 * i], bx[2 * i + 1], qx, qy, qz) : w_dist2(rec, qx, qy, qz)) <= T;
                if (fail == W_ALIVE && !want) fail = depth;
                if (__ballot(fail == W_ALIVE) != 0ull) break;
            }
            if (done) break;
        }
        }
        if (summing && members != found) ambiguous = true;                 // more (or fewer) photons within T than k: a tie
        const float tmp = (float)(((double)1.0f / 3.141592) / (double)T);  // (1.0f/PI)/(np.dist2[0])
        r *= tmp; g *= tmp; b *= tmp;
    }
    if (dbg) {
        const unsigned long long fl = __popcll(__ballot(valid && ambiguous));
        unsigned long long p1 = d_p1;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p1 += __shfl_xor(p1, off, 64);
        if (lane == 0) {
            atomicAdd(dbg + 0, p1);
            atomicAdd(dbg + 1, (unsigned long long)d_v2); atomicAdd(dbg + 2, (unsigned long long)d_c2);
            atomicAdd(dbg + 3, (unsigned long long)d_v3); atomicAdd(dbg + 4, (unsigned long long)d_m3); atomicAdd(dbg + 5, fl);
            atomicAdd(dbg + 6, 1ull);
            const unsigned long long dt = (unsigned long long)(wall_clock64() - t_begin);      // 100 MHz ticks
            atomicAdd(dbg + 7, dt);
            atomicMax(dbg + 8, (dt << 24) | (unsigned long long)(d_v2 > 0xffffffu ? 0xffffffu : d_v2));
            if (dt > 100000ull) atomicAdd(dbg + 9, 1ull);                                      // waves longer than 1 ms
            if (dt > 1000000ull) atomicAdd(dbg + 10, 1ull);                                    // longer than 10 ms
            if (bailed) atomicAdd(dbg + 11, 1ull);
            atomicAdd(dbg + 12, (unsigned long long)d_lanes);
        }
    }
    if (valid) {
        if (ambiguous) { r = 0.f; g = 0.f; b = 0.f; }
        out_rgb[3 * qi] = r; out_rgb[3 * qi + 1] = g; out_rgb[3 * qi + 2] = b;
        if (ambiguous) redo_list[atomicAdd(redo_count, 1u)] = (int32_t)qi;             // rare: exact ties, give-ups
    }
}

namespace {
struct WaveWork {            // carved from the caller's workspace
    unsigned *keys_in, *keys_out; int32_t *vals_in, *vals_out; void* temp; size_t temp_bytes, total;
};
WaveWork carve(void* base, int64_t q) {
    WaveWork W;
    size_t t1 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (const unsigned*)nullptr, (unsigned*)nullptr, (const int32_t*)nullptr,
                                             (int32_t*)nullptr, (int)q, 0, 30, (hipStream_t)nullptr);
    char* w = (char*)base;
    auto take = [&w](size_t bytes) { char* p = w; w += w_align(bytes); return (void*)p; };
    W.keys_in = (unsigned*)take((size_t)q * 4); W.keys_out = (unsigned*)take((size_t)q * 4);
    W.vals_in = (int32_t*)take((size_t)q * 4); W.vals_out = (int32_t*)take((size_t)q * 4);
    W.temp_bytes = t1;
    W.temp = take(W.temp_bytes);
    W.total = (size_t)(w - (char*)base) + 256;
    return W;
}
}  // namespace

// workspace of rtk_photon_irr_wave for q queries (sort keys / indices and the sort's own storage)
size_t rtk_photon_irr_wave_workspace(int64_t q) { return carve(nullptr, q).total; }

// k <= 127.  d_redo_list [q] / d_redo_count (zeroed by the caller): the queries left to the strict kernel.  d_work:
// rtk_photon_irr_wave_workspace(q) bytes.  d_stats: NULL, or 16 counters that the kernel adds to (rt_scene_knn_stats):
//   [0] phase-1 record fetches (all lanes)   [1] phase-2 node visits (per wave)   [2] photons inserted into some list
//   [3] phase-3 node visits   [4] photons that were a member for some lane   [5] queries flagged for the strict redo
//   [6] waves   [7] sum of wave times, 10 ns ticks   [8] longest wave << 24 | its phase-2 visits   [9] waves > 1 ms
//   [10] waves > 10 ms   [11] waves that gave up   [12] node visits summed over the lanes that were in the subtree
hipError_t rtk_photon_irr_wave(const DevPhoton* ph, int64_t n, int64_t half, const float* box6, const float* d_pos3, const float* d_nrm3,
                               int64_t q, int k, float maxDist, float* d_rgb, const float* d_tables, int32_t* d_redo_list,
                               unsigned* d_redo_count, void* d_work, unsigned long long* d_stats, hipStream_t st) {
    if (q <= 0) return hipSuccess;
    if (k < 1 || k > 127 || !d_redo_list || !d_redo_count || !d_work || q > 0x7fffffff) return hipErrorInvalidValue;
    const WaveWork W = carve(d_work, q);
    // queries per wave: 64 shares most, but a launch of few waves ends when its longest walk does, and a walk is
    // serial (~0.8 us per node when the chip is not full): short launches get narrower waves, so more and shorter ones
    static const int group_env = getenv("RT_KNN_GROUP") ? atoi(getenv("RT_KNN_GROUP")) : 0;
    int group = q >= 64 * 8192 ? 64 : (q >= 32 * 8192 ? 32 : (q >= 16 * 4096 ? 16 : 8));
    if (group_env == 64 || group_env == 32 || group_env == 16 || group_env == 8) group = group_env;
    static const bool no_sort = getenv("RT_KNN_NOSORT") != nullptr;
    const int32_t* perm = nullptr;
    if (q > 64 && !no_sort) {
        size_t temp = W.temp_bytes;
        WBox box;
        for (int a = 0; a < 3; ++a) {
            const float lo = (box6 ? box6[a] : 0.f) - maxDist, hi = (box6 ? box6[3 + a] : 0.f) + maxDist;
            box.lo[a] = lo; box.inv[a] = hi > lo ? 1.0f / (hi - lo) : 0.f;
        }
        hipLaunchKernelGGL(k_query_keys, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, d_pos3, q, box, W.keys_in, W.vals_in);
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(W.temp, temp, W.keys_in, W.keys_out, W.vals_in, W.vals_out, (int)q, 0, 30, st);
        if (e != hipSuccess) return e;
        perm = W.vals_out;
    }
    // phase-2 node visits a wave may spend before it gives up (the mean is 1 000 - 2 000 at k = 100): the launch
    // ends when its longest wave does, and the strict kernel spreads the 64 queries over 16 waves.
    const unsigned budget_env = getenv("RT_KNN_BUDGET") ? (unsigned)atoi(getenv("RT_KNN_BUDGET")) : 0u;      // (read at every call: the tests set it)
    const unsigned budget = budget_env ? budget_env : 16384u;
    // which walk: RT_KNN_WALK=wave|lane (read at every call), else by the photons per query: above 3 the queries'
    // neighbourhoods hardly overlap and the per-lane walk wins (10 M photons under 2.07 M queries: 117 against 104 M
    // lookups/s; 200 M photons: 74 against 20; the photon frame's 1 M-photon map under denser queries: 70 against 79)
    const int walk_env = [] { const char* e = getenv("RT_KNN_WALK"); return !e ? 0 : (strcmp(e, "lane") == 0 ? 2 : (strcmp(e, "wave") == 0 ? 1 : 0)); }();
    const bool lane_walk = walk_env ? walk_env == 2 : (double)n >= RT_KNN_LANE_RATIO * (double)q;
    if (lane_walk) { group = 64; }
    const unsigned grid2 = (unsigned)((q + group - 1) / group);
#define WAVE_ARGS dim3(grid2), dim3(64), 0, st, ph, n, half, d_pos3, d_nrm3, q, k, maxDist, d_rgb, d_tables, d_redo_list, d_redo_count, perm, budget, group, d_stats
    if (lane_walk) {
        if (k <= 31) hipLaunchKernelGGL((k_irr_main<32, true>), WAVE_ARGS);
        else if (k <= 63) hipLaunchKernelGGL((k_irr_main<64, true>), WAVE_ARGS);
        else if (k <= 103) hipLaunchKernelGGL((k_irr_main<104, true>), WAVE_ARGS);
        else hipLaunchKernelGGL((k_irr_main<128, true>), WAVE_ARGS);
    } else {
        if (k <= 31) hipLaunchKernelGGL((k_irr_main<32, false>), WAVE_ARGS);
        else if (k <= 63) hipLaunchKernelGGL((k_irr_main<64, false>), WAVE_ARGS);
        else if (k <= 103) hipLaunchKernelGGL((k_irr_main<104, false>), WAVE_ARGS);
        else hipLaunchKernelGGL((k_irr_main<128, false>), WAVE_ARGS);
    }
#undef WAVE_ARGS
    return hipGetLastError();
}
