// photon_api.cpp -- photon-map entry points of include/rt_hip.h: upload of a
// balanced map (final.cpp:1125-1129 hands it to the renderer), host balance
// (photon.cpp:133-254), kNN (photon.cpp:53-131) and irradiance (:342-443).
#include "rt_scene.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#define HIPCHK(expr, what) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return rt_hipfail(e__, what); } while (0)
#define RTCHK(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

namespace {

enum { S_Q_POS = 0, S_Q_NRM = 1, S_Q_IDX = 2, S_Q_D2 = 3, S_Q_FOUND = 4, S_Q_R2 = 5, S_Q_RGB = 6, S_Q_FLAG = 11 };
const int KNN_K_MAX = 287;       // (2(k+1)+64)*256 B of LDS must fit 160 KiB

// InitDirectionTable, photon.cpp:40-50 (PI = 3.141592, parameter.h:6)
int ensure_tables(rt_scene* s) {
    if (s->d_dir_tables) return RT_OK;
    float t[1024];
    for (int i = 0; i < 256; ++i) {
        const double angle = double(i) * (1.0 / 256.0) * 3.141592;
        t[i] = (float)std::cos(angle); t[256 + i] = (float)std::sin(angle);
        t[512 + i] = (float)std::cos(2.0 * angle); t[768 + i] = (float)std::sin(2.0 * angle);
    }
    void* p; RTCHK(s->dev_alloc(&p, sizeof t));
    HIPCHK(hipMemcpy(p, t, sizeof t, hipMemcpyHostToDevice), "copy tables");
    s->d_dir_tables = (float*)p;
    return RT_OK;
}

int check_map(rt_scene* s, int which, int k, const char* fn) {
    if (!s || which < 0 || which > 1) return rt_fail(RT_E_INVALID, "%s: bad scene or map selector", fn);
    if (k < 1 || k > KNN_K_MAX) return rt_fail(RT_E_LIMIT, "%s: k=%d outside [1,%d]", fn, k, KNN_K_MAX);
    if (!s->pmap[which].d_photons) return rt_fail(RT_E_STATE, "%s: photon map %d not uploaded", fn, which);
    return RT_OK;
}

// Where rt_photonmap_build balances and assembles (the same bytes either way, tests/test_gpu_photon.py): on the device
// (photon_build_dev.hip; 200 M photons 3.1 s against 21.6 s for the host's first heap alone) from 64 K photons up -- below
// that the host's threads are done before the kernels' launches are.  RT_PHOTON_BUILD=host|device overrides.
bool photon_build_on_device(int64_t n) {
    const char* e = getenv("RT_PHOTON_BUILD");
    if (e && strcmp(e, "device") == 0) return true;
    if (e && strcmp(e, "host") == 0) return false;
    return n >= 65536;
}

int upload_heap(rt_scene* s, int which, const rt_photon* heap, int64_t n) {
    PhotonMapDev& m = s->pmap[which];
    if (m.d_photons) { (void)hipFree(m.d_photons); m.d_photons = nullptr; m.n = 0; }
    std::vector<DevPhotonPos> dev;                  // [n+1] positions, [n+1] powers; second heap: [n+2] positions, [n+1] powers,
    std::vector<int32_t> src; std::vector<int8_t> pplane;   // [2 (n/2 + 2)] box corners
    const size_t nb = (size_t)n / 2;                // nodes of the second heap that have a child
    try {
        dev.resize(rt_photon_image_records(n));
        rt_photon_product_order(heap, n, src, pplane);
    } catch (const std::bad_alloc&) { return rt_fail(RT_E_NOMEM, "host allocation failed"); }
    memset(dev.data(), 0, sizeof(DevPhotonPos));
    DevPhotonPow* pw = reinterpret_cast<DevPhotonPow*>(dev.data() + (size_t)n + 1);
    memset(pw, 0, sizeof(DevPhotonPow));
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int64_t i = 1; i <= n; ++i) {
        const rt_photon& p = heap[i];
        DevPhotonPos& d = dev[(size_t)i];
        for (int a = 0; a < 3; ++a)
            if (p.position[a] > -1e30f && p.position[a] < 1e30f) { lo[a] = std::min(lo[a], p.position[a]); hi[a] = std::max(hi[a], p.position[a]); }
        memcpy(d.pos, p.position, 12); memcpy(pw[i].power, p.power, 12);
        d.packed = ((uint32_t)p.plane & 3u) | ((uint32_t)p.theta << 8) | ((uint32_t)p.phi << 16);
        pw[i].pad = 0;
    }
    {   // the second heap: same photons, spatially ordered everywhere
        DevPhotonPos* pp = dev.data() + 2 * ((size_t)n + 1);
        DevPhotonPow* pq = reinterpret_cast<DevPhotonPow*>(pp + (size_t)n + 2);
        memset(pp, 0, sizeof(DevPhotonPos)); memset(pp + (size_t)n + 1, 0, sizeof(DevPhotonPos)); memset(pq, 0, sizeof(DevPhotonPow));
        const int64_t half = n / 2 - 1;
        for (int64_t j = 1; j <= n; ++j) {
            const int64_t r = src[(size_t)j];
            const rt_photon& p = heap[r];
            const bool reachable = r == 1 || (r >> 1) < half;          // photon.cpp:58: children only below half_stored_photons
            memcpy(pp[j].pos, p.position, 12); memcpy(pq[j].power, p.power, 12);
            pp[j].packed = ((uint32_t)pplane[(size_t)j] & 3u) | (reachable ? 0u : RT_PH_UNREACHABLE) | ((uint32_t)p.theta << 8) | ((uint32_t)p.phi << 16);
            pq[j].pad = (uint32_t)r;
        }
        // bounding boxes of the sub-heaps, bottom up: box[i] = photon i + box[2i] + box[2i+1]; corners as two 16-byte records
        DevPhotonPos* bx = reinterpret_cast<DevPhotonPos*>(pq + (size_t)n + 1);
        memset(bx, 0, 2 * (nb + 2) * sizeof(DevPhotonPos));
        for (int64_t i = (int64_t)nb; i >= 1; --i) {
            float lo[3], hi[3];
            for (int a = 0; a < 3; ++a) lo[a] = hi[a] = pp[i].pos[a];
            for (int64_t c = 2 * i; c <= 2 * i + 1 && c <= n; ++c)
                for (int a = 0; a < 3; ++a) {
                    const float cl = c <= (int64_t)nb ? bx[2 * c].pos[a] : pp[c].pos[a], ch = c <= (int64_t)nb ? bx[2 * c + 1].pos[a] : pp[c].pos[a];
                    lo[a] = std::min(lo[a], cl); hi[a] = std::max(hi[a], ch);
                }
            memcpy(bx[2 * i].pos, lo, 12); memcpy(bx[2 * i + 1].pos, hi, 12);
        }
        // the per-lane walk's 64-byte node records (rt_internal.h): own photon + both children's sub-heap boxes
        float* nd = reinterpret_cast<float*>(dev.data() + rt_photon_nodes64_offset(n));
        memset(nd, 0, 64);
        for (int64_t i = 1; i <= n; ++i) {
            float* o = nd + 16 * (size_t)i;
            memcpy(o, pp[i].pos, 12);
            uint32_t pk = pp[i].packed & ~(8u | 16u);
            for (int side = 0; side < 2; ++side) {
                const int64_t c = 2 * i + side;
                float* b = o + 4 + 6 * side;
                if (c > n) { for (int a = 0; a < 6; ++a) b[a] = 0.f; continue; }
                if (c <= (int64_t)nb) { memcpy(b, bx[2 * c].pos, 12); memcpy(b + 3, bx[2 * c + 1].pos, 12); }
                else {
                    memcpy(b, pp[c].pos, 12); memcpy(b + 3, pp[c].pos, 12);
                    if (pp[c].packed & RT_PH_UNREACHABLE) pk |= side ? 16u : 8u;
                }
            }
            memcpy(o + 3, &pk, 4);
        }
    }
    hipError_t e = hipMalloc((void**)&m.d_photons, dev.size() * sizeof(DevPhotonPos));
    if (e != hipSuccess) return rt_fail(RT_E_NOMEM, "hipMalloc(photon map %zu B): %s", dev.size() * sizeof(DevPhotonPos), hipGetErrorString(e));
    HIPCHK(hipMemcpy(m.d_photons, dev.data(), dev.size() * sizeof(DevPhotonPos), hipMemcpyHostToDevice), "copy photon map");
    m.n = n;
    m.half = n / 2 - 1;                       // photon.cpp:168
    for (int a = 0; a < 3; ++a) { m.box[a] = lo[a] <= hi[a] ? lo[a] : 0.f; m.box[3 + a] = lo[a] <= hi[a] ? hi[a] : 0.f; }
    return ensure_tables(s);
}

}  // namespace

extern "C" {

int rt_photonmap_upload(rt_scene* s, int which, const rt_photon* heap_order, int64_t n) {
    if (!s || which < 0 || which > 1 || n < 0 || !heap_order) return rt_fail(RT_E_INVALID, "rt_photonmap_upload: bad argument");
    if (n >= ((int64_t)1 << 31) - 2) return rt_fail(RT_E_LIMIT, "rt_photonmap_upload: more than 2^31 photons");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    return upload_heap(s, which, heap_order, n);
}

int rt_photonmap_build(rt_scene* s, int which, rt_photon* photons, int64_t n, float power_scale) {
    if (!s || which < 0 || which > 1 || n < 0 || !photons) return rt_fail(RT_E_INVALID, "rt_photonmap_build: bad argument");
    if (n >= ((int64_t)1 << 31) - 2) return rt_fail(RT_E_LIMIT, "rt_photonmap_build: more than 2^31 photons");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    if (photon_build_on_device(n)) {                       // the same arrays, built by the kernels of photon_build_dev.hip
        void* d = nullptr;
        const size_t bytes = ((size_t)n + 1) * sizeof(rt_photon);
        if (hipMalloc(&d, bytes) != hipSuccess) return rt_fail(RT_E_NOMEM, "hipMalloc(%zu B of photons) failed", bytes);
        int rc = RT_OK;
        if (hipMemcpy(d, photons, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = rt_fail(RT_E_HIP, "copy photons");
        if (!rc) rc = rt_photonmap_build_dev(s, which, (rt_photon*)d, n, power_scale, nullptr);
        if (!rc && hipMemcpy(photons, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = rt_fail(RT_E_HIP, "copy photons back");
        (void)hipFree(d);
        return rc;
    }
    try { rt_photon_balance_host(photons, n); } catch (const std::bad_alloc&) { return rt_fail(RT_E_NOMEM, "host allocation failed"); }
    if (power_scale != 1.0f)                              // ScalePhotonPower(scale, 0, n): slots 0..n-1, photon.cpp:826-833
        for (int64_t i = 0; i < n; ++i)
            for (int j = 0; j < 3; ++j) photons[i].power[j] = photons[i].power[j] * power_scale;
    return upload_heap(s, which, photons, n);
}

// The same build with the photons resident on the device (rt_photonmap_emit_dev's output): balance, scale and image by
// the kernels of photon_build_dev.hip; d_photons comes back in the reference's heap order, as rt_photonmap_build leaves
// its host array.
int rt_photonmap_build_dev(rt_scene* s, int which, rt_photon* d_photons, int64_t n, float power_scale, void* stream) {
    if (!s || which < 0 || which > 1 || n < 0 || !d_photons) return rt_fail(RT_E_INVALID, "rt_photonmap_build_dev: bad argument");
    if (n >= ((int64_t)1 << 31) - 2) return rt_fail(RT_E_LIMIT, "rt_photonmap_build_dev: more than 2^31 photons");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    PhotonMapDev& m = s->pmap[which];
    if (m.d_photons) { (void)hipFree(m.d_photons); m.d_photons = nullptr; m.n = 0; }
    DevPhotonPos* img = nullptr; float box[6]; int status = 0;
    const hipError_t e = rtk_photon_build_device(d_photons, n, power_scale, &img, box, (hipStream_t)stream, &status);
    if (e == hipErrorOutOfMemory) return rt_fail(RT_E_NOMEM, "rt_photonmap_build_dev: device allocation failed");
    if (e != hipSuccess) return rt_hipfail(e, "rt_photonmap_build_dev");
    if (status) return rt_fail(RT_E_HIP, "rt_photonmap_build_dev: work queue overflow");
    m.d_photons = img; m.n = n; m.half = n / 2 - 1;
    memcpy(m.box, box, sizeof m.box);
    return ensure_tables(s);
}

// ---- one build per node: the device image of a map, copied between ranks' GPUs ---------------------------------------
static size_t image_records(int64_t n) { return rt_photon_image_records(n); }      // as upload_heap lays it out

int rt_photonmap_image_bytes(rt_scene* s, int which, int64_t* nphotons, int64_t* bytes) {
    if (!s || which < 0 || which > 1 || !nphotons || !bytes) return rt_fail(RT_E_INVALID, "rt_photonmap_image_bytes: bad argument");
    const PhotonMapDev& m = s->pmap[which];
    *nphotons = m.d_photons ? m.n : 0;
    *bytes = m.d_photons ? (int64_t)(image_records(m.n) * sizeof(DevPhotonPos)) : 0;
    return RT_OK;
}

int rt_photonmap_export_dev(rt_scene* s, int which, void* d_dst, int64_t cap_bytes, float box6[6], void* stream) {
    if (!s || which < 0 || which > 1 || !d_dst || !box6) return rt_fail(RT_E_INVALID, "rt_photonmap_export_dev: bad argument");
    const PhotonMapDev& m = s->pmap[which];
    if (!m.d_photons) return rt_fail(RT_E_STATE, "rt_photonmap_export_dev: photon map %d not uploaded", which);
    const size_t bytes = image_records(m.n) * sizeof(DevPhotonPos);
    if ((int64_t)bytes > cap_bytes) return rt_fail(RT_E_INVALID, "rt_photonmap_export_dev: %zu B do not fit %lld", bytes, (long long)cap_bytes);
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    HIPCHK(hipMemcpyAsync(d_dst, m.d_photons, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream), "copy photon map image");
    memcpy(box6, m.box, sizeof m.box);
    return RT_OK;
}

int rt_photonmap_import_dev(rt_scene* s, int which, const void* d_src, int64_t nphotons, int64_t bytes, const float box6[6], void* stream) {
    if (!s || which < 0 || which > 1 || !d_src || !box6 || nphotons < 0) return rt_fail(RT_E_INVALID, "rt_photonmap_import_dev: bad argument");
    if (nphotons >= ((int64_t)1 << 31) - 2) return rt_fail(RT_E_LIMIT, "rt_photonmap_import_dev: more than 2^31 photons");
    if ((int64_t)(image_records(nphotons) * sizeof(DevPhotonPos)) != bytes)
        return rt_fail(RT_E_INVALID, "rt_photonmap_import_dev: %lld B is not the image of %lld photons", (long long)bytes, (long long)nphotons);
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    PhotonMapDev& m = s->pmap[which];
    if (m.d_photons) { (void)hipFree(m.d_photons); m.d_photons = nullptr; m.n = 0; }
    hipError_t e = hipMalloc((void**)&m.d_photons, (size_t)bytes);
    if (e != hipSuccess) return rt_fail(RT_E_NOMEM, "hipMalloc(photon map %lld B): %s", (long long)bytes, hipGetErrorString(e));
    HIPCHK(hipMemcpyAsync(m.d_photons, d_src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream), "copy photon map image");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream), "sync");             // the caller may free d_src when this returns
    m.n = nphotons; m.half = nphotons / 2 - 1;
    memcpy(m.box, box6, sizeof m.box);
    return ensure_tables(s);
}

int rt_host_photon_balance(rt_photon* photons, int64_t n) {
    if (!photons || n < 0) return rt_fail(RT_E_INVALID, "rt_host_photon_balance: bad argument");
    try { rt_photon_balance_host(photons, n); } catch (const std::bad_alloc&) { return rt_fail(RT_E_NOMEM, "host allocation failed"); }
    return RT_OK;
}

int rt_photon_knn_dev(rt_scene* s, int which, const float* d_pos3, int64_t q, int k, float maxDist, int32_t* d_idx,
                      float* d_dist2, int32_t* d_found, float* d_radius2, void* stream) {
    RTCHK(check_map(s, which, k, "rt_photon_knn_dev"));
    if (q < 0 || (q > 0 && (!d_pos3 || !d_idx || !d_dist2 || !d_found))) return rt_fail(RT_E_INVALID, "rt_photon_knn_dev: bad argument");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    const PhotonMapDev& m = s->pmap[which];
    HIPCHK(rtk_photon_knn(m.d_photons, m.n, m.half, d_pos3, nullptr, q, k, maxDist, 0, d_idx, d_dist2, d_found, d_radius2,
                          nullptr, s->d_dir_tables, (hipStream_t)stream), "k_photon_knn");
    return RT_OK;
}

int rt_photon_knn(rt_scene* s, int which, const float* pos3, int64_t q, int k, float maxDist, int32_t* idx, float* dist2,
                  int32_t* found, float* radius2) {
    RTCHK(check_map(s, which, k, "rt_photon_knn"));
    if (q < 0 || (q > 0 && (!pos3 || !idx || !dist2 || !found))) return rt_fail(RT_E_INVALID, "rt_photon_knn: bad argument");
    if (q == 0) return RT_OK;
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    void *dp, *di, *dd, *df, *dr;
    RTCHK(s->scratch(S_Q_POS, (size_t)q * 12, &dp));
    RTCHK(s->scratch(S_Q_IDX, (size_t)q * k * 4, &di));
    RTCHK(s->scratch(S_Q_D2, (size_t)q * k * 4, &dd));
    RTCHK(s->scratch(S_Q_FOUND, (size_t)q * 4, &df));
    RTCHK(s->scratch(S_Q_R2, (size_t)q * 4, &dr));
    HIPCHK(hipMemcpy(dp, pos3, (size_t)q * 12, hipMemcpyHostToDevice), "copy queries");
    const PhotonMapDev& m = s->pmap[which];
    HIPCHK(rtk_photon_knn(m.d_photons, m.n, m.half, (const float*)dp, nullptr, q, k, maxDist, 0, (int32_t*)di, (float*)dd,
                          (int32_t*)df, (float*)dr, nullptr, s->d_dir_tables, nullptr), "k_photon_knn");
    HIPCHK(hipMemcpy(idx, di, (size_t)q * k * 4, hipMemcpyDeviceToHost), "copy idx");
    HIPCHK(hipMemcpy(dist2, dd, (size_t)q * k * 4, hipMemcpyDeviceToHost), "copy dist2");
    HIPCHK(hipMemcpy(found, df, (size_t)q * 4, hipMemcpyDeviceToHost), "copy found");
    if (radius2) HIPCHK(hipMemcpy(radius2, dr, (size_t)q * 4, hipMemcpyDeviceToHost), "copy radius2");
    return RT_OK;
}

int rt_photon_knn_set(rt_scene* s, int which, const float* pos3, int64_t q, int k, float maxDist, int32_t* idx, float* dist2,
                  int32_t* found, float* radius2) {
    RTCHK(check_map(s, which, k, "rt_photon_knn_set"));
    if (q < 0 || (q > 0 && (!pos3 || !idx || !dist2 || !found))) return rt_fail(RT_E_INVALID, "rt_photon_knn_set: bad argument");
    if (q == 0) return RT_OK;
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    void *dp, *di, *dd, *df, *dr;
    RTCHK(s->scratch(S_Q_POS, (size_t)q * 12, &dp));
    RTCHK(s->scratch(S_Q_IDX, (size_t)q * k * 4, &di));
    RTCHK(s->scratch(S_Q_D2, (size_t)q * k * 4, &dd));
    RTCHK(s->scratch(S_Q_FOUND, (size_t)q * 4, &df));
    RTCHK(s->scratch(S_Q_R2, (size_t)q * 4, &dr));
    HIPCHK(hipMemcpy(dp, pos3, (size_t)q * 12, hipMemcpyHostToDevice), "copy queries");
    const PhotonMapDev& m = s->pmap[which];
    if (s->knn_fast && k <= RT_KNN_FAST_KMAX) {
        void* fl; RTCHK(s->scratch(S_Q_FLAG, rtk_knn_fast_scratch_bytes(q), &fl));
        HIPCHK(rtk_photon_knn_fast(m.d_photons, m.n, m.half, m.box, s->d_knn_stats, (const float*)dp, nullptr, q, k, maxDist, 2, (int32_t*)di, (float*)dd,
                                   (int32_t*)df, (float*)dr, nullptr, s->d_dir_tables, (uint8_t*)fl, nullptr), "k_photon_knn_coop");
    } else
    HIPCHK(rtk_photon_knn(m.d_photons, m.n, m.half, (const float*)dp, nullptr, q, k, maxDist, 0, (int32_t*)di, (float*)dd,
                          (int32_t*)df, (float*)dr, nullptr, s->d_dir_tables, nullptr), "k_photon_knn");
    HIPCHK(hipMemcpy(idx, di, (size_t)q * k * 4, hipMemcpyDeviceToHost), "copy idx");
    HIPCHK(hipMemcpy(dist2, dd, (size_t)q * k * 4, hipMemcpyDeviceToHost), "copy dist2");
    HIPCHK(hipMemcpy(found, df, (size_t)q * 4, hipMemcpyDeviceToHost), "copy found");
    if (radius2) HIPCHK(hipMemcpy(radius2, dr, (size_t)q * 4, hipMemcpyDeviceToHost), "copy radius2");
    return RT_OK;
}

int rt_photon_irradiance_dev(rt_scene* s, int which, const float* d_pos3, const float* d_nrm3, int64_t q, int k,
                             float maxDist, float* d_rgb, void* stream) {
    RTCHK(check_map(s, which, k, "rt_photon_irradiance_dev"));
    if (q < 0 || (q > 0 && (!d_pos3 || !d_nrm3 || !d_rgb))) return rt_fail(RT_E_INVALID, "rt_photon_irradiance_dev: bad argument");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    const PhotonMapDev& m = s->pmap[which];
    if (s->knn_fast && k <= RT_KNN_FAST_KMAX) {
        void* fl; RTCHK(s->scratch(S_Q_FLAG, rtk_knn_fast_scratch_bytes(q), &fl));
        HIPCHK(rtk_photon_knn_fast(m.d_photons, m.n, m.half, m.box, s->d_knn_stats, d_pos3, d_nrm3, q, k, maxDist, 1, nullptr, nullptr, nullptr, nullptr,
                                   d_rgb, s->d_dir_tables, (uint8_t*)fl, (hipStream_t)stream), "k_photon_knn_coop");
        return RT_OK;
    }
    HIPCHK(rtk_photon_knn(m.d_photons, m.n, m.half, d_pos3, d_nrm3, q, k, maxDist, 1, nullptr, nullptr, nullptr, nullptr,
                          d_rgb, s->d_dir_tables, (hipStream_t)stream), "k_photon_knn");
    return RT_OK;
}

int rt_photon_irradiance(rt_scene* s, int which, const float* pos3, const float* nrm3, int64_t q, int k, float maxDist,
                         float* rgb) {
    RTCHK(check_map(s, which, k, "rt_photon_irradiance"));
    if (q < 0 || (q > 0 && (!pos3 || !nrm3 || !rgb))) return rt_fail(RT_E_INVALID, "rt_photon_irradiance: bad argument");
    if (q == 0) return RT_OK;
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    void *dp, *dn, *dc;
    RTCHK(s->scratch(S_Q_POS, (size_t)q * 12, &dp));
    RTCHK(s->scratch(S_Q_NRM, (size_t)q * 12, &dn));
    RTCHK(s->scratch(S_Q_RGB, (size_t)q * 12, &dc));
    HIPCHK(hipMemcpy(dp, pos3, (size_t)q * 12, hipMemcpyHostToDevice), "copy queries");
    HIPCHK(hipMemcpy(dn, nrm3, (size_t)q * 12, hipMemcpyHostToDevice), "copy normals");
    const PhotonMapDev& m = s->pmap[which];
    if (s->knn_fast && k <= RT_KNN_FAST_KMAX) {
        void* fl; RTCHK(s->scratch(S_Q_FLAG, rtk_knn_fast_scratch_bytes(q), &fl));
        HIPCHK(rtk_photon_knn_fast(m.d_photons, m.n, m.half, m.box, s->d_knn_stats, (const float*)dp, (const float*)dn, q, k, maxDist, 1, nullptr, nullptr,
                                   nullptr, nullptr, (float*)dc, s->d_dir_tables, (uint8_t*)fl, nullptr), "k_photon_knn_coop");
    } else
    HIPCHK(rtk_photon_knn(m.d_photons, m.n, m.half, (const float*)dp, (const float*)dn, q, k, maxDist, 1, nullptr, nullptr,
                          nullptr, nullptr, (float*)dc, s->d_dir_tables, nullptr), "k_photon_knn");
    HIPCHK(hipMemcpy(rgb, dc, (size_t)q * 12, hipMemcpyDeviceToHost), "copy rgb");
    return RT_OK;
}

static int emit_impl(rt_scene* s, const rt_emit_params* e, int64_t first, int64_t count, rt_photon* out, int out_is_device,
                     int64_t cap, int64_t* nstored, hipStream_t st) {
    if (!s || !e || first < 0 || count < 0 || cap < 0 || !nstored || (cap > 0 && !out))
        return rt_fail(RT_E_INVALID, "rt_photonmap_emit: bad argument");
    if (e->which < 0 || e->which > 1 || e->light < 0 || e->light >= s->view.nlights || e->ncaustic_sphere < 0 ||
        e->ncaustic_sphere > 4 || e->max_depth < 0 || e->max_depth > 6)
        return rt_fail(RT_E_INVALID, "rt_photonmap_emit: bad emission parameters (map %d, light %d of %d, depth %d)", e->which,
                       e->light, s->view.nlights, e->max_depth);
    if (count > ((int64_t)1 << 30)) return rt_fail(RT_E_LIMIT, "rt_photonmap_emit: more than 2^30 emissions in one call");
    *nstored = 0;
    if (count == 0) return RT_OK;
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    enum { S_E_COUNTS = 7, S_E_BSUMS = 8, S_E_OFFSETS = 9, S_E_OUT = 10 };
    const int64_t nb = rtk_scan_blocks_needed(count);
    void *dc, *db, *doff, *dout = nullptr;
    RTCHK(s->scratch(S_E_COUNTS, (size_t)count, &dc));
    RTCHK(s->scratch(S_E_BSUMS, (size_t)nb * 4 + 16, &db));
    RTCHK(s->scratch(S_E_OFFSETS, (size_t)count * 4, &doff));
    uint32_t* d_total = (uint32_t*)db + nb;
    // pass 0: how many photons does each emission index store; scan -> where they go
    HIPCHK(rtk_emit_photons(s->view, *e, first, count, s->eps_f, s->fast, 0, (uint8_t*)dc, nullptr, nullptr, 0, st), "k_emit_photons");
    HIPCHK(rtk_scan_flags((const uint8_t*)dc, count, (uint32_t*)db, d_total, st), "scan");
    HIPCHK(rtk_count_offsets((const uint8_t*)dc, count, (const uint32_t*)db, (uint32_t*)doff, st), "k_count_offsets");
    uint32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, st), "read photon count");
    HIPCHK(hipStreamSynchronize(st), "sync");
    *nstored = (int64_t)total;
    const int64_t nw = (int64_t)total < cap ? (int64_t)total : cap;
    if (nw == 0) return RT_OK;
    if (out_is_device) dout = out;
    else RTCHK(s->scratch(S_E_OUT, (size_t)nw * sizeof(rt_photon), &dout));
    // pass 1: the same walks again, photons written in emission order
    HIPCHK(rtk_emit_photons(s->view, *e, first, count, s->eps_f, s->fast, 1, nullptr, (const uint32_t*)doff, (rt_photon*)dout, nw,
                            st), "k_emit_photons");
    if (!out_is_device) HIPCHK(hipMemcpy(out, dout, (size_t)nw * sizeof(rt_photon), hipMemcpyDeviceToHost), "copy photons");
    return RT_OK;
}

int rt_photonmap_emit(rt_scene* s, const rt_emit_params* e, int64_t first, int64_t count, rt_photon* out, int64_t cap,
                      int64_t* nstored) {
    return emit_impl(s, e, first, count, out, 0, cap, nstored, nullptr);
}

int rt_photonmap_emit_dev(rt_scene* s, const rt_emit_params* e, int64_t first, int64_t count, rt_photon* d_out, int64_t cap,
                          int64_t* nstored, void* stream) {
    return emit_impl(s, e, first, count, d_out, 1, cap, nstored, (hipStream_t)stream);
}

}  // extern "C"
