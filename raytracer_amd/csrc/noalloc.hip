// noalloc.hip -- rt_photon_irradiance_noalloc_dev: the radiance estimate of rt_photon_irradiance_dev for callers that
// cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
//
// No counterpart in the reference (PhotonMap::CalcIrradiance mallocs two lists per lookup, photon.cpp:398-399).
// rt_photon_irradiance_dev keeps its per-call workspace -- the Morton keys and indices of the query sort and the sort's
// own storage, the list of queries left to the strict kernel and its counter -- in a grow-only hipMalloc'ed block of
// the scene.  Here that workspace is a fixed array in the code object; each GPU the library runs on holds its own copy
// at its own address until the process ends.  The kernels are the same and take the workspace as an argument, as
// they do there, so the results are the same bits.
#include "rt_scene.h"
#include <hip/hip_runtime.h>
#include <mutex>

#define RT_NOALLOC_SCRATCH_BYTES ((size_t)16 << 20)     /* 16 MiB per GPU */
#define RT_NOALLOC_MAX_DEVICES 64

// In host code this name is a stand-in for the symbol, not a device address: only noalloc_setup_device() mentions it,
// inside HIP_SYMBOL; every other use of the memory goes through the pointer that call returned.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {
std::mutex g_mu;
void* g_scratch_ptr[RT_NOALLOC_MAX_DEVICES] = {};       // per device: where that GPU's copy of `scratch` lives

// once per device, with that device current
int noalloc_setup_device(int dev, void** out) {
    if (dev < 0 || dev >= RT_NOALLOC_MAX_DEVICES) return rt_fail(RT_E_LIMIT, "device ordinal %d outside [0, %d)", dev, RT_NOALLOC_MAX_DEVICES);
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_scratch_ptr[dev]) {
        DeviceGuard g(dev);
        if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice(%d) failed", dev);
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return rt_hipfail(e, "hipGetSymbolAddress(scratch)");
        g_scratch_ptr[dev] = p;
    }
    if (out) *out = g_scratch_ptr[dev];
    return RT_OK;
}
}  // namespace

// called by rt_init for every gfx950 device, so that the entry point below finds its pointer ready
int rt_noalloc_setup(int dev) { return noalloc_setup_device(dev, nullptr); }

extern "C" int rt_photon_irradiance_noalloc_dev(rt_scene* s, int which, const float* d_pos3, const float* d_nrm3, int64_t q,
                                                int k, float maxDist, float* d_rgb, void* stream) {
    if (!s || which < 0 || which > 1) return rt_fail(RT_E_INVALID, "rt_photon_irradiance_noalloc_dev: bad scene or map selector");
    if (k < 1 || k > 287) return rt_fail(RT_E_LIMIT, "rt_photon_irradiance_noalloc_dev: k=%d outside [1,287]", k);
    const PhotonMapDev& m = s->pmap[which];
    if (!m.d_photons || !s->d_dir_tables) return rt_fail(RT_E_STATE, "rt_photon_irradiance_noalloc_dev: photon map %d not uploaded", which);
    if (q < 0 || (q > 0 && (!d_pos3 || !d_nrm3 || !d_rgb))) return rt_fail(RT_E_INVALID, "rt_photon_irradiance_noalloc_dev: bad argument");
    if (q > RT_NOALLOC_MAX_QUERIES)
        return rt_fail(RT_E_LIMIT, "rt_photon_irradiance_noalloc_dev: %lld queries in one call, the fixed workspace holds %lld "
                       "(split the call, or use rt_photon_irradiance_dev)", (long long)q, (long long)RT_NOALLOC_MAX_QUERIES);
    if (q == 0) return RT_OK;
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    if (s->knn_fast && k <= RT_KNN_FAST_KMAX) {
        void* ws = nullptr;
        int rc = noalloc_setup_device(s->device, &ws);         // a look-up after rt_init; the first call on a device otherwise
        if (rc) return rc;
        const size_t need = rtk_knn_fast_scratch_bytes(q);
        if (need > RT_NOALLOC_SCRATCH_BYTES)
            return rt_fail(RT_E_LIMIT, "rt_photon_irradiance_noalloc_dev: %lld queries need %zu B of workspace, the fixed one has %zu",
                           (long long)q, need, (size_t)RT_NOALLOC_SCRATCH_BYTES);
        // the workspace keeps the last call's contents; the one word that must start at zero (the length of the strict
        // kernel's list) is cleared on the stream by rtk_photon_knn_fast, everything else is written before it is read
        hipError_t e = rtk_photon_knn_fast(m.d_photons, m.n, m.half, m.box, s->d_knn_stats, d_pos3, d_nrm3, q, k, maxDist, 1, nullptr,
                                           nullptr, nullptr, nullptr, d_rgb, s->d_dir_tables, (uint8_t*)ws, st);
        if (e != hipSuccess) return rt_hipfail(e, "k_irr_main");
        return RT_OK;
    }
    // RT_KNN_STRICT / k > 112: the list code per query, candidate lists in LDS -- no workspace at all
    hipError_t e = rtk_photon_knn(m.d_photons, m.n, m.half, d_pos3, d_nrm3, q, k, maxDist, 1, nullptr, nullptr, nullptr, nullptr, d_rgb,
                                  s->d_dir_tables, st);
    if (e != hipSuccess) return rt_hipfail(e, "k_photon_knn");
    return RT_OK;
}
