// api.cpp -- the extern "C" surface of librt_hip.so (include/rt_hip.h): scene
// lifetime, traversal entry points, staging of caller-owned host buffers.
// There is no CPU path here: every entry point needs a gfx950 device.
#include "rt_scene.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <type_traits>

static thread_local std::string g_err;

int rt_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

int rt_hipfail(hipError_t e, const char* what) {
    return rt_fail(RT_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

DeviceGuard::DeviceGuard(int dev) : prev(-1), ok(false) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) return;
    ok = true; cur = dev;
}
DeviceGuard::~DeviceGuard() { if (ok && prev >= 0 && prev != cur) (void)hipSetDevice(prev); }

int rt_scene::dev_alloc(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return rt_fail(RT_E_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    allocs.push_back(*p);
    device_bytes += (int64_t)bytes;
    return RT_OK;
}

// grow-only scratch, reused across calls (one host thread per scene)
int rt_scene::scratch(int slot, size_t bytes, void** p) {
    if (slot < 0 || slot >= RT_NSCRATCH) return rt_fail(RT_E_INVALID, "bad scratch slot");
    if (scratch_cap[slot] < bytes) {
        if (scratch_ptr[slot]) {                       // nothing in flight on the caller's or the scene's second stream may still use it
            (void)hipDeviceSynchronize();
            (void)hipFree(scratch_ptr[slot]);
        }
        scratch_ptr[slot] = nullptr; scratch_cap[slot] = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&scratch_ptr[slot], want);
        if (e != hipSuccess) return rt_fail(RT_E_NOMEM, "hipMalloc(%zu) scratch: %s", want, hipGetErrorString(e));
        scratch_cap[slot] = want;
    }
    *p = scratch_ptr[slot];
    return RT_OK;
}

template <class V, class T>
static int upload(rt_scene* s, const V& v, const T** out) {
    static_assert(std::is_same<typename V::value_type, T>::value, "upload: element type");
    void* p = nullptr;
    int rc = s->dev_alloc(&p, v.size() * sizeof(T));
    if (rc) return rc;
    if (!v.empty()) {
        hipError_t e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) return rt_hipfail(e, "hipMemcpy(scene)");
    }
    *out = (const T*)p;
    return RT_OK;
}

// ---- algebra3 on the host (float ops of algebra3.cpp:280-282,295-302,345-347) ----
namespace {
struct H3 { float x, y, z; };
inline H3 h3(const float* p) { return H3{p[0], p[1], p[2]}; }
inline H3 hadd(H3 a, H3 b) { return H3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline H3 hsub(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 hmul(H3 a, float d) { return H3{d * a.x, d * a.y, d * a.z}; }
inline H3 hdiv(H3 a, float d) { const float di = (float)(1.0 / (double)d); return H3{a.x * di, a.y * di, a.z * di}; }
inline H3 hnorm(H3 a) { return hdiv(a, std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z)); }

// RectLight::GetSamplePoses, photon.h:90-111; PointLight / SimpleLight: the position itself (:60-64,:121-125)
void light_samples(const rt_light& l, std::vector<float>& out) {
    if (l.type != RT_LIGHT_RECT) { out.insert(out.end(), l.pos, l.pos + 3); return; }
    const H3 dir1 = hnorm(h3(l.dir1)), dir2 = hnorm(h3(l.dir2)), pos = h3(l.pos);
    const H3 startPos = hsub(hsub(pos, hdiv(hmul(dir1, l.size1), 2.0f)), hdiv(hmul(dir2, l.size2), 2.0f));
    for (int i = 0; i < l.n1; ++i) {
        const H3 nowPos = hadd(startPos, hdiv(hmul(hmul(dir1, l.size1), (float)i), (float)l.n1));
        for (int j = 0; j < l.n2; ++j) {
            const H3 t = hadd(nowPos, hdiv(hmul(hmul(dir2, l.size2), (float)j), (float)l.n2));
            out.push_back(t.x); out.push_back(t.y); out.push_back(t.z);
        }
    }
}

// the two triangles the RectL parser adds to lightMeshes, final.cpp:710-731, as v0 / edge1 / edge2
void light_quad(const rt_light& l, std::vector<float>& out) {
    const H3 pos = h3(l.pos);
    const H3 a = hmul(hmul(h3(l.dir1), l.size1), 0.5f), b = hmul(hmul(h3(l.dir2), l.size2), 0.5f);
    const H3 t[2][3] = {{hsub(hsub(pos, a), b), hsub(hadd(pos, a), b), hadd(hsub(pos, a), b)},
                        {hadd(hadd(pos, a), b), hsub(hadd(pos, a), b), hadd(hsub(pos, a), b)}};
    for (int k = 0; k < 2; ++k) {
        const H3 e1 = hsub(t[k][1], t[k][0]), e2 = hsub(t[k][2], t[k][0]);
        const float f[9] = {t[k][0].x, t[k][0].y, t[k][0].z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z};
        out.insert(out.end(), f, f + 9);
    }
}
}  // namespace

static bool is_gfx950(int dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_init(int ndev_wanted) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return rt_fail(RT_E_NODEVICE, "no HIP device visible (%s); librt_hip has no CPU path",
                       e == hipSuccess ? "count 0" : hipGetErrorString(e));
    int good = 0;
    for (int d = 0; d < n; ++d) good += is_gfx950(d) ? 1 : 0;
    if (good == 0) return rt_fail(RT_E_NODEVICE, "no gfx950 device among %d HIP devices", n);
    if (ndev_wanted > good) return rt_fail(RT_E_NODEVICE, "%d devices wanted, %d gfx950 visible", ndev_wanted, good);
    // the fixed workspace of rt_photon_irradiance_noalloc_dev: resolve its address now on the devices the caller says it
    // will use -- the first `ndev_wanted` gfx950 devices -- so that the entry point itself has nothing to set up
    // (noalloc.hip).  Resolving it creates a HIP context and loads the code object (with the 16 MiB array) on that device,
    // so it is NOT done for devices the caller did not ask for: a rank that calls rt_init(1) on an 8-GPU node touches one
    // GPU.  rt_scene_create does the same for its scene's device, and the entry point resolves lazily as a last resort.
    for (int d = 0, done = 0; d < n && done < ndev_wanted; ++d)
        if (is_gfx950(d)) { const int rc = rt_noalloc_setup(d); if (rc) return rc; ++done; }
    return good;
}

int rt_scene_create(const rt_scene_desc* d, rt_scene** out) {
    if (!d || !out) return rt_fail(RT_E_INVALID, "rt_scene_create: NULL argument");
    *out = nullptr;
    if (d->abi_version != RT_ABI_VERSION) return rt_fail(RT_E_INVALID, "abi_version %d != %d", d->abi_version, RT_ABI_VERSION);
    if (d->nprim < 0 || d->nmaterial <= 0 || d->nobject < 0 || d->nlight < 0)
        return rt_fail(RT_E_INVALID, "rt_scene_create: negative count or no material");
    if (d->nprim > 0 && (!d->prims || !d->prim_type || !d->prim_material))
        return rt_fail(RT_E_INVALID, "rt_scene_create: NULL primitive arrays");
    if (!d->materials || (d->nobject > 0 && (!d->obj_first || !d->obj_count)) || (d->nlight > 0 && !d->lights))
        return rt_fail(RT_E_INVALID, "rt_scene_create: NULL array");
    if (d->nprim >= (int64_t)1 << 30) return rt_fail(RT_E_LIMIT, "more than 2^30 primitives");

    int dev = d->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess)
        return rt_fail(RT_E_NODEVICE, "no HIP device; librt_hip has no CPU path");
    if (!is_gfx950(dev)) return rt_fail(RT_E_NODEVICE, "device %d is not a usable gfx950", dev);

    HostFlatScene flat;
    std::string err;
    int rc;
    // RT_HOST_LAYOUTS=1: all eight octant layouts from the host (the device otherwise derives seven of each hierarchy);
    // RT_CHECK_LAYOUTS=1: both, and the scene is created only if they agree byte for byte (tests/test_gpu_build.py)
    const bool check_layouts = getenv("RT_CHECK_LAYOUTS") && atoi(getenv("RT_CHECK_LAYOUTS")) != 0;
    const bool host_layouts = getenv("RT_HOST_LAYOUTS") && atoi(getenv("RT_HOST_LAYOUTS")) != 0;
    flat.host_layouts = host_layouts || check_layouts;
    flat.device = dev;
    const bool timing = getenv("RT_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tb0 = now();
    double tb1 = tb0;
    try {
        rc = rt_build_flat_scene(d, &flat, &err);
        tb1 = now();
        if (!rc) rc = rt_build_bvh(d, &flat, &err);
    } catch (const std::bad_alloc&) {
        return rt_fail(RT_E_NOMEM, "host allocation failed while building the KD-trees");
    }
    if (rc) return rt_fail(rc, "%s", err.c_str());
    const double tb2 = now();

    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return rt_fail(RT_E_NOMEM, "host allocation failed");
    s->device = dev;
    DeviceGuard g(dev);
    if (!g.ok) { delete s; return rt_fail(RT_E_NODEVICE, "hipSetDevice(%d) failed", dev); }
    // rt_photon_irradiance_noalloc_dev's fixed workspace on this scene's device (a look-up if rt_init already did it): a
    // caller that captures a HIP graph creates its scene before the capture, so the entry point never resolves under capture
    { const int rcn = rt_noalloc_setup(dev); if (rcn) { delete s; return rcn; } }

    std::vector<DevMaterial> mats((size_t)d->nmaterial);
    for (int i = 0; i < d->nmaterial; ++i) {
        const rt_material& m = d->materials[i];
        DevMaterial& o = mats[(size_t)i];
        memcpy(o.surf, m.surf, 12);
        o.Ka = m.Ka; o.Kd = m.Kd; o.Ks = m.Ks; o.specularity = m.specularity;
        o.reflect = m.reflect; o.refract = m.refract; o.Nr = m.Nr; o.pad[0] = o.pad[1] = 0;
    }
    std::vector<DevLight> lights((size_t)d->nlight);
    for (int i = 0; i < d->nlight; ++i) {
        const rt_light& l = d->lights[i];
        DevLight& o = lights[(size_t)i];
        memcpy(o.pos, l.pos, 12); o.intensity = l.intensity; memcpy(o.color, l.color, 12); o.type = l.type;
        memcpy(o.dir1, l.dir1, 12); o.size1 = l.size1; memcpy(o.dir2, l.dir2, 12); o.size2 = l.size2;
    }
    s->host_lights.assign(d->lights, d->lights + d->nlight);
    std::vector<float> lsamples, ltris;
    std::vector<int32_t> lfirst;
    for (int i = 0; i < d->nlight; ++i) {
        const rt_light& l = d->lights[i];
        if (l.type == RT_LIGHT_RECT && (l.n1 < 0 || l.n2 < 0 || (int64_t)l.n1 * l.n2 > 1 << 20)) {
            delete s; return rt_fail(RT_E_LIMIT, "light %d: %d x %d samples", i, l.n1, l.n2);
        }
        lfirst.push_back((int32_t)(lsamples.size() / 3));
        light_samples(l, lsamples);
        if (l.type == RT_LIGHT_RECT) light_quad(l, ltris);
    }
    lfirst.push_back((int32_t)(lsamples.size() / 3));
    std::vector<int32_t> pmat(d->prim_material, d->prim_material + d->nprim);
    std::vector<uint8_t> ptype(d->prim_type, d->prim_type + d->nprim);

    DevSceneView& v = s->view;
    memset(&v, 0, sizeof v);
    rc = RT_OK;
    // the hierarchies: the reference trees in the reference's order (layout 7) and the BVH in layout 0 go up, the other
    // seven layouts of each are derived there (kernels_build.hip) unless the host was asked for them
    const int32_t nkd = (int32_t)flat.nodes[7].size(), nbvh = (int32_t)flat.bvh_nodes[0].size();
    rc = upload(s, flat.nodes[7], &v.nodes_oct[7]);
    v.nodes = v.nodes_oct[7];
    if (!rc && !host_layouts) {
        DevNode* dst[7] = {};
        for (int o = 0; o < 7 && !rc; ++o) { void* p = nullptr; rc = s->dev_alloc(&p, (size_t)nkd * sizeof(DevNode)); dst[o] = (DevNode*)p; v.nodes_oct[o] = dst[o]; }
        flat.kd_roots.push_back(nkd);
        const int32_t* d_roots = nullptr;
        if (!rc) rc = upload(s, flat.kd_roots, &d_roots);
        if (!rc) { hipError_t e = rtk_kd_layouts(v.nodes_oct[7], nkd, d_roots, (int32_t)flat.kd_roots.size() - 1, dst, nullptr); if (e != hipSuccess) rc = rt_hipfail(e, "k_kd_layouts"); }
    } else {
        for (int o = 0; o < 7 && !rc; ++o) rc = upload(s, flat.nodes[o], &v.nodes_oct[o]);
    }
    if (!rc) rc = upload(s, flat.leaf_tris, &v.leaf_tris);
    if (!rc) rc = upload(s, flat.leaf_groups, &v.leaf_groups);
    if (!rc) rc = upload(s, flat.loose, &v.loose);
    if (!rc) rc = upload(s, flat.bvh_nodes[0], &v.bvh_oct[0]);
    if (!rc && !host_layouts) {
        DevBvhNode* dst[8] = {};
        for (int o = 1; o < 8 && !rc; ++o) { void* p = nullptr; rc = s->dev_alloc(&p, (size_t)nbvh * sizeof(DevBvhNode)); dst[o] = (DevBvhNode*)p; v.bvh_oct[o] = dst[o]; }
        if (!rc) { hipError_t e = rtk_bvh_layouts(v.bvh_oct[0], nbvh, dst, nullptr); if (e != hipSuccess) rc = rt_hipfail(e, "k_bvh_layouts"); }
    } else {
        for (int o = 1; o < 8 && !rc; ++o) rc = upload(s, flat.bvh_nodes[o], &v.bvh_oct[o]);
    }
    if (!rc && !host_layouts) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) rc = rt_hipfail(e, "octant layouts");
    }
    if (!rc && check_layouts && !host_layouts) {                  // the device's layouts against the host's, byte for byte
        std::vector<unsigned char> got;
        for (int o = 0; o < 8 && !rc; ++o) {
            got.resize((size_t)nkd * sizeof(DevNode));
            if (nkd && (hipMemcpy(got.data(), v.nodes_oct[o], got.size(), hipMemcpyDeviceToHost) != hipSuccess || memcmp(got.data(), flat.nodes[o].data(), got.size()) != 0))
                rc = rt_fail(RT_E_INVALID, "RT_CHECK_LAYOUTS: layout %d of the reference trees differs from the host's", o);
            got.resize((size_t)nbvh * sizeof(DevBvhNode));
            if (!rc && nbvh && (hipMemcpy(got.data(), v.bvh_oct[o], got.size(), hipMemcpyDeviceToHost) != hipSuccess || memcmp(got.data(), flat.bvh_nodes[o].data(), got.size()) != 0))
                rc = rt_fail(RT_E_INVALID, "RT_CHECK_LAYOUTS: layout %d of the BVH differs from the host's", o);
        }
    }
    v.nbvh = nbvh;
    if (!rc && !flat.bvh_c32.empty()) rc = upload(s, flat.bvh_c32, &v.bvh_c32);
    memcpy(v.bvh_root_centre, flat.bvh_root_centre, sizeof v.bvh_root_centre);
    v.bvh_root_rho = v.nbvh > 0 ? flat.bvh_nodes[0][0].rho : 0.f;
    if (v.nbvh > 0)                                    // layout 0 is unmirrored: b = {lo.x, hi.x, lo.y, hi.y, lo.z, hi.z}
        for (int k = 0; k < 3; ++k) { s->bvh_lo[k] = flat.bvh_nodes[0][0].b[2 * k]; s->bvh_hi[k] = flat.bvh_nodes[0][0].b[2 * k + 1]; }
    // per-lane walks all use layout 0: with random directions the eight layouts are eight times the cache footprint,
    // which costs more than the near-first order saves (1M-triangle mesh, passes of depth >= 2: 18 % faster than with
    // the ray's own octant; RT_LANE_LAYOUT_MASK=7 restores that)
    v.lane_layout_mask = 0;
    if (const char* e = getenv("RT_LANE_LAYOUT_MASK")) v.lane_layout_mask = atoi(e) & 7;
    if (v.lane_layout_mask != 0) v.bvh_c32 = nullptr;    // the compact nodes are layout 0's
    if (!rc) {
        void* q = nullptr;
        if (hipMalloc(&q, 64 * RT_RAY_COUNTERS) != hipSuccess) rc = rt_fail(RT_E_NOMEM, "hipMalloc failed");
        else { s->allocs.push_back(q); v.ray_counter = (unsigned long long*)q; }
    }
    if (!rc) rc = upload(s, flat.bvh_tris, &v.bvh_tris);
    if (!rc) rc = upload(s, flat.bvh_tri_ref, &v.bvh_tri_ref);
    if (!rc) rc = upload(s, flat.bvh_ref_leaf, &v.bvh_ref_leaf);
    if (!rc) rc = upload(s, flat.ref_parent, &v.ref_parent);
    if (!rc) rc = upload(s, flat.ref_chain_box, &v.ref_chain_box);
    // the single-step per-lane walk (trace_device.h: bvh_lane_step): its reachability records are made here, on the device
    v.lane_step = 0; v.lane_refill_batch = 8; v.lane_acc_batch = 4; v.lane_chunk = 128;
    if (!rc && v.bvh_c32 && !flat.bvh_tris.empty()) {
        void* q = nullptr;
        const int64_t nt = (int64_t)flat.bvh_tris.size();
        rc = s->dev_alloc(&q, (size_t)nt * sizeof(DevTriReach));
        if (!rc) {
            v.bvh_tri_reach = (const DevTriReach*)q;
            hipError_t e = rtk_tri_reach(v.bvh_tri_ref, v.bvh_ref_leaf, v.ref_chain_box, nt, (DevTriReach*)q, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) rc = rt_hipfail(e, "k_tri_reach");
            v.lane_step = 1;
        }
    }
    // ... and the tables for walks that start at the origin's leaf (trace_device.h: bvh_lane_start_from): one pre-order pass
    // over the compact nodes
    {
        const char* e = getenv("RT_LANE_ORIGIN");
        if (!rc && v.lane_step && (!e || atoi(e) != 0) && nbvh > 0) {
            const auto& nd = flat.bvh_c32;
            std::vector<int32_t> up((size_t)nbvh + 4, nbvh), parent((size_t)nbvh, -1), prim_start(ptype.size() * 4, -1);
            for (int32_t i = 0; i < nbvh; ++i) {
                const DevBvhNode32& n = nd[(size_t)i];
                const int32_t par = parent[(size_t)i];
                if (n.link >> 31) {
                    const uint32_t t0 = n.link & 0x7fffffffu, cnt = (n.sq >> 1) & 7u;
                    for (uint32_t t = t0; t < t0 + cnt; ++t) {
                        const int32_t p = flat.bvh_tris[t].prim;
                        if (p < 0 || (size_t)p >= ptype.size()) continue;
                        int32_t sib = nbvh;                                    // the leaf's own sibling: the walk's second range
                        if (par >= 0) {
                            const int32_t pl = par + 1;
                            sib = (i == pl) ? pl + 1 : pl;                     // (a leaf is one node: its right sibling follows it)
                        }
                        int32_t* q = &prim_start[4 * (size_t)p];
                        q[0] = i; q[1] = i + 1; q[2] = sib; q[3] = 0;
                    }
                } else {
                    const int32_t left = i + 1;
                    const int32_t right = (nd[(size_t)left].link >> 31) ? left + 1 : (int32_t)nd[(size_t)left].link;
                    parent[(size_t)left] = parent[(size_t)right] = i;
                    // after a child's subtree a climbing walk goes on at the sibling of i (i = the child's parent)
                    int32_t sib = nbvh;
                    if (par >= 0) {
                        const int32_t pl = par + 1;
                        sib = (i == pl) ? ((nd[(size_t)pl].link >> 31) ? pl + 1 : (int32_t)nd[(size_t)pl].link) : pl;
                    }
                    up[(size_t)left] = up[(size_t)right] = sib;
                }
            }
            rc = upload(s, up, &v.bvh_up);
            if (!rc) rc = upload(s, prim_start, &v.prim_start);
        }
    }
    if (const char* e = getenv("RT_LANE_STEP")) v.lane_step = (atoi(e) != 0 && v.bvh_tri_reach) ? 1 : 0;
    if (const char* e = getenv("RT_LANE_REFILL_BATCH")) { const int b = atoi(e); v.lane_refill_batch = b < 1 ? 1 : (b > 64 ? 64 : b); }
    if (const char* e = getenv("RT_LANE_CHUNK")) { const int b = atoi(e); v.lane_chunk = b < 16 ? 16 : (b > 4096 ? 4096 : b); }
    if (const char* e = getenv("RT_LANE_ACC_BATCH")) { const int b = atoi(e); v.lane_acc_batch = b < 1 ? 1 : (b > 64 ? 64 : b); }
    if (!rc) rc = upload(s, flat.prim_normal, &v.prim_normal);
    if (!rc) rc = upload(s, flat.prim_geom, &v.prim_geom);
    if (!rc) rc = upload(s, pmat, &v.prim_material);
    if (!rc) rc = upload(s, ptype, &v.prim_type);
    if (!rc) rc = upload(s, mats, &v.materials);
    if (!rc) rc = upload(s, lights, &v.lights);
    if (!rc) rc = upload(s, lsamples, &v.light_samples);
    if (!rc) rc = upload(s, lfirst, &v.light_sample_first);
    if (!rc) rc = upload(s, ltris, &v.light_tris);
    if (rc) { rt_scene_destroy(s); return rc; }
    if (timing) fprintf(stderr, "rt_scene_create: reference trees %.2f s, BVH %.2f s, uploads %.2f s\n", tb1 - tb0, tb2 - tb1, now() - tb2);
    v.nlight_samples = (int32_t)(lsamples.size() / 3);
    v.nlight_tris = (int32_t)(ltris.size() / 9);
    v.nnodes = (int32_t)flat.nodes[7].size();
    v.nloose = (int32_t)flat.loose.size();
    v.nlights = d->nlight;
    v.tol_f = rt_float_above(d->tolerance_depth);
    v.culling = (d->flags & RT_SCENE_CULLING) ? 1 : 0;
    v.shadow_bias = d->shadow_bias; v.refract_bias = d->refract_bias;
    s->eps_f = rt_float_above(0.000001);          // `epsilon`, hw4.cpp:18
    s->nprim = d->nprim;
    if (const char* e = getenv("RT_TRAVERSAL"))
        s->fast = (strcmp(e, "strict") == 0) ? 0 : (strcmp(e, "fast") == 0) ? 1 : (strcmp(e, "lane") == 0) ? 2 : (strcmp(e, "bvh") == 0) ? 3 : s->fast;
    if (const char* e = getenv("RT_FUSED_PRIMARY")) s->fused_primary = atoi(e) != 0;
    if (const char* e = getenv("RT_LANE_DEPTH")) s->lane_depth = atoi(e) < 0 ? 0 : atoi(e);
    if (const char* e = getenv("RT_PATH_DEPTH")) s->path_depth = atoi(e) < 0 ? 0 : atoi(e);
    if (const char* e = getenv("RT_RAY_SORT")) s->ray_sort = atoi(e) != 0;
    if (const char* e = getenv("RT_RENDER_OVERLAP")) s->overlap_shadow = atoi(e) != 0;
    if (const char* e = getenv("RT_KNN")) s->knn_fast = (strcmp(e, "strict") == 0) ? 0 : 1;

    rt_scene_info& inf = s->info;
    memset(&inf, 0, sizeof inf);
    inf.kd_nodes_built = flat.kd_nodes_built; inf.kd_max_depth = flat.kd_max_depth;
    inf.flat_nodes = (int64_t)flat.nodes[7].size(); inf.leaf_refs = (int64_t)flat.leaf_tris.size();
    inf.ntri = flat.ntri; inf.nsphere = flat.nsphere; inf.nloose = (int64_t)flat.loose.size();
    inf.device_bytes = s->device_bytes;
    *out = s;
    return RT_OK;
}

int rt_scene_destroy(rt_scene* s) {
    if (!s) return RT_OK;
    DeviceGuard g(s->device);
    for (void* p : s->allocs) (void)hipFree(p);
    for (int i = 0; i < RT_NSCRATCH; ++i) if (s->scratch_ptr[i]) (void)hipFree(s->scratch_ptr[i]);
    for (int w = 0; w < 2; ++w) if (s->pmap[w].d_photons) (void)hipFree(s->pmap[w].d_photons);
    if (s->d_tiles_cache) (void)hipFree(s->d_tiles_cache);
    if (s->aux_stream) (void)hipStreamDestroy(s->aux_stream);
    delete s;
    return RT_OK;
}

int rt_scene_knn_stats(rt_scene* s, int64_t out16[16]) {
    if (!s || !out16) return rt_fail(RT_E_INVALID, "rt_scene_knn_stats: NULL argument");
    memset(out16, 0, 16 * sizeof(int64_t));
    if (!s->d_knn_stats_buf) return RT_OK;
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out16, s->d_knn_stats_buf, 16 * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess)
        return rt_fail(RT_E_NODEVICE, "rt_scene_knn_stats: copy failed");
    return RT_OK;
}

int rt_scene_set_option(rt_scene* s, int option, int value) {
    if (!s) return rt_fail(RT_E_INVALID, "rt_scene_set_option: NULL scene");
    if (option == RT_OPT_TRAVERSAL && (value == RT_TRAVERSAL_STRICT || value == RT_TRAVERSAL_FAST || value == RT_TRAVERSAL_LANE || value == RT_TRAVERSAL_BVH)) { s->fast = value; return RT_OK; }
    if (option == RT_OPT_LANE_DEPTH && value >= 0) { s->lane_depth = value; return RT_OK; }
    if (option == RT_OPT_PATH_DEPTH && value >= 0) { s->path_depth = value; return RT_OK; }
    if (option == RT_OPT_FUSED_PRIMARY && (value == 0 || value == 1)) { s->fused_primary = value; return RT_OK; }
    if (option == RT_OPT_RAY_SORT && (value == 0 || value == 1)) { s->ray_sort = value; return RT_OK; }
    if (option == RT_OPT_RENDER_OVERLAP && (value == 0 || value == 1)) { s->overlap_shadow = value; return RT_OK; }
    if (option == RT_OPT_KNN_STATS && (value == 0 || value == 1)) {
        DeviceGuard g(s->device);
        if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
        if (value && !s->d_knn_stats_buf) {
            if (hipMalloc((void**)&s->d_knn_stats_buf, 16 * sizeof(unsigned long long)) != hipSuccess) return rt_fail(RT_E_NOMEM, "hipMalloc failed");
            s->allocs.push_back(s->d_knn_stats_buf);
        }
        if (value && (hipDeviceSynchronize() != hipSuccess || hipMemset(s->d_knn_stats_buf, 0, 16 * sizeof(unsigned long long)) != hipSuccess))
            return rt_fail(RT_E_NODEVICE, "hipMemset failed");
        s->d_knn_stats = value ? s->d_knn_stats_buf : nullptr;               // off: the buffer keeps the last counts
        return RT_OK;
    }
    if (option == RT_OPT_KNN && (value == RT_KNN_STRICT || value == RT_KNN_FAST)) { s->knn_fast = value; return RT_OK; }
    return rt_fail(RT_E_INVALID, "rt_scene_set_option: unknown option %d / value %d", option, value);
}

int rt_scene_info_get(const rt_scene* s, rt_scene_info* info) {
    if (!s || !info) return rt_fail(RT_E_INVALID, "rt_scene_info_get: NULL argument");
    *info = s->info;
    info->device_bytes = s->device_bytes;
    return RT_OK;
}

// ---- traversal ------------------------------------------------------------

int rt_trace_closest_dev(rt_scene* s, const rt_ray* d_rays, int64_t n, rt_hit* d_out, void* stream) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_out))) return rt_fail(RT_E_INVALID, "rt_trace_closest_dev: bad argument");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    hipError_t e = rtk_trace_closest(s->view, d_rays, nullptr, n, s->eps_f, s->fast, d_out, (hipStream_t)stream);
    if (e != hipSuccess) return rt_hipfail(e, "k_trace_closest");
    return RT_OK;
}

int rt_trace_closest(rt_scene* s, const rt_ray* rays, int64_t n, rt_hit* out) {
    if (!s || n < 0 || (n > 0 && (!rays || !out))) return rt_fail(RT_E_INVALID, "rt_trace_closest: bad argument");
    if (n == 0) return RT_OK;
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    void *dr, *dh; int rc;
    if ((rc = s->scratch(0, (size_t)n * sizeof(rt_ray), &dr))) return rc;
    if ((rc = s->scratch(1, (size_t)n * sizeof(rt_hit), &dh))) return rc;
    hipError_t e = hipMemcpy(dr, rays, (size_t)n * sizeof(rt_ray), hipMemcpyHostToDevice);
    if (e != hipSuccess) return rt_hipfail(e, "hipMemcpy(rays)");
    e = rtk_trace_closest(s->view, (const rt_ray*)dr, nullptr, n, s->eps_f, s->fast, (rt_hit*)dh, nullptr);
    if (e != hipSuccess) return rt_hipfail(e, "k_trace_closest");
    e = hipMemcpy(out, dh, (size_t)n * sizeof(rt_hit), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return rt_hipfail(e, "hipMemcpy(hits)");
    return RT_OK;
}

int rt_trace_any_dev(rt_scene* s, const rt_ray* d_rays, const float* d_tmax, int64_t n, uint8_t* d_occ, void* stream) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_occ))) return rt_fail(RT_E_INVALID, "rt_trace_any_dev: bad argument");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    hipError_t e = rtk_trace_any(s->view, d_rays, nullptr, d_tmax, n, s->eps_f, s->fast, d_occ, (hipStream_t)stream);
    if (e != hipSuccess) return rt_hipfail(e, "k_trace_any");
    return RT_OK;
}

int rt_trace_work_dev(rt_scene* s, const rt_ray* d_rays, const float* d_tmax, int64_t n, int any, int64_t counts[4],
                      void* stream) {
    if (!s || n < 0 || !counts || (n > 0 && !d_rays)) return rt_fail(RT_E_INVALID, "rt_trace_work_dev: bad argument");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    void* dc; int rc;
    if ((rc = s->scratch(2, 4 * sizeof(unsigned long long), &dc))) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(dc, 0, 4 * sizeof(unsigned long long), st);
    if (e != hipSuccess) return rt_hipfail(e, "hipMemsetAsync(counters)");
    e = rtk_trace_work(s->view, d_rays, nullptr, d_tmax, n, s->eps_f, s->fast, any, (unsigned long long*)dc, st);
    if (e != hipSuccess) return rt_hipfail(e, "k_trace_work");
    unsigned long long h[4];
    e = hipMemcpyAsync(h, dc, sizeof h, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return rt_hipfail(e, "read counters");
    for (int i = 0; i < 4; ++i) counts[i] = (int64_t)h[i];
    return RT_OK;
}

int rt_trace_any(rt_scene* s, const rt_ray* rays, const float* tmax, int64_t n, uint8_t* occluded) {
    if (!s || n < 0 || (n > 0 && (!rays || !occluded))) return rt_fail(RT_E_INVALID, "rt_trace_any: bad argument");
    if (n == 0) return RT_OK;
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    void *dr, *dt = nullptr, *dout; int rc;
    if ((rc = s->scratch(0, (size_t)n * sizeof(rt_ray), &dr))) return rc;
    if ((rc = s->scratch(1, (size_t)n, &dout))) return rc;
    hipError_t e = hipMemcpy(dr, rays, (size_t)n * sizeof(rt_ray), hipMemcpyHostToDevice);
    if (e != hipSuccess) return rt_hipfail(e, "hipMemcpy(rays)");
    if (tmax) {
        if ((rc = s->scratch(2, (size_t)n * sizeof(float), &dt))) return rc;
        e = hipMemcpy(dt, tmax, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) return rt_hipfail(e, "hipMemcpy(tmax)");
    }
    e = rtk_trace_any(s->view, (const rt_ray*)dr, nullptr, (const float*)dt, n, s->eps_f, s->fast, (uint8_t*)dout, nullptr);
    if (e != hipSuccess) return rt_hipfail(e, "k_trace_any");
    e = hipMemcpy(occluded, dout, (size_t)n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return rt_hipfail(e, "hipMemcpy(occluded)");
    return RT_OK;
}

int rt_generate_primary_dev(rt_scene* s, const rt_camera* cam, int64_t first, int64_t n, float ox, float oy,
                            rt_ray* d_rays, void* stream) {
    if (!s || !cam || n < 0 || first < 0 || (n > 0 && !d_rays)) return rt_fail(RT_E_INVALID, "rt_generate_primary_dev: bad argument");
    if (cam->resx <= 0 || cam->resy <= 0 || first + n > (int64_t)cam->resx * cam->resy)
        return rt_fail(RT_E_INVALID, "rt_generate_primary_dev: pixel range outside the frame");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    hipError_t e = rtk_generate_linear(*cam, first, n, ox, oy, d_rays, (hipStream_t)stream);
    if (e != hipSuccess) return rt_hipfail(e, "k_generate_linear");
    return RT_OK;
}

}  // extern "C"

// ---- host-only flatten (checks of the KD builder without a GPU) ------------
struct rt_host_flat { HostFlatScene flat; };

extern "C" int rt_host_flatten(const rt_scene_desc* d, rt_host_flat** out, rt_scene_info* info) {
    if (!d || !out) return rt_fail(RT_E_INVALID, "rt_host_flatten: NULL argument");
    *out = nullptr;
    if (d->nprim < 0 || d->nmaterial <= 0 || !d->materials || (d->nprim > 0 && (!d->prims || !d->prim_type || !d->prim_material)))
        return rt_fail(RT_E_INVALID, "rt_host_flatten: bad description");
    rt_host_flat* f = new (std::nothrow) rt_host_flat();
    if (!f) return rt_fail(RT_E_NOMEM, "host allocation failed");
    std::string err;
    int rc = rt_build_flat_scene(d, &f->flat, &err);
    if (rc) { delete f; return rt_fail(rc, "%s", err.c_str()); }
    if (info) {
        memset(info, 0, sizeof *info);
        info->kd_nodes_built = f->flat.kd_nodes_built; info->kd_max_depth = f->flat.kd_max_depth;
        info->flat_nodes = (int64_t)f->flat.nodes[7].size(); info->leaf_refs = (int64_t)f->flat.leaf_tris.size();
        info->ntri = f->flat.ntri; info->nsphere = f->flat.nsphere; info->nloose = (int64_t)f->flat.loose.size();
    }
    *out = f;
    return RT_OK;
}

extern "C" int64_t rt_host_flat_get(const rt_host_flat* f, int which, void* dst, int64_t cap) {
    if (!f) return rt_fail(RT_E_INVALID, "rt_host_flat_get: NULL");
    const HostFlatScene& h = f->flat;
    const int64_t nn = (int64_t)h.nodes[7].size(), nl = (int64_t)h.leaf_tris.size(), no = (int64_t)h.loose.size();
    const int64_t rows = which <= 4 ? nn : (which == 6 ? no : nl);
    if (which < 0 || which > 7) return rt_fail(RT_E_INVALID, "rt_host_flat_get: bad selector");
    if (!dst) return rows;
    if (cap < rows) return rt_fail(RT_E_INVALID, "rt_host_flat_get: buffer too small");
    for (int64_t i = 0; i < rows; ++i) {
        switch (which) {
            case 0: memcpy((float*)dst + 6 * i, h.nodes[7][(size_t)i].lo, 24); break;
            case 1: ((int32_t*)dst)[i] = h.nodes[7][(size_t)i].skip; break;
            case 2: ((int32_t*)dst)[i] = h.nodes[7][(size_t)i].meta; break;
            case 3: ((int32_t*)dst)[i] = h.nodes[7][(size_t)i].leaf_first; break;
            case 4: ((int32_t*)dst)[i] = h.nodes[7][(size_t)i].leaf_count; break;
            case 5: ((int32_t*)dst)[i] = h.rank_prim[(size_t)i]; break;                                 // in visit-rank order
            case 6: ((int32_t*)dst)[i] = h.loose[(size_t)i].prim; break;
            case 7: memcpy((float*)dst + 9 * (size_t)h.leaf_tris[(size_t)i].rank, h.leaf_tris[(size_t)i].v0, 36); break;
        }
    }
    return rows;
}

extern "C" void rt_host_flat_free(rt_host_flat* f) { delete f; }

// tools/time_host_build.py, tests/test_host_build.py: the host half of rt_scene_create (both hierarchies) with no device, for
// timing the builders and for checking that their output does not depend on the number of threads.  *digest (may be
// NULL): a hash of every array of the flattened scene.
extern "C" int rt_debug_host_build(const rt_scene_desc* d, uint64_t* digest) {
    if (!d) return rt_fail(RT_E_INVALID, "rt_debug_host_build: NULL argument");
    HostFlatScene flat;
    std::string err;
    int rc = rt_build_flat_scene(d, &flat, &err);
    if (!rc) rc = rt_build_bvh(d, &flat, &err);
    if (rc) return rt_fail(rc, "%s", err.c_str());
    if (digest) {
        uint64_t h = 1469598103934665603ull;
        auto add = [&h](const void* p, size_t bytes) {
            const unsigned char* c = (const unsigned char*)p;
            size_t i = 0;
            for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, c + i, 8); h = (h ^ w) * 1099511628211ull; }
            for (; i < bytes; ++i) h = (h ^ c[i]) * 1099511628211ull;
            h = (h ^ bytes) * 1099511628211ull;
        };
        auto vec = [&add](const auto& v) { add(v.data(), v.size() * sizeof(v[0])); };
        for (int o = 0; o < 8; ++o) { vec(flat.nodes[o]); vec(flat.bvh_nodes[o]); }
        vec(flat.leaf_tris); vec(flat.leaf_groups); vec(flat.rank_prim); vec(flat.loose); vec(flat.bvh_c32); vec(flat.bvh_tris);
        vec(flat.bvh_tri_ref); vec(flat.bvh_ref_leaf); vec(flat.ref_parent); vec(flat.ref_chain_box); vec(flat.prim_normal); vec(flat.prim_geom);
        *digest = h;
    }
    return RT_OK;
}
