// render_photon.cpp -- RT_RENDER_PHOTON frame driver: replaces the pixel loop of PhotonMappingImageSynthesis
// (Final/photon/final.cpp:1146-1202) and the recursion of SimpleMeshRayTrace(pmap, cpmap, ...) (:785-983) /
// TraceIrradiance (photon.cpp:714-820) with the queue pipeline described in kernels_pm.hip.
//
// The frame is processed in chunks of micro-tiles so that the queues (50 gather rays and one occlusion byte per
// light sample for every eye-path hit) stay within a fixed HBM budget whatever the resolution.
#include "rt_scene.h"
#include <algorithm>
#include <cstring>
#include <vector>

void rank_microtiles(const rt_camera& cam, const rt_params& p, std::vector<int32_t>& out);   // render.cpp

namespace {

#define HIPCHK(expr, what) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return rt_hipfail(e__, what); } while (0)
#define RTCHK(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

// scratch slots of the photon renderer (the Whitted renderer's slots 3.. are reused: one frame at a time per scene)
enum {
    P_TILES = 3, P_PIXMAP, P_WHITE, P_HITANY, P_ACCUM, P_CTR,
    P_QA_RAYS, P_QA_SID, P_QA_W, P_QA_CODE, P_QB_RAYS, P_QB_SID, P_QB_W, P_QB_CODE,
    P_HITS, P_FLAGS, P_NRM, P_NRML, P_QPOS, P_CAUSTIC, P_OCC, P_CONTRIB, P_BSUMS,
    P_G_SLOTS,  // the gather queues and their per-ray arrays, carved from one slot (Arena)
    P_KNN_FLAG,   // queries the fast kNN kernel hands to the strict one
    P_KNN_FLAG_C  // ... of the caustic map's calls, whose strict redo may still run while the global map's calls use the other
};
static_assert(P_KNN_FLAG_C + 1 <= RT_NSCRATCH, "not enough scratch slots");

struct Arena {   // bump allocator over one scratch slot
    char* base = nullptr; size_t cap = 0, off = 0;
    template <class T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return off <= cap ? p : nullptr;
    }
};

int knn_launch(rt_scene* s, int which, const float* pos, const float* nrm, int64_t q, int k, float radius, float* rgb, hipStream_t st, hipStream_t redo_st);

// the radiance estimate, bracketed by events when the caller wants rt_stats.knn_ms
// redo_st: the strict redo of the flagged queries goes to that stream (the caller joins it before reading rgb), or NULL
int knn(rt_scene* s, int which, const float* pos, const float* nrm, int64_t q, int k, float radius, float* rgb, hipStream_t st,
        EventBag* bag, std::vector<hipEvent_t>* kev, hipStream_t redo_st = nullptr) {
    hipError_t err;
    if (bag) { hipEvent_t e = bag->mark(st, &err); if (!e) return rt_hipfail(err, "event"); kev->push_back(e); }
    RTCHK(knn_launch(s, which, pos, nrm, q, k, radius, rgb, st, redo_st));
    if (bag) { hipEvent_t e = bag->mark(st, &err); if (!e) return rt_hipfail(err, "event"); kev->push_back(e); }
    return RT_OK;
}

int knn_launch(rt_scene* s, int which, const float* pos, const float* nrm, int64_t q, int k, float radius, float* rgb, hipStream_t st, hipStream_t redo_st) {
    const PhotonMapDev& m = s->pmap[which];
    if (!m.d_photons || m.n < 1) { HIPCHK(hipMemsetAsync(rgb, 0, (size_t)q * 12, st), "memset irradiance"); return RT_OK; }
    if (s->knn_fast && k <= RT_KNN_FAST_KMAX) {
        void* fl; RTCHK(s->scratch(which == RT_PMAP_CAUSTIC ? P_KNN_FLAG_C : P_KNN_FLAG, rtk_knn_fast_scratch_bytes(q), &fl));
        HIPCHK(rtk_photon_knn_fast(m.d_photons, m.n, m.half, m.box, s->d_knn_stats, pos, nrm, q, k, radius, 1, nullptr, nullptr, nullptr, nullptr, rgb,
                                   s->d_dir_tables, (uint8_t*)fl, st, redo_st), "k_photon_knn_coop");
        return RT_OK;
    }
    HIPCHK(rtk_photon_knn(m.d_photons, m.n, m.half, pos, nrm, q, k, radius, 1, nullptr, nullptr, nullptr, nullptr, rgb,
                          s->d_dir_tables, st), "k_photon_knn");
    return RT_OK;
}

}  // namespace

int render_photon_impl(rt_scene* s, const rt_camera* cam, const rt_params* p, uint8_t* d_rgb8, int32_t* d_hit_ids,
                       rt_stats* stats, hipStream_t st) {
    const int ntap = p->aa_taps > 0 ? std::min(p->aa_taps, 6) : 1;
    const int use_filter = p->aa_taps > 0 ? 1 : 0;
    const int G = std::max(0, p->gather_samples);
    const int S = s->view.nlight_samples;
    const int K = p->knn_k;
    if (K < 1 || K > 287) return rt_fail(RT_E_LIMIT, "rt_render: knn_k=%d outside [1,287]", K);
    if (p->irradiance_max_depth < 0 || p->irradiance_max_depth > 16) return rt_fail(RT_E_INVALID, "rt_render: irradiance_max_depth out of range");

    std::vector<int32_t> tiles;
    rank_microtiles(*cam, *p, tiles);
    const int64_t ntiles = (int64_t)tiles.size();
    // chunk: about 2^21 samples -- a 1080p frame whole: 105 M gather rays, and a gather arena of 84 GB at the reference's settings
    // (50 gather rays per sample, room for two refractive doublings, 202 B per entry) beside 0.8 GB of occlusion bytes and ~6 GB
    // of radiance-estimate workspace; the memory is taken for the frames that need it, a small frame is one small chunk.  The
    // radiance estimate is why: a call sorts ITS queries along the Morton curve, so the more queries a call holds the closer the
    // 64 of a wave lie and the more of their walks the caches share -- BASELINE configuration 4 at full size, per frame:
    // 2^17 3 561 ms, 2^18 3 187 (the default until round 3; 10.6 GB), 2^19 2 883, 2^20 2 630, 2^21 2 533, 2^22 the same (one chunk).
    // RT_PM_CHUNK_LOG2 = 12 .. 22 overrides; without it the chunk is halved until its arena fits into half of the device memory
    // that is free now (counting what this scene's arena already holds), so a crowded GPU gets smaller chunks, not RT_E_NOMEM.
    static const int chunk_env = [] { const char* e = getenv("RT_PM_CHUNK_LOG2"); const int v = e ? atoi(e) : 0; return v == 0 ? 0 : (v < 12 ? 12 : (v > 22 ? 22 : v)); }();
    int chunk_log2 = chunk_env ? chunk_env : 21;
    if (!chunk_env) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double room = 0.5 * ((double)free_b + (double)s->scratch_cap[P_G_SLOTS]);
            auto arena = [&](int lg) {
                const int64_t tl = std::min<int64_t>(ntiles, std::max<int64_t>(1, ((int64_t)1 << lg) / (64 * ntap)));
                return (double)(tl * ntap * 64) * (double)std::max(G, 1) * 4.0 * 202.0;      // two doublings: the worst case of depth 0
            };
            while (chunk_log2 > 14 && arena(chunk_log2) > room) --chunk_log2;
        }
    }
    const int64_t chunk_tiles = std::max<int64_t>(1, ((int64_t)1 << chunk_log2) / (64 * ntap));
    const int64_t T = std::min(ntiles, chunk_tiles);
    const int64_t nps = T * 64, nsmp = nps * ntap;

    void *d_tiles, *pixmap, *white, *hitany, *accum, *ctr;
    RTCHK(s->scratch(P_TILES, (size_t)std::max<int64_t>(ntiles, 1) * 4, &d_tiles));
    HIPCHK(hipMemcpyAsync(d_tiles, tiles.data(), (size_t)ntiles * 4, hipMemcpyHostToDevice, st), "copy tiles");
    RTCHK(s->scratch(P_PIXMAP, (size_t)nps * 4, &pixmap));
    RTCHK(s->scratch(P_WHITE, (size_t)nps, &white));
    RTCHK(s->scratch(P_HITANY, (size_t)nps, &hitany));
    RTCHK(s->scratch(P_ACCUM, (size_t)nsmp * 12, &accum));
    RTCHK(s->scratch(P_CTR, 8 * sizeof(unsigned long long) + 16, &ctr));
    unsigned long long* d_ctr = (unsigned long long*)ctr;
    uint32_t* d_total = (uint32_t*)(d_ctr + 8);
    HIPCHK(hipMemsetAsync(ctr, 0, 8 * sizeof(unsigned long long) + 16, st), "memset counters");

    // the caustic estimate's strict redo beside the gather rays (below): not while per-call times are asked for (rt_stats:
    // knn_ms brackets a whole call on one stream), not with RT_RENDER_OVERLAP=0
    const bool side_redo = s->overlap_shadow && !stats && s->knn_fast;
    if (side_redo && !s->aux_stream) HIPCHK(hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking), "hipStreamCreate");
    EventBag bag;                          // destroyed on every return path
    std::vector<hipEvent_t> knn_ev;        // begin / end of every radiance-estimate call
    EventBag* kb = stats ? &bag : nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (stats) { hipError_t err; ev0 = bag.mark(st, &err); if (!ev0) return rt_hipfail(err, "event"); }
    int64_t passes = 0;
    const float inv_g = G > 0 ? (float)(1.0 / (double)(float)G) : 0.f;       // vec3 / (float)G, algebra3.cpp:345-347

    for (int64_t tile0 = 0; tile0 < ntiles; tile0 += T) {
        const int64_t nt = std::min(T, ntiles - tile0);
        int64_t n = nt * ntap * 64;
        void *qr, *qs, *qw, *qc;
        RTCHK(s->scratch(P_QA_RAYS, (size_t)n * sizeof(rt_ray), &qr));
        RTCHK(s->scratch(P_QA_SID, (size_t)n * 4, &qs));
        RTCHK(s->scratch(P_QA_W, (size_t)n * 4, &qw));
        RTCHK(s->scratch(P_QA_CODE, (size_t)n * 4, &qc));
        HIPCHK(rtk_pm_generate(s->view, *cam, (const int32_t*)d_tiles, tile0, nt, ntap, use_filter, s->eps_f, (rt_ray*)qr,
                               (int32_t*)qs, (float*)qw, (uint32_t*)qc, (int32_t*)pixmap, (uint8_t*)white, (uint8_t*)hitany,
                               d_hit_ids, (float*)accum, d_ctr, st), "k_pm_generate");
        const rt_ray* q_rays = (const rt_ray*)qr; const int32_t* q_sid = (const int32_t*)qs;
        const float* q_w = (const float*)qw; const uint32_t* q_code = (const uint32_t*)qc;

        for (int depth = 0; depth <= p->max_depth && n > 0; ++depth) {
            ++passes;
            const bool spawn = depth < p->max_depth;
            void *hits, *flags, *nrm, *nrmL, *qpos, *caustic, *occ, *contrib, *bsums;
            RTCHK(s->scratch(P_HITS, (size_t)n * sizeof(rt_hit), &hits));
            RTCHK(s->scratch(P_FLAGS, (size_t)n * 3, &flags));
            RTCHK(s->scratch(P_NRM, (size_t)n * 12, &nrm));
            RTCHK(s->scratch(P_NRML, (size_t)n * 12, &nrmL));
            RTCHK(s->scratch(P_QPOS, (size_t)n * 12, &qpos));
            RTCHK(s->scratch(P_CAUSTIC, (size_t)n * 12, &caustic));
            RTCHK(s->scratch(P_OCC, (size_t)n * std::max(S, 1), &occ));
            RTCHK(s->scratch(P_CONTRIB, (size_t)n * 12, &contrib));
            // gather queues: a refracting hit spawns two children per gather ray (TraceIrradiance, photon.cpp:808), once per
            // level below this one: room for every doubling that IRRADIANCE_MAX_DEPTH still allows (two at depth 0 with the
            // reference's 3), at most three; deeper settings on glass-heavy scenes end in RT_E_LIMIT below
            const int doublings = std::min(3, std::max(1, p->irradiance_max_depth - depth - 1));
            const int64_t gmax = (n * std::max(G, 1)) << doublings;
            RTCHK(s->scratch(P_BSUMS, (size_t)rtk_scan_blocks_needed(std::max(3 * n, 2 * gmax)) * 4 + 16, &bsums));

            HIPCHK(rtk_trace_closest(s->view, q_rays, q_sid, n, s->eps_f, s->fast, (rt_hit*)hits, st), "k_trace_closest");
            HIPCHK(rtk_pm_shade(s->view, q_rays, (const rt_hit*)hits, q_sid, n, ntap, depth == 0, spawn, p->start_nr,
                                (uint8_t*)flags, (float*)nrm, (float*)nrmL, (float*)qpos, (uint8_t*)hitany,
                                depth == 0 ? d_hit_ids : nullptr, (const int32_t*)pixmap, d_ctr, st), "k_pm_shade");
            HIPCHK(rtk_pm_shadow(s->view, (const rt_hit*)hits, q_sid, (const float*)nrmL, n, S, s->eps_f, s->fast,
                                 (uint8_t*)occ, d_ctr, st), "k_pm_shadow");
            // The caustic estimate's strict redo -- a handful of queries with an exact tie, whose replay of the reference's walk
            // takes up to 200 ms where the caustic map's heap has no spatial order (coplanar photons on the floor; 1.6 of the
            // 4.3 s of config 4's frame) -- runs on the scene's second stream, followed there by what needs its result (direct
            // light + caustic), while this stream goes on with the gather rays.  That term is accumulated on the second stream
            // after the pass's gather terms (below), and the pass joins before its children are spawned: one fixed order of
            // additions per sample, whatever the tiling.
            const bool side = side_redo && G > 0 && depth + 1 <= p->irradiance_max_depth;
            hipStream_t sd = side ? s->aux_stream : st;
            RTCHK(knn(s, RT_PMAP_CAUSTIC, (const float*)qpos, (const float*)nrm, n, K, p->knn_radius_caustic, (float*)caustic, st, kb, &knn_ev,
                      side ? sd : nullptr));
            if (side) {         // (if nothing was flagged the redo stream has not waited for st yet: make it)
                hipError_t err; hipEvent_t e = bag.mark(st, &err);
                if (!e) return rt_hipfail(err, "event");
                HIPCHK(hipStreamWaitEvent(sd, e, 0), "hipStreamWaitEvent");
            }
            HIPCHK(rtk_pm_direct(s->view, (const rt_hit*)hits, q_sid, q_w, (const float*)nrmL, n, S, (const uint8_t*)occ,
                                 (const float*)caustic, cam->center, (float*)contrib, sd), "k_pm_direct");
            // (with the redo on the side, this pass's direct + caustic term joins the samples AFTER its gather terms: the whole
            // gather loop, not just its first level, runs beside the redo; a fixed order either way)
            if (!side) HIPCHK(rtk_accumulate(q_sid, (const float*)contrib, n, (float*)accum, st), "k_accumulate");

            // ---- global illumination: gather rays through TraceIrradiance (depth+1 .. IRRADIANCE_MAX_DEPTH) ----
            if (G > 0 && depth + 1 <= p->irradiance_max_depth) {
                // gather state lives in one arena: two queues (rays, sid, w) + per-ray hit / query / result arrays
                const size_t per = sizeof(rt_ray) * 2 + 4 * 4 + sizeof(rt_hit) + 2 + 12 * 4 + 4 + 64;
                void* gbase; RTCHK(s->scratch(P_G_SLOTS, (size_t)gmax * per + 8192, &gbase));
                Arena A; A.base = (char*)gbase; A.cap = (size_t)gmax * per + 8192;
                rt_ray* gr[2] = {A.take<rt_ray>(gmax), A.take<rt_ray>(gmax)};
                int32_t* gs[2] = {A.take<int32_t>(gmax), A.take<int32_t>(gmax)};
                float* gw[2] = {A.take<float>(gmax), A.take<float>(gmax)};
                rt_hit* ghits = A.take<rt_hit>(gmax);
                uint8_t* gflags = A.take<uint8_t>(2 * gmax);
                float* gqpos = A.take<float>(3 * gmax); float* gqnrm = A.take<float>(3 * gmax);
                float* girr = A.take<float>(3 * gmax); float* gcontrib = A.take<float>(3 * gmax);
                float* gkd = A.take<float>(gmax);
                if (!gkd) return rt_fail(RT_E_NOMEM, "rt_render: gather arena too small");
                HIPCHK(rtk_pm_gather_spawn((const rt_hit*)hits, q_sid, q_w, q_code, (const float*)nrmL, (const int32_t*)pixmap, n,
                                           G, ntap, p->seed, gr[0], gs[0], gw[0], st), "k_pm_gather_spawn");
                int64_t gn = n * G;
                int cur = 0;
                for (int gd = depth + 1; gd <= p->irradiance_max_depth && gn > 0; ++gd) {
                    const int last = gd + 1 > p->irradiance_max_depth;
                    HIPCHK(rtk_trace_closest(s->view, gr[cur], gs[cur], gn, s->eps_f, s->fast, ghits, st), "k_trace_closest");
                    HIPCHK(rtk_gi_shade(s->view, gr[cur], ghits, gs[cur], gn, last, p->start_nr, gflags, gqpos, gqnrm, gkd,
                                        d_ctr, st), "k_gi_shade");
                    RTCHK(knn(s, RT_PMAP_GLOBAL, gqpos, gqnrm, gn, K, p->knn_radius_global, girr, st, kb, &knn_ev));
                    HIPCHK(rtk_gi_contrib(gs[cur], gw[cur], gkd, girr, gn, inv_g, gcontrib, st), "k_gi_contrib");
                    HIPCHK(rtk_accumulate(gs[cur], gcontrib, gn, (float*)accum, st), "k_accumulate");
                    if (last) break;
                    HIPCHK(rtk_scan_flags(gflags, 2 * gn, (uint32_t*)bsums, d_total, st), "scan");
                    uint32_t total = 0;
                    HIPCHK(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, st), "read child count");
                    HIPCHK(hipStreamSynchronize(st), "sync");
                    if ((int64_t)total > gmax)
                        return rt_fail(RT_E_LIMIT, "rt_render: %u gather children do not fit the queue of %lld", total, (long long)gmax);
                    HIPCHK(rtk_gi_spawn(s->view, gr[cur], ghits, gs[cur], gw[cur], gflags, 2 * gn, (uint32_t*)bsums, d_total,
                                        p->start_nr, gr[cur ^ 1], gs[cur ^ 1], gw[cur ^ 1], st), "k_gi_spawn");
                    gn = (int64_t)total; cur ^= 1;
                }
            }
            if (side) {
                hipError_t err; hipEvent_t e = bag.mark(st, &err);          // the gather terms are in
                if (!e) return rt_hipfail(err, "event");
                HIPCHK(hipStreamWaitEvent(sd, e, 0), "hipStreamWaitEvent");
                HIPCHK(rtk_accumulate(q_sid, (const float*)contrib, n, (float*)accum, sd), "k_accumulate");
                hipEvent_t direct_done = bag.mark(sd, &err);
                if (!direct_done) return rt_hipfail(err, "event");
                HIPCHK(hipStreamWaitEvent(st, direct_done, 0), "hipStreamWaitEvent");
            }
            if (!spawn) break;
            // ---- children of the eye path -> the other queue ----
            const bool toB = (depth % 2) == 0;
            void *orr, *os, *ow, *oc;
            RTCHK(s->scratch(toB ? P_QB_RAYS : P_QA_RAYS, (size_t)n * 3 * sizeof(rt_ray), &orr));
            RTCHK(s->scratch(toB ? P_QB_SID : P_QA_SID, (size_t)n * 3 * 4, &os));
            RTCHK(s->scratch(toB ? P_QB_W : P_QA_W, (size_t)n * 3 * 4, &ow));
            RTCHK(s->scratch(toB ? P_QB_CODE : P_QA_CODE, (size_t)n * 3 * 4, &oc));
            HIPCHK(rtk_pm_spawn(s->view, q_rays, (const rt_hit*)hits, q_sid, q_w, q_code, (const uint8_t*)flags, 3 * n,
                                (uint32_t*)bsums, d_total, p->start_nr, (rt_ray*)orr, (int32_t*)os, (float*)ow, (uint32_t*)oc,
                                d_ctr, st), "k_pm_spawn");
            uint32_t total = 0;
            HIPCHK(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, st), "read child count");
            HIPCHK(hipStreamSynchronize(st), "sync");
            q_rays = (const rt_ray*)orr; q_sid = (const int32_t*)os; q_w = (const float*)ow; q_code = (const uint32_t*)oc;
            n = (int64_t)total;
        }
        HIPCHK(rtk_pm_finalize((const int32_t*)pixmap, nt * 64, ntap, use_filter, (const float*)accum, (const uint8_t*)white,
                               (const uint8_t*)hitany, p->background, d_rgb8, s->d_radiance_out, st), "k_pm_finalize");
    }
    if (stats) {
        { hipError_t err; ev1 = bag.mark(st, &err); if (!ev1) return rt_hipfail(err, "event"); }
        unsigned long long h[8];
        HIPCHK(hipMemcpyAsync(h, d_ctr, sizeof h, hipMemcpyDeviceToHost, st), "read counters");
        HIPCHK(hipStreamSynchronize(st), "sync");
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev0, ev1), "event time");
        memset(stats, 0, sizeof *stats);
        for (size_t i = 0; i + 1 < knn_ev.size(); i += 2) {
            float a = 0.f;
            HIPCHK(hipEventElapsedTime(&a, knn_ev[i], knn_ev[i + 1]), "event time");
            stats->knn_ms += a; stats->knn_launches += 1;
        }
        stats->rays_primary = (int64_t)h[0]; stats->rays_gather = (int64_t)h[2];
        stats->rays_reflect = (int64_t)h[3]; stats->rays_refract = (int64_t)h[4];
        stats->rays_shadow = (int64_t)h[1] * S;                   // k_pm_shadow: S rays per eye-path hit
        stats->knn_lookups = (int64_t)h[5] + (int64_t)h[1];       // global lookups + one caustic lookup per eye-path hit
        stats->passes = passes; stats->kernel_ms = ms;
    }
    return RT_OK;
}
