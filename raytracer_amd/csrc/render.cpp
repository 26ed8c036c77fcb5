// render.cpp -- rt_render / rt_render_dev: the frame driver that replaces
// GenerateColorImage (HW4/hw4/hw4.cpp:1081-1129) with a wavefront pipeline:
// generate -> [trace_closest -> shade -> trace_any -> light -> accumulate ->
// compact children]* -> quantise.  One pass per recursion depth of
// SimpleMeshRayTrace (hw4.cpp:794-955; depth limit :796).
#include "rt_scene.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

enum Slot {   // scratch slots of rt_scene used by the renderer
    S_TILES = 3, S_RAYS0, S_PIX0, S_HITS, S_SRAYS, S_SACTIVE, S_OCC, S_CONTRIB, S_FLAGS, S_BSUMS,
    S_QA_RAYS, S_QA_PIX, S_QA_W, S_QA_NR, S_QB_RAYS, S_QB_PIX, S_QB_W, S_QB_NR, S_ACCUM, S_HITANY, S_COUNTERS, S_PAIRS, S_WORK, S_W0, S_CODE0, S_PIXLIST, S_QA_CODE, S_QB_CODE, S_ORDER_C, S_ORDER_A,
    S_HITS1, S_SRAYS1, S_SACTIVE1, S_HITS2, S_SRAYS2, S_SACTIVE2, S_QC_RAYS, S_QC_PIX, S_QC_W, S_QC_NR, S_QC_CODE,
    S_PATH_CONTRIB, S_PATH_STACK, S_Q_OCT, S_QA_SRC, S_QB_SRC, S_QC_SRC
};
static_assert(S_QC_SRC < RT_NSCRATCH, "not enough scratch slots");

#define HIPCHK(expr, what) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return rt_hipfail(e__, what); } while (0)
#define RTCHK(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

}  // namespace

int render_photon_impl(rt_scene* s, const rt_camera* cam, const rt_params* p, uint8_t* d_rgb8, int32_t* d_hit_ids,
                       rt_stats* stats, hipStream_t st);                                           // render_photon.cpp

// micro-tiles (8x8 pixels) of the tiles this rank owns; tiles are tile_size^2 pixels,
// numbered row-major, dealt round-robin over ranks.
void rank_microtiles(const rt_camera& cam, const rt_params& p, std::vector<int32_t>& out) {
    const int mtw = (cam.resx + 7) / 8, mth = (cam.resy + 7) / 8;
    if (p.tile_size <= 0 || p.tile_nranks <= 1) {
        out.resize((size_t)mtw * mth);
        for (size_t i = 0; i < out.size(); ++i) out[i] = (int32_t)i;
        return;
    }
    const int ts = ((p.tile_size + 7) / 8) * 8;       // tiles are whole micro-tiles
    const int tw = (cam.resx + ts - 1) / ts, th = (cam.resy + ts - 1) / ts;
    const int per = ts / 8;
    out.clear();
    for (int t = p.tile_rank; t < tw * th; t += p.tile_nranks) {
        const int tx = t % tw, ty = t / tw;
        for (int my = ty * per; my < (ty + 1) * per && my < mth; ++my)
            for (int mx = tx * per; mx < (tx + 1) * per && mx < mtw; ++mx)
                out.push_back(my * mtw + mx);
    }
}

namespace {

int render_impl(rt_scene* s, const rt_camera* cam, const rt_params* p, uint8_t* d_rgb8, int32_t* d_hit_ids,
                rt_stats* stats, hipStream_t st) {
    if (p->mode == RT_RENDER_PHOTON) return render_photon_impl(s, cam, p, d_rgb8, d_hit_ids, stats, st);
    if (p->mode != RT_RENDER_WHITTED) return rt_fail(RT_E_INVALID, "rt_render: unknown mode %d", p->mode);
    const int64_t npix = (int64_t)cam->resx * cam->resy;
    // the rank's micro-tile list: kept from call to call (a frame loop asks for the same list; building and uploading
    // 130 k entries per 4K frame would cost a fraction of the frame itself)
    const bool sharded = p->tile_size > 0 && p->tile_nranks > 1;
    const int key[5] = {cam->resx, cam->resy, sharded ? p->tile_size : 0, sharded ? p->tile_rank : 0, sharded ? p->tile_nranks : 1};
    bool tiles_fresh = false;
    if (memcmp(key, s->tiles_key, sizeof key) != 0) {
        rank_microtiles(*cam, *p, s->tiles_host);
        memcpy(s->tiles_key, key, sizeof key);
        tiles_fresh = true;
    }
    if (tiles_fresh && s->d_tiles_cache) {                           // the device copy belongs to the old key
        HIPCHK(hipStreamSynchronize(st), "sync");
        (void)hipFree(s->d_tiles_cache); s->d_tiles_cache = nullptr;
    }
    const std::vector<int32_t>& tiles = s->tiles_host;
    const int64_t ntiles = (int64_t)tiles.size();
    // distributed ray tracing: any of the build-defined samplers switches the sample bookkeeping on
    const int spp = p->spp > 1 ? p->spp : 1;
    const bool drt = spp > 1 || p->lens_radius > 0.f || p->light_radius > 0.f || p->glossy > 0.f;
    if (spp > 4096) return rt_fail(RT_E_LIMIT, "rt_render: spp %d > 4096", spp);
    // the frame is traced in chunks of micro-tiles of at most 2^27 primary rays (about 25 GB of queues; fewer, larger launches are faster: every launch of incoherent
    // rays ends in a tail of long waves), so memory is
    // bounded whatever resolution x spp is asked for; BASELINE's 4K configuration is one chunk
    const int chunk_log2 = [] { const char* e = getenv("RT_CHUNK_LOG2"); const int v = e ? atoi(e) : 27; return v < 12 || v > 40 ? 27 : v; }();
    const int64_t chunk_tiles = std::max<int64_t>(1, (((int64_t)1 << chunk_log2) / spp) / 64);
    const int L = s->view.nlights;
    const bool whole = (p->tile_size <= 0 || p->tile_nranks <= 1) && ntiles <= chunk_tiles;

    void *d_tiles = nullptr, *rays0, *pix0, *accum, *hitany, *ctr;
    if (!whole) {
        if (tiles_fresh || !s->d_tiles_cache) {
            HIPCHK(hipStreamSynchronize(st), "sync");                 // nothing in flight may still read the old list
            if (s->d_tiles_cache) { (void)hipFree(s->d_tiles_cache); s->d_tiles_cache = nullptr; }
            HIPCHK(hipMalloc((void**)&s->d_tiles_cache, (size_t)std::max<int64_t>(ntiles, 1) * 4), "hipMalloc tiles");
            HIPCHK(hipMemcpy(s->d_tiles_cache, tiles.data(), (size_t)ntiles * 4, hipMemcpyHostToDevice), "copy tiles");
        }
        d_tiles = s->d_tiles_cache;
    }
    const int64_t max_n0 = std::min(ntiles, chunk_tiles) * 64 * spp;
    RTCHK(s->scratch(S_RAYS0, (size_t)max_n0 * sizeof(rt_ray), &rays0));
    RTCHK(s->scratch(S_PIX0, (size_t)max_n0 * 4, &pix0));
    RTCHK(s->scratch(S_ACCUM, (size_t)npix * 12, &accum));
    RTCHK(s->scratch(S_HITANY, (size_t)npix, &hitany));
    RTCHK(s->scratch(S_COUNTERS, 8 * sizeof(unsigned long long) + 16, &ctr));
    unsigned long long* d_ctr = (unsigned long long*)ctr;
    uint32_t* d_total = (uint32_t*)(d_ctr + 8);
    HIPCHK(hipMemsetAsync(ctr, 0, 8 * sizeof(unsigned long long) + 16, st), "memset counters");

    unsigned long long* d_work = nullptr;       // count_work: [0..3] closest, [4..7] any of the wave-synchronous passes, [8..15] the same of the
                                                // per-lane passes, [16..23] the path kernel (box, tri of its closest walks; [20], [21] of its shadow walks)
    if (stats && p->count_work) {
        void* w; RTCHK(s->scratch(S_WORK, 24 * sizeof(unsigned long long), &w));
        HIPCHK(hipMemsetAsync(w, 0, 24 * sizeof(unsigned long long), st), "memset work counters");
        d_work = (unsigned long long*)w;
    }

    EventBag bag;                     // every event of this call; destroyed on every return path
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> kev;      // per-kernel event pairs: [closest begin, closest end, any begin, any end] per pass
    std::vector<char> pass_lane;      // per pass: 1 = traced by the per-lane kernels, 2 = the path kernel (every depth from here on)
    int64_t path_roots = 0;
    auto mark = [&](hipStream_t on) -> int {
        if (!stats) return RT_OK;
        hipError_t err; hipEvent_t e = bag.mark(on, &err);
        if (!e) return rt_hipfail(err, "event");
        kev.push_back(e);
        return RT_OK;
    };
    if (stats) { hipError_t err; ev0 = bag.mark(st, &err); if (!ev0) return rt_hipfail(err, "event"); }

    // BASELINE configuration 2 (primary + shadow rays, one centre sample) has a kernel of its own
    const bool fused = s->fused_primary && p->primary_only && !drt && (s->fast == 1 || s->fast == 3) && !(stats && p->count_work) &&
                       L <= rtk_primary_only_max_lights();
    int64_t passes = 0;
    std::vector<int64_t> pass_n;      // rays of every pass (RT_TRACE_PASSES=1 prints them with the kernel times)
    for (int64_t tile0 = 0; tile0 < ntiles; tile0 += chunk_tiles) {
    const int64_t ctiles = std::min(chunk_tiles, ntiles - tile0);
    const int32_t* c_tiles = d_tiles ? (const int32_t*)d_tiles + tile0 : nullptr;
    const int64_t nslots = ctiles * 64;
    const int64_t n0 = nslots * spp;
    const int32_t* pixlist = (const int32_t*)pix0;      // one entry per pixel slot of this chunk (zero / finalize)
    const float* q_w = nullptr;          // NULL = weight 1
    const uint32_t* q_code = nullptr;    // NULL = no sample bookkeeping (the reference's single centre sample)
    if (fused) {
        // primary + shadow rays only: the chunk's whole frame in one kernel (kernels_primary.hip); nothing goes through HBM
        // but the pixels.  Timed as the "closest" kernel; there is no separate any-hit launch.
        const int64_t nblocks = ctiles;                       // one pair of counts per wave, if the caller wants rt_stats
        void* pairs = nullptr;
        if (stats) RTCHK(s->scratch(S_PAIRS, (size_t)nblocks * 8 + 16, &pairs));
        ++passes; pass_n.push_back(n0); pass_lane.push_back(0);
        RTCHK(mark(st));
        HIPCHK(rtk_primary_only(s->view, *cam, c_tiles, ctiles, s->eps_f, p->ambient_intensity, p->background, d_rgb8, d_hit_ids,
                                s->d_radiance_out, (unsigned int*)pairs, s->fast == 3 ? 1 : 0, p->packed_output ? tile0 * 64 : -1, st),
               "k_primary_only");
        RTCHK(mark(st)); RTCHK(mark(st)); RTCHK(mark(st));
        if (stats) HIPCHK(rtk_reduce_pairs((const unsigned int*)pairs, nblocks, d_ctr, st), "k_reduce_pairs");
        continue;
    }
    if (drt) {
        void *w0, *code0, *pl;
        RTCHK(s->scratch(S_W0, (size_t)n0 * 4, &w0));
        RTCHK(s->scratch(S_CODE0, (size_t)n0 * 4, &code0));
        RTCHK(s->scratch(S_PIXLIST, (size_t)nslots * 4, &pl));
        HIPCHK(rtk_generate_drt(*cam, c_tiles, ctiles, spp, p->seed, p->lens_radius, p->focal_distance,
                                (rt_ray*)rays0, (int32_t*)pix0, (float*)w0, (uint32_t*)code0, (int32_t*)pl, st), "k_generate_drt");
        pixlist = (const int32_t*)pl; q_w = (const float*)w0; q_code = (const uint32_t*)code0;
    } else {
        HIPCHK(rtk_generate_tiled(*cam, c_tiles, ctiles, 0.f, 0.f, (rt_ray*)rays0, (int32_t*)pix0, st), "k_generate_tiled");
    }
    HIPCHK(rtk_zero_pixels(pixlist, nslots, (float*)accum, (uint8_t*)hitany, st), "k_zero_pixels");

    // pass state
    const rt_ray* q_rays = (const rt_ray*)rays0;
    const int32_t* q_pix = (const int32_t*)pix0;
    const float* q_nr = nullptr;         // NULL = start_nr ... handled below
    int64_t n = n0;
    // The shadow half of a pass (any hit -> lights -> accumulate) does not feed the next pass: the children come from the
    // closest hits alone.  It runs on the scene's second stream, beside the next pass's closest-hit traversal -- in the deep
    // passes of an incoherent frame two per-lane kernels side by side take the time of one.  A buffer set is
    // rewritten only after the shadow half that reads it is done (events; below).  Accumulation stays in pass order on that
    // one stream, so the frame's bytes do not change.
    const bool overlap = s->overlap_shadow && L > 0 && !d_work && !s->ray_sort && !p->primary_only && p->max_depth > 0;
    hipStream_t sb = st;
    if (overlap) {
        if (!s->aux_stream) HIPCHK(hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking), "hipStreamCreate");
        sb = s->aux_stream;
    }
    DevSceneView view = s->view;
    if (overlap) view.lane_waves = 3072;                 // two per-lane kernels side by side (kernels_trace.hip: lanes_grid)
    // With two streams the per-pass buffers rotate through THREE sets (hits, shadow rays, ray queues): the closest-hit stream
    // may then run a whole pass ahead of the shadow stream -- with two sets it waited for the shadow half of pass d - 1
    // before it could write pass d + 1's queue (2.3 ms of the 85 ms frame, profiles/r02_timeline_drt_1m.txt).
    const int nsets = overlap ? 3 : 2;
    std::vector<hipEvent_t> b_done;                     // b_done[d]: the shadow half of pass d is done
    hipEvent_t b_last = nullptr;
    static const int HITS_SLOT[3] = {S_HITS, S_HITS1, S_HITS2}, SRAYS_SLOT[3] = {S_SRAYS, S_SRAYS1, S_SRAYS2},
                     SACT_SLOT[3] = {S_SACTIVE, S_SACTIVE1, S_SACTIVE2};
    static const int Q_RAYS[3] = {S_QA_RAYS, S_QB_RAYS, S_QC_RAYS}, Q_PIX[3] = {S_QA_PIX, S_QB_PIX, S_QC_PIX}, Q_W[3] = {S_QA_W, S_QB_W, S_QC_W},
                     Q_NR[3] = {S_QA_NR, S_QB_NR, S_QC_NR}, Q_CODE[3] = {S_QA_CODE, S_QB_CODE, S_QC_CODE},
                     Q_SRC[3] = {S_QA_SRC, S_QB_SRC, S_QC_SRC};
    // per-lane walks start at the leaf of the triangle their ray starts on (trace_device.h: bvh_lane_start_from): the children
    // of a queue carry that triangle beside them when the pass that takes them walks per lane
    const int32_t* q_src = nullptr;
    const bool origin_walks = s->view.prim_start != nullptr;
    hipEvent_t order_done = nullptr;                    // the second stream has ordered its shadow rays (rtk_ray_order)
    const uint8_t* q_oct = nullptr;                     // the queue's direction octants, one byte per ray (k_spawn), or NULL
    static const bool oct_partition = [] { const char* e = getenv("RT_OCT_PARTITION"); return e ? atoi(e) != 0 : true; }();
    // The rays of depth >= RT_OPT_PATH_DEPTH with everything below them: one persistent kernel (kernels_paths.hip) instead of a
    // pass of kernels and a host round trip per depth.  Needs the build's hierarchy and a queue made by k_spawn (depth >= 1).
    const bool use_paths = (s->fast == 1 || s->fast == 3) && s->view.nbvh > 0 && s->view.ray_counter != nullptr && s->path_depth >= 1 &&
                           s->path_depth <= p->max_depth && !p->primary_only && !s->ray_sort && L <= rtk_paths_max_lights();
    for (int depth = 0; depth <= p->max_depth && n > 0; ++depth) {
        // ... when the queue is small: the kernel needs 128 registers per lane (4 waves per SIMD), and on a queue that fills the
        // chip many times over the lean per-depth kernels (6 waves per SIMD, closest and shadow walks on two streams) get more
        // walks in flight -- 1 M-triangle mesh, 1080p x 64 spp, depth >= 2: 9.3 M entries, passes 40 ms, path kernel 51 ms; one
        // rank's share of that frame at 8 ranks: 0.97 M entries, passes 10.3 ms, path kernel 9.9 ms and no per-depth round trips
        // (frame 20.5 -> 16.5 ms).  RT_PATH_MAX_ROOTS overrides the threshold.
        // The choice is made once, at the option's depth (deeper queues are always smaller).
        static const int64_t path_max_roots = [] { const char* e = getenv("RT_PATH_MAX_ROOTS"); return e && atoll(e) > 0 ? atoll(e) : (int64_t)16000000; }();
        if (use_paths && depth == s->path_depth && n <= path_max_roots) {
            void *pc, *stk;
            RTCHK(s->scratch(S_PATH_CONTRIB, (size_t)n * 12, &pc));
            RTCHK(s->scratch(S_PATH_STACK, rtk_paths_stack_bytes(n, depth, p->max_depth), &stk));
            ++passes; pass_n.push_back(n); pass_lane.push_back(2); path_roots += n;
            if (order_done) { HIPCHK(hipStreamWaitEvent(st, order_done, 0), "hipStreamWaitEvent"); order_done = nullptr; }
            RTCHK(mark(st));
            // RT_PATH_SORT=1 (experiment): the entries are handed out in the order of their origins' cells, so the range an XCD
            // claims from is one region of the scene
            static const bool path_sort = [] { const char* e = getenv("RT_PATH_SORT"); return e && atoi(e) != 0; }();
            const int32_t* order_p = nullptr;
            if (path_sort) {
                void* ow; RTCHK(s->scratch(S_ORDER_C, rtk_ray_order_workspace(n), &ow));
                HIPCHK(rtk_ray_order(q_rays, q_pix, n, s->bvh_lo, s->bvh_hi, ow, &order_p, st), "ray order");
            }
            HIPCHK(rtk_paths(view, q_rays, q_pix, q_w, q_nr, q_code, n, depth, p->max_depth, s->eps_f, p->ambient_intensity, cam->center,
                             p->start_nr, p->seed, p->light_radius, p->glossy, (float*)pc, stk, d_ctr, d_work ? d_work + 16 : nullptr, st, order_p, q_src),
                   "k_paths_bvh_lanes");
            RTCHK(mark(st)); RTCHK(mark(st)); RTCHK(mark(st));
            // the entries' sums join the pixels after every earlier pass's (the accumulation stream keeps pass order)
            if (overlap) {
                hipError_t err; hipEvent_t e = bag.mark(st, &err);
                if (!e) return rt_hipfail(err, "event");
                HIPCHK(hipStreamWaitEvent(sb, e, 0), "hipStreamWaitEvent");
            }
            HIPCHK(rtk_accumulate(q_pix, (const float*)pc, n, (float*)accum, sb), "k_accumulate");
            if (overlap) {
                hipError_t err; b_last = bag.mark(sb, &err);
                if (!b_last) return rt_hipfail(err, "event");
            }
            break;
        }
        const int pass_mode = ((s->fast == 1 || s->fast == 3) && depth >= s->lane_depth) ? 2 : s->fast;   // RT_OPT_LANE_DEPTH
        ++passes; pass_n.push_back(n); pass_lane.push_back(pass_mode == 2 ? 1 : 0);
        const bool spawn = !p->primary_only && depth < p->max_depth;
        const int set = overlap ? depth % 3 : 0;
        void *hits, *srays, *sact, *occ, *contrib, *flags = nullptr, *bsums = nullptr, *pairs;
        RTCHK(s->scratch(S_PAIRS, (size_t)((n + 255) / 256) * 8 + 16, &pairs));
        RTCHK(s->scratch(HITS_SLOT[set], (size_t)n * sizeof(rt_hit), &hits));
        RTCHK(s->scratch(SRAYS_SLOT[set], (size_t)n * L * sizeof(rt_ray), &srays));
        RTCHK(s->scratch(SACT_SLOT[set], (size_t)n * L * 4, &sact));
        RTCHK(s->scratch(S_OCC, (size_t)n * L, &occ));
        RTCHK(s->scratch(S_CONTRIB, (size_t)n * 12, &contrib));
        if (spawn) {
            RTCHK(s->scratch(S_FLAGS, (size_t)n * 2, &flags));
            RTCHK(s->scratch(S_BSUMS, (size_t)rtk_scan_blocks_needed(2 * n) * 12 + 1024, &bsums));     // offsets + per-block child counts (rtk_compact_children)
        }
        // per-lane passes: rays go to lanes in spatial / octant order (RT_OPT_RAY_SORT); the sort is timed with the pass
        const bool sorted = pass_mode == 2 && s->ray_sort && s->view.nbvh > 0 && (s->fast == 1 || s->fast == 3);
        const int32_t* order_c = nullptr; const int32_t* order_a = nullptr;
        RTCHK(mark(st));
        // wave-synchronous passes of secondary rays: a wave takes the one-octant walk (no min / max per box, nearest child
        // first for every lane) only if all its rays point into one octant, and reflections off a curved surface do not.
        // The closest-hit kernel therefore takes its rays in the order of a stable partition by direction octant (one 4-bit
        // radix pass over (octant, index); within an octant the queue's pixel order stays) -- an order of work, results go to
        // the rays' own slots.  1 M-triangle mesh, 1080p x 64 spp, pass 1 (49.5 M rays): 16.0 -> 11.8 ms with the partition
        // inside.  The shadow rays of those hits all aim at the lights, so their octants say nothing, but their origins -- the
        // hit points of reflections that have spread over the scene -- do: they are ordered by the 32^3 cell of the origin
        // (two radix passes over the key's 15 cell bits): any-hit of pass 1 11.7 -> 8.9 ms, frame 90.5 -> 86.8 ms; cell +
        // octant or the full 30-bit key give the same (87.7 / 87.3), the cell order for the closest-hit rays instead of the
        // octants is slower (92.5), and either order for the per-lane passes is slower, again (90.4 - 97.7).  RT_OCT_PARTITION=0: off.
        const bool by_octant = oct_partition && pass_mode == 3 && depth >= 1 && s->view.nbvh > 0;
        if (sorted) {
            void* ow; RTCHK(s->scratch(S_ORDER_C, rtk_ray_order_workspace(n), &ow));
            HIPCHK(rtk_ray_order(q_rays, q_pix, n, s->bvh_lo, s->bvh_hi, ow, &order_c, st), "ray order");
        } else if (by_octant && q_oct) {
            // a stable partition by octant from the bytes k_spawn left beside the queue: count, scan, scatter
            void* ow; RTCHK(s->scratch(S_ORDER_C, rtk_octant_order_workspace(n), &ow));
            HIPCHK(rtk_octant_order(q_oct, n, ow, &order_c, st), "octant order");
        } else if (by_octant) {
            void* ow; RTCHK(s->scratch(S_ORDER_C, rtk_ray_order_workspace(n), &ow));
            HIPCHK(rtk_ray_order(q_rays, q_pix, n, s->bvh_lo, s->bvh_hi, ow, &order_c, st, RT_ORDER_OCTANT), "ray order");
        }
        if (order_done) { HIPCHK(hipStreamWaitEvent(st, order_done, 0), "hipStreamWaitEvent"); order_done = nullptr; }
        const RaySrc src_c = {(const char*)q_src, 4, 1}, src_a = {origin_walks && pass_mode == 2 ? (const char*)hits : nullptr, (int32_t)sizeof(rt_hit), L > 0 ? L : 1};
        HIPCHK(rtk_trace_closest(view, q_rays, q_pix, n, s->eps_f, pass_mode, (rt_hit*)hits, st, order_c, &src_c), "k_trace_closest");
        RTCHK(mark(st));
        HIPCHK(rtk_shade_flags(s->view, q_rays, (const rt_hit*)hits, q_pix, n, spawn ? 1 : 0, depth == 0 ? 1 : 0,
                               (uint8_t*)flags, (rt_ray*)srays, (int32_t*)sact, depth == 0 ? d_hit_ids : nullptr,
                               (uint8_t*)hitany, d_ctr, (unsigned int*)pairs, q_code, p->seed, p->light_radius, st), "k_shade_flags");
        if (overlap) {
            hipError_t err; hipEvent_t e = bag.mark(st, &err);
            if (!e) return rt_hipfail(err, "event");
            HIPCHK(hipStreamWaitEvent(sb, e, 0), "hipStreamWaitEvent");
        }
        RTCHK(mark(sb));
        if (L > 0) {
            if (sorted || by_octant) {      // wave-synchronous shadow rays of secondary hits: by the cell of their origin (see above)
                void* ow; RTCHK(s->scratch(S_ORDER_A, rtk_ray_order_workspace(n * L), &ow));
                HIPCHK(rtk_ray_order((const rt_ray*)srays, (const int32_t*)sact, n * L, s->bvh_lo, s->bvh_hi, ow, &order_a, sb,
                                     sorted ? RT_ORDER_FULL : RT_ORDER_CELL), "ray order");
                if (overlap && order_a) {       // the next pass's traversal starts after these radix passes: beside a per-lane
                    hipError_t err;             // kernel's scattered reads they get a twentieth of their bandwidth (7 ms for 1 ms)
                    order_done = bag.mark(sb, &err);
                    if (!order_done) return rt_hipfail(err, "event");
                }
            }
            HIPCHK(rtk_trace_any(view, (const rt_ray*)srays, (const int32_t*)sact, nullptr, n * L, s->eps_f, pass_mode, (uint8_t*)occ, sb, order_a, &src_a), "k_trace_any");
        }
        RTCHK(mark(sb));
        if (d_work) {
            unsigned long long* dw = d_work + (pass_mode == 2 ? 8 : 0);
            HIPCHK(rtk_trace_work(s->view, q_rays, q_pix, nullptr, n, s->eps_f, s->fast, 0, dw, st), "k_trace_work");
            if (L > 0)
                HIPCHK(rtk_trace_work(s->view, (const rt_ray*)srays, (const int32_t*)sact, nullptr, n * L, s->eps_f, s->fast, 1,
                                      dw + 4, st), "k_trace_work");
        }
        HIPCHK(rtk_shade_lights(s->view, q_rays, (const rt_hit*)hits, q_pix, q_w, n, (const uint8_t*)occ,
                                p->ambient_intensity, cam->center, (float*)contrib, d_ctr, sb), "k_shade_lights");
        HIPCHK(rtk_accumulate(q_pix, (const float*)contrib, n, (float*)accum, sb), "k_accumulate");
        if (overlap) {
            hipError_t err; b_last = bag.mark(sb, &err);
            if (!b_last) return rt_hipfail(err, "event");
            b_done.push_back(b_last);
        }
        if (!spawn) break;
        // children -> the other queue, which the shadow half of the previous pass may still be reading
        // (pass depth + 1 takes set (depth + 1) % 3, last used by pass depth - 2)
        if (overlap && depth >= 2) HIPCHK(hipStreamWaitEvent(st, b_done[(size_t)depth - 2], 0), "hipStreamWaitEvent");
        const int qo = (depth + 1) % nsets;               // the queue the children go to
        void *orays, *opix, *ow, *onr;
        RTCHK(s->scratch(Q_RAYS[qo], (size_t)n * 2 * sizeof(rt_ray), &orays));
        RTCHK(s->scratch(Q_PIX[qo], (size_t)n * 2 * 4, &opix));
        RTCHK(s->scratch(Q_W[qo], (size_t)n * 2 * 4, &ow));
        RTCHK(s->scratch(Q_NR[qo], (size_t)n * 2 * 4, &onr));
        const float* nr_in = q_nr;
        if (!nr_in && p->start_nr != 1.0f) {
            // first pass with a start index other than 1: materialise it (k_spawn treats NULL as 1.0)
            void* tmp; RTCHK(s->scratch(Q_NR[(qo + 1) % nsets], (size_t)n * 4, &tmp));
            int nr_bits; memcpy(&nr_bits, &p->start_nr, 4);
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)tmp, nr_bits, (size_t)n, st), "fill nr");      // on the device: no pageable host copy
            nr_in = (const float*)tmp;
        }
        void* ocode = nullptr;
        if (drt) RTCHK(s->scratch(Q_CODE[qo], (size_t)n * 2 * 4, &ocode));
        // the next pass, if wave-synchronous, takes its rays by direction octant: k_spawn notes the octants
        void* ooct = nullptr;
        {
            const int next_mode = ((s->fast == 1 || s->fast == 3) && depth + 1 >= s->lane_depth) ? 2 : s->fast;
            const bool next_paths = false;      // (the path kernel takes its queue in queue order; whether it runs is known only with the child count)
            if (oct_partition && next_mode == 3 && !next_paths && s->view.nbvh > 0 && !sorted) RTCHK(s->scratch(S_Q_OCT, (size_t)n * 2, &ooct));
        }
        void* osrc = nullptr;
        {
            const int next_lane = ((s->fast == 1 || s->fast == 3) && depth + 1 >= s->lane_depth) ? 1 : 0;
            if (origin_walks && (next_lane || (use_paths && depth + 1 == s->path_depth))) RTCHK(s->scratch(Q_SRC[qo], (size_t)n * 2 * 4, &osrc));
        }
        HIPCHK(rtk_compact_children(s->view, q_rays, (const rt_hit*)hits, q_pix, q_w, nr_in, (const uint8_t*)flags, 2 * n,
                                    (uint32_t*)bsums, d_total, (rt_ray*)orays, (int32_t*)opix, (float*)ow, (float*)onr,
                                    d_ctr, q_code, (uint32_t*)ocode, p->seed, p->glossy, st, (uint8_t*)ooct, (int32_t*)osrc), "compact children");
        uint32_t total = 0;
        HIPCHK(hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, st), "read child count");
        HIPCHK(hipStreamSynchronize(st), "sync");
        q_rays = (const rt_ray*)orays; q_pix = (const int32_t*)opix; q_w = (const float*)ow; q_nr = (const float*)onr;
        q_code = (const uint32_t*)ocode; q_oct = (const uint8_t*)ooct; q_src = (const int32_t*)osrc;
        n = (int64_t)total;
    }
    if (b_last) HIPCHK(hipStreamWaitEvent(st, b_last, 0), "hipStreamWaitEvent");       // the frame needs every pass's sum
    HIPCHK(rtk_finalize(pixlist, nslots, (const float*)accum, (const uint8_t*)hitany, p->background, d_rgb8,
                        p->packed_output ? tile0 * 64 : -1, st), "k_finalize");
    if (s->d_radiance_out) HIPCHK(rtk_copy_radiance(pixlist, nslots, (const float*)accum, s->d_radiance_out, st), "k_copy_radiance");
    }   // chunks
    if (stats) {
        { hipError_t err; ev1 = bag.mark(st, &err); if (!ev1) return rt_hipfail(err, "event"); }
        unsigned long long h[8], hw[24] = {};
        HIPCHK(hipMemcpyAsync(h, d_ctr, sizeof h, hipMemcpyDeviceToHost, st), "read counters");
        if (d_work) HIPCHK(hipMemcpyAsync(hw, d_work, sizeof hw, hipMemcpyDeviceToHost, st), "read work counters");
        HIPCHK(hipStreamSynchronize(st), "sync");
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev0, ev1), "event time");
        memset(stats, 0, sizeof *stats);
        stats->rays_reflect = (int64_t)h[3]; stats->rays_refract = (int64_t)h[4];
        stats->rays_primary = (int64_t)h[0] - stats->rays_reflect - stats->rays_refract;
        stats->rays_shadow = (int64_t)h[1] * L;      // one shadow ray per hit and light (hw4.cpp:929-931)
        stats->passes = passes;
        stats->kernel_ms = ms;
        for (int c = 0; c < 4; ++c) {
            // (the path kernel counts box tests and triangle records only; its slots 2, 3 hold the RT_PATH_DIAG diagnostics)
            stats->closest_work[c] = (int64_t)(hw[c] + hw[8 + c] + (c < 2 ? hw[16 + c] : 0)); stats->any_work[c] = (int64_t)(hw[4 + c] + hw[12 + c] + (c < 2 ? hw[20 + c] : 0));
            stats->lane_closest_work[c] = (int64_t)hw[8 + c]; stats->lane_any_work[c] = (int64_t)hw[12 + c];
        }
        if (d_work && getenv("RT_PATH_DIAG"))
            fprintf(stderr, "path kernel: %llu wave steps, %.1f lanes stepping per step; %llu events, %.1f lanes waiting per event\n", hw[18],
                    hw[18] ? (double)hw[19] / (double)hw[18] : 0.0, hw[22], hw[22] ? (double)hw[23] / (double)hw[22] : 0.0);
        stats->path_work[0] = (int64_t)hw[16]; stats->path_work[1] = (int64_t)hw[17]; stats->path_work[2] = (int64_t)hw[20]; stats->path_work[3] = (int64_t)hw[21];
        stats->path_roots = path_roots; stats->path_closest_rays = (int64_t)h[5]; stats->path_shadow_rays = (int64_t)h[6];
        for (size_t i = 0; i + 3 < kev.size(); i += 4) {
            float a = 0.f, b = 0.f;
            HIPCHK(hipEventElapsedTime(&a, kev[i], kev[i + 1]), "event time");
            HIPCHK(hipEventElapsedTime(&b, kev[i + 2], kev[i + 3]), "event time");
            if (pass_lane[i / 4] == 2) {               // the path kernel: closest and shadow walks of every depth from here on in one launch
                stats->path_ms += a; stats->path_launches += 1;
                if (getenv("RT_TRACE_PASSES"))
                    fprintf(stderr, "pass %zu: %lld queue entries and their subtrees, path kernel %.3f ms\n", i / 4, (long long)pass_n[i / 4], a);
                continue;
            }
            const bool any_launched = L > 0 && !fused;
            stats->trace_closest_ms += a; stats->trace_any_ms += b;
            stats->trace_closest_launches += 1; stats->trace_any_launches += any_launched ? 1 : 0;
            if (pass_lane[i / 4] == 1) {
                stats->lane_closest_ms += a; stats->lane_closest_launches += 1; stats->lane_closest_rays += pass_n[i / 4];
                if (any_launched) { stats->lane_any_ms += b; stats->lane_any_launches += 1; stats->lane_any_rays += pass_n[i / 4] * L; }
            }
            if (getenv("RT_TRACE_PASSES"))
                fprintf(stderr, "pass %zu: %lld rays, closest %.3f ms, any %.3f ms%s\n", i / 4, (long long)pass_n[i / 4], a, b, pass_lane[i / 4] ? " (per lane)" : "");
        }
    }
    return RT_OK;
}

int check_args(rt_scene* s, const rt_camera* cam, const rt_params* p, const void* rgb) {
    if (!s || !cam || !p || !rgb) return rt_fail(RT_E_INVALID, "rt_render: NULL argument");
    if (cam->resx <= 0 || cam->resy <= 0 || (int64_t)cam->resx * cam->resy > (int64_t)1 << 30)
        return rt_fail(RT_E_INVALID, "rt_render: bad resolution %dx%d", cam->resx, cam->resy);
    if (p->max_depth < 0 || p->max_depth > 64) return rt_fail(RT_E_INVALID, "rt_render: max_depth out of range");
    if (p->packed_output && p->mode != RT_RENDER_WHITTED) return rt_fail(RT_E_INVALID, "rt_render: packed_output needs RT_RENDER_WHITTED");
    if (p->tile_nranks > 1 && (p->tile_rank < 0 || p->tile_rank >= p->tile_nranks || p->tile_size <= 0))
        return rt_fail(RT_E_INVALID, "rt_render: bad tile sharding (rank %d of %d, tile %d)", p->tile_rank, p->tile_nranks, p->tile_size);
    return RT_OK;
}

}  // namespace

extern "C" int rt_render_dev(rt_scene* s, const rt_camera* cam, const rt_params* p, uint8_t* d_rgb8, int32_t* d_hit_ids,
                             rt_stats* stats, void* stream) {
    RTCHK(check_args(s, cam, p, d_rgb8));
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    return render_impl(s, cam, p, d_rgb8, d_hit_ids, stats, (hipStream_t)stream);
}

extern "C" int rt_render_pixel_list(const rt_camera* cam, const rt_params* p, int32_t* out, int64_t cap, int64_t* n) {
    if (!cam || !p || !n || (cap > 0 && !out)) return rt_fail(RT_E_INVALID, "rt_render_pixel_list: NULL argument");
    if (cam->resx <= 0 || cam->resy <= 0) return rt_fail(RT_E_INVALID, "rt_render_pixel_list: bad resolution");
    std::vector<int32_t> tiles;
    rank_microtiles(*cam, *p, tiles);
    const int mtw = (cam->resx + 7) / 8;
    *n = (int64_t)tiles.size() * 64;
    for (int64_t k = 0; k < *n && k < cap; ++k) {
        const int t = tiles[(size_t)(k >> 6)], lane = (int)(k & 63);
        const int x = (t % mtw) * 8 + (lane & 7), y = (t / mtw) * 8 + (lane >> 3);
        out[k] = (x < cam->resx && y < cam->resy) ? x + y * cam->resx : -1;
    }
    return RT_OK;
}

extern "C" int rt_render_radiance(rt_scene* s, const rt_camera* cam, const rt_params* p, float* radiance, rt_stats* stats) {
    RTCHK(check_args(s, cam, p, radiance));
    if (p->packed_output) return rt_fail(RT_E_INVALID, "rt_render_radiance: packed_output is for rt_render_dev");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    const size_t npix = (size_t)cam->resx * cam->resy;
    void *d_rgb, *d_rad;
    RTCHK(s->scratch(0, npix * 3, &d_rgb));
    RTCHK(s->scratch(1, npix * 12, &d_rad));
    HIPCHK(hipMemcpy(d_rad, radiance, npix * 12, hipMemcpyHostToDevice), "copy radiance in");   // other ranks' pixels stay
    s->d_radiance_out = (float*)d_rad;
    const int rc = render_impl(s, cam, p, (uint8_t*)d_rgb, nullptr, stats, nullptr);
    s->d_radiance_out = nullptr;
    if (rc) return rc;
    HIPCHK(hipMemcpy(radiance, d_rad, npix * 12, hipMemcpyDeviceToHost), "copy radiance out");
    return RT_OK;
}

extern "C" int rt_render(rt_scene* s, const rt_camera* cam, const rt_params* p, uint8_t* rgb8, int32_t* hit_ids,
                         rt_stats* stats) {
    RTCHK(check_args(s, cam, p, rgb8));
    if (p->packed_output) return rt_fail(RT_E_INVALID, "rt_render: packed_output is for rt_render_dev");
    DeviceGuard g(s->device);
    if (!g.ok) return rt_fail(RT_E_NODEVICE, "hipSetDevice failed");
    const size_t npix = (size_t)cam->resx * cam->resy;
    void *d_rgb, *d_ids = nullptr;
    RTCHK(s->scratch(0, npix * 3, &d_rgb));
    if (hit_ids) RTCHK(s->scratch(1, npix * 4, &d_ids));
    // pixels of other ranks' tiles keep the caller's bytes
    HIPCHK(hipMemcpy(d_rgb, rgb8, npix * 3, hipMemcpyHostToDevice), "copy rgb in");
    if (hit_ids) HIPCHK(hipMemcpy(d_ids, hit_ids, npix * 4, hipMemcpyHostToDevice), "copy ids in");
    RTCHK(render_impl(s, cam, p, (uint8_t*)d_rgb, (int32_t*)d_ids, stats, nullptr));
    HIPCHK(hipMemcpy(rgb8, d_rgb, npix * 3, hipMemcpyDeviceToHost), "copy rgb out");
    if (hit_ids) HIPCHK(hipMemcpy(hit_ids, d_ids, npix * 4, hipMemcpyDeviceToHost), "copy ids out");
    return RT_OK;
}
