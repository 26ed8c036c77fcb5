/*
 * rt_hip.h -- C ABI of librt_hip.so: the MI355X (gfx950) implementation of the
 * ray-traversal / intersection / shading / photon-kNN hot path of
 * clotha87762/RayTracer.
 *
 * The reference has no FFI or plugin interface: everything is free functions
 * over globals in one executable.  The seam is cut at function level, directly
 * under the frame drivers.  Every entry point below cites the reference
 * function it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, POD only, no C++ or torch types cross this boundary;
 *   - every call returns 0 on success and a negative RT_E* code on failure;
 *     rt_last_error() gives a text for the calling thread's last failure;
 *   - nothing here throws, exits or falls back to a CPU path: if no gfx950
 *     device is usable the calls fail with RT_E_NODEVICE;
 *   - "host" entry points take caller-owned host buffers (pageable is fine,
 *     the library stages them) and are synchronous; "_dev" entry points take
 *     device pointers that live on the scene's GPU and enqueue their work on
 *     the caller's HIP stream.  These return WITHOUT host synchronisation:
 *     rt_trace_closest_dev, rt_trace_any_dev, rt_generate_primary_dev,
 *     rt_photon_knn_dev, rt_photon_irradiance_dev / _noalloc_dev, and
 *     rt_render_dev for the primary + shadow frame (rt_params.primary_only, one
 *     sample, no rt_stats).  These synchronise the stream: rt_trace_work_dev
 *     and rt_photonmap_emit_dev (they return counts), rt_photonmap_build_dev
 *     (once per tree level and wide pass), rt_photonmap_import_dev, rt_render_dev
 *     with rt_stats, and rt_render_dev with recursion (once per depth above
 *     RT_OPT_PATH_DEPTH, to size the next pass by the number of child rays --
 *     twice per frame by default; the depths below that option run in one
 *     launch).  The first call of a size also grows the scene's scratch
 *     (hipMalloc; hipFree of the smaller block after a stream synchronise);
 *   - one host thread AND one stream per rt_scene at a time: the scene's
 *     scratch blocks are shared by its calls, so two calls on the same scene
 *     must be ordered on one stream (or by events).  rt_render_dev with
 *     recursion also uses a stream of the scene's own for the shadow half of
 *     every pass; it forks from and joins the caller's stream by events inside
 *     the call, so the caller sees one stream (RT_RENDER_OVERLAP=0: off).  The reference is
 *     single-threaded with global state (hw4.cpp:49-66); different scenes may
 *     be used from different threads and streams.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3

/* error codes */
#define RT_OK            0
#define RT_E_INVALID    -1   /* bad argument (NULL, negative size, bad index)    */
#define RT_E_NODEVICE   -2   /* no usable gfx950 device / HIP runtime failure    */
#define RT_E_NOMEM      -3   /* host or device allocation failed                 */
#define RT_E_LIMIT      -4   /* scene exceeds a compiled-in limit (KD depth ...) */
#define RT_E_STATE      -5   /* call order problem (e.g. kNN before upload)      */
#define RT_E_HIP        -6   /* a HIP call or kernel launch failed               */

typedef struct rt_scene rt_scene;      /* opaque: owns all device memory         */

/* primitive kinds (reference: class Triangle hw4.h:119-142, class Sphere hw4.h:95-117) */
#define RT_PRIM_TRIANGLE 0
#define RT_PRIM_SPHERE   1
#define RT_PRIM_DISABLED 255 /* parsed but unreachable: meshes after the last `O`, hw4.cpp:768 */

/* One primitive, in scene-file order (== index into the reference's
 * `simpleMeshes`, hw4.cpp:60,600,659).  48 bytes.
 *   triangle: f[0..8] = vertices[0..2] (xyz each), f[9..11] = normal[0] as the
 *             reference stores it after parsing (hw4.cpp:629-639: the file's
 *             normal, normalised once; Final: cross product, final.cpp:608-611)
 *   sphere:   f[0..2] = center, f[3] = radius, rest ignored                    */
typedef struct rt_prim {
    float f[12];
} rt_prim;

/* Material, the ten numbers of an `M` line in file order (hw4.cpp:707-728,
 * members hw4.h:53-58).                                                         */
typedef struct rt_material {
    float surf[3];      /* surfColor                                             */
    float Ka, Kd, Ks;
    float specularity;
    float reflect;
    float refract;
    float Nr;
} rt_material;

#define RT_LIGHT_POINT 0  /* `L` / `ML` lines, hw4.cpp:663-700                   */
#define RT_LIGHT_RECT  1  /* `RectL` line, final.cpp:683-707; photon.h:79-112    */
typedef struct rt_light {
    int32_t type;
    float pos[3];
    float color[3];
    float intensity;
    float dir1[3], dir2[3];   /* rect light axes (final.cpp:693-694)             */
    float size1, size2;
    int32_t n1, n2;           /* sampleCountDir1/2                               */
} rt_light;

/* scene flags */
#define RT_SCENE_CULLING       1u  /* `culling` (hw4.cpp:40 false; parameter.h:13 true) */
#define RT_SCENE_KD_REFERENCE  2u  /* build object KD-trees by the reference rule
                                      hw4.cpp:330-490 (the only builder so far)  */

/* Scene description handed over at the end of parsing (replaces hw4.cpp:729-765:
 * the `O` handler that calls KDNode::BuildKD, and final.cpp's plain
 * `simpleMeshes` list).  The library copies everything; the caller may free
 * its arrays when rt_scene_create returns.
 *
 * objects: each [obj_first[i], obj_first[i]+obj_count[i]) range of `prims` is
 * one `O` group and gets its own KD-tree (file order = traversal order,
 * hw4.cpp:808).  Primitives outside every range are "loose": they are tested
 * one by one in index order AFTER all objects (this is the whole scene for the
 * photon renderer, which never builds a KD-tree: final.cpp:802-816).  Spheres
 * inside an object range are treated as loose (the reference's BuildKD cannot
 * hold spheres: hw4.cpp:454-461 pushes a NULL).                                 */
typedef struct rt_scene_desc {
    int32_t            abi_version;     /* RT_ABI_VERSION                        */
    uint32_t           flags;
    int64_t            nprim;
    const rt_prim*     prims;
    const uint8_t*     prim_type;       /* [nprim] RT_PRIM_*                     */
    const int32_t*     prim_material;   /* [nprim] index into materials          */
    int32_t            nmaterial;
    const rt_material* materials;
    int32_t            nobject;
    const int64_t*     obj_first;       /* [nobject]                             */
    const int64_t*     obj_count;       /* [nobject]                             */
    int32_t            nlight;
    const rt_light*    lights;
    double             tolerance_depth; /* TOLERANCE_DEPTH: hw4.cpp:19 1e-5, parameter.h:9 1e-3 */
    float              shadow_bias;     /* hw4.cpp:34  1e-4                      */
    float              refract_bias;    /* hw4.cpp:35  1e-2                      */
    int32_t            device;          /* HIP device ordinal, -1 = current      */
} rt_scene_desc;

/* Numbers the reference prints into executionTime_<input> (hw4.cpp:1244-1245). */
typedef struct rt_scene_info {
    int64_t kd_nodes_built;   /* `builded`: BuildKD calls over all objects       */
    int32_t kd_max_depth;     /* `maxDepth`                                      */
    int64_t flat_nodes;       /* nodes uploaded                                  */
    int64_t leaf_refs;        /* triangle references in leaf lists               */
    int64_t ntri, nsphere, nloose;
    int64_t device_bytes;     /* HBM held by the scene                           */
} rt_scene_info;

typedef struct rt_ray { float o[3]; float d[3]; } rt_ray;                 /* 24 B */
/* prim = index into `prims` (scene-file order), -1 = miss; len and p are the
 * reference's nearestDis and hitPoint (hw4.cpp:855-857).                        */
typedef struct rt_hit { int32_t prim; float len; float p[3]; } rt_hit;    /* 20 B */

/* Camera after Camera::CalcPlaneParameters (fields of hw4.h:144-183 that
 * GetPixelCenter hw4.cpp:268-276 reads).                                        */
typedef struct rt_camera {
    float   center[3];
    float   corners[4][3];    /* planeCorners: topLeft, topRight, bottomLeft, bottomRight */
    float   xaxis[3], yaxis[3];
    int32_t resx, resy;
} rt_camera;

#define RT_RENDER_WHITTED 0   /* GenerateColorImage, hw4.cpp:1081-1166           */
#define RT_RENDER_PHOTON  1   /* PhotonMappingImageSynthesis, final.cpp:1045-1209 */

typedef struct rt_params {
    int32_t mode;               /* RT_RENDER_*                                   */
    int32_t max_depth;          /* literal 10 at hw4.cpp:796; 5 at final.cpp:789 */
    float   ambient_intensity;  /* hw4.cpp:33                                    */
    float   background[3];      /* hw4.cpp:32                                    */
    float   start_nr;           /* airReractivity 1.0 (hw4.cpp:38); 2.4 (final.cpp:1221) */
    int32_t primary_only;       /* 1: primary + one shadow ray per light per hit,
                                   no reflection/refraction (BASELINE config 2)  */
    /* tile sharding across ranks: this process renders the tiles
     * t with t % tile_nranks == tile_rank; other pixels are left untouched.     */
    int32_t tile_size;          /* 0 = no tiling (whole frame)                   */
    int32_t tile_rank, tile_nranks;
    /* photon-mode knobs (parameter.h:18-27, final.cpp:909,916)                  */
    int32_t gather_samples;     /* GLOBAL_ILLUMINATION_SAMPLE 50                 */
    int32_t knn_k;              /* PHOTON_QUERY_COUNT 100                        */
    float   knn_radius_global;  /* QUERY_MAX_DIS 1.0                             */
    float   knn_radius_caustic; /* 0.1                                           */
    int32_t irradiance_max_depth; /* IRRADIANCE_MAX_DEPTH 3                      */
    int32_t aa_taps;            /* KERNEL_SIZE 6 (0 = one centre sample)         */
    int32_t count_work;         /* 1: also run the counting traversal on every pass and fill
                                   rt_stats.closest_work / any_work (measurement only, slow) */
    uint64_t seed;              /* counter-based RNG key (build-defined)         */
    /* distributed ray tracing (RT_RENDER_WHITTED only).  NOT in the reference, which traces one centre sample per
     * pixel with point lights and mirror reflection: these are the build-defined samplers BASELINE's configuration 3
     * names.  All zero = the reference's image, bit for bit.                                        */
    int32_t spp;                /* samples per pixel, jittered inside the pixel; <= 1: one centre sample        */
    float   lens_radius;        /* depth of field: thin-lens radius, 0 = pinhole                                 */
    float   focal_distance;     /* distance along the primary ray of the point in focus                         */
    float   light_radius;       /* soft shadows: every light is a ball of this radius, one sample per shadow ray */
    float   glossy;             /* glossy reflection: reflected direction + glossy * (random point in unit ball) */
    int32_t packed_output;      /* rt_render_dev, RT_RENDER_WHITTED only.  1: the colour of the k-th pixel of this call's
                                   pixel list (rt_render_pixel_list) is written to rgb8[3k] instead of its place in the
                                   frame, so a rank's share of a tiled frame is one contiguous buffer (coalesced
                                   writes, no packing before the gather).  rgb8 then holds 3 x 64 x (number of 8x8
                                   micro-tiles of the call) bytes; hit_ids stay in frame order.  No counterpart in the
                                   reference (it has one process and one frame).                                   */
} rt_params;

/* Work counters (the reference keeps checkTime / totalIntersectionCheck,
 * hw4.cpp:52-54,807,843,986,1022).                                              */
typedef struct rt_stats {
    int64_t rays_primary, rays_shadow, rays_reflect, rays_refract, rays_gather;
    int64_t knn_lookups;
    int64_t passes;             /* wavefront iterations                          */
    double  kernel_ms;          /* device time of the whole render (HIP events on the call's stream) */
    double  trace_closest_ms;   /* sum over passes of the closest-hit kernel's time (HIP events)     */
    double  trace_any_ms;       /* sum over passes of the any-hit (shadow) kernel's time             */
    int64_t trace_closest_launches, trace_any_launches;
    int64_t closest_work[4];    /* rt_trace_work_dev's four counters over all closest-hit passes (count_work) */
    int64_t any_work[4];        /* the same over all any-hit passes                                          */
    /* appended with RT_ABI_VERSION 2 (measurement only) */
    double  knn_ms;             /* RT_RENDER_PHOTON: device time of all radiance-estimate launches (query sort, k_irr_main,
                                   strict redo), HIP events on the call's stream                              */
    int64_t knn_launches;
    double  lane_closest_ms;    /* the part of trace_closest_ms spent in per-lane passes (recursion depth >= RT_OPT_LANE_DEPTH) */
    double  lane_any_ms;        /* the same of trace_any_ms                                                  */
    int64_t lane_closest_launches, lane_any_launches;
    int64_t lane_closest_rays, lane_any_rays;   /* queue entries traced by those passes (inactive slots included) */
    int64_t lane_closest_work[4], lane_any_work[4];   /* count_work: the part of closest_work / any_work of those passes */
    /* appended with RT_ABI_VERSION 3 (measurement only): the persistent path kernel (RT_OPT_PATH_DEPTH), which traces the
     * closest-hit AND the shadow rays of every recursion depth >= that option in one launch.  Its time is NOT part of
     * trace_closest_ms / trace_any_ms / lane_*; its rays ARE part of rays_reflect / rays_refract / rays_shadow, and its
     * work part of closest_work / any_work.                                                                        */
    double  path_ms;            /* device time of that launch (HIP events)                                         */
    int64_t path_launches;      /* 0 or 1 per chunk of the frame                                                    */
    int64_t path_roots;         /* queue entries it started from                                                    */
    int64_t path_closest_rays, path_shadow_rays;   /* rays it traced                                               */
    int64_t path_work[4];       /* count_work: box tests, triangle records of its closest-hit walks; the same of its shadow walks */
} rt_stats;

/* 40-byte photon record, field order of PhotonStruct photon.h:133-141.          */
typedef struct rt_photon {
    float   position[3];
    float   inDir[3];
    float   power[3];
    uint8_t phi, theta;
    int16_t plane;
} rt_photon;

#define RT_PMAP_GLOBAL  0
#define RT_PMAP_CAUSTIC 1

/* ---- library ------------------------------------------------------------- */

/* Probe the HIP runtime; returns the number of gfx950 devices visible (>=1) or
 * a negative error.  ndev_wanted > visible count is an error. (no counterpart)
 * Also resolves rt_photon_irradiance_noalloc_dev's fixed workspace on the FIRST
 * ndev_wanted gfx950 devices (that creates a HIP context and loads the code
 * object there); devices the caller did not ask for are only probed.           */
int rt_init(int ndev_wanted);
const char* rt_last_error(void);
int rt_abi_version(void);

/* ---- scene --------------------------------------------------------------- */

/* Replaces: end of parsing + KDNode::BuildKD / CalcBound per `O` object
 * (hw4.cpp:729-765, 330-490).  Builds the KD-trees on the host by the
 * reference rule, flattens them to SoA and uploads.                            */
int rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
int rt_scene_destroy(rt_scene* s);                 /* replaces hw4.cpp:778-783   */
int rt_scene_info_get(const rt_scene* s, rt_scene_info* info);

/* Traversal algorithm of the trace / render entry points (no counterpart: the reference has one loop).  All four
 * return the reference's hits bit for bit (ids, len, point, occlusion); they differ in the work they do.
 *   RT_TRAVERSAL_STRICT  the reference's visit order, every box and triangle test it performs
 *   RT_TRAVERSAL_FAST    the reference's tree in near-first order + culling of subtrees that cannot hold the winner,
 *                        ties resolved by the reference's visit rank: same hits, fewer tests
 *   RT_TRAVERSAL_LANE    one path per lane instead of one per wave (per-lane record fetches, lanes refilled with
 *                        the launch's next ray): slower on coherent rays, faster on incoherent ones
 *   RT_TRAVERSAL_BVH     (default) a hierarchy of the build's own over the same triangles; a hit counts only if a
 *                        reference leaf holding the triangle is reachable for the ray.  Costs about
 *                        350 B of device memory per tree triangle on top of the reference's tree.
 * The environment variable RT_TRAVERSAL=strict|fast|lane|bvh sets the default of new scenes.
 * RT_OPT_LANE_DEPTH: with RT_TRAVERSAL_FAST or RT_TRAVERSAL_BVH, rt_render traces the passes of recursion depth >=
 * value (secondary rays and their shadow rays) per lane; default 2 (measured crossover on the 1 M-triangle scene,
 * DESIGN 4.1c); a value above max_depth keeps every pass wave-synchronous.                               */
#define RT_OPT_TRAVERSAL    1
#define RT_TRAVERSAL_STRICT 0
#define RT_TRAVERSAL_FAST   1
#define RT_TRAVERSAL_LANE   2
#define RT_TRAVERSAL_BVH    3
#define RT_OPT_LANE_DEPTH   3
/* RT_OPT_FUSED_PRIMARY (default 1): with rt_params.primary_only, RT_TRAVERSAL_FAST or _BVH, one sample per pixel and <= 32
 * lights, rt_render runs the whole frame (camera ray, closest hit, shadow rays, shading, pixel) as one kernel instead
 * of the wavefront passes: same bytes, no intermediate arrays.  0 = always the passes.  RT_FUSED_PRIMARY=0|1.       */
#define RT_OPT_FUSED_PRIMARY 5
/* RT_OPT_RAY_SORT (default 0: measured 3 % slower on BASELINE configuration 3; no counterpart: the reference traces one ray at a time): rt_render hands the rays of its
 * per-lane passes (RT_OPT_LANE_DEPTH) to lanes in the order of a key -- 32^3 cell of the origin, octant of the direction,
 * finer Morton bits -- instead of queue order (one radix sort of (key, index) pairs per pass; rays are not moved and
 * every result lands in the ray's own slot, so frames are the same bytes either way).  RT_RAY_SORT=0|1.                */
#define RT_OPT_RAY_SORT 6
/* RT_OPT_RENDER_OVERLAP (default 1; no counterpart): rt_render with recursion runs the shadow half of every pass (any hit,
 * lights, accumulate) on a stream of the scene's own, beside the next pass's closest-hit traversal; forked from and joined
 * to the caller's stream by events, per-pixel additions in the same order, so frames are the same bytes either way.
 * 0 = every kernel on the caller's stream (also what rt_stats' per-kernel times should be read from: kernels that share
 * the chip time each other).  RT_RENDER_OVERLAP=0|1.                                                                     */
#define RT_OPT_RENDER_OVERLAP 7
/* RT_OPT_PATH_DEPTH (default 3; no counterpart: the reference recurses): with RT_TRAVERSAL_FAST or _BVH, rt_render hands the
 * rays of recursion depth >= value to ONE persistent kernel in which a lane follows a ray's whole subtree of the
 * recursion -- closest hit, shadow rays, shading, reflected and refracted children, depth first -- instead of one
 * pass of kernels per depth with a host round trip between them.  Same hits, same shadow bits, same
 * children; a pixel's contributions are added subtree by subtree instead of depth by depth (float round-off).  The frame
 * stays byte-identical under tiling and chunking.  A value above max_depth, or 0, keeps the per-depth passes.
 * RT_PATH_DEPTH sets the default of new scenes.                                                                      */
#define RT_OPT_PATH_DEPTH 8
/* Algorithm of the radiance estimate (rt_photon_irradiance*, the photon renderer) and of rt_photon_knn_set:
 *   RT_KNN_STRICT  LocatePhoton's array / heap code statement by statement, one query per lane: np.index in the
 *                  reference's order, irradiance summed in that order (bit-identical to the reference).
 *                  rt_photon_knn / rt_photon_knn_dev always run this one.
 *   RT_KNN_FAST    (default, k <= 112) the SET the reference finds, by other means.  Irradiance: 64 neighbouring
 *                  queries share one walk of a second, spatially ordered heap over the same photons, each keeps
 *                  the k+1 smallest dist2 in registers; a short replay on the reference's heap tells which photon
 *                  its list code evicts; rt_photon_knn_set: 16 lanes per query with the reference's
 *                  visits.  Queries whose set depends on the reference's heap history (exact ties),
 *                  and waves whose walk does not end within its budget, are redone strictly.  The irradiance adds the same photons in
 *                  another order: equal up to float association (<= 1e-5 relative in the tests).  Above 3 photons
 *                  per query of a call every lane walks the second heap for its own query instead of sharing
 *                  the walk (the neighbourhoods no longer overlap); RT_KNN_WALK=wave|lane forces either.
 * RT_KNN=strict|fast sets the default of new scenes.                                                          */
#define RT_OPT_KNN          2
#define RT_KNN_STRICT       0
#define RT_KNN_FAST         1
/* RT_OPT_KNN_STATS (measurement only; no counterpart): 1 = zero the work counters of the RT_KNN_FAST irradiance kernels
 * and count from now on, 0 = stop.  rt_scene_knn_stats reads them (after a device synchronise):
 *   [0] phase-1 record fetches, all lanes   [1] phase-2 node visits, per wave   [2] photons inserted into some list
 *   [3] phase-3 node visits, per wave       [4] photons that were a member for some lane
 *   [5] queries handed to the strict kernel [6] waves   [7] sum of wave times in 10 ns ticks
 *   [8] longest wave time << 24 | its phase-2 visits   [9] waves > 1 ms   [10] waves > 10 ms   [11] waves that gave up
 *   [12] phase-2/3 node visits summed over the lanes that were inside the subtree (the per-lane algorithmic count) */
#define RT_OPT_KNN_STATS    4
int rt_scene_set_option(rt_scene* s, int option, int value);
int rt_scene_knn_stats(rt_scene* s, int64_t out16[16]);


/* Host-only view of what rt_scene_create builds and uploads (no GPU needed):
 * the reference-rule KD-trees (hw4.cpp:330-490) flattened in traversal order.
 * For checking the builder against the reference's own tree.  which:
 *   0 node lo/hi  float[nnode][6]      1 node skip   int32[nnode]
 *   2 node meta   int32[nnode] (depth | leaf<<8)
 *   3 leaf first  int32[nnode]         4 leaf count  int32[nnode]
 *   5 leaf prim   int32[nleafref]      6 loose prim  int32[nloose]
 *   7 leaf tri    float[nleafref][9] (v0, edge1, edge2)
 * rt_host_flat_get copies array `which` into dst (capacity cap elements of the
 * row type) and returns the row count, or a negative error.                    */
typedef struct rt_host_flat rt_host_flat;
int rt_host_flatten(const rt_scene_desc* desc, rt_host_flat** out, rt_scene_info* info);
int64_t rt_host_flat_get(const rt_host_flat* f, int which, void* dst, int64_t cap);
void rt_host_flat_free(rt_host_flat* f);
/* Host-only: PhotonMap::balance (photon.cpp:133-254) exactly as rt_photonmap_build
 * runs it before uploading; photons[1..n] are permuted into heap order in place. */
int rt_host_photon_balance(rt_photon* photons, int64_t n);

/* ---- traversal ----------------------------------------------------------- */

/* Replaces the closest-hit loop of SimpleMeshRayTrace (hw4.cpp:808-871 with
 * KDNode::HitTest :281-328, Triangle::IntersectionTest :171-231,
 * Sphere::IntersectionTest :73-116; brute-force form final.cpp:802-816).       */
int rt_trace_closest(rt_scene* s, const rt_ray* rays, int64_t n, rt_hit* out);
int rt_trace_closest_dev(rt_scene* s, const rt_ray* d_rays, int64_t n,
                         rt_hit* d_out, void* hip_stream);

/* Replaces IsInShadow's traversal (hw4.cpp:987-1046; final.cpp:997-1012).
 * `rays` are the already biased shadow rays (origin = point + normal*bias,
 * dir = normalize(light - origin)).  tmax == NULL: HW4 semantics, any hit with
 * len > tolerance occludes, also beyond the light (hw4.cpp:964,1031).
 * tmax != NULL: Final semantics, len < tmax[i] and primitives whose material
 * has refract > 0.1 are skipped (final.cpp:999,1007).                          */
int rt_trace_any(rt_scene* s, const rt_ray* rays, const float* tmax, int64_t n,
                 uint8_t* occluded);
int rt_trace_any_dev(rt_scene* s, const rt_ray* d_rays, const float* d_tmax,
                     int64_t n, uint8_t* d_occluded, void* hip_stream);

/* Measurement only (no counterpart; the reference keeps `totalIntersectionCheck`/`checkTime`,
 * hw4.cpp:52-54,843,1022): runs the traversal of rt_trace_closest (any == 0) or rt_trace_any (any != 0)
 * in the scene's current traversal mode with work counters instead of results.  counts[0] box tests,
 * counts[1] triangle records read by a ray inside a leaf (screened or exactly tested), counts[2] exact
 * reference-sequence primitive tests, counts[3] record fetches summed over rays.  d_rays is a device
 * pointer; the call synchronises the stream.                                                          */
int rt_trace_work_dev(rt_scene* s, const rt_ray* d_rays, const float* d_tmax, int64_t n, int any,
                      int64_t counts[4], void* hip_stream);

/* ---- photon map ---------------------------------------------------------- */

/* Replaces handing a balanced PhotonMap to the renderer (after
 * PhotonMap::balance + ScalePhotonPower, final.cpp:1125-1129).  `heap_order`
 * is the reference's 1-indexed array: heap_order[0] is unused, photons are
 * heap_order[1..n] with `plane` set by balance_segment (photon.cpp:197).       */
int rt_photonmap_upload(rt_scene* s, int which, const rt_photon* heap_order, int64_t n);

/* Replaces PhotonMap::balance (photon.cpp:133-254) for callers that hold an
 * unbalanced array photons[1..n] (as AddPhoton leaves it, photon.cpp:835-885):
 * balances exactly as the reference does (on the host, or from 65536 photons up
 * by the device kernels of rt_photonmap_build_dev: the same bytes), writes the heap
 * order (scaled, see below) back into `photons` and uploads it.                 */
int rt_photonmap_build(rt_scene* s, int which, rt_photon* photons, int64_t n, float power_scale);
/* power_scale: ScalePhotonPower(power_scale, 0, n) after the balance as final.cpp:1126 does it, i.e.
 * heap slots 0..n-1 are scaled and slot n keeps its power (photon.cpp:826-833); 1.0f = leave.   */

/* The same build with the photons RESIDENT ON THE DEVICE (the output of rt_photonmap_emit_dev, or of the all-gather
 * that assembles a map): PhotonMap::balance / balance_segment / median_split (photon.cpp:133-254) replayed by kernels --
 * every swap of the reference's quickselect, so photons that tie on a split axis land where the reference puts them --
 * then ScalePhotonPower and the device image.  d_photons[0..n] (slot 0 unused) comes back in the reference's heap order,
 * byte for byte what rt_photonmap_build leaves in its host array.  Synchronises the stream (one read-back per tree level).
 * rt_photonmap_build itself takes this path through a temporary device copy from 65536 photons up (RT_PHOTON_BUILD=host
 * or =device forces one or the other).
 * Memory and failure: the build holds about 220 B per photon of device memory at its peak (the caller's array, a copy in
 * emission order, the balance's index / key lists, both heaps and the image); the scene's previous map `which` is
 * released BEFORE the build starts so that the two never coexist.  If the call fails (RT_E_NOMEM, a HIP error) the
 * scene is left with NO map `which` (queries return RT_E_STATE until a build or upload succeeds) and the contents of
 * d_photons[0..n] are undefined -- they may already be permuted or scaled; the caller re-creates them to retry.     */
int rt_photonmap_build_dev(rt_scene* s, int which, rt_photon* d_photons, int64_t n, float power_scale, void* hip_stream);

/* One build per node (no counterpart: the reference is one process).  With the map replicated on every GPU, one rank
 * balances and uploads (rt_photonmap_build) and the others copy its DEVICE IMAGE -- both heaps and the sub-heap boxes as
 * the kernels read them -- instead of repeating the host build: rt_photonmap_image_bytes gives its size, rt_photonmap_export_dev
 * copies it into a caller's device buffer (what the rank hands to the RCCL broadcast) together with the photons'
 * bounding box, rt_photonmap_import_dev installs a received image.  export enqueues on the stream; import returns when
 * the copy is done.  raytracer_amd/dist.py: build_photon_maps.                                                       */
int rt_photonmap_image_bytes(rt_scene* s, int which, int64_t* nphotons, int64_t* bytes);
int rt_photonmap_export_dev(rt_scene* s, int which, void* d_dst, int64_t cap_bytes, float box6[6], void* hip_stream);
int rt_photonmap_import_dev(rt_scene* s, int which, const void* d_src, int64_t nphotons, int64_t bytes,
                            const float box6[6], void* hip_stream);

/* Parameters of one emission loop of PhotonMappingImageSynthesis (final.cpp:1053-1123).           */
typedef struct rt_emit_params {
    int32_t  which;              /* RT_PMAP_GLOBAL: PhotonTrace (photon.cpp:445-568);
                                    RT_PMAP_CAUSTIC: PhotonTraceCaustics (:570-710)              */
    int32_t  light;              /* index of the emitting (rect) light, final.cpp:1053            */
    float    power[3];           /* photon colour at the light: 3,3,3 (:1078) / .2,.2,.2 (:1115)  */
    float    start_nr;           /* 2.40, final.cpp:1080                                          */
    int32_t  max_depth;          /* PHOTON_TRACE_MAX_DEPTH 6, parameter.h:26                      */
    int32_t  ncaustic_sphere;    /* causticMeshes (final.cpp:754-758): proxy spheres a caustic    */
    float    caustic_sphere[4][4]; /*   photon must be aimed at: centre xyz, radius               */
    float    caustic_floor_y;    /* -0.3: caustic photons are stored only at y <= this (:697)     */
    uint64_t seed;               /* key of the counter-based RNG                                  */
} rt_emit_params;

/* Replaces the emission loops final.cpp:1057-1119 + PhotonTrace / PhotonTraceCaustics
 * (photon.cpp:445-710) + AddPhoton (:835-885).  The reference draws from one serial engine; here
 * every emission index has its own counter-based random stream, so emission indices
 * [first, first+count) can be traced in parallel and on any rank: the photons they store are
 * returned in emission order in out[0..*nstored) (capacity cap; *nstored may exceed cap, then
 * only the first cap were written).  `plane` is 0; the array is what AddPhoton leaves
 * behind, ready for rt_photonmap_build.                                                         */
int rt_photonmap_emit(rt_scene* s, const rt_emit_params* e, int64_t first, int64_t count,
                      rt_photon* out, int64_t cap, int64_t* nstored);
/* the same into a device buffer on the scene's GPU (what a rank hands to the RCCL all-gather that assembles the
 * map); runs on the caller's stream and synchronises it once to learn *nstored.                    */
int rt_photonmap_emit_dev(rt_scene* s, const rt_emit_params* e, int64_t first, int64_t count,
                          rt_photon* d_out, int64_t cap, int64_t* nstored, void* hip_stream);

/* Replaces PhotonMap::LocatePhoton (photon.cpp:53-131): for each query the k
 * nearest photons within maxDist.  idx: [q][k] heap-array indices (1-based as
 * in the reference's array) in the order the reference's np.index[1..found]
 * holds them, padded with -1; dist2: [q][k] matching squared distances;
 * found: [q]; radius2: [q] the final np.dist2[0] (may be NULL).                */
int rt_photon_knn(rt_scene* s, int which, const float* pos3, int64_t q, int k,
                  float maxDist, int32_t* idx, float* dist2, int32_t* found,
                  float* radius2);
int rt_photon_knn_dev(rt_scene* s, int which, const float* d_pos3, int64_t q, int k,
                      float maxDist, int32_t* d_idx, float* d_dist2,
                      int32_t* d_found, float* d_radius2, void* hip_stream);
/* The same photons as rt_photon_knn (same found, same radius2), listed in no particular order: what the scene's
 * kNN option computes (RT_KNN_FAST: the cooperative kernel).  No counterpart in the reference, whose only
 * consumer of np.index, CalcIrradiance (photon.cpp:418-429), sums over the list.                              */
int rt_photon_knn_set(rt_scene* s, int which, const float* pos3, int64_t q, int k,
                      float maxDist, int32_t* idx, float* dist2, int32_t* found,
                      float* radius2);

/* Replaces PhotonMap::CalcIrradiance in BALANCE_TREE mode (photon.cpp:391-441
 * with GetPhotonDirection :29-36 and the tables :40-50).  rgb: [q][3].         */
int rt_photon_irradiance(rt_scene* s, int which, const float* pos3, const float* nrm3,
                         int64_t q, int k, float maxDist, float* rgb);
int rt_photon_irradiance_dev(rt_scene* s, int which, const float* d_pos3,
                             const float* d_nrm3, int64_t q, int k, float maxDist,
                             float* d_rgb, void* hip_stream);

/* rt_photon_irradiance_dev for callers that cannot allocate device memory: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE.  Same arguments, same kernels,
 * same bits; the per-call workspace (the query sort's keys, indices and storage, the list of queries redone by the
 * strict kernel) is a fixed 16 MiB array inside the library's code object instead of a grow-only allocation of the
 * scene, and the call issues no hipMalloc / hipFree or any allocating routine.  Limits and rules:
 *   - at most RT_NOALLOC_MAX_QUERIES queries per call (RT_E_LIMIT beyond; split the call);
 *   - the array's address is resolved (one hipGetSymbolAddress per device) by rt_init for the first ndev_wanted gfx950
 *     devices and by rt_scene_create for the scene's device, so a caller that created its scene before it begins a graph
 *     capture never resolves under capture; the call itself only looks the address up;
 *   - every GPU has ONE such array, shared by all scenes and streams on it: calls that overlap in time on one GPU
 *     race on it (issue them on one stream, or order them with events).  The array keeps the last call's contents;
 *     the call clears what must start at zero on its stream.                                                     */
#define RT_NOALLOC_MAX_QUERIES 262144
int rt_photon_irradiance_noalloc_dev(rt_scene* s, int which, const float* d_pos3,
                                     const float* d_nrm3, int64_t q, int k, float maxDist,
                                     float* d_rgb, void* hip_stream);

/* ---- frame --------------------------------------------------------------- */

/* Replaces GenerateColorImage (hw4.cpp:1081-1129, up to the quantised image)
 * and PhotonMappingImageSynthesis' pixel loop (final.cpp:1146-1202).
 * rgb8: resx*resy*3 bytes, pixel (x,y) at 3*(x + y*resx) as ColorImage does
 * (imageIO.cpp:58-62).  hit_ids (optional): resx*resy primary-ray prim ids,
 * -1 = miss.  stats optional.                                                  */
int rt_render(rt_scene* s, const rt_camera* cam, const rt_params* params,
              uint8_t* rgb8, int32_t* hit_ids, rt_stats* stats);
int rt_render_dev(rt_scene* s, const rt_camera* cam, const rt_params* params,
                  uint8_t* d_rgb8, int32_t* d_hit_ids, rt_stats* stats,
                  void* hip_stream);

/* The pixel list of a render call: the 8x8 micro-tiles of the tiles this rank owns (tiles in rising number, a tile's
 * micro-tiles row-major), 64 pixels each (row-major inside the micro-tile).  Writes min(*n, cap) entries
 * x + y * resx, or -1 for a slot outside the frame, and the full count to *n.  This is the order of
 * rt_params.packed_output.  Host only; no counterpart in the reference.                                            */
int rt_render_pixel_list(const rt_camera* cam, const rt_params* p, int32_t* out, int64_t cap, int64_t* n);

/* The frame before quantisation: the `color` the pixel loops hold at hw4.cpp:1105-1109 / final.cpp:1158-1176,
 * float[resx*resy][3], pixel (x,y) at 3*(x + y*resx).  For checking radiance against a tolerance; pixels of
 * other ranks' tiles keep the caller's values.                                                          */
int rt_render_radiance(rt_scene* s, const rt_camera* cam, const rt_params* params,
                       float* radiance, rt_stats* stats);

/* Replaces Camera::GetPixelCenter + normalize (hw4.cpp:268-276,1053-1057):
 * primary rays for pixels [first, first+n) in the reference's loop order
 * (index = x*resy + y, hw4.cpp:1096-1098), offset (ox,oy) added to (x,y).      */
int rt_generate_primary_dev(rt_scene* s, const rt_camera* cam, int64_t first,
                            int64_t n, float ox, float oy, rt_ray* d_rays,
                            void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
