/*
 * rt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, single thread) of the hot path of
 * clotha87762/RayTracer, used as the checker by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  Nothing in raytracer_amd/ may include,
 * link or call this.  It shares only the POD struct definitions of
 * include/rt_hip.h with the product.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * against outputs of the reference itself compiled in the build container
 * (oracle/_ref, recipe in oracle/Makefile): the monkey 256x256 PPM (byte
 * exact), the KD statistics the reference logs (138 nodes / depth 9 / 133.332),
 * the ID-colour hit-id image, per-function known-answer vectors produced by
 * calling the reference's own HitTest / IntersectionTest / LocatePhoton /
 * CalcIrradiance / balance through oracle/ref_harness_*.cpp.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#include "../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct orc_counters {
    int64_t rays_primary, rays_shadow, rays_reflect, rays_refract;
    int64_t box_tests;      /* KDNode::HitTest calls                        */
    int64_t tri_tests;      /* primitive IntersectionTest calls             */
    int64_t leaf_refs_read; /* leaf list entries walked                     */
    int64_t knn_lookups, knn_node_visits, knn_found;
    double  average_check;  /* `averageCheck`, hw4.cpp:1106                  */
    int64_t node_pops;      /* nodes popped (each is box-tested a second time, hw4.cpp:825) */
    int64_t rays_traced;    /* closest + any traversals                      */
    int64_t rays_gather;    /* TraceIrradiance calls that traced (photon.cpp:714) */
} orc_counters;

/* flattened view of the KD trees, for comparing builders */
typedef struct orc_kd_flat {
    int64_t  nnode;
    float*   lo;        /* [nnode][3] center - extend/2.0f                  */
    float*   hi;        /* [nnode][3]                                       */
    int32_t* left;      /* [nnode] flat index or -1                         */
    int32_t* right;
    int32_t* is_leaf;
    int64_t* leaf_first;/* [nnode] offset into leaf_prims                   */
    int32_t* leaf_count;
    int64_t  nleafref;
    int32_t* leaf_prims;/* prim ids                                         */
    int32_t  nroot;
    int32_t* roots;     /* [nobject] flat index of each object's root       */
} orc_kd_flat;

orc_scene* orc_scene_create(const rt_scene_desc* d);
void orc_scene_destroy(orc_scene* s);
void orc_scene_info(const orc_scene* s, rt_scene_info* info);
void orc_scene_reset_normals(orc_scene* s);
void orc_counters_get(const orc_scene* s, orc_counters* c);
void orc_counters_reset(orc_scene* s);
int  orc_kd_flatten(const orc_scene* s, orc_kd_flat* out);  /* malloc'd; free with orc_kd_flat_free */
void orc_kd_flat_free(orc_kd_flat* f);

/* unit functions */
int orc_box_test(const float center[3], const float extend[3], const float ori[3], const float dir[3]);
int orc_tri_test(const float verts[9], const float ori[3], const float dir[3], float point[3]);
int orc_sphere_test(const float center[3], float radius, const float ori[3], const float dir[3], float point[3]);

void orc_trace_closest(orc_scene* s, const rt_ray* rays, int64_t n, rt_hit* out);
void orc_trace_any(orc_scene* s, const rt_ray* rays, const float* tmax, int64_t n, uint8_t* occ);

void orc_shadow_ray(const float point[3], const float light[3], const float normal[3], float bias,
                    rt_ray* ray, float* tlight);

/* Camera::CalcPlaneParameters, hw4.cpp:242-266 (view is normalised first as
 * the parser does, hw4.cpp:559) */
void orc_camera_setup(const float eye[3], const float view_raw[3], float fov_deg,
                      int resx, int resy, rt_camera* cam);
void orc_primary_ray(const rt_camera* cam, float px, float py, rt_ray* ray);

void orc_render_whitted(orc_scene* s, const rt_camera* cam, const rt_params* p,
                        uint8_t* rgb8, int32_t* hit_ids, float* radiance /* optional [resx*resy][3] */);

/* photon map */
void orc_direction_tables(float* cost, float* sint, float* cosp, float* sinp);
void orc_photon_quantize_dir(const float dir[3], uint8_t* phi, uint8_t* theta, float ndir[3]);
void orc_pmap_balance(rt_photon* photons /* [n+1], 1-based */, int64_t n);
int  orc_pmap_locate(const rt_photon* heap, int64_t n, const float pos[3], int k, float maxDist,
                     int32_t* idx /* [k] */, float* dist2 /* [k] */, float* radius2, int64_t* visits);
void orc_pmap_knn(const rt_photon* heap, int64_t n, const float* pos3, int64_t q, int k, float maxDist,
                  int32_t* idx, float* dist2, int32_t* found, float* radius2, int64_t* visits);
void orc_pmap_irradiance(const rt_photon* heap, int64_t n, const float* pos3, const float* nrm3,
                         int64_t q, int k, float maxDist, float* rgb);

/* photon-mapping renderer (final.cpp:785-1209, photon.cpp:714-820) with the build's counter RNG */
float orc_rng(uint64_t seed, uint64_t a, uint64_t b, uint64_t c);
int  orc_light_samples(const rt_light* l, float* out3, int cap);
void orc_light_quad(const rt_light* l, float tris[2][9]);
void orc_render_photon(orc_scene* s, const rt_camera* cam, const rt_params* p,
                       const rt_photon* gheap, int64_t gn, const rt_photon* cheap, int64_t cn,
                       uint8_t* rgb8, int32_t* hit_ids, float* radiance);
float orc_minstd_next(uint32_t* state);
uint32_t orc_minstd_seed(uint32_t seed);
int64_t orc_photon_trace_serial(orc_scene* s, const rt_emit_params* e, const float* pos3, const float* dir3,
                                int64_t n, uint32_t* lcg_state, rt_photon* out, int64_t cap);
void orc_trace_irradiance(orc_scene* s, const rt_params* p, const rt_photon* gheap, int64_t gn,
                          const float* ori3, const float* dir3, int depth, float nr, int64_t n, float* rgb);
void orc_eye_trace(orc_scene* s, const rt_camera* cam, const rt_params* p, const rt_photon* gheap, int64_t gn,
                   const rt_photon* cheap, int64_t cn, const float* ori3, const float* dir3, int depth, float nr,
                   int64_t n, uint32_t* lcg_state /* NULL: counter RNG */, float* rgb, uint8_t* hit_any);
int64_t orc_emit_photons(orc_scene* s, const rt_emit_params* e, int64_t first, int64_t count,
                         rt_photon* out, int64_t cap);
/* emission direction through the reference's own cosf / sinf (final.cpp:1075) instead of the build-defined rounded double */
void orc_set_emission_cosf(int on);
/* the emission rays of final.cpp:1057-1076 from the reference's serial engine (pinned by tests/golden/emission_golden.npz) */
void orc_emission_rays_serial(orc_scene* s, int light, int64_t n, uint32_t* lcg_state, float* pos3, float* dir3);

#ifdef __cplusplus
}
#endif
#endif
