/*
 * rt_host.h -- C ABI of librt_host.so: the host front end that sits ABOVE the
 * hot-path boundary (include/rt_hip.h).  It mirrors the parts of the
 * reference's `main` / Camera / ColorImage that stay on the CPU: camera set-up,
 * the PPM writer and the executionTime_<input> log.  No GPU code, no HIP.
 */
#ifndef RT_HOST_H
#define RT_HOST_H
#include <stdint.h>
#include "rt_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Camera::CalcPlaneParameters, HW4/hw4/hw4.cpp:242-266, after the parser's
 * viewDirection.normalize() (hw4.cpp:559); fovY = fovX/ASPECT_RATIO (:566),
 * up = (0,1,0) (:47), planeDistance = 1 (:25).                                 */
int rth_camera_setup(const float eye[3], const float view_raw[3], float fov_deg,
                     int32_t resx, int32_t resy, rt_camera* cam);

/* ColorImage::outputPPM, HW4/hw4/imageIO.cpp:64-74: "P6 w h 255\n" + raw RGB.   */
int rth_write_ppm(const char* path, const uint8_t* rgb8, int32_t resx, int32_t resy);

/* MeasureRenderTime's log, hw4.cpp:1232-1247 (same six lines, same labels).     */
int rth_write_execution_time(const char* path, int32_t resx, int32_t resy, int32_t ssaa,
                             double seconds, int64_t kd_nodes, int32_t kd_depth, double avg_check);

/* Scene-file front end: the token loops of the reference's `main` (hw4.cpp:538-765; final.cpp:509-735).
 * Keywords E V F R S T L ML M O (HW4) / RectL (Final) followed by numeric tokens read with atof / atoi
 * semantics; unknown tokens are skipped.  HW4: `T` carries 9 + 3 numbers (vertices, normal), `O` closes an object
 * (no `O` at all = one object, hw4.cpp:754-765; meshes after the last `O` are never traced, :768).  Final: `T` has
 * 9 numbers, normal = (v2-v0)^(v1-v0) (final.cpp:608); no objects.  Materials are de-duplicated in order of first
 * use; the default of hw4.cpp:521-526 applies before the first `M`.
 * The parse is multi-threaded (nthreads <= 0: one per core, at most 16); the result does not depend on the
 * thread count.  rth_scene_fill_desc points `desc` INTO the rth_scene: keep it alive until rt_scene_create
 * returns.  rth_scene_save / rth_scene_load: binary cache of the parsed arrays (same machine, no portability
 * promise), for inputs where even a parallel text parse is the start-up cost.                                   */
#define RTH_FLAVOUR_HW4   0
#define RTH_FLAVOUR_FINAL 1
typedef struct rth_scene rth_scene;
int  rth_scene_parse_file(const char* path, int flavour, int nthreads, rth_scene** out);
int  rth_scene_parse_text(const char* text, int64_t len, int flavour, int nthreads, rth_scene** out);
int  rth_scene_fill_desc(const rth_scene* s, rt_scene_desc* desc);
int  rth_scene_view(const rth_scene* s, float eye[3], float view[3], float* fov_deg, int32_t* resx, int32_t* resy);
int  rth_scene_save(const rth_scene* s, const char* path);
int  rth_scene_load(const char* path, rth_scene** out);
void rth_scene_free(rth_scene* s);

/* SSAA box down-sample of GenerateColorImage, hw4.cpp:1131-1159 (ratio 1 = copy). */
int rth_ssaa_downsample(const uint8_t* src, int32_t resx, int32_t resy, int32_t ratio, uint8_t* dst);

#ifdef __cplusplus
}
#endif
#endif
