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

/* SSAA box down-sample of GenerateColorImage, hw4.cpp:1131-1159 (ratio 1 = copy). */
int rth_ssaa_downsample(const uint8_t* src, int32_t resx, int32_t resy, int32_t ratio, uint8_t* dst);

#ifdef __cplusplus
}
#endif
#endif
