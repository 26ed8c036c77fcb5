// host_frontend.cpp -- librt_host.so: CPU-side front end above the hot-path
// boundary (see include/rt_host.h).  Mirrors Camera::CalcPlaneParameters
// (HW4/hw4/hw4.cpp:242-266), ColorImage::outputPPM (imageIO.cpp:64-74), the
// executionTime log (hw4.cpp:1232-1247) and the SSAA down-sample (:1131-1159).
#include "../../include/rt_host.h"
#include <cmath>
#include <cstdio>
#include <fstream>

namespace {
struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 scale(V3 a, float d) { return {d * a.x, d * a.y, d * a.z}; }             // algebra3.cpp:330
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 div(V3 a, float d) { float di = (float)(1.0 / (double)d); return {a.x * di, a.y * di, a.z * di}; }  // :345
inline V3 unit(V3 a) { return div(a, length(a)); }
inline void put(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
}

extern "C" int rth_camera_setup(const float eye[3], const float view_raw[3], float fov_deg,
                                int32_t resx, int32_t resy, rt_camera* cam) {
    if (!eye || !view_raw || !cam || resx <= 0 || resy <= 0) return RT_E_INVALID;
    const float planeDistance = 1.0f;
    const V3 up{0, 1.0f, 0};
    const V3 center{eye[0], eye[1], eye[2]};
    const V3 view = unit(V3{view_raw[0], view_raw[1], view_raw[2]});
    const float fovX = fov_deg, fovY = fov_deg / 1.0f;
    const V3 planeCenter = center + scale(div(view, length(view)), planeDistance);
    const V3 X = unit(cross(up, view));
    const V3 Y = unit(cross(X, view));
    const float thetaX = fovX / 2.0f, thetaY = fovY / 2.0f;
    // hw4.cpp:256-257: the X half-angle goes through the float tan, the Y one through double
    const float halfWidth = planeDistance * std::tan(thetaX * (3.14159f / 180.0f));
    const float halfHeight = (float)(planeDistance * std::tan(thetaY * (3.14159 / 180.0f)));
    put(cam->center, center);
    put(cam->corners[0], planeCenter - scale(X, halfWidth) - scale(Y, halfHeight));
    put(cam->corners[1], planeCenter + scale(X, halfWidth) - scale(Y, halfHeight));
    put(cam->corners[2], planeCenter - scale(X, halfWidth) + scale(Y, halfHeight));
    put(cam->corners[3], planeCenter + scale(X, halfWidth) + scale(Y, halfHeight));
    put(cam->xaxis, X); put(cam->yaxis, Y);
    cam->resx = resx; cam->resy = resy;
    return RT_OK;
}

extern "C" int rth_write_ppm(const char* path, const uint8_t* rgb8, int32_t resx, int32_t resy) {
    if (!path || !rgb8 || resx <= 0 || resy <= 0) return RT_E_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return RT_E_INVALID;
    fprintf(f, "P6 %d %d 255\n", resx, resy);
    size_t n = (size_t)3 * (size_t)resx * (size_t)resy;
    size_t w = fwrite(rgb8, 1, n, f);
    fclose(f);
    return w == n ? RT_OK : RT_E_INVALID;
}

extern "C" int rth_write_execution_time(const char* path, int32_t resx, int32_t resy, int32_t ssaa,
                                        double seconds, int64_t kd_nodes, int32_t kd_depth, double avg_check) {
    if (!path) return RT_E_INVALID;
    std::ofstream fout(path);
    if (!fout) return RT_E_INVALID;
    fout << "resolution:" << resx << "X" << resy << std::endl;
    fout << "SSAA ratio:" << ssaa << std::endl;
    fout << "elapse time:" << seconds << " seconds" << std::endl;
    fout << "total kd-tree node count:" << kd_nodes << std::endl;
    fout << "max depth of kd-tree subdivision:" << kd_depth << std::endl;
    fout << "average ray-mesh Intersection check count:" << avg_check << std::endl;
    return RT_OK;
}

extern "C" int rth_ssaa_downsample(const uint8_t* src, int32_t resx, int32_t resy, int32_t ratio, uint8_t* dst) {
    if (!src || !dst || ratio <= 0 || resx <= 0 || resy <= 0) return RT_E_INVALID;
    const int ox = resx / ratio, oy = resy / ratio;
    for (int x = 0; x < ox; ++x)
        for (int y = 0; y < oy; ++y) {
            int acc[3] = {0, 0, 0};
            for (int i = 0; i < ratio; ++i)
                for (int j = 0; j < ratio; ++j) {
                    const int sx = (x * ratio + i) >= resx ? resx - 1 : x * ratio + i;
                    const int sy = (y * ratio + j) >= resy ? resy - 1 : y * ratio + j;
                    const uint8_t* p = src + 3 * ((size_t)sx + (size_t)sy * (size_t)resx);
                    acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
                }
            uint8_t* q = dst + 3 * ((size_t)x + (size_t)y * (size_t)ox);
            for (int c = 0; c < 3; ++c) q[c] = (uint8_t)(acc[c] / (ratio * ratio));
        }
    return RT_OK;
}
