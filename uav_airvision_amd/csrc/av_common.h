// av_common.h -- shared declarations of libairvision_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/airvision.h"

#define AV_EXPORT extern "C" __attribute__((visibility("default")))

// ---- error plumbing ---------------------------------------------------------------------------
void av_set_error(const char* fmt, ...);
#define AV_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            av_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return AV_E_HIP;                                                                      \
        }                                                                                         \
    } while (0)
#define AV_LAUNCH_CHECK()                                                                         \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) {                                                                   \
            av_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return AV_E_HIP;                                                                      \
        }                                                                                         \
    } while (0)

// ---- pyramid geometry passed by value to kernels ----------------------------------------------
struct PyrGeom {
    int levels;
    int w[AV_MAX_LEVELS], h[AV_MAX_LEVELS], pitch[AV_MAX_LEVELS];
    int off[AV_MAX_LEVELS];      // byte offset of padded (0,0) of each level (fits int)
};
PyrGeom av_make_geom(const av_pyr_layout& l);

// ---- camera model, by value -------------------------------------------------------------------
struct CamModel {
    double fx, fy, cx, cy, k1, k2, p1, p2;       // equidistant: k1, k2, p1, p2 hold the model's k1 .. k4
    int model;                                    // AV_DISTORTION_RADTAN (0) / AV_DISTORTION_EQUIDISTANT (1)
};

// packed FAST keypoint word: score << 19 | (2^19-1 - raster)
#define AV_KP_RASTER_BITS 19
#define AV_KP_RASTER_MASK ((1u << AV_KP_RASTER_BITS) - 1u)

#ifdef __HIPCC__
// cv::borderInterpolate(p, len, BORDER_REFLECT_101) for |overshoot| < len
__device__ __forceinline__ int av_reflect101(int p, int len)
{
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return p;
}
// the same for any overshoot (cv::borderInterpolate loops): the general LK kernel's windows reach win + 1 pixels past a level
// that may be only win + 1 pixels wide
__device__ __forceinline__ int av_reflect101_any(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do { p = p < 0 ? -p : 2 * len - 2 - p; } while ((unsigned)p >= (unsigned)len);
    return p;
}

// cv2.undistortPoints core: 5 fixed-point iterations of the radtan inverse, then R*[x y 1].
// Expression order follows OpenCV's cvUndistortPointsInternal; fp64, no contraction.
// cv2.fisheye.undistortPoints core (camera_model.py:41-43): Newton's method on theta (1 + k1 theta^2 + ...) = theta_d, at most 10 steps,
// eps 1e-8 (OpenCV 4.x fisheye.cpp, default TermCriteria), then scale = tan(theta) / theta_d and R * [x y 1].  Parity unpinned
// (the CPU checker holds the same restatement; tan from the device math library vs libm: a few ulp).
__device__ __forceinline__ void av_undistort_fisheye(const CamModel& c, const double* R, double u, double v, double& ox, double& oy)
{
    const double pwx = (u - c.cx) / c.fx, pwy = (v - c.cy) / c.fy;
    const double half_pi = 3.1415926535897932384626433832795 / 2.;
    double theta_d = sqrt(pwx * pwx + pwy * pwy);
    theta_d = fmin(fmax(-half_pi, theta_d), half_pi);
    bool converged = false;
    double theta = theta_d, scale = 0.0;
    if (fabs(theta_d) > 1e-8) {
#pragma unroll 1
        for (int j = 0; j < 10; ++j) {
            const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
            const double k0_theta2 = c.k1 * theta2, k1_theta4 = c.k2 * theta4, k2_theta6 = c.p1 * theta6, k3_theta8 = c.p2 * theta8;
            const double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                     (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabs(theta_fix) < 1e-8) { converged = true; break; }
        }
        scale = tan(theta) / theta_d;
    } else {
        converged = true;
    }
    const bool flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
    if (converged && !flipped) {
        const double pux = pwx * scale, puy = pwy * scale;
        const double xx = R[0] * pux + R[1] * puy + R[2];
        const double yy = R[3] * pux + R[4] * puy + R[5];
        const double ww = R[6] * pux + R[7] * puy + R[8];
        ox = xx / ww; oy = yy / ww;
    } else {
        ox = -1000000.0; oy = -1000000.0;
    }
}
// cv2.fisheye.distortPoints (camera_model.py:69-70), alpha = 0
__device__ __forceinline__ void av_distort_fisheye(const CamModel& c, double x, double y, double& ou, double& ov)
{
    const double r2 = x * x + y * y, r = sqrt(r2);
    const double theta = atan(r);
    const double theta2 = theta * theta, theta3 = theta2 * theta, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const double theta_d = theta + c.k1 * theta3 + c.k2 * theta5 + c.p1 * theta7 + c.p2 * theta9;
    const double inv_r = r > 1e-8 ? 1.0 / r : 1;
    const double cdist = r > 1e-8 ? theta_d * inv_r : 1;
    ou = x * cdist * c.fx + c.cx;
    ov = y * cdist * c.fy + c.cy;
}

__device__ __forceinline__ void av_undistort(const CamModel& c, const double* R, double u, double v, double& ox, double& oy)
{
    if (c.model == 1) { av_undistort_fisheye(c, R, u, v, ox, oy); return; }
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    double x = (u - c.cx) * ifx, y = (v - c.cy) * ify;
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
        double r2 = x * x + y * y;
        double icdist = 1. / (1 + ((0 * r2 + c.k2) * r2 + c.k1) * r2);
        if (icdist < 0) { x = (u - c.cx) * ifx; y = (v - c.cy) * ify; break; }
        double deltaX = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
        double deltaY = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    double xx = R[0] * x + R[1] * y + R[2];
    double yy = R[3] * x + R[4] * y + R[5];
    double ww = 1. / (R[6] * x + R[7] * y + R[8]);
    ox = xx * ww;
    oy = yy * ww;
}

// cv2.projectPoints with zero rvec/tvec on (x, y, 1): radtan forward + K.
__device__ __forceinline__ void av_distort(const CamModel& c, double x, double y, double& ou, double& ov)
{
    if (c.model == 1) { av_distort_fisheye(c, x, y, ou, ov); return; }
    double r2 = x * x + y * y, r4 = r2 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cdist = 1 + c.k1 * r2 + c.k2 * r4;
    double xd = x * cdist + c.p1 * a1 + c.p2 * a2;
    double yd = y * cdist + c.p1 * a3 + c.p2 * a1;
    ou = xd * c.fx + c.cx;
    ov = yd * c.fy + c.cy;
}
#endif  // __HIPCC__

// ---- kernel launchers (defined in the .hip files, used by ops_api and the engine) -------------
// pyramid.hip
int av_launch_pyramid(const uint8_t* img0, const uint8_t* img1, int64_t img_stride, int n_streams, int imgs_per_stream,
                      const PyrGeom& g, uint8_t* pyr_base, int64_t stream_stride, int64_t slot_stride, int slot0, int slot1,
                      hipStream_t st, bool write_level0 = true, bool* wrote_level0 = nullptr, const int* index = nullptr);
// write_level0 = false: levels 1.. only, level 0 stays the caller's image (honoured by the fused kernel; *wrote_level0 tells)
// index (device, n_streams ints): image group i reads / writes storage entry index[i] of the image and pyramid arrays

// lk.hip
struct LKParams {
    int win, max_iter;
    double eps2, min_eig;
};
int av_launch_lk(const uint8_t* pyrI, const uint8_t* pyrJ, int64_t stream_stride, int n_set, const PyrGeom& g,
                 const float* prev, float* next, uint8_t* status, const int* count, int cap, int launch_pts,
                 const LKParams& p, hipStream_t st, const int* index = nullptr,
                 const uint8_t* imgI = nullptr, int64_t imgI_stride = 0, const uint8_t* imgJ = nullptr, int64_t imgJ_stride = 0,      // level 0 of I / J from the caller's image (lk.hip: LKArgs)
                 const int* mapI = nullptr, const int* mapJ = nullptr);      // set -> storage entry of the I / J pyramids and images (shared frame store)

// fast.hip
void av_fast_tiles(int w, int h, int* tiles, int* tile_cap);         // tile count of a w x h image, entries per tile list
int av_launch_fast(const uint8_t* img, int64_t img_stride, int img_pitch, int border, const uint8_t* mask, int64_t mask_stride,
                   int n_img, int w, int h, int threshold,
                   uint32_t* kp, int* count, int cap,                               // flat output (ops API) or NULL
                   uint32_t* tile_kp, int* tile_count,                              // per-tile output (front-end engine) or NULL
                   int* overflow, int stat_stride, hipStream_t st, const int* index = nullptr);      // index: image i of the launch is storage entry index[i]
