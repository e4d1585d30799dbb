// ops_api.hip -- error plumbing and the small stateless point operators of the C ABI.
#include <stdarg.h>

#include "av_common.h"

static thread_local char g_err[512] = "";

void av_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

AV_EXPORT const char* av_last_error(void) { return g_err; }
AV_EXPORT const char* av_version(void) { return "airvision-hip 0.1 (gfx950, round 1)"; }
AV_EXPORT int av_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

namespace {
struct RMat { double m[9]; };

// cv2.undistortPoints (camera_model.py:45, feature_publisher.py:57)
__global__ __launch_bounds__(256) void undistort_kernel(const double* pts, int n, CamModel c, RMat R, double* out)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double x, y;
    av_undistort(c, R.m, pts[2 * i], pts[2 * i + 1], x, y);
    out[2 * i] = x; out[2 * i + 1] = y;
}
// cv2.projectPoints with zero pose (camera_model.py:72-74)
__global__ __launch_bounds__(256) void distort_kernel(const double* pts, int n, CamModel c, double* out)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double u, v;
    av_distort(c, pts[2 * i], pts[2 * i + 1], u, v);
    out[2 * i] = u; out[2 * i + 1] = v;
}
}  // namespace

AV_EXPORT int av_undistort_points(const double* pts_dev, int n, const double* intr, const double* dist, const double* R,
                                  double* out_dev, void* stream)
{
    if (!pts_dev || !out_dev || !intr || !dist || n < 0) { av_set_error("av_undistort_points: bad arguments"); return AV_E_INVALID; }
    if (n == 0) return AV_OK;
    CamModel c{intr[0], intr[1], intr[2], intr[3], dist[0], dist[1], dist[2], dist[3]};
    RMat r;
    for (int i = 0; i < 9; ++i) r.m[i] = R ? R[i] : ((i % 4 == 0) ? 1.0 : 0.0);
    hipLaunchKernelGGL(undistort_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts_dev, n, c, r, out_dev);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

AV_EXPORT int av_distort_points(const double* pts_dev, int n, const double* intr, const double* dist, double* out_dev, void* stream)
{
    if (!pts_dev || !out_dev || !intr || !dist || n < 0) { av_set_error("av_distort_points: bad arguments"); return AV_E_INVALID; }
    if (n == 0) return AV_OK;
    CamModel c{intr[0], intr[1], intr[2], intr[3], dist[0], dist[1], dist[2], dist[3]};
    hipLaunchKernelGGL(distort_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts_dev, n, c, out_dev);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

// the same two operators with the distortion model of camera_model.py:41, 69 as an argument
AV_EXPORT int av_undistort_points_model(const double* pts_dev, int n, const double* intr, const double* dist, const double* R, int model,
                                        double* out_dev, void* stream)
{
    if (!pts_dev || !out_dev || !intr || !dist || n < 0) { av_set_error("av_undistort_points_model: bad arguments"); return AV_E_INVALID; }
    if (model != AV_DISTORTION_RADTAN && model != AV_DISTORTION_EQUIDISTANT) { av_set_error("av_undistort_points_model: unknown distortion model %d", model); return AV_E_INVALID; }
    if (n == 0) return AV_OK;
    CamModel c{intr[0], intr[1], intr[2], intr[3], dist[0], dist[1], dist[2], dist[3], model};
    RMat r;
    for (int i = 0; i < 9; ++i) r.m[i] = R ? R[i] : ((i % 4 == 0) ? 1.0 : 0.0);
    hipLaunchKernelGGL(undistort_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts_dev, n, c, r, out_dev);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
AV_EXPORT int av_distort_points_model(const double* pts_dev, int n, const double* intr, const double* dist, int model, double* out_dev, void* stream)
{
    if (!pts_dev || !out_dev || !intr || !dist || n < 0) { av_set_error("av_distort_points_model: bad arguments"); return AV_E_INVALID; }
    if (model != AV_DISTORTION_RADTAN && model != AV_DISTORTION_EQUIDISTANT) { av_set_error("av_distort_points_model: unknown distortion model %d", model); return AV_E_INVALID; }
    if (n == 0) return AV_OK;
    CamModel c{intr[0], intr[1], intr[2], intr[3], dist[0], dist[1], dist[2], dist[3], model};
    hipLaunchKernelGGL(distort_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts_dev, n, c, out_dev);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
