// lk.hip -- pyramidal iterative Lucas-Kanade tracker for gfx950, one wavefront per point.
//
// Replaces cv2.calcOpticalFlowPyrLK as the reference calls it (winSize 15x15, maxLevel 3,
// criteria (EPS|COUNT, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW, minEigThreshold 1e-4):
//   src/image_processing/feature_tracker.py:102-108   (temporal, prev cam0 -> curr cam0)
//   src/image_processing/stereo_matcher.py:64-68      (stereo forward, cam0 -> cam1)
//   src/image_processing/stereo_matcher.py:70-74      (stereo backward, cam1 -> cam0)
//   parameters: src/config.py:31-44
//
// Semantics follow OpenCV 4.x video/lkpyramid.cpp (LKTrackerInvoker): W_BITS = 14 fixed-point
// bilinear weights, I patch scaled by 32, Scharr derivatives of the FIRST image (zero outside
// the image), 2x2 normal equations in float, <= max_iter Newton steps per level with the
// eps^2 and the oscillation stop, status decided at level 0 only.
//
// MI355X mapping
//   * one 64-lane wavefront owns one point for all pyramid levels (levels are a dependent chain
//     per point, points are independent) -> no inter-workgroup synchronisation, one launch per
//     LK call for every stream of the batch; 4 points per 256-thread workgroup.
//   * the 15x15 window is spread over the lanes (pixel p -> lane p % 64, <= 4 pixels per lane);
//     the I patch and its two derivative patches live in 12 VGPRs per lane for the whole level.
//   * Scharr derivatives are computed on the fly from a 4x4 neighbourhood of the padded image
//     instead of materialising OpenCV's int16x2 derivative image (saves 4 B/pixel/level of HBM
//     writes + reads per image per call).
//   * pyramid levels carry a 16-pixel reflect-101 frame, so no tap is ever clamped.
//   * the window sums are sums of integers; they are accumulated exactly (int32 per lane, int64
//     across lanes with a butterfly) and rounded once, which makes the result independent of the
//     reduction order and bit-identical to the scalar CPU oracle (DESIGN.md "bit-exactness").
// Bound: latency/VALU (gathers hit L1/L2; the padded pyramids of a stream are ~560 KB).
#include "av_common.h"

namespace {

struct LKArgs {
    const uint8_t* pyrI;
    const uint8_t* pyrJ;
    int64_t stream_stride;
    PyrGeom g;
    const float* prev;
    float* next;
    uint8_t* status;
    const int* count;
    int cap;
    int max_iter;
    double eps2, min_eig;
};

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

#define AV_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

template <int WIN>
__global__ __launch_bounds__(256) void lk_track_kernel(LKArgs a)
{
    constexpr int NPIX = WIN * WIN;
    constexpr int PPL = (NPIX + 63) / 64;          // pixels per lane
    constexpr int W_BITS = 14;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.y;
    const int pidx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(a.count[s], a.cap);
    if (pidx >= n) return;                          // wave-uniform

    const uint8_t* PI = a.pyrI + s * a.stream_stride;
    const uint8_t* PJ = a.pyrJ + s * a.stream_stride;
    const size_t pi = (size_t)s * a.cap + pidx;
    const float prevx0 = a.prev[2 * pi], prevy0 = a.prev[2 * pi + 1];
    float curx = a.next[2 * pi], cury = a.next[2 * pi + 1];       // nextPts[ptidx]

    int px[PPL], py[PPL];
    bool act[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        int p = lane + 64 * k;
        act[k] = p < NPIX;
        p = act[k] ? p : 0;
        py[k] = p / WIN;
        px[k] = p - py[k] * WIN;
    }

    const float halfWin = (WIN - 1) * 0.5f;
    bool ok = true;

    for (int level = a.g.levels - 1; level >= 0; --level) {
        const int w = a.g.w[level], h = a.g.h[level], pitch = a.g.pitch[level];
        const uint8_t* I = PI + a.g.off[level] + AV_PYR_BORDER * pitch + AV_PYR_BORDER;
        const uint8_t* J = PJ + a.g.off[level] + AV_PYR_BORDER * pitch + AV_PYR_BORDER;
        const float scale = (float)(1. / (1 << level));
        float pvx = prevx0 * scale, pvy = prevy0 * scale;
        if (level == a.g.levels - 1) { curx = curx * scale; cury = cury * scale; }
        else                         { curx = curx * 2.f;   cury = cury * 2.f; }

        pvx -= halfWin; pvy -= halfWin;
        const int ipx = (int)floorf(pvx), ipy = (int)floorf(pvy);
        if (ipx < -WIN || ipx >= w || ipy < -WIN || ipy >= h) {
            if (level == 0) ok = false;
            continue;
        }
        float fa = pvx - ipx, fb = pvy - ipy;
        int iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
        int iw01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
        int iw10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

        // ---- I patch + Scharr derivative patches, covariance of derivatives -------------------
        int iv[PPL], ixv[PPL], iyv[PPL];
        int a11 = 0, a12 = 0, a22 = 0;                 // <= 4 terms of <= 4080^2 each: fits int32
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            iv[k] = ixv[k] = iyv[k] = 0;
            if (act[k]) {
                const int X = ipx + px[k], Y = ipy + py[k];
                const uint8_t* p = I + (Y - 1) * pitch + (X - 1);
                int nb[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) nb[r][c] = p[r * pitch + c];
                iv[k] = AV_DESCALE(nb[1][1] * iw00 + nb[1][2] * iw01 + nb[2][1] * iw10 + nb[2][2] * iw11, W_BITS - 5);
                int gx[2][2], gy[2][2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int r = 1 + dy, c = 1 + dx;
                        const bool in = (unsigned)(X + dx) < (unsigned)w && (unsigned)(Y + dy) < (unsigned)h;
                        int t0l = (nb[r - 1][c - 1] + nb[r + 1][c - 1]) * 3 + nb[r][c - 1] * 10;
                        int t0r = (nb[r - 1][c + 1] + nb[r + 1][c + 1]) * 3 + nb[r][c + 1] * 10;
                        int t1l = nb[r + 1][c - 1] - nb[r - 1][c - 1];
                        int t1c = nb[r + 1][c] - nb[r - 1][c];
                        int t1r = nb[r + 1][c + 1] - nb[r - 1][c + 1];
                        gx[dy][dx] = in ? (t0r - t0l) : 0;
                        gy[dy][dx] = in ? ((t1r + t1l) * 3 + t1c * 10) : 0;
                    }
                ixv[k] = AV_DESCALE(gx[0][0] * iw00 + gx[0][1] * iw01 + gx[1][0] * iw10 + gx[1][1] * iw11, W_BITS);
                iyv[k] = AV_DESCALE(gy[0][0] * iw00 + gy[0][1] * iw01 + gy[1][0] * iw10 + gy[1][1] * iw11, W_BITS);
                a11 += ixv[k] * ixv[k];
                a12 += ixv[k] * iyv[k];
                a22 += iyv[k] * iyv[k];
            }
        }
        const long long sA11 = wave_sum_i64(a11), sA12 = wave_sum_i64(a12), sA22 = wave_sum_i64(a22);
        const double FLT_SCALE_D = 1.0 / (1 << 20);
        const float A11 = (float)((double)sA11 * FLT_SCALE_D);
        const float A12 = (float)((double)sA12 * FLT_SCALE_D);
        const float A22 = (float)((double)sA22 * FLT_SCALE_D);

        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if ((double)minEig < a.min_eig || D < 1.1920928955078125e-7f) {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.f / D;

        float wx = curx - halfWin, wy = cury - halfWin;       // nextPt -= halfWin
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_iter; ++j) {
            const int inx = (int)floorf(wx), iny = (int)floorf(wy);
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                if (level == 0) ok = false;
                break;
            }
            fa = wx - inx; fb = wy - iny;
            iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
            iw01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
            iw10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int b1 = 0, b2 = 0;                          // <= 4 terms of <= 8160*4080: fits int32
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                if (act[k]) {
                    const uint8_t* p = J + (iny + py[k]) * pitch + (inx + px[k]);
                    int v = p[0] * iw00 + p[1] * iw01 + p[pitch] * iw10 + p[pitch + 1] * iw11;
                    int diff = AV_DESCALE(v, W_BITS - 5) - iv[k];
                    b1 += diff * ixv[k];
                    b2 += diff * iyv[k];
                }
            }
            const long long sb1 = wave_sum_i64(b1), sb2 = wave_sum_i64(b2);
            const float fb1 = (float)((double)sb1 * FLT_SCALE_D);
            const float fb2 = (float)((double)sb2 * FLT_SCALE_D);
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            wx += dx; wy += dy;
            curx = wx + halfWin; cury = wy + halfWin;
            if ((double)dx * dx + (double)dy * dy <= a.eps2) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                curx -= dx * 0.5f; cury -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
    }
    if (lane == 0) {
        a.next[2 * pi] = curx;
        a.next[2 * pi + 1] = cury;
        a.status[pi] = ok ? 1 : 0;
    }
}

}  // namespace

int av_launch_lk(const uint8_t* pyrI, const uint8_t* pyrJ, int64_t stream_stride, int n_set, const PyrGeom& g,
                 const float* prev, float* next, uint8_t* status, const int* count, int cap, int launch_pts,
                 const LKParams& p, hipStream_t st)
{
    if (n_set <= 0 || launch_pts <= 0) return AV_OK;
    if (p.win != 15) {
        av_set_error("av_lk_track: only winSize 15x15 is built (the reference's config.py:35); got %d", p.win);
        return AV_E_INVALID;
    }
    if (p.win + 1 > AV_PYR_BORDER) { av_set_error("av_lk_track: window exceeds pyramid frame"); return AV_E_INVALID; }
    LKArgs a;
    a.pyrI = pyrI; a.pyrJ = pyrJ; a.stream_stride = stream_stride; a.g = g;
    a.prev = prev; a.next = next; a.status = status; a.count = count; a.cap = cap;
    a.max_iter = p.max_iter; a.eps2 = p.eps2; a.min_eig = p.min_eig;
    if (launch_pts > cap) launch_pts = cap;
    dim3 grid((launch_pts + 3) / 4, n_set);
    hipLaunchKernelGGL(lk_track_kernel<15>, grid, dim3(256), 0, st, a);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

AV_EXPORT int av_lk_track(const uint8_t* pyrI_dev, const uint8_t* pyrJ_dev, int64_t pyr_stride, int n_set,
                          int w, int h, int levels,
                          const float* prev_dev, float* next_dev, uint8_t* status_dev, const int32_t* count_dev, int cap,
                          int win, int max_iter, double eps, double min_eig_threshold, void* stream)
{
    av_pyr_layout lay;
    int rc = av_pyramid_layout(w, h, levels, &lay);
    if (rc) return rc;
    if (!pyrI_dev || !pyrJ_dev || !prev_dev || !next_dev || !status_dev || !count_dev || cap <= 0 || n_set < 0 ||
        pyr_stride < lay.bytes) {
        av_set_error("av_lk_track: bad arguments");
        return AV_E_INVALID;
    }
    LKParams p;
    p.win = win;
    p.max_iter = max_iter < 0 ? 0 : (max_iter > 100 ? 100 : max_iter);        // as calc() clamps criteria.maxCount
    double e = eps < 0 ? 0. : (eps > 10. ? 10. : eps);
    p.eps2 = e * e;
    p.min_eig = min_eig_threshold;
    return av_launch_lk(pyrI_dev, pyrJ_dev, pyr_stride, n_set, av_make_geom(lay), prev_dev, next_dev, status_dev,
                        count_dev, cap, cap, p, (hipStream_t)stream);
}
