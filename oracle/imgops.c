/*
 * oracle/imgops.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never on the product path).
 *
 * Scalar C restatement of the third-party native calls that UAV-Airvision's image front-end
 * delegates to OpenCV (module `opencv-python`, version UNPINNED in the reference's
 * requirements.txt:2-3; semantics restated here are those of the OpenCV 4.x sources:
 * imgproc/pyramids.cpp, video/lkpyramid.cpp, features2d/fast.cpp + fast_score.cpp,
 * calib3d undistortPoints / projectPoints).  Reference call sites:
 *
 *   cv2.calcOpticalFlowPyrLK   src/image_processing/feature_tracker.py:102-108
 *                              src/image_processing/stereo_matcher.py:64-68, 70-74
 *                              with the parameters of src/config.py:31-44
 *   cv2.FastFeatureDetector    src/image_processing/pipeline.py:23-25 (ctor),
 *                              src/image_processing/feature_initializer.py:52,
 *                              src/image_processing/feature_adder.py:64
 *   cv2.undistortPoints        src/image_processing/camera_model.py:45,
 *                              src/image_processing/feature_publisher.py:57
 *   cv2.projectPoints          src/image_processing/camera_model.py:72-74
 *
 * PARITY STATUS: "parity unpinned" for this file.  OpenCV is not installed in the build
 * container, the reference pins no version and ships no golden vectors for these calls
 * (SURVEY.md section 8c), so this restatement cannot be checked against the real library here.
 *
 * One deliberate, documented choice: OpenCV accumulates the LK window sums (A11,A12,A22,b1,b2)
 * in float, in an order that differs between its scalar and SIMD builds.  Every term of those
 * sums is an integer, so this oracle accumulates them EXACTLY in int64 and rounds once.  That
 * is order-independent, which is what lets a 64-lane GPU reduction be bit-identical to this
 * scalar code.  All later float arithmetic follows OpenCV's expression order, with FP
 * contraction disabled (see oracle/Makefile).
 *
 * A second documented deviation -- the `err` block.  After the iterations of level 0 OpenCV's
 * LKTrackerInvoker runs `if (status[ptidx] && err && level == 0 && (flags & LK_GET_MIN_EIGENVALS) == 0)`:
 * it recomputes the patch difference at the FINAL position and, if that final window leaves the
 * image (`inextPoint.x < -winSize.width || >= cols` ...), CLEARS status[ptidx].  The Python binding
 * always passes an `err` output, so in cv2 a point that converged but whose final window is outside the
 * image reports status 0; this restatement (which computes no err) reports 1.  The difference cannot
 * reach the reference's results: every LK call site gates the returned points itself --
 * feature_tracker.py:111-115 keeps only 0 <= x <= w-1, 0 <= y <= h-1 (a point inside the image has its
 * 15x15 window within `winSize` of the image, so the err block's test passes for it), stereo_matcher.py:82-88
 * keeps 0 <= x < w, 0 <= y < h for the forward pass, and the backward pass's status (`rev_mask`) is never
 * read (stereo_matcher.py:70-80).  So status differs only for points the reference discards anyway.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else       p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* ------------------------------------------------------------------------------------------
 * pyrDown, 8-bit single channel: separable [1 4 6 4 1], integer, (sum + 128) >> 8,
 * BORDER_REFLECT_101, dst size ((w+1)/2, (h+1)/2).  This is what calcOpticalFlowPyrLK runs
 * internally for every level of both images (the reference's own
 * pyramid_builder.py:31-48 is a pass-through).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_pyr_down_u8(const uint8_t* src, int sw, int sh, uint8_t* dst)
{
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    int* hrow = (int*)malloc(sizeof(int) * 5 * (size_t)dw);
    for (int y = 0; y < dh; ++y) {
        for (int k = 0; k < 5; ++k) {
            const uint8_t* s = src + (size_t)reflect101(2 * y - 2 + k, sh) * sw;
            int* hr = hrow + (size_t)k * dw;
            for (int x = 0; x < dw; ++x) {
                int c = 2 * x;
                hr[x] = s[reflect101(c - 2, sw)] + s[reflect101(c + 2, sw)] +
                        4 * (s[reflect101(c - 1, sw)] + s[reflect101(c + 1, sw)]) + 6 * s[reflect101(c, sw)];
            }
        }
        for (int x = 0; x < dw; ++x) {
            int v = hrow[x] + hrow[4 * dw + x] + 4 * (hrow[dw + x] + hrow[3 * dw + x]) + 6 * hrow[2 * dw + x];
            dst[(size_t)y * dw + x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(hrow);
}

/* ------------------------------------------------------------------------------------------
 * Scharr derivative image (lkpyramid.cpp calcScharrDeriv): int16 pairs (dx, dy); vertical pass
 * (3,10,3)/( -1,0,1 ), then horizontal; rows and columns just outside the image mirror row/col 1
 * resp. n-2 (i.e. reflect-101).
 * ---------------------------------------------------------------------------------------- */
static void scharr_deriv(const uint8_t* img, int w, int h, int16_t* d)
{
    int* t0 = (int*)malloc(sizeof(int) * (size_t)(w + 2));
    int* t1 = (int*)malloc(sizeof(int) * (size_t)(w + 2));
    for (int y = 0; y < h; ++y) {
        const uint8_t* r0 = img + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * w;
        const uint8_t* r1 = img + (size_t)y * w;
        const uint8_t* r2 = img + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * w;
        for (int x = 0; x < w; ++x) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
        t0[0] = t0[x0 + 1]; t0[w + 1] = t0[x1 + 1];
        t1[0] = t1[x0 + 1]; t1[w + 1] = t1[x1 + 1];
        int16_t* dr = d + (size_t)y * w * 2;
        for (int x = 0; x < w; ++x) {
            dr[2 * x]     = (int16_t)(t0[x + 2] - t0[x]);
            dr[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0); free(t1);
}

/* image sample with the winSize-wide REFLECT_101 border buildOpticalFlowPyramid adds */
static inline int pix(const uint8_t* img, int w, int h, int x, int y)
{
    return img[(size_t)reflect101(y, h) * w + reflect101(x, w)];
}
/* derivative sample: calc() pads the derivative image with BORDER_CONSTANT (zeros) */
static inline int dpix(const int16_t* d, int w, int h, int x, int y, int c)
{
    if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return 0;
    return d[((size_t)y * w + x) * 2 + c];
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* round half to even */
static inline int cv_floor_f(float v) { return (int)floorf(v); }
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

#define LK_MAX_WIN 31

/* debug statistics of the last orc_lk_track call: [0] level passes that built an I patch,
 * [1] Newton iterations, [2] points */
ORC_API long long orc_lk_stats[3];

/* ------------------------------------------------------------------------------------------
 * calcOpticalFlowPyrLK with OPTFLOW_USE_INITIAL_FLOW (always set by the reference,
 * config.py:44), minEigThreshold = 1e-4 (the OpenCV default; the reference does not pass it).
 *   pyrI/pyrJ : nlev level pointers (level 0 = full resolution), tightly packed u8
 *   W/H       : level sizes
 *   prev      : [n][2] float32, next: [n][2] float32 in = initial guess, out = result
 *   status    : [n] u8
 *   eps       : criteria epsilon as passed by the caller (0.01); squared internally like calc()
 * ---------------------------------------------------------------------------------------- */
/* diagnostic tap (analysis scripts only): iterations run per point and level of the last orc_lk_track call, [n][8]; NULL = off */
static int* orc_lk_iter_tap = 0;
ORC_API void orc_lk_set_iter_tap(int* buf) { orc_lk_iter_tap = buf; }

ORC_API void orc_lk_track(int nlev, const uint8_t* const* pyrI, const uint8_t* const* pyrJ,
                          const int* W, const int* H,
                          const float* prev, float* next, uint8_t* status, int n,
                          int win, int max_iter, double eps, double min_eig_thr)
{
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const double eps2 = eps * eps;
    const float halfWin = (win - 1) * 0.5f;
    if (win > LK_MAX_WIN) return;

    for (int i = 0; i < n; ++i) status[i] = 1;
    orc_lk_stats[0] = orc_lk_stats[1] = 0; orc_lk_stats[2] = n;

    short Iw[LK_MAX_WIN * LK_MAX_WIN], Ix[LK_MAX_WIN * LK_MAX_WIN], Iy[LK_MAX_WIN * LK_MAX_WIN];

    for (int level = nlev - 1; level >= 0; --level) {
        const int w = W[level], h = H[level];
        const uint8_t* I = pyrI[level];
        const uint8_t* J = pyrJ[level];
        int16_t* deriv = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)w * h);
        scharr_deriv(I, w, h, deriv);

        for (int p = 0; p < n; ++p) {
            const float scale = (float)(1. / (1 << level));
            float prevx = prev[2 * p] * scale, prevy = prev[2 * p + 1] * scale;
            float nextx, nexty;
            if (level == nlev - 1) { nextx = next[2 * p] * scale; nexty = next[2 * p + 1] * scale; }
            else                   { nextx = next[2 * p] * 2.f;   nexty = next[2 * p + 1] * 2.f; }
            next[2 * p] = nextx; next[2 * p + 1] = nexty;

            prevx -= halfWin; prevy -= halfWin;
            int ipx = cv_floor_f(prevx), ipy = cv_floor_f(prevy);
            if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
                if (level == 0) status[p] = 0;
                continue;
            }
            float a = prevx - ipx, b = prevy - ipy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

            orc_lk_stats[0]++;
            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < win; ++y)
                for (int x = 0; x < win; ++x) {
                    int X = ipx + x, Y = ipy + y;
                    int ival = DESCALE(pix(I, w, h, X, Y) * iw00 + pix(I, w, h, X + 1, Y) * iw01 +
                                       pix(I, w, h, X, Y + 1) * iw10 + pix(I, w, h, X + 1, Y + 1) * iw11, W_BITS - 5);
                    int ixv = DESCALE(dpix(deriv, w, h, X, Y, 0) * iw00 + dpix(deriv, w, h, X + 1, Y, 0) * iw01 +
                                      dpix(deriv, w, h, X, Y + 1, 0) * iw10 + dpix(deriv, w, h, X + 1, Y + 1, 0) * iw11, W_BITS);
                    int iyv = DESCALE(dpix(deriv, w, h, X, Y, 1) * iw00 + dpix(deriv, w, h, X + 1, Y, 1) * iw01 +
                                      dpix(deriv, w, h, X, Y + 1, 1) * iw10 + dpix(deriv, w, h, X + 1, Y + 1, 1) * iw11, W_BITS);
                    Iw[y * win + x] = (short)ival; Ix[y * win + x] = (short)ixv; Iy[y * win + x] = (short)iyv;
                    sA11 += (int64_t)ixv * ixv; sA12 += (int64_t)ixv * iyv; sA22 += (int64_t)iyv * iyv;
                }
            /* exact integer sum, one rounding (see file header) */
            float A11 = (float)((double)sA11 * (double)FLT_SCALE);
            float A12 = (float)((double)sA12 * (double)FLT_SCALE);
            float A22 = (float)((double)sA22 * (double)FLT_SCALE);

            float D = A11 * A22 - A12 * A12;
            float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
            if ((double)minEig < min_eig_thr || D < 1.1920928955078125e-7f /* FLT_EPSILON */) {
                if (level == 0) status[p] = 0;
                continue;
            }
            D = 1.f / D;

            nextx -= halfWin; nexty -= halfWin;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_iter; ++j) {
                int inx = cv_floor_f(nextx), iny = cv_floor_f(nexty);
                if (inx < -win || inx >= w || iny < -win || iny >= h) {
                    if (level == 0) status[p] = 0;
                    break;
                }
                a = nextx - inx; b = nexty - iny;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                orc_lk_stats[1]++;
                if (orc_lk_iter_tap && level < 8) orc_lk_iter_tap[8 * p + level] = j + 1;
                int64_t sb1 = 0, sb2 = 0;
                for (int y = 0; y < win; ++y)
                    for (int x = 0; x < win; ++x) {
                        int X = inx + x, Y = iny + y;
                        int diff = DESCALE(pix(J, w, h, X, Y) * iw00 + pix(J, w, h, X + 1, Y) * iw01 +
                                           pix(J, w, h, X, Y + 1) * iw10 + pix(J, w, h, X + 1, Y + 1) * iw11, W_BITS - 5) -
                                   Iw[y * win + x];
                        sb1 += (int64_t)diff * Ix[y * win + x];
                        sb2 += (int64_t)diff * Iy[y * win + x];
                    }
                float b1 = (float)((double)sb1 * (double)FLT_SCALE);
                float b2 = (float)((double)sb2 * (double)FLT_SCALE);
                float dx = (A12 * b2 - A22 * b1) * D;
                float dy = (A12 * b1 - A11 * b2) * D;
                nextx += dx; nexty += dy;
                next[2 * p] = nextx + halfWin; next[2 * p + 1] = nexty + halfWin;
                if ((double)dx * dx + (double)dy * dy <= eps2) break;
                if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                    next[2 * p] -= dx * 0.5f; next[2 * p + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
        }
        free(deriv);
    }
}

/* ------------------------------------------------------------------------------------------
 * FAST-9/16 (features2d/fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>), with non-max
 * suppression, 3-pixel border, raster-order output; then KeyPointsFilter::runByPixelsMask.
 * Returns the number of keypoints found (may exceed cap; only the first cap are written).
 * ---------------------------------------------------------------------------------------- */
static const int FAST_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int FAST_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int fast_is_corner(const uint8_t* p, const int* off, int t)
{
    int v = p[0];
    int bright = 0, dark = 0; /* 16-bit circular masks */
    for (int k = 0; k < 16; ++k) {
        int x = p[off[k]];
        if (x > v + t) bright |= 1 << k;
        if (x < v - t) dark |= 1 << k;
    }
    for (int pass = 0; pass < 2; ++pass) {
        unsigned m = pass ? dark : bright;
        m |= m << 16;
        for (int s = 0; s < 16; ++s)
            if (((m >> s) & 0x1FF) == 0x1FF) return 1;
    }
    return 0;
}

static int fast_corner_score(const uint8_t* p, const int* off, int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int d[25];
    int v = p[0];
    for (int k = 0; k < N; ++k) d[k] = v - p[off[k & 15]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int q = 4; q <= 8; ++q) if (d[k + q] < a) a = d[k + q];
        int m0 = a < d[k] ? a : d[k];
        int m9 = a < d[k + 9] ? a : d[k + 9];
        if (m0 > a0) a0 = m0;
        if (m9 > a0) a0 = m9;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int q = 3; q <= 4; ++q) if (d[k + q] > b) b = d[k + q];
        if (b >= b0) continue;
        for (int q = 5; q <= 8; ++q) if (d[k + q] > b) b = d[k + q];
        int m0 = b > d[k] ? b : d[k];
        int m9 = b > d[k + 9] ? b : d[k + 9];
        if (m0 < b0) b0 = m0;
        if (m9 < b0) b0 = m9;
    }
    return -b0 - 1;
}

ORC_API int orc_fast9_detect(const uint8_t* img, int w, int h, int threshold, const uint8_t* mask,
                             int cap, int* xs, int* ys, int* scores)
{
    int off[16];
    for (int k = 0; k < 16; ++k) off[k] = FAST_DY[k] * w + FAST_DX[k];
    int* sc = (int*)calloc((size_t)w * h, sizeof(int));
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            const uint8_t* p = img + (size_t)y * w + x;
            if (fast_is_corner(p, off, threshold)) sc[(size_t)y * w + x] = fast_corner_score(p, off, threshold);
        }
    int cnt = 0;
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            int s = sc[(size_t)y * w + x];
            if (s == 0) continue;
            const int* c = sc + (size_t)y * w + x;
            if (s > c[-1] && s > c[1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] &&
                s > c[w - 1] && s > c[w] && s > c[w + 1]) {
                if (mask && mask[(size_t)y * w + x] == 0) continue;
                if (cnt < cap) { xs[cnt] = x; ys[cnt] = y; scores[cnt] = s; }
                ++cnt;
            }
        }
    free(sc);
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * cv2.undistortPoints(src, K, D, None, R, P=I): pinhole + radtan (k1,k2,p1,p2), the default
 * TermCriteria(MAX_ITER, 5, 0.01) => exactly 5 fixed-point iterations, then R*[x y 1], all in
 * double.  kd = [fx fy cx cy], dist = [k1 k2 p1 p2], R row-major 3x3.  The float32-vs-float64
 * output dtype rule is applied by the caller (it depends on the numpy dtype of the input).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_undistort_points(const double* pts, int n, const double* kd, const double* dist,
                                  const double* R, double* out)
{
    const double fx = kd[0], fy = kd[1], cx = kd[2], cy = kd[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
    for (int i = 0; i < n; ++i) {
        double u = pts[2 * i], v = pts[2 * i + 1];
        double x = (u - cx) * ifx, y = (v - cy) * ify;
        double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((0 * r2 + k2) * r2 + k1) * r2);
            if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
            double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        double xx = R[0] * x + R[1] * y + R[2];
        double yy = R[3] * x + R[4] * y + R[5];
        double ww = 1. / (R[6] * x + R[7] * y + R[8]);
        out[2 * i] = xx * ww;
        out[2 * i + 1] = yy * ww;
    }
}

/* cv2.projectPoints(convertPointsToHomogeneous(pts), rvec=0, tvec=0, K, D): radtan forward + K */
ORC_API void orc_distort_points(const double* pts, int n, const double* kd, const double* dist, double* out)
{
    const double fx = kd[0], fy = kd[1], cx = kd[2], cy = kd[3];
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
    for (int i = 0; i < n; ++i) {
        double x = pts[2 * i], y = pts[2 * i + 1];
        double r2 = x * x + y * y, r4 = r2 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + k1 * r2 + k2 * r4;
        double xd = x * cdist + p1 * a1 + p2 * a2;
        double yd = y * cdist + p1 * a3 + p2 * a1;
        out[2 * i] = xd * fx + cx;
        out[2 * i + 1] = yd * fy + cy;
    }
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * The 'equidistant' (fisheye, Kannala-Brandt) model of camera_model.py:41-42, 69-70 / feature_publisher.py:53-54, 82-83:
 * cv2.fisheye.undistortPoints(pts, K, D, R, P = identity) and cv2.fisheye.distortPoints(pts, K, D).
 * PARITY UNPINNED: cv2 cannot be installed here and the reference's tests hold no vector for this model (EuRoC is radtan,
 * config.py:98,117); restated from OpenCV 4.x modules/calib3d/src/fisheye.cpp (cv::fisheye::undistortPoints with its default
 * TermCriteria(MAX_ITER + EPS, 10, 1e-8), cv::fisheye::distortPoints with alpha = 0), expression order kept.
 * ------------------------------------------------------------------------------------------------------------------------- */
ORC_API void orc_undistort_points_fisheye(const double* pts, int n, const double* kd, const double* dist, const double* R, double* out)
{
    const double fx = kd[0], fy = kd[1], cx = kd[2], cy = kd[3];
    const double k0 = dist[0], k1 = dist[1], k2 = dist[2], k3 = dist[3];
    const double eps = 1e-8, half_pi = 3.1415926535897932384626433832795 / 2.;
    for (int i = 0; i < n; ++i) {
        const double pwx = (pts[2 * i] - cx) / fx, pwy = (pts[2 * i + 1] - cy) / fy;
        double theta_d = sqrt(pwx * pwx + pwy * pwy);
        /* the model is only valid up to 180 degrees of field of view: clip */
        theta_d = fmin(fmax(-half_pi, theta_d), half_pi);
        int converged = 0;
        double theta = theta_d, scale = 0.0;
        if (fabs(theta_d) > eps) {
            for (int j = 0; j < 10; ++j) {               /* Newton's method on theta (1 + k0 theta^2 + ...) = theta_d */
                const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
                const double k0_theta2 = k0 * theta2, k1_theta4 = k1 * theta4, k2_theta6 = k2 * theta6, k3_theta8 = k3 * theta8;
                const double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                         (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
                theta = theta - theta_fix;
                if (fabs(theta_fix) < eps) { converged = 1; break; }
            }
            scale = tan(theta) / theta_d;
        } else {
            converged = 1;
        }
        /* theta must keep its sign: a flip means convergence on the other side of the camera centre */
        const int flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
        if (converged && !flipped) {
            const double pux = pwx * scale, puy = pwy * scale;
            const double xx = R[0] * pux + R[1] * puy + R[2];
            const double yy = R[3] * pux + R[4] * puy + R[5];
            const double ww = R[6] * pux + R[7] * puy + R[8];
            out[2 * i] = xx / ww;
            out[2 * i + 1] = yy / ww;
        } else {
            out[2 * i] = -1000000.0;
            out[2 * i + 1] = -1000000.0;
        }
    }
}

ORC_API void orc_distort_points_fisheye(const double* pts, int n, const double* kd, const double* dist, double* out)
{
    const double fx = kd[0], fy = kd[1], cx = kd[2], cy = kd[3];
    const double k0 = dist[0], k1 = dist[1], k2 = dist[2], k3 = dist[3];
    for (int i = 0; i < n; ++i) {
        const double x = pts[2 * i], y = pts[2 * i + 1];
        const double r2 = x * x + y * y, r = sqrt(r2);
        const double theta = atan(r);
        const double theta2 = theta * theta, theta3 = theta2 * theta, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
        const double theta_d = theta + k0 * theta3 + k1 * theta5 + k2 * theta7 + k3 * theta9;
        const double inv_r = r > 1e-8 ? 1.0 / r : 1;
        const double cdist = r > 1e-8 ? theta_d * inv_r : 1;
        const double xd = x * cdist, yd = y * cdist;
        out[2 * i] = xd * fx + cx;                       /* alpha (skew) = 0 */
        out[2 * i + 1] = yd * fy + cy;
    }
}
