// lk.hip -- pyramidal iterative Lucas-Kanade tracker for gfx950, 16 lanes per point.
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
//   * one DPP row (16 lanes) owns one point for all pyramid levels (levels are a dependent chain per
//     point, points are independent): 4 points per wavefront, 16 per 256-thread workgroup, no
//     inter-workgroup synchronisation, one launch per LK call for every stream of the batch.
//   * lane r of a row owns window row r (15 pixels; lane 15 only feeds the interpolation of row 14).
//     The I patch and its two derivative patches live in 45 VGPRs per lane per level.  The float
//     Newton update of a point is identical in all its lanes, so one VALU instruction serves 4 points
//     (an earlier one-wavefront-per-point version spent half of every iteration on that replicated math).
//   * windows are staged through LDS: per level one coalesced 18x24-byte stage of the I
//     neighbourhood and one 32x32-byte stage of J around the predicted position (aligned dword
//     loads); every Newton iteration then reads 5 dwords of its row from LDS and takes the row below
//     from the neighbouring lane (DPP row_shl).  The J tile is restaged only when the window drifts
//     more than ~6 px.  1152 B of LDS per point.
//   * Scharr derivatives are computed on the fly from the staged neighbourhood (packed 16-bit
//     arithmetic on the even columns, odd pairs by v_perm) instead of materialising OpenCV's int16x2
//     derivative image (saves 4 B/pixel/level of HBM traffic).
//   * pyramid levels carry a 16-pixel reflect-101 frame, so no tap is ever clamped.
//   * the window sums are sums of integers; they are accumulated exactly (int32 per lane; across the
//     16 lanes by a DPP butterfly whose last two steps run on 16-bit halves, recombined in fp64)
//     and rounded once, which makes the result independent of the reduction order and bit-identical
//     to the scalar CPU oracle.
// Bound: VALU issue (PMC: ~1,400 VALU instructions per point pass; the staged tiles come from
// L2/MALL: the padded pyramids of a stream are ~560 KB); HBM sees each pyramid once per frame.
#include <stdlib.h>

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

// LDS traffic of one wave is ordered by issue; this keeps the compiler from moving a read of the
// staged tile above the (other lanes') writes that fill it.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define AV_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

typedef short av_v2s __attribute__((ext_vector_type(2)));
// a.lo*b.lo + a.hi*b.hi + c on packed signed 16-bit halves (v_dot2c_i32_i16, full rate, exact)
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(av_v2s, a), __builtin_bit_cast(av_v2s, b), c, false);
}
// the same with the accumulator in its own register: v_dot2c (what the builtin selects) accumulates in place, so an
// accumulator that has to survive -- the per-pixel seeds, the rounding constants -- costs a v_mov per use; the VOP3P
// form takes it as a third source
__device__ __forceinline__ int dot2_keep(uint32_t a, uint32_t b, int c)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }

constexpr int TILE = 32;          // staged tile: 32 rows x 32 bytes
constexpr int TPITCH = 36;        // LDS row pitch in bytes (9 dwords: spreads rows over banks)
constexpr int TILE_DWORDS = TILE * TPITCH / 4;

// =================================================================================================
// 16 lanes per point: lane r of a DPP row owns window row r (15 pixels); one wavefront tracks 4 points.
// Per Newton iteration the pixel work per lane is 15 x (2 perm + 2 dot2 + shift + sub + 2 mad) while the
// float update of the point (weights, 2x2 solve, stop tests), which is identical for every lane of a point,
// is now shared by four points per instruction instead of one.  Window sums stay exact: int32 per lane
// (15 x 8160 x 4080 < 2^31), split into 16-bit halves for the 16-lane DPP butterfly, recombined in fp64.
// =================================================================================================
typedef unsigned short av_v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(av_v2u, a) + __builtin_bit_cast(av_v2u, b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(av_v2u, a) - __builtin_bit_cast(av_v2u, b)); }
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, unsigned short k) { av_v2u kk = {k, k}; return __builtin_bit_cast(uint32_t, __builtin_bit_cast(av_v2u, a) * kk); }

// exact sum over the 16 lanes of a DPP row of an int32 that may overflow when summed: 16-bit halves
__device__ __forceinline__ double row_sum16_exact(int v)
{
    // |v| <= 15 * 8160 * 4080 < 2^29: the sum over a quad still fits int32, the sum over 16 lanes does not
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    int lo = v & 0xFFFF, hi = v >> 16;
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false); // row_half_mirror
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false);
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, false); // row_mirror
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, false);
    return (double)hi * 65536.0 + (double)lo;
}

template <int WIN>
__global__ __launch_bounds__(256) void lk_track_g16_kernel(LKArgs a)
{
    static_assert(WIN == 15, "lane map is built for the reference's 15x15 window");
    constexpr int W_BITS = 14;
    __shared__ uint32_t tile_all[16][TILE_DWORDS];
    const int g = threadIdx.x >> 4;                 // point slot of this 16-lane group inside the workgroup
    const int r = threadIdx.x & 15;                 // window row owned by this lane (row 15 only feeds row 14)
    uint32_t* tile = tile_all[g];
    const int s = blockIdx.y;
    const int pidx = blockIdx.x * 16 + g;
    const int n = min(a.count[s], a.cap);
    if (pidx >= n) return;                          // uniform per 16-lane group

    const uint8_t* PI = a.pyrI + s * a.stream_stride;
    const uint8_t* PJ = a.pyrJ + s * a.stream_stride;
    const size_t pi = (size_t)s * a.cap + pidx;
    const float prevx0 = a.prev[2 * pi], prevy0 = a.prev[2 * pi + 1];
    float curx = a.next[2 * pi], cury = a.next[2 * pi + 1];
    const float halfWin = (WIN - 1) * 0.5f;
    const bool rowact = r < WIN;
    bool ok = true;
    const double FLT_SCALE_D = 1.0 / (1 << 20);

    for (int level = a.g.levels - 1; level >= 0; --level) {
        const int w = a.g.w[level], h = a.g.h[level], pitch = a.g.pitch[level];
        const uint8_t* I = PI + a.g.off[level] + AV_PYR_BORDER * pitch + AV_PYR_BORDER;
        const uint8_t* J = PJ + a.g.off[level] + AV_PYR_BORDER * pitch + AV_PYR_BORDER;
        const int col_lo = -AV_PYR_BORDER, col_hi = pitch - AV_PYR_BORDER;
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
        uint32_t wtop = pack16(iw00, iw01), wbot = pack16(iw10, iw11);

        // ---- stage the 18 x 24-byte neighbourhood of the I window (rows ipy-1 .., aligned dwords from cs) ---
        const int cs = (ipx - 1) & ~3, io = (ipx - 1) - cs;
        wave_lds_sync();
        {
            uint32_t sv[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int idx = r + 16 * k;                 // 18 rows x 6 dwords = 108
                const int row = idx / 6, dw = idx - row * 6;
                const int c = cs + 4 * dw;
                sv[k] = 0;
                if (idx < 108 && c >= col_lo && c + 4 <= col_hi) sv[k] = *reinterpret_cast<const uint32_t*>(I + __mul24(ipy - 1 + row, pitch) + c);
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int idx = r + 16 * k;
                const int row = idx / 6, dw = idx - row * 6;
                if (idx < 108) tile[__mul24(row, TPITCH / 4) + dw] = sv[k];
            }
        }
        wave_lds_sync();

        // ---- I patch + Scharr patches of window row r: staged rows r..r+3, bytes io..io+17 of each -------------
        int iv[WIN], ixv[WIN], iyv[WIN];
        int a11 = 0, a12 = 0, a22 = 0;
        {
            // staged byte index of window column c is c+1.  E[t][k] = (byte 2k | byte 2k+1 << 16) of staged row r+t,
            // O[t][k] = (byte 2k+1 | byte 2k+2 << 16): window pair (c, c+1) is O[k] for c = 2k and E[k+1] for c = 2k+1.
            uint32_t E[4][9], O12[2][8];
            const int rr = rowact ? r : 14;                   // spare lane 15 recomputes row 14 (dropped at the reductions)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t* rowp = tile + __mul24(rr + t, TPITCH / 4);
                uint32_t d[6], B[5];
#pragma unroll
                for (int k = 0; k < 6; ++k) d[k] = rowp[k];
#pragma unroll
                for (int k = 0; k < 5; ++k) B[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], io);
#pragma unroll
                for (int k = 0; k < 9; ++k) {                // bytes 2k, 2k+1
                    const int b0 = 2 * k;
                    E[t][k] = __builtin_amdgcn_perm(0, B[b0 >> 2], 0x0C000C00u | (uint32_t)(b0 & 3) | ((uint32_t)((b0 & 3) + 1) << 16));
                }
            }
            // (hi half of x | lo half of y << 16): the pair one column to the right of x, given y = the next pair
            auto mid = [](uint32_t x, uint32_t y) -> uint32_t { return __builtin_amdgcn_perm(y, x, 0x05040302u); };
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int k = 0; k < 8; ++k) O12[t][k] = mid(E[1 + t][k], E[1 + t][k + 1]);
            // Scharr on the even window columns (pairs c = 2k, 2k+1); the odd-aligned pairs are byte permutes of those:
            //   gx[c] = t0[s+1] - t0[s-1],  gy[c] = (t1[s+1] + t1[s-1]) * 3 + t1[s] * 10,  s = c+1,
            //   t0 = (row above + row below) * 3 + row * 10,  t1 = row below - row above
            uint32_t gxp[2][WIN + 1], gyp[2][WIN + 1];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                uint32_t t0E[9], t1E[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    t0E[k] = pk_add(pk_mul(pk_add(E[dy][k], E[dy + 2][k]), 3), pk_mul(E[dy + 1][k], 10));
                    t1E[k] = pk_sub(E[dy + 2][k], E[dy][k]);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {                 // window columns (2k, 2k+1): centres O[k], left E[k], right E[k+1]
                    gxp[dy][2 * k] = pk_sub(t0E[k + 1], t0E[k]);
                    gyp[dy][2 * k] = pk_add(pk_mul(pk_add(t1E[k + 1], t1E[k]), 3), pk_mul(mid(t1E[k], t1E[k + 1]), 10));
                }
            }
            // the derivative image is zero outside the image: only windows at the border pay for the masks
            const bool inside = ipx >= 0 && ipx + WIN < w && ipy >= 0 && ipy + WIN < h;
            if (__builtin_amdgcn_ballot_w64(!inside) != 0) {
                __builtin_amdgcn_s_sleep(0);        // a side effect keeps the compiler from if-converting this (rare) block into selects on the common path
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const bool rowin = (unsigned)(ipy + rr + dy) < (unsigned)h;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t m0 = (rowin && (unsigned)(ipx + 2 * k) < (unsigned)w) ? 0xFFFFu : 0u;
                        const uint32_t m1 = (rowin && (unsigned)(ipx + 2 * k + 1) < (unsigned)w) ? 0xFFFF0000u : 0u;
                        gxp[dy][2 * k] &= (m0 | m1); gyp[dy][2 * k] &= (m0 | m1);
                    }
                }
            }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    gxp[dy][2 * k + 1] = mid(gxp[dy][2 * k], gxp[dy][2 * k + 2]);
                    gyp[dy][2 * k + 1] = mid(gyp[dy][2 * k], gyp[dy][2 * k + 2]);
                }
            const int rnd_i = 1 << (W_BITS - 6), rnd_d = 1 << (W_BITS - 1);      // rounding terms, one register each
#pragma unroll
            for (int c = 0; c < WIN; ++c) {
                const int k = c >> 1;
                const uint32_t ptop = (c & 1) == 0 ? O12[0][k] : E[1][k + 1];
                const uint32_t pbot = (c & 1) == 0 ? O12[1][k] : E[2][k + 1];
                const int ival = dot2(ptop, wtop, dot2_keep(pbot, wbot, rnd_i)) >> (W_BITS - 5);
                ixv[c] = dot2(gxp[0][c], wtop, dot2_keep(gxp[1][c], wbot, rnd_d)) >> W_BITS;
                iyv[c] = dot2(gyp[0][c], wtop, dot2_keep(gyp[1][c], wbot, rnd_d)) >> W_BITS;
                // accumulator seed of the iteration's interpolation: ((dot + R) >> 9) - ival == (dot + R - (ival << 9)) >> 9
                iv[c] = (1 << (W_BITS - 6)) - (ival << (W_BITS - 5));
                a11 += __mul24(ixv[c], ixv[c]);
                a12 += __mul24(ixv[c], iyv[c]);
                a22 += __mul24(iyv[c], iyv[c]);
            }
            if (!rowact) { a11 = 0; a12 = 0; a22 = 0; }
        }
        const float A11 = (float)(row_sum16_exact(a11) * FLT_SCALE_D);
        const float A12 = (float)(row_sum16_exact(a12) * FLT_SCALE_D);
        const float A22 = (float)(row_sum16_exact(a22) * FLT_SCALE_D);
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if ((double)minEig < a.min_eig || D < 1.1920928955078125e-7f) {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.f / D;

        float wx = curx - halfWin, wy = cury - halfWin;
        float pdx = 0.f, pdy = 0.f;
        int X0 = 0, Y0 = 0;
        bool staged = false;
        for (int j = 0; j < a.max_iter; ++j) {
            const int inx = (int)floorf(wx), iny = (int)floorf(wy);
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                if (level == 0) ok = false;
                break;
            }
            int dx0 = inx - X0, dy0 = iny - Y0;
            if (!staged || (unsigned)dx0 > 15u || (unsigned)dy0 > 15u) {
                X0 = min(max((inx - 6) & ~3, col_lo), col_hi - TILE);
                Y0 = min(max(iny - 8, -AV_PYR_BORDER), h + AV_PYR_BORDER - TILE);
                wave_lds_sync();
                uint32_t sv[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int idx = r + 16 * k;                // 32 rows x 8 dwords
                    sv[k] = *reinterpret_cast<const uint32_t*>(J + __mul24(Y0 + (idx >> 3), pitch) + X0 + 4 * (idx & 7));
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int idx = r + 16 * k;
                    tile[__mul24(idx >> 3, TPITCH / 4) + (idx & 7)] = sv[k];
                }
                wave_lds_sync();
                staged = true;
                dx0 = inx - X0; dy0 = iny - Y0;
            }
            fa = wx - inx; fb = wy - iny;
            iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
            iw01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
            iw10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            wtop = pack16(iw00, iw01); wbot = pack16(iw10, iw11);

            // this lane's J row (dy0 + r): 16 pixels starting at byte dx0 -> 5 aligned dwords -> 4 byte-aligned dwords
            const int sh = dx0 & 3;
            const uint32_t* rp = tile + __mul24(dy0 + r, TPITCH / 4) + (dx0 >> 2);
            uint32_t d0 = rp[0], d1 = rp[1], d2 = rp[2], d3 = rp[3], d4 = rp[4];
            uint32_t T[5], Bt[5];
            T[0] = __builtin_amdgcn_alignbyte(d1, d0, sh); T[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
            T[2] = __builtin_amdgcn_alignbyte(d3, d2, sh); T[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
            T[4] = 0;
            // the row below comes from the next lane of the DPP row (row_shl:1)
#pragma unroll
            for (int k = 0; k < 4; ++k) Bt[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)T[k], 0x101, 0xF, 0xF, false);
            Bt[4] = 0;
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int c = 0; c < WIN; ++c) {
                const int q = c >> 2, o = c & 3;
                const uint32_t sel = 0x0C000C00u | (uint32_t)o | ((uint32_t)(o + 1) << 16);     // o+1 == 4 selects byte 0 of the next dword
                const uint32_t tp = __builtin_amdgcn_perm(T[q + 1], T[q], sel);
                const uint32_t bp = __builtin_amdgcn_perm(Bt[q + 1], Bt[q], sel);
                const int diff = dot2(tp, wtop, dot2_keep(bp, wbot, iv[c])) >> (W_BITS - 5);
                b1 += __mul24(diff, ixv[c]);
                b2 += __mul24(diff, iyv[c]);
            }
            if (!rowact) { b1 = 0; b2 = 0; }
            const float fb1 = (float)(row_sum16_exact(b1) * FLT_SCALE_D);
            const float fb2 = (float)(row_sum16_exact(b2) * FLT_SCALE_D);
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
    if (r == 0) {
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
    dim3 grid((launch_pts + 15) / 16, n_set);
    hipLaunchKernelGGL(lk_track_g16_kernel<15>, grid, dim3(256), 0, st, a);
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
