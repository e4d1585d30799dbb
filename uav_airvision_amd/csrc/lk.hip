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
#include <math.h>
#include <stdio.h>
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
    const int* index;           // optional [n_set][cap]: point j of set s is index[s*cap + j] (a subset of the point arrays)
    int cap;
    int max_iter;
    double eps2;
    // float forms of the two fp64 thresholds (av_launch_lk): minEig < min_eig (double) <=> minEig < min_eig_up, the smallest float
    // >= min_eig; the step-length test is decided in float outside (eps_lo, eps_hi) = eps2 (1 -+ 2^-18) and in fp64 inside
    float min_eig_up, eps_lo, eps_hi;
    double min_eig;             // the general kernel's fp64 form
    int n_set, gx;              // XCD-aware 1-D launch (gx > 0): gx workgroups per point set, see lk_track_g16_body
    // Level 0 straight from the caller's image (w x h, tightly packed rows) instead of a padded copy in the pyramid: non-null =
    // the I (resp. J) side reads level 0 at img + set * img_stride, with BORDER_REFLECT_101 indexing for the few windows that
    // reach over the image border (what the padded level's frame holds).  Levels >= 1 always come from the padded pyramid.
    const uint8_t* imgI; const uint8_t* imgJ;
    int64_t imgI_stride, imgJ_stride;
    // Optional storage maps (the front-end's shared frame store, frontend.hip): point set s reads the pyramids / level-0 images of
    // storage entry mapI[s] (I side) and mapJ[s] (J side) instead of entry s; a negative entry = the set has no frame in this step
    // and tracks nothing.  Null = identity.
    const int* mapI; const int* mapJ;
    // diagnostic (AV_LK_PROF=1, lk_track_g16_prof_kernel): per-phase s_memtime sums over every wavefront of the launch
    unsigned long long* prof;
};

// phases of a wavefront's life in the 16-lane kernel, as the stamps cut it
enum { LKP_HEAD = 0, LKP_ISTAGE, LKP_SETUP, LKP_JSTAGE, LKP_ITER, LKP_N, LKP_WAVES = LKP_N, LKP_ITERS, LKP_RESTAGES, LKP_LEVELS, LKP_SLOTS };

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

constexpr int TILE_ROWS = 24;     // staged J tile: 24 rows x 32 bytes (window 16 rows + 4 px of drift either way)
constexpr int TILE_COLS = 32;
constexpr int TPITCH = 36;        // LDS row pitch in bytes (9 dwords: the 16 rows of a point fall on 16 different banks)
constexpr int TILE_DWORDS = 240;  // >= 24 * 9, and == 16 mod 32: the two points of a 32-lane LDS access group use disjoint banks

// =================================================================================================
// 16 lanes per point: lane r of a DPP row owns window row r (15 pixels + the column right of them); lane 15 owns
// row 15, which only feeds the bilinear interpolation of row 14.  A lane builds ONLY its own row of the I patch
// and of the two Scharr derivative patches; the row below, which every bilinear sample needs, is read out of the
// next lane's registers by the DPP operand of v_dot2c_i32_i16 (row_shl:1) -- no second copy of the row, no
// v_mov_dpp, no second byte permute per pixel.  The float Newton update of a point is identical in all its
// lanes, so one VALU instruction serves 4 points.  Window sums stay exact: int32 per lane
// (15 x 8160 x 4080 < 2^31), split into 16-bit halves for the 16-lane DPP butterfly, recombined in fp64.
// =================================================================================================
typedef unsigned short av_v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(av_v2u, a) + __builtin_bit_cast(av_v2u, b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(av_v2u, a) - __builtin_bit_cast(av_v2u, b)); }
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, unsigned short k) { av_v2u kk = {k, k}; return __builtin_bit_cast(uint32_t, __builtin_bit_cast(av_v2u, a) * kk); }

// exact sum S over the 16 lanes of a DPP row of an int32 that may overflow when summed (16-bit halves), returned as
// (float)(S * 2^-20) rounded ONCE -- the value of (float)((double)S * 2^-20), without the fp64 instructions: the two half sums are
// exact as floats (|hi| < 2^17, lo < 2^18), so is lo * 2^-20, and the fused multiply-add rounds the exact hi * 2^-4 + lo * 2^-20 =
// S * 2^-20 once.  (fp64 form: 2 x v_cvt_f64_i32, v_ldexp_f64, v_add_f64, v_ldexp_f64, v_cvt_f32_f64 per sum -- six half-rate
// instructions against two half-rate conversions, one multiply and one fused multiply-add here; five sums per Newton iteration and
// level set-up: profiles/r05/valu_issue_microbench.json.)
__device__ __forceinline__ float row_sum16_scaled(int v)
{
    // |v| <= 15 * 8160 * 4080 < 2^29: the sum over a quad still fits int32, the sum over 16 lanes does not
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    int lo = v & 0xFFFF, hi = v >> 16;
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false); // row_half_mirror
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false);
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, false); // row_mirror
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, false);
    return __builtin_fmaf((float)hi, 0x1p-4f, (float)lo * 0x1p-20f);
}

// Bilinear sample of five packed pixel (or derivative) pairs x_i = (v[c], v[c+1]) of this lane's row with the same
// pairs of the row below (next lane, DPP row_shl:1):  o_i = (x_i . wtop + below(x_i) . wbot + rnd) >> SH.
// One asm statement, for two reasons: the compiler does not fold a DPP move into v_dot2c (it has a tied accumulator),
// and its hazard recogniser does not see into inline asm -- a DPP operand written by a VALU instruction needs 2 wait
// states, guaranteed here by issuing the five plain dot products first.
template <int SH>
__device__ __forceinline__ void bilin5(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t wtop, uint32_t wbot, int rnd,
                                       int& o0, int& o1, int& o2, int& o3, int& o4)
{
    asm("v_dot2_i32_i16 %0, %5, %10, %12\n\t"
        "v_dot2_i32_i16 %1, %6, %10, %12\n\t"
        "v_dot2_i32_i16 %2, %7, %10, %12\n\t"
        "v_dot2_i32_i16 %3, %8, %10, %12\n\t"
        "v_dot2_i32_i16 %4, %9, %10, %12\n\t"
        "v_dot2c_i32_i16_dpp %0, %5, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %1, %6, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %2, %7, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %3, %8, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %4, %9, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ashrrev_i32 %0, %13, %0\n\t"
        "v_ashrrev_i32 %1, %13, %1\n\t"
        "v_ashrrev_i32 %2, %13, %2\n\t"
        "v_ashrrev_i32 %3, %13, %3\n\t"
        "v_ashrrev_i32 %4, %13, %4"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(wtop), "v"(wbot), "v"(rnd), "n"(SH));
}

// The I patch itself is only ever used as the accumulator seed of the iteration's interpolation:
//   ((dot + R) >> 9) - ival == (dot + R - (ival << 9)) >> 9,  seed = R - (ival << 9),  ival << 9 == acc & ~511 (acc >= 0)
__device__ __forceinline__ void bilin5_seed(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t wtop, uint32_t wbot, int rnd,
                                            int& o0, int& o1, int& o2, int& o3, int& o4)
{
    asm("v_dot2_i32_i16 %0, %5, %10, %12\n\t"
        "v_dot2_i32_i16 %1, %6, %10, %12\n\t"
        "v_dot2_i32_i16 %2, %7, %10, %12\n\t"
        "v_dot2_i32_i16 %3, %8, %10, %12\n\t"
        "v_dot2_i32_i16 %4, %9, %10, %12\n\t"
        "v_dot2c_i32_i16_dpp %0, %5, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %1, %6, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %2, %7, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %3, %8, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %4, %9, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_and_b32 %0, %13, %0\n\t"
        "v_and_b32 %1, %13, %1\n\t"
        "v_and_b32 %2, %13, %2\n\t"
        "v_and_b32 %3, %13, %3\n\t"
        "v_and_b32 %4, %13, %4\n\t"
        "v_sub_u32 %0, %14, %0\n\t"
        "v_sub_u32 %1, %14, %1\n\t"
        "v_sub_u32 %2, %14, %2\n\t"
        "v_sub_u32 %3, %14, %3\n\t"
        "v_sub_u32 %4, %14, %4"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4)
        // (the two constants in scalar registers: no 32-bit literal dwords in the instruction stream)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(wtop), "v"(wbot), "v"(rnd), "s"(0xfffffe00u), "s"(0x100u));
}

// a * b + c on the low 24 bits of a and b, ONE instruction (left to itself the compiler turns `c += __mul24(a, b)` chains into
// v_mul_i32_i24 + v_add3_u32 plus a sign-extension v_bfe per operand: 1.5 - 2x the instructions of the window loops)
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 16 x 16 -> 32-bit multiply-add on selected halves: the x and y derivative of a pixel share one register (|Ix|, |Iy| <= 4080,
// |J - I| <= 8160 all fit 16 bits), which is what brings the kernel under 96 VGPRs
template <int HA, int HB>
__device__ __forceinline__ int mad16(uint32_t a, uint32_t b, int c)
{
    int r;
    if (HA == 0 && HB == 0) asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if (HA == 0 && HB == 1) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The Newton iteration's version for five window columns c0..c0+4 of this lane's J row (16 bytes in T0..T3): byte pairs by
// v_perm, then as above with a per-column accumulator seed (rounding term minus the I sample):
//   d_i = ((J[c], J[c+1]) . wtop + below(...) . wbot + seed_i) >> 9
template <int C0>
__device__ __forceinline__ void jdiff5(uint32_t T0, uint32_t T1, uint32_t T2, uint32_t T3, uint32_t sel0, uint32_t sel1, uint32_t sel2, uint32_t sel3,
                                       uint32_t wtop, uint32_t wbot, int s0, int s1, int s2, int s3, int s4,
                                       int& d0, int& d1, int& d2, int& d3, int& d4)
{
    // column c reads bytes (c, c+1): dword q = c >> 2, byte o = c & 3; o == 3 takes byte 0 of the next dword
    uint32_t p0, p1, p2, p3, p4;
#define AV_TQ(c) (((c) >> 2) == 0 ? T0 : ((c) >> 2) == 1 ? T1 : ((c) >> 2) == 2 ? T2 : T3)
#define AV_TN(c) (((c) >> 2) == 0 ? T1 : ((c) >> 2) == 1 ? T2 : T3)         /* never needed for c >> 2 == 3 (bytes 12..15 only) */
#define AV_SL(c) (((c) & 3) == 0 ? sel0 : ((c) & 3) == 1 ? sel1 : ((c) & 3) == 2 ? sel2 : sel3)
    asm("v_perm_b32 %5, %10, %11, %20\n\t"
        "v_perm_b32 %6, %12, %13, %21\n\t"
        "v_perm_b32 %7, %14, %15, %22\n\t"
        "v_perm_b32 %8, %16, %17, %23\n\t"
        "v_perm_b32 %9, %18, %19, %24\n\t"
        "v_dot2_i32_i16 %0, %5, %25, %27\n\t"
        "v_dot2_i32_i16 %1, %6, %25, %28\n\t"
        "v_dot2_i32_i16 %2, %7, %25, %29\n\t"
        "v_dot2_i32_i16 %3, %8, %25, %30\n\t"
        "v_dot2_i32_i16 %4, %9, %25, %31\n\t"
        "v_dot2c_i32_i16_dpp %0, %5, %26 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %1, %6, %26 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %2, %7, %26 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %3, %8, %26 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_dot2c_i32_i16_dpp %4, %9, %26 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_ashrrev_i32 %0, 9, %0\n\t"
        "v_ashrrev_i32 %1, 9, %1\n\t"
        "v_ashrrev_i32 %2, 9, %2\n\t"
        "v_ashrrev_i32 %3, 9, %3\n\t"
        "v_ashrrev_i32 %4, 9, %4"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4)
        : "v"(AV_TN(C0)), "v"(AV_TQ(C0)), "v"(AV_TN(C0 + 1)), "v"(AV_TQ(C0 + 1)), "v"(AV_TN(C0 + 2)), "v"(AV_TQ(C0 + 2)),
          "v"(AV_TN(C0 + 3)), "v"(AV_TQ(C0 + 3)), "v"(AV_TN(C0 + 4)), "v"(AV_TQ(C0 + 4)),
          "s"(AV_SL(C0)), "s"(AV_SL(C0 + 1)), "s"(AV_SL(C0 + 2)), "s"(AV_SL(C0 + 3)), "s"(AV_SL(C0 + 4)),
          "v"(wtop), "v"(wbot), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4));
#undef AV_TQ
#undef AV_TN
#undef AV_SL
}

// This lane's row of the I patch and of the two Scharr derivative patches from staged rows r .. r+2 of the window's 18 x 18-byte
// neighbourhood (load_row(t, B): the five dwords of staged row r + t, byte 0 = window column -1), and the lane's share of the
// normal-equation sums.
template <int WIN, typename LoadRow>
__device__ __forceinline__ void lk_patch_rows(LoadRow load_row, uint32_t wtop, uint32_t wbot, int ipx, int ipy, int w, int h, int r,
                                              int (&iv)[WIN], uint32_t (&ixx)[(WIN + 1) / 2], uint32_t (&iyy)[(WIN + 1) / 2], int& a11, int& a12, int& a22)
{
    constexpr int W_BITS = 14;
    a11 = 0; a12 = 0; a22 = 0;
    // staged byte index of window column c is c+1.  E[t][k] = (byte 2k | byte 2k+1 << 16) of staged row r+t; the window
    // pair (c, c+1) of the centre row is O1[k] = (byte 2k+1 | byte 2k+2 << 16) for c = 2k and E[1][k+1] for c = 2k+1.
    uint32_t E[3][9];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        uint32_t B[5];
        load_row(t, B);
#pragma unroll
        for (int k = 0; k < 9; ++k) {                // bytes 2k, 2k+1
            const int b0 = 2 * k;
            E[t][k] = __builtin_amdgcn_perm(0, B[b0 >> 2], 0x0C000C00u | (uint32_t)(b0 & 3) | ((uint32_t)((b0 & 3) + 1) << 16));
        }
    }
    // (hi half of x | lo half of y << 16): the pair one column to the right of x, given y = the next pair
    auto mid = [](uint32_t x, uint32_t y) -> uint32_t { return __builtin_amdgcn_perm(y, x, 0x05040302u); };
    // pixel pairs (c, c+1) of the centre row, c = 0..14
    uint32_t px[WIN];
#pragma unroll
    for (int k = 0; k < 8; ++k) px[2 * k] = mid(E[1][k], E[1][k + 1]);
#pragma unroll
    for (int k = 0; k < 7; ++k) px[2 * k + 1] = E[1][k + 1];
    // Scharr on the even window columns (pairs c = 2k, 2k+1); the odd-aligned pairs are byte permutes of those:
    //   gx[c] = t0[s+1] - t0[s-1],  gy[c] = (t1[s+1] + t1[s-1]) * 3 + t1[s] * 10,  s = c+1,
    //   t0 = (row above + row below) * 3 + row * 10,  t1 = row below - row above
    uint32_t gxp[WIN + 1], gyp[WIN + 1];
    {
        uint32_t t0E[9], t1E[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            t0E[k] = pk_add(pk_mul(pk_add(E[0][k], E[2][k]), 3), pk_mul(E[1][k], 10));
            t1E[k] = pk_sub(E[2][k], E[0][k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {                 // window columns (2k, 2k+1): centres O[k], left E[k], right E[k+1]
            gxp[2 * k] = pk_sub(t0E[k + 1], t0E[k]);
            gyp[2 * k] = pk_add(pk_mul(pk_add(t1E[k + 1], t1E[k]), 3), pk_mul(mid(t1E[k], t1E[k + 1]), 10));
        }
    }
    // the derivative image is zero outside the image: only windows at the border pay for the masks
    const bool inside = ipx >= 0 && ipx + WIN < w && ipy >= 0 && ipy + WIN < h;
    if (__builtin_amdgcn_ballot_w64(!inside) != 0) {
        // the masks are built from an opaque copy of ipx made INSIDE the block: without it the compiler hoists the 60-odd
        // compares / selects of the mask values onto the common path and keeps only the 16 ANDs in here
        int ipx_o = ipx;
        asm volatile("" : "+v"(ipx_o));
        const bool rowin = (unsigned)(ipy + r) < (unsigned)h;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t m0 = (rowin && (unsigned)(ipx_o + 2 * k) < (unsigned)w) ? 0xFFFFu : 0u;
            const uint32_t m1 = (rowin && (unsigned)(ipx_o + 2 * k + 1) < (unsigned)w) ? 0xFFFF0000u : 0u;
            gxp[2 * k] &= (m0 | m1); gyp[2 * k] &= (m0 | m1);
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        gxp[2 * k + 1] = mid(gxp[2 * k], gxp[2 * k + 2]);
        gyp[2 * k + 1] = mid(gyp[2 * k], gyp[2 * k + 2]);
    }
    const int rnd_i = 1 << (W_BITS - 6), rnd_d = 1 << (W_BITS - 1);      // rounding terms, one register each
    // The derivative patches are kept as PAIRS of adjacent columns, (Ix[2p] | Ix[2p+1] << 16) and the same for Iy (|Ix|, |Iy| <= 4080;
    // the 16th column is zero): every window sum is then a chain of v_dot2_i32_i16 -- two columns per instruction -- instead of one
    // v_mad_i32_i16 per column (24 instead of 45 here, 16 + 7 pair packs instead of 30 in every Newton iteration; the same exact
    // integers, whatever the grouping).
    int ixv[WIN + 1], iyv[WIN + 1];
    ixv[WIN] = 0; iyv[WIN] = 0;
#pragma unroll
    for (int c = 0; c < WIN; c += 5) {
        bilin5_seed(px[c], px[c + 1], px[c + 2], px[c + 3], px[c + 4], wtop, wbot, rnd_i, iv[c], iv[c + 1], iv[c + 2], iv[c + 3], iv[c + 4]);
        bilin5<W_BITS>(gxp[c], gxp[c + 1], gxp[c + 2], gxp[c + 3], gxp[c + 4], wtop, wbot, rnd_d, ixv[c], ixv[c + 1], ixv[c + 2], ixv[c + 3], ixv[c + 4]);
        bilin5<W_BITS>(gyp[c], gyp[c + 1], gyp[c + 2], gyp[c + 3], gyp[c + 4], wtop, wbot, rnd_d, iyv[c], iyv[c + 1], iyv[c + 2], iyv[c + 3], iyv[c + 4]);
    }
#pragma unroll
    for (int p = 0; p < (WIN + 1) / 2; ++p) {
        ixx[p] = __builtin_amdgcn_perm((uint32_t)ixv[2 * p + 1], (uint32_t)ixv[2 * p], 0x05040100u);
        iyy[p] = __builtin_amdgcn_perm((uint32_t)iyv[2 * p + 1], (uint32_t)iyv[2 * p], 0x05040100u);
        a11 = dot2(ixx[p], ixx[p], a11);
        a12 = dot2(ixx[p], iyy[p], a12);
        a22 = dot2(iyy[p], iyy[p], a22);
    }
    if (r >= WIN) { a11 = 0; a12 = 0; a22 = 0; }
}

// PTS: points per workgroup (16 = four wavefronts, 4 = one).  The wavefronts of a workgroup share nothing -- every point has its own
// LDS tile and there is no barrier -- so the workgroup size is purely a placement matter: see lk_track_g16_kernel.
template <int WIN, int OCC, bool PROF = false, int PTS = 16>
__device__ __forceinline__ bool lk_track_g16_body(const LKArgs& a)
{
    static_assert(WIN == 15, "lane map is built for the reference's 15x15 window");
    constexpr int W_BITS = 14;
    __shared__ uint32_t tile_all[PTS][TILE_DWORDS];
    // phase stamps (PROF only): cycles since the previous stamp are booked on the phase that just ended
    unsigned pt[LKP_N] = {0, 0, 0, 0, 0}, plast = 0;             // 32-bit: a wavefront lives ~10^5 cycles
    unsigned pn_iter = 0, pn_restage = 0, pn_level = 0;
    unsigned long long rt0 = 0;
    if (PROF) { rt0 = __builtin_amdgcn_s_memrealtime(); plast = (unsigned)__builtin_readcyclecounter(); }
#define LK_STAMP(ph) do { if (PROF) { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); pt[ph] += t_ - plast; plast = t_; } } while (0)
    const int g = threadIdx.x >> 4;                 // point slot of this 16-lane group inside the workgroup
    const int r = threadIdx.x & 15;                 // window row owned by this lane (row 15 only feeds row 14)
    uint32_t* tile = tile_all[g];
    // Workgroup -> (point set, block of 16 points).  The dispatcher deals workgroups round-robin over the 8 XCDs (linear id mod 8,
    // MI355X_MICROARCH.md "Workgroup dispatch"), and each XCD has its own L2: with the plain (block, set) grid the ~19 workgroups of
    // one stream land on all eight XCDs and every L2 fetches that stream's two pyramids (profiles/r03/fetch_calib.json, rows32: a
    // window gather in image order reads 2.6x the image bytes).  The 1-D launch keeps a set's workgroups on ONE XCD: id L ->
    // XCD label L & 7, set = label + 8 * ((L >> 3) / gx), block = (L >> 3) % gx.  Placement only: any mapping is correct.
    int s, bx;
    if (a.gx > 0) { const int L = blockIdx.x, j = L >> 3; s = (L & 7) + 8 * (j / a.gx); bx = j % a.gx; if (s >= a.n_set) return false; }
    else { s = blockIdx.y; bx = blockIdx.x; }
    const int slot = bx * PTS + g;
    const int sI = a.mapI ? a.mapI[s] : s, sJ = a.mapJ ? a.mapJ[s] : s;      // storage entries of the two images (scalar loads)
    if (sI < 0 || sJ < 0) return false;
    // Without an index list the point of a slot is known before the set's count is: its loads are issued ahead of the count's (one
    // dependent memory round trip less at the head of every wavefront; a slot past the count reads a valid entry and leaves)
    const bool direct = a.index == nullptr && slot < a.cap;
    float prevx0 = 0.f, prevy0 = 0.f, curx = 0.f, cury = 0.f;
    if (direct) {
        const size_t p0 = (size_t)s * a.cap + slot;
        prevx0 = a.prev[2 * p0]; prevy0 = a.prev[2 * p0 + 1]; curx = a.next[2 * p0]; cury = a.next[2 * p0 + 1];
    }
    const int n = min(a.count[s], a.cap);
    if (slot >= n) return false;                    // uniform per 16-lane group
    const int pidx = a.index ? a.index[(size_t)s * a.cap + slot] : slot;

    const uint8_t* PI = a.pyrI + sI * a.stream_stride;
    const uint8_t* PJ = a.pyrJ + sJ * a.stream_stride;
    const size_t pi = (size_t)s * a.cap + pidx;
    if (!direct) { prevx0 = a.prev[2 * pi]; prevy0 = a.prev[2 * pi + 1]; curx = a.next[2 * pi]; cury = a.next[2 * pi + 1]; }
    const float halfWin = (WIN - 1) * 0.5f;
    const bool rowact = r < WIN;
    bool ok = true;
    // byte-pair selectors of v_perm: (byte o, byte o+1) zero-extended into the two 16-bit halves; o+1 == 4 = byte 0 of the next dword
    const uint32_t sel0 = 0x0C010C00u, sel1 = 0x0C020C01u, sel2 = 0x0C030C02u, sel3 = 0x0C040C03u;

    for (int level = a.g.levels - 1; level >= 0; --level) {
        const int w = a.g.w[level], h = a.g.h[level], pitch = a.g.pitch[level];
        const int col_lo = -AV_PYR_BORDER, col_hi = pitch - AV_PYR_BORDER;
        const float scale = __int_as_float((127 - level) << 23);       // 2^-level (written as 1. / (1 << level) it is an fp64 division per level)
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

        // ---- stage the 18 x 18-byte neighbourhood of the I window: rows ipy-1 .., bytes ipx-1 .. ipx+16 of each as five dwords
        //      read at their (unaligned) byte address, so that every staged row starts at window column -1.  All bytes lie inside
        //      the padded pyramid row: ipx - 1 >= -16 and ipx + 16 <= w + 15 (the 5th dword is fetched 2 bytes early and shifted).
        // Lane map of both staging loops: lane r handles dword (r & 7) of rows 2k + (r >> 3): the global address is a
        // wave-uniform base (padded level origin + 2k rows, scalar) plus ONE per-lane 32-bit offset, the LDS address one
        // per-lane base plus an immediate -- no per-load index arithmetic on the vector unit.
        const int sub = r >> 3, dwl = r & 7;
        const uint32_t tofs = (uint32_t)(sub * (TPITCH / 4) + dwl);
        LK_STAMP(LKP_HEAD);
        if (PROF) ++pn_level;
        wave_lds_sync();
        typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
        {
            // source of this level's I rows: the padded pyramid level (origin at row -16, column -16, frame included) or, for
            // level 0 in place, the caller's image (origin at pixel (0, 0), no frame: windows over the border take the slow path)
            const bool extI = level == 0 && a.imgI != nullptr;                      // wave-uniform
            const uint8_t* L0 = extI ? a.imgI + (size_t)sI * (size_t)a.imgI_stride : PI + a.g.off[level];
            const int pI = extI ? w : pitch, bI = extI ? 0 : AV_PYR_BORDER;
            const bool act = dwl < 5;                                    // 5 of the 8 slots of a row carry a dword
            if (!extI || (ipx >= 1 && ipx + 16 < w && ipy >= 1 && ipy + 16 < h)) {
                const uint32_t loff = (uint32_t)(__mul24(ipy - 1 + sub + bI, pI) + (ipx - 1 + bI) + (dwl < 4 ? 4 * dwl : 14));
                const uint32_t shamt = dwl == 4 ? 16u : 0u;
                uint32_t sv[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {                               // 18 rows
                    sv[k] = 0;
                    if (act) sv[k] = *reinterpret_cast<const u32_unaligned*>(L0 + (size_t)(2 * k) * (size_t)pI + loff);
                }
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    if (act) tile[tofs + 2 * k * (TPITCH / 4)] = sv[k] >> shamt;
            } else if (act) {
                // the window reaches over the image border: the bytes the padded level's BORDER_REFLECT_101 frame would hold
                const int xb = ipx - 1 + (dwl < 4 ? 4 * dwl : 16), nbyte = dwl < 4 ? 4 : 2;
#pragma unroll 1
                for (int k = 0; k < 9; ++k) {
                    const uint8_t* row = L0 + (size_t)av_reflect101(ipy - 1 + sub + 2 * k, h) * (size_t)w;
                    uint32_t v = 0;
#pragma unroll 1
                    for (int bb = 0; bb < nbyte; ++bb) v |= (uint32_t)row[av_reflect101(xb + bb, w)] << (8 * bb);
                    tile[tofs + 2 * k * (TPITCH / 4)] = v;
                }
            }
        }
        wave_lds_sync();
        LK_STAMP(LKP_ISTAGE);

        // ---- this lane's row of the I patch and of the Scharr patches: staged rows r..r+2, 18 bytes of each -------------
        int iv[WIN];
        uint32_t ixx[(WIN + 1) / 2], iyy[(WIN + 1) / 2];      // (Ix[2p] | Ix[2p+1] << 16), the same for Iy: pairs of window columns
        int a11, a12, a22;
        lk_patch_rows<WIN>([&](int t, uint32_t (&B)[5]) {
                               const uint32_t* rowp = tile + __mul24(r + t, TPITCH / 4);
#pragma unroll
                               for (int k = 0; k < 5; ++k) B[k] = rowp[k];
                           }, wtop, wbot, ipx, ipy, w, h, r, iv, ixx, iyy, a11, a12, a22);
        const float A11 = row_sum16_scaled(a11);
        const float A12 = row_sum16_scaled(a12);
        const float A22 = row_sum16_scaled(a22);
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if (minEig < a.min_eig_up || D < 1.1920928955078125e-7f) {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.f / D;

        float wx = curx - halfWin, wy = cury - halfWin;
        float pdx = 0.f, pdy = 0.f;
        int X0 = 0, Y0 = 0;
        bool staged = false;
        LK_STAMP(LKP_SETUP);
        for (int j = 0; j < a.max_iter; ++j) {
            if (PROF) ++pn_iter;
            const int inx = (int)floorf(wx), iny = (int)floorf(wy);
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                if (level == 0) ok = false;
                break;
            }
            int dx0 = inx - X0, dy0 = iny - Y0;
            if (!staged || (unsigned)dx0 > (unsigned)(TILE_COLS - 17) || (unsigned)dy0 > (unsigned)(TILE_ROWS - 16)) {
                X0 = min(max((inx - 6) & ~3, col_lo), col_hi - TILE_COLS);
                Y0 = min(max(iny - 4, -AV_PYR_BORDER), h + AV_PYR_BORDER - TILE_ROWS);
                LK_STAMP(LKP_ITER);
                if (PROF) ++pn_restage;
                wave_lds_sync();
                {
                    const bool extJ = level == 0 && a.imgJ != nullptr;              // wave-uniform
                    const uint8_t* LJ0 = extJ ? a.imgJ + (size_t)sJ * (size_t)a.imgJ_stride : PJ + a.g.off[level];
                    const int pJ = extJ ? w : pitch, bJ = extJ ? 0 : AV_PYR_BORDER;
                    if (!extJ || (X0 >= 0 && X0 + TILE_COLS <= w && Y0 >= 0 && Y0 + TILE_ROWS <= h)) {
                        const uint32_t joff = (uint32_t)(__mul24(Y0 + sub + bJ, pJ) + X0 + bJ + 4 * dwl);
                        uint32_t sv[TILE_ROWS / 2];
#pragma unroll
                        for (int k = 0; k < TILE_ROWS / 2; ++k)        // 24 rows x 8 dwords
                            sv[k] = *reinterpret_cast<const u32_unaligned*>(LJ0 + (size_t)(2 * k) * (size_t)pJ + joff);
#pragma unroll
                        for (int k = 0; k < TILE_ROWS / 2; ++k) tile[tofs + 2 * k * (TPITCH / 4)] = sv[k];
                    } else {
#pragma unroll 1
                        for (int k = 0; k < TILE_ROWS / 2; ++k) {
                            const uint8_t* row = LJ0 + (size_t)av_reflect101(Y0 + sub + 2 * k, h) * (size_t)w;
                            uint32_t v = 0;
#pragma unroll 1
                            for (int bb = 0; bb < 4; ++bb) v |= (uint32_t)row[av_reflect101(X0 + 4 * dwl + bb, w)] << (8 * bb);
                            tile[tofs + 2 * k * (TPITCH / 4)] = v;
                        }
                    }
                }
                wave_lds_sync();
                LK_STAMP(LKP_JSTAGE);
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
            const uint32_t T0 = __builtin_amdgcn_alignbyte(d1, d0, sh), T1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
            const uint32_t T2 = __builtin_amdgcn_alignbyte(d3, d2, sh), T3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
            int df[WIN];
            jdiff5<0>(T0, T1, T2, T3, sel0, sel1, sel2, sel3, wtop, wbot, iv[0], iv[1], iv[2], iv[3], iv[4], df[0], df[1], df[2], df[3], df[4]);
            jdiff5<5>(T0, T1, T2, T3, sel0, sel1, sel2, sel3, wtop, wbot, iv[5], iv[6], iv[7], iv[8], iv[9], df[5], df[6], df[7], df[8], df[9]);
            jdiff5<10>(T0, T1, T2, T3, sel0, sel1, sel2, sel3, wtop, wbot, iv[10], iv[11], iv[12], iv[13], iv[14], df[10], df[11], df[12], df[13], df[14]);
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int p = 0; p < (WIN + 1) / 2; ++p) {
                // (|J - I| <= 8160 fits 16 bits; the high half of the last pair multiplies the zero 16th column: df[14] goes in as it is)
                const uint32_t dp = 2 * p + 1 < WIN ? __builtin_amdgcn_perm((uint32_t)df[2 * p + 1 < WIN ? 2 * p + 1 : 0], (uint32_t)df[2 * p], 0x05040100u) : (uint32_t)df[2 * p];
                b1 = dot2(dp, ixx[p], b1);
                b2 = dot2(dp, iyy[p], b2);
            }
            if (!rowact) { b1 = 0; b2 = 0; }
            const float fb1 = row_sum16_scaled(b1);
            const float fb2 = row_sum16_scaled(b2);
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            wx += dx; wy += dy;
            curx = wx + halfWin; cury = wy + halfWin;
            // delta.ddot(delta) <= eps^2 is an fp64 test in OpenCV.  e32 lies within 2^-22 (relative) of the exact dx^2 + dy^2, so outside
            // (eps_lo, eps_hi) the float compare decides the same way; inside (once in ~10^5 iterations) the fp64 form does.
            const float e32 = __builtin_fmaf(dx, dx, dy * dy);
            bool conv = e32 <= a.eps_lo;
            if (__builtin_amdgcn_ballot_w64(e32 > a.eps_lo && e32 < a.eps_hi) != 0) {
                asm volatile("; fp64 step test");          // keeps the block a branch (if-converted, its fp64 instructions ran every iteration)
                conv = (double)dx * dx + (double)dy * dy <= a.eps2;
            }
            if (conv) break;
            // |x| < 0.01 (double) for a float x <=> |x| <= 0.01f: 0.01f = 0.00999999977... < 0.01 < the next float up
            if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                curx -= dx * 0.5f; cury -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        LK_STAMP(LKP_ITER);
    }
    if (r == 0) {
        a.next[2 * pi] = curx;
        a.next[2 * pi + 1] = cury;
        a.status[pi] = ok ? 1 : 0;
    }
    if (PROF) {
        LK_STAMP(LKP_HEAD);
        // one record per wavefront: its first lane books the wave's clock; iteration / level counts are the maxima over its points
        // (what the wave executed); every point leader books its own counts for the per-point means
        unsigned wi = pn_iter, wr = pn_restage, wl = pn_level;
#pragma unroll
        for (int d = 16; d < 64; d <<= 1) { wi = max(wi, (unsigned)__shfl_xor((int)wi, d, 64)); wr = max(wr, (unsigned)__shfl_xor((int)wr, d, 64)); wl = max(wl, (unsigned)__shfl_xor((int)wl, d, 64)); }
        // (every wavefront stamps; one workgroup in 61 books: 10^7 same-address atomics per launch made the profiled kernel 9x slower)
        const bool book = blockIdx.x % 61 == 0;
        if (book && (threadIdx.x & 63) == 0) {
            for (int k = 0; k < LKP_N; ++k) atomicAdd(a.prof + k, (unsigned long long)pt[k]);
            atomicAdd(a.prof + LKP_WAVES, 1ull);
            atomicAdd(a.prof + LKP_SLOTS + 3, __builtin_amdgcn_s_memrealtime() - rt0);      // the same span on the constant 100 MHz clock
            atomicAdd(a.prof + LKP_ITERS, (unsigned long long)wi);
            atomicAdd(a.prof + LKP_RESTAGES, (unsigned long long)wr);
            atomicAdd(a.prof + LKP_LEVELS, (unsigned long long)wl);
        }
        if (book && r == 0) { atomicAdd(a.prof + LKP_SLOTS, 1ull); atomicAdd(a.prof + LKP_SLOTS + 1, (unsigned long long)pn_iter); atomicAdd(a.prof + LKP_SLOTS + 2, (unsigned long long)pn_restage); }
    }
#undef LK_STAMP
    return true;
}

// (Round 5, archived -- profiles/r05/lk_dma_experiment.md, code in git 741c321: both stagings by LDS-DMA, the J tile requested before
//  the patch set-up and the next level's I neighbourhood before the iterations.  Bit-exact, the wavefront's life 13 % shorter, the
//  launch 1.7 % LONGER (0.955 vs 0.939 ms): at five wavefronts per SIMD the other four already fill the slots a staging wavefront
//  leaves; the kernel is bound by VALU issue.)
// 96 VGPRs: 5 waves per SIMD without spills (the kernel is VALU-issue bound: 4 -> 5 waves bought 1 %, a 6-wave build with
// 11 spilled values lost 5 %).
// One wavefront (four points) per workgroup: the wavefronts of this kernel share nothing, so the workgroup size is a placement matter
// only -- and it matters twice.  (1) A 256-thread workgroup holds its four slots until its SLOWEST wavefront is done (the points of a
// launch take 3 .. 30 Newton iterations); single wavefronts hand their slot back one by one: front-end alone 0.917 -> 0.862 ms per
// launch (211.1 -> 220.5 k frames/s).  (2) A 256-thread workgroup needs a free slot on all four SIMDs of a CU at once: beside the
// filter's single-wavefront tasks, each of which takes ONE slot on ONE SIMD, a CU with three free slots starts nothing -- every
// resident filter wavefront can keep a whole LK workgroup out: complete path 161.6 -> 168.9 k.  (profiles/r05/README.md)
template <int WIN> __global__ __launch_bounds__(64, 5) void lk_track_g16_kernel(LKArgs a) { lk_track_g16_body<WIN, 5, false, 4>(a); }
// AV_LK_WG=256 (A/B): four wavefronts (16 points) per workgroup, rounds 2-4's launch shape.  (Two wavefronts per workgroup lie in
// between: front-end alone 218.3 k against 220.2 k, complete path 166.2 against 168.7 k.  profiles/r05/README.md)
template <int WIN> __global__ __launch_bounds__(256, 5) void lk_track_g16_w4_kernel(LKArgs a) { lk_track_g16_body<WIN, 5>(a); }
// (Forward and backward pass of a stereo match as ONE launch -- the 16 lanes that tracked a point forward track it back, three launches
//  fewer per front-end step -- was built and measured in round 5: front-end alone 207.9 / 206.2 k against 206.1 / 206.8 k frames/s, one
//  stream 0.649 against 0.645 ms per frame: nothing, removed.  profiles/r05/README.md)
// (a four-wave build that leaves 128 register rows per SIMD to the filter's kernels: 146.7 k against 152.8 k frames/s, round 4)
// AV_LK_PROF=1: the same body with s_memtime stamps at the phase boundaries (I staging / patch set-up / J staging / Newton
// iterations), summed over the wavefronts of every launch and printed at exit (profiles/r05/lk_phase_stamps.txt).
template <int WIN> __global__ __launch_bounds__(256, 5) void lk_track_g16_prof_kernel(LKArgs a) { lk_track_g16_body<WIN, 5, true>(a); }

static unsigned long long* g_lk_prof = nullptr;
static void lk_prof_report()
{
    if (!g_lk_prof) return;
    unsigned long long h[LKP_SLOTS + 4];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, g_lk_prof, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
    static const char* names[LKP_N] = {"head+tail", "I staging", "set-up", "J staging", "iterations"};
    double tot = 0;
    for (int k = 0; k < LKP_N; ++k) tot += (double)h[k];
    const double nw = (double)h[LKP_WAVES];
    fprintf(stderr, "AV_LK_PROF: %.0f wavefronts booked (one workgroup in 61), %.0f s_memtime ticks per wavefront = %.1f us on the 100 MHz clock: %.0f ticks per us\n",
            nw, tot / nw, (double)h[LKP_SLOTS + 3] / nw / 100., tot / ((double)h[LKP_SLOTS + 3] / 100.));
    for (int k = 0; k < LKP_N; ++k) fprintf(stderr, "AV_LK_PROF:   %-10s %8.0f ticks per wavefront  %5.1f %%\n", names[k], (double)h[k] / nw, 100. * (double)h[k] / tot);
    fprintf(stderr, "AV_LK_PROF: per wavefront (max over its 4 points): %.2f levels, %.2f iterations, %.2f J stagings; per point: %.2f iterations, %.2f J stagings\n",
            (double)h[LKP_LEVELS] / nw, (double)h[LKP_ITERS] / nw, (double)h[LKP_RESTAGES] / nw,
            (double)h[LKP_SLOTS + 1] / (double)h[LKP_SLOTS], (double)h[LKP_SLOTS + 2] / (double)h[LKP_SLOTS]);
}


// =================================================================================================
// Any other window (config.win_size, src/config.py:35-38; 3 <= win <= 31): one WAVEFRONT per point, the win x win window
// pixels dealt over the 64 lanes (at most 16 per lane, the I / Ix / Iy values of a lane's pixels in registers), the same
// arithmetic in the same order as the 16-lane kernel and the oracle: W_BITS = 14 bilinear weights, Scharr derivatives of the
// first image taken at integer pixels (BORDER_REFLECT_101 neighbours inside the image, zero outside), exact integer window
// sums (int64 per lane, butterfly over the wavefront) rounded once, float Newton steps.  Pixels are read straight from the
// padded pyramid level (or the caller's image for level 0) through the cache -- this is the general path, not the fast one.
// =================================================================================================
constexpr int LKG_MAX_WIN = 31, LKG_PER_LANE = (LKG_MAX_WIN * LKG_MAX_WIN + 63) / 64;

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

struct LKLevel { const uint8_t* base; int pitch, w, h; };      // base = address of pixel (0, 0)
__device__ __forceinline__ int lkg_pix(const LKLevel& L, int x, int y) { return L.base[(size_t)av_reflect101_any(y, L.h) * L.pitch + av_reflect101_any(x, L.w)]; }

__global__ __launch_bounds__(256) void lk_track_generic_kernel(LKArgs a, int WIN)
{
    constexpr int W_BITS = 14;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.y, slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int sI = a.mapI ? a.mapI[s] : s, sJ = a.mapJ ? a.mapJ[s] : s;
    if (sI < 0 || sJ < 0) return;
    const int n = min(a.count[s], a.cap);
    if (slot >= n) return;                                   // wave-uniform
    const int pidx = a.index ? a.index[(size_t)s * a.cap + slot] : slot;
    const size_t pi = (size_t)s * a.cap + pidx;
    const float prevx0 = a.prev[2 * pi], prevy0 = a.prev[2 * pi + 1];
    float curx = a.next[2 * pi], cury = a.next[2 * pi + 1];
    const float halfWin = (WIN - 1) * 0.5f;
    const int win2 = WIN * WIN;
    const double FLT_SCALE_D = 1.0 / (1 << 20);
    bool ok = true;
    for (int level = a.g.levels - 1; level >= 0; --level) {
        const int w = a.g.w[level], h = a.g.h[level];
        LKLevel LI, LJ;
        {
            const bool extI = level == 0 && a.imgI != nullptr, extJ = level == 0 && a.imgJ != nullptr;
            const int pitch = a.g.pitch[level];
            LI.base = extI ? a.imgI + (size_t)sI * (size_t)a.imgI_stride : a.pyrI + sI * a.stream_stride + a.g.off[level] + (size_t)AV_PYR_BORDER * pitch + AV_PYR_BORDER;
            LI.pitch = extI ? w : pitch; LI.w = w; LI.h = h;
            LJ.base = extJ ? a.imgJ + (size_t)sJ * (size_t)a.imgJ_stride : a.pyrJ + sJ * a.stream_stride + a.g.off[level] + (size_t)AV_PYR_BORDER * pitch + AV_PYR_BORDER;
            LJ.pitch = extJ ? w : pitch; LJ.w = w; LJ.h = h;
        }
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
        // ---- I patch and derivative patches of this lane's window pixels
        short iv[LKG_PER_LANE], ixv[LKG_PER_LANE], iyv[LKG_PER_LANE];
        long long sA11 = 0, sA12 = 0, sA22 = 0;
#pragma unroll
        for (int t = 0; t < LKG_PER_LANE; ++t) {
            const int idx = lane + 64 * t;
            iv[t] = 0; ixv[t] = 0; iyv[t] = 0;
            if (idx < win2) {
                const int y = idx / WIN, x = idx - y * WIN, X = ipx + x, Y = ipy + y;
                // 4 x 4 neighbourhood (X-1 .. X+2, Y-1 .. Y+2), reflected: the four bilinear taps and their Scharr stencils
                int P[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) P[r][c] = lkg_pix(LI, X - 1 + c, Y - 1 + r);
                const int ival = AV_DESCALE(P[1][1] * iw00 + P[1][2] * iw01 + P[2][1] * iw10 + P[2][2] * iw11, W_BITS - 5);
                // derivative at the integer pixel (X + cx, Y + cy), cx, cy in {0, 1}: zero outside the image (calc() pads the derivative
                // image with BORDER_CONSTANT), else Scharr over reflected neighbours.  The stencil of (X+cx, Y+cy) needs pixels
                // X+cx-1 .. X+cx+1: for cx = 1 that is X .. X+2, all inside P; reflection of a neighbour of an INSIDE pixel equals the
                // reflected read already taken (reflect101 is applied per coordinate).
                int dxv[2][2], dyv[2][2];
#pragma unroll
                for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                    for (int cx = 0; cx < 2; ++cx) {
                        const bool in = (unsigned)(X + cx) < (unsigned)w && (unsigned)(Y + cy) < (unsigned)h;
                        // t0(c) = (above + below) * 3 + centre * 10, t1(c) = below - above, at columns cx-1 .. cx+1 (P columns cx .. cx+2)
                        int t0[3], t1[3];
#pragma unroll
                        for (int k = 0; k < 3; ++k) { t0[k] = (P[cy][cx + k] + P[cy + 2][cx + k]) * 3 + P[cy + 1][cx + k] * 10; t1[k] = P[cy + 2][cx + k] - P[cy][cx + k]; }
                        dxv[cy][cx] = in ? (int)(short)(t0[2] - t0[0]) : 0;
                        dyv[cy][cx] = in ? (int)(short)((t1[2] + t1[0]) * 3 + t1[1] * 10) : 0;
                    }
                const int ix = AV_DESCALE(dxv[0][0] * iw00 + dxv[0][1] * iw01 + dxv[1][0] * iw10 + dxv[1][1] * iw11, W_BITS);
                const int iy = AV_DESCALE(dyv[0][0] * iw00 + dyv[0][1] * iw01 + dyv[1][0] * iw10 + dyv[1][1] * iw11, W_BITS);
                iv[t] = (short)ival; ixv[t] = (short)ix; iyv[t] = (short)iy;
                sA11 += (long long)ixv[t] * ixv[t]; sA12 += (long long)ixv[t] * iyv[t]; sA22 += (long long)iyv[t] * iyv[t];
            }
        }
        const float A11 = (float)((double)wave_sum_i64(sA11) * FLT_SCALE_D);
        const float A12 = (float)((double)wave_sum_i64(sA12) * FLT_SCALE_D);
        const float A22 = (float)((double)wave_sum_i64(sA22) * FLT_SCALE_D);
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if ((double)minEig < a.min_eig || D < 1.1920928955078125e-7f) {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.f / D;
        float wx = curx - halfWin, wy = cury - halfWin;
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
            long long sb1 = 0, sb2 = 0;
#pragma unroll
            for (int t = 0; t < LKG_PER_LANE; ++t) {
                const int idx = lane + 64 * t;
                if (idx < win2) {
                    const int y = idx / WIN, x = idx - y * WIN, X = inx + x, Y = iny + y;
                    const int diff = AV_DESCALE(lkg_pix(LJ, X, Y) * iw00 + lkg_pix(LJ, X + 1, Y) * iw01 + lkg_pix(LJ, X, Y + 1) * iw10 + lkg_pix(LJ, X + 1, Y + 1) * iw11, W_BITS - 5) - iv[t];
                    sb1 += (long long)diff * ixv[t]; sb2 += (long long)diff * iyv[t];
                }
            }
            const float fb1 = (float)((double)wave_sum_i64(sb1) * FLT_SCALE_D);
            const float fb2 = (float)((double)wave_sum_i64(sb2) * FLT_SCALE_D);
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

static int lk_fill_args(LKArgs& a, const uint8_t* pyrI, const uint8_t* pyrJ, int64_t stream_stride, const PyrGeom& g,
                        const float* prev, float* next, uint8_t* status, const int* count, int cap,
                        const LKParams& p, const int* index,
                        const uint8_t* imgI, int64_t imgI_stride, const uint8_t* imgJ, int64_t imgJ_stride, const int* mapI, const int* mapJ)
{
    if (p.win < 3 || p.win > LKG_MAX_WIN) { av_set_error("av_lk_track: winSize %d outside 3 .. %d", p.win, LKG_MAX_WIN); return AV_E_INVALID; }
    a.pyrI = pyrI; a.pyrJ = pyrJ; a.stream_stride = stream_stride; a.g = g;
    // cv::buildOpticalFlowPyramid stops at the first level whose successor would be no larger than the window in either
    // dimension (OpenCV 4.x lkpyramid.cpp: `if (sz.width <= winSize.width || sz.height <= winSize.height) return level`);
    // level 0 is always tracked.  Not reached by the reference's 752 x 480 / 15 / 3.
    for (int l = 1; l < a.g.levels; ++l)
        if (a.g.w[l] <= p.win || a.g.h[l] <= p.win) { a.g.levels = l; break; }
    a.prev = prev; a.next = next; a.status = status; a.count = count; a.index = index; a.cap = cap;
    a.max_iter = p.max_iter; a.eps2 = p.eps2;
    a.min_eig_up = (float)p.min_eig;
    if ((double)a.min_eig_up < p.min_eig) a.min_eig_up = nextafterf(a.min_eig_up, INFINITY);
    if (p.eps2 >= 1e-30) { a.eps_lo = (float)(p.eps2 * (1. - 0x1p-18)); a.eps_hi = (float)(p.eps2 * (1. + 0x1p-18)); }
    else { a.eps_lo = -1.f; a.eps_hi = INFINITY; }                 // e32 may underflow: every test in fp64
    a.min_eig = p.min_eig;
    a.imgI = imgI; a.imgJ = imgJ; a.imgI_stride = imgI_stride; a.imgJ_stride = imgJ_stride; a.mapI = mapI; a.mapJ = mapJ; a.prof = nullptr;
    a.n_set = 0; a.gx = 0;
    return AV_OK;
}

static bool lk_prof_on()
{
    static const bool prof = [] {
        const char* e = getenv("AV_LK_PROF");
        if (!(e && atoi(e) == 1)) return false;
        if (hipMalloc(&g_lk_prof, sizeof(unsigned long long) * (LKP_SLOTS + 4)) != hipSuccess) { g_lk_prof = nullptr; return false; }
        (void)hipMemset(g_lk_prof, 0, sizeof(unsigned long long) * (LKP_SLOTS + 4));
        (void)hipDeviceSynchronize();
        atexit(lk_prof_report);
        return true;
    }();
    return prof;
}

// grid of the 16-lane kernels: XCD-aware 1-D launch (lk_track_g16_body) unless AV_LK_XCD=0
static dim3 lk_g16_grid(LKArgs& a, int n_set, int launch_pts, int pts_per_wg)
{
    static const bool xcd_map = [] { const char* e = getenv("AV_LK_XCD"); return !(e && atoi(e) == 0); }();      // A/B switch
    const int gx = (launch_pts + pts_per_wg - 1) / pts_per_wg;
    a.n_set = n_set; a.gx = xcd_map ? gx : 0;
    return xcd_map ? dim3((unsigned)gx * 8u * (unsigned)((n_set + 7) / 8)) : dim3(gx, n_set);
}

int av_launch_lk(const uint8_t* pyrI, const uint8_t* pyrJ, int64_t stream_stride, int n_set, const PyrGeom& g,
                 const float* prev, float* next, uint8_t* status, const int* count, int cap, int launch_pts,
                 const LKParams& p, hipStream_t st, const int* index,
                 const uint8_t* imgI, int64_t imgI_stride, const uint8_t* imgJ, int64_t imgJ_stride, const int* mapI, const int* mapJ)
{
    if (n_set <= 0 || launch_pts <= 0) return AV_OK;
    LKArgs a;
    int rc = lk_fill_args(a, pyrI, pyrJ, stream_stride, g, prev, next, status, count, cap, p, index, imgI, imgI_stride, imgJ, imgJ_stride, mapI, mapJ);
    if (rc) return rc;
    if (launch_pts > cap) launch_pts = cap;
    if (p.win != 15) {           // any other window of config.win_size: the general kernel (one wavefront per point)
        a.n_set = n_set; a.gx = 0;
        hipLaunchKernelGGL(lk_track_generic_kernel, dim3((launch_pts + 3) / 4, n_set), dim3(256), 0, st, a, p.win);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    const bool prof = lk_prof_on();
    static const bool w1 = [] { const char* e = getenv("AV_LK_WG"); return !(e && atoi(e) == 256); }();      // one wavefront per workgroup; AV_LK_WG=256 (A/B): four
    const int wgp = (prof || !w1) ? 256 : 64;
    const dim3 grid = lk_g16_grid(a, n_set, launch_pts, wgp / 16);
    a.prof = g_lk_prof;
    if (prof) hipLaunchKernelGGL(lk_track_g16_prof_kernel<15>, grid, dim3(256), 0, st, a);
    else if (wgp == 64) hipLaunchKernelGGL(lk_track_g16_kernel<15>, grid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL(lk_track_g16_w4_kernel<15>, grid, dim3(256), 0, st, a);
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
                        count_dev, cap, cap, p, (hipStream_t)stream, nullptr);
}
