// msckf.hip -- the MSCKF measurement update's batched small-dense fp64 linear algebra (gfx950).
//
// Reference being replaced (src/msckf.py unless noted), SURVEY.md section 8(a) rows a14-a25:
//   process_model:275-339          F, G, 3rd-order Phi, OC-KF fix, P <- Phi P Phi^T + Q, cross terms   -> propagate_kernel
//   state_augmentation:390-423     P grows by the 6x21 Jacobian J                                      -> augment_kernel
//   feature/feature_position_initializer.py:6-76 (+ feature_observation.py, feature_depth_estimator.py)
//                                  Levenberg-Marquardt triangulation in (alpha, beta, rho)             -> triangulate_kernel
//   measurement_jacobian:443-507, feature_jacobian:509-546, gating_test:604-612
//                                  per-feature 4x6 / 4x3 blocks, null-space projection, chi^2 gate     -> feature_kernel
//   measurement_update:548-602     thin QR (m > n), S = H P H^T + s^2 I, K, delta_x, P <- (I-KH)P, sym -> update_kernel
//   prune_cam_state_buffer:774-786 delete 6 rows/cols per removed camera state                         -> remove_cam_kernel
//
// The reference uses numpy/LAPACK: full SVD of H_f for the null space (msckf.py:540), Householder
// QR (msckf.py:555) and LU solves (msckf.py:565,607).  Here: Householder QR of H_f (3 reflectors;
// any orthonormal basis of the left null space gives the same gamma, delta_x and P -- SURVEY 8a
// "invariants"), Householder QR for the compression, Cholesky for the SPD solves.  fp64 everywhere;
// parity with the reference is to tolerance (stated in tests/test_gpu_msckf.py), not bitwise.
//
// MI355X mapping: one wavefront per feature for triangulation (views across lanes, butterfly
// reductions); one 256-thread workgroup per feature for the Jacobian/projection/gate with the
// (4M x 6M) block and the (4M-3)^2 innovation covariance in LDS (sized by the largest track in the
// batch); one 1024-thread workgroup for the stacked update, matrices <= 141x141 in L2-resident
// scratch with the Cholesky factor packed in LDS (80 KB).  MFMA is not used: the only dense
// contraction is <= 141^3 and is latency-, not throughput-bound at this size.
#include <math.h>

#include <new>
#include <vector>

#include "av_common.h"

namespace {

constexpr int IMU_DIM = 21;

// The batched filter's kernels are short and on the step's critical path, and they share their CUs with the front-end's
// long-running throughput waves (other HIP streams): VALU issue is arbitrated by wave priority first, age second
// (MI355X_MICROARCH.md, "Two waves per SIMD" item 2), so these kernels raise their waves' priority once at entry.
#ifndef AV_FILTER_WAVE_PRIO
#define AV_FILTER_WAVE_PRIO 3
#endif
#define AV_FILTER_PRIO() __builtin_amdgcn_s_setprio(AV_FILTER_WAVE_PRIO)

// 1 / sqrt(d) for d > 0: the hardware estimate refined by two Newton steps (~1 ulp).  The diagonal block of a panel is a chain of eight
// square roots and eight reciprocals on ONE thread while 255 wait at the barrier: the library sqrt and the division are ~25 dependent
// instructions each, this is ~10 for both (sqrt(d) = d * rsqrt(d)).
__device__ __forceinline__ double av_rsqrt_f64(double d)
{
    // (pivots below 1e-280 -- the 1e-300 floor of the Gram regularisation, or a denormal -- are scaled into range first: the hardware
    //  estimate of a denormal is not usable and y * y would overflow)
    const bool tiny = d < 1e-280;
    const double ds = tiny ? d * 0x1p200 : d;
    double y = __builtin_amdgcn_rsq(ds);
    const double h = 0.5 * ds;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return tiny ? y * 0x1p100 : y;
}
__device__ __forceinline__ void quat_to_rot(const double* qin, double* R)
{
    // utils.py:12-23: normalise, R = (2w^2-1) I - 2w [v]x + 2 v v^T
    double n = sqrt(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
    double x = qin[0] / n, y = qin[1] / n, z = qin[2] / n, w = qin[3] / n;
    double c = 2 * w * w - 1;
    R[0] = c + 2 * x * x;          R[1] = 2 * w * z + 2 * x * y;   R[2] = -2 * w * y + 2 * x * z;
    R[3] = -2 * w * z + 2 * y * x; R[4] = c + 2 * y * y;           R[5] = 2 * w * x + 2 * y * z;
    R[6] = 2 * w * y + 2 * z * x;  R[7] = -2 * w * x + 2 * z * y;  R[8] = c + 2 * z * z;
}

// Sum over the 64 lanes, returned to every lane.  Four DPP butterfly steps inside each 16-lane row (two v_mov_dpp on
// the halves + one v_add_f64 each), then the four row sums are read through SGPRs -- an order of magnitude less latency
// than six dependent ds_bpermute round trips, and this reduction sits on the critical path of every Householder step.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lane_f64(double v, int lane)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);      // row_half_mirror
    v += dpp_f64<0x140>(v);      // row_mirror
    return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}
// sum over the 16 lanes of a DPP row, returned to each of them (teams of 16 lanes: four independent sums per wavefront)
__device__ __forceinline__ double row16_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}

// sum over the 8 / 4 lanes of a team that is a fraction of a DPP row
__device__ __forceinline__ double row8_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    return v;
}
__device__ __forceinline__ double quad_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    return v;
}

// ================================================================================================
// Triangulation: one wavefront per feature, lane = view (obs j, camera c) with view = 2j + c.
// ================================================================================================
struct TriArgs {
    int n_feat;
    const int* obs_off;          // [n_feat + 1]
    const int* obs_cam;          // camera-state index of every observation
    const double* obs_z;         // [.][4] = u0 v0 u1 v1
    const double* cam_q;         // [n_cam][4]   (+ feat_stream[f] * cam_stride entries when batched over streams)
    const double* cam_p;         // [n_cam][3]
    const int* feat_stream;      // stream of every feature, or NULL (single filter)
    int cam_stride;              // camera slots per stream
    double R01[9], t01[3];       // cam0 -> cam1 (config.T_cn_cnm1)
    double huber, precision, damping;
    int outer_max, inner_max;
    double* out_pos;             // [n_feat][3] world position
    int* out_valid;              // [n_feat]
    // device-resident filter (msckf_dev.inc): the features to triangulate are list[0 .. *n_list_dev) (indices into the per-feature
    // arrays above; the stream of feature f is f / fs_stride), the kernel strides over the list whatever its grid; a valid position
    // is also stored in the stream's feature table (camera pruning: the feature lives on), slot f_mslot[f] of stream f / fs_stride
    const int* list; const int* n_list_dev; int fs_stride;
    const int* f_mslot; double* meta_pos; int* meta_init; int meta_stride;
};

__device__ __forceinline__ void solve3(const double A[9], const double b[3], double x[3])
{
    // Gaussian elimination with partial pivoting (what np.linalg.solve's gesv does)
    double M[3][4] = {{A[0], A[1], A[2], b[0]}, {A[3], A[4], A[5], b[1]}, {A[6], A[7], A[8], b[2]}};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int piv = k;
        double best = fabs(M[k][k]);
#pragma unroll
        for (int i = k + 1; i < 3; ++i) if (fabs(M[i][k]) > best) { best = fabs(M[i][k]); piv = i; }
        if (piv != k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { double t = M[k][j]; M[k][j] = M[piv][j]; M[piv][j] = t; }
#pragma unroll
        for (int i = k + 1; i < 3; ++i) {
            double f = M[i][k] / M[k][k];
#pragma unroll
            for (int j = k; j < 4; ++j) M[i][j] -= f * M[k][j];
        }
    }
    x[2] = M[2][3] / M[2][2];
    x[1] = (M[1][3] - M[1][2] * x[2]) / M[1][1];
    x[0] = (M[0][3] - M[0][1] * x[1] - M[0][2] * x[2]) / M[0][0];
}

__device__ __forceinline__ void triangulate_one(const TriArgs& a, int f, int lane)
{
    const int o0 = a.obs_off[f], M = a.obs_off[f + 1] - o0;
    const int nv = 2 * M;                                   // views; host guarantees nv <= 64
    const bool act = lane < nv;
    const int j = act ? (lane >> 1) : 0, cam = lane & 1;

    // pose of this view (camera -> world), feature_position_initializer.py:19-26
    double Rv[9], tv[3], z[2] = {0, 0};
    {
        const int ci = a.obs_cam[o0 + j] + (a.feat_stream ? a.feat_stream[f] * a.cam_stride : (a.fs_stride > 0 ? (f / a.fs_stride) * a.cam_stride : 0));
        double Rwc[9];
        quat_to_rot(a.cam_q + 4 * ci, Rwc);                 // world -> cam0
        double R0[9];                                       // cam0 -> world = Rwc^T
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) R0[r * 3 + c] = Rwc[c * 3 + r];
        const double* p = a.cam_p + 3 * ci;
        if (cam == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) Rv[i] = R0[i];
            tv[0] = p[0]; tv[1] = p[1]; tv[2] = p[2];
        } else {
            // cam1 pose = cam0 * inverse(T_cam0_cam1): R = R0 * R01^T, t = R0 * (-R01^T t01) + p
            double t10[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) t10[r] = -(a.R01[0 * 3 + r] * a.t01[0] + a.R01[1 * 3 + r] * a.t01[1] + a.R01[2 * 3 + r] * a.t01[2]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) Rv[r * 3 + c] = R0[r * 3 + 0] * a.R01[c * 3 + 0] + R0[r * 3 + 1] * a.R01[c * 3 + 1] + R0[r * 3 + 2] * a.R01[c * 3 + 2];
                tv[r] = R0[r * 3 + 0] * t10[0] + R0[r * 3 + 1] * t10[1] + R0[r * 3 + 2] * t10[2] + p[r];
            }
        }
        z[0] = a.obs_z[4 * (o0 + j) + 2 * cam];
        z[1] = a.obs_z[4 * (o0 + j) + 2 * cam + 1];
    }
    // first view's pose (T_c0_w) broadcast, then relative pose  inv(pose_v) * pose_0
    double R00[9], t00[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R00[i] = __shfl(Rv[i], 0, 64);
#pragma unroll
    for (int i = 0; i < 3; ++i) t00[i] = __shfl(tv[i], 0, 64);
    double R[9], t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = Rv[0 * 3 + r] * R00[0 * 3 + c] + Rv[1 * 3 + r] * R00[1 * 3 + c] + Rv[2 * 3 + r] * R00[2 * 3 + c];
        t[r] = Rv[0 * 3 + r] * (t00[0] - tv[0]) + Rv[1 * 3 + r] * (t00[1] - tv[1]) + Rv[2 * 3 + r] * (t00[2] - tv[2]);
    }

    // two-view initial guess from views 0 and 1 (feature_depth_estimator.py:4-14)
    double x[3];
    {
        double R1[9], t1[3], z0[2], z1[2];
#pragma unroll
        for (int i = 0; i < 9; ++i) R1[i] = __shfl(R[i], 1, 64);
#pragma unroll
        for (int i = 0; i < 3; ++i) t1[i] = __shfl(t[i], 1, 64);
        z0[0] = __shfl(z[0], 0, 64); z0[1] = __shfl(z[1], 0, 64);
        z1[0] = __shfl(z[0], 1, 64); z1[1] = __shfl(z[1], 1, 64);
        double m[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) m[r] = R1[r * 3 + 0] * z0[0] + R1[r * 3 + 1] * z0[1] + R1[r * 3 + 2];
        double a0 = m[0] - z1[0] * m[2], a1 = m[1] - z1[1] * m[2];
        double b0 = z1[0] * t1[2] - t1[0], b1 = z1[1] * t1[2] - t1[1];
        double depth = (a0 * b0 + a1 * b1) / (a0 * a0 + a1 * a1);
        double p0[3] = {z0[0] * depth, z0[1] * depth, depth};
        x[0] = p0[0] / p0[2]; x[1] = p0[1] / p0[2]; x[2] = 1.0 / p0[2];
    }

    auto cost_of = [&](const double* xx) -> double {
        double h0 = R[0] * xx[0] + R[1] * xx[1] + R[2] + xx[2] * t[0];
        double h1 = R[3] * xx[0] + R[4] * xx[1] + R[5] + xx[2] * t[1];
        double h2 = R[6] * xx[0] + R[7] * xx[1] + R[8] + xx[2] * t[2];
        double d0 = h0 / h2 - z[0], d1 = h1 / h2 - z[1];
        return wave_sum_f64(act ? d0 * d0 + d1 * d1 : 0.0);
    };

    double lam = a.damping;
    int outer = 0, inner = 0;
    double delta_norm = INFINITY;
    double total = cost_of(x);
    while (outer < a.outer_max && delta_norm > a.precision) {
        // J (2x3), r (2), Huber weight (feature_observation.py:14-39)
        double h0 = R[0] * x[0] + R[1] * x[1] + R[2] + x[2] * t[0];
        double h1 = R[3] * x[0] + R[4] * x[1] + R[5] + x[2] * t[1];
        double h2 = R[6] * x[0] + R[7] * x[1] + R[8] + x[2] * t[2];
        double W0[3] = {R[0], R[1], t[0]}, W1[3] = {R[3], R[4], t[1]}, W2[3] = {R[6], R[7], t[2]};
        double J0[3], J1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            J0[c] = W0[c] / h2 - W2[c] * h0 / (h2 * h2);
            J1[c] = W1[c] / h2 - W2[c] * h1 / (h2 * h2);
        }
        double r0 = h0 / h2 - z[0], r1 = h1 / h2 - z[1];
        double e = sqrt(r0 * r0 + r1 * r1);
        double w = (e <= a.huber) ? 1.0 : a.huber / (2 * e);
        double ww = (w == 1.0) ? 1.0 : w * w;
        if (!act) ww = 0.0;
        double A[9], b[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) A[r * 3 + c] = wave_sum_f64(ww * (J0[r] * J0[c] + J1[r] * J1[c]));
            b[r] = wave_sum_f64(ww * (J0[r] * r0 + J1[r] * r1));
        }
        bool reduced = false;
        while (inner < a.inner_max && !reduced) {
            double Ad[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) Ad[i] = A[i] + ((i % 4 == 0) ? lam : 0.0);
            double d[3];
            solve3(Ad, b, d);
            double xn[3] = {x[0] - d[0], x[1] - d[1], x[2] - d[2]};
            delta_norm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            double cn = cost_of(xn);
            if (cn < total) {
                reduced = true;
                x[0] = xn[0]; x[1] = xn[1]; x[2] = xn[2];
                total = cn;
                lam = fmax(lam / 10., 1e-10);
            } else {
                lam = fmin(lam * 10., 1e12);
            }
            ++inner;
        }
        ++outer;
    }
    double pf[3] = {x[0] / x[2], x[1] / x[2], 1.0 / x[2]};
    double depth_v = R[6] * pf[0] + R[7] * pf[1] + R[8] * pf[2] + t[2];
    unsigned long long bad = __ballot(act && !(depth_v > 0));
    if (lane == 0) {
        // position in the world: T_c0_w.R @ pf + T_c0_w.t (this lane is view 0, so Rv/tv = T_c0_w)
        double pw[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { pw[r] = Rv[r * 3 + 0] * pf[0] + Rv[r * 3 + 1] * pf[1] + Rv[r * 3 + 2] * pf[2] + tv[r]; a.out_pos[3 * f + r] = pw[r]; }
        const int ok = bad == 0ull ? 1 : 0;
        a.out_valid[f] = ok;
        if (a.meta_pos) {                                   // feature_position_initializer.py:72-76: position and is_initialized of the Feature
            const size_t ms = (size_t)(f / a.fs_stride) * a.meta_stride + a.f_mslot[f];
            a.meta_pos[3 * ms] = pw[0]; a.meta_pos[3 * ms + 1] = pw[1]; a.meta_pos[3 * ms + 2] = pw[2];
            a.meta_init[ms] = ok;
        }
    }
}
__global__ __launch_bounds__(256) void triangulate_kernel(TriArgs a)
{
    AV_FILTER_PRIO();
    const int lane = threadIdx.x & 63, wpb = (int)blockDim.x >> 6;      // wavefronts per workgroup: 4, or 1 (device-resident path)
    if (a.list) {                                           // wave-uniform loop: a wavefront per listed feature
        const int n = *a.n_list_dev;
        for (int t = blockIdx.x * wpb + (threadIdx.x >> 6); t < n; t += gridDim.x * wpb) triangulate_one(a, a.list[t], lane);
        return;
    }
    const int f = blockIdx.x * wpb + (threadIdx.x >> 6);
    if (f >= a.n_feat) return;
    triangulate_one(a, f, lane);
}

// ================================================================================================
// Per-feature Jacobian, null-space projection and chi^2 gate: one 256-thread workgroup per feature.
// ================================================================================================
struct FeatArgs {
    int n_feat, n_cam, ld;               // ld = leading dimension of P and of the output rows (>= 21 + 6 n_cam)
    const int* obs_off; const int* obs_cam; const double* obs_z;
    const double* pos;                   // [n_feat][3]
    const int* dof;                      // chi^2 degrees of freedom per feature (msckf.py:662,761)
    const int* row_off;                  // first output row of each feature (prefix of 4M-3)
    const double* cam_q; const double* cam_p; const double* cam_qn; const double* cam_pn;
    const double* P;                     // [n][ld]
    const double* chi2;                  // chi2.ppf(0.05, dof), index dof (1..99)
    double R01[9], t01[3], gravity[3], obs_noise;
    double* Hout;                        // [rows][ld]
    double* rout;                        // [rows]
    double* gamma; int* pass;            // [n_feat]
    int Mmax;
    // batching over streams (all NULL / 0 for a single filter)
    const int* feat_stream;              // stream of every feature
    const int* stream_ncam;              // camera states per stream
    const double* stream_gravity;        // [S][3]
    int cam_stride;                      // camera slots per stream in cam_q / cam_p / cam_qn / cam_pn
    size_t p_stride, h_stride, r_stride; // elements between streams in P, Hout, rout
    const int* feat_list;                // optional: team t processes feature feat_list[t] (launch buckets by track length)
    int n_list;                          // teams to run in this launch
    int team_doubles;                    // LDS doubles per team (wavefront teams: four per workgroup)
    unsigned long long* prof;            // optional [8] phase stamps of team 0 (diagnostic runs, AV_MSCKF_TIMING)
    int compact;                         // > 0: the output rows hold ONLY the feature's own 6 M camera columns, `compact` doubles per row (the pruning phase of the
                                         // device-resident filter: every candidate has the same two cameras, and the update gathers exactly these columns)
    int zero_fill;                       // 1: clear the full-width output rows first (columns of cameras the feature was not seen
                                         // from must read as zero); 0: the caller gathers only this feature's own camera columns
    // chained launches (batched filter): a feature triangulated by the triangulate_kernel launch just ahead on the same HIP
    // stream takes its position straight from that kernel's output (no host round trip between the two); a feature whose
    // triangulation failed is not evaluated and reads pass = 0 (msckf.py:646-656, 749-757: it is dropped before the Jacobian)
    const int* tri_idx;                  // [n_feat] index into tri_pos / tri_valid, or -1 = position known (`pos`); NULL = none
    const double* tri_pos; const int* tri_valid;
    // device-resident filter (msckf_dev.inc): teams stride over feat_list[0 .. *n_list_dev) whatever the grid; valid[f] = 0 marks a
    // feature without a usable position (triangulation or check_motion failed: not evaluated, pass = 0); stream of f = f / fs_stride
    const int* n_list_dev; const int* valid; int fs_stride;
};

// dynamic LDS of feature_kernel for tracks of at most Mx observations (layout at the top of the kernel)
__host__ __device__ static inline size_t feature_lds_bytes(int Mx, int team = 256)
{
    const size_t red = team == 256 ? 256 : 0;           // tree-reduction scratch of the workgroup teams only
    // Hb [4Mx][6] + alpha [3][6Mx] + Hf [4Mx][3] + r [4Mx] + S [4Mx][4Mx+1] (+ reduction scratch) + camera indices
    return sizeof(double) * ((size_t)(4 * Mx) * 6 + 3 * (size_t)(6 * Mx) + (4 * Mx) * 3 + 4 * Mx + (size_t)(4 * Mx) * (4 * Mx + 1) + red) + sizeof(int) * Mx + 16;
}

// TEAM = threads that cooperate on one feature: a whole 256-thread workgroup for long tracks, one wavefront (four
// features per workgroup, no workgroup barriers) for tracks of at most 4 observations -- the two-camera prune of
// msckf.py:714-800 produces ~300 two-observation features per stream and frame, which are latency- not work-bound.
// MEM: the lanes also hand GLOBAL memory to each other across this point (the zero fill of the output rows must have landed before
// other lanes overwrite parts of it).  Without it a team inside one wavefront only needs its LDS traffic ordered -- a wavefront's LDS
// operations execute in issue order -- so the fence is wavefront-scoped and costs no s_waitcnt: the workgroup-scoped form drains
// vmcnt as well, i.e. every phase boundary after the output rows were stored waited for those stores to be acknowledged.
template <int TEAM, bool MEM = false>
__device__ __forceinline__ void team_sync()
{
    if (TEAM == 256) {
        if (MEM) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS hand-over only: no vmcnt(0), stores stay in flight
    } else if (MEM) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int TEAM>
__device__ __forceinline__ void feature_one(const FeatArgs& a, const int slot, double* sm_all);

template <int TEAM>
__device__ __forceinline__ void feature_body(const FeatArgs& a)
{
    AV_FILTER_PRIO();
    extern __shared__ double sm_all[];
    const int tpb = (int)blockDim.x / TEAM;               // teams per workgroup (the launch picks the workgroup size: 256 threads, 128 for 4-lane teams)
    const int slot0 = TEAM == 256 ? (int)blockIdx.x : (int)(blockIdx.x * tpb + threadIdx.x / TEAM);
    if (a.n_list_dev) {                                   // device-side count: every team strides over the list (team-uniform trip count)
        const int n = *a.n_list_dev, stride = (int)gridDim.x * tpb;
        for (int slot = slot0; slot < n; slot += stride) { feature_one<TEAM>(a, slot, sm_all); team_sync<TEAM>(); }
        return;
    }
    if (slot0 >= a.n_list) return;                        // whole team
    feature_one<TEAM>(a, slot0, sm_all);
}

template <int TEAM>
__device__ __forceinline__ void feature_one(const FeatArgs& a, const int slot, double* sm_all)
{
    const int tid = threadIdx.x % TEAM;
    double* sm = sm_all + (TEAM == 256 ? 0 : (threadIdx.x / TEAM) * a.team_doubles);
    const int f = a.feat_list ? a.feat_list[slot] : slot;
    const int sidx = a.feat_stream ? a.feat_stream[f] : (a.fs_stride > 0 ? f / a.fs_stride : 0);
    const int cam0 = sidx * a.cam_stride;
    const double* Pm = a.P + sidx * a.p_stride;
    double* Hout = a.Hout + sidx * a.h_stride;
    double* rout = a.rout + sidx * a.r_stride;
    const double* grav = a.stream_gravity ? a.stream_gravity + 3 * sidx : a.gravity;
    // 16-lane teams only ever see two-observation features (the pruning candidates: one observation from each of the two cameras
    // that go; launch_feature_kernel picks them for Mmax <= 2): M is a compile-time constant there and every loop below unrolls
    const int o0 = a.obs_off[f], M = TEAM <= 16 ? 2 : a.obs_off[f + 1] - o0;
    const int R4 = 4 * M, C6 = 6 * M, K = R4 - 3;
    const int Mx = TEAM <= 16 ? 2 : a.Mmax;
    const int tri = a.tri_idx ? a.tri_idx[f] : -1;
    if (tri == -2 || (tri >= 0 && !a.tri_valid[tri]) || (a.valid && !a.valid[f]) || (TEAM <= 16 && a.obs_off[f + 1] - o0 != 2)) {   // -2: failed check_motion on the host                  // whole team (uniform in f): triangulation failed, nothing to gate
        if (tid == 0) { a.gamma[f] = 0.0; a.pass[f] = 0; }
        return;
    }
    // H_x is block diagonal (observation j: rows 4j..4j+3, its camera's 6 columns) and stays that way in LDS: only the blocks
    // are stored.  The null-space projection Q^T H_x (three Householder reflectors from H_f) is never materialised either --
    // column c of it is  H_x[:, c] - alpha0[c] u0 - alpha1[c] u1 - alpha2[c] u2  with three scalars per column (below), and
    // its rows go straight to global memory.  (A dense [4M][6M] copy was 85 KB of the 147 KB this kernel held at 21
    // observations, which kept whole CUs away from the front-end's kernels.)
    double* Hb = sm;                             // [4Mx][6]   H_x blocks: row 4j+r, column c of camera block j
    double* al = Hb + (4 * Mx) * 6;              // [3][6Mx]   reflector coefficients per column
    double* Hf = al + 3 * (6 * Mx);              // [4Mx][3]
    double* rr = Hf + (4 * Mx) * 3;              // [4Mx]
    double* S = rr + 4 * Mx;                     // [4Mx][4Mx+1]  (odd row pitch: row-wise reflector jobs stay off the same banks)
    double* red = S + (4 * Mx) * (4 * Mx + 1);   // [256] reduction scratch
    const int SP = R4 + 1;
    int* cidx = reinterpret_cast<int*>(red + (TEAM == 256 ? 256 : 0));   // [Mx] camera index per observation

    for (int i = tid; i < M; i += TEAM) cidx[i] = a.obs_cam[o0 + i];
    team_sync<TEAM>();
    if (a.prof && slot == 0 && tid == 0) a.prof[0] = __builtin_amdgcn_s_memrealtime();

    // ---- measurement_jacobian per observation (msckf.py:443-507): thread j < M ---------------------
    if (tid < M) {
        const int j = tid, ci = cam0 + a.obs_cam[o0 + j];
        double Rw0[9], Rw1[9];
        quat_to_rot(a.cam_q + 4 * ci, Rw0);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Rw1[r * 3 + c] = a.R01[r * 3 + 0] * Rw0[0 * 3 + c] + a.R01[r * 3 + 1] * Rw0[1 * 3 + c] + a.R01[r * 3 + 2] * Rw0[2 * 3 + c];
        const double* t0 = a.cam_p + 3 * ci;
        double t1[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) t1[r] = t0[r] - (Rw1[0 * 3 + r] * a.t01[0] + Rw1[1 * 3 + r] * a.t01[1] + Rw1[2 * 3 + r] * a.t01[2]);
        const double* pw = tri >= 0 ? a.tri_pos + 3 * tri : a.pos + 3 * f;
        double d0[3] = {pw[0] - t0[0], pw[1] - t0[1], pw[2] - t0[2]}, d1[3] = {pw[0] - t1[0], pw[1] - t1[1], pw[2] - t1[2]};
        double p0[3], p1[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            p0[r] = Rw0[r * 3] * d0[0] + Rw0[r * 3 + 1] * d0[1] + Rw0[r * 3 + 2] * d0[2];
            p1[r] = Rw1[r * 3] * d1[0] + Rw1[r * 3 + 1] * d1[1] + Rw1[r * 3 + 2] * d1[2];
        }
        // dz/dp (rows 0,1 from cam0, rows 2,3 from cam1)
        double dz[4][3] = {{1 / p0[2], 0, -p0[0] / (p0[2] * p0[2])}, {0, 1 / p0[2], -p0[1] / (p0[2] * p0[2])},
                           {1 / p1[2], 0, -p1[0] / (p1[2] * p1[2])}, {0, 1 / p1[2], -p1[1] / (p1[2] * p1[2])}};
        double sk[9] = {0, -p0[2], p0[1], p0[2], 0, -p0[0], -p0[1], p0[0], 0};
        double dp0[3][6], dp1[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dp0[r][c] = sk[r * 3 + c];
                dp0[r][3 + c] = -Rw0[r * 3 + c];
                dp1[r][c] = a.R01[r * 3 + 0] * sk[0 * 3 + c] + a.R01[r * 3 + 1] * sk[1 * 3 + c] + a.R01[r * 3 + 2] * sk[2 * 3 + c];
                dp1[r][3 + c] = -Rw1[r * 3 + c];
            }
        double A[4][6];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double(*dp)[6] = r < 2 ? dp0 : dp1;
                A[r][c] = dz[r][0] * dp[0][c] + dz[r][1] * dp[1][c] + dz[r][2] * dp[2][c];
            }
        // observability-constrained projection with the null-space states
        double Rn[9], u[6];
        quat_to_rot(a.cam_qn + 4 * ci, Rn);
        const double* g = grav;
#pragma unroll
        for (int r = 0; r < 3; ++r) u[r] = Rn[r * 3] * g[0] + Rn[r * 3 + 1] * g[1] + Rn[r * 3 + 2] * g[2];
        const double* pn = a.cam_pn + 3 * ci;
        double e[3] = {pw[0] - pn[0], pw[1] - pn[1], pw[2] - pn[2]};
        u[3] = -e[2] * g[1] + e[1] * g[2];
        u[4] = e[2] * g[0] - e[0] * g[2];
        u[5] = -e[1] * g[0] + e[0] * g[1];
        double uu = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) uu += u[c] * u[c];
        const double* zz = a.obs_z + 4 * (o0 + j);
        double zh[4] = {p0[0] / p0[2], p0[1] / p0[2], p1[0] / p1[2], p1[1] / p1[2]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double Au = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) Au += A[r][c] * u[c];
            double hx[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) { hx[c] = A[r][c] - Au * u[c] / uu; Hb[(4 * j + r) * 6 + c] = hx[c]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) Hf[(4 * j + r) * 3 + c] = -hx[3 + c];
            rr[4 * j + r] = zz[r] - zh[r];
        }
    }
    team_sync<TEAM>();

    if (a.prof && slot == 0 && tid == 0) a.prof[1] = __builtin_amdgcn_s_memrealtime();
    // ---- G = H_x Psub H_x^T (R4 x R4).  Before the projection H_x is block diagonal (one 4x6 block per observation),
    //      so G is M x M blocks  Hj P[cam j][cam l] Hl^T: one thread per block pair reads its 6x6 block of P once.  The gate
    //      matrix A^T H_x Psub H_x^T A (msckf.py:604-612 on the projected Jacobian) is then the trailing K x K block of
    //      Q^T G Q, with Q the product of the three reflectors below applied from both sides.
    for (int pr = tid; pr < M * M; pr += TEAM) {
        const int j = pr / M, l = pr - j * M;
        const double* Pb = Pm + (size_t)(IMU_DIM + 6 * cidx[j]) * a.ld + IMU_DIM + 6 * cidx[l];
        const double* Hj = Hb + (4 * j) * 6;
        const double* Hl = Hb + (4 * l) * 6;
        double t[4][6] = {{0}};
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double pv[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) pv[d] = Pb[(size_t)c * a.ld + d];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double h = Hj[r * 6 + c];
#pragma unroll
                for (int d = 0; d < 6; ++d) t[r][d] += h * pv[d];
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            double hl[6];
#pragma unroll
            for (int d = 0; d < 6; ++d) hl[d] = Hl[s2 * 6 + d];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double g = 0;
#pragma unroll
                for (int d = 0; d < 6; ++d) g += t[r][d] * hl[d];
                S[(4 * j + r) * SP + 4 * l + s2] = g;
            }
        }
    }
    team_sync<TEAM>();

    if (a.prof && slot == 0 && tid == 0) a.prof[2] = __builtin_amdgcn_s_memrealtime();
    // ---- left null space of H_f by 3 Householder reflectors, applied to H and r (msckf.py:540-544) --
    double v0s[3], taus[3];                          // reflector k: u_k = (0.., v0s[k] at row k, Hf[k+1.., k]), H_k = I - taus[k] u_k u_k^T
    for (int k = 0; k < 3; ++k) {
        // norm of Hf[k:, k]
        double part = 0;
        for (int i = k + tid; i < R4; i += TEAM) { double v = Hf[i * 3 + k]; part += v * v; }
        double nrm;
        if (TEAM == 256) {
            // wavefront sums by DPP, the four of them through LDS: two barriers (the 256-entry tree took nine per reflector)
            part = wave_sum_f64(part);
            if ((tid & 63) == 0) red[tid >> 6] = part;
            team_sync<TEAM>();
            nrm = sqrt((red[0] + red[1]) + (red[2] + red[3]));
            team_sync<TEAM>();
        } else {
            nrm = sqrt(TEAM == 64 ? wave_sum_f64(part) : (TEAM == 16 ? row16_sum_f64(part) : (TEAM == 8 ? row8_sum_f64(part) : quad_sum_f64(part))));
        }
        const double akk = Hf[k * 3 + k];
        const double alpha = akk >= 0 ? -nrm : nrm;
        const double v0 = akk - alpha;                     // v = x - alpha e1, stored in place in Hf[k:, k] (v0 kept apart)
        const double vtv = nrm * nrm - 2 * alpha * akk + alpha * alpha;      // |v|^2
        const double tau = vtv > 0 ? 2.0 / vtv : 0.0;
        v0s[k] = v0; taus[k] = tau;
        // columns to transform: the remaining Hf columns (k+1..2), r, and the R4 columns of G (H_x: see below)
        auto reflect = [&](double* col, int stride) {
            double dot = v0 * col[k * stride];
#pragma unroll 8
            for (int i = k + 1; i < R4; ++i) dot += Hf[i * 3 + k] * col[i * stride];        // unrolled: the LDS loads of 8 rows are in flight together
            dot *= tau;
            col[k * stride] -= dot * v0;
#pragma unroll 8
            for (int i = k + 1; i < R4; ++i) col[i * stride] -= dot * Hf[i * 3 + k];
        };
        const int nfix = (2 - k) + 1, njobs = nfix + R4;
        // workgroup teams (long tracks): FOUR lanes per column, each a quarter of the rows, the partial products summed inside the quad
        // (DPP): a column of 80 rows walked by one thread was 4 us per application with two thirds of the workgroup idle
        auto reflect4 = [&](double* col, int stride, int part) {
            const int len = R4 - k, chunk = (len + 3) >> 2, i0 = k + part * chunk, i1 = min(R4, i0 + chunk);
            double dot = 0;
#pragma unroll 5
            for (int i = i0; i < i1; ++i) dot += (i == k ? v0 : Hf[i * 3 + k]) * col[i * stride];       // unrolled: the LDS loads of five rows are in flight together
            dot = quad_sum_f64(dot) * tau;
#pragma unroll 5
            for (int i = i0; i < i1; ++i) col[i * stride] -= dot * (i == k ? v0 : Hf[i * 3 + k]);
        };
        if (TEAM == 256) {
            for (int job = tid >> 2; job < ((njobs + 63) & ~63); job += 64) {       // (whole quads run the DPP sum: the bound is rounded up)
                double* col = rr; int stride = 1;
                const bool on = job < njobs;
                if (on) {
                    if (job < 2 - k) { col = Hf + (k + 1 + job); stride = 3; }
                    else if (job < nfix) { col = rr; stride = 1; }
                    else { col = S + (job - nfix); stride = SP; }
                }
                if (on) reflect4(col, stride, tid & 3);
            }
        } else
        for (int job = tid; job < njobs; job += TEAM) {
            // one call site: the lanes of a wavefront differ only in (column, stride), not in control flow
            double* col; int stride;
            if (job < 2 - k) { col = Hf + (k + 1 + job); stride = 3; }
            else if (job < nfix) { col = rr; stride = 1; }
            else { col = S + (job - nfix); stride = SP; }
            reflect(col, stride);
        }
        team_sync<TEAM>();
        if (TEAM == 256) {
            for (int job = tid >> 2; job < R4; job += 64) reflect4(S + job * SP, 1, tid & 3);       // ... and G from the right
        } else
        for (int job = tid; job < R4; job += TEAM) reflect(S + job * SP, 1);       // ... and G from the right
        team_sync<TEAM>();
    }
    if (a.prof && slot == 0 && tid == 0) a.prof[3] = __builtin_amdgcn_s_memrealtime();
    // rows 3..R4-1 of H and r are A^T H_x and A^T r.  Write them out (dense row of width ld).
    // (Round 5: rows WITHOUT the zero fill -- three quarters of what a five-observation feature writes are zeros -- and readers that mask
    //  by the cameras of every stacked block were built, parity-green and measured: 5.87 / 5.91 against 5.97 / 5.87 ms of exclusive chain,
    //  156.9 / 157.5 against 156.7 / 156.4 k frames/s: within the noise, not kept.  profiles/r05/README.md)
    const int row0 = a.row_off[f];                 // < 0: gate only, nothing is stored (two-pass streams, msckf_batch.inc)
    if (row0 >= 0) {
        if (a.zero_fill) {
            for (int i = tid; i < K * a.ld; i += TEAM) {
                const int rI = i / a.ld, c = i - rI * a.ld;
                Hout[(size_t)(row0 + rI) * a.ld + c] = 0.0;
            }
            team_sync<TEAM, true>();
        }
        // Q^T H_x = H2 H1 H0 H_x column by column: with d_k = u_k^T col (four products: the column has four non-zeros),
        //   alpha0 = tau0 d0,  alpha1 = tau1 (d1 - alpha0 u1.u0),  alpha2 = tau2 (d2 - alpha0 u2.u0 - alpha1 u2.u1)
        // and rows >= 3 of u_k are Hf[., k] (the rows that are kept).
        auto uk = [&](int k, int i) -> double { return i < k ? 0.0 : (i == k ? v0s[k] : Hf[i * 3 + k]); };
        double g10 = 0, g20 = 0, g21 = 0;                     // u1.u0, u2.u0, u2.u1 (every thread: <= 3 x 84 LDS broadcasts)
        for (int i = 1; i < R4; ++i) {
            const double u0 = uk(0, i), u1 = uk(1, i), u2 = uk(2, i);
            g10 += u1 * u0; g20 += u2 * u0; g21 += u2 * u1;
        }
        for (int c = tid; c < C6; c += TEAM) {
            const int j = c / 6, cc = c - 6 * j;
            double d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double h = Hb[(4 * j + r) * 6 + cc];
                d0 += uk(0, 4 * j + r) * h; d1 += uk(1, 4 * j + r) * h; d2 += uk(2, 4 * j + r) * h;
            }
            const double a0 = taus[0] * d0;
            const double a1 = taus[1] * (d1 - a0 * g10);
            const double a2 = taus[2] * (d2 - a0 * g20 - a1 * g21);
            al[c] = a0; al[C6 + c] = a1; al[2 * C6 + c] = a2;
        }
        team_sync<TEAM>();
        for (int i = tid; i < K * C6; i += TEAM) {
            const int rI = i / C6, c = i - rI * C6, row = 3 + rI, j = c / 6;
            double v = (row >> 2) == j ? Hb[row * 6 + (c - 6 * j)] : 0.0;
            v -= al[c] * Hf[row * 3] + al[C6 + c] * Hf[row * 3 + 1] + al[2 * C6 + c] * Hf[row * 3 + 2];
            if (a.compact > 0) Hout[(size_t)(row0 + rI) * a.compact + c] = v;
            else Hout[(size_t)(row0 + rI) * a.ld + IMU_DIM + 6 * cidx[j] + (c - 6 * j)] = v;
        }
        for (int i = tid; i < K; i += TEAM) rout[row0 + i] = rr[3 + i];
    }

    if (a.prof && slot == 0 && tid == 0) a.prof[4] = __builtin_amdgcn_s_memrealtime();
    // ---- gating test (msckf.py:604-612): S = H' Psub H'^T + s^2 I = (Q^T G Q)[3:, 3:] + s^2 I, gamma = r'^T S^-1 r' ----
    double* Sg = S + 3 * SP + 3;                 // K x K, row pitch SP
    for (int i = tid; i < K; i += TEAM) Sg[i * SP + i] += a.obs_noise;
    team_sync<TEAM>();
    // gamma = r'^T S^-1 r' by a right-looking LDL^T that carries r' along as one more row: at step j the pivot d_j and
    // w_j = (L^-1 r')_j * sqrt(d_j) are final, gamma += w_j^2 / d_j, and the trailing block and the rest of w get their
    // rank-1 update.  One barrier per column, no square roots, no separate substitution pass (the single-thread forward
    // solve used to be half of this kernel's time on long tracks).
    double* wv = rr + 3;
    constexpr int GX = TEAM == 256 ? 16 : (TEAM == 64 ? 8 : (TEAM == 4 ? 2 : 4)), GY = TEAM / GX;
    const int tx = tid % GX, ty = tid / GX;
    double g = 0;
    if (TEAM == 256 && K > 16) {
        // Workgroup teams (long tracks, K up to 77): BLOCKED Cholesky of the bordered matrix [[S, r'], [r'^T, .]] -- the border
        // row comes out as y = L^-1 r' and gamma = y^T y.  Panels of 8 columns: the 8 x 8 diagonal block by one thread in
        // registers, the panel (and the border row) one row per thread, the trailing block by a 16 x 16 thread grid: three
        // barriers per eight columns where the column-by-column form below pays one per column, and at these sizes the
        // barrier is most of a step (60 us of a 150 us block at 17 observations).
        constexpr int NB = 8;
        for (int j0 = 0; j0 < K; j0 += NB) {
            const int nb = min(NB, K - j0), jb = j0 + nb;
            if (tid == 0) {
                double A[NB][NB];
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) A[i][j] = (i < nb) ? Sg[(j0 + i) * SP + j0 + j] : (i == j ? 1.0 : 0.0);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    double d = A[j][j];
#pragma unroll
                    for (int t = 0; t < j; ++t) d -= A[j][t] * A[j][t];
                    const double inv = av_rsqrt_f64(d);       // (two Newton steps on the hardware estimate: chol_packed_lds has the reasoning)
                    d = d * inv;
                    A[j][j] = d;
                    red[j] = inv;                              // reciprocal of the diagonal for the panel rows below
#pragma unroll
                    for (int i = j + 1; i < NB; ++i) {
                        double v = A[i][j];
#pragma unroll
                        for (int t = 0; t < j; ++t) v -= A[i][t] * A[j][t];
                        A[i][j] = v * inv;
                    }
                }
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) if (i < nb) Sg[(j0 + i) * SP + j0 + j] = A[i][j];
            }
            __syncthreads();
            {   // panel rows jb..K-1 and the border row (index K): x <- x L11^-T
                const int i = jb + tid;
                if (i <= K) {
                    double* xr = i < K ? Sg + i * SP + j0 : wv + j0;
                    double x[NB];
#pragma unroll
                    for (int t = 0; t < NB; ++t) x[t] = t < nb ? xr[t] : 0.0;
#pragma unroll
                    for (int t = 0; t < NB; ++t) if (t < nb) {
                        double v = x[t];
#pragma unroll
                        for (int u = 0; u < t; ++u) v -= x[u] * Sg[(j0 + t) * SP + j0 + u];
                        x[t] = v * red[t];
                    }
#pragma unroll
                    for (int t = 0; t < NB; ++t) if (t < nb) xr[t] = x[t];
                }
            }
            __syncthreads();
            if (jb < K) {
                // trailing block, border row included (i == K); two rows (i, i + GY) per thread and pass share the panel entries of a
                // column read from LDS (chol_packed_lds: the factorisation is bound by LDS traffic)
                for (int i = jb + ty; i <= K; i += 2 * GY) {
                    const int i2 = i + GY;
                    const bool two = i2 <= K;
                    const double* li = i < K ? Sg + i * SP + j0 : wv + j0;
                    const double* li2 = i2 < K ? Sg + i2 * SP + j0 : wv + j0;
                    double lr[NB], lr2[NB];
#pragma unroll
                    for (int t = 0; t < NB; ++t) { lr[t] = t < nb ? li[t] : 0.0; lr2[t] = (two && t < nb) ? li2[t] : 0.0; }
                    const int ilast = two ? i2 : i, cmax = ilast < K ? ilast : K - 1;     // the border row spans columns jb .. K-1
                    for (int c = jb + tx; c <= cmax; c += GX) {
                        double lc[NB];
#pragma unroll
                        for (int t = 0; t < NB; ++t) lc[t] = t < nb ? Sg[c * SP + j0 + t] : 0.0;
                        if (i == K || c <= i) {
                            double* dst = i < K ? Sg + i * SP + c : wv + c;
                            double v = *dst;
#pragma unroll
                            for (int t = 0; t < NB; ++t) v -= lr[t] * lc[t];
                            *dst = v;
                        }
                        if (two) {                                    // (c <= cmax covers row i2's columns exactly)
                            double* dst = i2 < K ? Sg + i2 * SP + c : wv + c;
                            double v = *dst;
#pragma unroll
                            for (int t = 0; t < NB; ++t) v -= lr2[t] * lc[t];
                            *dst = v;
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (tid == 0) { for (int k = 0; k < K; ++k) g += wv[k] * wv[k]; }
    } else
    for (int k = 0; k < K; ++k) {
        const double inv = 1.0 / Sg[k * SP + k];
        const double wk = wv[k], zk = wk * inv;
        g += wk * zk;
        for (int i = k + 1 + ty; i < K; i += GY) {
            const double aik = Sg[i * SP + k];
            const double lik = aik * inv;
#pragma unroll 4
            for (int jj = k + 1 + tx; jj <= i; jj += GX) Sg[i * SP + jj] -= lik * Sg[jj * SP + k];
            if (tx == 0) wv[i] -= aik * zk;
        }
        team_sync<TEAM>();
    }
    if (tid == 0) {
        a.gamma[f] = g;
        a.pass[f] = g < a.chi2[a.dof[f]] ? 1 : 0;
        if (a.prof && slot == 0) a.prof[5] = __builtin_amdgcn_s_memrealtime();
    }
}

// Long tracks: one workgroup per feature.  Short tracks: one wavefront per feature; those are latency-bound chains of
// small fp64 steps, so the register budget is capped for 8 waves per SIMD (the Jacobian phase spills, but only its two
// active lanes touch scratch) -- twice the features in flight per CU.
template <int TEAM> __global__ __launch_bounds__(256) void feature_kernel(FeatArgs a);
template <> __global__ __launch_bounds__(256, 4) void feature_kernel<256>(FeatArgs a) { feature_body<256>(a); }
template <> __global__ __attribute__((amdgpu_waves_per_eu(4, 8))) __launch_bounds__(256) void feature_kernel<64>(FeatArgs a) { feature_body<64>(a); }
template <> __global__ __attribute__((amdgpu_waves_per_eu(4, 8))) __launch_bounds__(256) void feature_kernel<16>(FeatArgs a) { feature_body<16>(a); }
// The device-resident filter's build of the 16-lane kernel: 207 registers, nothing spilled, two waves per SIMD.  Measured per 2,048-stream
// launch (round 4, profiles/r04/README.md): 564 us, against 688 us at 128 registers (256 B of scratch per lane), 814 at 96, 920 at 80 --
// PMC showed 73 % of the wave cycles waiting on memory, and the spill traffic was part of what they waited for.
// eight and four lanes per two-observation feature (8 / 16 features per wavefront): the Jacobian phase keeps 2 lanes of a team busy and the
// gate-matrix blocks 4, whatever the team size -- smaller teams waste fewer lanes there and take more rounds in the later phases
__global__ __attribute__((amdgpu_waves_per_eu(2, 3))) __launch_bounds__(256) void feature_kernel8(FeatArgs a) { feature_body<8>(a); }
// (Held to 128 registers -- so that a wave fits the hole ONE retiring LK wave leaves on a SIMD instead of waiting for two -- this kernel
//  and upd_info_kernel were measured beside the front-end in round 5: 156.0 / 156.9 k against 157.5 k frames/s.  profiles/r05/README.md)

static int msckf_lds_opt_in();       // raises the dynamic-LDS limit of every kernel below once per process

// Launch `cnt` features (a.feat_list or 0..cnt-1) whose tracks hold at most Mx observations.
static int launch_feature_kernel(FeatArgs a, int cnt, int Mx, hipStream_t st)
{
    int rc0 = msckf_lds_opt_in();
    if (rc0) return rc0;
    a.Mmax = Mx; a.n_list = cnt;
    // team size: 16 lanes (one DPP row, 16 features per workgroup) for the two-observation features of the prune path,
    // one wavefront up to 4 observations, one workgroup beyond.  (env: A/B and debugging aid, forces workgroup teams)
    const int team = getenv("AV_FEATURE_BLOCK_TEAMS") ? 256 : (Mx <= 2 ? 16 : (Mx <= 4 ? 64 : 256));
    const size_t per = (feature_lds_bytes(Mx, team) + 7) / 8 * 8;
    a.team_doubles = (int)(per / 8);
    const size_t lds = per * (256 / team);
    if (lds > 160 * 1024) { av_set_error("MSCKF feature blocks: %d observations per feature need %zu B of LDS", Mx, lds); return AV_E_CAPACITY; }
    if (team == 16) hipLaunchKernelGGL(feature_kernel<16>, dim3((cnt + 15) / 16), dim3(256), lds, st, a);
    else if (team == 64) hipLaunchKernelGGL(feature_kernel<64>, dim3((cnt + 3) / 4), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(feature_kernel<256>, dim3(cnt), dim3(256), lds, st, a);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

// ================================================================================================
// Covariance propagation for one IMU sample (msckf.py:275-339, covariance part), one workgroup.
// ================================================================================================
struct PropArgs {
    double* P; int n, ld;
    double dt;
    double gyro[3], acc[3];          // bias-corrected
    double q_old[4];                 // orientation before predict_new_state (for F, G)
    double q_new[4];                 // orientation after
    double q_null[4], v_null[3], p_null[3];
    double v_new[3], p_new[3];
    double gravity[3];
    double noise[4];                 // gyro, gyro_bias, acc, acc_bias (continuous)
};

// P11s / PhiT (LDS, both or neither): batched mode.  The IMU block of P lives in P11s across the samples of one frame
// and the transition matrices are accumulated into PhiT, so the cross block P12 <- Phi P12 is applied ONCE per frame
// with the product of the frame's transition matrices instead of once per IMU sample (same map; ten passes over the
// 21 x 6N block in global memory become one).
constexpr int PROP_LDS_DOUBLES = 5 * IMU_DIM * IMU_DIM + IMU_DIM * 12;      // F, F2, Phi, T, Q, G
// EXT: the six work matrices live in `ext` (PROP_LDS_DOUBLES doubles of LDS the caller also uses for something else at other times)
// NT: threads of the workgroup -- 256, or 64 (dk_cov: the device-resident filter runs a stream's covariance steps in ONE wavefront; the
// barriers below are then wavefront-local and the scalar preparation sits on neighbouring lanes instead of on the first lanes of
// three wavefronts)
template <bool EXT = false, int NT = 256>
__device__ __forceinline__ void propagate_body(const PropArgs& a, double* P11s = nullptr, double* PhiT = nullptr, double* ext = nullptr)
{
    __shared__ double own[EXT ? 1 : PROP_LDS_DOUBLES];
    double* const base = EXT ? ext : own;
    double* const F = base, * const F2 = F + IMU_DIM * IMU_DIM, * const Phi = F2 + IMU_DIM * IMU_DIM, * const T = Phi + IMU_DIM * IMU_DIM,
          * const Q = T + IMU_DIM * IMU_DIM, * const G = Q + IMU_DIM * IMU_DIM;
    __shared__ double Rwi[9], Rnull[9], Rnew[9], u[3], sv[3], w1[3], w2[3];
    const int tid = threadIdx.x;
    const int N = IMU_DIM;
    constexpr int Q1 = NT == 256 ? 64 : 1, Q2 = NT == 256 ? 128 : 2;          // lanes of the three quaternion conversions (0, Q1, Q2)
    constexpr int R1 = NT == 256 ? 64 : 9, R2 = NT == 256 ? 128 : 22, R3 = NT == 256 ? 192 : 43;      // first lanes of the side jobs (beside lanes 0 .. 8)
    for (int i = tid; i < N * N; i += NT) F[i] = 0.0;
    for (int i = tid; i < N * 12; i += NT) G[i] = 0.0;
    // the scalar preparation is spread over the first lanes of three wavefronts (it used to be lane 0's alone: with the back end batched
    // this serial stretch, not the 21 x 21 products, was most of a sample's time)
    if (tid == 0 || tid == Q1 || tid == Q2) quat_to_rot(tid == 0 ? a.q_old : (tid == Q1 ? a.q_null : a.q_new), tid == 0 ? Rwi : (tid == Q1 ? Rnull : Rnew));
    __syncthreads();
    if (tid < 9) {
        const int r = tid / 3, c = tid - 3 * r;
        const double* w = a.gyro; const double* ac = a.acc;
        const double skw[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        const double ska[9] = {0, -ac[2], ac[1], ac[2], 0, -ac[0], -ac[1], ac[0], 0};
        F[r * N + c] = -skw[r * 3 + c];
        F[r * N + 3 + c] = r == c ? -1.0 : 0.0;
        double s = 0;
        for (int k = 0; k < 3; ++k) s += Rwi[k * 3 + r] * ska[k * 3 + c];          // R^T skew(acc)
        F[(6 + r) * N + c] = -s;
        F[(6 + r) * N + 9 + c] = -Rwi[c * 3 + r];
        F[(12 + r) * N + 6 + c] = r == c ? 1.0 : 0.0;
        G[r * 12 + c] = r == c ? -1.0 : 0.0;
        G[(3 + r) * 12 + 3 + c] = r == c ? 1.0 : 0.0;
        G[(6 + r) * 12 + 6 + c] = -Rwi[c * 3 + r];
        G[(9 + r) * 12 + 9 + c] = r == c ? 1.0 : 0.0;
    } else if (tid == R1) {
        for (int r = 0; r < 3; ++r) u[r] = Rnull[r * 3] * a.gravity[0] + Rnull[r * 3 + 1] * a.gravity[1] + Rnull[r * 3 + 2] * a.gravity[2];
        double uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
        for (int r = 0; r < 3; ++r) sv[r] = u[r] / uu;
    } else if (tid == R2) {
        double dv[3] = {a.v_null[0] - a.v_new[0], a.v_null[1] - a.v_new[1], a.v_null[2] - a.v_new[2]};
        double dp[3] = {a.dt * a.v_null[0] + a.p_null[0] - a.p_new[0], a.dt * a.v_null[1] + a.p_null[1] - a.p_new[1], a.dt * a.v_null[2] + a.p_null[2] - a.p_new[2]};
        const double* g = a.gravity;
        w1[0] = -dv[2] * g[1] + dv[1] * g[2]; w1[1] = dv[2] * g[0] - dv[0] * g[2]; w1[2] = -dv[1] * g[0] + dv[0] * g[1];
        w2[0] = -dp[2] * g[1] + dp[1] * g[2]; w2[1] = dp[2] * g[0] - dp[0] * g[2]; w2[2] = -dp[1] * g[0] + dp[0] * g[1];
    }
    __syncthreads();
    // F, and with it Phi = I + F dt + (F dt)^2 / 2 + (F dt)^3 / 6, is block sparse (five 3 x 3 blocks of F; the rows of Phi hold 1 .. 11 non-zeros
    // of 21), and every product below reads both operands out of LDS -- at two LDS reads per multiply-add the dense products were
    // bound by LDS bandwidth (1.2 G multiply-adds per 2,048-stream step, half of dk_begin's time).  The zero patterns are read off the
    // matrices themselves (a bit mask per row / column) and the sums run over the non-zeros only, in the same ascending order: a
    // skipped term is an exact zero, so the results are bit-identical to the dense sums.
    __shared__ unsigned frow[IMU_DIM], fcol[IMU_DIM], prow[IMU_DIM];
    for (int i = tid; i < N * N; i += NT) F[i] *= a.dt;                       // Fdt
    if (tid >= R2 && tid < R2 + N) {                                            // (scaling by dt keeps the pattern; a dt of 0 only zeroes more)
        const int r = tid - R2; unsigned m = 0;
        for (int k = 0; k < N; ++k) if (F[r * N + k] != 0.0) m |= 1u << k;
        frow[r] = m;
    } else if (tid >= R3 && tid < R3 + N) {
        const int c = tid - R3; unsigned m = 0;
        for (int k = 0; k < N; ++k) if (F[k * N + c] != 0.0) m |= 1u << k;
        fcol[c] = m;
    }
    __syncthreads();
    for (int i = tid; i < N * N; i += NT) {                                    // Fdt^2
        int r = i / N, c = i - r * N; double s = 0;
        for (unsigned m = frow[r]; m; m &= m - 1) { const int k = __builtin_ctz(m); s += F[r * N + k] * F[k * N + c]; }
        F2[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < N * N; i += NT) {                                    // Fdt^3, Phi
        int r = i / N, c = i - r * N; double s = 0;
        for (unsigned m = fcol[c]; m; m &= m - 1) { const int k = __builtin_ctz(m); s += F2[r * N + k] * F[k * N + c]; }
        Phi[i] = (r == c ? 1.0 : 0.0) + F[i] + F2[i] / 2. + s / 6.;
    }
    __syncthreads();
    // two patches of Phi that do not meet (each thread reads and writes only its own entries of Phi):
    if (tid < 9) {                                                              // Phi[:3,:3] = R_new R_null^T
        int r = tid / 3, c = tid % 3; double s = 0;
        for (int k = 0; k < 3; ++k) s += Rnew[r * 3 + k] * Rnull[c * 3 + k];
        Phi[r * N + c] = s;
    } else if (tid >= R1 && tid < R1 + 6) {                                     // OC-KF corrections of rows 6..8 and 12..14
        const int t6 = tid - R1, blk = t6 / 3, r = t6 % 3;
        const int row = (blk == 0 ? 6 : 12) + r;
        const double* wv = blk == 0 ? w1 : w2;
        const double p0 = Phi[row * N], p1 = Phi[row * N + 1], p2 = Phi[row * N + 2];
        const double Au = p0 * u[0] + p1 * u[1] + p2 * u[2];
        const double d = Au - wv[r];
        Phi[row * N] = p0 - d * sv[0]; Phi[row * N + 1] = p1 - d * sv[1]; Phi[row * N + 2] = p2 - d * sv[2];
    } else if (tid >= R2 && tid < R2 + N) {
        // the pattern of the finished Phi: the entries the two patches rewrite at this moment (columns 0 .. 2 of rows 0 .. 2, 6 .. 8, 12 .. 14)
        // count as non-zero whatever they hold
        const int r = tid - R2; unsigned m = 0;
        for (int k = 0; k < N; ++k) if (Phi[r * N + k] != 0.0) m |= 1u << k;
        if (r < 3 || (r >= 6 && r < 9) || (r >= 12 && r < 15)) m |= 7u;
        prow[r] = m;
    }
    __syncthreads();
    // three products with the finished Phi, side by side: T = Phi G (21 x 12, for Q = Phi G Qc G^T Phi^T dt), F2 = Phi P11, and in
    // batched mode Q <- Phi PhiT (the frame's accumulated transition; Q is only a buffer here, the noise term is added below)
    for (int i = tid; i < N * 12; i += NT) {
        int r = i / 12, c = i - r * 12; double s = 0;
        for (unsigned m = prow[r]; m; m &= m - 1) { const int k = __builtin_ctz(m); s += Phi[r * N + k] * G[k * 12 + c]; }
        T[i] = s;
    }
    for (int i = tid; i < N * N; i += NT) {
        int r = i / N, c = i - r * N; double s = 0;
        if (P11s) { for (unsigned m = prow[r]; m; m &= m - 1) { const int k = __builtin_ctz(m); s += Phi[r * N + k] * P11s[k * N + c]; } }
        else      { for (unsigned m = prow[r]; m; m &= m - 1) { const int k = __builtin_ctz(m); s += Phi[r * N + k] * a.P[(size_t)k * a.ld + c]; } }
        F2[i] = s;                                                              // Phi P11
        if (P11s) {
            double s2 = 0;
            for (unsigned m = prow[r]; m; m &= m - 1) { const int k = __builtin_ctz(m); s2 += Phi[r * N + k] * PhiT[k * N + c]; }
            Q[i] = s2;
        }
    }
    __syncthreads();
    // P11 <- Phi P11 Phi^T + Q
    for (int i = tid; i < N * N; i += NT) {
        int r = i / N, c = i - r * N; double s = 0, q = 0;
        for (int k = 0; k < 12; ++k) q += T[r * 12 + k] * a.noise[k / 3] * T[c * 12 + k];
        q = q * a.dt;
        for (unsigned m = prow[c]; m; m &= m - 1) { const int k = __builtin_ctz(m); s += F2[r * N + k] * Phi[c * N + k]; }
        F[i] = s + q;
    }
    __syncthreads();
    if (P11s) {
        // batched: symmetrised IMU block back to LDS ((P + P^T)/2 per sample, msckf.py:334-335), PhiT <- Phi PhiT
        for (int i = tid; i < N * N; i += NT) {
            int r = i / N, c = i - r * N;
            PhiT[i] = Q[i];
            P11s[i] = (F[r * N + c] + F[c * N + r]) / 2.;
        }
        __syncthreads();
        return;
    }
    // P12 <- Phi P12 (computed from the OLD P12 into scratch rows of T in chunks), P21 = P12^T
    const int nc = a.n - N;
    for (int c0 = 0; c0 < nc; c0 += N) {
        const int w = min(N, nc - c0);
        for (int i = tid; i < N * w; i += NT) {
            int r = i / w, c = i - r * w; double s = 0;
            for (unsigned m = prow[r]; m; m &= m - 1) { const int k = __builtin_ctz(m); s += Phi[r * N + k] * a.P[(size_t)k * a.ld + N + c0 + c]; }
            T[i] = s;
        }
        __syncthreads();
        for (int i = tid; i < N * w; i += NT) {
            int r = i / w, c = i - r * w;
            a.P[(size_t)r * a.ld + N + c0 + c] = T[i];
            a.P[(size_t)(N + c0 + c) * a.ld + r] = T[i];
        }
        __syncthreads();
    }
    // write P11 symmetrised ((P + P^T)/2 of msckf.py:334-335; the cross blocks are exact transposes already)
    for (int i = tid; i < N * N; i += NT) {
        int r = i / N, c = i - r * N;
        a.P[(size_t)r * a.ld + c] = (F[r * N + c] + F[c * N + r]) / 2.;
    }
}

__global__ __launch_bounds__(256) void propagate_kernel(PropArgs a) { propagate_body(a); }
// batched: block b applies samples first[b] .. first[b+1]-1 in order (one stream per block)
__global__ __launch_bounds__(256) void propagate_batch_kernel(const PropArgs* arr, const int* first)
{
    AV_FILTER_PRIO();
    __shared__ double P11s[IMU_DIM * IMU_DIM], PhiT[IMU_DIM * IMU_DIM];
    const int i0 = first[blockIdx.x], i1 = first[blockIdx.x + 1];
    if (i0 >= i1) return;                                   // block-uniform
    const int tid = threadIdx.x, N = IMU_DIM;
    double* P = arr[i0].P; const int n = arr[i0].n, ld = arr[i0].ld;
    for (int i = tid; i < N * N; i += 256) { int r = i / N, c = i - r * N; P11s[i] = P[(size_t)r * ld + c]; PhiT[i] = r == c ? 1.0 : 0.0; }
    __syncthreads();
    for (int i = i0; i < i1; ++i) propagate_body(arr[i], P11s, PhiT);
    // cross block with the frame's accumulated transition: every element of the new block is formed in registers
    // from the old one before anything is overwritten
    const int nc = n - N, tot = N * nc;
    constexpr int PER = (IMU_DIM * 6 * 31 + 255) / 256;     // up to 31 camera states
    double acc[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + 256 * u;
        double s = 0;
        if (e < tot) {
            const int r = e / nc, c = e - r * nc;
            for (int k = 0; k < N; ++k) s += PhiT[r * N + k] * P[(size_t)k * ld + N + c];
        }
        acc[u] = s;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + 256 * u;
        if (e < tot) {
            const int r = e / nc, c = e - r * nc;
            P[(size_t)r * ld + N + c] = acc[u];
            P[(size_t)(N + c) * ld + r] = acc[u];
        }
    }
    for (int i = tid; i < N * N; i += 256) { int r = i / N, c = i - r * N; P[(size_t)r * ld + c] = P11s[i]; }
}

// ================================================================================================
// State augmentation (msckf.py:407-423)
// ================================================================================================
struct AugArgs { double* P; int n, ld; double R_ic[9]; double sk[9]; };     // sk = skew(R_w_i^T t_c_i)

template <int NT = 256>
__device__ __forceinline__ void augment_body(const AugArgs& a)
{
    __shared__ double J[6 * IMU_DIM];
    __shared__ double C[36];
    const int tid = threadIdx.x, N = IMU_DIM, n = a.n;
    for (int i = tid; i < 6 * N; i += NT) J[i] = 0.0;
    __syncthreads();
    if (tid < 9) {
        int r = tid / 3, c = tid % 3;
        J[r * N + c] = a.R_ic[tid];
        J[r * N + 15 + c] = r == c ? 1.0 : 0.0;
        J[(3 + r) * N + c] = a.sk[tid];
        J[(3 + r) * N + 12 + c] = r == c ? 1.0 : 0.0;
        J[(3 + r) * N + 18 + c] = r == c ? 1.0 : 0.0;
    }
    __syncthreads();
    // new rows: J P[:21, :n]
    for (int i = tid; i < 6 * n; i += NT) {
        int r = i / n, c = i - r * n; double s = 0;
        for (int k = 0; k < N; ++k) s += J[r * N + k] * a.P[(size_t)k * a.ld + c];
        a.P[(size_t)(n + r) * a.ld + c] = s;
    }
    __syncthreads();
    // corner: (J P11) J^T, then mirror the new rows into the new columns
    if (tid < 36) {
        int r = tid / 6, c = tid % 6; double s = 0;
        for (int k = 0; k < N; ++k) s += a.P[(size_t)(n + r) * a.ld + k] * J[c * N + k];
        C[tid] = s;
    }
    __syncthreads();
    for (int i = tid; i < 6 * n; i += NT) {
        int r = i / n, c = i - r * n;
        a.P[(size_t)c * a.ld + n + r] = a.P[(size_t)(n + r) * a.ld + c];
    }
    if (tid < 36) {
        int r = tid / 6, c = tid % 6;
        a.P[(size_t)(n + r) * a.ld + n + c] = (C[r * 6 + c] + C[c * 6 + r]) / 2.;
    }
}

__global__ __launch_bounds__(256) void augment_kernel(AugArgs a) { augment_body(a); }
__global__ __launch_bounds__(256) void augment_batch_kernel(const AugArgs* arr) { AV_FILTER_PRIO(); augment_body(arr[blockIdx.x]); }

// ================================================================================================
// Delete the 6 rows/cols of one camera state (msckf.py:774-786)
// ================================================================================================
struct RemArgs { double* P; double* scratch; int n, ld, start0, start1; };     // start1 < 0: only one removal

__device__ __forceinline__ void remove_cam_body(double* P, double* scratch, int n, int ld, int start)
{
    // compact into scratch, then copy back (single workgroup; n <= 147)
    const int tid = threadIdx.x, m = n - 6;
    for (int i = tid; i < m * m; i += (int)blockDim.x) {
        int r = i / m, c = i - r * m;
        int sr = r < start ? r : r + 6, sc = c < start ? c : c + 6;
        scratch[i] = P[(size_t)sr * ld + sc];
    }
    __syncthreads();
    for (int i = tid; i < m * m; i += (int)blockDim.x) {
        int r = i / m, c = i - r * m;
        P[(size_t)r * ld + c] = scratch[i];
    }
}
// Both removals of a pruning step in one pass: rows / columns [s0, s0 + 6) and [s1, s1 + 6) of the n x n matrix leave (s0 < s1, both in the
// numbering BEFORE the removal).  Entries above and left of s0 stay where they are and are not touched; the rest goes through the
// scratch copy once instead of twice (dk_finish is a memory-bound kernel: P is 170 KB per stream).
__device__ __forceinline__ void remove_two_cams_body(double* P, double* scratch, int n, int ld, int s0, int s1)
{
    const int tid = threadIdx.x, m = n - 12;
    for (int i = tid; i < m * m; i += (int)blockDim.x) {
        const int r = i / m, c = i - r * m;
        if (r < s0 && c < s0) continue;
        const int sr = r < s0 ? r : (r + 6 < s1 ? r + 6 : r + 12), sc = c < s0 ? c : (c + 6 < s1 ? c + 6 : c + 12);
        scratch[i] = P[(size_t)sr * ld + sc];
    }
    __syncthreads();
    for (int i = tid; i < m * m; i += (int)blockDim.x) {
        const int r = i / m, c = i - r * m;
        if (r < s0 && c < s0) continue;
        P[(size_t)r * ld + c] = scratch[i];
    }
}
__global__ __launch_bounds__(256) void remove_cam_kernel(double* P, double* scratch, int n, int ld, int start) { remove_cam_body(P, scratch, n, ld, start); }
// batched: two removals per stream, the second index already refers to the matrix after the first removal
__global__ __launch_bounds__(256) void remove_cam_batch_kernel(const RemArgs* arr)
{
    AV_FILTER_PRIO();
    const RemArgs a = arr[blockIdx.x];
    if (a.start0 < 0) return;
    remove_cam_body(a.P, a.scratch, a.n, a.ld, a.start0);
    if (a.start1 >= 0) { __syncthreads(); remove_cam_body(a.P, a.scratch, a.n - 6, a.ld, a.start1); }
}

// ================================================================================================
// Stacked measurement update (msckf.py:548-602), one 1024-thread workgroup.
// ================================================================================================
struct UpdArgs {
    double* P; int n, ld;
    const double* Hsrc; const double* rsrc;      // per-feature blocks produced by feature_kernel
    const int* blk_row; const int* blk_len; int n_blk;
    double* W;                                   // [ld+1][ldt] TRANSPOSED work copy of [H | r] (column c at W + c*ldt)
    int ldt;                                     // leading dimension of W (>= m)
    const int* cols; int nc;                     // the non-zero columns of the stacked Jacobian (ascending), device pointer
    unsigned long long* prof;                    // optional [16] phase timestamps (100 MHz ticks), diagnostic builds/runs only
    double* T;                                   // [k][ld]
    double* Kt;                                  // [k][ld]
    double* Pn;                                  // [n][ld]
    double* dx;                                  // [n]
    double obs_noise;
    int m;                                       // total stacked rows
    double* Sbuf;                                // batched back end: S = H P H^T + s^2 I, [k][ld] (lower triangle used)
    int mode;                                    // batched back end: 0 = Cholesky pipeline, 1 = information form (upd_info_kernel)
    int kdir;                                    // batched back end: > 0 = rows kept (no QR compression: the stacked rows come in chunks), 0 = upd_k(m, nc)
    int round;                                   // batched back end: chunk number of a sequential update (> 0: residual minus H dx so far, dx accumulates)
    int hld;                                     // batched back end: > 0 = Hsrc rows are COMPACT, hld doubles per row, column q of the row is state column cols[q] (FeatArgs::compact)
    int* status;                                 // batched back end: the stream's entry of StackArgs::stacked (NULL: single filter); a factorisation that meets a
                                                 // non-positive or non-finite pivot writes UPD_BAD_PIVOT there and turns the update into a no-op
};
constexpr int UPD_BAD_PIVOT = -2;

// The batched kernels read their UpdArgs from an array in memory, and a pointer that arrives that way is GENERIC to the compiler: every
// access through it becomes a FLAT load / store, which takes the slower flat path and counts on lgkmcnt (the LDS counter) as well as
// vmcnt.  UpdArgsG is the same record with its pointers typed as global memory (address space 1: global_load / global_store); the
// kernels below work on that view, their shared bodies are templates over the record type (a kernel may keep its UpdArgs in LDS).
template <typename T> using av_gptr = T __attribute__((address_space(1)))*;
struct UpdArgsG {
    av_gptr<double> P; int n, ld;
    av_gptr<const double> Hsrc; av_gptr<const double> rsrc;
    av_gptr<const int> blk_row; av_gptr<const int> blk_len; int n_blk;
    av_gptr<double> W; int ldt;
    av_gptr<const int> cols; int nc;
    av_gptr<unsigned long long> prof;
    av_gptr<double> T; av_gptr<double> Kt; av_gptr<double> Pn; av_gptr<double> dx;
    double obs_noise; int m;
    av_gptr<double> Sbuf;
    int mode, kdir, round, hld;
    av_gptr<int> status;
};
__device__ __forceinline__ UpdArgsG upd_load(const UpdArgs* __restrict__ arr, int i)
{
    const UpdArgs a = arr[i];
    UpdArgsG g;
    g.P = (av_gptr<double>)a.P; g.n = a.n; g.ld = a.ld; g.Hsrc = (av_gptr<const double>)a.Hsrc; g.rsrc = (av_gptr<const double>)a.rsrc;
    g.blk_row = (av_gptr<const int>)a.blk_row; g.blk_len = (av_gptr<const int>)a.blk_len; g.n_blk = a.n_blk;
    g.W = (av_gptr<double>)a.W; g.ldt = a.ldt; g.cols = (av_gptr<const int>)a.cols; g.nc = a.nc; g.prof = (av_gptr<unsigned long long>)a.prof;
    g.T = (av_gptr<double>)a.T; g.Kt = (av_gptr<double>)a.Kt; g.Pn = (av_gptr<double>)a.Pn; g.dx = (av_gptr<double>)a.dx;
    g.obs_noise = a.obs_noise; g.m = a.m; g.Sbuf = (av_gptr<double>)a.Sbuf; g.mode = a.mode; g.kdir = a.kdir; g.round = a.round; g.hld = a.hld;
    g.status = (av_gptr<int>)a.status;
    return g;
}

constexpr int UT = 1024;

// When is the stacked Jacobian QR-compressed to its nc non-zero columns?  The reference compresses when m > n
// (msckf.py:554); any thin QR leaves delta_x and P+ unchanged, so the choice is free.  A Householder QR costs nc
// dependent reflector steps, the back end scales with k = rows kept: compressing 130 rows to 114 costs more than it
// saves, compressing 1480 rows to 12 is the whole point.  k never exceeds 144 (S and T^T live in [ld][ld] buffers).
// (136, not 144: the packed triangle of a 136-row Cholesky is 74.6 KB, so TWO upd_chol workgroups share a CU's 160 KB of LDS;
//  the device-resident path sends streams with more than 136 stacked rows through the compression, msckf_dev_host.inc DEV_KCH.)
__host__ __device__ inline bool upd_compress(int m, int nc) { return m > nc && (2 * m > 3 * nc || m > 136); }
__host__ __device__ inline int upd_k(int m, int nc) { return upd_compress(m, nc) ? nc : m; }

__device__ __forceinline__ double block_sum(double v, double* red)
{
    const int tid = threadIdx.x;
    v = wave_sum_f64(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int i = 0; i < UT / 64; ++i) s += red[i];
    return s;
}


// Reflector scalars off the slow path: the owner's norm -> sqrt -> divide chain sits between two barriers of every
// Householder step, and the IEEE fp64 sqrt / divide sequences are long dependent chains.  v_rsq_f64 / v_rcp_f64 plus two
// Newton steps give the same values to an ulp or two, which is all a reflector needs: H = I - tau v v^T is orthogonal for
// tau = 2 / v^T v whatever the rounding of alpha; an inexact alpha only leaves rounding-level residue below the diagonal.
__device__ __forceinline__ void reflector_scalars(double nrm2, double ajj, double& alpha, double& v0, double& tau)
{
    double r = nrm2 > 0 ? __builtin_amdgcn_rsq(nrm2) : 0.0;
    r = r * (1.5 - 0.5 * nrm2 * r * r);
    r = r * (1.5 - 0.5 * nrm2 * r * r);
    const double nrm = nrm2 * r;
    alpha = ajj >= 0 ? -nrm : nrm;
    v0 = ajj - alpha;
    const double vtv = nrm2 - 2 * alpha * ajj + alpha * alpha;
    double t = vtv > 0 ? __builtin_amdgcn_rcp(vtv) : 0.0;
    t = t * (2.0 - vtv * t);
    t = t * (2.0 - vtv * t);
    tau = 2.0 * t;
}

// Thin Householder QR of the stacked [H | r] held entirely in registers.  Column c lives in wavefront c % 16 (slot
// c / 16), row i in lane i % 64 (slot i / 64): CPW x RPL doubles per lane, so 16*CPW >= nc+1 and 64*RPL >= m.
// Per reflector: the owner wavefront norms its column and publishes v (LDS, double buffered) and (v0, tau); after ONE
// barrier every wavefront applies it to its own columns without touching memory.  The generic path below keeps the
// columns in global memory (L2) and pays two dependent round trips per reflector.
template <int CPW, int RPL>
__device__ __forceinline__ void qr_in_registers(const UpdArgs& a, int m, int nc, const int* srcrow, double* vb0, double* vb1, double* qsc,
                                                double* Wt, size_t ldt, double* rcol)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double x[CPW][RPL];
#pragma unroll
    for (int l = 0; l < CPW; ++l) {
        const int c = wave + 16 * l;
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            const int row = lane + 64 * t;
            double v = 0.0;
            if (row < m && c <= nc) v = c < nc ? a.Hsrc[(size_t)srcrow[row] * a.ld + a.cols[c]] : a.rsrc[srcrow[row]];
            x[l][t] = v;
        }
    }
    __syncthreads();                                   // srcrow shares its LDS with the reflector buffers
    if (a.prof && tid == 0) a.prof[1] = __builtin_amdgcn_s_memrealtime();     // diagnostic: 'gather' ends once the columns sit in registers
    for (int j = 0; j < nc; ++j) {
        double* vb = (j & 1) ? vb1 : vb0;
        double* sc = qsc + 2 * (j & 1);
        if (wave == (j & 15)) {
            const int lj = j >> 4, tj = j >> 6;
            double part = 0, cand = 0;
#pragma unroll
            for (int l = 0; l < CPW; ++l) if (l == lj) {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const int row = lane + 64 * t;
                    if (row >= j) part += x[l][t] * x[l][t];
                    if (t == tj) cand = x[l][t];
                }
            }
            const double nrm2 = wave_sum_f64(part);
            const double ajj = __shfl(cand, j & 63, 64);
            double alpha, v0o, tauo;
            reflector_scalars(nrm2, ajj, alpha, v0o, tauo);
            if (lane == 0) { sc[0] = v0o; sc[1] = tauo; }
#pragma unroll
            for (int l = 0; l < CPW; ++l) if (l == lj) {
#pragma unroll
                for (int t = 0; t < RPL; ++t) {
                    const int row = lane + 64 * t;
                    if (row > j && row < m) vb[row] = x[l][t];
                    if (row == j) x[l][t] = alpha; else if (row > j) x[l][t] = 0.0;
                }
            }
        }
        __syncthreads();
        const double v0 = sc[0], tau = sc[1];
        auto vv = [&](int t) -> double { const int row = lane + 64 * t; return row == j ? v0 : ((row > j && row < m) ? vb[row] : 0.0); };
        if (RPL <= 8) {
            // few rows per lane: v stays in registers for all CPW columns (re-reading it per column makes the step LDS-bandwidth bound)
            double v[RPL <= 8 ? RPL : 1];
#pragma unroll
            for (int t = 0; t < (RPL <= 8 ? RPL : 1); ++t) v[t] = vv(t);
#pragma unroll
            for (int l = 0; l < CPW; ++l) {
                const int c = wave + 16 * l;
                if (c > j && c <= nc) {                    // wavefront-uniform
                    double dot = 0;
#pragma unroll
                    for (int t = 0; t < (RPL <= 8 ? RPL : 1); ++t) dot += v[t] * x[l][t];
                    dot = wave_sum_f64(dot) * tau;
#pragma unroll
                    for (int t = 0; t < (RPL <= 8 ? RPL : 1); ++t) x[l][t] -= dot * v[t];
                }
            }
        } else {
            // many rows, one or two columns per lane: v is re-read from LDS (keeping it next to x would double the footprint)
#pragma unroll
            for (int l = 0; l < CPW; ++l) {
                const int c = wave + 16 * l;
                if (c > j && c <= nc) {
                    double dot = 0;
#pragma unroll
                    for (int t = 0; t < RPL; ++t) dot += vv(t) * x[l][t];
                    dot = wave_sum_f64(dot) * tau;
#pragma unroll
                    for (int t = 0; t < RPL; ++t) x[l][t] -= dot * vv(t);
                }
            }
        }
    }
    // R (upper triangular, zeros below the diagonal) and Q^T r: the first nc rows of every column
#pragma unroll
    for (int l = 0; l < CPW; ++l) {
        const int c = wave + 16 * l;
        if (c <= nc) {
            double* dst = c < nc ? Wt + (size_t)c * ldt : rcol;
#pragma unroll
            for (int t = 0; t < RPL; ++t) { const int row = lane + 64 * t; if (row < nc) dst[row] = x[l][t]; }
        }
    }
    __syncthreads();
}



// sum over each 32-lane half of the wavefront, returned to the lanes of that half
__device__ __forceinline__ double half32_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    const double lo = lane_f64(v, 0) + lane_f64(v, 16), hi = lane_f64(v, 32) + lane_f64(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}

// The register-resident QR for the lost-feature shape (m <= 256 rows, up to 128 columns): HALF a wavefront per column.
// Column c lives in wavefront (c mod 32) / 2, half c & 1, slot c / 32 (4 slots); row i in lane i mod 32 of the half, slot
// i / 32 (8 slots): 32 doubles per lane.  One dot-product reduction then serves two columns, so a reflector costs each
// wavefront 4 reductions instead of the 8 of the one-wavefront-per-column mapping (the reductions are what a step is
// made of), and the 8 values of v a lane needs stay in registers.
__device__ __forceinline__ void qr_half32(const UpdArgs& a, int m, int nc, const int* srcrow, double* vb0, double* vb1, double* qsc,
                                          double* Wt, size_t ldt, double* rcol)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hf = lane >> 5, l32 = lane & 31;
    double x[4][8];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int c = 2 * wave + hf + 32 * l;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = l32 + 32 * t;
            double v = 0.0;
            if (row < m && c <= nc) v = c < nc ? a.Hsrc[(size_t)srcrow[row] * a.ld + a.cols[c]] : a.rsrc[srcrow[row]];
            x[l][t] = v;
        }
    }
    __syncthreads();                                   // srcrow shares its LDS with the reflector buffers
    if (a.prof && tid == 0) a.prof[1] = __builtin_amdgcn_s_memrealtime();
    for (int j = 0; j < nc; ++j) {
        double* vb = (j & 1) ? vb1 : vb0;
        double* sc = qsc + 2 * (j & 1);
        if (wave == ((j & 31) >> 1)) {
            const int hj = j & 1, lj = j >> 5, tj = j >> 5;          // row j sits in lane j & 31 of a half, slot j >> 5
            double part = 0, cand = 0;
#pragma unroll
            for (int l = 0; l < 4; ++l) if (l == lj) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int row = l32 + 32 * t;
                    if (row >= j) part += x[l][t] * x[l][t];
                    if (t == tj) cand = x[l][t];
                }
            }
            const double nrm2 = lane_f64(half32_sum_f64(part), 32 * hj);
            const double ajj = lane_f64(cand, (j & 31) + 32 * hj);
            double alpha, v0o, tauo;
            reflector_scalars(nrm2, ajj, alpha, v0o, tauo);
            if (lane == 0) { sc[0] = v0o; sc[1] = tauo; }
            if (hf == hj) {
#pragma unroll
                for (int l = 0; l < 4; ++l) if (l == lj) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int row = l32 + 32 * t;
                        if (row > j && row < m) vb[row] = x[l][t];
                        if (row == j) x[l][t] = alpha; else if (row > j) x[l][t] = 0.0;
                    }
                }
            }
        }
        __syncthreads();
        const double v0 = sc[0], tau = sc[1];
        double v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int row = l32 + 32 * t; v[t] = row == j ? v0 : ((row > j && row < m) ? vb[row] : 0.0); }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int c = 2 * wave + hf + 32 * l;
            if (2 * wave + 1 + 32 * l > j && 2 * wave + 32 * l <= nc) {          // some column of this slot is live (wavefront-uniform)
                double dot = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) dot += v[t] * x[l][t];
                dot = half32_sum_f64(dot) * tau;
                if (c > j && c <= nc) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) x[l][t] -= dot * v[t];
                }
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int c = 2 * wave + hf + 32 * l;
        if (c <= nc) {
            double* dst = c < nc ? Wt + (size_t)c * ldt : rcol;
#pragma unroll
            for (int t = 0; t < 8; ++t) { const int row = l32 + 32 * t; if (row < nc) dst[row] = x[l][t]; }
        }
    }
    __syncthreads();
}

// The stacked update runs as two kernels per launch group: update_front (row map, gather, thin QR -> rows 0..k-1 of
// [H_thin | r_thin] in W) and update_back (gain, covariance).  Separate kernels keep the register-resident QR and the
// 4x4-tiled products out of each other's register allocation (fused, the compiler spilled in every phase).
__device__ __forceinline__ void update_front(const UpdArgs& a)
{
    extern __shared__ double Lp[];               // front: row map / scan scratch, then the two reflector buffers (2 m doubles)
    __shared__ double red[UT / 64];
    __shared__ double vnorm[3];
    const int tid = threadIdx.x, m = a.m;
    // Column compression: the stacked Jacobian is zero outside the 6-wide blocks of the camera states that the
    // stacked features were observed from (the 21 IMU columns are always zero, msckf.py:535).  Only those nc
    // columns (a.cols, ascending) are factorised.  Replacing (H, r) by (R, Q^T r) of ANY thin QR of H leaves
    // delta_x and P+ unchanged (isotropic noise), so compressing to k = min(m, nc) rows is exact; the reference
    // compresses only when m > n (msckf.py:554) which is the same map in exact arithmetic.
    auto stamp = [&](int i) { if (a.prof && tid == 0) a.prof[i] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    const int nc = a.nc;
    const size_t ldt = a.ldt;                    // W is stored TRANSPOSED: Wt[q * ldt + i] = H[i][cols[q]]; r lives in row nc.
    double* Wt = a.W;                            // Column operations of the QR are then contiguous (coalesced) row segments.
    double* rcol = Wt + (size_t)nc * ldt;
    // 1. row map of the stacked matrix: srcrow[i] = row of Hsrc / rsrc that becomes stacked row i (inclusive scan of the
    //    block lengths).  The map and the scan scratch live where the QR keeps its reflector buffers later.
    __shared__ double qsc[4];
    int* srcrow = reinterpret_cast<int*>(Lp);
    {
        int* scan = srcrow + m;                          // n_blk <= m (every block has at least one row); host checks n_blk <= 4 * UT
        const int nb = a.n_blk;
        for (int b = tid; b < nb; b += UT) scan[b] = a.blk_len[b];
        __syncthreads();
        for (int off = 1; off < nb; off <<= 1) {
            int tmp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int b = tid + u * UT; tmp[u] = (b < nb && b >= off) ? scan[b - off] : 0; }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int b = tid + u * UT; if (b < nb) scan[b] += tmp[u]; }
            __syncthreads();
        }
        for (int b = tid; b < nb; b += UT) {
            const int len = a.blk_len[b], r0 = a.blk_row[b], d0 = scan[b] - len;
            for (int r = 0; r < len; ++r) srcrow[d0 + r] = r0 + r;
        }
        __syncthreads();
    }
    const bool comp = upd_compress(m, nc);
    const bool qr_regs = comp && ((nc + 1 <= 128 && m <= 256) || (nc + 1 <= 64 && m <= 512) || (nc + 1 <= 32 && m <= 1024) || (nc + 1 <= 16 && m <= 2048));
    if (!qr_regs) {
        // gather the gated feature blocks into Wt (coalesced writes along a column)
        for (int i = tid; i < m * nc; i += UT) {
            const int q = i / m, row = i - q * m;
            Wt[(size_t)q * ldt + row] = a.Hsrc[(size_t)srcrow[row] * a.ld + a.cols[q]];
        }
        for (int row = tid; row < m; row += UT) rcol[row] = a.rsrc[srcrow[row]];
    }
    __syncthreads();
    stamp(1);
    // 2. thin QR by Householder when m > nc; afterwards rows 0..k-1 hold [H_thin | r_thin]
    if (qr_regs) {
        double* vb0 = Lp;
        if (nc + 1 <= 128 && m <= 256) qr_half32(a, m, nc, srcrow, vb0, vb0 + m, qsc, Wt, ldt, rcol);
        else if (nc + 1 <= 64 && m <= 512) qr_in_registers<4, 8>(a, m, nc, srcrow, vb0, vb0 + m, qsc, Wt, ldt, rcol);
        else if (nc + 1 <= 32 && m <= 1024) qr_in_registers<2, 16>(a, m, nc, srcrow, vb0, vb0 + m, qsc, Wt, ldt, rcol);
        else if (nc + 1 <= 16 && m <= 1536) qr_in_registers<1, 24>(a, m, nc, srcrow, vb0, vb0 + m, qsc, Wt, ldt, rcol);
        else qr_in_registers<1, 32>(a, m, nc, srcrow, vb0, vb0 + m, qsc, Wt, ldt, rcol);
    } else if (comp) {
        const int wave = tid >> 6, lane = tid & 63, nw = UT / 64;
        // Two LDS buffers hold the current and the next reflector column.  The wavefront that updates column j+1
        // in step j also accumulates its squared norm and stages it as the next reflector, so a Householder step
        // costs one pass over the trailing columns and two barriers (no separate norm pass).
        double* vbuf[2] = {Lp, Lp + m};
        {
            const double* c0p = Wt;
            double part = 0;
            for (int i = tid; i < m; i += UT) { double v = c0p[i]; vbuf[0][i] = v; part += v * v; }
            const double n0 = block_sum(part, red);
            if (tid == 0) vnorm[2] = n0;
            __syncthreads();
        }
        for (int j = 0; j < nc; ++j) {
            double* vsh = vbuf[j & 1];
            double* vnext = vbuf[(j + 1) & 1];
            double* vj = Wt + (size_t)j * ldt;                   // column j in global memory
            if (tid == 0) {
                const double nrm2 = vnorm[2];
                const double ajj = vsh[j];
                const double nrm = sqrt(nrm2);
                const double alpha = ajj >= 0 ? -nrm : nrm;
                vnorm[0] = ajj - alpha;                             // v0
                const double vtv = nrm2 - 2 * alpha * ajj + alpha * alpha;
                vnorm[1] = vtv > 0 ? 2.0 / vtv : 0.0;               // tau
                vj[j] = alpha;                                      // R_jj
            }
            __syncthreads();
            const double v0 = vnorm[0], tau = vnorm[1];
            const int ncols = (nc - 1 - j) + 1;                     // trailing columns j+1..nc-1 plus the r column
            for (int c0 = wave * 4; c0 < ncols; c0 += nw * 4) {
                double* col[4]; double dot[4] = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cj = min(c0 + u, ncols - 1);
                    col[u] = cj < ncols - 1 ? Wt + (size_t)(j + 1 + cj) * ldt : rcol;
                }
                for (int i = j + 1 + lane; i < m; i += 64) {
                    const double v = vsh[i];
#pragma unroll
                    for (int u = 0; u < 4; ++u) dot[u] += v * col[u][i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) dot[u] = (wave_sum_f64(dot[u]) + v0 * col[u][j]) * tau;
                const int nvalid = min(4, ncols - c0);            // duplicates of the last column are not written twice
                const bool stage_next = (c0 == 0) && (j + 1 < nc);   // col[0] is column j+1: the next reflector
                double nn = 0;
                for (int i = j + 1 + lane; i < m; i += 64) {
                    const double v = vsh[i];
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (u < nvalid) {
                        const double nv = col[u][i] - dot[u] * v;
                        col[u][i] = nv;
                        if (u == 0 && stage_next) { vnext[i] = nv; nn += nv * nv; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (lane == u && u < nvalid) col[u][j] -= dot[u] * v0;
                if (stage_next) { nn = wave_sum_f64(nn); if (lane == 0) vnorm[2] = nn; }
            }
            __syncthreads();
        }
        // zero the reflector storage inside the leading nc x nc block (strictly below the diagonal)
        for (int i = tid; i < nc * nc; i += UT) { int c = i / nc, r = i - c * nc; if (c < r) Wt[(size_t)c * ldt + r] = 0.0; }
        __syncthreads();
    }
    stamp(2);
}


// C[r][c] = sum_q A(q, r) * B(q, c)  for r < R, c < Cn, q < Q, by one 1024-thread workgroup: 4x4 register tiles (at most
// two per thread), operands staged through LDS in panels of 16 values of q (wavefront w stages panel row w, lanes along
// the row: coalesced).  The unstaged form re-reads every operand value from L2 once per tile that needs it and is bound
// by the one CU's vector-memory path (64 B/clk); staged, global traffic drops from O(tiles * Q) to O((R + Cn) * Q).
constexpr int PANEL_Q = 16, PANEL_W = 208;      // PANEL_W >= 4 * ceil(max(R, Cn) / 4): n <= 21 + 6 * 31 = 207 (and <= 256 = 4 values per lane)
template <typename FA, typename FB, typename FOUT>
__device__ __forceinline__ void panel_gemm(int R, int Cn, int Q, double* pa, double* pb, bool lower_only, FA loadA, FB loadB, FOUT out)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tr = (R + 3) >> 2, tc = (Cn + 3) >> 2;
    const int RA = tr * 4, CB = tc * 4;
    // one 4x4 tile per thread and pass (n <= 141: 900 tiles of T, 666 lower tiles of the covariance -> a single pass); a
    // small register footprint matters more here than a second tile per thread: the kernel should leave room on its CU
    // for a wavefront of the front-end's kernels
    int ntiles = tr * tc;
    for (int tbase = 0; tbase < ntiles; tbase += UT) {
        const int t = tbase + tid;
        bool act = t < ntiles;
        int r0 = 0, c0 = 0;
        if (act) {
            if (lower_only) {
                // enumerate only the tiles on and below the diagonal: t -> (row, col) of the triangular index
                r0 = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
                while ((r0 + 1) * (r0 + 2) / 2 <= t) ++r0;
                while (r0 * (r0 + 1) / 2 > t) --r0;
                c0 = t - r0 * (r0 + 1) / 2;
                act = r0 < tr;
                r0 *= 4; c0 *= 4;
            } else {
                r0 = (t / tc) * 4; c0 = (t % tc) * 4;
            }
        }
        double acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
        for (int q0 = 0; q0 < Q; q0 += PANEL_Q) {
            const int qn = min(PANEL_Q, Q - q0);
            __syncthreads();
            if (wave < qn) {
                for (int e = lane; e < RA; e += 64) pa[wave * PANEL_W + e] = e < R ? loadA(q0 + wave, e) : 0.0;
                for (int e = lane; e < CB; e += 64) pb[wave * PANEL_W + e] = e < Cn ? loadB(q0 + wave, e) : 0.0;
            }
            __syncthreads();
            if (act) {
                const double* ap = pa + r0;
                const double* bp = pb + c0;
#pragma unroll 4
                for (int q = 0; q < qn; ++q) {
                    double av[4], bv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { av[i] = ap[q * PANEL_W + i]; bv[i] = bp[q * PANEL_W + i]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
                }
            }
        }
        if (act) out(r0, c0, acc);                  // the tile's owner writes it (rows/columns beyond R / Cn are the functor's to skip)
        __syncthreads();
    }
}
typedef double av_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void update_back(const UpdArgs& a)
{
    extern __shared__ double Lp[];               // packed lower triangle of S / its Cholesky factor: k(k+1)/2
    const int tid = threadIdx.x, n = a.n, m = a.m, nc = a.nc;
    auto stamp = [&](int i) { if (a.prof && tid == 0) a.prof[i] = __builtin_amdgcn_s_memrealtime(); };
    stamp(7);
    const size_t ldt = a.ldt;
    double* Wt = a.W;
    double* rcol = Wt + (size_t)nc * ldt;
    const int k = upd_k(m, nc);
    // 3. T = H_thin P (k x n): T[r][c] = sum_q Wt[q][r] P[cols[q]][c];  4. S = T H_thin^T + s^2 I (packed lower triangle in
    //    LDS): S[r][c] = sum_q T[r][cols[q]] Wt[q][c].  Both through panel_gemm (LDS-staged operands, 4x4 register tiles).
    double* pa = Lp + (size_t)k * (k + 1) / 2 + 8;
    double* pb = pa + PANEL_Q * PANEL_W;
    panel_gemm(k, n, nc, pa, pb, false,
               [&](int q, int r) { return Wt[(size_t)q * ldt + r]; },
               [&](int q, int c) { return a.P[(size_t)a.cols[q] * a.ld + c]; },
               [&](int r0, int c0, const double (&t)[4][4]) {
                   // rows of 4 doubles (32 B aligned: ld is a multiple of 8, c0 of 4); columns n..ld-1 of T are padding
#pragma unroll
                   for (int i = 0; i < 4; ++i) if (r0 + i < k) { av_d4 o = {t[i][0], t[i][1], t[i][2], t[i][3]}; *reinterpret_cast<av_d4*>(a.T + (size_t)(r0 + i) * a.ld + c0) = o; }
               });
    stamp(8);
    panel_gemm(k, k, nc, pa, pb, true,
               [&](int q, int r) { return a.T[(size_t)r * a.ld + a.cols[q]]; },
               [&](int q, int c) { return Wt[(size_t)q * ldt + c]; },
               [&](int r0, int c0, const double (&t)[4][4]) {
#pragma unroll
                   for (int i = 0; i < 4; ++i)
#pragma unroll
                       for (int j = 0; j < 4; ++j) {
                           const int r = r0 + i, c = c0 + j;
                           if (r < k && c <= r) Lp[(size_t)r * (r + 1) / 2 + c] = t[i][j] + (r == c ? a.obs_noise : 0.0);
                       }
               });
    stamp(3);
    // 5. Cholesky in LDS, right-looking with the column scaling deferred: step j only subtracts a_rj a_cj / d_j from the
    //    trailing block (one barrier per column instead of three); columns are divided by sqrt(d_j) in one pass at the end.
    for (int j = 0; j < k; ++j) {
        const double inv = 1.0 / Lp[j * (j + 1) / 2 + j];
        // 32 x 32 thread grid over the trailing block (no integer division by the shrinking block size)
        for (int r = j + 1 + (tid >> 5); r < k; r += UT / 32) {
            const double lrj = Lp[r * (r + 1) / 2 + j] * inv;
            for (int c = j + 1 + (tid & 31); c <= r; c += 32) Lp[r * (r + 1) / 2 + c] -= lrj * Lp[c * (c + 1) / 2 + j];
        }
        __syncthreads();
    }
    for (int e = tid; e < k * k; e += UT) {
        const int r = e / k, c = e - r * k;
        if (c < r) Lp[r * (r + 1) / 2 + c] /= sqrt(Lp[c * (c + 1) / 2 + c]);
    }
    __syncthreads();
    for (int j = tid; j < k; j += UT) Lp[j * (j + 1) / 2 + j] = sqrt(Lp[j * (j + 1) / 2 + j]);
    __syncthreads();
    stamp(4);
    // 6. Y = L^-1 [T | r_thin]: forward substitution only, one thread per right-hand side (n columns of T plus r).
    //    With S = L L^T:  K r = T^T S^-1 r = Y^T y_r  and  (I - K H) P = P - T^T S^-1 T = P - Y^T Y   (msckf.py:565-602);
    //    the backward substitution of an explicit K^T = S^-1 T is not needed.  Y overwrites Kt, y_r overwrites rcol.
    //    Blocked by 8 rows: the 8 partial sums share every load of an already solved y (one global load feeds 8 FMAs;
    //    the L entries are LDS broadcasts, all right-hand sides read the same address).
    for (int c = tid; c <= n; c += UT) {
        double* y = c < n ? a.Kt + c : rcol;
        const double* src = c < n ? a.T + c : rcol;
        const size_t st_ = c < n ? (size_t)a.ld : 1;
        for (int i0 = 0; i0 < k; i0 += 8) {
            double acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = i0 + u < k ? src[(size_t)(i0 + u) * st_] : 0.0;
            const double* Lr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = min(i0 + u, k - 1); Lr[u] = Lp + (size_t)i * (i + 1) / 2; }
#pragma unroll 4
            for (int q = 0; q < i0; ++q) {
                const double yq = y[(size_t)q * st_];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] -= Lr[u][q] * yq;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u < k) {
#pragma unroll
                    for (int w = 0; w < u; ++w) acc[u] -= Lr[u][i0 + w] * acc[w];
                    acc[u] /= Lr[u][i0 + u];
                    y[(size_t)(i0 + u) * st_] = acc[u];
                }
            }
        }
    }
    __syncthreads();
    stamp(5);
    // 7. delta_x = Y^T y_r
    for (int c = tid; c < n; c += UT) {
        double s = 0;
#pragma unroll 8
        for (int i = 0; i < k; ++i) s += a.Kt[(size_t)i * a.ld + c] * rcol[i];
        a.dx[c] = s;
    }
    stamp(9);
    // 8. P <- sym(P - Y^T Y) (msckf.py:597-602).  The (r,c) and (c,r) elements of Y^T Y are the same products summed in
    //    the same order, so both halves of the symmetrisation are available to the tile that owns (r,c): it reads the
    //    4x4 tile of P and its mirror tile (both row segments of 32 B), and the transposed, uncoalesced pass over the
    //    whole matrix that the separate (P + P^T)/2 needed is gone.  Pn then holds the result; a coalesced copy moves it back.
    //    Only the tiles on and below the diagonal are formed (half the FMAs and LDS traffic); each writes its mirror too.
    panel_gemm(n, n, k, pa, pb, true,
               [&](int q, int r) { return a.Kt[(size_t)q * a.ld + r]; },
               [&](int q, int c) { return a.Kt[(size_t)q * a.ld + c]; },
               [&](int r0, int c0, const double (&t)[4][4]) {
                   if (r0 + 3 < n && c0 + 3 < n) {
                       av_d4 prow[4], mrow[4];
#pragma unroll
                       for (int i = 0; i < 4; ++i) {
                           prow[i] = *reinterpret_cast<const av_d4*>(a.P + (size_t)(r0 + i) * a.ld + c0);
                           mrow[i] = *reinterpret_cast<const av_d4*>(a.P + (size_t)(c0 + i) * a.ld + r0);
                       }
                       av_d4 o[4], om[4];
#pragma unroll
                       for (int i = 0; i < 4; ++i)
#pragma unroll
                           for (int j = 0; j < 4; ++j) { const double v = ((prow[i][j] - t[i][j]) + (mrow[j][i] - t[i][j])) / 2.; o[i][j] = v; om[j][i] = v; }
#pragma unroll
                       for (int i = 0; i < 4; ++i) *reinterpret_cast<av_d4*>(a.Pn + (size_t)(r0 + i) * a.ld + c0) = o[i];
                       if (r0 != c0) {
#pragma unroll
                           for (int i = 0; i < 4; ++i) *reinterpret_cast<av_d4*>(a.Pn + (size_t)(c0 + i) * a.ld + r0) = om[i];
                       }
                   } else {
#pragma unroll
                       for (int i = 0; i < 4; ++i)
#pragma unroll
                           for (int j = 0; j < 4; ++j) {
                               const int r = r0 + i, c = c0 + j;
                               if (r < n && c < n) {
                                   const double v = ((a.P[(size_t)r * a.ld + c] - t[i][j]) + (a.P[(size_t)c * a.ld + r] - t[i][j])) / 2.;
                                   a.Pn[(size_t)r * a.ld + c] = v;
                                   a.Pn[(size_t)c * a.ld + r] = v;
                               }
                           }
                   }
               });
    stamp(10);
    for (int i0 = tid; i0 < n * a.ld; i0 += 8 * UT) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * UT; x[u] = i < n * a.ld ? a.Pn[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * UT; if (i < n * a.ld) a.P[i] = x[u]; }
    }
    __syncthreads();
    stamp(6);
}
// ================================================================================================
// Back end of the stacked update for MANY streams (av_msckf_batch_*): msckf_mfma.inc (included below, after the kernels whose helpers it
// shares) -- T^T = P[:, cols] H_thin^T, S = H_thin P_cc H_thin^T + s^2 I, L L^T = S, Y = L^-1 [T | r], P <- sym(P - Y^T Y) as
// single-wavefront tasks on the fp64 matrix instruction.  The single-workgroup update_back above does the same arithmetic inside one
// 1024-thread workgroup at 128 VGPRs (the single-filter path): it owns a whole CU for ~0.4 ms per stream while using a few percent of it.
// (Rounds 3-5 ran these products as 64 x 64 LDS tile kernels -- 256-thread workgroups, 4 x 4 fp64 register tiles, panels of 16 -- around
//  a 256-thread factor kernel with the packed triangle in 74 KB of LDS and a lane-per-right-hand-side substitution: git 26affda.  A 48-wide
//  wave-tile variant of them, one 576-thread workgroup per stream, was slower still: 148.6 k against 155.8 k frames/s.  profiles/r05/README.md
//  has the numbers that retired them: 315 / 247 / 490 / 426 / 375 us per 2,048-stream launch alone and 2-5x that beside the front-end.)
// ================================================================================================
// Pointers that arrive inside a struct read from memory (UpdArgs) are generic to the compiler: it emits FLAT loads, which count on
// lgkmcnt as well as vmcnt -- an LDS-only barrier would wait for them.  These casts say "global memory" (global_load: vmcnt only).
typedef const double __attribute__((address_space(1)))* gcd_ptr;
typedef const int __attribute__((address_space(1)))* gci_ptr;
#define AV_GD(p) ((gcd_ptr)(p))
#define AV_GI(p) ((gci_ptr)(p))

// Workgroup barrier for LDS hand-overs only: the LDS writes of this wavefront are done (lgkmcnt(0)) and every wavefront has arrived.
// __syncthreads() also drains vmcnt -- every global load or store in flight has to come back first.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// ================================================================================================
// Information form of the stacked update for streams whose stacked Jacobian touches FEW columns (nc <= INFO_NC: the
// two-camera pruning update, msckf.py:712-786, stacks ~5 rows per live feature -- 1,500 rows at 300 features -- over
// exactly 12 columns).  With isotropic noise s^2 I and H = [0 Hc] (columns `cols`):
//      K r        = P[:,c] (A Pcc + s^2 I)^-1 b,          A = Hc^T Hc  (nc x nc),   b = Hc^T r
//      (I - K H)P = P - P[:,c] (A Pcc + s^2 I)^-1 A P[c,:]
// (push-through identity Hc^T (Hc Pcc Hc^T + s^2 I)^-1 = (A Pcc + s^2 I)^-1 Hc^T), which is what msckf.py:548-602 computes
// by thin QR + solve, without ever forming a triangular factor of the 1,500-row matrix: A and b are sums over the rows,
// the only factorisation is a pivoted elimination of an nc x nc matrix.  A is singular (the projected Jacobian is blind to
// a rigid motion of the camera pair), which is why the Gram matrix is never inverted -- only M = A Pcc + s^2 I is, and its
// eigenvalues are s^2 + eig(Pcc^1/2 A Pcc^1/2) >= s^2.  One 256-thread workgroup per stream, ~36 KB of LDS, instead of a
// 1024-thread QR workgroup that owns a CU plus five back-end launches.
// ================================================================================================
constexpr int INFO_NC = 24, INFO_CH = 96, INFO_MAXROWS = 8192;
__host__ __device__ inline bool upd_info_form(int m, int nc) { return nc <= INFO_NC && m > nc && m <= INFO_MAXROWS; }
static inline size_t upd_info_lds(int nc, int n, int m)
{
    return sizeof(double) * ((size_t)nc * nc + nc + (size_t)nc * (n + 1) + (size_t)nc * (nc + n + 2) + (size_t)INFO_CH * (INFO_NC + 1)) + sizeof(int) * ((size_t)m + 2) + 64;
}

template <typename UA> __device__ __forceinline__ void upd_info_body(const UA& a, double* Li);
__global__ __launch_bounds__(256) void upd_info_kernel(const UpdArgs* __restrict__ arr)
{
    AV_FILTER_PRIO();
    extern __shared__ double Li_dyn[];
    const UpdArgsG a = upd_load(arr, blockIdx.x);
    if (a.m <= 0 || a.mode != 1) return;
    upd_info_body(a, Li_dyn);
}
template <typename UA> __device__ __forceinline__ void upd_info_body(const UA& a, double* Li)
{
    const int tid = threadIdx.x, n = a.n, nc = a.nc, nb = a.n_blk;
    double* A = Li;                          // [nc][nc]
    double* bv = A + nc * nc;                // [nc]
    double* Pc = bv + nc;                    // [nc][n+1]   rows cols[q] of P (P is symmetric: also its columns), pitch n+1
    double* G = Pc + (size_t)nc * (n + 1);   // [nc][W]     augmented [M | F | b], W = nc + n + 1 (+1 pad)
    const int PW_ = n + 1, W = nc + n + 2;
    __shared__ int piv_row;
    auto stamp = [&](int i) { if (a.prof && tid == 0) a.prof[i] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    // 1. A = Hc^T Hc (upper triangle, then mirrored), b = Hc^T r.  The stacked rows are staged through LDS in chunks of INFO_CH
    //    rows x (nc + 1) values by all 256 threads (the next chunk's loads are in flight while this one is summed); a thread
    //    per entry of A / b then accumulates over the chunk.  Row map: stacked row i -> row of the feature-block buffer.
    int* srow = reinterpret_cast<int*>(G + (size_t)nc * W);          // [m]
    double* chunk = reinterpret_cast<double*>(srow + ((a.m + 1) & ~1));   // [INFO_CH][nc + 1]
    {
        int* pre = reinterpret_cast<int*>(chunk);                      // block prefix (inclusive), lives in the chunk buffer until the map is built
        // inclusive scan of the block lengths: each thread sums a contiguous run, thread 0 scans the 256 run totals
        const int per = (nb + 255) / 256;
        int run = 0;
        for (int u = 0; u < per; ++u) { const int b = tid * per + u; if (b < nb) { run += a.blk_len[b]; pre[b] = run; } }
        __shared__ int tot[256];
        tot[tid] = run;
        __syncthreads();
        if (tid == 0) { int acc = 0; for (int t = 0; t < 256; ++t) { const int v = tot[t]; tot[t] = acc; acc += v; } }
        __syncthreads();
        for (int u = 0; u < per; ++u) { const int b = tid * per + u; if (b < nb) pre[b] += tot[tid]; }
        __syncthreads();
        for (int b = tid; b < nb; b += 256) {
            const int len = a.blk_len[b], r0 = a.blk_row[b], d0 = pre[b] - len;
            for (int r = 0; r < len; ++r) srow[d0 + r] = r0 + r;
        }
        __syncthreads();
    }
    stamp(1);
    {
        const int m = a.m, cw = nc + 1;
        // entries: the nc (nc + 1) / 2 pairs (r, c <= r) of A, then the nc entries of b -- at most 324 for nc = 24: two per thread
        const int ntri = nc * (nc + 1) / 2, nent = ntri + nc;
        int ei[2], ej[2]; bool act[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int e = tid + 256 * v;
            act[v] = e < nent;
            if (e < ntri) {
                int r = (int)((sqrtf(8.f * e + 1.f) - 1.f) * 0.5f);
                while ((r + 1) * (r + 2) / 2 <= e) ++r;
                while (r * (r + 1) / 2 > e) --r;
                ei[v] = r; ej[v] = e - r * (r + 1) / 2;
            } else { ei[v] = min(e - ntri, nc - 1); ej[v] = nc; }           // b: column nc of the staged rows is the residual
        }
        double acc[2] = {0, 0};
        constexpr int PER = (INFO_CH * (INFO_NC + 1) + 255) / 256;
        double stage[PER];
        auto fetch = [&](int base) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int idx = tid + 256 * u, rr = idx / cw, cc = idx - rr * cw;
                stage[u] = 0.0;
                if (rr < INFO_CH && base + rr < m) { const int sr = srow[base + rr]; stage[u] = cc < nc ? (a.hld > 0 ? a.Hsrc[(size_t)sr * a.hld + cc] : a.Hsrc[(size_t)sr * a.ld + a.cols[cc]]) : a.rsrc[sr]; }
            }
        };
        fetch(0);
        for (int base = 0; base < m; base += INFO_CH) {
#pragma unroll
            for (int u = 0; u < PER; ++u) { const int idx = tid + 256 * u; if (idx < INFO_CH * cw) chunk[idx] = stage[u]; }
            lds_barrier();                                            // (LDS-only barriers: __syncthreads() would wait for the prefetch below)
            if (base + INFO_CH < m) fetch(base + INFO_CH);            // next chunk's loads overlap this chunk's sums
            const int rows = min(INFO_CH, m - base);
#pragma unroll
            for (int v = 0; v < 2; ++v)
                if (act[v]) for (int rr = 0; rr < rows; ++rr) acc[v] = __builtin_fma(chunk[rr * cw + ei[v]], chunk[rr * cw + ej[v]], acc[v]);
            lds_barrier();
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
            if (act[v]) { if (ej[v] == nc) bv[ei[v]] = acc[v]; else { A[ei[v] * nc + ej[v]] = acc[v]; A[ej[v] * nc + ei[v]] = acc[v]; } }
    }
    stamp(2);
    for (int e = tid; e < nc * n; e += 256) { const int q = e / n, c = e - q * n; Pc[q * PW_ + c] = a.P[(size_t)a.cols[q] * a.ld + c]; }
    __syncthreads();
    stamp(3);
    // 2. F = A Pc (nc x n);  G = [F[:, cols] + s^2 I | F | b]
    for (int e = tid; e < nc * n; e += 256) {
        const int i = e / n, c = e - i * n;
        double acc = 0;
        for (int q = 0; q < nc; ++q) acc = __builtin_fma(A[i * nc + q], Pc[q * PW_ + c], acc);
        G[i * W + nc + c] = acc;
    }
    for (int i = tid; i < nc; i += 256) G[i * W + nc + n] = bv[i];
    __syncthreads();
    for (int e = tid; e < nc * nc; e += 256) { const int i = e / nc, j = e - i * nc; G[i * W + j] = G[i * W + nc + a.cols[j]] + (i == j ? a.obs_noise : 0.0); }
    __syncthreads();
    stamp(4);
    // 3. Gauss-Jordan with partial pivoting on the augmented rows: afterwards G[:, nc:] = M^-1 [F | b] (scaled by the pivots)
    for (int p = 0; p < nc; ++p) {
        if (tid == 0) {
            int best = p; double bm = fabs(G[p * W + p]);
            for (int r = p + 1; r < nc; ++r) { const double v = fabs(G[r * W + p]); if (v > bm) { bm = v; best = r; } }
            piv_row = best;
        }
        __syncthreads();
        const int pr = piv_row;
        if (pr != p) for (int c = tid; c < W - 1; c += 256) { const double t = G[p * W + c]; G[p * W + c] = G[pr * W + c]; G[pr * W + c] = t; }
        __syncthreads();
        const double inv = 1.0 / G[p * W + p];
        // eliminate column p from every other row (columns > p only: the rest of column p is never read again)
        for (int e = tid; e < nc * (W - 1 - (p + 1)); e += 256) {
            const int r = e / (W - 1 - (p + 1)), c = p + 1 + (e - r * (W - 1 - (p + 1)));
            if (r != p) G[r * W + c] = __builtin_fma(-G[r * W + p] * inv, G[p * W + c], G[r * W + c]);
        }
        __syncthreads();
    }
    stamp(5);
    for (int e = tid; e < nc * (n + 1); e += 256) { const int i = e / (n + 1), c = e - i * (n + 1); G[i * W + nc + c] /= G[i * W + i]; }
    __syncthreads();
    stamp(6);
    // 4. dx = P[:,c] X_b;   P <- sym(P - P[:,c] X): a thread per pair (r, c <= r), in place
    const double* X = G + nc;                // X[q][c] = G[q*W + nc + c], c < n; X_b = column n
    for (int c = tid; c < n; c += 256) {
        double acc = 0;
        for (int q = 0; q < nc; ++q) acc = __builtin_fma(Pc[q * PW_ + c], X[q * W + n], acc);
        a.dx[c] = acc;
    }
    // 2 x 2 tiles of the lower triangle per thread: eight LDS reads per eight multiply-adds (a pair at a time made two reads per
    // multiply-add and a double-precision square root per pair for its index)
    const int nt = (n + 1) / 2;
    for (int e = tid; e < nt * (nt + 1) / 2; e += 256) {
        int R = (int)((sqrtf(8.f * e + 1.f) - 1.f) * 0.5f);
        while ((R + 1) * (R + 2) / 2 <= e) ++R;
        while (R * (R + 1) / 2 > e) --R;
        const int C = e - R * (R + 1) / 2, r0 = 2 * R, c0 = 2 * C;
        double trc[2][2] = {{0, 0}, {0, 0}}, tcr[2][2] = {{0, 0}, {0, 0}};
        for (int q = 0; q < nc; ++q) {
            const double pr[2] = {Pc[q * PW_ + r0], Pc[q * PW_ + r0 + 1]}, xc[2] = {X[q * W + c0], X[q * W + c0 + 1]};
            const double pc[2] = {Pc[q * PW_ + c0], Pc[q * PW_ + c0 + 1]}, xr[2] = {X[q * W + r0], X[q * W + r0 + 1]};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    trc[i][j] = __builtin_fma(pr[i], xc[j], trc[i][j]);
                    tcr[i][j] = __builtin_fma(pc[j], xr[i], tcr[i][j]);
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = r0 + i, c = c0 + j;
                if (r < n && c <= r) {                       // (c <= r < n; the upper entries of a diagonal tile belong to their mirror)
                    const double v = ((a.P[(size_t)r * a.ld + c] - trc[i][j]) + (a.P[(size_t)c * a.ld + r] - tcr[i][j])) / 2.;
                    a.P[(size_t)r * a.ld + c] = v;
                    a.P[(size_t)c * a.ld + r] = v;
                }
            }
    }
    stamp(7);
}

__global__ __launch_bounds__(UT) void update_front_kernel(UpdArgs a) { update_front(a); }
__global__ __launch_bounds__(UT) void update_back_kernel(UpdArgs a) { update_back(a); }
// `list` (optional): the streams to run, so that the 1024-thread workgroups exist only for the streams that really compress
__global__ __launch_bounds__(UT) void update_front_batch_kernel(const UpdArgs* __restrict__ arr, const int* __restrict__ list)
{
    const UpdArgs a = arr[list ? list[blockIdx.x] : (int)blockIdx.x];  // block-uniform: lives in scalar registers
    // (a listed stream may turn out not to compress: the chained path lists every stream that COULD exceed a chunk)
    if (a.m > 0 && a.mode == 0 && a.kdir <= 0 && upd_compress(a.m, a.nc)) update_front(a);
}
// Streams whose stacked Jacobian is NOT compressed (upd_compress: at most 144 rows) only need the gated feature blocks
// gathered into the transposed work matrix: a 256-thread workgroup, no QR machinery.
__global__ __launch_bounds__(256) void upd_gather_kernel(const UpdArgs* __restrict__ arr)
{
    AV_FILTER_PRIO();
    const UpdArgsG a = upd_load(arr, blockIdx.x);
    if (a.m <= 0 || a.mode != 0 || (a.kdir <= 0 && upd_compress(a.m, a.nc))) return;
    __shared__ int srow[256], bstart[256];
    const int tid = threadIdx.x, m = a.m, nc = a.nc, nb = a.n_blk;        // m <= 144, every block has at least one row
    if (tid == 0) { int run = 0; for (int b = 0; b < nb; ++b) { bstart[b] = run; run += a.blk_len[b]; } }
    __syncthreads();
    if (tid < nb) { const int len = a.blk_len[tid], r0 = a.blk_row[tid], d0 = bstart[tid]; for (int r = 0; r < len; ++r) srow[d0 + r] = r0 + r; }
    __syncthreads();
    const size_t ldt = a.ldt;
    for (int i = tid; i < m * nc; i += 256) {
        const int q = i / m, row = i - q * m;
        a.W[(size_t)q * ldt + row] = a.Hsrc[(size_t)srow[row] * a.ld + a.cols[q]];
    }
    const auto rcol = a.W + (size_t)nc * ldt;
    // later chunks of a sequential update see the state correction of the earlier ones: r' = r - H dx (the batch update of
    // [H0; H1] equals the update with H0 followed by the update with H1 on (P1, r1 - H1 dx1); dx is injected once, at the end)
    for (int row = tid; row < m; row += 256) {
        double r = a.rsrc[srow[row]];
        if (a.round > 0) for (int q = 0; q < nc; ++q) r = __builtin_fma(-a.Hsrc[(size_t)srow[row] * a.ld + a.cols[q]], a.dx[a.cols[q]], r);
        rcol[row] = r;
    }
}
// ================================================================================================
// Row compression WITHOUT a QR for the streams that stack more rows than one back-end pass keeps (m > 144 over n_c <= 144 columns).
// With isotropic noise the update depends on the stacked Jacobian only through A = Hc^T Hc and b = Hc^T r (information form:
// P+ = (P^-1 + A / s^2)^-1, dx = P+ b / s^2), so ANY (F, f) with F^T F = A and F^T f = b may stand in for (H, r) -- the thin QR's
// (R, Q^T r) that msckf.py:554-557 uses is one such pair, the Cholesky factor of the Gram matrix another:
//      [A + E  b; b^T  rho] = L L^T  (bordered, E = 1e-12 diag(A)),   F = L_nc^T,   f = L_nc^-1 b = the border row of L.
// A is singular (the projected Jacobian is blind to a rigid motion of the observing cameras), hence E: a relative 1e-12 on the
// diagonal, six orders below the 1e-6 parity tolerance and four above the rounding of the sums.  The Gram matrix is a sum over the
// rows -- a wide-grid tile GEMM, split over chunks of 256 rows with the partial sums added in a fixed order (no atomics) -- and the
// only dependent chain is one n_c-sized Cholesky in a 256-thread workgroup.  This replaces update_front_batch_kernel (a
// 1024-thread, 128-VGPR Householder workgroup that had to wait ~1 ms for a whole free CU) on the batched path.
//   upd_rowmap_kernel     stacked row -> row of the block buffer (scan of the gated blocks), int[m] in the stream's Kt buffer
//   upd_gram_mfma_kernel       Gram matrix of [Hc | r], a 32 x 32 block of the lower triangle per wavefront, into the stream's Sbuf (msckf_mfma.inc)
//   upd_gram_chol_mfma_kernel  bordered Cholesky by one wavefront -> [F | f] written as the k = n_c rows of W
// ================================================================================================
constexpr int GRAM_CHUNK = 256;
template <typename UA> __device__ __forceinline__ bool upd_front_needed(const UA& a) { return a.m > 0 && a.mode == 0 && a.kdir <= 0 && upd_compress(a.m, a.nc); }

// (n_list_dev != NULL: the list was written on the device -- upd_stack_kernel -- and holds *n_list_dev streams; the workgroups
//  stride over it.  NULL: one workgroup per listed stream, as launched by the host.)
template <typename UA> __device__ __forceinline__ void upd_rowmap_one(const UA& a);
// (one wavefront per workgroup: beside the front-end's single-wavefront workgroups a 256-thread workgroup waits for a free slot on all
//  four SIMDs of a CU at once -- this kernel, 7 us alone, took 1.1 ms in the shared run)
__global__ __launch_bounds__(64) void upd_rowmap_kernel(const UpdArgs* __restrict__ arr, const int* __restrict__ list, const int* __restrict__ n_list_dev)
{
    AV_FILTER_PRIO();
    if (n_list_dev) { const int n = *n_list_dev; for (int l = blockIdx.x; l < n; l += gridDim.x) upd_rowmap_one(upd_load(arr, list[l])); return; }
    upd_rowmap_one(upd_load(arr, list[blockIdx.x]));
}
template <typename UA> __device__ __forceinline__ void upd_rowmap_one(const UA& a)
{
    if (!upd_front_needed(a)) return;
    const auto srcrow = (av_gptr<int>)(a.Kt);
    const int lane = threadIdx.x, nb = a.n_blk;
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 64) {                // blocks in order: exclusive scan of their lengths, a wavefront's worth at a time
        const int b = b0 + lane;
        const int len = b < nb ? a.blk_len[b] : 0;
        int incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        const int d0 = carry + incl - len;
        if (b < nb) { const int r0 = a.blk_row[b]; for (int r = 0; r < len; ++r) srcrow[d0 + r] = r0 + r; }
        carry += __shfl(incl, 63, 64);
    }
}

// ================================================================================================
// Stacking decisions of remove_lost_features / prune_cam_state_buffer ON THE DEVICE (msckf.py:658-668, 759-763): which gated
// blocks are stacked (in feature = map order, up to the `> 1500 rows` cut), which camera columns they touch, and how the
// stacked rows are split into the sequential chunks of the batched back end.  One wavefront per stream.  The host only
// knows upper bounds (it launches `rounds` rounds of back-end kernels; rounds a stream does not need see m = 0 and exit),
// so the gate flags never travel to the host between the feature kernels and the update.
// ================================================================================================
constexpr int STACK_LDS_BLOCKS = 4096;          // = 4 * UT, the block-list limit of the back end
struct StackArgs {
    const int* rbeg;                 // [S + 1] feature range of every stream (features are stream-major, map order inside)
    const int* pass;                 // [n_feat] gate flags of feature_kernel
    const int* obs_off; const int* obs_cam;      // the feature kernels' observation CSR (camera slot per observation)
    const int* row_off;              // [n_feat] first row of every feature's block
    int* blk_row; int* blk_len;      // out: stacked blocks of stream s at [rbeg[s], rbeg[s] + n_blk)
    int* cols; int cols_stride;      // out: touched state columns of stream s at cols + s * cols_stride (ascending)
    const UpdArgs* base;             // [S] per-stream constants (pointers, n, ld, ...); base.mode = 1: information form allowed, 2: no update
    UpdArgs* out;                    // [rounds][S]
    int S, rounds, cut1500, kch;
    int compress;                    // 1: a stream with more than kch rows is QR-compressed (update_front_batch_kernel) and updated in ONE round
    int* stacked;                    // out [S]: rows stacked (0 = no update), -1 = more chunks than `rounds` (nothing is updated)
    // ---- device-resident filter (msckf_dev.inc); all NULL / 0 on the host-driven path ------------------------------------------
    const int* n_cand; int fs_stride;        // rbeg == NULL: the features of stream s are [s * fs_stride, s * fs_stride + n_cand[s])
    const int* stream_n;                     // state dimension of every stream (overrides base[s].n: the host does not know it)
    const int* valid;                        // [n_feat] 0 = the feature was never evaluated (no position); work counters only
    const int* over;                         // [S] 1: the stream's candidates reserved more rows than the block buffer holds, so they were
    int* row_off_w; int* again_list; int* again_count;   //  gated without storage (row_off = -1): the stacked ones get compact offsets here and are listed to run again
    int* clist; int* clist_count;            // out: streams whose stacked rows exceed one back-end pass (compression kernels stride over this list)
    double* work;                            // [S][8] accumulators of av_msckf_batch_work: gate flops, update flops, reference-QR flops, gated, updates, rows; [6], [7]: executed gate / update flops (av_msckf_batch_work_executed)
    int hld;                                 // UpdArgs::hld of this phase's updates (0: dense rows)
    int no_info;                             // 1: this phase launches no information-form kernel (every stream takes the Cholesky back end)
};
__global__ __launch_bounds__(64) void upd_stack_kernel(StackArgs a)
{
    AV_FILTER_PRIO();
    __shared__ int s_len[STACK_LDS_BLOCKS];
    __shared__ int s_chunk[64];                          // first block of chunk c (c < rounds <= 63), then the end
    __shared__ int s_nch;
    const int s = blockIdx.x, lane = threadIdx.x;
    const int i0 = a.rbeg ? a.rbeg[s] : s * a.fs_stride, i1 = a.rbeg ? a.rbeg[s + 1] : i0 + a.n_cand[s];
    const bool two_pass = a.over && a.over[s];
    const int n_state = a.stream_n ? a.stream_n[s] : a.base[s].n;
    if (a.work) {        // per gated feature with r = 4M-3 rows against n columns  2 r n^2 + 2 r^2 n + r^3 / 3
        double fl = 0, fx = 0; int cnt = 0;
        for (int i = i0 + lane; i < i1; i += 64) {
            if (a.valid && !a.valid[i]) continue;
            const double r = 4.0 * (a.obs_off[i + 1] - a.obs_off[i]) - 3.0, n = (double)n_state;
            fl += 2 * r * n * n + 2 * r * r * n + r * r * r / 3.0; ++cnt;
            // what feature_kernel EXECUTES for a track of M observations (block-sparse: H_x is 4 x 6 per observation, the gate
            // matrix is built from the M^2 6 x 6 blocks of P before the projection, the reflectors are applied to it from both
            // sides -- DESIGN.md 3b): Jacobians ~600 M; three reflectors from H_f and the projected residual 96 M; reflectors
            // over the 6 M columns of H_x (four products per column and reflector, then 4 M rows x 3 updates) 6 M (34 + 24 M);
            // G = H_x P_sub H_x^T 480 M^2; reflectors on G from both sides 384 M^2; Cholesky r^3 / 3 and the solve r^2
            const double M = (double)(a.obs_off[i + 1] - a.obs_off[i]);
            fx += 696 * M + 6 * M * (34 + 24 * M) + 864 * M * M + r * r * r / 3.0 + r * r;
        }
        fl = wave_sum_f64(fl); fx = wave_sum_f64(fx);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if (lane == 0) { a.work[8 * s] += fl; a.work[8 * s + 3] += (double)cnt; a.work[8 * s + 6] += fx; }
    }
    int stacked = 0, nb = 0;
    unsigned long long used = 0ull;
    bool overflow = false;
    for (int base = i0; base < i1; base += 64) {
        const int i = base + lane;
        const bool in = i < i1;
        const int o0 = in ? a.obs_off[i] : 0, nobs = in ? a.obs_off[i + 1] - o0 : 0;
        const int rows = 4 * nobs - 3;
        const int v = (in && a.pass[i]) ? rows : 0;
        int incl = v;                                    // inclusive prefix sum over the wavefront
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        const int before = stacked + incl - v;           // rows stacked when the reference reaches this feature
        const bool take = v > 0 && (!a.cut1500 || before <= 1500);       // the loop breaks once stacked > 1500 (msckf.py:667-668)
        const unsigned long long mask = __ballot(take);
        const int pos = nb + __popcll(mask & ((1ull << lane) - 1ull));
        if (take) {
            if (pos < STACK_LDS_BLOCKS) s_len[pos] = rows; else overflow = true;
            int ro = a.row_off[i];
            if (two_pass) {      // every passing feature up to the cut is stacked, so the rows stacked before this one are its compact offset
                ro = before; a.row_off_w[i] = ro;
                a.again_list[atomicAdd(a.again_count, 1)] = i;
            }
            a.blk_row[i0 + pos] = ro; a.blk_len[i0 + pos] = rows;
            for (int q = 0; q < nobs; ++q) used |= 1ull << a.obs_cam[o0 + q];
        }
        nb += __popcll(mask);
        int add = take ? v : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) add += __shfl_xor(add, d, 64);
        stacked += add;
        if (a.cut1500 && stacked > 1500) break;          // uniform
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { used |= __shfl_xor(used, d, 64); }
    overflow = __ballot(overflow) != 0ull;
    UpdArgs b = a.base[s];
    b.n = n_state;
    const int nc = 6 * __popcll(used);
    const bool dead = b.mode == 2;                      // stopped by the host (msckf_batch.inc: b_fail_stream): no update, stacked = -1
    const int m = (overflow || dead) ? 0 : stacked;
    // (the information form holds for any m >= 1; a stream the host marked for it never needs a Cholesky round)
    const int mode = (m > 0 && b.mode == 1 && !a.no_info && nc <= INFO_NC && m <= INFO_MAXROWS) ? 1 : 0;
    __syncthreads();
    if (lane == 0) {
        int c = 0;
        for (int ci = 0; ci < 64; ++ci) if (used >> ci & 1ull) for (int e = 0; e < 6; ++e) a.cols[(size_t)s * a.cols_stride + c++] = IMU_DIM + 6 * ci + e;
        int nch = 0;
        if (m > 0 && mode == 0 && a.compress) nch = 1;   // one round: direct if m <= kch, else thin QR to nc rows first
        else if (m > 0 && mode == 0) {                   // next-fit chunks of at most kch rows (a block is never split)
            int acc = 0;
            s_chunk[nch++] = 0;
            for (int k = 0; k < nb; ++k) {
                const int len = s_len[k];
                if (acc + len > a.kch) { if (nch < 63) s_chunk[nch] = k; ++nch; acc = 0; }
                acc += len;
            }
            if (nch <= 62) s_chunk[nch] = nb;
        }
        s_nch = nch;
    }
    __syncthreads();
    const int nch = s_nch;
    const bool too_many = nch > a.rounds || nch > 62;
    if (lane == 0) {
        a.stacked[s] = overflow || too_many || dead ? -1 : m;
        const bool upd = m > 0 && !too_many;
        if (upd && a.clist && mode == 0 && a.compress && m > a.kch) a.clist[atomicAdd(a.clist_count, 1)] = s;
        if (upd && a.work) {     // per update of m stacked rows with k = min(m, n) rows kept, the reference's thin QR counted apart
            const double mm = m, n = (double)n_state, k = mm < n ? mm : n;
            a.work[8 * s + 1] += 2 * k * n * n + 2 * k * k * n + k * k * k / 3.0 + 2 * k * k * n + 4 * k * n * n;
            if (mm > n) a.work[8 * s + 2] += 2 * mm * n * n - 2.0 / 3.0 * n * n * n;
            a.work[8 * s + 4] += 1.0; a.work[8 * s + 5] += mm;
            // what the update kernels EXECUTE over the nc touched columns (analytic).  Information form (nc <= 24, vector FMAs): Gram
            // m nc^2, b 2 m nc, A P_cc and the 12 x 12 elimination 4 nc^3, the two products with P[:, c] 2 n nc^2 + 2 n^2 nc.  Cholesky
            // back end (msckf_mfma.inc): matrix instructions of 16 x 16 x 4, 2,048 flops each, PADDING INCLUDED -- a product over t output
            // tiles and K summed rows issues t ceil(K / 4) of them: the Gram compression of a stream with more rows than one pass
            // (lower tiles of nc + 1, K = m; then its factor and kk = nc rows), T^T (tiles of n x kk, K = nc), S (lower tiles of kk, K = nc),
            // the factor (left-looking: (nb - J) tiles with K = 16 J, plus 4 per sub-diagonal tile for the diagonal block's inverse; the
            // 16 x 16 diagonal blocks themselves ~2 x 16^3 / 3 vector flops each way), the substitution (per 16 right-hand sides and block
            // row I: 4 I + 4), P - Y^T Y (lower tiles of n, K = kk) and delta_x 2 kk n
            const double c = (double)nc;
            if (mode == 1) a.work[8 * s + 7] += mm * c * c + 2 * mm * c + 4 * c * c * c + 2 * n * c * c + 2 * n * n * c;
            else {
                const bool comp = a.compress && m > a.kch;
                const int kki = comp ? nc : m;
                auto t16 = [](int x) { return (double)((x + 15) / 16); };
                auto q4 = [](int x) { return (double)((x + 3) / 4); };
                auto lower = [](double t) { return t * (t + 1) / 2; };
                auto factor = [](double nb) { double f = 0; for (int J = 0; J < (int)nb; ++J) f += (nb - J) * 4.0 * J + 4.0 * (nb - J - 1); return f * 2048.0 + nb * 2.0 * 2731.0; };
                const double tn = t16(n_state), tk = t16(kki);
                double fl = 2048.0 * (tn * tk * q4(nc) + lower(tk) * q4(nc) + t16(n_state + 1) * 2.0 * tk * (tk + 1) + lower(tn) * q4(kki)) + factor(tk) + 2.0 * kki * n;
                if (comp) fl += 2048.0 * lower(t16(nc + 1)) * q4(m) + factor(t16(nc + 1));
                a.work[8 * s + 7] += fl;
            }
        }
    }
    for (int r = lane; r < a.rounds; r += 64) {
        UpdArgs u = b;
        u.mode = mode; u.round = r; u.kdir = 0; u.nc = nc; u.hld = a.hld; u.cols = a.cols + (size_t)s * a.cols_stride; u.status = a.stacked + s;
        u.blk_row = a.blk_row + i0; u.blk_len = a.blk_len + i0; u.n_blk = nb; u.m = (too_many || overflow) ? 0 : m;
        if (u.m > 0 && mode == 0 && a.compress) {
            if (r > 0) u.m = 0;
            else u.kdir = m <= a.kch ? m : 0;            // 0: k = upd_k(m, nc) = nc rows after the QR (m > kch >= nc compresses)
        } else if (u.m > 0 && mode == 0) {
            if (r < nch) {
                const int b0 = s_chunk[r], b1 = s_chunk[r + 1];
                int rows = 0;
                for (int k = b0; k < b1; ++k) rows += s_len[k];
                u.blk_row += b0; u.blk_len += b0; u.n_blk = b1 - b0; u.m = rows; u.kdir = rows;
            } else u.m = 0;
        } else if (r > 0) u.m = 0;                       // information form: one launch, round 0 only
        a.out[(size_t)r * a.S + s] = u;
    }
}

__global__ __launch_bounds__(UT) void update_back_batch_kernel(const UpdArgs* __restrict__ arr)
{
    const UpdArgs a = arr[blockIdx.x];
    if (a.m > 0) update_back(a);
}
// LDS of the two halves for a stream with m stacked rows over nc columns
static inline size_t update_front_lds(int m) { return sizeof(double) * (2 * (size_t)m + 8); }
static inline size_t update_back_lds(int m, int nc) { const size_t k = upd_k(m, nc); return sizeof(double) * (k * (k + 1) / 2 + 8 + 2 * (size_t)PANEL_Q * PANEL_W); }
// The kernels with dynamic LDS are allowed the whole 160 KB once, up front: the limit is process-wide state, and the
// stream groups of the batched filter launch concurrently from several host threads (a per-launch hipFuncSetAttribute
// with the launch's own size could lower the limit under another thread's launch).
static int msckf_lds_opt_in()
{
    static const int rc = [] {
        const int lim = 160 * 1024;
        const void* fns[6] = {reinterpret_cast<const void*>(feature_kernel8),
                              reinterpret_cast<const void*>(feature_kernel<16>), reinterpret_cast<const void*>(feature_kernel<64>), reinterpret_cast<const void*>(feature_kernel<256>),
                              reinterpret_cast<const void*>(update_front_kernel), reinterpret_cast<const void*>(update_front_batch_kernel)};
        const void* fns2[3] = {reinterpret_cast<const void*>(update_back_kernel), reinterpret_cast<const void*>(update_back_batch_kernel),
                               reinterpret_cast<const void*>(upd_info_kernel)};
        for (const void* f : fns2) {
            hipFuncAttributes at;
            hipError_t e = hipFuncGetAttributes(&at, f);
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lim - (int)at.sharedSizeBytes);
            if (e != hipSuccess) { av_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e)); return (int)AV_E_HIP; }
        }
        for (const void* f : fns) {
            // the dynamic limit excludes the kernel's static LDS (a few hundred bytes in the update kernels)
            hipFuncAttributes at;
            hipError_t e = hipFuncGetAttributes(&at, f);
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lim - (int)at.sharedSizeBytes);
            if (e != hipSuccess) { av_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e)); return (int)AV_E_HIP; }
        }
        return (int)AV_OK;
    }();
    return rc;
}

// diag(P)[12..14] of every stream (online_reset, msckf.py:829-835)
__global__ void pos_var_kernel(const double* P, size_t p_stride, int ld, int S, double* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * S) { int s = i / 3, k = 12 + i % 3; out[i] = P[s * p_stride + (size_t)k * ld + k]; }
}

}  // namespace

// ================================================================================================
// Host side / C ABI
// ================================================================================================
struct av_msckf {
    int device = 0;
    int max_cam = 0, ld = 0, rows_cap = 0, n = IMU_DIM;
    double* P = nullptr; double* Pn = nullptr; double* T = nullptr; double* Kt = nullptr; double* W = nullptr;
    double* Hblk = nullptr; double* rblk = nullptr; double* dx = nullptr; double* chi2 = nullptr; double* scratch = nullptr; int* cols_dev = nullptr;
    std::vector<void*> allocs;
};

namespace {
template <typename T>
int mk_alloc(av_msckf* c, T** p, size_t count)
{
    void* q = nullptr;
    AV_HIP(hipMalloc(&q, count * sizeof(T) + 256));
    AV_HIP(hipMemset(q, 0, count * sizeof(T) + 256));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return AV_OK;
}
}  // namespace

AV_EXPORT int av_msckf_create(int max_cam_states, int rows_cap, const double* chi2_table_100, int device, av_msckf** out)
{
    if (!out || max_cam_states < 2 || max_cam_states > 40 || rows_cap < 64 || !chi2_table_100) { av_set_error("av_msckf_create: bad arguments"); return AV_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { av_set_error("av_msckf_create: no HIP device visible"); return AV_E_NODEVICE; }
    AV_HIP(hipSetDevice(device));
    av_msckf* c = new (std::nothrow) av_msckf();
    if (!c) { av_set_error("out of host memory"); return AV_E_INVALID; }
    c->device = device; c->max_cam = max_cam_states; c->rows_cap = rows_cap;
    const int nmax = IMU_DIM + 6 * (max_cam_states + 1);
    c->ld = (nmax + 7) & ~7;
    int rc;
#define A(p, cnt) if ((rc = mk_alloc(c, &(p), (size_t)(cnt)))) { av_msckf_destroy(c); return rc; }
    A(c->P, (size_t)c->ld * c->ld) A(c->Pn, (size_t)c->ld * c->ld) A(c->T, (size_t)c->ld * c->ld) A(c->Kt, (size_t)c->ld * c->ld)
    A(c->scratch, (size_t)c->ld * c->ld)
    A(c->W, (size_t)rows_cap * (c->ld + 1)) A(c->Hblk, (size_t)rows_cap * c->ld) A(c->rblk, rows_cap) A(c->dx, c->ld) A(c->chi2, 100)
    A(c->cols_dev, c->ld)
#undef A
    AV_HIP(hipMemcpy(c->chi2, chi2_table_100, sizeof(double) * 100, hipMemcpyHostToDevice));
    *out = c;
    return AV_OK;
}

AV_EXPORT void av_msckf_destroy(av_msckf* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
}

AV_EXPORT int av_msckf_ld(const av_msckf* c) { return c ? c->ld : AV_E_INVALID; }
AV_EXPORT int av_msckf_dim(const av_msckf* c) { return c ? c->n : AV_E_INVALID; }

AV_EXPORT int av_msckf_set_cov(av_msckf* c, const double* P_host, int n, void* stream)
{
    if (!c || !P_host || n < IMU_DIM || n > c->ld || (n - IMU_DIM) % 6) { av_set_error("av_msckf_set_cov: bad arguments"); return AV_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    AV_HIP(hipSetDevice(c->device));
    AV_HIP(hipMemcpy2DAsync(c->P, sizeof(double) * c->ld, P_host, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyHostToDevice, st));
    AV_HIP(hipStreamSynchronize(st));
    c->n = n;
    return AV_OK;
}

AV_EXPORT int av_msckf_get_cov(av_msckf* c, double* P_host, int n, void* stream)
{
    if (!c || !P_host || n != c->n) { av_set_error("av_msckf_get_cov: n = %d but the filter has %d states", n, c ? c->n : -1); return AV_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    AV_HIP(hipSetDevice(c->device));
    AV_HIP(hipMemcpy2DAsync(P_host, sizeof(double) * n, c->P, sizeof(double) * c->ld, sizeof(double) * n, n, hipMemcpyDeviceToHost, st));
    AV_HIP(hipStreamSynchronize(st));
    return AV_OK;
}

AV_EXPORT int av_msckf_propagate(av_msckf* c, double dt, const double gyro[3], const double acc[3], const double q_old[4],
                                 const double q_new[4], const double q_null[4], const double v_null[3], const double p_null[3],
                                 const double v_new[3], const double p_new[3], const double gravity[3], const double noise[4],
                                 void* stream)
{
    if (!c || !gyro || !acc || !q_old || !q_new || !q_null || !v_null || !p_null || !v_new || !p_new || !gravity || !noise) {
        av_set_error("av_msckf_propagate: bad arguments");
        return AV_E_INVALID;
    }
    PropArgs a;
    a.P = c->P; a.n = c->n; a.ld = c->ld; a.dt = dt;
    for (int i = 0; i < 3; ++i) { a.gyro[i] = gyro[i]; a.acc[i] = acc[i]; a.v_null[i] = v_null[i]; a.p_null[i] = p_null[i]; a.v_new[i] = v_new[i]; a.p_new[i] = p_new[i]; a.gravity[i] = gravity[i]; }
    for (int i = 0; i < 4; ++i) { a.q_old[i] = q_old[i]; a.q_new[i] = q_new[i]; a.q_null[i] = q_null[i]; a.noise[i] = noise[i]; }
    AV_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(propagate_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

AV_EXPORT int av_msckf_augment(av_msckf* c, const double R_imu_cam0[9], const double skew_Rt_t[9], void* stream)
{
    if (!c || !R_imu_cam0 || !skew_Rt_t) { av_set_error("av_msckf_augment: bad arguments"); return AV_E_INVALID; }
    if (c->n + 6 > c->ld) { av_set_error("av_msckf_augment: more than %d camera states", c->max_cam + 1); return AV_E_CAPACITY; }
    AugArgs a;
    a.P = c->P; a.n = c->n; a.ld = c->ld;
    for (int i = 0; i < 9; ++i) { a.R_ic[i] = R_imu_cam0[i]; a.sk[i] = skew_Rt_t[i]; }
    AV_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(augment_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    AV_LAUNCH_CHECK();
    c->n += 6;
    return AV_OK;
}

AV_EXPORT int av_msckf_remove_cam(av_msckf* c, int cam_index, void* stream)
{
    const int ncam = c ? (c->n - IMU_DIM) / 6 : 0;
    if (!c || cam_index < 0 || cam_index >= ncam) { av_set_error("av_msckf_remove_cam: bad index"); return AV_E_INVALID; }
    AV_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(remove_cam_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, c->P, c->scratch, c->n, c->ld, IMU_DIM + 6 * cam_index);
    AV_LAUNCH_CHECK();
    c->n -= 6;
    return AV_OK;
}

AV_EXPORT int av_msckf_triangulate(av_msckf* c, int n_feat, const int32_t* obs_off_dev, const int32_t* obs_cam_dev, const double* obs_z_dev,
                                   const double* cam_q_dev, const double* cam_p_dev, const double* T_cam0_cam1_rowmajor44,
                                   const double* opt5, int max_views, double* pos_dev, int32_t* valid_dev, void* stream)
{
    if (!c || n_feat < 0 || !obs_off_dev || !obs_cam_dev || !obs_z_dev || !cam_q_dev || !cam_p_dev || !T_cam0_cam1_rowmajor44 || !opt5 || !pos_dev || !valid_dev) {
        av_set_error("av_msckf_triangulate: bad arguments");
        return AV_E_INVALID;
    }
    if (max_views > 64) { av_set_error("av_msckf_triangulate: %d views per feature exceed one wavefront (64)", max_views); return AV_E_CAPACITY; }
    if (n_feat == 0) return AV_OK;
    TriArgs a; memset(&a, 0, sizeof(a));
    a.n_feat = n_feat; a.obs_off = obs_off_dev; a.obs_cam = obs_cam_dev; a.obs_z = obs_z_dev; a.cam_q = cam_q_dev; a.cam_p = cam_p_dev;
    a.feat_stream = nullptr; a.cam_stride = 0;
    for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) a.R01[r * 3 + cc] = T_cam0_cam1_rowmajor44[r * 4 + cc]; a.t01[r] = T_cam0_cam1_rowmajor44[r * 4 + 3]; }
    a.huber = opt5[0]; a.precision = opt5[1]; a.damping = opt5[2]; a.outer_max = (int)opt5[3]; a.inner_max = (int)opt5[4];
    a.out_pos = pos_dev; a.out_valid = valid_dev;
    AV_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(triangulate_kernel, dim3((n_feat + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

AV_EXPORT int av_msckf_feature_blocks(av_msckf* c, int n_feat, int n_cam, int max_obs, const int32_t* obs_off_dev, const int32_t* obs_cam_dev,
                                      const double* obs_z_dev, const double* pos_dev, const int32_t* dof_dev, const int32_t* row_off_dev, int total_rows,
                                      const double* cam_q_dev, const double* cam_p_dev, const double* cam_qn_dev, const double* cam_pn_dev,
                                      const double* T_cam0_cam1_rowmajor44, const double gravity[3], double obs_noise,
                                      double* gamma_dev, int32_t* pass_dev, void* stream)
{
    if (!c || n_feat < 0 || !obs_off_dev || !obs_cam_dev || !obs_z_dev || !pos_dev || !dof_dev || !row_off_dev || !cam_q_dev || !cam_p_dev ||
        !cam_qn_dev || !cam_pn_dev || !T_cam0_cam1_rowmajor44 || !gravity || !gamma_dev || !pass_dev) {
        av_set_error("av_msckf_feature_blocks: bad arguments");
        return AV_E_INVALID;
    }
    if (IMU_DIM + 6 * n_cam != c->n) { av_set_error("av_msckf_feature_blocks: %d camera states but the covariance has %d states", n_cam, c->n); return AV_E_INVALID; }
    if (total_rows > c->rows_cap) { av_set_error("av_msckf_feature_blocks: %d stacked rows exceed rows_cap %d", total_rows, c->rows_cap); return AV_E_CAPACITY; }
    if (max_obs < 2 || max_obs > c->max_cam + 1) { av_set_error("av_msckf_feature_blocks: max_obs %d out of range", max_obs); return AV_E_INVALID; }
    if (n_feat == 0) return AV_OK;
    FeatArgs a; memset(&a, 0, sizeof(a));
    a.n_feat = n_feat; a.n_cam = n_cam; a.ld = c->ld;
    a.obs_off = obs_off_dev; a.obs_cam = obs_cam_dev; a.obs_z = obs_z_dev; a.pos = pos_dev; a.dof = dof_dev; a.row_off = row_off_dev;
    a.cam_q = cam_q_dev; a.cam_p = cam_p_dev; a.cam_qn = cam_qn_dev; a.cam_pn = cam_pn_dev; a.P = c->P; a.chi2 = c->chi2;
    for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) a.R01[r * 3 + cc] = T_cam0_cam1_rowmajor44[r * 4 + cc]; a.t01[r] = T_cam0_cam1_rowmajor44[r * 4 + 3]; a.gravity[r] = gravity[r]; }
    a.obs_noise = obs_noise; a.Hout = c->Hblk; a.rout = c->rblk; a.gamma = gamma_dev; a.pass = pass_dev; a.Mmax = max_obs;
    a.feat_stream = nullptr; a.stream_ncam = nullptr; a.stream_gravity = nullptr; a.cam_stride = 0; a.p_stride = a.h_stride = a.r_stride = 0;
    a.feat_list = nullptr; a.prof = nullptr; a.zero_fill = 1; a.tri_idx = nullptr; a.tri_pos = nullptr; a.tri_valid = nullptr;
    AV_HIP(hipSetDevice(c->device));
    return launch_feature_kernel(a, n_feat, max_obs, (hipStream_t)stream);
}

AV_EXPORT int av_msckf_update(av_msckf* c, const int32_t* blk_row_dev, const int32_t* blk_len_dev, int n_blk, int total_rows,
                              double obs_noise, double* dx_host, void* stream)
{
    if (!c || n_blk < 0 || total_rows < 0 || !dx_host || (n_blk > 0 && (!blk_row_dev || !blk_len_dev))) { av_set_error("av_msckf_update: bad arguments"); return AV_E_INVALID; }
    if (total_rows > c->rows_cap) { av_set_error("av_msckf_update: %d rows exceed rows_cap %d", total_rows, c->rows_cap); return AV_E_CAPACITY; }
    hipStream_t st = (hipStream_t)stream;
    if (total_rows == 0) { for (int i = 0; i < c->n; ++i) dx_host[i] = 0.0; return AV_OK; }
    UpdArgs a;
    memset(&a, 0, sizeof(a));
    a.P = c->P; a.n = c->n; a.ld = c->ld; a.Hsrc = c->Hblk; a.rsrc = c->rblk; a.blk_row = blk_row_dev; a.blk_len = blk_len_dev; a.n_blk = n_blk;
    a.prof = nullptr;
    a.W = c->W; a.ldt = c->rows_cap; a.T = c->T; a.Kt = c->Kt; a.Pn = c->Pn; a.dx = c->dx; a.obs_noise = obs_noise; a.m = total_rows;
    // non-zero columns: every camera block (the caller's blocks may involve any of them); the IMU columns are zero
    std::vector<int> cols;
    for (int q = IMU_DIM; q < c->n; ++q) cols.push_back(q);
    AV_HIP(hipSetDevice(c->device));
    AV_HIP(hipMemcpyAsync(c->cols_dev, cols.data(), sizeof(int) * cols.size(), hipMemcpyHostToDevice, st));
    AV_HIP(hipStreamSynchronize(st));
    a.cols = c->cols_dev; a.nc = (int)cols.size();
    const size_t lds_f = update_front_lds(total_rows), lds_b = update_back_lds(total_rows, a.nc);
    if (lds_f > 160 * 1024 || lds_b > 160 * 1024) { av_set_error("av_msckf_update: %d rows need %zu B of LDS", total_rows, lds_f > lds_b ? lds_f : lds_b); return AV_E_CAPACITY; }
    if (n_blk > 4 * UT) { av_set_error("av_msckf_update: %d blocks exceed %d", n_blk, 4 * UT); return AV_E_CAPACITY; }
    { int rc0 = msckf_lds_opt_in(); if (rc0) return rc0; }
    hipLaunchKernelGGL(update_front_kernel, dim3(1), dim3(UT), lds_f, st, a);
    hipLaunchKernelGGL(update_back_kernel, dim3(1), dim3(UT), lds_b, st, a);
    AV_LAUNCH_CHECK();
    AV_HIP(hipMemcpyAsync(dx_host, c->dx, sizeof(double) * c->n, hipMemcpyDeviceToHost, st));
    AV_HIP(hipStreamSynchronize(st));
    return AV_OK;
}


#include "msckf_mfma.inc"

// The back end of one round of batched updates: T^T, S, its factor, the substitution, the covariance update -- for all S streams of
// `arr` (streams without an update of this kind exit on their first instruction).  kmax / nmax: the longest pass / largest state the
// grids have to cover.  skipk: timing experiments of the host-store path (AV_MSCKF_SKIP).
static inline void launch_upd_back(const UpdArgs* arr, int S, int kmax, int nmax, hipStream_t stm, int skipk = 0)
{
    const int bk = (kmax + MBW - 1) / MBW, bn = (nmax + MBW - 1) / MBW;
    if (!(skipk & 8)) hipLaunchKernelGGL(upd_tt_mfma_kernel, dim3(mfma_grid(bn * bk, S)), dim3(64), 0, stm, arr, S, bn, bk);
    if (!(skipk & 8)) hipLaunchKernelGGL(upd_s_mfma_kernel, dim3(mfma_grid(bk * (bk + 1) / 2, S)), dim3(64), 0, stm, arr, S, bk);
    if (!(skipk & 16)) hipLaunchKernelGGL(upd_chol_mfma_kernel, dim3(mfma_grid(1, S)), dim3(64), 0, stm, arr, S);
    const int parts = nmax / 16 + 1;
    if (!(skipk & 32)) hipLaunchKernelGGL(upd_fsolve_mfma_kernel, dim3(mfma_grid(parts, S)), dim3(64), 0, stm, arr, S, parts);
    if (!(skipk & 64)) hipLaunchKernelGGL(upd_p_mfma_kernel, dim3(mfma_grid(bn * (bn + 1) / 2, S)), dim3(64), 0, stm, arr, S, bn);
}
// Row compression of the streams on `list` (n_list_dev: device-written count, the tasks stride over the list; NULL: n_list entries):
// Gram matrix of [Hc | r], then its bordered Cholesky factor -> the k = n_c rows [F | f] of W
static inline void launch_upd_compress(const UpdArgs* arr, const int* list, const int* n_list_dev, unsigned n_list, int k1max, hipStream_t stm)
{
    const int bk = (k1max + MBW - 1) / MBW, slots = 8 * (int)((n_list + 7) / 8);
    hipLaunchKernelGGL(upd_gram_mfma_kernel, dim3((unsigned)(bk * (bk + 1) / 2) * (unsigned)slots), dim3(64), 0, stm, arr, list, n_list_dev, (int)n_list, slots, bk);
    hipLaunchKernelGGL(upd_gram_chol_mfma_kernel, dim3(n_list), dim3(64), 0, stm, arr, list, n_list_dev, (int)n_list);
}
#include "msckf_batch.inc"
