/*
 * airvision.h -- C ABI of libairvision_hip.so, the MI355X (gfx950) implementation of
 * UAV-Airvision's per-frame hot path.
 *
 * The reference (BUBLET/uav-airvision) is pure Python; its "FFI" for this path is the set of
 * third-party native calls listed in SURVEY.md section 2.1 (K1-K13) plus the Python call surface of
 * section 8(b).  Each entry point below names the reference interface it replaces (file:line,
 * relative to the reference's src/).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions: every function returns 0 on success and a negative AV_E_*
 * code on error, never throws; `*_dev` pointers are device (HBM) pointers, everything else is host
 * memory owned by the caller and only read/written during the call; the library owns the device
 * memory of a context; there is no process-global state; `stream` is a hipStream_t (NULL = the
 * default stream).  Unless a function says it synchronises, work is only ENQUEUED on `stream`.
 */
#ifndef AIRVISION_H
#define AIRVISION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AV_OK            0
#define AV_E_INVALID    -1   /* bad argument / unsupported shape */
#define AV_E_HIP        -2   /* a HIP runtime call failed (see av_last_error) */
#define AV_E_CAPACITY   -3   /* a device-side capacity was exceeded (e.g. FAST corners > max_corners) */
#define AV_E_NODEVICE   -4   /* no gfx950 device visible */
#define AV_E_NUMERIC    -5   /* a factorisation met a non-positive or non-finite pivot (per-stream: that stream is stopped) */

#define AV_MAX_LEVELS    5   /* pyramid levels 0..4 (the reference uses maxLevel = 3, config.py:34) */
#define AV_PYR_BORDER   16   /* border (pixels) of every padded pyramid level; >= LK win + 1 */

/* Thread-local text of the last error raised on the calling thread. */
const char* av_last_error(void);
/* Library version / build info string. */
const char* av_version(void);
/* Number of visible HIP devices (does not initialise a context beyond the count). */
int av_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Padded u8 pyramids.  A pyramid is one contiguous device allocation holding levels 0..levels-1,
 * each level stored with an AV_PYR_BORDER-pixel BORDER_REFLECT_101 frame (what OpenCV's
 * buildOpticalFlowPyramid produces internally for calcOpticalFlowPyrLK).
 * ------------------------------------------------------------------------------------------- */
typedef struct av_pyr_layout {
    int32_t levels;
    int32_t w[AV_MAX_LEVELS], h[AV_MAX_LEVELS];      /* interior size of each level                */
    int32_t pitch[AV_MAX_LEVELS];                    /* bytes per padded row                       */
    int64_t offset[AV_MAX_LEVELS];                   /* byte offset of the padded level's (0,0)    */
    int64_t bytes;                                   /* total bytes of one pyramid (16-B multiple) */
} av_pyr_layout;

int av_pyramid_layout(int w, int h, int levels, av_pyr_layout* out);

/* Build n_img pyramids.  Image i is at img_dev + i*img_stride (tightly packed w*h u8), pyramid i
 * at pyr_dev + i*pyr_stride.  Replaces the pyrDown chain inside cv2.calcOpticalFlowPyrLK
 * (reference: image_processing/pyramid_builder.py:22-48 is a pass-through). */
int av_pyramid_build(const uint8_t* img_dev, int64_t img_stride, int n_img, int w, int h, int levels,
                     uint8_t* pyr_dev, int64_t pyr_stride, void* stream);

/* ---------------------------------------------------------------------------------------------
 * cv2.calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts, winSize=(win,win),
 *   maxLevel=levels-1, criteria=(EPS|COUNT, max_iter, eps), flags=OPTFLOW_USE_INITIAL_FLOW)
 * Reference call sites: image_processing/feature_tracker.py:102-108,
 * image_processing/stereo_matcher.py:64-68 and 70-74, parameters config.py:31-44.
 * Batched over n_set point sets: set i tracks count_dev[i] (<= cap) points from pyramid
 * pyrI_dev + i*pyr_stride into pyrJ_dev + i*pyr_stride.  prev/next are float32 (x,y) pairs at
 * [i*cap + k]; next holds the initial guess on entry and the result on return; status is u8.
 * win: 3 .. 31 (config.win_size: the reference's 15 x 15 runs the 16-lanes-per-point kernel, any other size the general
 * one-wavefront-per-point kernel: same arithmetic, same results rule -- bit-identical to the CPU oracle); levels: 1 .. AV_MAX_LEVELS.
 * ------------------------------------------------------------------------------------------- */
int av_lk_track(const uint8_t* pyrI_dev, const uint8_t* pyrJ_dev, int64_t pyr_stride, int n_set,
                int w, int h, int levels,
                const float* prev_dev, float* next_dev, uint8_t* status_dev, const int32_t* count_dev, int cap,
                int win, int max_iter, double eps, double min_eig_threshold, void* stream);

/* ---------------------------------------------------------------------------------------------
 * cv2.FastFeatureDetector_create(threshold).detect(img, mask)   (TYPE_9_16, NMS on)
 * Reference: image_processing/pipeline.py:23-25, feature_initializer.py:52, feature_adder.py:64.
 * Batched over n_img tightly packed images (and optional masks, NULL = no mask).  For image i the
 * number of keypoints is written to count_dev[i] and keypoint k is packed into
 * kp_dev[i*cap + k] = score << 19 | (2^19 - 1 - (y*w + x)); keypoints are UNORDERED (sort the
 * packed words descending within equal score to recover raster order).  Requires w*h <= 2^19.
 * If more than cap keypoints exist count_dev[i] still reports the true number.
 * ------------------------------------------------------------------------------------------- */
int av_fast_detect(const uint8_t* img_dev, int64_t img_stride, const uint8_t* mask_dev, int64_t mask_stride,
                   int n_img, int w, int h, int threshold, uint32_t* kp_dev, int32_t* count_dev, int cap,
                   void* stream);

/* ---------------------------------------------------------------------------------------------
 * cv2.undistortPoints(pts, K, D, None, R, P = identity) for pinhole + radtan
 * (reference: image_processing/camera_model.py:24-47, feature_publisher.py:24-59), and
 * cv2.projectPoints(convertPointsToHomogeneous(pts), 0, 0, K, D)
 * (reference: image_processing/camera_model.py:49-75).  fp64 in, fp64 out; intr = [fx fy cx cy],
 * dist = [k1 k2 p1 p2], R = row-major 3x3 (host pointers, copied by value).
 * ------------------------------------------------------------------------------------------- */
int av_undistort_points(const double* pts_dev, int n, const double* intr, const double* dist, const double* R,
                        double* out_dev, void* stream);
int av_distort_points(const double* pts_dev, int n, const double* intr, const double* dist,
                      double* out_dev, void* stream);

/* The same two operators with the distortion model as an argument (camera_model.py:41-46, 69-74; feature_publisher.py:53-58,
 * 82-87): AV_DISTORTION_RADTAN = the two above; AV_DISTORTION_EQUIDISTANT = cv2.fisheye.undistortPoints(pts, K, D, R, P = identity)
 * and cv2.fisheye.distortPoints(pts, K, D), dist = [k1 k2 k3 k4] of the Kannala-Brandt model.  (Parity of the equidistant branch
 * is unpinned: restated from OpenCV 4.x fisheye.cpp, no cv2 to check against) */
#define AV_DISTORTION_RADTAN 0
#define AV_DISTORTION_EQUIDISTANT 1
int av_undistort_points_model(const double* pts_dev, int n, const double* intr, const double* dist, const double* R, int model,
                              double* out_dev, void* stream);
int av_distort_points_model(const double* pts_dev, int n, const double* intr, const double* dist, int model,
                            double* out_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The image front-end as one device-resident engine over n_streams independent stereo streams:
 * ImageProcessingPipeline.__init__ / imu_callback / stereo_callback
 * (reference: image_processing/pipeline.py:15-40, 42-44, 46-150) with all of its feature_* stages
 * (feature_initializer.py:45-85, feature_tracker.py:74-177, stereo_matcher.py:33-115,
 * feature_adder.py:52-108, feature_pruner.py:8-19, feature_publisher.py:90-121,
 * imu_processor.py:22-67).
 * ------------------------------------------------------------------------------------------- */
typedef struct av_frontend_config {
    int32_t width, height;                   /* config.cam0_resolution (config.py:102)             */
    int32_t grid_row, grid_col;              /* config.py:23-24                                     */
    int32_t grid_min_feature_num;            /* config.py:26                                        */
    int32_t grid_max_feature_num;            /* config.py:27                                        */
    int32_t fast_threshold;                  /* config.py:28                                        */
    int32_t lk_win;                          /* config.patch_size (config.py:35); 3 .. 31           */
    int32_t lk_levels;                       /* config.pyramid_levels + 1 (config.py:34)            */
    int32_t lk_max_iter;                     /* config.max_iteration (config.py:31)                 */
    int32_t max_corners;                     /* capacity for FAST keypoints of one image            */
    int32_t flags;                           /* AV_FE_* bits below                                  */
    double  lk_eps;                          /* config.track_precision (config.py:32)               */
    double  lk_min_eig;                      /* OpenCV default minEigThreshold = 1e-4               */
    double  stereo_threshold;                /* config.py:30                                        */
    double  cam0_intrinsics[4], cam0_distortion[4];   /* config.py:99-101                           */
    double  cam1_intrinsics[4], cam1_distortion[4];   /* config.py:118-120                          */
    double  R_cam0_imu[9], R_cam1_imu[9];    /* inv(T_imu_cam*)[:3,:3] (imu_processor.py:10-16)     */
    double  R0to1[9];                        /* R_cam1_imu.T @ R_cam0_imu (stereo_matcher.py:47)    */
    double  E[9];                            /* skew(t01) @ R0to1 (stereo_matcher.py:90-91)         */
    double  norm_unit;                       /* 4/(2fx+2fy) of cam0 (stereo_matcher.py:103-104)     */
    int32_t cam0_distortion_model;           /* AV_DISTORTION_* (config.py:98); 0 = radtan          */
    int32_t cam1_distortion_model;           /* config.py:117                                       */
} av_frontend_config;

typedef struct av_frontend av_frontend;

int  av_frontend_create(const av_frontend_config* cfg, int n_streams, int device, av_frontend** out);
void av_frontend_destroy(av_frontend* fe);

/* ImageProcessingPipeline.imu_callback (pipeline.py:42-44 -> imu_processor.py:22-26).  Only the
 * gyro is used by the front-end.  Thread-safe against av_frontend_step* on other threads. */
int av_frontend_push_imu(av_frontend* fe, int stream, double timestamp, const double gyro[3]);

/* n IMU samples in one call: sample i goes to stream stream_idx[i]; gyro is [n][3]. */
int av_frontend_push_imu_batch(av_frontend* fe, const int32_t* stream_idx, const double* timestamps,
                               const double* gyro, int n);

/* av_frontend_config.flags.  AV_FE_INPUTS_PERSIST: the device images handed to av_frontend_step stay valid and unmodified
 * until the kernels of the NEXT av_frontend_step of this engine have completed.  The engine then reads pyramid level 0 in place
 * (LK and FAST index the caller's image, with BORDER_REFLECT_101 arithmetic at the border) and builds only levels 1..3 -- no
 * padded copy of the 752x480 level: 0.8 MB less HBM traffic per stereo frame.  Without the flag the inputs are only read during
 * the call's own kernels and level 0 is copied (the round-1/2 behaviour).  av_frontend_step_host always works in place: the
 * images live in the library's own staging slots.  Results are bit-identical either way. */
#define AV_FE_INPUTS_PERSIST 1

/* ImageProcessingPipeline.stereo_callback for every stream at once (pipeline.py:46-150).
 * Stream s reads its cam0/cam1 images (tightly packed width*height u8, device memory) at
 * img0_dev + s*img_stride and img1_dev + s*img_stride; timestamps[s] is the frame time.  All
 * kernels are enqueued on `stream`; nothing is synchronised. */
int av_frontend_step(av_frontend* fe, const uint8_t* img0_dev, const uint8_t* img1_dev, int64_t img_stride,
                     const double* timestamps, void* stream);
/* Same with host images (the drop-in boundary hands over numpy arrays): copies H2D, steps. */
/* Builds the pyramids of the images the NEXT av_frontend_step will be given, now, behind whatever is enqueued on `stream` (needs
 * AV_FE_INPUTS_PERSIST; the step that follows with the same pointers skips its own pyramid launch; results are identical -- the
 * launch only changes its place in the stream).  pipeline.py:46-150 builds a frame's pyramids when the frame arrives; a caller
 * that knows its next frame (a replay, a queue of camera frames: streaming/dataset.py:93-158) can have them built while the
 * filter works on the frame before. */
int av_frontend_prestage(av_frontend* fe, const uint8_t* img0_dev, const uint8_t* img1_dev, int64_t img_stride, void* stream);

int av_frontend_step_host(av_frontend* fe, const uint8_t* img0_host, const uint8_t* img1_host, int64_t img_stride,
                          const double* timestamps, void* stream);

/* Shared frame store -- for sweeps that replay the SAME frames on several streams: the reference's run.bat:4-12 runs every
 * sequence from several start offsets, and an offset only moves the start index (streaming/dataset.py:206-214), so every frame
 * of the sequence is read by every offset stream a few steps apart.  A frame put into the store is copied to the device, gets
 * both its pyramids (levels 1.., level 0 is read in place) and its FAST pass ONCE (the detector's lists are independent of the
 * stream: the per-stream mask of feature_adder.py:56-62 is applied when the lists are binned into a stream's cells); the streams
 * only carry an entry number per step.
 *   av_frontend_frames_reserve   allocate n_slots entries (~2 MB each at 752 x 480 and four levels); once per engine.
 *   av_frontend_frames_upload    n host frame pairs (frame i: img0_host + i*img_stride, img1_host + i*img_stride, tightly packed
 *                                width*height u8) into entries slots[i].  The copies and kernels run on the engine's copy stream,
 *                                behind the newest step ENQUEUED so far (the entries must be free as of that step) and beside
 *                                whatever is enqueued afterwards: upload the frames of step k+1 before enqueueing step k to overlap
 *                                the two.  The host arrays are free again when the call returns.
 *   av_frontend_step_frames      av_frontend_step with stream s reading entry slot_of_stream[s] (its previous frame's entry is
 *                                remembered by the engine and must still hold that frame).  A NEGATIVE entry = the stream has no
 *                                frame in this step (its sequence is over): none of its kernels' workgroups do anything, its
 *                                published count reads 0, its state stays as it is.  Results are bit-identical to av_frontend_step
 *                                on the same images. */
int av_frontend_frames_reserve(av_frontend* fe, int n_slots);
int av_frontend_frames_upload(av_frontend* fe, const int32_t* slots, int n, const uint8_t* img0_host, const uint8_t* img1_host,
                              int64_t img_stride, void* stream);
int av_frontend_step_frames(av_frontend* fe, const int32_t* slot_of_stream, const double* timestamps, void* stream);

/* Host-side staging for sweeps: decode n 8-bit greyscale, non-interlaced PNG files of width x height (the EuRoC camera
 * frames; reference: streaming/dataset.py:101 `cv2.imread(path, -1)` on the reader threads of dataset.py:93-158) on
 * `threads` host threads, file i into out + i*out_stride -- e.g. straight into the [S][h][w] batches handed to
 * av_frontend_step_host, one step ahead of the GPU.  status[i] (optional) = 0 ok, 1 unsupported flavour of PNG (other depth /
 * colour type / size / interlaced: the caller may decode that file by other means), 2 unreadable or corrupt.  Returns AV_OK,
 * AV_E_CAPACITY if the worst status is 1, AV_E_INVALID if any file was unreadable.  Pure host function. */
int av_png_decode_gray8(const char* const* paths, int n, int width, int height, uint8_t* out, int64_t out_stride, int threads, int32_t* status);

/* Capacity (features per stream) of the published feature message = grid_num * grid_max. */
int av_frontend_max_features(const av_frontend* fe);

/* The feature_msg of the last step (feature_publisher.py:109-121), synchronising `stream` first.
 * For stream s: n_out[s] features; ids at ids_out[s*cap + k]; (u0,v0,u1,v1) at uv_out[(s*cap+k)*4].
 * cap must be >= av_frontend_max_features.  Returns AV_E_CAPACITY if any stream overflowed a
 * device-side buffer during the step (results of that stream are then not parity-exact). */
int av_frontend_read_features(av_frontend* fe, int64_t* ids_out, double* uv_out, int32_t* n_out, int cap, void* stream);
/* The same read-back in two halves: _begin enqueues the device-to-host copies into pinned slot 0/1 behind the work
 * already on `stream` and returns; _end waits for those copies only and unpacks them.  Enqueueing the next
 * av_frontend_step between the two overlaps it with the consumption of this frame's features. */
int av_frontend_read_features_begin(av_frontend* fe, int slot, void* stream);
int av_frontend_read_features_end(av_frontend* fe, int slot, int64_t* ids_out, double* uv_out, int32_t* n_out, int cap);
/* The same feature_msg where the last step left it, ON THE DEVICE: ids int64[S][cap], uv double[S][cap][4], n int32[S] with
 * cap = av_frontend_max_features.  The buffers are rewritten by the next av_frontend_step; consume them with work enqueued on the
 * stream the step ran on (av_msckf_batch_submit_dev copies them there).  This is the hand-over of pipeline.py:131-143 ->
 * modules/vio.py:34-36,46-51 (feature_queue) without the host in between. */
int av_frontend_features_dev(av_frontend* fe, const int64_t** ids_dev, const double** uv_dev, const int32_t** n_dev, int* cap);

/* Pipeline state visible to callers (pipeline.py:33-40): the grid of the frame just published
 * (= prev_features after the callback returns).  Per feature k of stream `stream`:
 * ids[k], lifetime[k], cell[k], pts[k*4] = cam0 x,y, cam1 x,y (pixels, float32).  Synchronises. */
int av_frontend_read_grid(av_frontend* fe, int stream_idx, int64_t* ids, int32_t* lifetime, int32_t* cell,
                          float* pts, int cap, int32_t* n_out, int64_t* next_feature_id, void* stream);

/* Stage counters of the last step for one stream (feature_tracker.py:96,123,133,157 and the
 * adder): [before_tracking, after_tracking, after_matching, n_fast_corners, n_candidates, n_new,
 * n_published, overflow_flags].  Synchronises. */
int av_frontend_read_counters(av_frontend* fe, int stream_idx, int32_t out[8], void* stream);
/* New-feature candidates are stereo-matched lazily (feature_adder.py:80-108 matches all of them and keeps the grid_min
 * best inliers per cell; the first grid_min + 2 candidates of a cell decide that unless too few of them are inliers):
 * [candidates matched in round 1, in round 2] of the last step, i.e. the LK point passes actually run.  Synchronises. */
int av_frontend_read_match_counts(av_frontend* fe, int stream_idx, int32_t out[2], void* stream);

/* ---------------------------------------------------------------------------------------------
 * MSCKF back-end: the batched small-dense fp64 linear algebra of MSCKF.feature_callback
 * (reference: msckf.py:177-228).  The covariance P lives on the device inside the context (row-major,
 * leading dimension av_msckf_ld); state vectors and the feature/camera bookkeeping stay with the
 * caller (the Python MSCKF class mirrors the reference's dict bookkeeping).  Per-feature inputs are
 * CSR-style: observations of feature f are obs_off[f] .. obs_off[f+1]-1, each with the index of its
 * camera state (position in the cam_states dict) and z = (u0, v0, u1, v1).
 * ------------------------------------------------------------------------------------------- */
typedef struct av_msckf av_msckf;

/* chi2_table_100[d] = chi2.ppf(0.05, d) for d = 1..99 (msckf.py:111-113); rows_cap = capacity of the
 * stacked Jacobian in rows (the prune path of the reference is uncapped). */
int  av_msckf_create(int max_cam_states, int rows_cap, const double* chi2_table_100, int device, av_msckf** out);
void av_msckf_destroy(av_msckf* ctx);
int  av_msckf_ld(const av_msckf* ctx);
int  av_msckf_dim(const av_msckf* ctx);
/* state_cov <- host matrix n x n (reset_state_cov msckf.py:788-798, online_reset :843); / read back. */
int  av_msckf_set_cov(av_msckf* ctx, const double* P_host, int n, void* stream);
int  av_msckf_get_cov(av_msckf* ctx, double* P_host, int n, void* stream);
/* MSCKF.process_model covariance part (msckf.py:282-335) for one IMU sample: builds F, G, Phi (3rd
 * order), applies the observability fix with the null-space states, Q = Phi G Qc G^T Phi^T dt,
 * P11 <- Phi P11 Phi^T + Q, P12 <- Phi P12, symmetrise.  gyro/acc are bias-corrected; q_old is the
 * orientation before predict_new_state, q_new/v_new/p_new after; noise = continuous variances
 * [gyro, gyro_bias, acc, acc_bias] (msckf.py:123-127). */
int  av_msckf_propagate(av_msckf* ctx, double dt, const double gyro[3], const double acc[3], const double q_old[4],
                        const double q_new[4], const double q_null[4], const double v_null[3], const double p_null[3],
                        const double v_new[3], const double p_new[3], const double gravity[3], const double noise[4],
                        void* stream);
/* MSCKF.state_augmentation covariance part (msckf.py:407-423): J = [R_i_c 0 0 0 0 I 0; skew(R_w_i^T t_c_i) 0 0 0 I 0 I]. */
int  av_msckf_augment(av_msckf* ctx, const double R_imu_cam0[9], const double skew_Rt_t[9], void* stream);
/* prune_cam_state_buffer (msckf.py:774-786): drop the 6 rows/cols of camera state #cam_index. */
int  av_msckf_remove_cam(av_msckf* ctx, int cam_index, void* stream);
/* Feature.initialize_position (feature/feature_position_initializer.py:6-76) for n_feat features.
 * opt5 = [huber_epsilon, estimation_precision, initial_damping, outer_loop_max, inner_loop_max]
 * (config.py:7-17); max_views = 2 * (largest observation count) <= 64. */
int  av_msckf_triangulate(av_msckf* ctx, int n_feat, const int32_t* obs_off_dev, const int32_t* obs_cam_dev, const double* obs_z_dev,
                          const double* cam_q_dev, const double* cam_p_dev, const double* T_cam0_cam1_rowmajor44,
                          const double* opt5, int max_views, double* pos_dev, int32_t* valid_dev, void* stream);
/* MSCKF.feature_jacobian + gating_test (msckf.py:509-546, 604-612) for n_feat features: writes the
 * null-space-projected rows (4M-3 per feature, starting at row_off[f]) into the context's block
 * buffer and gamma / pass per feature.  dof[f] = chi^2 degrees of freedom (msckf.py:662, 761). */
int  av_msckf_feature_blocks(av_msckf* ctx, int n_feat, int n_cam, int max_obs, const int32_t* obs_off_dev, const int32_t* obs_cam_dev,
                             const double* obs_z_dev, const double* pos_dev, const int32_t* dof_dev, const int32_t* row_off_dev, int total_rows,
                             const double* cam_q_dev, const double* cam_p_dev, const double* cam_qn_dev, const double* cam_pn_dev,
                             const double* T_cam0_cam1_rowmajor44, const double gravity[3], double obs_noise,
                             double* gamma_dev, int32_t* pass_dev, void* stream);
/* MSCKF.measurement_update (msckf.py:548-602) on the blocks (first row, length) selected by the
 * caller after gating: thin QR when rows > n, S, gain, P <- sym((I-KH)P).  Synchronises and returns
 * delta_x (n doubles) for the caller's state injection (msckf.py:568-595). */
int  av_msckf_update(av_msckf* ctx, const int32_t* blk_row_dev, const int32_t* blk_len_dev, int n_blk, int total_rows,
                     double obs_noise, double* dx_host, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Quaternion / rotation helpers of the filter, for host callers (replaces the functions of src/utils.py; JPL
 * quaternions [x, y, z, w]).  Pure host functions (no device needed); rotation matrices are 9 doubles, row-major.
 *   av_quat_to_rotation       to_rotation(q)                   utils.py:12-23   (normalises q first)
 *   av_rotation_to_quat       to_quaternion(R)                 utils.py:25-47
 *   av_quat_multiply          quaternion_multiplication(a, b)  utils.py:61-76   (normalises inputs and output)
 *   av_quat_small_angle       small_angle_quaternion(dtheta)   utils.py:79-93   (|dtheta/2|^2 <= 1 branch rule)
 *   av_quat_from_two_vectors  from_two_vectors(v0, v1)         utils.py:96-120  (antiparallel / parallel fallbacks)
 * ------------------------------------------------------------------------------------------- */
int  av_quat_to_rotation(const double q[4], double R_rowmajor9[9]);
int  av_rotation_to_quat(const double R_rowmajor9[9], double q[4]);
int  av_quat_multiply(const double q1[4], const double q2[4], double out[4]);
int  av_quat_small_angle(const double dtheta[3], double q[4]);
int  av_quat_from_two_vectors(const double v0[3], const double v1[3], double q[4]);

/* ---------------------------------------------------------------------------------------------
 * n_streams independent MSCKF filters stepped together (the throughput path of the back-end).  The
 * bookkeeping of MSCKF.feature_callback (msckf.py:177-228) runs in C++ inside the library for all
 * streams; each numeric phase is one batched launch over all streams (same kernels as av_msckf_*).
 *   R_imu_cam0_t_cam0_imu12 = [inv(T_imu_cam0)[:3,:3].T (9, row-major), inv(T_imu_cam0)[:3,3] (3)] (msckf.py:132-134)
 *   cov_init5 = [gyro_bias_cov, velocity_cov, acc_bias_cov, extrinsic_rotation_cov, extrinsic_translation_cov]
 *   noise4    = [gyro_noise, gyro_bias_noise, acc_noise, acc_bias_noise]        (config.py:71-75)
 *   opt6      = [huber_epsilon, estimation_precision, initial_damping, outer_max, inner_max, translation_threshold]
 * ------------------------------------------------------------------------------------------- */
typedef struct av_msckf_batch av_msckf_batch;
int  av_msckf_batch_create(int n_streams, int max_cam_states, int rows_cap, const double* chi2_table_100, const double gravity[3],
                           const double* T_cam0_cam1_rowmajor44, const double* R_imu_cam0_t_cam0_imu12, const double cov_init5[5],
                           const double noise4[4], double obs_noise, double position_std_threshold, const double velocity0[3],
                           const double* opt6, int device, av_msckf_batch** out);
void av_msckf_batch_destroy(av_msckf_batch* b);
/* MSCKF.imu_callback for n samples (msckf.py:162-175 incl. initialize_gravity_and_bias); gyro/acc are [n][3]. */
int  av_msckf_batch_push_imu(av_msckf_batch* b, const int32_t* stream_idx, const double* timestamps, const double* gyro, const double* acc, int n);
/* MSCKF.feature_callback for every stream.  Host inputs: stream s has n_feat[s] features, ids[s*cap+k],
 * uv[(s*cap+k)*4..] = u0 v0 u1 v1.  out[s*12..] = {published (0/1; -1 = stream stopped, av_msckf_batch_stream_status),
 * t, p[3], q[4] (JPL xyzw), v[3]}.
 * A stream is live for a frame iff its gravity initialisation was completed by an IMU sample not newer than the frame
 * (msckf.py:182-183 under the deterministic replay); timestamps[s] < 0 means "no frame for stream s in this
 * step" (sequences of different lengths stepped together): the stream idles and reports published = 0.
 * The first step fixes the message capacity `cap` the device buffers are sized for; rows_cap must be >= 5*cap
 * (camera-pruning update) and >= 1664 (lost-feature update: 1500-row cut + one block), else AV_E_CAPACITY;
 * rows_cap = 0 at create sizes the block buffers from that first `cap` (max(2048, 5*cap + 64) rows).  A later step with a larger
 * cap REBUILDS them (and the device-resident observation store) for the wider message: the batch is drained and the device
 * synchronised, the capacity grows geometrically (x1.5 at least) and the outgrown allocations stay parked until destroy -- pass
 * the largest cap at the first step (BatchedMSCKF(max_features=...)) when the width of the messages varies.  An explicit
 * rows_cap is never grown: a wider message than it can hold fails with AV_E_CAPACITY.
 * rows_cap rows are what each stream owns.  The lost features of one frame may need more (a blank frame drops every track at once:
 * ~4,500 rows at 150 tracks): the device-resident filter then takes the rows from a pool all streams share, allocated behind the
 * last stream's region -- max(131,072, n_streams * rows_cap) rows of ld doubles, AV_MSCKF_POOL_ROWS overrides -- and stops a
 * stream that finds the pool exhausted with AV_E_CAPACITY (av_msckf_batch_stream_status).
 * max_cam_states <= 24 (one back-end pass holds 144 columns = 6 per camera state); the reference's value is 20. */
int  av_msckf_batch_step(av_msckf_batch* b, const int64_t* ids, const double* uv, const int32_t* n_feat, int cap,
                         const double* timestamps, double* out, void* stream);
/* The same step, queued: returns at once; the stream groups of the batch consume their queues independently (a group
 * that is done with frame k starts frame k+1 without waiting for the slowest group, the way the reference's VIO
 * thread runs behind a queue, modules/vio.py:46-58).  All buffers of a submitted step, inputs and `out`, must stay
 * valid until av_msckf_batch_wait has let it retire.  Steps retire in submission order.
 * av_msckf_batch_wait blocks until at most max_pending submitted steps are unfinished (0 = drain) and returns the
 * first error any of them raised (later queued steps are then skipped).  av_msckf_batch_get_cov / _sizes / _push_imu for
 * a frame whose step is already queued must not race with pending steps: drain (wait 0) before reading state. */
int  av_msckf_batch_submit(av_msckf_batch* b, const int64_t* ids, const double* uv, const int32_t* n_feat, int cap,
                           const double* timestamps, double* out, void* stream);
int  av_msckf_batch_wait(av_msckf_batch* b, int max_pending);
/* av_msckf_batch_submit with the feature message in DEVICE arrays (av_frontend_features_dev): the three arrays are consumed by
 * copies enqueued on msg_stream -- the stream that produced them -- before the call returns, so the producer may overwrite them
 * at once; the filter's kernels run on the stream groups' own streams (a batch of one group: on `stream`) behind an event.
 * timestamps and out are host arrays and must stay valid until av_msckf_batch_wait has let the step retire.  Blocks while
 * more than two earlier steps are unfinished.  Needs the device-resident filter state (the default; AV_E_INVALID under
 * AV_MSCKF_STORE=host). */
int  av_msckf_batch_device_resident(const av_msckf_batch* b);     /* 1: state + observation map on the device (default), 0: host bookkeeping */
int  av_msckf_batch_submit_dev(av_msckf_batch* b, const int64_t* ids_dev, const double* uv_dev, const int32_t* n_feat_dev, int cap,
                               const double* timestamps, double* out, void* msg_stream, void* stream);
int  av_msckf_batch_get_cov(av_msckf_batch* b, int stream_idx, double* P_host, int n, void* stream);
int  av_msckf_batch_sizes(av_msckf_batch* b, int stream_idx, int32_t out3[3]);     /* [state dim, camera states, map features] */
/* Full host-side state of one stream -- every target of measurement_update's injection (msckf.py:568-595), for parity tests
 * (SURVEY 8b get_state): imu32 = [imu_state.timestamp, orientation q(4, JPL xyzw), position(3), velocity(3), gyro_bias(3),
 * acc_bias(3), R_imu_cam0 (9, row-major), t_cam0_imu(3), IMUState.gravity(3)]; the camera states of the window in
 * state_server.cam_states order: cam_ids[k] and cam_qp7[7k..] = orientation(4), position(3).  *n_cam = their number;
 * cam_cap = 0 only asks for imu32 and the count.  Drain (wait 0) first. */
int  av_msckf_batch_get_state(av_msckf_batch* b, int stream_idx, double imu32[32], int64_t* cam_ids, double* cam_qp7, int cam_cap, int32_t* n_cam);
/* A failure that concerns ONE stream (its camera window or the block list of one of its updates outgrew a fixed capacity)
 * stops that stream only: from then on it idles and its out[s*12] reads -1; the other streams of the batch keep stepping
 * (the reference's sequences are separate processes, run.bat:4-12).  *status = 0 while the stream runs, else the
 * AV_E_* code that stopped it, with the reason in msg.  Configuration errors (rows_cap vs cap, LDS limits) and HIP errors
 * concern the whole batch and are returned by the step / wait as before. */
int  av_msckf_batch_stream_status(av_msckf_batch* b, int stream_idx, int32_t* status, char* msg, int msg_cap);
/* Run statistics over all streams (drain first): [steps, stream-steps that ran prune_cam_state_buffer (msckf.py:712-786),
 * stream-steps whose lost-feature candidates outgrew rows_cap (device-resident filter: rows taken from the shared overflow pool;
 * AV_MSCKF_STORE=host: gated first, stored in a second pass), device-buffer
 * reallocations after the first step (0 in a correctly pre-sized run), min camera states, max camera states,
 * min map features, max map features].  bench.py uses it to prove that the timed region is the steady state. */
int  av_msckf_batch_counters(av_msckf_batch* b, int64_t out8[8]);
/* Parity-test tap (SURVEY 8b "get_state / get_cov for parity tests"): the values the reference computes inside gating_test and
 * measurement_update (msckf.py:604-612, 548-602) but never returns.  After av_msckf_batch_debug_capture(b, 1) every stream keeps
 * them for the two update phases of its LAST step: phase 0 = remove_lost_features, 1 = prune_cam_state_buffer.  debug_read:
 * gamma[*n_gamma] in the reference's evaluation order (features behind the > 1500-row cut are not evaluated, msckf.py:667-668);
 * *rows = stacked rows of the phase's update (0: no update ran, dx / P_after untouched); dx[*n_state]; P_after[*n_state ** 2]
 * = state_cov right after the update (before the pruning phase removes its two camera states).  P_after rows have pitch
 * *n_state; n_cap = capacity of dx (n_cap doubles) and P_after (n_cap * n_cap).  Synchronous copies per step: tests only. */
int  av_msckf_batch_debug_capture(av_msckf_batch* b, int enable);
int  av_msckf_batch_debug_read(av_msckf_batch* b, int stream_idx, int phase, double* gamma, int gamma_cap, int32_t* n_gamma,
                               double* dx, double* P_after, int n_cap, int32_t* n_state, int32_t* rows);
/* Work done so far, for the filter stage's roofline (drain first): out8 = [algorithmic fp64 flops of the gating tests
 * (per feature with r = 4M-3 rows, n columns: 2rn^2 + 2r^2n + r^3/3; msckf.py:604-612), of the measurement updates on the
 * k = min(m, n) rows kept (S 2kn^2 + 2k^2n, Cholesky k^3/3, solve 2k^2n, (I-KH)P 4kn^2; msckf.py:562-602), of the reference's
 * thin QR by the survey's formula (2mn^2 - 2/3 n^3 when m > n, msckf.py:554-557 -- reported apart: the column-compressed update
 * never runs a QR of that size), features gated, updates run, rows stacked, milliseconds the phase chains (triangulation ..
 * covariance update) spent on the device summed over stream groups, 0].  The time needs enable = 1 on an earlier call (two HIP
 * events per phase and group); enable = 0 switches it off, enable < 0 only reads. */
int  av_msckf_batch_work(av_msckf_batch* b, int enable, double out8[8]);
/* The same two stages counted as the kernels EXECUTE them (drain first; device-resident path, zeros otherwise): out2 = [fp64 flops of
 * the gating tests on the block-sparse shapes -- H_x as 4 x 6 blocks, the gate matrix from the M^2 6 x 6 blocks of P, reflectors applied
 * to it from both sides: ~0.6 MFLOP for a 21-observation track where the dense formula above says 5.5 --, fp64 flops of the updates
 * over the touched columns -- Gram compression m (nc+1)^2 + (nc+1)^3/3 where a stream stacks more rows than one pass, T^T 2 n nc k,
 * S k^2 nc, Cholesky k^3/3, substitution k^2 (n+1), P - Y^T Y n^2 k; information form m nc^2 + 4 nc^3 + 2 n nc^2 + 2 n^2 nc].
 * Analytic per gated feature / per update (upd_stack_kernel), tile padding not counted.  No reference counterpart: bench.py's
 * roofline_msckf.executed. */
int  av_msckf_batch_work_executed(av_msckf_batch* b, double out2[2]);

/* Measurement hooks (bench.py's roofline leg; no reference counterpart): when enabled, every
 * launch group of av_frontend_step is bracketed by a HIP event pair ON THE STEP'S STREAM.
 * max_spans = capacity in event pairs (0 disables).  av_frontend_read_timing synchronises the
 * device and returns, per kernel class [0 pyramid, 1 LK, 2 FAST, 3 glue], the summed elapsed
 * milliseconds and the number of launch groups measured since the last read. */
int av_frontend_enable_timing(av_frontend* fe, int max_spans);
int av_frontend_read_timing(av_frontend* fe, double ms_out[4], int32_t spans_out[4]);

#ifdef __cplusplus
}
#endif
#endif /* AIRVISION_H */
