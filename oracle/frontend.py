"""oracle/frontend.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference image front-end (ImageProcessingPipeline and its feature_*
stages), one function per reference stage, on top of oracle/cvops.py.  It reproduces the
reference's glue semantics including the quirks listed in SURVEY.md Appendix A; every function
cites the reference lines it follows.  Parity status: glue = restated line by line from the
reference sources; cv2 calls = "parity unpinned" (oracle/imgops.c header).

Tiny fp64 3x3 products (homography, epipolar line) are written out term by term in a fixed order
(sum over k = 0,1,2, no FMA) where the reference calls numpy's BLAS-backed `@`: BLAS is free to
fuse or reorder those three terms, so the last bit of the reference is not defined anyway, and a
fixed order is what lets the HIP path be compared bit for bit.
"""
import math
from collections import namedtuple

import numpy as np

from . import cvops

feature_msg_t = namedtuple('feature_msg', ['timestamp', 'features'])


class Feat(object):
    """FeatureMetaData (reference: src/image_processing/feature_meta_data.py:1-10)."""
    __slots__ = ('id', 'response', 'lifetime', 'cam0_point', 'cam1_point')

    def __init__(self):
        self.id = None
        self.response = None
        self.lifetime = None
        self.cam0_point = None
        self.cam1_point = None


class Meas(object):
    """FeatureMeasurement (reference: src/image_processing/feature_measurment.py:1-9)."""
    __slots__ = ('id', 'u0', 'v0', 'u1', 'v1')


def matmul3(A, B):
    """3x3 @ 3x3 in fp64, k-sequential."""
    out = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            out[i, j] = (float(A[i, 0]) * float(B[0, j]) + float(A[i, 1]) * float(B[1, j])) + float(A[i, 2]) * float(B[2, j])
    return out


def matvec3(A, v):
    return np.array([(float(A[i, 0]) * float(v[0]) + float(A[i, 1]) * float(v[1])) + float(A[i, 2]) * float(v[2])
                     for i in range(3)])


def skew(v):
    x, y, z = v
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])


def tracking_homography(R_p_c, intr):
    """H = K R K^-1 (feature_tracker.py:166-171) with the analytic inverse of K."""
    fx, fy, cx, cy = [float(v) for v in intr]
    K = np.array([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]])
    Kinv = np.array([[1. / fx, 0., -cx / fx], [0., 1. / fy, -cy / fy], [0., 0., 1.]])
    return matmul3(matmul3(K, R_p_c), Kinv)


def predict_feature_tracking(pts, H):
    """feature_tracker.py:173-177: p' ~ H [x y 1], fp64 math, float32 store."""
    out = np.empty((len(pts), 2), np.float32)
    for i, p in enumerate(pts):
        h = matvec3(H, (float(p[0]), float(p[1]), 1.0))
        out[i, 0] = h[0] / h[2]
        out[i, 1] = h[1] / h[2]
    return out


class StereoGeometry(object):
    """Extrinsics-derived constants of stereo_match (imu_processor.py:10-17,
    stereo_matcher.py:47,90-91,103-104).  Computed with numpy exactly as the reference does."""

    def __init__(self, config):
        T_cam0_imu = np.linalg.inv(config.T_imu_cam0)
        T_cam1_imu = np.linalg.inv(config.T_imu_cam1)
        self.R_cam0_imu = T_cam0_imu[:3, :3]
        self.t_cam0_imu = T_cam0_imu[:3, 3]
        self.R_cam1_imu = T_cam1_imu[:3, :3]
        self.t_cam1_imu = T_cam1_imu[:3, 3]
        self.R0to1 = self.R_cam1_imu.T @ self.R_cam0_imu
        t01 = self.R_cam1_imu.T @ (self.t_cam0_imu - self.t_cam1_imu)
        self.E = skew(t01) @ self.R0to1
        k = config.cam0_intrinsics
        self.norm_unit = 4.0 / (2 * k[0] + 2 * k[1])


def stereo_match(img0, img1, cam0_points, config, geom, cache_pyramids=False):
    """StereoMatcher.stereo_match (stereo_matcher.py:33-115).  Uses the cam0 model for both
    cameras (Appendix A.3), ignores the backward LK status (A.4), x-term-only epipolar error (A.2).
    Returns (p1 float32[N,2], inlier bool[N], dbg dict)."""
    if len(cam0_points) == 0:
        return np.array([]), np.array([], dtype=bool), {}
    K0, D0 = config.cam0_intrinsics, config.cam0_distortion_coeffs
    pts0 = np.array(cam0_points, dtype=np.float32)
    M0 = getattr(config, 'cam0_distortion_model', 'radtan')
    und0 = cvops.undistort_points(pts0, K0, D0, geom.R0to1, distortion_model=M0)
    proj1 = cvops.distort_points(und0, K0, D0, distortion_model=M0)
    lk = dict(config.lk_params)
    p1, track_mask, _ = cvops.calc_optical_flow_pyr_lk(img0, img1, pts0, np.array(proj1, dtype=np.float32),
                                                       cache_pyramids=cache_pyramids, **lk)
    p0r, _rev, _ = cvops.calc_optical_flow_pyr_lk(img1, img0, p1, pts0.copy(), cache_pyramids=cache_pyramids, **lk)
    d = pts0 - p0r
    err = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])            # float32, as np.linalg.norm(axis=1)
    disp = np.abs(proj1[:, 1] - p1[:, 1])
    inlier = track_mask.reshape(-1).astype(bool) & (err < 3) & (disp < 20)
    h, w = img1.shape[:2]
    for i in range(len(p1)):
        if inlier[i]:
            x, y = p1[i]
            if x < 0 or x >= w or y < 0 or y >= h:
                inlier[i] = False
    undist0 = cvops.undistort_points(pts0, K0, D0, distortion_model=M0)
    undist1 = cvops.undistort_points(p1, K0, D0, distortion_model=M0)
    thr = config.stereo_threshold * geom.norm_unit
    E = geom.E
    for i in range(len(p1)):
        if not inlier[i]:
            continue
        line = matvec3(E, (float(undist0[i, 0]), float(undist0[i, 1]), 1.0))
        err_epi = abs(float(undist1[i, 0]) * line[0]) / math.sqrt(line[0] * line[0] + line[1] * line[1])
        if err_epi > thr:
            inlier[i] = False
    return p1, inlier, dict(proj1=proj1, p0r=p0r, track_mask=track_mask)


def integrate_imu(imu_buffer, t_prev, t_curr, geom):
    """IMUProcessor.integrate_imu_data (imu_processor.py:28-67).  Returns
    (cam0_R_p_c, cam1_R_p_c, trimmed_buffer)."""
    idx_begin = idx_end = None
    for i, m in enumerate(imu_buffer):
        if m.timestamp >= t_prev - 0.01:
            idx_begin = i
            break
    for i, m in enumerate(imu_buffer):
        if m.timestamp >= t_curr - 0.004:
            idx_end = i
            break
    if idx_begin is None or idx_end is None:
        return np.identity(3), np.identity(3), imu_buffer
    mean = np.zeros(3)
    for m in imu_buffer[idx_begin:idx_end]:
        mean += m.angular_velocity
    count = idx_end - idx_begin
    if count > 0:
        mean /= count
    cam0_mean = matvec3(geom.R_cam0_imu.T, mean)
    cam1_mean = matvec3(geom.R_cam1_imu.T, mean)
    dt = t_curr - t_prev
    R0 = cvops.rodrigues(cam0_mean * dt).T
    R1 = cvops.rodrigues(cam1_mean * dt).T
    return R0, R1, imu_buffer[idx_end:]


def grid_size(img, config):
    """get_grid_size (feature_tracker.py:65-72)."""
    h, w = img.shape[:2]
    return int(np.ceil(h / config.grid_row)), int(np.ceil(w / config.grid_col))


def cell_of(pt, gh, gw, config):
    """feature_tracker.py:144-146 / feature_adder.py:68-70 / feature_initializer.py:69-71."""
    return int(pt[1] / gh) * config.grid_col + int(pt[0] / gw)


class OracleFrontend(object):
    """ImageProcessingPipeline (pipeline.py:14-150) restated; same callback surface."""

    def __init__(self, config, cache_pyramids=True):
        self.config = config
        self.geom = StereoGeometry(config)
        self.cache_pyramids = cache_pyramids
        self.imu_buffer = []
        self.prev_cam0_msg = None
        self.prev_img0 = None
        self.next_feature_id = 0
        self.prev_features = [[] for _ in range(config.grid_num)]
        self.curr_features = [[] for _ in range(config.grid_num)]
        self.first_frame = True
        self.num_features = {}
        self.debug = {}

    def imu_callback(self, msg):
        self.imu_buffer.append(msg)            # imu_processor.py:22-26

    # ---- stages --------------------------------------------------------------------------
    def _stereo(self, img0, img1, pts):
        return stereo_match(img0, img1, pts, self.config, self.geom, self.cache_pyramids)

    def _initialize_first_frame(self, img0, img1):
        """FeatureInitializer.initialize_first_frame (feature_initializer.py:45-85)."""
        cfg = self.config
        gh, gw = grid_size(img0, cfg)
        xs, ys, sc = cvops.fast_detect(img0, cfg.fast_threshold)
        cam0_points = [(float(x), float(y)) for x, y in zip(xs, ys)]
        cam1_points, inl, _ = self._stereo(img0, img1, cam0_points)
        cells = [[] for _ in range(cfg.grid_num)]
        for i, ok in enumerate(inl):
            if not ok:
                continue
            f = Feat()
            f.response = float(sc[i]); f.cam0_point = cam0_points[i]; f.cam1_point = cam1_points[i]
            cells[cell_of(f.cam0_point, gh, gw, cfg)].append(f)
        for idx, feats in enumerate(cells):
            for f in sorted(feats, key=lambda q: q.response, reverse=True)[:cfg.grid_min_feature_num]:
                f.id = self.next_feature_id
                f.lifetime = 1
                self.curr_features[idx].append(f)
                self.next_feature_id += 1

    def _track(self, prev_img0, img0, img1, t_prev, t_curr):
        """FeatureTracker.track_features (feature_tracker.py:74-157)."""
        cfg = self.config
        gh, gw = grid_size(img0, cfg)
        R0, _R1, self.imu_buffer = integrate_imu(self.imu_buffer, t_prev, t_curr, self.geom)
        prev = [f for cell in self.prev_features for f in cell]
        self.num_features['before_tracking'] = len(prev)
        if not prev:
            return
        prev_pts = np.array([f.cam0_point for f in prev], dtype=np.float32)
        H = tracking_homography(R0, cfg.cam0_intrinsics)
        pred = predict_feature_tracking(prev_pts, H)
        curr_pts, mask, _ = cvops.calc_optical_flow_pyr_lk(prev_img0, img0, prev_pts, pred,
                                                           cache_pyramids=self.cache_pyramids, **cfg.lk_params)
        h, w = img0.shape[:2]
        keep = []
        for i, p in enumerate(curr_pts):
            if not mask[i]:
                continue
            if p[0] < 0 or p[0] > w - 1 or p[1] < 0 or p[1] > h - 1:
                continue
            keep.append(i)
        self.num_features['after_tracking'] = len(keep)
        tracked = [curr_pts[i] for i in keep]
        cam1_pts, match, _ = self._stereo(img0, img1, tracked)
        n = 0
        for k, i in enumerate(keep):
            if not match[k]:
                continue
            f = Feat()
            f.id = prev[i].id
            f.lifetime = prev[i].lifetime + 1
            f.cam0_point = tracked[k]
            f.cam1_point = cam1_pts[k]
            self.curr_features[cell_of(f.cam0_point, gh, gw, cfg)].append(f)
            n += 1
        self.num_features['after_matching'] = n
        self.num_features['after_ransac'] = n          # no RANSAC exists
        self.debug['track'] = dict(prev_pts=prev_pts, pred=pred, curr_pts=curr_pts, mask=mask, H=H)

    def _add_new(self, img0, img1):
        """FeatureAdder.add_new_features (feature_adder.py:52-108)."""
        cfg = self.config
        gh, gw = grid_size(img0, cfg)
        mask = np.ones(img0.shape[:2], dtype='uint8')
        for cell in self.curr_features:
            for f in cell:
                x, y = int(f.cam0_point[0]), int(f.cam0_point[1])
                mask[y - 3:y + 4, x - 3:x + 4] = 0       # numpy slice semantics on purpose (Appendix A.6)
        xs, ys, sc = cvops.fast_detect(img0, cfg.fast_threshold, mask)
        sieve = [[] for _ in range(cfg.grid_num)]
        for x, y, s in zip(xs, ys, sc):
            pt = (float(x), float(y))
            sieve[cell_of(pt, gh, gw, cfg)].append((pt, float(s)))
        cand = []
        for cell in sieve:
            if len(cell) > cfg.grid_max_feature_num:
                cell = sorted(cell, key=lambda q: q[1], reverse=True)[:cfg.grid_max_feature_num]
            cand.extend(cell)
        cam0_points = [c[0] for c in cand]
        cam1_points, inl, _ = self._stereo(img0, img1, cam0_points)
        cells = [[] for _ in range(cfg.grid_num)]
        for i, ok in enumerate(inl):
            if not ok:
                continue
            f = Feat()
            f.response = cand[i][1]; f.cam0_point = cam0_points[i]; f.cam1_point = cam1_points[i]
            cells[cell_of(f.cam0_point, gh, gw, cfg)].append(f)
        n_new = 0
        for idx, feats in enumerate(cells):
            for f in sorted(feats, key=lambda q: q.response, reverse=True)[:cfg.grid_min_feature_num]:
                f.id = self.next_feature_id
                f.lifetime = 1
                self.curr_features[idx].append(f)
                self.next_feature_id += 1
                n_new += 1
        self.debug['add'] = dict(n_candidates=len(cand), n_new=n_new, n_fast=len(xs))

    def _prune(self):
        """FeaturePruner.prune_features (feature_pruner.py:8-19)."""
        gmax = self.config.grid_max_feature_num
        for i, feats in enumerate(self.curr_features):
            if len(feats) > gmax:
                self.curr_features[i] = sorted(feats, key=lambda q: q.lifetime, reverse=True)[:gmax]

    def _publish(self, timestamp):
        """FeaturePublisher.publish (feature_publisher.py:90-121) including the dtype rule of
        Appendix A.19: cam0 points are float64 whenever any of them is a FAST tuple."""
        cfg = self.config
        feats = [f for cell in self.curr_features for f in cell]
        ids = [f.id for f in feats]
        p0 = [f.cam0_point for f in feats]
        p1 = [f.cam1_point for f in feats]
        if feats:
            u0 = cvops.undistort_points(np.reshape(p0, (-1, 1, 2)), cfg.cam0_intrinsics, cfg.cam0_distortion_coeffs,
                                        distortion_model=getattr(cfg, 'cam0_distortion_model', 'radtan'))
            u1 = cvops.undistort_points(np.reshape(p1, (-1, 1, 2)), cfg.cam1_intrinsics, cfg.cam1_distortion_coeffs,
                                        distortion_model=getattr(cfg, 'cam1_distortion_model', 'radtan'))
        out = []
        for i in range(len(ids)):
            m = Meas()
            m.id = ids[i]
            m.u0 = u0[i][0]; m.v0 = u0[i][1]; m.u1 = u1[i][0]; m.v1 = u1[i][1]
            out.append(m)
        return feature_msg_t(timestamp, out)

    # ---- driver --------------------------------------------------------------------------
    def stereo_callback(self, stereo_msg):
        """pipeline.py:46-150."""
        cam0_msg, cam1_msg = stereo_msg.cam0_msg, stereo_msg.cam1_msg
        img0, img1 = cam0_msg.image, cam1_msg.image
        if self.first_frame:
            self._initialize_first_frame(img0, img1)
            self.first_frame = False
        else:
            self._track(self.prev_img0, img0, img1, self.prev_cam0_msg.timestamp, cam0_msg.timestamp)
            self._add_new(img0, img1)
            self._prune()
        msg = self._publish(cam0_msg.timestamp)
        self.prev_cam0_msg = cam0_msg
        self.prev_img0 = img0
        self.prev_features = self.curr_features
        self.curr_features = [[] for _ in range(self.config.grid_num)]
        return msg
