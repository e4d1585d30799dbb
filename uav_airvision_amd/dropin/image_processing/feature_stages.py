"""FeatureInitializer / FeatureTracker / FeatureAdder / FeaturePruner / FeaturePublisher with the reference's
constructor keywords (reference: src/image_processing/feature_initializer.py, feature_tracker.py,
feature_adder.py, feature_pruner.py, feature_publisher.py; wiring in pipeline.py:73-143).

These are the host-list "plugin" flavour of the stages: Python lists of FeatureMetaData as in the reference,
every OpenCV call replaced by a HIP operator (uav_airvision_amd/ops.py).  The throughput path
(ImageProcessingPipeline / FrontendEngine) runs the same semantics device-resident; both are bit-identical to
the CPU oracle (tests/test_gpu_stages.py)."""
from collections import namedtuple
from itertools import chain

import numpy as np

from uav_airvision_amd import ops

from .camera_model import CameraModel, predict_points
from .feature_measurment import FeatureMeasurement
from .feature_meta_data import FeatureMetaData


def _grid_size(img, grid_row, grid_col):
    h, w = img.shape[:2]
    return int(np.ceil(h / grid_row)), int(np.ceil(w / grid_col))


class _FastDetector(object):
    """Stand-in for cv2.FastFeatureDetector_create(threshold) (pipeline.py:23-25): detect(img, mask=None)
    returns keypoints with .pt and .response, in raster order."""
    KeyPoint = namedtuple('KeyPoint', ['pt', 'response'])

    def __init__(self, threshold):
        self.threshold = int(threshold)

    def detect(self, img, mask=None):
        xs, ys, sc = ops.fast_detect(img, self.threshold, mask)
        return [self.KeyPoint((float(x), float(y)), float(s)) for x, y, s in zip(xs, ys, sc)]


def FastFeatureDetector_create(threshold):
    return _FastDetector(threshold)


class FeatureInitializer(object):
    def __init__(self, detector, stereo_matcher, config, cam0_curr_img_msg, curr_features, next_feature_id,
                 grid_row, grid_col, grid_min_feature_num):
        self.detector = detector
        self.stereo_match = stereo_matcher.stereo_match
        self.config = config
        self.cam0_curr_img_msg = cam0_curr_img_msg
        self.curr_features = curr_features
        self.next_feature_id = next_feature_id
        self.grid_row, self.grid_col = grid_row, grid_col
        self.grid_min_feature_num = grid_min_feature_num

    def initialize_first_frame(self):
        """feature_initializer.py:45-85."""
        img = self.cam0_curr_img_msg.image
        gh, gw = _grid_size(img, self.grid_row, self.grid_col)
        kps = self.detector.detect(img)
        cam0_points = [kp.pt for kp in kps]
        cam1_points, inl = self.stereo_match(cam0_points)
        cells = [[] for _ in range(self.config.grid_num)]
        for i, ok in enumerate(inl):
            if not ok:
                continue
            f = FeatureMetaData()
            f.response, f.cam0_point, f.cam1_point = kps[i].response, cam0_points[i], cam1_points[i]
            cells[int(f.cam0_point[1] / gh) * self.grid_col + int(f.cam0_point[0] / gw)].append(f)
        for idx, feats in enumerate(cells):
            for f in sorted(feats, key=lambda q: q.response, reverse=True)[:self.grid_min_feature_num]:
                f.id, f.lifetime = self.next_feature_id, 1
                self.curr_features[idx].append(f)
                self.next_feature_id += 1


class FeatureTracker(object):
    def __init__(self, lk_params, imu_processor, stereo_matcher, cam0_intrinsics, cam0_distortion_model,
                 cam0_distortion_coeffs, cam1_intrinsics, cam1_distortion_model, cam1_distortion_coeffs,
                 prev_cam0_pyramid, curr_cam0_pyramid, prev_features, curr_features, num_features,
                 grid_row, grid_col, ransac_threshold):
        self.lk_params = lk_params
        self.integrate_imu_data = imu_processor.integrate_imu_data
        self.R_cam0_imu, self.R_cam1_imu = imu_processor.R_cam0_imu, imu_processor.R_cam1_imu
        self.stereo_match = stereo_matcher.stereo_match
        self.cam0_intrinsics, self.cam0_dist_model, self.cam0_dist_coeffs = cam0_intrinsics, cam0_distortion_model, cam0_distortion_coeffs
        self.cam1_intrinsics, self.cam1_dist_model, self.cam1_dist_coeffs = cam1_intrinsics, cam1_distortion_model, cam1_distortion_coeffs
        self.prev_cam0_pyramid, self.curr_cam0_pyramid = prev_cam0_pyramid, curr_cam0_pyramid
        self.prev_features, self.curr_features, self.num_features = prev_features, curr_features, num_features
        self.grid_row, self.grid_col = grid_row, grid_col
        self.ransac_threshold = ransac_threshold            # stored, never read (no RANSAC exists)

    def get_grid_size(self, img):
        return _grid_size(img, self.grid_row, self.grid_col)

    def predict_feature_tracking(self, input_pts, R_p_c, intrinsics):
        if len(input_pts) == 0:
            return np.array([], dtype=np.float32)
        return predict_points(input_pts, R_p_c, intrinsics)

    def track_features(self):
        """feature_tracker.py:74-157."""
        img = self.curr_cam0_pyramid
        gh, gw = self.get_grid_size(img)
        cam0_R_p_c, _cam1_R_p_c = self.integrate_imu_data()
        prev = list(chain.from_iterable(self.prev_features))
        self.num_features['before_tracking'] = len(prev)
        if not prev:
            return
        prev_pts = np.array([f.cam0_point for f in prev], dtype=np.float32)
        pred = self.predict_feature_tracking(prev_pts, cam0_R_p_c, self.cam0_intrinsics)
        curr_pts, mask, _ = ops.calc_optical_flow_pyr_lk(self.prev_cam0_pyramid, img, prev_pts, pred, **self.lk_params)
        h, w = img.shape[:2]
        keep = [i for i, p in enumerate(curr_pts)
                if mask[i] and not (p[0] < 0 or p[0] > w - 1 or p[1] < 0 or p[1] > h - 1)]
        self.num_features['after_tracking'] = len(keep)
        tracked = [curr_pts[i] for i in keep]
        cam1_pts, match = self.stereo_match(tracked)
        n = 0
        for k, i in enumerate(keep):
            if not match[k]:
                continue
            f = FeatureMetaData()
            f.id, f.lifetime = prev[i].id, prev[i].lifetime + 1
            f.cam0_point, f.cam1_point = tracked[k], cam1_pts[k]
            self.curr_features[int(f.cam0_point[1] / gh) * self.grid_col + int(f.cam0_point[0] / gw)].append(f)
            n += 1
        self.num_features['after_matching'] = n
        self.num_features['after_ransac'] = n


class FeatureAdder(object):
    def __init__(self, detector, stereo_matcher, config, cam0_curr_img_msg, curr_features, next_feature_id,
                 grid_row, grid_col, grid_max_feature_num, grid_min_feature_num):
        self.detector = detector
        self.stereo_matcher = stereo_matcher
        self.stereo_match = stereo_matcher.stereo_match
        self.config = config
        self.cam0_curr_img_msg = cam0_curr_img_msg
        self.curr_features = curr_features
        self.next_feature_id = next_feature_id
        self.grid_row, self.grid_col = grid_row, grid_col
        self.grid_max_feature_num, self.grid_min_feature_num = grid_max_feature_num, grid_min_feature_num

    def add_new_features(self):
        """feature_adder.py:52-108 (numpy slice semantics of the 7x7 mask kept on purpose)."""
        img = self.cam0_curr_img_msg.image
        gh, gw = _grid_size(img, self.grid_row, self.grid_col)
        mask = np.ones(img.shape[:2], dtype='uint8')
        for f in chain.from_iterable(self.curr_features):
            x, y = int(f.cam0_point[0]), int(f.cam0_point[1])
            mask[y - 3:y + 4, x - 3:x + 4] = 0
        sieve = [[] for _ in range(self.config.grid_num)]
        for kp in self.detector.detect(img, mask=mask):
            sieve[int(kp.pt[1] / gh) * self.grid_col + int(kp.pt[0] / gw)].append(kp)
        cand = []
        for cell in sieve:
            if len(cell) > self.grid_max_feature_num:
                cell = sorted(cell, key=lambda q: q.response, reverse=True)[:self.grid_max_feature_num]
            cand.extend(cell)
        cam0_points = [kp.pt for kp in cand]
        cam1_points, inl = self.stereo_match(cam0_points)
        cells = [[] for _ in range(self.config.grid_num)]
        for i, ok in enumerate(inl):
            if not ok:
                continue
            f = FeatureMetaData()
            f.response, f.cam0_point, f.cam1_point = cand[i].response, cam0_points[i], cam1_points[i]
            cells[int(f.cam0_point[1] / gh) * self.grid_col + int(f.cam0_point[0] / gw)].append(f)
        for idx, feats in enumerate(cells):
            for f in sorted(feats, key=lambda q: q.response, reverse=True)[:self.grid_min_feature_num]:
                f.id, f.lifetime = self.next_feature_id, 1
                self.curr_features[idx].append(f)
                self.next_feature_id += 1


class FeaturePruner(object):
    def __init__(self, grid_max_feature_num):
        self.grid_max_feature_num = grid_max_feature_num
        self.curr_features = None
        self.config = None

    def prune_features(self):
        """feature_pruner.py:8-19 (stable sort by lifetime)."""
        gmax = self.config.grid_max_feature_num if self.config is not None else self.grid_max_feature_num
        for i, feats in enumerate(self.curr_features):
            if len(feats) > gmax:
                self.curr_features[i] = sorted(feats, key=lambda q: q.lifetime, reverse=True)[:gmax]


class FeaturePublisher(object):
    def __init__(self, cam0_intrinsics, cam0_dist_model, cam0_dist_coeffs, cam1_intrinsics, cam1_dist_model, cam1_dist_coeffs):
        self.cam0_intrinsics, self.cam0_dist_model, self.cam0_dist_coeffs = cam0_intrinsics, cam0_dist_model, cam0_dist_coeffs
        self.cam1_intrinsics, self.cam1_dist_model, self.cam1_dist_coeffs = cam1_intrinsics, cam1_dist_model, cam1_dist_coeffs
        self.cam0_curr_img_msg = self.cam1_curr_img_msg = self.curr_features = None
        self._cm = CameraModel(cam0_intrinsics, cam0_dist_model, cam0_dist_coeffs)

    def undistort_points(self, pts_in, intrinsics, distortion_model, distortion_coeffs,
                         rectification_matrix=np.identity(3), new_intrinsics=np.array([1, 1, 0, 0])):
        return self._cm.undistort_points(pts_in, intrinsics, distortion_model, distortion_coeffs, rectification_matrix, new_intrinsics)

    def distort_points(self, pts_in, intrinsics, distortion_model, distortion_coeffs):
        return self._cm.distort_points(pts_in, intrinsics, distortion_model, distortion_coeffs)

    def publish(self):
        """feature_publisher.py:90-121; np.reshape of the mixed point list reproduces the dtype rule (A.19)."""
        feats = list(chain.from_iterable(self.curr_features))
        u0 = self.undistort_points([f.cam0_point for f in feats], self.cam0_intrinsics, self.cam0_dist_model, self.cam0_dist_coeffs)
        u1 = self.undistort_points([f.cam1_point for f in feats], self.cam1_intrinsics, self.cam1_dist_model, self.cam1_dist_coeffs)
        out = []
        for i, f in enumerate(feats):
            m = FeatureMeasurement()
            m.id, m.u0, m.v0, m.u1, m.v1 = f.id, u0[i][0], u0[i][1], u1[i][0], u1[i][1]
            out.append(m)
        return namedtuple('feature_msg', ['timestamp', 'features'])(self.cam0_curr_img_msg.timestamp, out)
