"""StereoMatcher (reference: src/image_processing/stereo_matcher.py:6-115): cam0 -> cam1 matching by forward
and backward pyramidal LK on the GPU, then the reference's gates (forward-backward error < 3 px, |dy| < 20,
in bounds, x-term epipolar error with the cam0 model -- SURVEY Appendix A.2-A.5)."""
import math

import numpy as np

from uav_airvision_amd import ops


def _skew(v):
    x, y, z = v
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])


class StereoMatcher(object):
    def __init__(self, lk_params, imu_processor, pyramid_builder, camera_model, stereo_threshold):
        self.lk_params = lk_params
        self.integrate_imu = imu_processor.integrate_imu_data
        self.R_cam0_imu = imu_processor.R_cam0_imu
        self.R_cam1_imu = imu_processor.R_cam1_imu
        self.t_cam0_imu = imu_processor.t_cam0_imu
        self.t_cam1_imu = imu_processor.t_cam1_imu
        self.pyr0 = pyramid_builder.curr_cam0_pyramid
        self.pyr1 = pyramid_builder.curr_cam1_pyramid
        self.camera_model = camera_model
        self.stereo_threshold = stereo_threshold

    def stereo_match(self, cam0_points):
        if len(cam0_points) == 0:
            return np.array([]), np.array([], dtype=bool)
        cm = self.camera_model
        K, D, model = cm.intrinsics, cm.distortion_coeffs, cm.distortion_model
        pts0 = np.array(cam0_points, dtype=np.float32)
        R0to1 = self.R_cam1_imu.T @ self.R_cam0_imu
        und0 = cm.undistort_points(pts0, K, model, D, rectification_matrix=R0to1)
        proj1 = cm.distort_points(und0, K, model, D)
        p1, track_mask, _ = ops.calc_optical_flow_pyr_lk(self.pyr0, self.pyr1, pts0, np.array(proj1, dtype=np.float32), **self.lk_params)
        p0r, _rev, _ = ops.calc_optical_flow_pyr_lk(self.pyr1, self.pyr0, p1, pts0.copy(), **self.lk_params)
        d = pts0 - p0r
        err = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
        disp = np.abs(proj1[:, 1] - p1[:, 1])
        inlier = track_mask.reshape(-1).astype(bool) & (err < 3) & (disp < 20)
        h, w = self.pyr1.shape[:2]
        for i, (x, y) in enumerate(p1):
            if inlier[i] and (x < 0 or x >= w or y < 0 or y >= h):
                inlier[i] = False
        t01 = self.R_cam1_imu.T @ (self.t_cam0_imu - self.t_cam1_imu)
        E = _skew(t01) @ R0to1
        undist0 = cm.undistort_points(pts0, K, model, D)
        undist1 = cm.undistort_points(p1, K, model, D)
        thr = self.stereo_threshold * (4.0 / (2 * K[0] + 2 * K[1]))
        for i in range(len(p1)):
            if not inlier[i]:
                continue
            u0x, u0y, u1x = float(undist0[i, 0]), float(undist0[i, 1]), float(undist1[i, 0])
            l0 = (float(E[0, 0]) * u0x + float(E[0, 1]) * u0y) + float(E[0, 2]) * 1.0
            l1 = (float(E[1, 0]) * u0x + float(E[1, 1]) * u0y) + float(E[1, 2]) * 1.0
            if abs(u1x * l0) / math.sqrt(l0 * l0 + l1 * l1) > thr:
                inlier[i] = False
        return p1, inlier
