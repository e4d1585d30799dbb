"""Configuration bag for the hot path.

Mirrors the attribute surface of the reference's ``ConfigEuRoC`` / ``OptimizationConfigEuRoC``
(reference: src/config.py:7-17, 19-123) so that objects of either class can be handed to
``ImageProcessor(config)`` / ``MSCKF(config)``.  The reference module imports ``cv2`` only for
three integer constants (config.py:41,44); they are spelled out here so no OpenCV is needed.
"""
import numpy as np

# cv2 constants used by config.py:37-44
TERM_CRITERIA_COUNT = 1
TERM_CRITERIA_EPS = 2
OPTFLOW_USE_INITIAL_FLOW = 4


class OptimizationConfigEuRoC(object):
    """LM triangulation knobs (reference: src/config.py:7-17)."""

    def __init__(self):
        self.translation_threshold = -1.0
        self.huber_epsilon = 0.01
        self.estimation_precision = 5e-7
        self.initial_damping = 1e-3
        self.outer_loop_max_iteration = 5
        self.inner_loop_max_iteration = 5


class ConfigEuRoC(object):
    """EuRoC calibration + tunables (reference: src/config.py:19-123)."""

    def __init__(self, grid_row=4, grid_col=5, grid_min_feature_num=3, grid_max_feature_num=5):
        self.optimization_config = OptimizationConfigEuRoC()

        # front-end (config.py:23-35)
        self.grid_row = grid_row
        self.grid_col = grid_col
        self.grid_num = self.grid_row * self.grid_col
        self.grid_min_feature_num = grid_min_feature_num
        self.grid_max_feature_num = grid_max_feature_num
        self.fast_threshold = 15
        self.ransac_threshold = 3      # stored, never read by the reference
        self.stereo_threshold = 5
        self.max_iteration = 30
        self.track_precision = 0.01
        self.pyramid_levels = 3
        self.patch_size = 15
        self.win_size = (self.patch_size, self.patch_size)
        self.lk_params = dict(
            winSize=self.win_size,
            maxLevel=self.pyramid_levels,
            criteria=(TERM_CRITERIA_EPS | TERM_CRITERIA_COUNT, self.max_iteration, self.track_precision),
            flags=OPTFLOW_USE_INITIAL_FLOW)

        # filter (config.py:47-88)
        self.gravity_acc = 9.81
        self.gravity = np.array([0.0, 0.0, -self.gravity_acc])
        self.frame_rate = 20
        self.max_cam_state_size = 20
        self.position_std_threshold = 2.0
        self.rotation_threshold = 0.15
        self.translation_threshold = 0.2
        self.tracking_rate_threshold = 0.5
        self.gyro_noise = 0.005 ** 2
        self.acc_noise = 0.05 ** 2
        self.gyro_bias_noise = 0.001 ** 2
        self.acc_bias_noise = 0.01 ** 2
        self.observation_noise = 0.035 ** 2
        self.velocity = np.zeros(3)
        self.velocity_cov = 0.25
        self.gyro_bias_cov = 0.01
        self.acc_bias_cov = 0.01
        self.extrinsic_rotation_cov = 3.0462e-4
        self.extrinsic_translation_cov = 2.5e-5

        # calibration (config.py:93-123)
        self.T_imu_cam0 = np.array([
            [0.014865542981794, 0.999557249008346, -0.025774436697440, 0.065222909535531],
            [-0.999880929698575, 0.014967213324719, 0.003756188357967, -0.020706385492719],
            [0.004140296794224, 0.025715529947966, 0.999660727177902, -0.008054602460030],
            [0, 0, 0, 1.000000000000000]])
        self.cam0_camera_model = 'pinhole'
        self.cam0_distortion_model = 'radtan'
        self.cam0_distortion_coeffs = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
        self.cam0_intrinsics = np.array([458.654, 457.296, 367.215, 248.375])
        self.cam0_resolution = np.array([752, 480])

        self.T_imu_cam1 = np.array([
            [0.012555267089103, 0.999598781151433, -0.025389800891747, -0.044901980682509],
            [-0.999755099723116, 0.013011905181504, 0.017900583825251, -0.020569771258915],
            [0.018223771455443, 0.025158836311552, 0.999517347077547, -0.008638135126028],
            [0, 0, 0, 1.000000000000000]])
        self.T_cn_cnm1 = np.array([
            [0.999997256477881, 0.002312067192424, 0.000376008102415, -0.110073808127187],
            [-0.002317135723281, 0.999898048506644, 0.014089835846648, 0.000399121547014],
            [-0.000343393120525, -0.014090668452714, 0.999900662637729, -0.000853702503357],
            [0, 0, 0, 1.000000000000000]])
        self.cam1_camera_model = 'pinhole'
        self.cam1_distortion_model = 'radtan'
        self.cam1_distortion_coeffs = np.array([-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05])
        self.cam1_intrinsics = np.array([457.587, 456.134, 379.999, 255.238])
        self.cam1_resolution = np.array([752, 480])

        self.T_imu_body = np.identity(4)
