"""IMUProcessor (reference: src/image_processing/imu_processor.py:4-67): IMU buffer and the mean-gyro
rotation prediction between two frames.  Host-side scalar code (negligible cost)."""
import math

import numpy as np


def rodrigues(rvec):
    """cv2.Rodrigues(vector)[0] (imu_processor.py:63-64)."""
    r = [float(v) for v in rvec]
    theta = math.sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2])
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = math.cos(theta), math.sin(theta)
    x, y, z = [v * (1. / theta) for v in r]
    rrt = np.array([[x * x, x * y, x * z], [x * y, y * y, y * z], [x * z, y * z, z * z]])
    rx = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return c * np.eye(3) + (1. - c) * rrt + s * rx


def _mtv(R, v):
    return np.array([(float(R[0, i]) * float(v[0]) + float(R[1, i]) * float(v[1])) + float(R[2, i]) * float(v[2]) for i in range(3)])


class IMUProcessor(object):
    def __init__(self, T_imu_cam0, T_imu_cam1):
        self.T_cam0_imu = np.linalg.inv(T_imu_cam0)
        self.R_cam0_imu = self.T_cam0_imu[:3, :3]
        self.t_cam0_imu = self.T_cam0_imu[:3, 3]
        self.T_cam1_imu = np.linalg.inv(T_imu_cam1)
        self.R_cam1_imu = self.T_cam1_imu[:3, :3]
        self.t_cam1_imu = self.T_cam1_imu[:3, 3]
        self.imu_buffer = []
        self.cam0_prev_img_msg = None
        self.cam0_curr_img_msg = None

    def imu_callback(self, msg):
        self.imu_buffer.append(msg)

    def integrate_imu_data(self):
        """imu_processor.py:28-67."""
        t_prev, t_curr = self.cam0_prev_img_msg.timestamp, self.cam0_curr_img_msg.timestamp
        idx_begin = idx_end = None
        for i, m in enumerate(self.imu_buffer):
            if m.timestamp >= t_prev - 0.01:
                idx_begin = i
                break
        for i, m in enumerate(self.imu_buffer):
            if m.timestamp >= t_curr - 0.004:
                idx_end = i
                break
        if idx_begin is None or idx_end is None:
            return np.identity(3), np.identity(3)
        mean = np.zeros(3)
        for m in self.imu_buffer[idx_begin:idx_end]:
            mean += m.angular_velocity
        if idx_end - idx_begin > 0:
            mean /= (idx_end - idx_begin)
        dt = t_curr - t_prev
        R0 = rodrigues(_mtv(self.R_cam0_imu, mean) * dt).T
        R1 = rodrigues(_mtv(self.R_cam1_imu, mean) * dt).T
        self.imu_buffer = self.imu_buffer[idx_end:]
        return R0, R1
