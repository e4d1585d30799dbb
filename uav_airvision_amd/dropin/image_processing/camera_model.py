"""CameraModel (reference: src/image_processing/camera_model.py:4-108): radtan undistort / distort of point
lists on the GPU (av_undistort_points / av_distort_points replace cv2.undistortPoints / cv2.projectPoints)."""
import numpy as np

from uav_airvision_amd import ops


class CameraModel(object):
    def __init__(self, intrinsics, distortion_model, distortion_coeffs):
        self.intrinsics = intrinsics
        self.distortion_model = distortion_model
        self.distortion_coeffs = distortion_coeffs
        fx, fy, cx, cy = intrinsics
        self.K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=float)

    @staticmethod
    def _check(distortion_model, new_intrinsics=None):
        if new_intrinsics is not None and list(new_intrinsics) != [1, 1, 0, 0]:
            raise NotImplementedError('new_intrinsics other than [1,1,0,0] are not used by the reference')

    def undistort_points(self, pts_in, intrinsics, distortion_model, distortion_coeffs,
                         rectification_matrix=np.identity(3), new_intrinsics=np.array([1, 1, 0, 0])):
        """camera_model.py:24-47."""
        if len(pts_in) == 0:
            return []
        self._check(distortion_model, new_intrinsics)
        return ops.undistort_points(np.reshape(pts_in, (-1, 1, 2)), intrinsics, distortion_coeffs, rectification_matrix,
                                    distortion_model=distortion_model)       # 'equidistant' -> cv2.fisheye.undistortPoints, else radtan (:41-46)

    def distort_points(self, pts_in, intrinsics, distortion_model, distortion_coeffs):
        """camera_model.py:49-75."""
        if len(pts_in) == 0:
            return []
        self._check(distortion_model)
        return ops.distort_points(pts_in, intrinsics, distortion_coeffs, distortion_model=distortion_model)

    def predict_feature_tracking(self, input_pts, R_p_c, intrinsics):
        """camera_model.py:77-93 (unused by the pipeline; kept for API compatibility)."""
        if len(input_pts) == 0:
            return []
        return predict_points(input_pts, R_p_c, intrinsics)


def homography(R_p_c, intrinsics):
    """K R K^-1 with the analytic inverse of K, evaluated k-sequentially."""
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    K = [[fx, 0., cx], [0., fy, cy], [0., 0., 1.]]
    Ki = [[1. / fx, 0., -cx / fx], [0., 1. / fy, -cy / fy], [0., 0., 1.]]

    def mm(A, B):
        return [[(float(A[i][0]) * float(B[0][j]) + float(A[i][1]) * float(B[1][j])) + float(A[i][2]) * float(B[2][j])
                 for j in range(3)] for i in range(3)]
    return mm(mm(K, np.asarray(R_p_c).tolist()), Ki)


def predict_points(input_pts, R_p_c, intrinsics):
    """feature_tracker.py:159-177: p' ~ K R K^-1 [x y 1], fp64 math, float32 store."""
    H = homography(R_p_c, intrinsics)
    out = np.empty((len(input_pts), 2), np.float32)
    for i, p in enumerate(input_pts):
        x, y = float(p[0]), float(p[1])
        h = [(H[r][0] * x + H[r][1] * y) + H[r][2] * 1.0 for r in range(3)]
        out[i, 0] = h[0] / h[2]
        out[i, 1] = h[1] / h[2]
    return out
