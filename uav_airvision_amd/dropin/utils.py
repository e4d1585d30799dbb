"""Quaternion / SE3 helpers with the reference's names and conventions (reference: src/utils.py:2-141;
JPL quaternions [x, y, z, w], `to_rotation(q)` maps world -> body).  Host-side scalar math only;
everything matrix-sized runs in libairvision_hip.so."""
import numpy as np


def skew(vec):
    x, y, z = vec
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])


def to_rotation(q):
    """utils.py:12-23 (re-normalises q on every call)."""
    q = q / np.linalg.norm(q)
    vec, w = q[:3], q[3]
    return (2 * w * w - 1) * np.identity(3) - 2 * w * skew(vec) + 2 * vec[:, None] * vec


def to_quaternion(R):
    """utils.py:25-47."""
    if R[2, 2] < 0:
        if R[0, 0] > R[1, 1]:
            q = [1 + R[0, 0] - R[1, 1] - R[2, 2], R[0, 1] + R[1, 0], R[2, 0] + R[0, 2], R[1, 2] - R[2, 1]]
        else:
            q = [R[0, 1] + R[1, 0], 1 - R[0, 0] + R[1, 1] - R[2, 2], R[2, 1] + R[1, 2], R[2, 0] - R[0, 2]]
    elif R[0, 0] < -R[1, 1]:
        q = [R[0, 2] + R[2, 0], R[2, 1] + R[1, 2], 1 - R[0, 0] - R[1, 1] + R[2, 2], R[0, 1] - R[1, 0]]
    else:
        q = [R[1, 2] - R[2, 1], R[2, 0] - R[0, 2], R[0, 1] - R[1, 0], 1 + R[0, 0] + R[1, 1] + R[2, 2]]
    q = np.array(q)
    return q / np.linalg.norm(q)


def quaternion_normalize(q):
    return q / np.linalg.norm(q)


def quaternion_conjugate(q):
    return np.array([*-q[:3], q[3]])


def quaternion_multiplication(q1, q2):
    """utils.py:61-76."""
    q1 = q1 / np.linalg.norm(q1)
    q2 = q2 / np.linalg.norm(q2)
    L = np.array([[q1[3], q1[2], -q1[1], q1[0]],
                  [-q1[2], q1[3], q1[0], q1[1]],
                  [q1[1], -q1[0], q1[3], q1[2]],
                  [-q1[0], -q1[1], -q1[2], q1[3]]])
    q = L @ q2
    return q / np.linalg.norm(q)


def small_angle_quaternion(dtheta):
    """utils.py:79-93."""
    dq = dtheta / 2.
    n2 = dq @ dq
    if n2 <= 1:
        return np.array([*dq, np.sqrt(1 - n2)])
    q = np.array([*dq, 1.])
    return q / np.sqrt(1 + n2)


def from_two_vectors(v0, v1):
    """utils.py:96-120."""
    v0 = v0 / np.linalg.norm(v0)
    v1 = v1 / np.linalg.norm(v1)
    d = v0 @ v1
    if d < -0.999999:
        axis = np.cross([1, 0, 0], v0)
        if np.linalg.norm(axis) < 0.000001:
            axis = np.cross([0, 1, 0], v0)
        q = np.array([*axis, 0.])
    elif d > 0.999999:
        q = np.array([0., 0., 0., 1.])
    else:
        s = np.sqrt((1 + d) * 2)
        q = np.array([*(np.cross(v0, v1) / s), 0.5 * s])
    q = q / np.linalg.norm(q)
    return quaternion_conjugate(q)


class Isometry3d(object):
    """utils.py:124-141."""

    def __init__(self, R, t):
        self.R = R
        self.t = t

    def matrix(self):
        m = np.identity(4)
        m[:3, :3] = self.R
        m[:3, 3] = self.t
        return m

    def inverse(self):
        return Isometry3d(self.R.T, -self.R.T @ self.t)

    def __mul__(self, T1):
        return Isometry3d(self.R @ T1.R, self.R @ T1.t + self.t)
