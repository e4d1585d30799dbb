"""`utils` of the drop-in: the names modules importing the reference's src/utils.py expect, served by the
library's own host helpers (include/airvision.h, av_quat_* -- the same C++ the batched filter runs), so the
JPL-quaternion formulas exist once.  numpy in, numpy out; every wrapper raises on a library error."""
import ctypes as C

import numpy as np

from uav_airvision_amd import _native as N


def _vec(a, n):
    v = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    if v.size != n:
        raise ValueError('expected %d values, got %d' % (n, v.size))
    return v


def _call(name, out_shape, *args):
    out = np.empty(out_shape)
    ptrs = [a.ctypes.data_as(C.c_void_p) for a in args] + [out.ctypes.data_as(C.c_void_p)]
    N.check(getattr(N.lib(), name)(*ptrs))
    return out


def to_rotation(q):
    """C(q), world -> body for a JPL quaternion (reference: utils.py:12-23)."""
    return _call('av_quat_to_rotation', (3, 3), _vec(q, 4))


def to_quaternion(R):
    """Unit JPL quaternion of a rotation matrix (reference: utils.py:25-47)."""
    return _call('av_rotation_to_quat', 4, _vec(R, 9))


def quaternion_multiplication(q1, q2):
    """q1 (x) q2, inputs and output normalised (reference: utils.py:61-76)."""
    return _call('av_quat_multiply', 4, _vec(q1, 4), _vec(q2, 4))


def small_angle_quaternion(dtheta):
    """Quaternion of a small rotation vector (reference: utils.py:79-93)."""
    return _call('av_quat_small_angle', 4, _vec(dtheta, 3))


def from_two_vectors(v0, v1):
    """Quaternion taking direction v0 to v1 (reference: utils.py:96-120)."""
    return _call('av_quat_from_two_vectors', 4, _vec(v0, 3), _vec(v1, 3))


def quaternion_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.sqrt(np.dot(q, q))


def quaternion_conjugate(q):
    return np.asarray(q, dtype=np.float64) * np.array([-1.0, -1.0, -1.0, 1.0])


def skew(vec):
    """[v]x: skew(v) @ w == cross(v, w)."""
    m = np.zeros((3, 3))
    m[[2, 0, 1], [1, 2, 0]] = vec
    m[[1, 2, 0], [2, 0, 1]] = np.negative(vec)
    return m


class Isometry3d(object):
    """Rigid transform x -> R x + t (reference: utils.py:124-141; modules/viewer.py reads `.t`, the filter `.R`)."""
    __slots__ = ('R', 't')

    def __init__(self, R, t):
        self.R, self.t = R, t

    @classmethod
    def from_matrix(cls, T):
        T = np.asarray(T, dtype=np.float64)
        return cls(T[:3, :3].copy(), T[:3, 3].copy())

    def matrix(self):
        return np.block([[np.asarray(self.R, dtype=np.float64), np.reshape(self.t, (3, 1))], [np.zeros((1, 3)), np.ones((1, 1))]])

    def inverse(self):
        Rt = np.transpose(self.R)
        return Isometry3d(Rt, np.negative(np.dot(Rt, self.t)))

    def __mul__(self, other):
        return Isometry3d(np.dot(self.R, other.R), np.dot(self.R, other.t) + self.t)

    def __repr__(self):
        return 'Isometry3d(R=%r, t=%r)' % (self.R, self.t)
