"""dropin/utils.py (the `utils` module of the drop-in) against vectors produced by the reference's own src/utils.py
(tests/golden/msckf_units.npz, written by tests/golden/make_msckf_golden.py).  The functions are ctypes wrappers over the
library's host helpers av_quat_* (include/airvision.h), so this also pins those; no GPU needed."""
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'uav_airvision_amd', 'dropin'))
import utils as U          # noqa: E402


def test_quaternion_helpers_match_reference_vectors():
    u = np.load(os.path.join(ROOT, 'tests', 'golden', 'msckf_units.npz'))
    tol = dict(rtol=0, atol=1e-14)
    assert np.allclose([U.to_rotation(q) for q in u['u_q']], u['u_R'], **tol)
    assert np.allclose([U.to_quaternion(R) for R in u['u_R']], u['u_q_of_R'], **tol)
    assert np.allclose([U.to_quaternion(R) for R in u['u_Rb']], u['u_q_of_Rb'], **tol)          # every branch of to_quaternion
    assert np.allclose([U.quaternion_multiplication(a, b) for a, b in zip(u['u_q'], u['u_q2'])], u['u_qmul'], **tol)
    assert np.allclose([U.small_angle_quaternion(d) for d in u['u_dth']], u['u_small'], **tol)
    assert np.allclose([U.from_two_vectors(a, b) for a, b in zip(u['u_v0'], u['u_v1'])], u['u_two'], **tol)


def test_skew_conjugate_isometry():
    v, w = np.array([1., 2., 3.]), np.array([-2., .5, 4.])
    assert np.array_equal(U.skew(v) @ w, np.cross(v, w))
    q = U.quaternion_normalize(np.array([.1, -.2, .3, .9]))
    assert np.allclose(U.to_rotation(U.quaternion_conjugate(q)), U.to_rotation(q).T, atol=1e-15)
    A = U.Isometry3d(U.to_rotation(q), v)
    B = U.Isometry3d(U.to_rotation(np.array([.3, .1, -.2, .8])), w)
    assert np.allclose((A * B).matrix(), A.matrix() @ B.matrix(), atol=1e-15)
    assert np.allclose((A * A.inverse()).matrix(), np.eye(4), atol=1e-15)
    assert np.array_equal(U.Isometry3d.from_matrix(A.matrix()).t, v)


def test_bad_shapes_raise():
    import pytest
    with pytest.raises(ValueError):
        U.to_rotation([1.0, 0.0, 0.0])
