"""oracle/cvops.py -- TEST INFRASTRUCTURE ONLY (ctypes front of oracle/imgops.c).

numpy-level stand-ins for the cv2 calls the reference front-end makes; argument meaning follows the
reference call sites (cited per function).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  Parity status: "parity unpinned" (see imgops.c header).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')


def build(force=False):
    """Compile oracle/imgops.c with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, 'imgops.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_fast9_detect.restype = C.c_int
    return _lib


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def pyr_down(img):
    """cv2.pyrDown on uint8[h,w] (what calcOpticalFlowPyrLK runs internally per level)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down_u8(_u8p(img), C.c_int(w), C.c_int(h), _u8p(out))
    return out


def build_pyramid(img, max_level):
    """Levels 0..max_level (buildOpticalFlowPyramid without derivatives)."""
    pyr = [np.ascontiguousarray(img, dtype=np.uint8)]
    for _ in range(max_level):
        pyr.append(pyr_down(pyr[-1]))
    return pyr


_pyr_cache = {}


def _cached_pyramid(img, max_level, cache):
    if not cache:
        return build_pyramid(img, max_level)
    key = (id(img), max_level)
    hit = _pyr_cache.get(key)
    if hit is not None and hit[0] is img:
        return hit[1]
    pyr = build_pyramid(img, max_level)
    if len(_pyr_cache) > 8:
        _pyr_cache.clear()
    _pyr_cache[key] = (img, pyr)
    return pyr


def calc_optical_flow_pyr_lk(prev_img, next_img, prev_pts, next_pts, winSize=(15, 15), maxLevel=3,
                             criteria=(3, 30, 0.01), flags=4, minEigThreshold=1e-4, cache_pyramids=False):
    """cv2.calcOpticalFlowPyrLK as called at feature_tracker.py:102-108 and
    stereo_matcher.py:64-74 (always OPTFLOW_USE_INITIAL_FLOW, config.py:44).
    Returns (next_pts float32[N,2], status uint8[N,1], None)."""
    assert flags & 4, 'the reference always passes OPTFLOW_USE_INITIAL_FLOW'
    assert winSize[0] == winSize[1]
    ctype, max_iter, eps = criteria
    max_iter = min(max(int(max_iter), 0), 100) if (ctype & 1) else 30
    eps = min(max(float(eps), 0.), 10.) if (ctype & 2) else 0.01
    prev = np.ascontiguousarray(np.asarray(prev_pts, dtype=np.float32).reshape(-1, 2))
    nxt = np.array(np.asarray(next_pts, dtype=np.float32).reshape(-1, 2), copy=True, order='C')
    n = prev.shape[0]
    assert nxt.shape[0] == n
    status = np.zeros((n, 1), np.uint8)
    if n == 0:
        return nxt, status, None
    pI = _cached_pyramid(prev_img, maxLevel, cache_pyramids)
    pJ = _cached_pyramid(next_img, maxLevel, cache_pyramids)
    # cv::buildOpticalFlowPyramid (OpenCV 4.x lkpyramid.cpp) returns the last level whose successor is still larger than the
    # window in both dimensions; level 0 is always used.  Not reached at 752 x 480, winSize 15, maxLevel 3.
    nlev = 1
    while nlev <= maxLevel and pI[nlev].shape[1] > winSize[0] and pI[nlev].shape[0] > winSize[1]:
        nlev += 1
    pI, pJ = pI[:nlev], pJ[:nlev]
    PI = (C.POINTER(C.c_uint8) * nlev)(*[_u8p(a) for a in pI])
    PJ = (C.POINTER(C.c_uint8) * nlev)(*[_u8p(a) for a in pJ])
    W = (C.c_int * nlev)(*[a.shape[1] for a in pI])
    H = (C.c_int * nlev)(*[a.shape[0] for a in pI])
    lib().orc_lk_track(C.c_int(nlev), PI, PJ, W, H,
                       prev.ctypes.data_as(C.POINTER(C.c_float)), nxt.ctypes.data_as(C.POINTER(C.c_float)),
                       _u8p(status), C.c_int(n), C.c_int(winSize[0]), C.c_int(max_iter),
                       C.c_double(eps), C.c_double(minEigThreshold))
    return nxt, status, None


def fast_detect(img, threshold, mask=None, cap=1 << 17):
    """cv2.FastFeatureDetector_create(threshold).detect(img, mask) (pipeline.py:23-25,
    feature_initializer.py:52, feature_adder.py:64).  Returns int arrays (x, y, response) in
    raster order; kp.pt == (float(x), float(y)), kp.response == float(score)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    xs = np.empty(cap, np.int32); ys = np.empty(cap, np.int32); sc = np.empty(cap, np.int32)
    mp = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        mp = _u8p(mask)
    ip = C.POINTER(C.c_int)
    n = lib().orc_fast9_detect(_u8p(img), C.c_int(w), C.c_int(h), C.c_int(int(threshold)), mp, C.c_int(cap),
                               xs.ctypes.data_as(ip), ys.ctypes.data_as(ip), sc.ctypes.data_as(ip))
    if n > cap:
        raise RuntimeError('oracle FAST capacity exceeded: %d > %d' % (n, cap))
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def undistort_points(pts_in, intrinsics, distortion_coeffs, rectification_matrix=None, distortion_model='radtan'):
    """cv2.undistortPoints(pts, K, D, None, R, P=identity) as wrapped by camera_model.py:24-47 and
    feature_publisher.py:24-59 (new_intrinsics = [1,1,0,0]); distortion_model 'equidistant' =
    cv2.fisheye.undistortPoints(pts, K, D, R, P=identity) (camera_model.py:41-43; parity unpinned, see imgops.c).  Output dtype
    follows the input array dtype the way OpenCV does: float32 in -> float32 out, anything else -> float64."""
    arr = np.asarray(pts_in)
    out_f32 = arr.dtype == np.float32
    pts = np.ascontiguousarray(arr.reshape(-1, 2), dtype=np.float64)
    R = np.ascontiguousarray(np.eye(3) if rectification_matrix is None else rectification_matrix, dtype=np.float64)
    kd = np.ascontiguousarray(intrinsics, dtype=np.float64)
    dd = np.ascontiguousarray(distortion_coeffs, dtype=np.float64)
    out = np.empty_like(pts)
    fn = lib().orc_undistort_points_fisheye if distortion_model == 'equidistant' else lib().orc_undistort_points
    fn(_dp(pts), C.c_int(pts.shape[0]), _dp(kd), _dp(dd), _dp(R), _dp(out))
    return out.astype(np.float32) if out_f32 else out


def distort_points(pts_in, intrinsics, distortion_coeffs, distortion_model='radtan'):
    """cv2.projectPoints(convertPointsToHomogeneous(pts), 0, 0, K, D) (camera_model.py:49-75); distortion_model
    'equidistant' = cv2.fisheye.distortPoints(pts, K, D) (camera_model.py:69-70; parity unpinned, see imgops.c)."""
    arr = np.asarray(pts_in)
    out_f32 = arr.dtype == np.float32
    pts = np.ascontiguousarray(arr.reshape(-1, 2), dtype=np.float64)
    kd = np.ascontiguousarray(intrinsics, dtype=np.float64)
    dd = np.ascontiguousarray(distortion_coeffs, dtype=np.float64)
    out = np.empty_like(pts)
    fn = lib().orc_distort_points_fisheye if distortion_model == 'equidistant' else lib().orc_distort_points
    fn(_dp(pts), C.c_int(pts.shape[0]), _dp(kd), _dp(dd), _dp(out))
    return out.astype(np.float32) if out_f32 else out


def rodrigues(rvec):
    """cv2.Rodrigues(vector)[0] (imu_processor.py:63-64): axis-angle -> rotation matrix."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = math.sqrt((float(r[0]) * float(r[0]) + float(r[1]) * float(r[1])) + float(r[2]) * float(r[2]))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = math.cos(theta), math.sin(theta)      # libm, like cv::Rodrigues
    c1 = 1. - c
    x, y, z = r * (1. / theta)
    rrt = np.array([[x * x, x * y, x * z], [x * y, y * y, y * z], [x * z, y * z, z * z]])
    rx = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return c * np.eye(3) + c1 * rrt + s * rx
