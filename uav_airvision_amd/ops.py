"""Operator-level Python front of the C ABI: numpy in / numpy out, computed on cuda:<device>.

One function per third-party native call of the reference, with the
argument meaning of the reference call sites so the parity tests read like the reference code.
torch is used only to own device memory and the stream; all arithmetic is in libairvision_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N


def _dev(device):
    return torch.device('cuda', device)


def pyramid_layout(w, h, levels):
    lay = N.PyrLayout()
    N.check(N.lib().av_pyramid_layout(w, h, levels, C.byref(lay)))
    return lay


def build_pyramids(images, levels, device=0):
    """images: uint8[n,h,w] (numpy or cuda tensor) -> (uint8 cuda tensor [n, bytes], layout)."""
    t = torch.as_tensor(images)
    if t.dim() == 2:
        t = t[None]
    t = t.to(_dev(device), dtype=torch.uint8).contiguous()
    n, h, w = t.shape
    lay = pyramid_layout(w, h, levels)
    pyr = torch.empty((n, lay.bytes), dtype=torch.uint8, device=_dev(device))
    with torch.cuda.device(device):
        N.check(N.lib().av_pyramid_build(N.dptr(t), w * h, n, w, h, levels, N.dptr(pyr), lay.bytes, N.current_stream()))
    return pyr, lay


def pyramid_level(pyr_row, lay, level, with_border=False):
    """numpy view of one level of one pyramid (for tests)."""
    a = pyr_row.cpu().numpy()
    B = N.AV_PYR_BORDER
    w, h, pitch, off = lay.w[level], lay.h[level], lay.pitch[level], lay.offset[level]
    full = a[off:off + pitch * (h + 2 * B)].reshape(h + 2 * B, pitch)
    return full[:, :w + 2 * B] if with_border else full[B:B + h, B:B + w]


def calc_optical_flow_pyr_lk(prev_img, next_img, prev_pts, next_pts, winSize=(15, 15), maxLevel=3,
                             criteria=(3, 30, 0.01), flags=4, minEigThreshold=1e-4, device=0):
    """cv2.calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, **lk_params) on the GPU
    (reference: feature_tracker.py:102-108, stereo_matcher.py:64-74, config.py:37-44).
    Returns (next_pts float32[N,2], status uint8[N,1], None)."""
    if not (flags & 4):
        raise ValueError('only OPTFLOW_USE_INITIAL_FLOW is implemented (the reference always sets it)')
    ctype, max_iter, eps = criteria
    max_iter = int(max_iter) if (ctype & 1) else 30
    eps = float(eps) if (ctype & 2) else 0.01
    prev = np.ascontiguousarray(np.asarray(prev_pts, dtype=np.float32).reshape(-1, 2))
    nxt = np.ascontiguousarray(np.asarray(next_pts, dtype=np.float32).reshape(-1, 2))
    n = prev.shape[0]
    if n == 0:
        return nxt.copy(), np.zeros((0, 1), np.uint8), None
    imgs = np.stack([np.asarray(prev_img, dtype=np.uint8), np.asarray(next_img, dtype=np.uint8)])
    # levels OpenCV would not build are not laid out either (cv::buildOpticalFlowPyramid stops before the first level that is
    # no larger than the window; av_lk_track applies the same rule to whatever it is handed)
    lw, lh, eff = imgs.shape[2], imgs.shape[1], 0
    while eff < maxLevel and (lw + 1) // 2 > winSize[0] and (lh + 1) // 2 > winSize[1]:
        lw, lh, eff = (lw + 1) // 2, (lh + 1) // 2, eff + 1
    maxLevel = eff
    pyr, lay = build_pyramids(imgs, maxLevel + 1, device)
    dev = _dev(device)
    d_prev = torch.from_numpy(prev).to(dev)
    d_next = torch.from_numpy(nxt).to(dev)
    d_status = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_count = torch.tensor([n], dtype=torch.int32, device=dev)
    h, w = imgs.shape[1:]
    with torch.cuda.device(device):
        N.check(N.lib().av_lk_track(N.dptr(pyr[0]), N.dptr(pyr[1]), lay.bytes, 1, w, h, maxLevel + 1,
                                    N.dptr(d_prev), N.dptr(d_next), N.dptr(d_status), N.dptr(d_count), n,
                                    int(winSize[0]), max_iter, eps, float(minEigThreshold), N.current_stream()))
        torch.cuda.synchronize()
    return d_next.cpu().numpy(), d_status.cpu().numpy().reshape(-1, 1), None


def fast_detect(img, threshold, mask=None, cap=1 << 16, device=0):
    """cv2.FastFeatureDetector_create(threshold).detect(img, mask): (x, y, response) int arrays in
    raster order (reference: pipeline.py:23-25, feature_initializer.py:52, feature_adder.py:64)."""
    dev = _dev(device)
    t = torch.as_tensor(np.ascontiguousarray(img, dtype=np.uint8)).to(dev)
    h, w = t.shape
    m = None if mask is None else torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8)).to(dev)
    kp = torch.empty(cap, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(device):
        N.check(N.lib().av_fast_detect(N.dptr(t), w * h, None if m is None else N.dptr(m), w * h, 1, w, h, int(threshold),
                                       N.dptr(kp), N.dptr(cnt), cap, N.current_stream()))
        torch.cuda.synchronize()
    n = int(cnt.item())
    if n > cap:
        raise N.AirvisionError(N.AV_E_CAPACITY, 'FAST found %d keypoints, capacity %d' % (n, cap))
    words = kp[:n].cpu().numpy().view(np.uint32)
    raster = (np.uint32((1 << 19) - 1) - (words & np.uint32((1 << 19) - 1))).astype(np.int64)
    score = (words >> np.uint32(19)).astype(np.int32)
    order = np.argsort(raster, kind='stable')
    raster, score = raster[order], score[order]
    return (raster % w).astype(np.int32), (raster // w).astype(np.int32), score


def _points_op(fn_name, pts_in, extra, device):
    arr = np.asarray(pts_in)
    out_f32 = arr.dtype == np.float32
    pts = np.ascontiguousarray(arr.reshape(-1, 2), dtype=np.float64)
    n = pts.shape[0]
    if n == 0:
        return pts.astype(np.float32) if out_f32 else pts
    dev = _dev(device)
    d_in = torch.from_numpy(pts).to(dev)
    d_out = torch.empty_like(d_in)
    with torch.cuda.device(device):
        N.check(getattr(N.lib(), fn_name)(N.dptr(d_in), n, *extra, N.dptr(d_out), N.current_stream()))
        torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    return out.astype(np.float32) if out_f32 else out


def undistort_points(pts_in, intrinsics, distortion_coeffs, rectification_matrix=None, device=0, distortion_model='radtan'):
    """cv2.undistortPoints(pts, K, D, None, R, P=I) -- or, for distortion_model 'equidistant', cv2.fisheye.undistortPoints(pts, K,
    D, R, P=I) (reference: camera_model.py:24-47)."""
    R = np.eye(3) if rectification_matrix is None else np.asarray(rectification_matrix, dtype=np.float64)
    extra = (N.darr(intrinsics), N.darr(distortion_coeffs), N.darr(R.reshape(-1)), N.distortion_model_code(distortion_model))
    return _points_op('av_undistort_points_model', pts_in, extra, device)


def distort_points(pts_in, intrinsics, distortion_coeffs, device=0, distortion_model='radtan'):
    """cv2.projectPoints(homogeneous(pts), 0, 0, K, D) -- or cv2.fisheye.distortPoints(pts, K, D) for 'equidistant' (reference:
    camera_model.py:49-75)."""
    extra = (N.darr(intrinsics), N.darr(distortion_coeffs), N.distortion_model_code(distortion_model))
    return _points_op('av_distort_points_model', pts_in, extra, device)
