"""Known-answer and property tests that pin the CPU oracle's image operators (oracle/imgops.c).

OpenCV itself is not available, so these are pinned by independent restatements of
the published definitions written a second way in numpy, and by analytic cases."""
import numpy as np
import pytest

from oracle import cvops


def _pyr_down_numpy(img):
    """pyrDown by its definition: reflect-101 pad, separable [1 4 6 4 1], decimate, (s+128)>>8."""
    h, w = img.shape
    k = np.array([1, 4, 6, 4, 1], np.int64)
    p = np.pad(img.astype(np.int64), 2, mode='reflect')
    dh, dw = (h + 1) // 2, (w + 1) // 2
    rows = sum(k[i] * p[:, i:i + 2 * dw:2][:, :dw] for i in range(5))           # horizontal, decimated
    out = sum(k[i] * rows[i:i + 2 * dh:2][:dh] for i in range(5))
    return ((out + 128) >> 8).astype(np.uint8)


@pytest.mark.parametrize('shape', [(480, 752), (61, 95), (60, 94), (33, 40)])
def test_pyr_down_matches_definition(shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(cvops.pyr_down(img), _pyr_down_numpy(img))


def test_pyr_down_constant_and_sizes():
    img = np.full((480, 752), 77, np.uint8)
    pyr = cvops.build_pyramid(img, 3)
    assert [p.shape for p in pyr] == [(480, 752), (240, 376), (120, 188), (60, 94)]
    assert all((p == 77).all() for p in pyr)


def _fast_bruteforce(img, t):
    """FAST-9/16 + score + NMS straight from the textbook definition (slow, tiny images only)."""
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    h, w = img.shape
    sc = np.zeros((h, w), np.int32)
    im = img.astype(np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = [im[y, x] - im[y + dy[k], x + dx[k]] for k in range(16)]
            best = -999
            for s in range(16):
                arc = [d[(s + j) % 16] for j in range(9)]
                best = max(best, min(arc), min(-v for v in arc))
            if best > t:
                sc[y, x] = best - 1          # largest threshold for which it is still a corner
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = sc[y, x]
            if s and all(s > sc[y + j, x + i] for j in (-1, 0, 1) for i in (-1, 0, 1) if (i, j) != (0, 0)):
                out.append((x, y, s))
    return out


def test_fast_matches_bruteforce_definition():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 52), dtype=np.uint8)
    img[10:25, 12:30] = 200
    img[15:20, 35:45] = 20
    for t in (10, 30):
        xs, ys, sc = cvops.fast_detect(img, t)
        got = list(zip(xs.tolist(), ys.tolist(), sc.tolist()))
        assert got == _fast_bruteforce(img, t)
        assert len(got) > 3


def test_fast_score_is_max_threshold():
    """response == the largest threshold at which the pixel is still detected (ignoring NMS)."""
    img = np.full((21, 21), 100, np.uint8)
    img[10, 10] = 160                                   # isolated bright dot: all 16 ring pixels darker by 60
    xs, ys, sc = cvops.fast_detect(img, 15)
    assert (xs.tolist(), ys.tolist(), sc.tolist()) == ([10], [10], [59])
    assert len(cvops.fast_detect(img, 59)[0]) == 1 and len(cvops.fast_detect(img, 60)[0]) == 0


def test_fast_mask_filters_after_detection_and_border():
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    xs, ys, sc = cvops.fast_detect(img, 20)
    assert xs.min() >= 3 and ys.min() >= 3 and xs.max() <= 60 and ys.max() <= 60
    mask = np.ones_like(img); mask[:, :32] = 0
    mx, my, ms = cvops.fast_detect(img, 20, mask)
    keep = xs >= 32
    assert np.array_equal(mx, xs[keep]) and np.array_equal(my, ys[keep]) and np.array_equal(ms, sc[keep])
    order = ys.astype(np.int64) * 64 + xs
    assert (np.diff(order) > 0).all()                   # raster order


def _blob_image(shift=(0.0, 0.0), seed=1, size=(240, 320)):
    """Smooth random texture sampled with a sub-pixel shift (analytic ground truth for LK)."""
    h, w = size
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    xx = xx - shift[0]; yy = yy - shift[1]
    img = np.zeros((h, w))
    for _ in range(120):
        cx, cy, s, a = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(4, 12), rng.uniform(-60, 60)
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return np.clip(img + 128, 0, 255).round().astype(np.uint8)


def test_lk_recovers_subpixel_translation():
    I = _blob_image()
    J = _blob_image(shift=(3.4, -2.25))
    ys, xs = np.mgrid[40:200:20, 40:280:20]
    prev = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32)
    nxt, st, _ = cvops.calc_optical_flow_pyr_lk(I, J, prev, prev.copy())
    ok = st.ravel() > 0
    assert ok.mean() > 0.6                                  # smooth blobs: some windows fail the minEig gate
    err = np.abs(nxt - (prev + np.float32([3.4, -2.25])))[ok]
    assert err.max() < 0.25 and err.mean() < 0.05


def test_lk_uses_initial_flow_and_status_rules():
    I = _blob_image()
    prev = np.array([[100, 100], [5000, 5000], [60.5, 40.25]], np.float32)
    init = prev.copy(); init[0] += 1.5
    nxt, st, _ = cvops.calc_optical_flow_pyr_lk(I, I, prev, init)
    assert st.ravel().tolist() == [1, 0, 1]                     # far outside the image -> status 0
    assert np.abs(nxt[0] - prev[0]).max() < 0.05                # pulled back from the wrong initial guess
    assert np.array_equal(nxt[2], prev[2])                      # zero residual: delta == 0 exactly, bits untouched
    assert np.array_equal(nxt[1], prev[1])
    flat = np.full_like(I, 50)
    _, st2, _ = cvops.calc_optical_flow_pyr_lk(flat, flat, prev[:1], prev[:1].copy())
    assert st2.ravel().tolist() == [0]                          # minEig gate


def test_lk_zero_iterations_and_empty():
    I = _blob_image()
    prev = np.array([[100, 100]], np.float32)
    nxt, st, _ = cvops.calc_optical_flow_pyr_lk(I, I, prev, prev + 2, criteria=(3, 0, 0.01))
    assert st.all() and np.array_equal(nxt, prev + 2)
    nxt, st, _ = cvops.calc_optical_flow_pyr_lk(I, I, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert nxt.shape == (0, 2) and st.shape == (0, 1)


def test_undistort_inverts_distort(cfg):
    rng = np.random.default_rng(9)
    norm = rng.uniform(-0.6, 0.6, (500, 2))
    for K, D in ((cfg.cam0_intrinsics, cfg.cam0_distortion_coeffs), (cfg.cam1_intrinsics, cfg.cam1_distortion_coeffs)):
        pix = cvops.distort_points(norm, K, D)
        back = cvops.undistort_points(pix, K, D)
        assert np.abs(back - norm).max() < 2e-4                 # 5 fixed-point iterations, as OpenCV
        # closed form of the forward model
        x, y = norm[:, 0], norm[:, 1]
        r2 = x * x + y * y
        cd = 1 + D[0] * r2 + D[1] * r2 * r2
        u = (x * cd + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)) * K[0] + K[2]
        v = (y * cd + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y) * K[1] + K[3]
        assert np.allclose(pix, np.stack([u, v], 1), rtol=0, atol=1e-9)


def test_undistort_dtype_rule_and_rectification(cfg):
    K, D = cfg.cam0_intrinsics, cfg.cam0_distortion_coeffs
    p32 = np.array([[100.5, 200.25], [700, 400]], np.float32)
    a = cvops.undistort_points(p32, K, D)
    b = cvops.undistort_points(p32.astype(np.float64), K, D)
    assert a.dtype == np.float32 and b.dtype == np.float64
    assert np.array_equal(a, b.astype(np.float32))
    R = cvops.rodrigues([0.0, 0.1, 0.0])
    c = cvops.undistort_points(p32.astype(np.float64), K, D, R)
    ray = np.c_[b, np.ones(2)] @ R.T
    assert np.allclose(c, ray[:, :2] / ray[:, 2:], atol=1e-12)


def test_rodrigues_matches_scipy():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    for _ in range(20):
        v = rng.normal(0, 0.5, 3)
        assert np.allclose(cvops.rodrigues(v), Rotation.from_rotvec(v).as_matrix(), atol=1e-14)
    assert np.array_equal(cvops.rodrigues(np.zeros(3)), np.eye(3))



def test_equidistant_model_against_an_independent_solution():
    """The 'equidistant' branch of camera_model.py:41-43, 69-70 (cv2.fisheye.undistortPoints / distortPoints).  cv2 is not
    installable here and the reference holds no vector for this model, so the C restatement (oracle/imgops.c, from OpenCV 4.x
    fisheye.cpp) is checked against the model's definition solved another way: theta_d = theta (1 + k1 theta^2 + ... + k4
    theta^8), r_d = theta_d, r_u = tan(theta) -- forward by the formula in numpy, inverse by bisection on the monotone
    polynomial -- plus the round trip, the principal point, the rotation and the dtype rule."""
    from oracle import cvops
    K = np.array([461.6, 460.3, 362.7, 248.1])
    D = np.array([-0.0126, 0.0129, -0.0161, 0.0062])            # a typical Kannala-Brandt calibration of a wide-angle lens
    rng = np.random.default_rng(5)
    px = np.stack([rng.uniform(5, 747, 3000), rng.uniform(5, 475, 3000)], 1)
    und = cvops.undistort_points(px, K, D, distortion_model='equidistant')
    assert und.dtype == np.float64 and und.shape == px.shape
    # independent inverse: bisection for theta in [0, pi/2)
    xd = (px[:, 0] - K[2]) / K[0]; yd = (px[:, 1] - K[3]) / K[1]
    rd = np.hypot(xd, yd)
    poly = lambda t: t * (1 + D[0] * t**2 + D[1] * t**4 + D[2] * t**6 + D[3] * t**8)
    lo, hi = np.zeros_like(rd), np.full_like(rd, 1.5)
    for _ in range(80):
        mid = 0.5 * (lo + hi); big = poly(mid) > rd
        hi = np.where(big, mid, hi); lo = np.where(big, lo, mid)
    theta = 0.5 * (lo + hi)
    scale = np.tan(theta) / rd
    assert np.abs(und - np.stack([xd * scale, yd * scale], 1)).max() < 1e-9
    # forward model by the formula, and the round trip through both C functions
    r = np.hypot(und[:, 0], und[:, 1]); th = np.arctan(r)
    ref = np.stack([und[:, 0] * poly(th) / r * K[0] + K[2], und[:, 1] * poly(th) / r * K[1] + K[3]], 1)
    back = cvops.distort_points(und, K, D, distortion_model='equidistant')
    assert np.abs(back - ref).max() < 1e-9 and np.abs(back - px).max() < 1e-6
    # the principal point maps to the optical axis (theta_d below the criteria's epsilon: scale 0), and back to itself
    c = cvops.undistort_points(np.array([[K[2], K[3]]]), K, D, distortion_model='equidistant')
    assert np.array_equal(c, np.zeros((1, 2)))
    assert np.allclose(cvops.distort_points(np.zeros((1, 2)), K, D, distortion_model='equidistant'), [[K[2], K[3]]], atol=0, rtol=0)
    # rectification matrix applied to (x, y, 1), then the perspective division -- as in the radtan wrapper
    R = cvops.rodrigues(np.array([0.02, -0.01, 0.03]))
    ur = cvops.undistort_points(px[:50], K, D, R, distortion_model='equidistant')
    h = (R @ np.concatenate([und[:50], np.ones((50, 1))], 1).T).T
    assert np.abs(ur - h[:, :2] / h[:, 2:]).max() < 1e-12
    # float32 in -> float32 out (OpenCV's depth rule, which the feature publisher relies on)
    assert cvops.undistort_points(px.astype(np.float32), K, D, distortion_model='equidistant').dtype == np.float32
    # and the two models really differ
    assert np.abs(und - cvops.undistort_points(px, K, D)).max() > 1e-3
