"""Parity of every HIP operator against the CPU oracle, called through the C ABI (bit-exact).

The bar (task statement, section 3): bit-exact for integer / byte / index work.  The LK points
are float32 but are compared bit-for-bit too: the window sums are exact integers in both
implementations and all float expressions are evaluated in the same order without contraction.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from uav_airvision_amd import ops as o
    return o


def test_pyramid_matches_oracle_including_border(ops, frames0):
    from oracle import cvops
    img = frames0[0].cam0_image
    pyr, lay = ops.build_pyramids(np.stack([img, frames0[0].cam1_image]), 4)
    ref0 = cvops.build_pyramid(img, 3)
    ref1 = cvops.build_pyramid(frames0[0].cam1_image, 3)
    for l in range(4):
        for row, ref in ((0, ref0), (1, ref1)):
            got = ops.pyramid_level(pyr[row], lay, l)
            assert got.shape == ref[l].shape
            assert np.array_equal(got, ref[l]), 'level %d interior differs' % l
            full = ops.pyramid_level(pyr[row], lay, l, with_border=True)
            exp = np.pad(ref[l], 16, mode='reflect')          # numpy 'reflect' == BORDER_REFLECT_101
            assert np.array_equal(full, exp), 'level %d border differs' % l


def test_pyramid_odd_sizes(ops):
    from oracle import cvops
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (203, 317), dtype=np.uint8)
    pyr, lay = ops.build_pyramids(img, 3)
    ref = cvops.build_pyramid(img, 2)
    for l in range(3):
        assert np.array_equal(ops.pyramid_level(pyr[0], lay, l), ref[l])


def _lk_both(ops, I, J, prev, init, cfg):
    from oracle import cvops
    a = ops.calc_optical_flow_pyr_lk(I, J, prev, init, **cfg.lk_params)
    b = cvops.calc_optical_flow_pyr_lk(I, J, prev, init, **cfg.lk_params)
    return a, b


def test_lk_temporal_bit_exact(ops, frames0, cfg):
    from oracle import cvops
    xs, ys, sc = cvops.fast_detect(frames0[0].cam0_image, cfg.fast_threshold)
    sel = np.linspace(0, len(xs) - 1, 700).astype(int)
    prev = np.stack([xs[sel], ys[sel]], 1).astype(np.float32) + np.float32(0.25)
    (p_gpu, s_gpu, _), (p_cpu, s_cpu, _) = _lk_both(ops, frames0[0].cam0_image, frames0[1].cam0_image, prev, prev.copy(), cfg)
    assert np.array_equal(s_gpu, s_cpu)
    assert s_cpu.sum() > 600
    assert np.array_equal(p_gpu.view(np.uint32), p_cpu.view(np.uint32)), np.abs(p_gpu - p_cpu).max()
    assert np.abs(p_cpu - prev)[s_cpu[:, 0] > 0].max() > 0.05      # the points really moved


def test_lk_stereo_and_edge_cases_bit_exact(ops, frames0, cfg):
    rng = np.random.default_rng(7)
    n = 600
    prev = np.stack([rng.uniform(-20, 772, n), rng.uniform(-20, 500, n)], 1).astype(np.float32)   # incl. out of image
    init = prev + rng.normal(0, 6, (n, 2)).astype(np.float32)
    init[:50] += 300.0                                                                                 # windows that leave the image
    (p_gpu, s_gpu, _), (p_cpu, s_cpu, _) = _lk_both(ops, frames0[2].cam0_image, frames0[2].cam1_image, prev, init, cfg)
    assert np.array_equal(s_gpu, s_cpu)
    assert 0 < s_cpu.sum() < n
    assert np.array_equal(p_gpu.view(np.uint32), p_cpu.view(np.uint32))


@pytest.mark.parametrize('win,max_level', [(9, 3), (21, 3), (31, 2), (16, 1), (5, 4), (15, 1), (15, 4), (29, 4), (31, 4)])
def test_lk_other_windows_and_levels_bit_exact(ops, frames0, win, max_level):
    """config.win_size / pyramid levels other than the reference's defaults (config.py:31-44 are configuration, not constants):
    windows other than 15 take the general one-wavefront-per-point kernel, 15 x 15 with another maxLevel the 16-lane kernel.
    Same bar as the default: status equal, points bit-identical, incl. points outside the image and windows that leave it."""
    from oracle import cvops
    rng = np.random.default_rng(100 + win)
    xs, ys, _ = cvops.fast_detect(frames0[0].cam0_image, 15)
    sel = np.linspace(0, len(xs) - 1, 260).astype(int)
    prev = np.concatenate([np.stack([xs[sel], ys[sel]], 1).astype(np.float32) + np.float32(0.25),
                           np.stack([rng.uniform(-25, 777, 60), rng.uniform(-25, 505, 60)], 1).astype(np.float32)])
    init = prev + rng.normal(0, 3, prev.shape).astype(np.float32)
    init[-10:] += 400.0
    kw = dict(winSize=(win, win), maxLevel=max_level, criteria=(3, 30, 0.01), flags=4, minEigThreshold=1e-4)
    for I, J in ((frames0[0].cam0_image, frames0[1].cam0_image), (frames0[2].cam0_image, frames0[2].cam1_image)):
        p_gpu, s_gpu, _ = ops.calc_optical_flow_pyr_lk(I, J, prev, init, **kw)
        p_cpu, s_cpu, _ = cvops.calc_optical_flow_pyr_lk(I, J, prev, init, **kw)
        assert np.array_equal(s_gpu, s_cpu), (win, max_level, int((s_gpu != s_cpu).sum()))
        assert 50 < s_cpu.sum() < len(prev)
        assert np.array_equal(p_gpu.view(np.uint32), p_cpu.view(np.uint32)), (win, max_level, np.abs(p_gpu - p_cpu).max())


@pytest.mark.parametrize('w,h,win,max_level', [(18, 20, 31, 2), (40, 36, 21, 2), (33, 17, 15, 1), (64, 48, 23, 1)])
def test_lk_small_images_bit_exact(ops, w, h, win, max_level):
    """Images barely larger than -- or smaller than -- the window: windows reach more than one image width past the border (the
    border interpolation has to loop like cv::borderInterpolate) and the levels OpenCV would not build are dropped on both sides
    ((29, 4) / (31, 4) above: the 47 x 30 top level is kept at win 29 and dropped at win 31)."""
    from oracle import cvops
    rng = np.random.default_rng(w * 100 + h)
    I = rng.integers(0, 256, (h, w), dtype=np.uint8)
    I = ((I.astype(np.int32) + np.roll(I, 1, 0) + np.roll(I, 1, 1) + np.roll(I, (1, 1), (0, 1))) // 4).astype(np.uint8)
    J = np.roll(I, (1, -1), (0, 1))
    prev = np.stack([rng.uniform(-4, w + 4, 80), rng.uniform(-4, h + 4, 80)], 1).astype(np.float32)
    init = prev + rng.normal(0, 1.5, prev.shape).astype(np.float32)
    kw = dict(winSize=(win, win), maxLevel=max_level, criteria=(3, 30, 0.01), flags=4, minEigThreshold=1e-4)
    p_gpu, s_gpu, _ = ops.calc_optical_flow_pyr_lk(I, J, prev, init, **kw)
    p_cpu, s_cpu, _ = cvops.calc_optical_flow_pyr_lk(I, J, prev, init, **kw)
    assert np.array_equal(s_gpu, s_cpu), int((s_gpu != s_cpu).sum())
    assert s_cpu.sum() > 0
    assert np.array_equal(p_gpu.view(np.uint32), p_cpu.view(np.uint32)), np.abs(p_gpu - p_cpu).max()


@pytest.mark.parametrize('eps,min_eig', [(0.0, 1e-4), (1e-20, 1e-4), (0.003, 1e-4), (0.05, 3e-3), (0.3, 0.02), (10.0, 1e-4)])
def test_lk_termination_tests_in_float_match_the_fp64_forms(ops, frames0, eps, min_eig):
    """The 16-lane kernel takes OpenCV's fp64 tests (`delta.ddot(delta) <= eps^2`, `minEig < minEigThreshold`, the 0.01 oscillation
    stop) in float where float decides the same way (lk.hip: eps_lo / eps_hi / min_eig_up) and in fp64 next to the threshold.
    eps = 0 and 1e-20 send EVERY step test down the fp64 branch (the float sum can underflow), the others down the float one;
    the minEig thresholds cut the point set in different places.  All against the oracle's fp64 forms, bit for bit."""
    from oracle import cvops
    rng = np.random.default_rng(int(eps * 1000) + 17)
    xs, ys, _ = cvops.fast_detect(frames0[0].cam0_image, 15)
    sel = np.linspace(0, len(xs) - 1, 500).astype(int)
    prev = np.concatenate([np.stack([xs[sel], ys[sel]], 1).astype(np.float32) + np.float32(0.25),
                           np.stack([rng.uniform(0, 752, 200), rng.uniform(0, 480, 200)], 1).astype(np.float32)])   # incl. weak texture
    init = prev + rng.normal(0, 2, prev.shape).astype(np.float32)
    kw = dict(winSize=(15, 15), maxLevel=3, criteria=(3, 30, eps), flags=4, minEigThreshold=min_eig)
    for I, J in ((frames0[0].cam0_image, frames0[1].cam0_image), (frames0[2].cam0_image, frames0[2].cam1_image)):
        p_gpu, s_gpu, _ = ops.calc_optical_flow_pyr_lk(I, J, prev, init, **kw)
        p_cpu, s_cpu, _ = cvops.calc_optical_flow_pyr_lk(I, J, prev, init, **kw)
        assert np.array_equal(s_gpu, s_cpu), (eps, min_eig, int((s_gpu != s_cpu).sum()))
        assert 100 < s_cpu.sum() <= len(prev)
        assert np.array_equal(p_gpu.view(np.uint32), p_cpu.view(np.uint32)), (eps, min_eig, np.abs(p_gpu - p_cpu).max())


def test_lk_flat_image_fails_min_eig(ops, cfg):
    I = np.full((480, 752), 90, np.uint8)
    prev = np.array([[100.5, 100.25], [300, 200]], np.float32)
    (p_gpu, s_gpu, _), (p_cpu, s_cpu, _) = _lk_both(ops, I, I, prev, prev.copy(), cfg)
    assert s_gpu.sum() == 0 and s_cpu.sum() == 0
    assert np.array_equal(p_gpu.view(np.uint32), p_cpu.view(np.uint32))


def test_lk_empty(ops, cfg, frames0):
    p, s, _ = ops.calc_optical_flow_pyr_lk(frames0[0].cam0_image, frames0[1].cam0_image, np.zeros((0, 2), np.float32),
                                           np.zeros((0, 2), np.float32), **cfg.lk_params)
    assert p.shape == (0, 2) and s.shape == (0, 1)


def test_fast_matches_oracle(ops, frames0, cfg):
    from oracle import cvops
    img = frames0[1].cam0_image
    gx, gy, gs = ops.fast_detect(img, cfg.fast_threshold)
    cx, cy, cs = cvops.fast_detect(img, cfg.fast_threshold)
    assert len(cx) > 2000
    assert np.array_equal(gx, cx) and np.array_equal(gy, cy) and np.array_equal(gs, cs)


def test_fast_mask_and_threshold(ops, frames0):
    from oracle import cvops
    img = frames0[3].cam1_image
    rng = np.random.default_rng(5)
    mask = (rng.uniform(size=img.shape) > 0.3).astype(np.uint8)
    for thr in (7, 40):
        g = ops.fast_detect(img, thr, mask)
        c = cvops.fast_detect(img, thr, mask)
        assert len(c[0]) > 50
        for a, b in zip(g, c):
            assert np.array_equal(a, b)


def test_fast_random_noise_image(ops):
    from oracle import cvops
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (97, 131), dtype=np.uint8)          # ragged size, dense corners, many score ties
    g = ops.fast_detect(img, 20)
    c = cvops.fast_detect(img, 20)
    assert len(c[0]) > 100
    for a, b in zip(g, c):
        assert np.array_equal(a, b)


def test_fast_capacity_error(ops, frames0, cfg):
    from uav_airvision_amd._native import AirvisionError
    with pytest.raises(AirvisionError):
        ops.fast_detect(frames0[0].cam0_image, cfg.fast_threshold, cap=64)


def test_equidistant_undistort_distort_match_the_oracle(ops, cfg):
    """camera_model.py:41-43, 69-70 with distortion_model 'equidistant' (cv2.fisheye.*): same restatement on both sides; the only
    difference is tan / atan from the device math library against libm, so the comparison allows a few ulp of fp64 and float32
    results must agree to the last bit almost everywhere."""
    from oracle import cvops
    K = np.array([461.6, 460.3, 362.7, 248.1]); D = np.array([-0.0126, 0.0129, -0.0161, 0.0062])
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(-30, 780, 4000), rng.uniform(-30, 510, 4000)], 1)
    pts[0] = (K[2], K[3])                                           # the principal point: theta_d below epsilon
    R = cvops.rodrigues(np.array([0.01, -0.02, 0.005]))
    for dtype in (np.float32, np.float64):
        p = pts.astype(dtype)
        a = ops.undistort_points(p, K, D, R, distortion_model='equidistant')
        b = cvops.undistort_points(p, K, D, R, distortion_model='equidistant')
        assert a.dtype == b.dtype == dtype
        if dtype == np.float64:
            assert np.abs(a - b).max() <= 4e-16 * max(1.0, np.abs(b).max())
        else:
            assert (a != b).mean() < 1e-3 and np.abs(a - b).max() < 1e-6
        a2 = ops.distort_points(b, K, D, distortion_model='equidistant')
        b2 = cvops.distort_points(b, K, D, distortion_model='equidistant')
        if dtype == np.float64:
            assert np.abs(a2 - b2).max() <= 1e-12
        else:
            assert (a2 != b2).mean() < 1e-3 and np.abs(a2 - b2).max() < 1e-3
    # the radtan entry points and the model argument agree
    assert np.array_equal(ops.undistort_points(pts, cfg.cam0_intrinsics, cfg.cam0_distortion_coeffs, R, distortion_model='radtan'),
                          cvops.undistort_points(pts, cfg.cam0_intrinsics, cfg.cam0_distortion_coeffs, R))
    from uav_airvision_amd import _native as N
    import torch
    d = torch.zeros(4, dtype=torch.float64, device='cuda')
    with pytest.raises(N.AirvisionError, match='distortion model'):
        N.check(N.lib().av_undistort_points_model(N.dptr(d), 2, N.darr(K), N.darr(D), None, 7, N.dptr(d), None))


def test_undistort_distort_bit_exact(ops, cfg):
    from oracle import cvops
    rng = np.random.default_rng(2)
    pts = np.stack([rng.uniform(-30, 780, 2000), rng.uniform(-30, 510, 2000)], 1)
    R = cvops.rodrigues(np.array([0.01, -0.02, 0.005]))
    for dtype in (np.float32, np.float64):
        p = pts.astype(dtype)
        for K, D in ((cfg.cam0_intrinsics, cfg.cam0_distortion_coeffs), (cfg.cam1_intrinsics, cfg.cam1_distortion_coeffs)):
            a = ops.undistort_points(p, K, D, R)
            b = cvops.undistort_points(p, K, D, R)
            assert a.dtype == b.dtype == dtype
            assert np.array_equal(a, b)
            a2 = ops.distort_points(b, K, D)
            b2 = cvops.distort_points(b, K, D)
            assert np.array_equal(a2, b2)
    assert ops.undistort_points(np.zeros((0, 2)), cfg.cam0_intrinsics, cfg.cam0_distortion_coeffs).shape == (0, 2)
