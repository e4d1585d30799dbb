"""Pins oracle/imgops.c to the REAL OpenCV -- when tests/golden/cv2_golden.npz exists.

The file is written by tests/golden/make_cv2_golden.py on any box that has opencv-python (this container and the GPU boxes have
not: SURVEY.md 8c), from the integer-built inputs of tests/golden/cv2_inputs.py; this module rebuilds the same inputs, checks
their CRCs against the file and replays every call through the CPU oracle.  Without the file every comparison SKIPS (the oracle's
image half stays "parity unpinned"); the input builder itself is always exercised.

Bars: pyrDown, FAST keypoints / responses: bit-exact.  Camera model: 1e-12 (fp64) / float32 rounding.  LK: OpenCV accumulates the
window sums in float in a build-dependent order where the oracle sums them exactly, so positions agree to the iteration's
stopping precision, not bit for bit: status equal on >= 99.5 % of the points the call sites keep, positions of the commonly
tracked points within 0.02 px at the 99th percentile and 1e-3 px in the median.  cv2's `err` block clears the status of points
whose FINAL window leaves the image (imgops.c header, second deviation): such points are excluded from the status comparison,
exactly as the reference's own bounds gates drop them (feature_tracker.py:111-115, stereo_matcher.py:82-88).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import cv2_inputs as ci      # noqa: E402

from oracle import cvops     # noqa: E402

GOLD = os.path.join(HERE, 'golden', 'cv2_golden.npz')


@pytest.fixture(scope='module')
def gold():
    if not os.path.exists(GOLD):
        pytest.skip('tests/golden/cv2_golden.npz not generated yet (needs opencv-python: python tests/golden/make_cv2_golden.py)')
    return np.load(GOLD, allow_pickle=False)


@pytest.fixture(scope='module')
def frames():
    return ci.frames()


def test_input_builder_is_deterministic_and_trackable(frames):
    """Runs with or without the golden file: the inputs are what the generator's docstring says they are."""
    I, Jt, Js = frames
    I2, _, _ = ci.frames()
    assert I.shape == (ci.H, ci.W) and I.dtype == np.uint8 and np.array_equal(I, I2)
    prev, init = ci.points(11, 700, 100)
    assert prev.dtype == np.float32 and np.array_equal(prev * 8, np.round(prev * 8))
    nxt, st, _ = cvops.calc_optical_flow_pyr_lk(I, Jt, prev, init, **ci.LK_REFERENCE)
    ok = st[:, 0] > 0
    assert ok.sum() > 600
    assert np.allclose(np.median((nxt - prev)[ok], 0), [2.5, 1.5], atol=0.01)
    xs, _, _ = cvops.fast_detect(I, 15)
    assert len(xs) > 2000                                   # FAST@15 must yield >= 2,000 corners
    R = ci.rectification()
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15)


def test_inputs_match_the_generators(gold, frames):
    I, Jt, Js = frames
    assert [ci.crc(I), ci.crc(Jt), ci.crc(Js)] == gold['crc_inputs'].tolist()
    prev, init = ci.points(11, 700, 100)
    assert [ci.crc(prev), ci.crc(init)] == gold['crc_points'].tolist()
    assert [ci.crc(ci.fast_mask())] == gold['crc_mask'].tolist()


def test_pyr_down_vs_cv2(gold, frames):
    I = frames[0]
    lev = cvops.build_pyramid(I, 4)
    assert [list(a.shape) for a in lev] == gold['pyr_shapes'].tolist()
    assert np.array_equal(lev[3], gold['pyr_level3'])
    assert [ci.crc(a) for a in lev] == gold['pyr_crc'].tolist()
    odd = np.ascontiguousarray(I[:479, :751])
    d1 = cvops.pyr_down(odd)
    assert [ci.crc(d1), ci.crc(cvops.pyr_down(d1))] == gold['pyr_odd_crc'].tolist()


@pytest.mark.parametrize('t', ci.FAST_THRESHOLDS)
@pytest.mark.parametrize('masked', [False, True])
def test_fast_vs_cv2(gold, frames, t, masked):
    want = gold['fast_t%d%s' % (t, '_mask' if masked else '')]
    xs, ys, sc = cvops.fast_detect(frames[0], t, ci.fast_mask() if masked else None)
    got = np.stack([xs, ys, sc], 1).astype(np.float64).reshape(-1, 3)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want)                        # raster order, integer-valued pt, response = corner score


def _final_window_inside(pts, w, h, win):
    """cv2's err block keeps status only if the final window's corner lies in [-win, cols) x [-win, rows)."""
    half = (win - 1) * 0.5
    ix, iy = np.floor(pts[:, 0] - half), np.floor(pts[:, 1] - half)
    return (ix >= -win) & (ix < w) & (iy >= -win) & (iy < h)


def _compare_lk(nxt, st, want_next, want_status, w, h, win, min_tracked):
    st = st.reshape(-1).astype(bool)
    ws = want_status.reshape(-1).astype(bool)
    # status: compared where the oracle's final window is inside the image (elsewhere cv2's err block clears it; see module docstring)
    cmp = _final_window_inside(nxt, w, h, win)
    agree = (st == ws)[cmp]
    assert agree.mean() >= 0.995, 'status differs on %d of %d points' % ((~agree).sum(), cmp.sum())
    assert not (ws & ~cmp & ~st).any()                       # cv2 never tracks a point the oracle rejects outright
    both = st & ws
    assert both.sum() >= min_tracked, both.sum()
    d = np.abs(nxt[both] - want_next[both]).max(1)
    assert np.median(d) < 1e-3 and np.percentile(d, 99) < 0.02, (float(np.median(d)), float(np.percentile(d, 99)), float(d.max()))


def test_lk_reference_parameters_vs_cv2(gold, frames):
    I, Jt, Js = frames
    prev, init = ci.points(11, 700, 100)
    n, s, _ = cvops.calc_optical_flow_pyr_lk(I, Jt, prev, init, **ci.LK_REFERENCE)
    _compare_lk(n, s, gold['lk_t_next'], gold['lk_t_status'], ci.W, ci.H, 15, 600)
    init_s = init.copy(); init_s[:, 0] -= np.float32(11)
    n, s, _ = cvops.calc_optical_flow_pyr_lk(I, Js, prev, init_s, **ci.LK_REFERENCE)
    _compare_lk(n, s, gold['lk_s_next'], gold['lk_s_status'], ci.W, ci.H, 15, 600)
    # backward pass from cv2's own forward result (stereo_matcher.py:70-74), so a forward difference does not compound
    n, s, _ = cvops.calc_optical_flow_pyr_lk(Js, I, gold['lk_s_next'].astype(np.float32), prev.copy(), **ci.LK_REFERENCE)
    _compare_lk(n, s, gold['lk_b_next'], gold['lk_b_status'], ci.W, ci.H, 15, 600)


@pytest.mark.parametrize('win,max_level', ci.LK_OTHER)
def test_lk_other_windows_vs_cv2(gold, frames, win, max_level):
    I, Jt, _ = frames
    prev, init = ci.points(12, 260, 60, sigma8=16)
    kw = dict(winSize=(win, win), maxLevel=max_level, criteria=(3, 30, 0.01), flags=4, minEigThreshold=1e-4)
    n, s, _ = cvops.calc_optical_flow_pyr_lk(I, Jt, prev, init, **kw)
    _compare_lk(n, s, gold['lk_w%d_l%d_next' % (win, max_level)], gold['lk_w%d_l%d_status' % (win, max_level)], ci.W, ci.H, win, 150)


@pytest.mark.parametrize('w,h,win,max_level', ci.LK_SMALL)
def test_lk_small_images_vs_cv2(gold, w, h, win, max_level):
    a, b = ci.small_frames(w, h, 1000 + w)
    p, g = ci.points(w * 100 + h, 60, 20, w=w, h=h, sigma8=10)
    kw = dict(winSize=(win, win), maxLevel=max_level, criteria=(3, 30, 0.01), flags=4, minEigThreshold=1e-4)
    n, s, _ = cvops.calc_optical_flow_pyr_lk(a, b, p, g, **kw)
    _compare_lk(n, s, gold['lk_small_%dx%d_w%d_next' % (w, h, win)], gold['lk_small_%dx%d_w%d_status' % (w, h, win)], w, h, win, 10)


@pytest.mark.parametrize('cam', ['cam0', 'cam1'])
def test_camera_model_vs_cv2(gold, cam):
    k, d = (ci.CAM0_K, ci.CAM0_D) if cam == 'cam0' else (ci.CAM1_K, ci.CAM1_D)
    px, nrm = ci.camera_points()
    R = ci.rectification()
    for dt, tol in ((np.float64, 1e-12), (np.float32, 0.0)):
        dn = np.dtype(dt).name
        for key, got in (('undist_%s_%s' % (cam, dn), cvops.undistort_points(px.astype(dt), k, d)),
                         ('undist_R_%s_%s' % (cam, dn), cvops.undistort_points(px.astype(dt), k, d, R)),
                         ('dist_%s_%s' % (cam, dn), cvops.distort_points(nrm.astype(dt), k, d))):
            want = gold[key]
            assert got.dtype == want.dtype == dt, key
            if dt is np.float32:
                # computed in double on both sides, rounded once: at most one float32 ulp where the doubles straddle a tie
                assert np.all(np.abs(got - want) <= np.spacing(np.abs(want)).astype(np.float32)), key
            else:
                assert np.allclose(got, want, rtol=tol, atol=tol), (key, np.abs(got - want).max())
    got = cvops.undistort_points(px, k, ci.FISH_D, R, distortion_model='equidistant')
    assert np.allclose(got, gold['fish_undist_%s' % cam], rtol=1e-9, atol=1e-9)
    got = cvops.distort_points(nrm, k, ci.FISH_D, distortion_model='equidistant')
    assert np.allclose(got, gold['fish_dist_%s' % cam], rtol=1e-12, atol=1e-9)


def test_rodrigues_vs_cv2(gold):
    for v, want in zip(ci.RODRIGUES_VECS, gold['rodrigues']):
        assert np.allclose(cvops.rodrigues(v), want, rtol=0, atol=1e-15), v
