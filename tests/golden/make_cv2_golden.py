"""Generates tests/golden/cv2_golden.npz: the outputs of the REAL OpenCV calls the reference makes, on deterministic inputs.

    pip install opencv-python-headless numpy        # any box with network access; OpenCV 4.x (the reference pins no version:
    python tests/golden/make_cv2_golden.py          # /root/reference/requirements.txt:2-3)

cv2 is not installable in the build container, so this script cannot run there: it is the one command that turns
"parity unpinned" for oracle/imgops.c into a pinned oracle.  It imports nothing of this repository except the integer input
builder next to it (tests/golden/cv2_inputs.py, numpy only) -- copy the two files anywhere, run, commit the .npz.
tests/test_oracle_cv2_golden.py replays the same inputs through the CPU oracle and skips while the file does not exist.

Calls covered, at the reference's call sites and parameters:
  cv2.pyrDown                          (inside calcOpticalFlowPyrLK; pyramid_builder.py:22-48 is a pass-through)
  cv2.calcOpticalFlowPyrLK             feature_tracker.py:102-108, stereo_matcher.py:64-74, parameters config.py:31-44;
                                       plus other windows / depths (config.win_size, pyramid_levels are configuration) and images
                                       barely larger than the window (level dropping, looping border interpolation)
  cv2.FastFeatureDetector_create(t)    pipeline.py:23-25, feature_initializer.py:52, feature_adder.py:64 (with and without mask)
  cv2.undistortPoints / projectPoints  camera_model.py:42-45, 70-74, feature_publisher.py:57
  cv2.fisheye.undistortPoints / distortPoints   camera_model.py:41-43, 69-70 (distortion_model 'equidistant')
  cv2.Rodrigues                        imu_processor.py:63-64
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cv2_inputs as ci      # noqa: E402


def main():
    import cv2
    out = {'cv2_version': np.array(cv2.__version__), 'numpy_version': np.array(np.__version__)}
    I, Jt, Js = ci.frames()
    out['crc_inputs'] = np.array([ci.crc(I), ci.crc(Jt), ci.crc(Js)], np.int64)

    # ---- pyrDown: four levels of I; level CRCs + the top level itself
    lev = [I]
    for _ in range(4):
        lev.append(cv2.pyrDown(lev[-1]))
    out['pyr_crc'] = np.array([ci.crc(a) for a in lev], np.int64)
    out['pyr_shapes'] = np.array([a.shape for a in lev], np.int64)
    out['pyr_level3'] = lev[3]
    odd = I[:479, :751]                                          # odd sizes: (w + 1) / 2 rounding, right / bottom border taps
    out['pyr_odd_crc'] = np.array([ci.crc(cv2.pyrDown(odd)), ci.crc(cv2.pyrDown(cv2.pyrDown(odd)))], np.int64)

    # ---- FAST: keypoints in detection order (x, y, response), with and without mask
    mask = ci.fast_mask()
    out['crc_mask'] = np.array([ci.crc(mask)], np.int64)
    for t in ci.FAST_THRESHOLDS:
        det = cv2.FastFeatureDetector_create(t)
        for tag, m in (('', None), ('_mask', mask)):
            kps = det.detect(I, m)
            out['fast_t%d%s' % (t, tag)] = np.array([[kp.pt[0], kp.pt[1], kp.response] for kp in kps], np.float64).reshape(-1, 3)

    # ---- LK at the reference's parameters: temporal pair, stereo pair (forward and backward)
    def lk(a, b, prev, init, **kw):
        nxt, st, err = cv2.calcOpticalFlowPyrLK(a, b, prev.reshape(-1, 1, 2).copy(), init.reshape(-1, 1, 2).copy(), **kw)
        return nxt.reshape(-1, 2), st.reshape(-1), err.reshape(-1)
    prev, init = ci.points(11, 700, 100)
    out['crc_points'] = np.array([ci.crc(prev), ci.crc(init)], np.int64)
    out['lk_t_next'], out['lk_t_status'], out['lk_t_err'] = lk(I, Jt, prev, init, **ci.LK_REFERENCE)
    init_s = init.copy(); init_s[:, 0] -= np.float32(11)
    out['lk_s_next'], out['lk_s_status'], out['lk_s_err'] = lk(I, Js, prev, init_s, **ci.LK_REFERENCE)
    back0 = out['lk_s_next'].copy()
    out['lk_b_next'], out['lk_b_status'], out['lk_b_err'] = lk(Js, I, back0, prev.copy(), **ci.LK_REFERENCE)      # stereo_matcher.py:70-74
    # ---- other windows / depths
    prev2, init2 = ci.points(12, 260, 60, sigma8=16)
    for win, ml in ci.LK_OTHER:
        kw = dict(winSize=(win, win), maxLevel=ml, criteria=(3, 30, 0.01), flags=4, minEigThreshold=1e-4)
        n, s, e = lk(I, Jt, prev2, init2, **kw)
        out['lk_w%d_l%d_next' % (win, ml)], out['lk_w%d_l%d_status' % (win, ml)] = n, s
    # ---- images barely larger than (or smaller than) the window
    for w, h, win, ml in ci.LK_SMALL:
        a, b = ci.small_frames(w, h, 1000 + w)
        p, g = ci.points(w * 100 + h, 60, 20, w=w, h=h, sigma8=10)
        kw = dict(winSize=(win, win), maxLevel=ml, criteria=(3, 30, 0.01), flags=4, minEigThreshold=1e-4)
        n, s, e = lk(a, b, p, g, **kw)
        out['lk_small_%dx%d_w%d_next' % (w, h, win)], out['lk_small_%dx%d_w%d_status' % (w, h, win)] = n, s

    # ---- camera model
    px, nrm = ci.camera_points()
    R = ci.rectification()
    def K(k): return np.array([[k[0], 0, k[2]], [0, k[1], k[3]], [0, 0, 1.0]])
    Pn = np.eye(3)
    for tag, k, d in (('cam0', ci.CAM0_K, ci.CAM0_D), ('cam1', ci.CAM1_K, ci.CAM1_D)):
        for dt in (np.float32, np.float64):
            dn = np.dtype(dt).name
            src = px.astype(dt).reshape(-1, 1, 2)
            out['undist_%s_%s' % (tag, dn)] = cv2.undistortPoints(src, K(k), np.array(d), None, None, Pn).reshape(-1, 2)
            out['undist_R_%s_%s' % (tag, dn)] = cv2.undistortPoints(src, K(k), np.array(d), None, R, Pn).reshape(-1, 2)
            h3 = cv2.convertPointsToHomogeneous(nrm.astype(dt).reshape(-1, 1, 2))
            out['dist_%s_%s' % (tag, dn)] = cv2.projectPoints(h3, np.zeros(3), np.zeros(3), K(k), np.array(d))[0].reshape(-1, 2)
        src64 = px.reshape(-1, 1, 2)
        out['fish_undist_%s' % tag] = cv2.fisheye.undistortPoints(src64, K(k), np.array(ci.FISH_D), None, R, Pn).reshape(-1, 2)
        out['fish_dist_%s' % tag] = cv2.fisheye.distortPoints(nrm.reshape(-1, 1, 2), K(k), np.array(ci.FISH_D)).reshape(-1, 2)
    out['rodrigues'] = np.stack([cv2.Rodrigues(np.array(v, np.float64))[0] for v in ci.RODRIGUES_VECS])

    path = os.path.join(HERE, 'cv2_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%d arrays, OpenCV %s)' % (path, len(out), cv2.__version__))


if __name__ == '__main__':
    main()
