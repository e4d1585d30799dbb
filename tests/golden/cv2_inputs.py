"""Inputs of the cv2 golden vectors -- shared by the generator (make_cv2_golden.py, runs where OpenCV is installed) and by the
test that replays them through the CPU oracle (tests/test_oracle_cv2_golden.py, runs anywhere).

Everything here is INTEGER arithmetic on numpy's PCG64 stream, so both sides build bit-identical images and points on any
platform; the .npz written by the generator therefore carries only cv2's OUTPUTS plus a CRC of every input as a guard.
"""
import zlib

import numpy as np

W, H = 752, 480                      # the reference's EuRoC frame (src/config.py: cam0_resolution)

# EuRoC cam0 / cam1 calibration as the reference holds it (src/config.py:93-123): intrinsics fu fv cu cv, radtan k1 k2 p1 p2
CAM0_K = [458.654, 457.296, 367.215, 248.375]
CAM0_D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]
CAM1_K = [457.587, 456.134, 379.999, 255.238]
CAM1_D = [-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05]
# equidistant (cv2.fisheye) coefficients of a typical 190-degree lens (camera_model.py:41-43, 69-70 take the same four numbers)
FISH_D = [-0.013721808247486035, 0.020727425669427896, -0.012786476702685545, 0.0025242267320687625]


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)


def _box(a, r):
    """(2r+1)^2 integer box sum with wrap-around, int64."""
    s = np.zeros_like(a)
    for d in range(-r, r + 1):
        s += np.roll(a, d, 1)
    t = np.zeros_like(a)
    for d in range(-r, r + 1):
        t += np.roll(s, d, 0)
    return t


def texture(seed, h=H + 64, w=W + 64):
    """Smooth random texture with corners: two octaves of blocky noise, box-filtered, plus dark / bright squares."""
    rng = np.random.default_rng(seed)
    coarse = np.kron(rng.integers(0, 256, ((h + 15) // 16, (w + 15) // 16), dtype=np.int64), np.ones((16, 16), np.int64))[:h, :w]
    fine = np.kron(rng.integers(0, 256, ((h + 3) // 4, (w + 3) // 4), dtype=np.int64), np.ones((4, 4), np.int64))[:h, :w]
    img = (_box(coarse, 3) // 49 * 5 + _box(fine, 1) // 9 * 3) // 8
    for _ in range(260):
        y, x = int(rng.integers(4, h - 14)), int(rng.integers(4, w - 14))
        s, v = int(rng.integers(5, 10)), int(rng.integers(0, 2)) * 215 + 20
        img[y:y + s, x:x + s] = v
    return np.clip(img, 0, 255).astype(np.uint8)


def shifted(tex, dx2, dy2, h=H, w=W):
    """Crop of the texture displaced by (dx2 / 2, dy2 / 2) pixels: half-pixel phases by 2 x 2 integer averaging."""
    t = tex.astype(np.int64)
    ox, oy = 32 + dx2 // 2, 32 + dy2 // 2
    a = t[oy:oy + h + 1, ox:ox + w + 1]
    fx, fy = dx2 & 1, dy2 & 1
    out = ((2 - fx) * (2 - fy) * a[:h, :w] + fx * (2 - fy) * a[:h, 1:w + 1] + (2 - fx) * fy * a[1:h + 1, :w] + fx * fy * a[1:h + 1, 1:w + 1] + 2) // 4
    return out.astype(np.uint8)


def frames():
    """I (frame t, cam0), J_t (frame t+1, cam0: 2.5 px right, 1.5 px down), J_s (cam1: 11 px of disparity, half a pixel up)."""
    tex = texture(20240901)
    return shifted(tex, 0, 0), shifted(tex, -5, -3), shifted(tex, 22, 1)


def small_frames(w, h, seed):
    tex = texture(seed, h + 64, w + 64)
    return shifted(tex, 0, 0, h, w), shifted(tex, -2, 1, h, w)


def points(seed, n_in, n_out, w=W, h=H, sigma8=24):
    """float32 points on a 1/8-pixel lattice: n_in inside the image, n_out up to 25 px outside; initial guesses = the points
    displaced by integer-drawn eighths (|d| <= sigma8 / 8 px); the last 10 guesses are thrown 400 px away (windows leave the image)."""
    rng = np.random.default_rng(seed)
    inside = np.stack([rng.integers(8 * 8, (w - 8) * 8, n_in), rng.integers(8 * 8, (h - 8) * 8, n_in)], 1)
    outside = np.stack([rng.integers(-25 * 8, (w + 25) * 8, n_out), rng.integers(-25 * 8, (h + 25) * 8, n_out)], 1)
    p8 = np.concatenate([inside, outside]).astype(np.int64)
    d8 = rng.integers(-sigma8, sigma8 + 1, p8.shape)
    prev = (p8.astype(np.float32) / np.float32(8))
    init = ((p8 + d8).astype(np.float32) / np.float32(8))
    if len(init) >= 10:
        init[-10:] += np.float32(400)
    return prev, init


LK_REFERENCE = dict(winSize=(15, 15), maxLevel=3, criteria=(3, 30, 0.01), flags=4, minEigThreshold=1e-4)    # config.py:37-44
LK_OTHER = [(9, 3), (21, 3), (31, 2), (16, 1), (5, 4), (15, 1), (15, 4), (29, 4), (31, 4)]                   # (win, maxLevel)
LK_SMALL = [(18, 20, 31, 2), (40, 36, 21, 2), (33, 17, 15, 1), (64, 48, 23, 1)]                               # (w, h, win, maxLevel)
FAST_THRESHOLDS = [7, 15, 40]


def fast_mask():
    """feature_adder.py:56-62 style mask: 7 x 7 holes around a lattice of points."""
    m = np.ones((H, W), np.uint8)
    rng = np.random.default_rng(77)
    for _ in range(300):
        y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
        m[max(y - 3, 0):y + 4, max(x - 3, 0):x + 4] = 0
    return m


def camera_points(n=400):
    rng = np.random.default_rng(5)
    px = np.stack([rng.integers(-40 * 16, (W + 40) * 16, n), rng.integers(-40 * 16, (H + 40) * 16, n)], 1).astype(np.float64) / 16.0
    nrm = np.stack([rng.integers(-900, 901, n), rng.integers(-600, 601, n)], 1).astype(np.float64) / 1024.0
    return px, nrm


def rectification():
    """A rotation about (1, 2, 3) / sqrt(14) by 0.05 rad built from rationals (no libm: the same doubles everywhere)."""
    c, s = 0.99875026039496628, 0.049979169270678331
    x, y, z = 0.2672612419124244, 0.53452248382484879, 0.80178372573727319
    C = 1 - c
    return np.array([[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


RODRIGUES_VECS = [[0.0, 0.0, 0.0], [1e-9, -2e-9, 1e-9], [0.01, -0.02, 0.005], [0.3, 0.2, -0.1], [1.2, -0.7, 2.1], [3.0, 0.5, -0.25]]
