"""Seeded synthetic EuRoC-like stereo + IMU streams.

A textured, slanted wall is rendered through the real EuRoC pinhole+radtan calibration of both
cameras (reference: src/config.py:93-121) from a smooth 6-DoF trajectory, so that the reference's
own gates (forward-backward LK error, |dy| < 20, epipolar test with the cam0 model) pass for true
matches, and a 200 Hz IMU stream consistent with the same trajectory is produced for the IMU
rotation prediction and for the MSCKF.  Message shapes are the reference's namedtuples
(reference: src/streaming/dataset.py:56-57,101,168-174).  Pure numpy; no GPU, no OpenCV.
"""
from collections import namedtuple

import numpy as np

imu_msg_t = namedtuple('imu_msg', ['timestamp', 'angular_velocity', 'linear_acceleration'])
img_msg_t = namedtuple('img_msg', ['timestamp', 'image'])
stereo_msg_t = namedtuple('stereo_msg', ['timestamp', 'cam0_image', 'cam1_image', 'cam0_msg', 'cam1_msg'])

W, H = 752, 480
TEX_SCALE = 100.0          # texels per metre on the wall
FRAME_RATE = 20.0
IMU_RATE = 200.0
IMU_PHASE = 0.0012         # IMU clock offset vs. camera clock [s] (unsynchronised sensors)
GRAVITY = np.array([0.0, 0.0, -9.81])


def make_texture(seed, size=(1536, 2048)):
    """Multi-octave value noise plus a few thousand random rectangles: enough corner structure for
    FAST@15 to find >= 2000 corners per 752x480 view and enough gradient for LK everywhere."""
    rng = np.random.default_rng(seed)
    th, tw = size
    tex = np.full((th, tw), 110.0, np.float32)
    for cell, amp in ((128, 30.0), (32, 22.0), (8, 14.0)):
        g = rng.uniform(-1.0, 1.0, (th // cell + 2, tw // cell + 2)).astype(np.float32)
        yy = np.arange(th, dtype=np.float32) / cell
        xx = np.arange(tw, dtype=np.float32) / cell
        y0 = yy.astype(np.int32); x0 = xx.astype(np.int32)
        fy = (yy - y0)[:, None]; fx = (xx - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        tex += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    n_rect = (th * tw) // 450
    ys = rng.integers(0, th - 16, n_rect); xs = rng.integers(0, tw - 16, n_rect)
    hs = rng.integers(5, 14, n_rect); ws = rng.integers(5, 14, n_rect)
    dv = rng.choice(np.array([-70., -50., 50., 70.], np.float32), n_rect)
    for y, x, h, w, v in zip(ys, xs, hs, ws, dv):
        tex[y:y + h, x:x + w] += v
    return np.clip(tex, 0, 255).astype(np.float32)


def _so3_exp(v):
    th = np.linalg.norm(v)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / (th * th) * (K @ K)


def _so3_log(R):
    c = min(1.0, max(-1.0, (np.trace(R) - 1) / 2))
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-9:
        return w / 2
    return w * th / (2 * np.sin(th))


def _undistorted_rays(intr, dist, iters=25):
    """Normalised undistorted coordinates of every pixel centre (radtan inverse, fixed point)."""
    fx, fy, cx, cy = intr
    k1, k2, p1, p2 = dist
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    x0 = (u - cx) / fx; y0 = (v - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        ic = 1.0 / (1 + (k2 * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * ic; y = (y0 - dy) * ic
    return np.stack([x, y, np.ones_like(x)], axis=-1)


class SyntheticStream(object):
    """One seeded stream: `frames()` yields stereo_msg, `imu` is the full list of imu_msg.

    Frame k has timestamp t0 + k / 20; the IMU stream starts `lead_in` seconds earlier (the filter
    needs 200 samples before its first update, reference: src/msckf.py:172-175) and the platform is
    at rest until frame 0.
    """

    def __init__(self, config, seed=0, n_frames=20, t0=100.0, lead_in=1.0, motion_scale=1.0,
                 pixel_noise=1.0, texture=None, render=True, rest=0.0, tex_offset=(0.0, 0.0)):
        self.config = config
        self.seed = int(seed)
        self.n_frames = int(n_frames)
        self.t0 = float(t0)
        self.lead_in = float(lead_in)
        self.rest = float(rest)                 # seconds of standstill after frame 0 (EuRoC sequences start at rest on the ground)
        self.motion_scale = float(motion_scale)
        self.tex_offset = (float(tex_offset[0]), float(tex_offset[1]))      # texels: which part of the (wrapping) texture the wall shows
        self.pixel_noise = float(pixel_noise)
        self.rng = np.random.default_rng(0xA1B0 + self.seed)
        self.tex = (make_texture(0xA1B0 + self.seed) if texture is None else texture) if render else None

        # camera <-> imu geometry (T_imu_cam*: imu-frame vector -> camera frame)
        self.T_c0_i = np.linalg.inv(config.T_imu_cam0)      # cam0 -> imu
        self.T_c1_i = np.linalg.inv(config.T_imu_cam1)
        # world: z up; at rest cam0 looks along world +x, image-right = -y, image-down = -z
        R_c0_w = np.array([[0., 0., 1.], [-1., 0., 0.], [0., -1., 0.]])
        self.R_i_w0 = R_c0_w @ config.T_imu_cam0[:3, :3]     # imu -> world at rest
        # the wall: n . X = d, slanted so depth spans ~4..7 m across the view
        n = np.array([1.0, 0.25, 0.10]); self.wall_n = n / np.linalg.norm(n)
        self.wall_d = 5.2
        e1 = np.cross(self.wall_n, [0., 0., 1.]); self.wall_e1 = e1 / np.linalg.norm(e1)
        self.wall_e2 = np.cross(self.wall_n, self.wall_e1)
        if render:
            self.rays0 = _undistorted_rays(config.cam0_intrinsics, config.cam0_distortion_coeffs)
            self.rays1 = _undistorted_rays(config.cam1_intrinsics, config.cam1_distortion_coeffs)
        self.imu = self._make_imu()

    # ---- trajectory ------------------------------------------------------------------------
    def _s(self, t):
        """Smooth start: s(0)=0, s'(0)=0, s(t)->t-1; t is seconds since frame 0 (rest for t<0)."""
        t = max(0.0, t)
        return t * t / (t + 1.0)

    def position(self, t):
        s = self._s(t - self.t0 - self.rest) * 1.0
        m = self.motion_scale
        return m * np.array([0.5 * np.sin(0.5 * s), 0.3 * np.sin(0.7 * s), 0.1 * np.sin(0.3 * s)])

    def R_i_w(self, t):
        s = self._s(t - self.t0 - self.rest)
        m = self.motion_scale
        th = m * np.array([0.05 * np.sin(0.4 * s), 0.04 * np.sin(0.6 * s), 0.06 * np.sin(0.5 * s)])
        return self.R_i_w0 @ _so3_exp(th)

    def _make_imu(self):
        dt = 1.0 / IMU_RATE
        t_begin = self.t0 - self.lead_in
        t_end = self.t0 + (self.n_frames - 1) / FRAME_RATE + 2 * dt
        n = int(np.floor((t_end - t_begin) / dt)) + 1
        out = []
        h = 1e-4
        for k in range(n):
            t = t_begin + IMU_PHASE + k * dt
            R = self.R_i_w(t)
            w = _so3_log(self.R_i_w(t - h).T @ self.R_i_w(t + h)) / (2 * h)      # body rate, imu frame
            a = (self.position(t + h) - 2 * self.position(t) + self.position(t - h)) / (h * h)
            f = R.T @ (a - GRAVITY)
            w = w + self.rng.normal(0, 0.005, 3)
            f = f + self.rng.normal(0, 0.05, 3)
            out.append(imu_msg_t(t, w, f))
        return out

    # ---- rendering -------------------------------------------------------------------------
    def _render(self, rays, R_c_w, c_w, noise_rng):
        r = rays @ R_c_w.T                                        # ray directions in world
        s = (self.wall_d - self.wall_n @ c_w) / (r @ self.wall_n)
        X = c_w + r * s[..., None]
        u = (X @ self.wall_e1) * TEX_SCALE + self.tex.shape[1] / 2 + self.tex_offset[0]
        v = (X @ self.wall_e2) * TEX_SCALE + self.tex.shape[0] / 2 + self.tex_offset[1]
        th, tw = self.tex.shape
        u0 = np.floor(u); v0 = np.floor(v)
        fu = (u - u0).astype(np.float32); fv = (v - v0).astype(np.float32)
        ui = u0.astype(np.int64) % tw; vi = v0.astype(np.int64) % th
        ui1 = (ui + 1) % tw; vi1 = (vi + 1) % th
        t = self.tex
        img = (t[vi, ui] * (1 - fu) + t[vi, ui1] * fu) * (1 - fv) + (t[vi1, ui] * (1 - fu) + t[vi1, ui1] * fu) * fv
        if self.pixel_noise > 0:
            img = img + noise_rng.normal(0, self.pixel_noise, img.shape).astype(np.float32)
        return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))

    def frame_time(self, k):
        return self.t0 + k / FRAME_RATE

    def cam0_pose(self, t):
        """(R_c0_w, c0_w): cam0 -> world rotation and cam0 centre in world."""
        R_i_w, p = self.R_i_w(t), self.position(t)
        return R_i_w @ self.T_c0_i[:3, :3], p + R_i_w @ self.T_c0_i[:3, 3]

    def frame(self, k):
        t = self.frame_time(k)
        R_i_w, p = self.R_i_w(t), self.position(t)
        nrng = np.random.default_rng((0xA1B0 + self.seed) * 100003 + k)
        img0 = self._render(self.rays0, R_i_w @ self.T_c0_i[:3, :3], p + R_i_w @ self.T_c0_i[:3, 3], nrng)
        img1 = self._render(self.rays1, R_i_w @ self.T_c1_i[:3, :3], p + R_i_w @ self.T_c1_i[:3, 3], nrng)
        m0, m1 = img_msg_t(t, img0), img_msg_t(t, img1)
        return stereo_msg_t(t, img0, img1, m0, m1)

    def frames(self):
        for k in range(self.n_frames):
            yield self.frame(k)

    # ---- the same renderer in torch (bench data generation: many distinct streams rendered on the GPU) ----------
    def torch_state(self, device):
        """Device copies of what `frame_torch` needs (texture, per-pixel rays); build once per stream (textures may be shared)."""
        import torch
        return dict(tex=torch.from_numpy(self.tex).to(device), rays0=torch.from_numpy(self.rays0).to(device),
                    rays1=torch.from_numpy(self.rays1).to(device))

    def frame_torch(self, k, state, generator=None):
        """(cam0, cam1) uint8 tensors of frame k: `_render` restated with torch ops (fp64 geometry, fp32 bilinear lookup).  The
        sensor noise comes from torch's generator, not numpy's, so the pixels are NOT those of `frame(k)`: same scene, same
        statistics.  Streams that must equal the CPU path's input are rendered with `frame`."""
        import torch
        t = self.frame_time(k)
        R_i_w, p = self.R_i_w(t), self.position(t)
        out = []
        tex = state['tex']
        th, tw = tex.shape
        dev = tex.device
        for rays, T in ((state['rays0'], self.T_c0_i), (state['rays1'], self.T_c1_i)):
            R_c_w = torch.from_numpy(np.ascontiguousarray(R_i_w @ T[:3, :3])).to(dev)
            c_w = torch.from_numpy(p + R_i_w @ T[:3, 3]).to(dev)
            n = torch.from_numpy(self.wall_n).to(dev)
            e1 = torch.from_numpy(self.wall_e1).to(dev); e2 = torch.from_numpy(self.wall_e2).to(dev)
            r = rays @ R_c_w.T
            s_ = (self.wall_d - torch.dot(n, c_w)) / (r @ n)
            X = c_w + r * s_[..., None]
            u = (X @ e1) * TEX_SCALE + tw / 2 + self.tex_offset[0]
            v = (X @ e2) * TEX_SCALE + th / 2 + self.tex_offset[1]
            u0 = torch.floor(u); v0 = torch.floor(v)
            fu = (u - u0).to(torch.float32); fv = (v - v0).to(torch.float32)
            ui = torch.remainder(u0.to(torch.int64), tw); vi = torch.remainder(v0.to(torch.int64), th)
            ui1 = torch.remainder(ui + 1, tw); vi1 = torch.remainder(vi + 1, th)
            img = (tex[vi, ui] * (1 - fu) + tex[vi, ui1] * fu) * (1 - fv) + (tex[vi1, ui] * (1 - fu) + tex[vi1, ui1] * fu) * fv
            if self.pixel_noise > 0:
                img = img + torch.randn(img.shape, generator=generator, device=dev, dtype=torch.float32) * self.pixel_noise
            out.append(torch.clamp(torch.round(img), 0, 255).to(torch.uint8))
        return out[0], out[1]


def replay(stream, imu_sinks, on_frame):
    """Deterministic sequential replay: before each stereo frame at time t
    deliver every IMU message with timestamp <= t to each sink (order of vio.py:43-44), then call
    on_frame(stereo_msg)."""
    it = iter(stream.imu)
    pending = next(it, None)
    for k in range(stream.n_frames):
        msg = stream.frame(k)
        while pending is not None and pending.timestamp <= msg.timestamp:
            for sink in imu_sinks:
                sink(pending)
            pending = next(it, None)
        on_frame(msg)


feature_msg_t = namedtuple('feature_msg', ['timestamp', 'features'])


class _Meas(object):
    __slots__ = ('id', 'u0', 'v0', 'u1', 'v1')


class SyntheticFeatureStream(object):
    """Stereo feature measurements (normalised coordinates) for the filter half, without images:
    random landmarks in view of the moving stereo rig of a SyntheticStream, track length
    U{3..24} frames, pixel noise sigma ~0.5 px (MSCKF microbench recipe).
    `frames()` yields feature_msg(timestamp, [FeatureMeasurement-like]); `imu` as SyntheticStream."""

    def __init__(self, config, seed=0, n_frames=100, n_features=100, pixel_sigma=0.5, motion_scale=1.0, t0=100.0,
                 outlier_rate=0.0, outlier_px=12.0):
        self.base = SyntheticStream(config, seed=seed, n_frames=n_frames, motion_scale=motion_scale, t0=t0, render=False)
        self.config = config
        self.n_frames = n_frames
        self.n_features = n_features
        self.imu = self.base.imu
        self.rng = np.random.default_rng(0xFEA7 + seed)
        self.sigma = pixel_sigma / float(config.cam0_intrinsics[0])
        # gross mismatches (so that the chi-square gate rejects something); own generator: rate 0 leaves the stream as it was
        self.outlier_rate = outlier_rate
        self.outlier_sigma = outlier_px / float(config.cam0_intrinsics[0])
        self.rng_out = np.random.default_rng(0x0071 + seed)
        self.T_c0_i = self.base.T_c0_i
        self.T_c1_i = self.base.T_c1_i
        self._next_id = 0
        self._tracks = []            # [id, p_world, frames_left]
        self._msgs = [self._make(k) for k in range(n_frames)]

    def _spawn(self, R_c0_w, c0_w):
        depth = self.rng.uniform(2.0, 8.0)
        pc = np.array([self.rng.uniform(-0.6, 0.6), self.rng.uniform(-0.4, 0.4), 1.0]) * depth
        tr = [self._next_id, R_c0_w @ pc + c0_w, int(self.rng.integers(3, 25))]
        self._next_id += 1
        return tr

    def _make(self, k):
        t = self.base.frame_time(k)
        R_i_w, p = self.base.R_i_w(t), self.base.position(t)
        R0, c0 = R_i_w @ self.T_c0_i[:3, :3], p + R_i_w @ self.T_c0_i[:3, 3]
        R1, c1 = R_i_w @ self.T_c1_i[:3, :3], p + R_i_w @ self.T_c1_i[:3, 3]
        self._tracks = [tr for tr in self._tracks if tr[2] > 0]
        while len(self._tracks) < self.n_features:
            self._tracks.append(self._spawn(R0, c0))
        feats = []
        for tr in self._tracks:
            q0 = R0.T @ (tr[1] - c0)
            q1 = R1.T @ (tr[1] - c1)
            tr[2] -= 1
            if q0[2] < 0.5 or q1[2] < 0.5 or abs(q0[0] / q0[2]) > 0.75 or abs(q0[1] / q0[2]) > 0.5:
                tr[2] = 0
                continue
            n = self.rng.normal(0, self.sigma, 4)
            if self.outlier_rate > 0 and self.rng_out.random() < self.outlier_rate:
                n = n + self.rng_out.normal(0, self.outlier_sigma, 4)
            m = _Meas()
            m.id = tr[0]
            m.u0, m.v0 = q0[0] / q0[2] + n[0], q0[1] / q0[2] + n[1]
            m.u1, m.v1 = q1[0] / q1[2] + n[2], q1[1] / q1[2] + n[3]
            feats.append(m)
        return feature_msg_t(t, feats)

    def frame(self, k):
        return self._msgs[k]

    def frames(self):
        return iter(self._msgs)

    def truth(self, k):
        """(R_i_w, p_w) of the IMU at frame k."""
        t = self.base.frame_time(k)
        return self.base.R_i_w(t), self.base.position(t)


def replay_features(fstream, imu_sinks, on_features):
    """Deterministic replay for the filter: IMU with timestamp <= t first, then the feature message."""
    it = iter(fstream.imu)
    pending = next(it, None)
    for k in range(fstream.n_frames):
        msg = fstream.frame(k)
        while pending is not None and pending.timestamp <= msg.timestamp:
            for sink in imu_sinks:
                sink(pending)
            pending = next(it, None)
        on_features(msg)
