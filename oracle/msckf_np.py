"""oracle/msckf_np.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's stereo MSCKF back-end (src/msckf.py, src/feature/*,
src/utils.py), written independently but following the reference step by step, including the
quirks of SURVEY.md Appendix A.8-A.17.  Each function cites the reference lines it follows.

PARITY STATUS: pinned.  The reference filter is importable in the build container (numpy + scipy
only); tests/golden/make_msckf_golden.py imports it from /root/reference/src, drives it and this
restatement with the same inputs and commits the reference's outputs as tests/golden/msckf_*.npz.
tests/test_oracle_msckf.py checks this file against those vectors.
"""
from collections import namedtuple

import numpy as np
from scipy.stats import chi2

vio_result_t = namedtuple('vio_result', ['timestamp', 'pose', 'velocity', 'cam0_pose'])


# ---- quaternion / SE3 helpers (utils.py:2-141; JPL convention [x y z w]) ----------------------
def skew(v):
    x, y, z = v
    return np.array([[0., -z, y], [z, 0., -x], [-y, x, 0.]])


def to_rotation(q):
    """utils.py:12-23 (normalises its input, A.15)."""
    q = q / np.linalg.norm(q)
    v, w = q[:3], q[3]
    return (2 * w * w - 1) * np.identity(3) - 2 * w * skew(v) + 2 * v[:, None] * v


def to_quaternion(R):
    """utils.py:25-47."""
    if R[2, 2] < 0:
        if R[0, 0] > R[1, 1]:
            q = [1 + R[0, 0] - R[1, 1] - R[2, 2], R[0, 1] + R[1, 0], R[2, 0] + R[0, 2], R[1, 2] - R[2, 1]]
        else:
            q = [R[0, 1] + R[1, 0], 1 - R[0, 0] + R[1, 1] - R[2, 2], R[2, 1] + R[1, 2], R[2, 0] - R[0, 2]]
    else:
        if R[0, 0] < -R[1, 1]:
            q = [R[0, 2] + R[2, 0], R[2, 1] + R[1, 2], 1 - R[0, 0] - R[1, 1] + R[2, 2], R[0, 1] - R[1, 0]]
        else:
            q = [R[1, 2] - R[2, 1], R[2, 0] - R[0, 2], R[0, 1] - R[1, 0], 1 + R[0, 0] + R[1, 1] + R[2, 2]]
    q = np.array(q)
    return q / np.linalg.norm(q)


def quaternion_multiplication(q1, q2):
    """utils.py:61-76 (normalises inputs and output)."""
    q1 = q1 / np.linalg.norm(q1)
    q2 = q2 / np.linalg.norm(q2)
    L = np.array([[q1[3], q1[2], -q1[1], q1[0]],
                  [-q1[2], q1[3], q1[0], q1[1]],
                  [q1[1], -q1[0], q1[3], q1[2]],
                  [-q1[0], -q1[1], -q1[2], q1[3]]])
    q = L @ q2
    return q / np.linalg.norm(q)


def small_angle_quaternion(dtheta):
    """utils.py:79-93."""
    dq = dtheta / 2.
    n2 = dq @ dq
    if n2 <= 1:
        return np.array([*dq, np.sqrt(1 - n2)])
    q = np.array([*dq, 1.])
    return q / np.sqrt(1 + n2)


def from_two_vectors(v0, v1):
    """utils.py:96-120 (Hamilton rotation v0->v1, returned as JPL)."""
    v0 = v0 / np.linalg.norm(v0)
    v1 = v1 / np.linalg.norm(v1)
    d = v0 @ v1
    if d < -0.999999:
        axis = np.cross([1, 0, 0], v0)
        if np.linalg.norm(axis) < 0.000001:
            axis = np.cross([0, 1, 0], v0)
        q = np.array([*axis, 0.])
    elif d > 0.999999:
        q = np.array([0., 0., 0., 1.])
    else:
        s = np.sqrt((1 + d) * 2)
        q = np.array([*(np.cross(v0, v1) / s), 0.5 * s])
    q = q / np.linalg.norm(q)
    return np.array([*-q[:3], q[3]])


class Iso(object):
    """Isometry3d (utils.py:124-141)."""
    __slots__ = ('R', 't')

    def __init__(self, R, t):
        self.R = R
        self.t = t

    def inverse(self):
        return Iso(self.R.T, -self.R.T @ self.t)

    def __mul__(self, o):
        return Iso(self.R @ o.R, self.R @ o.t + self.t)


# ---- feature triangulation (feature/*.py) -----------------------------------------------------
class OFeature(object):
    """feature.Feature (feature/__init__.py:7-16, base_feature.py:3-13)."""

    def __init__(self, fid, opt):
        self.id = fid
        self.observations = dict()
        self.position = np.zeros(3)
        self.is_initialized = False
        self.opt = opt


def tri_cost(R, t, x, z):
    """feature_observation.py:4-12."""
    h = R @ np.array([x[0], x[1], 1.0]) + x[2] * t
    d = h[:2] / h[2] - z
    return (d ** 2).sum()


def tri_jacobian(R, t, x, z, huber_eps):
    """feature_observation.py:14-39."""
    h = R @ np.array([x[0], x[1], 1.0]) + x[2] * t
    Wm = np.zeros((3, 3))
    Wm[:, :2] = R[:, :2]
    Wm[:, 2] = t
    J = np.zeros((2, 3))
    J[0] = Wm[0] / h[2] - Wm[2] * h[0] / (h[2] * h[2])
    J[1] = Wm[1] / h[2] - Wm[2] * h[1] / (h[2] * h[2])
    r = np.array([h[0] / h[2], h[1] / h[2]]) - z
    e = np.linalg.norm(r)
    w = 1.0 if e <= huber_eps else huber_eps / (2 * e)
    return J, r, w


def initialize_position(feat, cam_states, R_cam0_cam1, t_cam0_cam1):
    """feature_position_initializer.py:6-76 (+ feature_depth_estimator.py:4-14).  Note the inner
    iteration counter is never reset between outer iterations (A.11)."""
    opt = feat.opt
    T10 = Iso(R_cam0_cam1, t_cam0_cam1).inverse()
    poses, meas = [], []
    for cid, m in feat.observations.items():
        if cid not in cam_states:
            continue
        meas.extend([m[:2], m[2:]])
        c0 = Iso(to_rotation(cam_states[cid].orientation).T, cam_states[cid].position)
        poses.extend([c0, c0 * T10])
    T_c0_w = poses[0]
    poses = [p.inverse() * T_c0_w for p in poses]
    # two-view initial guess
    m = poses[1].R @ np.array([*meas[0], 1.0])
    a = m[:2] - meas[1] * m[2]
    b = meas[1] * poses[1].t[2] - poses[1].t[:2]
    depth = a @ b / (a @ a)
    p0 = np.array([*meas[0], 1.0]) * depth
    x = np.array([p0[0], p0[1], 1.0]) / p0[2]
    lam = opt.initial_damping
    outer = inner = 0
    delta_norm = float('inf')
    cost = sum(tri_cost(p.R, p.t, x, z) for p, z in zip(poses, meas))
    while outer < opt.outer_loop_max_iteration and delta_norm > opt.estimation_precision:
        A = np.zeros((3, 3))
        bv = np.zeros(3)
        for p, z in zip(poses, meas):
            J, r, w = tri_jacobian(p.R, p.t, x, z, opt.huber_epsilon)
            if w == 1.0:
                A += J.T @ J
                bv += J.T @ r
            else:
                A += w * w * J.T @ J
                bv += w * w * J.T @ r
        reduced = False
        while inner < opt.inner_loop_max_iteration and not reduced:
            delta = np.linalg.solve(A + lam * np.eye(3), bv)
            xn = x - delta
            delta_norm = np.linalg.norm(delta)
            cn = sum(tri_cost(p.R, p.t, xn, z) for p, z in zip(poses, meas))
            if cn < cost:
                reduced = True
                x, cost = xn, cn
                lam = max(lam / 10., 1e-10)
            else:
                lam = min(lam * 10., 1e12)
            inner += 1
        outer += 1
    pf = np.array([x[0], x[1], 1.0]) / x[2]
    valid = all((p.R @ pf + p.t)[2] > 0 for p in poses)
    feat.position = T_c0_w.R @ pf + T_c0_w.t
    feat.is_initialized = valid
    return valid


# ---- filter state -------------------------------------------------------------------------------
class OIMUState(object):
    """IMUState (msckf.py:18-58); gravity / T_imu_body are per-filter here, not class statics."""

    def __init__(self):
        self.id = None
        self.timestamp = None
        self.orientation = np.array([0., 0., 0., 1.])
        self.position = np.zeros(3)
        self.velocity = np.zeros(3)
        self.gyro_bias = np.zeros(3)
        self.acc_bias = np.zeros(3)
        self.orientation_null = np.array([0., 0., 0., 1.])
        self.position_null = np.zeros(3)
        self.velocity_null = np.zeros(3)
        self.R_imu_cam0 = np.identity(3)
        self.t_cam0_imu = np.zeros(3)


class OCAMState(object):
    """CAMState (msckf.py:61-77)."""

    def __init__(self, cid):
        self.id = cid
        self.timestamp = None
        self.orientation = np.array([0., 0., 0., 1.])
        self.position = np.zeros(3)
        self.orientation_null = np.array([0., 0., 0., 1.])
        self.position_null = np.zeros(3)


class OracleMSCKF(object):
    """MSCKF (msckf.py:95-867) restated.  next_imu_id starts at 0 per filter instance (the reference
    keeps it in the class attribute IMUState.next_id, msckf.py:20,270-271)."""

    def __init__(self, config, next_imu_id=0):
        self.config = config
        self.opt = config.optimization_config
        self.imu_msg_buffer = []
        self.imu_state = OIMUState()
        self.cam_states = dict()
        self.map_server = dict()
        self.chi2_table = {i: chi2.ppf(0.05, i) for i in range(1, 100)}           # msckf.py:111-113 (A.9)
        self.imu_state.velocity = config.velocity
        self.state_cov = self._initial_cov()
        Qc = np.identity(12)
        Qc[:3, :3] *= config.gyro_noise
        Qc[3:6, 3:6] *= config.gyro_bias_noise
        Qc[6:9, 6:9] *= config.acc_noise
        Qc[9:, 9:] *= config.acc_bias_noise
        self.Qc = Qc
        self.gravity = config.gravity
        T_cam0_imu = np.linalg.inv(config.T_imu_cam0)
        self.imu_state.R_imu_cam0 = T_cam0_imu[:3, :3].T
        self.imu_state.t_cam0_imu = T_cam0_imu[:3, 3]          # a view into a private matrix, mutated in place later
        self.R_cam0_cam1 = config.T_cn_cnm1[:3, :3]
        self.t_cam0_cam1 = config.T_cn_cnm1[:3, 3]
        self.T_imu_body = Iso(config.T_imu_body[:3, :3], config.T_imu_body[:3, 3])
        self.next_imu_id = next_imu_id
        self.tracking_rate = None
        self.is_gravity_set = False
        self.is_first_img = True
        self.trajectory = []            # what _write_state appends to the txt file (msckf.py:152-160)
        self.debug = {}

    def _initial_cov(self):
        """reset_state_cov (msckf.py:788-798)."""
        c = self.config
        P = np.zeros((21, 21))
        P[3:6, 3:6] = c.gyro_bias_cov * np.identity(3)
        P[6:9, 6:9] = c.velocity_cov * np.identity(3)
        P[9:12, 9:12] = c.acc_bias_cov * np.identity(3)
        P[15:18, 15:18] = c.extrinsic_rotation_cov * np.identity(3)
        P[18:21, 18:21] = c.extrinsic_translation_cov * np.identity(3)
        return P

    # ---- callbacks -------------------------------------------------------------------------
    def imu_callback(self, msg):
        """msckf.py:162-175."""
        self.imu_msg_buffer.append(msg)
        if not self.is_gravity_set and len(self.imu_msg_buffer) >= 200:
            self.initialize_gravity_and_bias()
            self.is_gravity_set = True

    def initialize_gravity_and_bias(self):
        """msckf.py:230-249 (A.16)."""
        sw = np.zeros(3)
        sa = np.zeros(3)
        for m in self.imu_msg_buffer:
            sw += m.angular_velocity
            sa += m.linear_acceleration
        n = len(self.imu_msg_buffer)
        self.imu_state.gyro_bias = sw / n
        g_imu = sa / n
        self.gravity = np.array([0., 0., -np.linalg.norm(g_imu)])
        self.imu_state.orientation = from_two_vectors(-self.gravity, g_imu)

    def feature_callback(self, feature_msg):
        """msckf.py:177-228."""
        if not self.is_gravity_set:
            return None
        if self.is_first_img:
            self.is_first_img = False
            self.imu_state.timestamp = feature_msg.timestamp
        self.batch_imu_processing(feature_msg.timestamp)
        self.state_augmentation(feature_msg.timestamp)
        self.add_feature_observations(feature_msg)
        self.remove_lost_features()
        self.prune_cam_state_buffer()
        try:
            return self.publish(feature_msg.timestamp)
        finally:
            self.online_reset()

    # ---- propagation -----------------------------------------------------------------------
    def batch_imu_processing(self, time_bound):
        """msckf.py:251-273."""
        used = 0
        for m in self.imu_msg_buffer:
            if m.timestamp < self.imu_state.timestamp:
                used += 1
                continue
            if m.timestamp > time_bound:
                break
            self.process_model(m.timestamp, m.angular_velocity, m.linear_acceleration)
            used += 1
            self.imu_state.timestamp = m.timestamp
        self.imu_state.id = self.next_imu_id
        self.next_imu_id += 1
        self.imu_msg_buffer = self.imu_msg_buffer[used:]

    def process_model(self, time, m_gyro, m_acc):
        """msckf.py:275-339 (A.12)."""
        s = self.imu_state
        dt = time - s.timestamp
        gyro = m_gyro - s.gyro_bias
        acc = m_acc - s.acc_bias
        F = np.zeros((21, 21))
        G = np.zeros((21, 12))
        R_w_i = to_rotation(s.orientation)
        F[:3, :3] = -skew(gyro)
        F[:3, 3:6] = -np.identity(3)
        F[6:9, :3] = -R_w_i.T @ skew(acc)
        F[6:9, 9:12] = -R_w_i.T
        F[12:15, 6:9] = np.identity(3)
        G[:3, :3] = -np.identity(3)
        G[3:6, 3:6] = np.identity(3)
        G[6:9, 6:9] = -R_w_i.T
        G[9:12, 9:12] = np.identity(3)
        Fdt = F * dt
        Fdt2 = Fdt @ Fdt
        Fdt3 = Fdt2 @ Fdt
        Phi = np.identity(21) + Fdt + Fdt2 / 2. + Fdt3 / 6.
        self.predict_new_state(dt, gyro, acc)
        R_kk_1 = to_rotation(s.orientation_null)
        Phi[:3, :3] = to_rotation(s.orientation) @ R_kk_1.T
        u = R_kk_1 @ self.gravity
        sv = u / (u @ u)
        A1 = Phi[6:9, :3]
        w1 = skew(s.velocity_null - s.velocity) @ self.gravity
        Phi[6:9, :3] = A1 - (A1 @ u - w1)[:, None] * sv
        A2 = Phi[12:15, :3]
        w2 = skew(dt * s.velocity_null + s.position_null - s.position) @ self.gravity
        Phi[12:15, :3] = A2 - (A2 @ u - w2)[:, None] * sv
        Q = Phi @ G @ self.Qc @ G.T @ Phi.T * dt
        P = self.state_cov
        P[:21, :21] = Phi @ P[:21, :21] @ Phi.T + Q
        if len(self.cam_states) > 0:
            P[:21, 21:] = Phi @ P[:21, 21:]
            P[21:, :21] = P[21:, :21] @ Phi.T
        self.state_cov = (P + P.T) / 2.
        s.orientation_null = s.orientation
        s.position_null = s.position
        s.velocity_null = s.velocity

    def predict_new_state(self, dt, gyro, acc):
        """msckf.py:341-388 (A.13: k2 and k3 share the half-step rotation)."""
        s = self.imu_state
        gn = np.linalg.norm(gyro)
        Om = np.zeros((4, 4))
        Om[:3, :3] = -skew(gyro)
        Om[:3, 3] = gyro
        Om[3, :3] = -gyro
        q, v, p = s.orientation, s.velocity, s.position
        if gn > 1e-5:
            dq_dt = (np.cos(gn * dt * 0.5) * np.identity(4) + np.sin(gn * dt * 0.5) / gn * Om) @ q
            dq_dt2 = (np.cos(gn * dt * 0.25) * np.identity(4) + np.sin(gn * dt * 0.25) / gn * Om) @ q
        else:
            dq_dt = np.cos(gn * dt * 0.5) * (np.identity(4) + Om * dt * 0.5) @ q
            dq_dt2 = np.cos(gn * dt * 0.25) * (np.identity(4) + Om * dt * 0.25) @ q
        Rt = to_rotation(dq_dt).T
        Rt2 = to_rotation(dq_dt2).T
        g = self.gravity
        k1v = to_rotation(q).T @ acc + g
        k1p = v
        k2v = Rt2 @ acc + g
        k2p = v + k1v * dt / 2.
        k3v = Rt2 @ acc + g
        k3p = v + k2v * dt / 2
        k4v = Rt @ acc + g
        k4p = v + k3v * dt
        s.orientation = dq_dt / np.linalg.norm(dq_dt)
        s.velocity = v + (k1v + 2 * k2v + 2 * k3v + k4v) * dt / 6.
        s.position = p + (k1p + 2 * k2p + 2 * k3p + k4p) * dt / 6.

    def state_augmentation(self, time):
        """msckf.py:390-423 (A.14)."""
        s = self.imu_state
        R_i_c, t_c_i = s.R_imu_cam0, s.t_cam0_imu
        R_w_i = to_rotation(s.orientation)
        cs = OCAMState(s.id)
        cs.timestamp = time
        cs.orientation = to_quaternion(R_i_c @ R_w_i)
        cs.position = s.position + R_w_i.T @ t_c_i
        cs.orientation_null = cs.orientation
        cs.position_null = cs.position
        self.cam_states[s.id] = cs
        J = np.zeros((6, 21))
        J[:3, :3] = R_i_c
        J[:3, 15:18] = np.identity(3)
        J[3:6, :3] = skew(R_w_i.T @ t_c_i)
        J[3:6, 12:15] = np.identity(3)
        J[3:6, 18:21] = np.identity(3)
        n = self.state_cov.shape[0]
        P = np.zeros((n + 6, n + 6))
        P[:n, :n] = self.state_cov
        P[n:, :n] = J @ P[:21, :n]
        P[:n, n:] = P[n:, :n].T
        P[n:, n:] = J @ P[:21, :21] @ J.T
        self.state_cov = (P + P.T) / 2.

    def add_feature_observations(self, feature_msg):
        """msckf.py:425-441."""
        sid = self.imu_state.id
        n_before = len(self.map_server)
        tracked = 0
        for f in feature_msg.features:
            z = np.array([f.u0, f.v0, f.u1, f.v1])
            if f.id not in self.map_server:
                mf = OFeature(f.id, self.opt)
                mf.observations[sid] = z
                self.map_server[f.id] = mf
            else:
                self.map_server[f.id].observations[sid] = z
                tracked += 1
        self.tracking_rate = tracked / (n_before + 1e-5)

    # ---- measurement model -------------------------------------------------------------------
    def measurement_jacobian(self, cam_id, fid):
        """msckf.py:443-507 (observability-constrained projection with the null-space states)."""
        cs = self.cam_states[cam_id]
        feat = self.map_server[fid]
        R_w_c0 = to_rotation(cs.orientation)
        t_c0_w = cs.position
        R_w_c1 = self.R_cam0_cam1 @ R_w_c0
        t_c1_w = t_c0_w - R_w_c1.T @ self.t_cam0_cam1
        p_w = feat.position
        z = feat.observations[cam_id]
        p0 = R_w_c0 @ (p_w - t_c0_w)
        p1 = R_w_c1 @ (p_w - t_c1_w)
        dz0 = np.zeros((4, 3))
        dz0[0, 0] = 1 / p0[2]
        dz0[1, 1] = 1 / p0[2]
        dz0[0, 2] = -p0[0] / (p0[2] * p0[2])
        dz0[1, 2] = -p0[1] / (p0[2] * p0[2])
        dz1 = np.zeros((4, 3))
        dz1[2, 0] = 1 / p1[2]
        dz1[3, 1] = 1 / p1[2]
        dz1[2, 2] = -p1[0] / (p1[2] * p1[2])
        dz1[3, 2] = -p1[1] / (p1[2] * p1[2])
        dp0 = np.zeros((3, 6))
        dp0[:, :3] = skew(p0)
        dp0[:, 3:] = -R_w_c0
        dp1 = np.zeros((3, 6))
        dp1[:, :3] = self.R_cam0_cam1 @ skew(p0)
        dp1[:, 3:] = -R_w_c1
        A = dz0 @ dp0 + dz1 @ dp1
        u = np.zeros(6)
        u[:3] = to_rotation(cs.orientation_null) @ self.gravity
        u[3:] = skew(p_w - cs.position_null) @ self.gravity
        H_x = A - (A @ u)[:, None] * u / (u @ u)
        H_f = -H_x[:4, 3:6]
        r = z - np.array([*p0[:2] / p0[2], *p1[:2] / p1[2]])
        return H_x, H_f, r

    def feature_jacobian(self, fid, cam_ids):
        """msckf.py:509-546: stack, project onto the left null space of H_f (full SVD)."""
        feat = self.map_server[fid]
        valid = [c for c in cam_ids if c in feat.observations]
        rows = 4 * len(valid)
        keys = list(self.cam_states.keys())
        H_xj = np.zeros((rows, 21 + len(keys) * 6))
        H_fj = np.zeros((rows, 3))
        r_j = np.zeros(rows)
        for k, cid in enumerate(valid):
            Hx, Hf, r = self.measurement_jacobian(cid, fid)
            idx = keys.index(cid)
            H_xj[4 * k:4 * k + 4, 21 + 6 * idx:27 + 6 * idx] = Hx
            H_fj[4 * k:4 * k + 4] = Hf
            r_j[4 * k:4 * k + 4] = r
        U, _, _ = np.linalg.svd(H_fj)
        A = U[:, 3:]
        return A.T @ H_xj, A.T @ r_j

    def gating_test(self, H, r, dof):
        """msckf.py:604-612."""
        S = H @ self.state_cov @ H.T + self.config.observation_noise * np.identity(len(H))
        gamma = r @ np.linalg.solve(S, r)
        self.debug.setdefault('gamma', []).append(gamma)
        return gamma < self.chi2_table[dof]

    def measurement_update(self, H, r):
        """msckf.py:548-602 (A.10: QR only if m>n, non-Joseph covariance update)."""
        if len(H) == 0 or len(r) == 0:
            return
        if H.shape[0] > H.shape[1]:
            Q, R = np.linalg.qr(H, mode='reduced')
            H_thin, r_thin = R, Q.T @ r
        else:
            H_thin, r_thin = H, r
        P = self.state_cov
        S = H_thin @ P @ H_thin.T + self.config.observation_noise * np.identity(len(H_thin))
        K = np.linalg.solve(S, H_thin @ P).T
        dx = K @ r_thin
        self.debug['delta_x'] = dx
        s = self.imu_state
        # NOTE the reference updates these numpy arrays IN PLACE (msckf.py:580-587,595).  Because
        # process_model aliases position_null/velocity_null to the same array objects
        # (msckf.py:337-339) and state_augmentation aliases a camera's position_null to its
        # position (msckf.py:402-404), the "null" positions/velocity silently follow the update,
        # while orientation_null (rebound, not mutated) does not.  Reproduced on purpose.
        s.orientation = quaternion_multiplication(small_angle_quaternion(dx[:3]), s.orientation)
        s.gyro_bias += dx[3:6]
        s.velocity += dx[6:9]
        s.acc_bias += dx[9:12]
        s.position += dx[12:15]
        s.R_imu_cam0 = to_rotation(small_angle_quaternion(dx[15:18])) @ s.R_imu_cam0
        s.t_cam0_imu += dx[18:21]
        for i, cs in enumerate(self.cam_states.values()):
            d = dx[21 + 6 * i:27 + 6 * i]
            cs.orientation = quaternion_multiplication(small_angle_quaternion(d[:3]), cs.orientation)
            cs.position += d[3:]
        Pn = (np.identity(len(K)) - K @ H_thin) @ P
        self.state_cov = (Pn + Pn.T) / 2.

    # ---- feature / camera-state management ---------------------------------------------------
    def _try_init(self, feat):
        if feat.is_initialized:
            return True
        # check_motion is always True at translation_threshold = -1 (config.py:12, feature_motion_checker.py:14-15)
        if self.opt.translation_threshold >= 0 and not self._check_motion(feat):
            return False
        return bool(initialize_position(feat, self.cam_states, self.R_cam0_cam1, self.t_cam0_cam1))

    def _check_motion(self, feat):
        """feature_motion_checker.py:6-39."""
        ids = list(feat.observations.keys())
        a, b = self.cam_states[ids[0]], self.cam_states[ids[-1]]
        Ra = to_rotation(a.orientation).T
        d = np.array([*feat.observations[ids[0]][:2], 1.0])
        d = Ra @ (d / np.linalg.norm(d))
        t = b.position - a.position
        return np.linalg.norm(t - (t @ d) * d) > self.opt.translation_threshold

    def remove_lost_features(self):
        """msckf.py:614-676 (A.8: the > 1500 row cut happens after adding the crossing block)."""
        rows = 0
        invalid, processed = [], []
        for feat in self.map_server.values():
            if self.imu_state.id in feat.observations:
                continue
            if len(feat.observations) < 3:
                invalid.append(feat.id)
                continue
            if not self._try_init(feat):
                invalid.append(feat.id)
                continue
            rows += 4 * len(feat.observations) - 3
            processed.append(feat.id)
        for fid in invalid:
            del self.map_server[fid]
        if not processed:
            return
        H = np.zeros((rows, 21 + 6 * len(self.cam_states)))
        r = np.zeros(rows)
        k = 0
        for fid in processed:
            feat = self.map_server[fid]
            cam_ids = list(feat.observations.keys())
            Hj, rj = self.feature_jacobian(fid, cam_ids)
            if self.gating_test(Hj, rj, len(cam_ids) - 1):
                H[k:k + Hj.shape[0], :Hj.shape[1]] = Hj
                r[k:k + len(rj)] = rj
                k += Hj.shape[0]
            if k > 1500:
                break
        self.measurement_update(H[:k], r[:k])
        for fid in processed:
            del self.map_server[fid]

    def find_redundant_cam_states(self):
        """msckf.py:678-709."""
        pairs = list(self.cam_states.items())
        key_idx = len(pairs) - 4
        idx = key_idx + 1
        first = 0
        key_p = pairs[key_idx][1].position
        key_R = to_rotation(pairs[key_idx][1].orientation)
        out = []
        for _ in range(2):
            p = pairs[idx][1].position
            R = to_rotation(pairs[idx][1].orientation)
            dist = np.linalg.norm(p - key_p)
            ang = 2 * np.arccos(to_quaternion(R @ key_R.T)[-1])
            if ang < 0.2618 and dist < 0.4 and self.tracking_rate > 0.5:
                out.append(pairs[idx][0])
                idx += 1
            else:
                out.append(pairs[first][0])
                first += 1
                idx += 1
        return sorted(out)

    def prune_cam_state_buffer(self):
        """msckf.py:712-786."""
        if len(self.cam_states) < self.config.max_cam_state_size:
            return
        rm = self.find_redundant_cam_states()
        rows = 0
        for feat in self.map_server.values():
            inv = [c for c in rm if c in feat.observations]
            if not inv:
                continue
            if len(inv) == 1:
                del feat.observations[inv[0]]
                continue
            if not self._try_init(feat):
                for c in inv:
                    del feat.observations[c]
                continue
            rows += 4 * len(inv) - 3
        H = np.zeros((rows, 21 + 6 * len(self.cam_states)))
        r = np.zeros(rows)
        k = 0
        for feat in self.map_server.values():
            inv = [c for c in rm if c in feat.observations]
            if not inv:
                continue
            Hj, rj = self.feature_jacobian(feat.id, inv)
            if self.gating_test(Hj, rj, len(inv)):
                H[k:k + Hj.shape[0], :Hj.shape[1]] = Hj
                r[k:k + len(rj)] = rj
                k += Hj.shape[0]
            for c in inv:
                del feat.observations[c]
        self.measurement_update(H[:k], r[:k])
        for cid in rm:
            i = list(self.cam_states.keys()).index(cid)
            a, b = 21 + 6 * i, 27 + 6 * i
            keep = np.r_[0:a, b:self.state_cov.shape[0]]
            self.state_cov = self.state_cov[np.ix_(keep, keep)].copy()
            del self.cam_states[cid]

    def online_reset(self):
        """msckf.py:821-843."""
        if self.config.position_std_threshold <= 0:
            return
        sd = np.sqrt(np.array([self.state_cov[12, 12], self.state_cov[13, 13], self.state_cov[14, 14]]))
        if max(sd) < self.config.position_std_threshold:
            return
        self.cam_states.clear()
        self.map_server.clear()
        self.state_cov = self._initial_cov()

    def publish(self, time):
        """msckf.py:845-867."""
        s = self.imu_state
        T_i_w = Iso(to_rotation(s.orientation).T, s.position)
        T_b_w = self.T_imu_body * T_i_w * self.T_imu_body.inverse()
        vel = self.T_imu_body.R @ s.velocity
        R_w_c = s.R_imu_cam0 @ T_i_w.R.T
        t_c_w = s.position + T_i_w.R @ s.t_cam0_imu
        self.trajectory.append(np.array([s.timestamp, *s.position, *s.orientation]))
        return vio_result_t(time, T_b_w, vel, Iso(R_w_c.T, t_c_w))
