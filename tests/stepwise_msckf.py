"""TEST INFRASTRUCTURE: a step-by-step Python driver of the single-filter operator API (av_msckf_propagate / _augment /
_triangulate / _feature_blocks / _update / _remove_cam through uav_airvision_amd.msckf_ops.MsckfDevice).

Until round 2 this file was the drop-in `MSCKF` class.  The product class (uav_airvision_amd/dropin/msckf.py) is now a view over
the batched filter (av_msckf_batch_* with one stream), whose bookkeeping lives once, in C++; this driver keeps the operator
entry points under test: it sequences them the way MSCKF.feature_callback does (reference: src/msckf.py:177-228 and the methods
it calls, cited per method below) with the host-side dict bookkeeping in Python, and is compared with the reference's golden
vectors exactly like the product class.  Like the oracle it is not importable from product code.
"""
import os
from collections import namedtuple

import numpy as np

from feature import BaseFeature, Feature
from utils import (Isometry3d, from_two_vectors, quaternion_multiplication, skew, small_angle_quaternion,
                   to_quaternion, to_rotation)
from uav_airvision_amd.msckf_ops import FeatureBatch, MsckfDevice

_vio_result = namedtuple('vio_result', ['timestamp', 'pose', 'velocity', 'cam0_pose'])


def _make_output_filepath():
    """msckf.py:10-16."""
    base = 'results/txts'
    os.makedirs(base, exist_ok=True)
    return os.path.join(base, 'output_%s_offset%s.txt' % (os.getenv('DATASET_NAME', 'unknown'), os.getenv('TIME_OFFSET', '0')))


class IMUState(object):
    """msckf.py:18-58."""
    next_id = 0
    gravity = np.array([0., 0., -9.81])
    T_imu_body = Isometry3d(np.identity(3), np.zeros(3))

    def __init__(self, new_id=None):
        self.id = new_id
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


class CAMState(object):
    """msckf.py:61-77."""
    R_cam0_cam1 = None
    t_cam0_cam1 = None

    def __init__(self, new_id=None):
        self.id = new_id
        self.timestamp = None
        self.orientation = np.array([0., 0., 0., 1.])
        self.position = np.zeros(3)
        self.orientation_null = np.array([0., 0., 0., 1.])
        self.position_null = np.zeros(3)


class StateServer(object):
    """msckf.py:80-91; `state_cov` is a view of the device-resident covariance (downloaded on read)."""

    def __init__(self, dev):
        self._dev = dev
        self.imu_state = IMUState()
        self.cam_states = dict()
        self.continuous_noise_cov = np.zeros((12, 12))

    @property
    def state_cov(self):
        return self._dev.get_cov()

    @state_cov.setter
    def state_cov(self, P):
        self._dev.set_cov(P)


class MSCKF(object):
    def __init__(self, config, device=0, rows_cap=8192, write_trajectory=True):
        self.config = config
        self.optimization_config = config.optimization_config
        self.imu_msg_buffer = []
        self._dev = MsckfDevice(max_cam_states=config.max_cam_state_size, rows_cap=rows_cap, device=device)
        self.state_server = StateServer(self._dev)
        self.map_server = dict()
        self.chi_squared_test_table = {i: self._dev.chi2_table[i] for i in range(1, 100)}
        self.state_server.imu_state.velocity = config.velocity
        self.reset_state_cov()
        Qc = np.identity(12)
        Qc[:3, :3] *= config.gyro_noise
        Qc[3:6, 3:6] *= config.gyro_bias_noise
        Qc[6:9, 6:9] *= config.acc_noise
        Qc[9:, 9:] *= config.acc_bias_noise
        self.state_server.continuous_noise_cov = Qc
        self._noise = [config.gyro_noise, config.gyro_bias_noise, config.acc_noise, config.acc_bias_noise]
        self.gravity = np.array(config.gravity, dtype=np.float64)
        IMUState.gravity = self.gravity
        T_cam0_imu = np.linalg.inv(config.T_imu_cam0)
        self.state_server.imu_state.R_imu_cam0 = T_cam0_imu[:3, :3].T
        self.state_server.imu_state.t_cam0_imu = T_cam0_imu[:3, 3]
        self.T_cam0_cam1 = np.array(config.T_cn_cnm1, dtype=np.float64)
        CAMState.R_cam0_cam1 = self.T_cam0_cam1[:3, :3]
        CAMState.t_cam0_cam1 = self.T_cam0_cam1[:3, 3]
        BaseFeature.R_cam0_cam1 = CAMState.R_cam0_cam1
        BaseFeature.t_cam0_cam1 = CAMState.t_cam0_cam1
        self.T_imu_body = Isometry3d(config.T_imu_body[:3, :3], config.T_imu_body[:3, 3])
        IMUState.T_imu_body = self.T_imu_body
        self._next_imu_id = 0
        self.tracking_rate = None
        self.is_gravity_set = False
        self.is_first_img = True
        self._outfile = _make_output_filepath() if write_trajectory else None
        self.debug = {}

    def close(self):
        self._dev.close()

    # ---- callbacks ---------------------------------------------------------------------------
    def _write_state(self, s):
        """msckf.py:152-160."""
        if self._outfile is None:
            return
        with open(self._outfile, 'a') as f:
            f.write('%.6f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n' % (s.timestamp, s.position[0], s.position[1], s.position[2],
                                                                    s.orientation[0], s.orientation[1], s.orientation[2], s.orientation[3]))

    def imu_callback(self, imu_msg):
        """msckf.py:162-175."""
        self.imu_msg_buffer.append(imu_msg)
        if not self.is_gravity_set and len(self.imu_msg_buffer) >= 200:
            self.initialize_gravity_and_bias()
            self.is_gravity_set = True

    def feature_callback(self, feature_msg):
        """msckf.py:177-228."""
        if not self.is_gravity_set:
            return None
        if self.is_first_img:
            self.is_first_img = False
            self.state_server.imu_state.timestamp = feature_msg.timestamp
        self.batch_imu_processing(feature_msg.timestamp)
        self.state_augmentation(feature_msg.timestamp)
        self.add_feature_observations(feature_msg)
        self.remove_lost_features()
        self.prune_cam_state_buffer()
        try:
            return self.publish(feature_msg.timestamp)
        finally:
            self.online_reset()

    def initialize_gravity_and_bias(self):
        """msckf.py:230-249."""
        sw, sa = np.zeros(3), np.zeros(3)
        for m in self.imu_msg_buffer:
            sw += m.angular_velocity
            sa += m.linear_acceleration
        n = len(self.imu_msg_buffer)
        s = self.state_server.imu_state
        s.gyro_bias = sw / n
        g_imu = sa / n
        self.gravity = np.array([0., 0., -np.linalg.norm(g_imu)])
        IMUState.gravity = self.gravity
        s.orientation = from_two_vectors(-self.gravity, g_imu)

    # ---- propagation -------------------------------------------------------------------------
    def batch_imu_processing(self, time_bound):
        """msckf.py:251-273."""
        s = self.state_server.imu_state
        used = 0
        for m in self.imu_msg_buffer:
            if m.timestamp < s.timestamp:
                used += 1
                continue
            if m.timestamp > time_bound:
                break
            self.process_model(m.timestamp, m.angular_velocity, m.linear_acceleration)
            used += 1
            s.timestamp = m.timestamp
        s.id = self._next_imu_id
        self._next_imu_id += 1
        IMUState.next_id = self._next_imu_id
        self.imu_msg_buffer = self.imu_msg_buffer[used:]

    def process_model(self, time, m_gyro, m_acc):
        """msckf.py:275-339: RK4 state on the host, F/G/Phi/Q and the covariance on the GPU."""
        s = self.state_server.imu_state
        dt = time - s.timestamp
        gyro = m_gyro - s.gyro_bias
        acc = m_acc - s.acc_bias
        q_old = s.orientation
        self.predict_new_state(dt, gyro, acc)
        self._dev.propagate(dt, gyro, acc, q_old, s.orientation, s.orientation_null, s.velocity_null, s.position_null,
                            s.velocity, s.position, self.gravity, self._noise)
        s.orientation_null = s.orientation          # aliases on purpose (msckf.py:337-339), see measurement_update
        s.position_null = s.position
        s.velocity_null = s.velocity

    def predict_new_state(self, dt, gyro, acc):
        """msckf.py:341-388 (4th-order Runge-Kutta; k2 and k3 share the half-step rotation)."""
        s = self.state_server.imu_state
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
        Rt, Rt2 = to_rotation(dq_dt).T, to_rotation(dq_dt2).T
        g = self.gravity
        k1v = to_rotation(q).T @ acc + g
        k2v = Rt2 @ acc + g
        k3v = Rt2 @ acc + g
        k4v = Rt @ acc + g
        k1p = v
        k2p = v + k1v * dt / 2.
        k3p = v + k2v * dt / 2
        k4p = v + k3v * dt
        s.orientation = dq_dt / np.linalg.norm(dq_dt)
        s.velocity = v + (k1v + 2 * k2v + 2 * k3v + k4v) * dt / 6.
        s.position = p + (k1p + 2 * k2p + 2 * k3p + k4p) * dt / 6.

    def state_augmentation(self, time):
        """msckf.py:390-423."""
        s = self.state_server.imu_state
        R_w_i = to_rotation(s.orientation)
        cs = CAMState(s.id)
        cs.timestamp = time
        cs.orientation = to_quaternion(s.R_imu_cam0 @ R_w_i)
        cs.position = s.position + R_w_i.T @ s.t_cam0_imu
        cs.orientation_null = cs.orientation
        cs.position_null = cs.position              # alias on purpose (msckf.py:403-404)
        self.state_server.cam_states[s.id] = cs
        self._dev.augment(s.R_imu_cam0, skew(R_w_i.T @ s.t_cam0_imu))

    def add_feature_observations(self, feature_msg):
        """msckf.py:425-441."""
        sid = self.state_server.imu_state.id
        n_before = len(self.map_server)
        tracked = 0
        for f in feature_msg.features:
            z = np.array([f.u0, f.v0, f.u1, f.v1])
            if f.id not in self.map_server:
                mf = Feature(f.id, self.optimization_config)
                mf.observations[sid] = z
                self.map_server[f.id] = mf
            else:
                self.map_server[f.id].observations[sid] = z
                tracked += 1
        self.tracking_rate = tracked / (n_before + 1e-5)

    # ---- batched measurement model -----------------------------------------------------------
    def _cam_arrays(self):
        cs = list(self.state_server.cam_states.values())
        return ([c.orientation for c in cs], [c.position for c in cs], [c.orientation_null for c in cs], [c.position_null for c in cs])

    def _initialize_batch(self, feats):
        """check_motion + initialize_position (msckf.py:629-637 / 732-742) for many features at once;
        features are independent, so batching does not change the result.  Returns {id: ok}."""
        res = {}
        todo = []
        for f in feats:
            if f.is_initialized:
                res[f.id] = True
            elif not f.check_motion(self.state_server.cam_states):
                res[f.id] = False
            else:
                todo.append(f)
        if todo:
            keys = {k: i for i, k in enumerate(self.state_server.cam_states.keys())}
            cams, zs = [], []
            for f in todo:
                ids = [c for c in f.observations if c in keys]
                cams.append([keys[c] for c in ids])
                zs.append([f.observations[c] for c in ids])
            q, p, _, _ = self._cam_arrays()
            batch = FeatureBatch(cams, zs, self._dev.device)
            pos, ok = self._dev.triangulate(batch, q, p, self.T_cam0_cam1, self.optimization_config)
            for i, f in enumerate(todo):
                f.position = pos[i]
                f.is_initialized = bool(ok[i])
                res[f.id] = bool(ok[i])
        return res

    def _blocks(self, feats, cam_id_lists, dofs):
        """feature_jacobian + gating_test for many features (msckf.py:509-546, 604-612)."""
        keys = {k: i for i, k in enumerate(self.state_server.cam_states.keys())}
        cams = [[keys[c] for c in ids] for ids in cam_id_lists]
        zs = [[f.observations[c] for c in ids] for f, ids in zip(feats, cam_id_lists)]
        q, p, qn, pn = self._cam_arrays()
        batch = FeatureBatch(cams, zs, self._dev.device)
        row_off, rows, gamma, ok = self._dev.feature_blocks(batch, [f.position for f in feats], dofs, q, p, qn, pn,
                                                            self.T_cam0_cam1, self.gravity, self.config.observation_noise)
        self._gamma = gamma            # debug['gamma'] gets the ones the reference would have evaluated (see the callers)
        return row_off, rows, ok

    def measurement_update(self, blk_rows, blk_lens):
        """msckf.py:548-602: the stacked update runs on the GPU, the state injection on the host."""
        if len(blk_lens) == 0:
            return
        dx = self._dev.update(blk_rows, blk_lens, self.config.observation_noise)
        self.debug['delta_x'] = dx
        if np.linalg.norm(dx[6:9]) > 0.5 or np.linalg.norm(dx[12:15]) > 1.0:
            print('[Warning] Update change is too large')
        s = self.state_server.imu_state
        s.orientation = quaternion_multiplication(small_angle_quaternion(dx[:3]), s.orientation)
        s.gyro_bias += dx[3:6]                       # in place like the reference: the aliased *_null states follow
        s.velocity += dx[6:9]
        s.acc_bias += dx[9:12]
        s.position += dx[12:15]
        s.R_imu_cam0 = to_rotation(small_angle_quaternion(dx[15:18])) @ s.R_imu_cam0
        s.t_cam0_imu += dx[18:21]
        for i, cs in enumerate(self.state_server.cam_states.values()):
            d = dx[21 + 6 * i:27 + 6 * i]
            cs.orientation = quaternion_multiplication(small_angle_quaternion(d[:3]), cs.orientation)
            cs.position += d[3:]

    # ---- feature / camera-state management -----------------------------------------------------
    def remove_lost_features(self):
        """msckf.py:614-676."""
        sid = self.state_server.imu_state.id
        invalid, cand = [], []
        for feat in self.map_server.values():
            if sid in feat.observations:
                continue
            if len(feat.observations) < 3:
                invalid.append(feat.id)
                continue
            cand.append(feat)
        ok = self._initialize_batch(cand)
        processed = []
        for feat in cand:
            (processed if ok[feat.id] else invalid).append(feat if ok[feat.id] else feat.id)
        for fid in invalid:
            del self.map_server[fid]
        if not processed:
            return
        cam_lists = [list(f.observations.keys()) for f in processed]
        row_off, rows, passed = self._blocks(processed, cam_lists, [len(c) - 1 for c in cam_lists])
        blk_r, blk_l, stacked, n_eval = [], [], 0, 0
        for i in range(len(processed)):
            n_eval = i + 1
            if passed[i]:
                blk_r.append(row_off[i]); blk_l.append(rows[i])
                stacked += int(rows[i])
            if stacked > 1500:                      # the cut happens after adding the crossing block (A.8)
                break
        # the batched kernel gated every candidate; the reference never evaluates the ones behind the cut (msckf.py:667-668)
        self.debug.setdefault('gamma', []).extend(self._gamma[:n_eval].tolist())
        self.measurement_update(blk_r, blk_l)
        for f in processed:
            del self.map_server[f.id]

    def find_redundant_cam_states(self):
        """msckf.py:678-709."""
        pairs = list(self.state_server.cam_states.items())
        key_idx = len(pairs) - 4
        idx, first = key_idx + 1, 0
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
            else:
                out.append(pairs[first][0])
                first += 1
            idx += 1
        return sorted(out)

    def prune_cam_state_buffer(self):
        """msckf.py:712-786."""
        cam_states = self.state_server.cam_states
        if len(cam_states) < self.config.max_cam_state_size:
            return
        rm = self.find_redundant_cam_states()
        involved, need_init = {}, []
        for feat in self.map_server.values():
            inv = [c for c in rm if c in feat.observations]
            if not inv:
                continue
            if len(inv) == 1:
                del feat.observations[inv[0]]
                continue
            involved[feat.id] = inv
            need_init.append(feat)
        ok = self._initialize_batch(need_init)
        feats, lists = [], []
        for feat in need_init:
            if not ok[feat.id]:
                for c in involved[feat.id]:
                    del feat.observations[c]
                continue
            feats.append(feat)
            lists.append(involved[feat.id])
        if feats:
            row_off, rows, passed = self._blocks(feats, lists, [len(c) for c in lists])
            self.debug.setdefault('gamma', []).extend(self._gamma.tolist())          # no cut on this path (msckf.py:759-763)
            blk_r = [row_off[i] for i in range(len(feats)) if passed[i]]
            blk_l = [rows[i] for i in range(len(feats)) if passed[i]]
            for feat, inv in zip(feats, lists):
                for c in inv:
                    del feat.observations[c]
            self.measurement_update(blk_r, blk_l)
        for cid in rm:
            self._dev.remove_cam(list(cam_states.keys()).index(cid))
            del cam_states[cid]

    def reset_state_cov(self):
        """msckf.py:788-798."""
        c = self.config
        P = np.zeros((21, 21))
        P[3:6, 3:6] = c.gyro_bias_cov * np.identity(3)
        P[6:9, 6:9] = c.velocity_cov * np.identity(3)
        P[9:12, 9:12] = c.acc_bias_cov * np.identity(3)
        P[15:18, 15:18] = c.extrinsic_rotation_cov * np.identity(3)
        P[18:21, 18:21] = c.extrinsic_translation_cov * np.identity(3)
        self._dev.set_cov(P)

    def reset(self):
        """msckf.py:800-819."""
        old = self.state_server.imu_state
        s = IMUState()
        s.id = old.id
        s.R_imu_cam0 = old.R_imu_cam0
        s.t_cam0_imu = old.t_cam0_imu
        self.state_server.imu_state = s
        self.state_server.cam_states.clear()
        self.reset_state_cov()
        self.map_server.clear()
        self.imu_msg_buffer.clear()
        self.is_gravity_set = False
        self.is_first_img = True

    def online_reset(self):
        """msckf.py:821-843."""
        if self.config.position_std_threshold <= 0:
            return
        P = self._dev.get_cov()
        if max(np.sqrt(P[12, 12]), np.sqrt(P[13, 13]), np.sqrt(P[14, 14])) < self.config.position_std_threshold:
            return
        print('Start online reset...')
        self.state_server.cam_states.clear()
        self.map_server.clear()
        self.reset_state_cov()

    def publish(self, time):
        """msckf.py:845-867."""
        s = self.state_server.imu_state
        T_i_w = Isometry3d(to_rotation(s.orientation).T, s.position)
        T_b_w = self.T_imu_body * T_i_w * self.T_imu_body.inverse()
        body_velocity = self.T_imu_body.R @ s.velocity
        R_w_c = s.R_imu_cam0 @ T_i_w.R.T
        t_c_w = s.position + T_i_w.R @ s.t_cam0_imu
        self._write_state(s)
        return _vio_result(time, T_b_w, body_velocity, Isometry3d(R_w_c.T, t_c_w))
