"""Python handle of the MSCKF device context (av_msckf_* C ABI): covariance resident on the GPU, batched
per-feature kernels, stacked update.  Used by dropin/msckf.py and dropin/feature; numpy in, numpy out."""
import ctypes as C

import numpy as np
import torch
from scipy.stats import chi2

from . import _native as N


def _d(a):
    return (C.c_double * len(a))(*[float(v) for v in a])


class FeatureBatch(object):
    """CSR description of a batch of features on the device."""

    def __init__(self, obs_cam_lists, obs_z_lists, device):
        dev = torch.device('cuda', device)
        counts = [len(c) for c in obs_cam_lists]
        self.n = len(counts)
        self.max_obs = max(counts) if counts else 0
        off = np.zeros(self.n + 1, np.int32)
        off[1:] = np.cumsum(counts)
        cam = np.concatenate([np.asarray(c, np.int32) for c in obs_cam_lists]) if self.n else np.zeros(0, np.int32)
        z = np.concatenate([np.asarray(zz, np.float64).reshape(-1, 4) for zz in obs_z_lists]) if self.n else np.zeros((0, 4))
        self.counts = counts
        self.off = torch.from_numpy(off).to(dev)
        self.cam = torch.from_numpy(np.ascontiguousarray(cam)).to(dev)
        self.z = torch.from_numpy(np.ascontiguousarray(z)).to(dev)


class MsckfDevice(object):
    def __init__(self, max_cam_states=20, rows_cap=8192, device=0):
        self.device = int(device)
        self.dev = torch.device('cuda', self.device)
        table = np.zeros(100)
        table[1:] = [chi2.ppf(0.05, i) for i in range(1, 100)]          # msckf.py:111-113
        self.chi2_table = table
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_create(int(max_cam_states), int(rows_cap), _d(table), self.device, C.byref(self._h)))
        self.rows_cap = rows_cap

    def close(self):
        if self._h:
            N.lib().av_msckf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def dim(self):
        return N.lib().av_msckf_dim(self._h)

    def _st(self):
        return N.current_stream()

    def set_cov(self, P):
        P = np.ascontiguousarray(P, dtype=np.float64)
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_set_cov(self._h, P.ctypes.data_as(C.c_void_p), P.shape[0], self._st()))

    def get_cov(self):
        n = self.dim
        P = np.empty((n, n), np.float64)
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_get_cov(self._h, P.ctypes.data_as(C.c_void_p), n, self._st()))
        return P

    def propagate(self, dt, gyro, acc, q_old, q_new, q_null, v_null, p_null, v_new, p_new, gravity, noise):
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_propagate(self._h, float(dt), _d(gyro), _d(acc), _d(q_old), _d(q_new), _d(q_null), _d(v_null),
                                               _d(p_null), _d(v_new), _d(p_new), _d(gravity), _d(noise), self._st()))

    def augment(self, R_imu_cam0, skew_Rt_t):
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_augment(self._h, _d(np.asarray(R_imu_cam0).reshape(-1)), _d(np.asarray(skew_Rt_t).reshape(-1)), self._st()))

    def remove_cam(self, idx):
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_remove_cam(self._h, int(idx), self._st()))

    def _cams(self, cam_q, cam_p):
        q = torch.from_numpy(np.ascontiguousarray(cam_q, dtype=np.float64).reshape(-1, 4)).to(self.dev)
        p = torch.from_numpy(np.ascontiguousarray(cam_p, dtype=np.float64).reshape(-1, 3)).to(self.dev)
        return q, p

    def triangulate(self, batch, cam_q, cam_p, T_cam0_cam1, opt):
        """-> (positions float64[n,3], valid bool[n])."""
        if batch.n == 0:
            return np.zeros((0, 3)), np.zeros(0, bool)
        q, p = self._cams(cam_q, cam_p)
        pos = torch.empty((batch.n, 3), dtype=torch.float64, device=self.dev)
        valid = torch.empty(batch.n, dtype=torch.int32, device=self.dev)
        opt5 = _d([opt.huber_epsilon, opt.estimation_precision, opt.initial_damping,
                   opt.outer_loop_max_iteration, opt.inner_loop_max_iteration])
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_triangulate(self._h, batch.n, N.dptr(batch.off), N.dptr(batch.cam), N.dptr(batch.z), N.dptr(q), N.dptr(p),
                                                 _d(np.asarray(T_cam0_cam1, dtype=np.float64).reshape(-1)), opt5, 2 * batch.max_obs,
                                                 N.dptr(pos), N.dptr(valid), self._st()))
            torch.cuda.synchronize()
        return pos.cpu().numpy(), valid.cpu().numpy().astype(bool)

    def feature_blocks(self, batch, positions, dofs, cam_q, cam_p, cam_qn, cam_pn, T_cam0_cam1, gravity, obs_noise):
        """Jacobian + null-space projection + gate for the batch.  -> (row_off int[n], gamma[n], pass bool[n])."""
        n_cam = len(cam_q)
        rows = np.array([4 * c - 3 for c in batch.counts], np.int32)
        row_off = np.zeros(batch.n, np.int32)
        if batch.n:
            row_off[1:] = np.cumsum(rows)[:-1]
        total = int(rows.sum())
        if batch.n == 0:
            return row_off, rows, np.zeros(0), np.zeros(0, bool)
        q, p = self._cams(cam_q, cam_p)
        qn, pn = self._cams(cam_qn, cam_pn)
        pos = torch.from_numpy(np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)).to(self.dev)
        dof = torch.from_numpy(np.ascontiguousarray(dofs, dtype=np.int32)).to(self.dev)
        ro = torch.from_numpy(row_off).to(self.dev)
        gamma = torch.empty(batch.n, dtype=torch.float64, device=self.dev)
        ok = torch.empty(batch.n, dtype=torch.int32, device=self.dev)
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_feature_blocks(self._h, batch.n, n_cam, batch.max_obs, N.dptr(batch.off), N.dptr(batch.cam), N.dptr(batch.z),
                                                    N.dptr(pos), N.dptr(dof), N.dptr(ro), total, N.dptr(q), N.dptr(p), N.dptr(qn), N.dptr(pn),
                                                    _d(np.asarray(T_cam0_cam1, dtype=np.float64).reshape(-1)), _d(gravity), float(obs_noise),
                                                    N.dptr(gamma), N.dptr(ok), self._st()))
            torch.cuda.synchronize()
        return row_off, rows, gamma.cpu().numpy(), ok.cpu().numpy().astype(bool)

    def update(self, blk_rows, blk_lens, obs_noise):
        """Stacked EKF update on the selected blocks; returns delta_x (n doubles)."""
        n = self.dim
        dx = np.zeros(n)
        total = int(np.sum(blk_lens)) if len(blk_lens) else 0
        if total == 0:
            return dx
        br = torch.from_numpy(np.ascontiguousarray(blk_rows, dtype=np.int32)).to(self.dev)
        bl = torch.from_numpy(np.ascontiguousarray(blk_lens, dtype=np.int32)).to(self.dev)
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_update(self._h, N.dptr(br), N.dptr(bl), len(blk_lens), total, float(obs_noise),
                                            dx.ctypes.data_as(C.c_void_p), self._st()))
        return dx


class BatchedMSCKF(object):
    """S independent filters stepped together (av_msckf_batch_* C ABI): C++ bookkeeping inside the
    library, one batched launch per numeric phase.  Mirrors MSCKF.imu_callback / feature_callback
    (reference: src/msckf.py:162-228) for many streams at once."""

    def __init__(self, config, n_streams, device=0, rows_cap=None, max_features=None):
        """rows_cap: rows of the per-stream block buffer.  None = sized by the library from the capacity of the first
        feature message (the camera-pruning update stacks 5 rows per feature, the lost-feature update at most 1500 + one
        block); `max_features`, when given, sizes it up front instead.  With the automatic size a later step with a wider
        `ids.shape[1]` REBUILDS the block buffers and the device-resident observation store for the wider message (drains the
        batch, synchronises the device, grows by at least x1.5, keeps the outgrown allocations until close) -- pass
        max_features when the width of the feature arrays can vary.  An explicit rows_cap is never grown.
        config.max_cam_state_size <= 24 (reference: 20)."""
        if rows_cap is None:
            rows_cap = 0 if max_features is None else max(2048, 5 * int(max_features) + 64)
        self.rows_cap = int(rows_cap)
        self.config = config
        self.S = int(n_streams)
        self.device = int(device)
        table = np.zeros(100)
        table[1:] = [chi2.ppf(0.05, i) for i in range(1, 100)]
        T_cam0_imu = np.linalg.inv(config.T_imu_cam0)
        rt = np.concatenate([T_cam0_imu[:3, :3].T.reshape(-1), T_cam0_imu[:3, 3]])
        o = config.optimization_config
        self._h = C.c_void_p()
        self._inflight = []
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_batch_create(
                self.S, int(config.max_cam_state_size), int(rows_cap), _d(table), _d(config.gravity),
                _d(np.asarray(config.T_cn_cnm1, dtype=np.float64).reshape(-1)), _d(rt),
                _d([config.gyro_bias_cov, config.velocity_cov, config.acc_bias_cov, config.extrinsic_rotation_cov, config.extrinsic_translation_cov]),
                _d([config.gyro_noise, config.gyro_bias_noise, config.acc_noise, config.acc_bias_noise]),
                float(config.observation_noise), float(config.position_std_threshold), _d(config.velocity),
                _d([o.huber_epsilon, o.estimation_precision, o.initial_damping, o.outer_loop_max_iteration, o.inner_loop_max_iteration, o.translation_threshold]),
                self.device, C.byref(self._h)))

    def close(self):
        if self._h:
            if getattr(self, '_inflight', None):
                N.lib().av_msckf_batch_wait(self._h, 0)          # queued steps still reference the arrays in _inflight
                self._inflight = []
            N.lib().av_msckf_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push_imu(self, stream_idx, timestamps, gyro, acc):
        si = np.ascontiguousarray(stream_idx, dtype=np.int32)
        ts = np.ascontiguousarray(timestamps, dtype=np.float64)
        g = np.ascontiguousarray(gyro, dtype=np.float64).reshape(-1, 3)
        a = np.ascontiguousarray(acc, dtype=np.float64).reshape(-1, 3)
        N.check(N.lib().av_msckf_batch_push_imu(self._h, si.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p),
                                                g.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), len(si)))

    def step(self, ids, uv, n_feat, timestamps):
        """ids int64[S,cap], uv float64[S,cap,4], n_feat int32[S], timestamps float64[S] (host arrays).
        Returns float64[S,12]: published, t, p(3), q(4), v(3)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        uv = np.ascontiguousarray(uv, dtype=np.float64)
        nf = np.ascontiguousarray(n_feat, dtype=np.int32)
        ts = np.ascontiguousarray(timestamps, dtype=np.float64)
        cap = ids.shape[1]
        out = np.zeros((self.S, 12))
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_batch_step(self._h, ids.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p),
                                                nf.ctypes.data_as(C.c_void_p), cap, ts.ctypes.data_as(C.c_void_p),
                                                out.ctypes.data_as(C.c_void_p), N.current_stream()))
        return out

    def submit(self, ids, uv, n_feat, timestamps):
        """Queue one step (av_msckf_batch_submit) and return its float64[S,12] output array, which is filled once the
        step has retired -- i.e. after a `wait(k)` that leaves fewer than the steps submitted after it pending.  The
        stream groups of the batch run behind their queues independently of each other."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        uv = np.ascontiguousarray(uv, dtype=np.float64)
        nf = np.ascontiguousarray(n_feat, dtype=np.int32)
        ts = np.ascontiguousarray(timestamps, dtype=np.float64)
        out = np.zeros((self.S, 12))
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_batch_submit(self._h, ids.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p),
                                                  nf.ctypes.data_as(C.c_void_p), ids.shape[1], ts.ctypes.data_as(C.c_void_p),
                                                  out.ctypes.data_as(C.c_void_p), N.current_stream()))
        self._inflight.append((ids, uv, nf, ts, out))          # the library reads / writes these until the step retires
        return out

    def device_resident(self):
        """True when the filter state and the observation map live on the device (the default; `submit_dev` needs it)."""
        return N.lib().av_msckf_batch_device_resident(self._h) == 1

    def submit_dev(self, engine, timestamps, msg_stream=None):
        """Queue one step on the feature message `engine` (a FrontendEngine with the same streams) has just published, read
        where it lies on the device (av_msckf_batch_submit_dev): no host round trip between the front-end and the filter.
        msg_stream: the stream handle the engine's step ran on (default: torch's current stream, i.e. call this right after
        `engine.step` on the same stream).  The filter's kernels run on the current stream / the stream groups' own streams.
        Returns the float64[S,12] output array, filled once a `wait` has let the step retire."""
        ids, uv, n, cap = engine.features_dev()
        ts = np.ascontiguousarray(timestamps, dtype=np.float64)
        assert ts.shape == (self.S,) and engine.n_streams == self.S
        out = np.zeros((self.S, 12))
        with torch.cuda.device(self.device):
            ms = N.current_stream() if msg_stream is None else msg_stream
            N.check(N.lib().av_msckf_batch_submit_dev(self._h, ids, uv, n, cap, ts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                                      ms, N.current_stream()))
        self._inflight.append((ts, out))
        return out

    def wait(self, max_pending=0):
        """Block until at most `max_pending` submitted steps are unfinished; raises the first error of a retired step."""
        N.check(N.lib().av_msckf_batch_wait(self._h, int(max_pending)))
        while len(self._inflight) > max_pending:
            self._inflight.pop(0)

    def sizes(self, s):
        o = (C.c_int32 * 3)()
        N.check(N.lib().av_msckf_batch_sizes(self._h, int(s), C.byref(o)))
        return int(o[0]), int(o[1]), int(o[2])

    COUNTER_NAMES = ('steps', 'prune_stream_steps', 'two_pass_streams', 'devbuf_growths', 'min_cam_states', 'max_cam_states',
                     'min_map_features', 'max_map_features')

    def counters(self):
        """Run statistics over all streams (av_msckf_batch_counters); drains the queue first.  `two_pass_streams` counts the stream-steps
        whose lost-feature candidates needed more rows than `rows_cap`: the device-resident filter serves them from the overflow pool all
        streams share (AV_MSCKF_POOL_ROWS; an exhausted pool stops the stream with AV_E_CAPACITY), the host-bookkeeping path in two passes."""
        self.wait(0)
        o = (C.c_int64 * 8)()
        N.check(N.lib().av_msckf_batch_counters(self._h, C.byref(o)))
        return dict(zip(self.COUNTER_NAMES, [int(v) for v in o]))

    def get_cov(self, s):
        n = self.sizes(s)[0]
        P = np.empty((n, n))
        with torch.cuda.device(self.device):
            N.check(N.lib().av_msckf_batch_get_cov(self._h, int(s), P.ctypes.data_as(C.c_void_p), n, N.current_stream()))
        return P

    def get_state(self, s):
        """Everything measurement_update injects into (msckf.py:568-595) for stream s: dict(t, q, p, v, bg, ba, R_ic, t_ci,
        gravity, cam_ids, cam_q, cam_p).  Drain (`wait(0)`) first."""
        imu = (C.c_double * 32)()
        n = C.c_int32(0)
        cap = int(self.config.max_cam_state_size) + 1
        ids = np.zeros(cap, np.int64); qp = np.zeros((cap, 7))
        N.check(N.lib().av_msckf_batch_get_state(self._h, int(s), C.byref(imu), ids.ctypes.data_as(C.c_void_p), qp.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        a = np.array(imu[:])
        k = int(n.value)
        return dict(t=a[0], q=a[1:5], p=a[5:8], v=a[8:11], bg=a[11:14], ba=a[14:17], R_ic=a[17:26].reshape(3, 3), t_ci=a[26:29],
                    gravity=a[29:32], cam_ids=ids[:k].copy(), cam_q=qp[:k, :4].copy(), cam_p=qp[:k, 4:].copy())

    def stream_status(self, s):
        """(0, '') while stream s runs; (AV_E_* code, reason) once a per-stream capacity failure stopped it (its step output
        then reads published = -1; the other streams keep running)."""
        st = C.c_int32(0)
        msg = C.create_string_buffer(160)
        N.check(N.lib().av_msckf_batch_stream_status(self._h, int(s), C.byref(st), msg, 160))
        return int(st.value), msg.value.decode()

    WORK_NAMES = ('gate_flops', 'update_flops', 'reference_qr_flops', 'features_gated', 'updates', 'rows_stacked', 'chain_ms', 'chain_timings_dropped')

    def work(self, enable=-1):
        """Algorithmic fp64 flops of the gates / updates run so far and the device time of the phase chains
        (av_msckf_batch_work); enable=1/0 switches the event timing on / off, -1 only reads.  Drains the queue first."""
        self.wait(0)
        o = (C.c_double * 8)()
        N.check(N.lib().av_msckf_batch_work(self._h, int(enable), C.byref(o)))
        return dict(zip(self.WORK_NAMES, [float(v) for v in o]))

    def work_executed(self):
        """fp64 flops the gate / update kernels EXECUTED so far on their block-sparse shapes (av_msckf_batch_work_executed; analytic per
        gated feature / per update): {'gate_flops_executed', 'update_flops_executed'}.  Drains the queue first."""
        self.wait(0)
        o = (C.c_double * 2)()
        N.check(N.lib().av_msckf_batch_work_executed(self._h, C.byref(o)))
        return {'gate_flops_executed': float(o[0]), 'update_flops_executed': float(o[1])}

    def debug_capture(self, enable=True):
        """Parity-test tap: keep gamma / delta_x / P+ of the two update phases of every stream's last step (tests only)."""
        N.check(N.lib().av_msckf_batch_debug_capture(self._h, 1 if enable else 0))

    def debug_read(self, s, phase, gamma_cap=8192):
        """(gamma float64[k], rows, dx float64[n] or None, P_after float64[n,n] or None) of phase 0 (remove_lost_features) or
        1 (prune_cam_state_buffer) of stream s's last step."""
        ncap = 21 + 6 * (int(self.config.max_cam_state_size) + 1)
        g = np.zeros(gamma_cap); dx = np.zeros(ncap); P = np.zeros(ncap * ncap)
        ng, n, rows = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        N.check(N.lib().av_msckf_batch_debug_read(self._h, int(s), int(phase), g.ctypes.data_as(C.c_void_p), gamma_cap, C.byref(ng),
                                                  dx.ctypes.data_as(C.c_void_p), P.ctypes.data_as(C.c_void_p), ncap, C.byref(n), C.byref(rows)))
        k = int(n.value)
        if rows.value <= 0:
            return g[:ng.value].copy(), 0, None, None
        return g[:ng.value].copy(), int(rows.value), dx[:k].copy(), P[:k * k].reshape(k, k).copy()
