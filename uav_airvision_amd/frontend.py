"""FrontendEngine: Python handle of the device-resident image front-end (av_frontend_* C ABI).

One engine owns S independent stereo streams on one GPU.  This is the throughput path; the drop-in
`image_processing.ImageProcessor` (uav_airvision_amd/dropin) is the same engine with S = 1.
Reference surface mirrored: ImageProcessingPipeline.{__init__, imu_callback, stereo_callback}
(reference: src/image_processing/pipeline.py:14-150).
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N


def pack_frontend_config(config, max_corners=8192):
    """Build the packed av_frontend_config from a reference-style config object.  The
    extrinsics-derived matrices are computed with numpy exactly as the reference does
    (imu_processor.py:10-16, stereo_matcher.py:47,90-91,103-104)."""
    c = N.FrontendConfig()
    w, h = [int(v) for v in config.cam0_resolution]
    c.width, c.height = w, h
    c.grid_row, c.grid_col = int(config.grid_row), int(config.grid_col)
    c.grid_min_feature_num = int(config.grid_min_feature_num)
    c.grid_max_feature_num = int(config.grid_max_feature_num)
    c.fast_threshold = int(config.fast_threshold)
    lk = config.lk_params
    if lk['winSize'][0] != lk['winSize'][1]:
        raise ValueError('square LK windows only')
    if not (lk['flags'] & 4):
        raise ValueError('the front-end requires OPTFLOW_USE_INITIAL_FLOW (config.py:44)')
    ctype, max_iter, eps = lk['criteria']
    c.lk_win = int(lk['winSize'][0])
    c.lk_levels = int(lk['maxLevel']) + 1
    c.lk_max_iter = int(max_iter) if (ctype & 1) else 30
    c.lk_eps = float(eps) if (ctype & 2) else 0.01
    c.lk_min_eig = 1e-4
    c.max_corners = int(max_corners)
    c.stereo_threshold = float(config.stereo_threshold)
    for name, src in (('cam0_intrinsics', config.cam0_intrinsics), ('cam0_distortion', config.cam0_distortion_coeffs),
                      ('cam1_intrinsics', config.cam1_intrinsics), ('cam1_distortion', config.cam1_distortion_coeffs)):
        getattr(c, name)[:] = [float(v) for v in src]
    c.cam0_distortion_model = N.distortion_model_code(config.cam0_distortion_model)      # 'equidistant' = cv2.fisheye.* (camera_model.py:41,69)
    c.cam1_distortion_model = N.distortion_model_code(config.cam1_distortion_model)
    T_cam0_imu = np.linalg.inv(config.T_imu_cam0)
    T_cam1_imu = np.linalg.inv(config.T_imu_cam1)
    R_cam0_imu, t_cam0_imu = T_cam0_imu[:3, :3], T_cam0_imu[:3, 3]
    R_cam1_imu, t_cam1_imu = T_cam1_imu[:3, :3], T_cam1_imu[:3, 3]
    R0to1 = R_cam1_imu.T @ R_cam0_imu
    t01 = R_cam1_imu.T @ (t_cam0_imu - t_cam1_imu)
    x, y, z = t01
    E = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]]) @ R0to1
    c.R_cam0_imu[:] = [float(v) for v in R_cam0_imu.reshape(-1)]
    c.R_cam1_imu[:] = [float(v) for v in R_cam1_imu.reshape(-1)]
    c.R0to1[:] = [float(v) for v in R0to1.reshape(-1)]
    c.E[:] = [float(v) for v in E.reshape(-1)]
    c.norm_unit = float(4.0 / (2 * config.cam0_intrinsics[0] + 2 * config.cam0_intrinsics[1]))
    return c


COUNTER_NAMES = ('before_tracking', 'after_tracking', 'after_matching', 'n_fast', 'n_candidates', 'n_new',
                 'n_published', 'overflow')


class FrontendEngine(object):
    def __init__(self, config, n_streams=1, device=0, max_corners=8192, inputs_persist=False):
        """inputs_persist: promise that the cuda tensors handed to `step` stay unmodified until the NEXT step has run
        (AV_FE_INPUTS_PERSIST, include/airvision.h): pyramid level 0 is then read in place instead of being copied.  The engine
        keeps a reference to the last cam0 tensor, so dropping yours is fine; overwriting it in place is not.  `step_host`
        always works in place on the library's own staging slots.  Same results either way."""
        self.config = config
        self.n_streams = int(n_streams)
        self.device = int(device)
        self._cfg = pack_frontend_config(config, max_corners)
        self._cfg.flags = N.AV_FE_INPUTS_PERSIST if inputs_persist else 0
        self._keep = None
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_create(C.byref(self._cfg), self.n_streams, self.device, C.byref(self._h)))
        self.max_features = N.lib().av_frontend_max_features(self._h)
        self.width, self.height = self._cfg.width, self._cfg.height
        S, cap = self.n_streams, self.max_features
        self._ids = np.zeros((S, cap), np.int64)
        self._uv = np.zeros((S, cap, 4), np.float64)
        self._n = np.zeros(S, np.int32)

    def close(self):
        if self._h:
            N.lib().av_frontend_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return N.current_stream()

    def push_imu(self, stream, timestamp, gyro):
        g = (C.c_double * 3)(float(gyro[0]), float(gyro[1]), float(gyro[2]))
        N.check(N.lib().av_frontend_push_imu(self._h, int(stream), float(timestamp), g))

    def push_imu_batch(self, stream_idx, timestamps, gyro):
        """stream_idx int32[n], timestamps float64[n], gyro float64[n,3] in one C call."""
        si = np.ascontiguousarray(stream_idx, dtype=np.int32)
        ts = np.ascontiguousarray(timestamps, dtype=np.float64)
        g = np.ascontiguousarray(gyro, dtype=np.float64).reshape(-1, 3)
        assert len(si) == len(ts) == len(g)
        N.check(N.lib().av_frontend_push_imu_batch(self._h, si.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p),
                                                   g.ctypes.data_as(C.c_void_p), len(si)))

    def step(self, img0, img1, timestamps):
        """img0/img1: uint8 cuda tensors [S,h,w] (contiguous); timestamps: S floats.  Enqueues only."""
        S = self.n_streams
        assert img0.is_cuda and img1.is_cuda and img0.dtype == torch.uint8 and img1.dtype == torch.uint8
        assert tuple(img0.shape) == (S, self.height, self.width) == tuple(img1.shape), (img0.shape, img1.shape)
        assert img0.is_contiguous() and img1.is_contiguous()
        ts = (C.c_double * S)(*[float(t) for t in timestamps])
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_step(self._h, N.dptr(img0), N.dptr(img1), self.height * self.width, ts, self._stream()))
        self._keep = (self._keep[1] if self._keep else None, (img0, img1))       # this frame's and the previous frame's tensors stay alive

    def prestage(self, img0, img1):
        """Build the pyramids of the NEXT step's images now (av_frontend_prestage): `step` with the same tensors then starts with its
        tracking launch.  Same results; the engine must have been created with inputs_persist=True."""
        S = self.n_streams
        assert img0.is_cuda and img1.is_cuda and img0.dtype == torch.uint8 and img1.dtype == torch.uint8
        assert tuple(img0.shape) == (S, self.height, self.width) == tuple(img1.shape) and img0.is_contiguous() and img1.is_contiguous()
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_prestage(self._h, N.dptr(img0), N.dptr(img1), self.height * self.width, self._stream()))
        self._pre = (img0, img1)                                                # alive until the step that uses them

    def step_host(self, img0, img1, timestamps):
        """numpy uint8 [S,h,w] (or [h,w] when S == 1)."""
        S = self.n_streams
        a0 = np.ascontiguousarray(img0, dtype=np.uint8).reshape(S, self.height, self.width)
        a1 = np.ascontiguousarray(img1, dtype=np.uint8).reshape(S, self.height, self.width)
        ts = (C.c_double * S)(*[float(t) for t in np.atleast_1d(timestamps)])
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_step_host(self._h, a0.ctypes.data_as(C.c_void_p), a1.ctypes.data_as(C.c_void_p),
                                                  self.height * self.width, ts, self._stream()))

    def frames_reserve(self, n_slots):
        """Allocate the shared frame store (av_frontend_frames_reserve): `n_slots` resident stereo frames."""
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_frames_reserve(self._h, int(n_slots)))

    def frames_upload(self, slots, img0, img1):
        """Put host frames into the store: slots int32[n], img0 / img1 uint8[n,h,w] (C-contiguous).  Copy, pyramids and FAST of
        every frame happen once, here, on the engine's copy stream; the arrays are free again on return."""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        n = len(sl)
        if n == 0:
            return
        a0 = np.ascontiguousarray(img0, dtype=np.uint8).reshape(n, self.height, self.width)
        a1 = np.ascontiguousarray(img1, dtype=np.uint8).reshape(n, self.height, self.width)
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_frames_upload(self._h, sl.ctypes.data_as(C.c_void_p), n, a0.ctypes.data_as(C.c_void_p),
                                                      a1.ctypes.data_as(C.c_void_p), self.height * self.width, self._stream()))

    def step_frames(self, slot_of_stream, timestamps):
        """One step with stream s reading store entry slot_of_stream[s]; < 0 = no frame for that stream in this step."""
        S = self.n_streams
        sl = np.ascontiguousarray(slot_of_stream, dtype=np.int32)
        assert sl.shape == (S,)
        ts = (C.c_double * S)(*[float(t) for t in np.atleast_1d(timestamps)])
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_step_frames(self._h, sl.ctypes.data_as(C.c_void_p), ts, self._stream()))

    def read_features(self):
        """Synchronises; returns [(ids int64[n], uv float64[n,4])] per stream (u0, v0, u1, v1)."""
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_read_features(self._h, self._ids.ctypes.data_as(C.c_void_p),
                                                      self._uv.ctypes.data_as(C.c_void_p),
                                                      self._n.ctypes.data_as(C.c_void_p), self.max_features, self._stream()))
        return [(self._ids[s, :self._n[s]].copy(), self._uv[s, :self._n[s]].copy()) for s in range(self.n_streams)]

    def read_features_raw(self):
        """Synchronises; returns the engine's own host arrays (ids int64[S,cap], uv float64[S,cap,4],
        n int32[S]) without per-stream copies -- valid until the next read."""
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_read_features(self._h, self._ids.ctypes.data_as(C.c_void_p),
                                                      self._uv.ctypes.data_as(C.c_void_p),
                                                      self._n.ctypes.data_as(C.c_void_p), self.max_features, self._stream()))
        return self._ids, self._uv, self._n

    def features_dev(self):
        """(ids_ptr, uv_ptr, n_ptr, cap): device addresses of the feature message the last `step` published (int64[S,cap],
        float64[S,cap,4], int32[S]); rewritten by the next step.  For `BatchedMSCKF.submit_dev`."""
        ids, uv, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int(0)
        N.check(N.lib().av_frontend_features_dev(self._h, C.byref(ids), C.byref(uv), C.byref(n), C.byref(cap)))
        return ids, uv, n, int(cap.value)

    def read_features_begin(self, slot=0):
        """Enqueue the device-to-host copy of the features published by the last step into pinned slot 0/1 (returns
        at once).  Call `step` for the next frame before `read_features_end(slot)` to overlap the two."""
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_read_features_begin(self._h, int(slot), self._stream()))

    def read_features_end(self, slot=0):
        """Wait for the copy begun on `slot` and return NEW host arrays (ids int64[S,cap], uv float64[S,cap,4],
        n int32[S]); entries beyond n[s] are unspecified."""
        ids = np.empty((self.n_streams, self.max_features), np.int64)
        uv = np.empty((self.n_streams, self.max_features, 4), np.float64)
        n = np.empty(self.n_streams, np.int32)
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_read_features_end(self._h, int(slot), ids.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p),
                                                        n.ctypes.data_as(C.c_void_p), self.max_features))
        return ids, uv, n

    def read_grid(self, stream=0):
        cap = self.max_features
        ids = np.zeros(cap, np.int64); life = np.zeros(cap, np.int32); cell = np.zeros(cap, np.int32)
        pts = np.zeros((cap, 4), np.float32)
        n = C.c_int32(0); nid = C.c_int64(0)
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_read_grid(self._h, int(stream), ids.ctypes.data_as(C.c_void_p),
                                                  life.ctypes.data_as(C.c_void_p), cell.ctypes.data_as(C.c_void_p),
                                                  pts.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(nid), self._stream()))
        k = n.value
        return dict(ids=ids[:k], lifetime=life[:k], cell=cell[:k], cam0=pts[:k, :2].copy(), cam1=pts[:k, 2:].copy(),
                    next_feature_id=nid.value)

    def read_counters(self, stream=0):
        out = (C.c_int32 * 8)()
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_read_counters(self._h, int(stream), C.byref(out), self._stream()))
        return dict(zip(COUNTER_NAMES, [int(v) for v in out]))

    def read_match_counts(self, stream=0):
        """(candidates stereo-matched in round 1, in round 2) of the last step (lazy matching, airvision.h)."""
        out = (C.c_int32 * 2)()
        with torch.cuda.device(self.device):
            N.check(N.lib().av_frontend_read_match_counts(self._h, int(stream), C.byref(out), self._stream()))
        return int(out[0]), int(out[1])

    def enable_timing(self, max_spans):
        """Bracket every launch group with HIP events on the step's stream (bench roofline leg)."""
        N.check(N.lib().av_frontend_enable_timing(self._h, int(max_spans)))

    def read_timing(self):
        """{class: (total_ms, n_launch_groups)} since the last read; synchronises the device."""
        ms = (C.c_double * 4)(); n = (C.c_int32 * 4)()
        N.check(N.lib().av_frontend_read_timing(self._h, C.byref(ms), C.byref(n)))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(('pyramid', 'lk', 'fast', 'glue'))}

    def read_all_counters(self):
        return [self.read_counters(s) for s in range(self.n_streams)]
