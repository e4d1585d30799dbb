"""EuRoC MAV dataset reader + deterministic replay.

Counterpart of the reference's `streaming/dataset.py:12-219` (CSV/PNG readers, namedtuple messages,
start-time offset) and of `streaming/publisher.py:8-53` -- except that the wall-clock-paced publisher
threads are replaced by the deterministic sequential replay of SURVEY section 3.5 (IMU messages with
timestamp <= t are delivered before the stereo frame at t), which is what a throughput run needs and
what makes results reproducible.  PNGs are decoded with Pillow (the reference uses cv2.imread(path, -1),
dataset.py:110; OpenCV is not a dependency here).
"""
import os
from collections import namedtuple

import numpy as np

imu_msg = namedtuple('imu_msg', ['timestamp', 'angular_velocity', 'linear_acceleration'])
img_msg = namedtuple('img_msg', ['timestamp', 'image'])
stereo_msg = namedtuple('stereo_msg', ['timestamp', 'cam0_image', 'cam1_image', 'cam0_msg', 'cam1_msg'])
gt_msg = namedtuple('gt_msg', ['timestamp', 'p', 'q', 'v', 'bw', 'ba'])


def read_image(path):
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im)
    if a.ndim == 3:
        a = a[..., 0]
    if a.dtype != np.uint8:                      # cv2.imread(path, -1) of the reference would hand a uint16 array on: never truncate it silently
        raise ValueError('%s: %s image, expected uint8 (8-bit greyscale camera frames)' % (path, a.dtype))
    return np.ascontiguousarray(a)


class EuRoCDataset(object):
    """`EuRoCDataset(path)` with `.imu`, `.stereo`, `.groundtruth` iterables and
    `.set_starttime(offset)` (reference: dataset.py:189-219)."""

    def __init__(self, path):
        self.path = path
        m = os.path.join(path, 'mav0')
        self._imu_csv = os.path.join(m, 'imu0', 'data.csv')
        self._gt_csv = os.path.join(m, 'state_groundtruth_estimate0', 'data.csv')
        self._cam = [self._list_images(os.path.join(m, 'cam%d' % c, 'data')) for c in (0, 1)]
        self.timestamps = self._cam[0][1]
        self._imu = self._load_csv(self._imu_csv, 7)
        self._gt = self._load_csv(self._gt_csv, 17) if os.path.exists(self._gt_csv) else np.zeros((0, 17))
        # dataset.py:203: starttime = max(first imu, first stereo)
        self.starttime0 = max(self._imu[0, 0] if len(self._imu) else -np.inf, self.timestamps[0] if self.timestamps else -np.inf)
        self.starttime = self.starttime0

    @staticmethod
    def _list_images(d):
        names = sorted((n for n in os.listdir(d) if n.endswith('.png')), key=lambda n: float(n[:-4]))
        return [os.path.join(d, n) for n in names], [float(n[:-4]) * 1e-9 for n in names]

    @staticmethod
    def _load_csv(path, ncol):
        rows = []
        with open(path) as f:
            next(f)
            for line in f:
                v = [float(x) for x in line.strip().split(',')]
                if len(v) >= ncol:
                    rows.append(v[:ncol])
        a = np.array(rows, dtype=np.float64).reshape(-1, ncol)
        if len(a):
            a[:, 0] *= 1e-9                              # ns -> s (dataset.py:29,67)
        return a

    def set_starttime(self, offset):
        """dataset.py:206-214: everything before starttime0 + offset is skipped."""
        self.starttime = self.starttime0 + float(offset)

    @property
    def imu(self):
        for r in self._imu:
            if r[0] >= self.starttime:
                yield imu_msg(r[0], r[1:4].copy(), r[4:7].copy())

    @property
    def groundtruth(self):
        for r in self._gt:
            if r[0] >= self.starttime:
                yield gt_msg(r[0], r[1:4].copy(), r[4:8].copy(), r[8:11].copy(), r[11:14].copy(), r[14:17].copy())

    @property
    def stereo(self):
        (p0, t0), (p1, _t1) = self._cam
        for k in range(min(len(p0), len(p1))):
            if t0[k] < self.starttime:
                continue
            i0, i1 = read_image(p0[k]), read_image(p1[k])
            yield stereo_msg(t0[k], i0, i1, img_msg(t0[k], i0), img_msg(t0[k], i1))

    @property
    def stereo_files(self):
        """(timestamp, cam0 path, cam1 path) of every frame from the start time on -- what `stereo` decodes; the batch
        stager (`FrameStager`) decodes these on host threads one step ahead instead."""
        (p0, t0), (p1, _t1) = self._cam
        for k in range(min(len(p0), len(p1))):
            if t0[k] >= self.starttime:
                yield t0[k], p0[k], p1[k]

    def groundtruth_array(self):
        """float64[n, 8]: t, p(3), q as stored by EuRoC (w, x, y, z)."""
        g = self._gt[self._gt[:, 0] >= self.starttime]
        return g[:, :8].copy()


def decode_batch(paths, out, threads=16):
    """Decode len(paths) greyscale PNG files into out[i] (uint8 [n, h, w], C-contiguous) on `threads` host threads with the
    library's decoder (av_png_decode_gray8: zlib + PNG row filters in C++, no GIL); a file of another PNG flavour (16-bit,
    RGB, palette ...) is decoded by Pillow.  paths[i] = None leaves out[i] untouched."""
    import ctypes as C
    from . import _native as N
    idx = [i for i, p in enumerate(paths) if p is not None]
    if not idx:
        return
    n, h, w = out.shape
    assert out.flags['C_CONTIGUOUS'] and out.dtype == np.uint8
    arr = (C.c_char_p * len(idx))(*[os.fsencode(paths[i]) for i in idx])
    status = (C.c_int32 * len(idx))()
    if len(idx) == n:
        dst, tmp = out, None
    else:
        tmp = np.empty((len(idx), h, w), np.uint8); dst = tmp
    rc = N.lib().av_png_decode_gray8(arr, len(idx), w, h, dst.ctypes.data_as(C.c_void_p), h * w, int(threads), status)
    if rc == N.AV_E_CAPACITY:                 # some file is not 8-bit greyscale: that file alone goes through Pillow
        for k, i in enumerate(idx):
            if status[k] == 1:
                im = np.asarray(read_image(paths[i]))
                if im.dtype != np.uint8 or im.shape != (h, w):         # e.g. a 16-bit PNG: never truncated silently
                    raise ValueError('%s: %s %s image, expected uint8 %s' % (paths[i], im.dtype, im.shape, (h, w)))
                dst[k] = im
            elif status[k] != 0:
                N.check(N.AV_E_INVALID)
    else:
        N.check(rc)
    if tmp is not None:
        out[idx] = tmp


class FrameStager(object):
    """Batch staging for sweeps (reference: the reader threads of streaming/dataset.py:93-158): the frames of
    step k+1 of all S streams are decoded on host threads while step k runs on the GPU.  `next()` returns
    (timestamps float64[S] (-1 = stream finished), img0 uint8[S,h,w], img1) of the next step, or None when every stream
    is done; the arrays stay valid until the call after the next one (three buffers: one being filled by the
    loader thread, the one just returned, the one returned before it)."""

    def __init__(self, datasets, height, width, max_frames=None, threads=16):
        from concurrent.futures import ThreadPoolExecutor
        self.S = len(datasets)
        self.its = [iter(d.stereo_files) for d in datasets]
        self.buf = [np.zeros((2, self.S, height, width), np.uint8) for _ in range(3)]      # [slot][camera][stream]
        self.max_frames, self.threads, self.k = max_frames, threads, 0
        self.pool = ThreadPoolExecutor(1)
        self.fut = self.pool.submit(self._load, 0)

    def _load(self, k):
        if self.max_frames is not None and k >= self.max_frames:
            return None
        ent = [next(it, None) for it in self.its]
        if all(e is None for e in ent):
            return None
        ts = np.array([-1.0 if e is None else e[0] for e in ent])
        buf = self.buf[k % 3]
        both = [None if e is None else e[1] for e in ent] + [None if e is None else e[2] for e in ent]
        decode_batch(both, buf.reshape(2 * self.S, buf.shape[2], buf.shape[3]), self.threads)     # cam0 and cam1 of all streams in one threaded call
        for s, e in enumerate(ent):
            if e is None:
                buf[:, s] = 0                    # a finished stream idles on blank images
        return ts, buf[0], buf[1]

    def next(self):
        res = self.fut.result()
        self.k += 1
        self.fut = self.pool.submit(self._load, self.k) if res is not None else None
        return res

    def close(self):
        if self.fut is not None:
            try:
                self.fut.result()
            except Exception:
                pass
        self.pool.shutdown(wait=True)


class SharedFramePlan(object):
    """Which DISTINCT frames a batch of streams reads at which step, and where each lives in the front-end's device frame store
    (`FrontendEngine.frames_*`).  The reference's sweep replays one sequence from several start offsets (run.bat:4-12; an offset
    only moves the start index, dataset.py:206-214): stream i reads frame start_i + k at step k, so every frame is read by every
    offset stream a few steps apart -- here it is decoded, uploaded, pyramided and FAST-scanned once, when the first of them
    reaches it, and dropped when the last has tracked away from it.  Streams of different sequences simply share nothing.

    Frames are keyed by their (cam0 path, cam1 path).  Computed up front from the datasets' file lists:
      n_steps, n_slots            steps of the longest stream; store entries needed (the most frames ever live at once)
      slots[k]    int32[S]        store entry stream s reads at step k; -1 = the stream is over
      ts[k]       float64[S]      frame time; -1 = over
      new[k]      [(entry, cam0 path, cam1 path)]   frames first read at step k: upload them before the step (at the earliest
                                  right before step k - 1 is enqueued: an entry is handed out again only two steps after its last reader)
    A frame is read as the current image at step k and as the previous image at step k + 1 of the same stream."""

    def __init__(self, datasets, max_frames=None):
        files = []
        for d in datasets:
            f = list(d.stereo_files)
            files.append(f if max_frames is None else f[:max_frames])
        S = self.S = len(files)
        self.n_steps = max([len(f) for f in files] + [0])
        last_need = {}
        for f in files:
            for k, (_t, p0, p1) in enumerate(f):
                need = k + 1 if k + 1 < len(f) else k
                key = (p0, p1)
                if last_need.get(key, -1) < need:
                    last_need[key] = need
        self.slots = np.full((self.n_steps, S), -1, np.int32)
        self.ts = np.full((self.n_steps, S), -1.0)
        self.new = [[] for _ in range(self.n_steps)]
        import heapq
        free, where, n_slots = [], {}, 0             # free: heap of (first step the entry may be rewritten for, entry)
        for k in range(self.n_steps):
            for s, f in enumerate(files):
                if k >= len(f):
                    continue
                t, p0, p1 = f[k]
                key = (p0, p1)
                e = where.get(key)
                if e is None:
                    if free and free[0][0] <= k:
                        e = heapq.heappop(free)[1]
                    else:
                        e = n_slots; n_slots += 1
                    where[key] = e
                    self.new[k].append((e, p0, p1))
                self.slots[k, s] = e
                self.ts[k, s] = t
            for key in [key for key, e in where.items() if last_need[key] == k]:
                heapq.heappush(free, (k + 2, where.pop(key)))
        self.n_slots = max(n_slots, 1)
        self.n_frames_distinct = sum(len(n) for n in self.new)
        self.n_stream_frames = int((self.slots >= 0).sum())


class SharedFrameStager(object):
    """Decodes the NEW frames of every step of a `SharedFramePlan` on host threads, a few steps ahead of the GPU (reference: the
    reader threads of streaming/dataset.py:93-158).  `get(k)` -> (entries int32[n], img0 uint8[n,h,w], img1) of step k; steps must be
    asked for in order."""

    def __init__(self, plan, height, width, threads=16, ahead=3):
        from concurrent.futures import ThreadPoolExecutor
        self.plan, self.h, self.w, self.threads, self.ahead = plan, height, width, threads, ahead
        self.pool = ThreadPoolExecutor(1)
        self.futs = {}
        self.submitted = 0
        self._fill(0)

    def _decode(self, k):
        new = self.plan.new[k]
        n = len(new)
        buf = np.empty((2 * n, self.h, self.w), np.uint8)
        if n:
            decode_batch([e[1] for e in new] + [e[2] for e in new], buf, self.threads)
        return np.array([e[0] for e in new], np.int32), buf[:n], buf[n:]

    def _fill(self, k):
        while self.submitted < min(self.plan.n_steps, k + 1 + self.ahead):
            self.futs[self.submitted] = self.pool.submit(self._decode, self.submitted)
            self.submitted += 1

    def get(self, k):
        self._fill(k)
        res = self.futs.pop(k).result()
        self._fill(k + 1)
        return res

    def close(self):
        for f in self.futs.values():
            try:
                f.result()
            except Exception:
                pass
        self.pool.shutdown(wait=True)


def replay(dataset, imu_sinks, on_stereo, max_frames=None):
    """Deterministic replay: for each stereo frame at time t deliver every IMU message with
    timestamp <= t to each sink (order of vio.py:43-44), then call on_stereo(msg)."""
    it = iter(dataset.imu)
    pending = next(it, None)
    n = 0
    for msg in dataset.stereo:
        while pending is not None and pending.timestamp <= msg.timestamp:
            for sink in imu_sinks:
                sink(pending)
            pending = next(it, None)
        on_stereo(msg)
        n += 1
        if max_frames is not None and n >= max_frames:
            break
    return n


def write_euroc_layout(root, stream, groundtruth_rate_hz=200.0, frame_range=None, write_csv=True, compress_level=6):
    """Write a seeded synthetic stream (uav_airvision_amd.synth.SyntheticStream) as an EuRoC-layout directory
    `root/mav0/{cam0,cam1}/data/<ns>.png`, `imu0/data.csv`, `state_groundtruth_estimate0/data.csv`, so that the same
    reader / replay / sweep code that runs on the real dataset (which is not redistributable and not present on the build
    or GPU boxes) can be exercised end to end.  PNG is lossless: the reader returns the rendered pixels bit for bit.
    Ground truth = the analytic trajectory of the stream (position of the IMU frame; identity orientation columns).
    `frame_range=(a, b)` writes only the images of frames a..b-1 (several writer processes can share one sequence);
    `write_csv=False` skips the IMU / ground-truth files; `compress_level` is zlib's (any level is lossless; 1 writes 5x faster)."""
    from PIL import Image
    for cam in ('cam0', 'cam1'):
        os.makedirs(os.path.join(root, 'mav0', cam, 'data'), exist_ok=True)
    a, b = (0, stream.n_frames) if frame_range is None else frame_range
    for k in range(a, b):
        m = stream.frame(k)
        name = '%d.png' % int(round(m.timestamp * 1e9))
        Image.fromarray(m.cam0_image).save(os.path.join(root, 'mav0', 'cam0', 'data', name), compress_level=compress_level)
        Image.fromarray(m.cam1_image).save(os.path.join(root, 'mav0', 'cam1', 'data', name), compress_level=compress_level)
    if not write_csv:
        return root
    os.makedirs(os.path.join(root, 'mav0', 'imu0'), exist_ok=True)
    with open(os.path.join(root, 'mav0', 'imu0', 'data.csv'), 'w') as f:
        f.write('#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\n')
        for m in stream.imu:
            f.write('%d,%.12f,%.12f,%.12f,%.12f,%.12f,%.12f\n' % (int(round(m.timestamp * 1e9)), *m.angular_velocity, *m.linear_acceleration))
    os.makedirs(os.path.join(root, 'mav0', 'state_groundtruth_estimate0'), exist_ok=True)
    with open(os.path.join(root, 'mav0', 'state_groundtruth_estimate0', 'data.csv'), 'w') as f:
        f.write('#timestamp,p_RS_R_x [m],p_RS_R_y [m],p_RS_R_z [m],q_RS_w [],q_RS_x [],q_RS_y [],q_RS_z [],v_RS_R_x,v_RS_R_y,v_RS_R_z,bw_x,bw_y,bw_z,ba_x,ba_y,ba_z\n')
        t_begin, t_end = stream.imu[0].timestamp, stream.imu[-1].timestamp
        n = int((t_end - t_begin) * groundtruth_rate_hz) + 1
        for i in range(n):
            t = t_begin + i / groundtruth_rate_hz
            p = stream.position(t)
            f.write(','.join(['%d' % int(round(t * 1e9))] + ['%.9f' % v for v in p] + ['1', '0', '0', '0'] + ['0'] * 9) + '\n')
    return root
