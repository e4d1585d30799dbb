"""Several independent pipelines on one GPU: `EngineSet` / `FilterSet` split a batch of S streams into P contiguous parts, each with
its own `FrontendEngine`, its own `BatchedMSCKF`, its own host thread and its own HIP streams, behind the interface of ONE engine / ONE
filter (the methods `bench.py` and `sweep.py` call).

Why (profiles/r05/README.md, `scripts/two_engines.py`): an engine's step is one in-order chain of ~20 launches on one HIP stream --
every kernel boundary drains the machine before the next kernel fills it, and the chain's kernels are bound by different things (LK by
VALU issue, the pyramids by HBM, the glue kernels by latency).  Independent pipelines fill each other's gaps: FRONT-END ALONE 218.5 ->
229.3 k frames/s with two pipelines of the same 2,048 streams (`bench.py --frontend-only --pipelines 2`; 233-236 k in the front-end-only leg
that follows a complete-path run; four pipelines: 215-236 k, no better than two).  In the complete path
the batched filter's kernels already are that second tenant: there one pipeline stays the best (174.3 k against 160.6 / 167.5 / 165.5-170.4 k
with 2 / 3 / 4), which is why `--pipelines` defaults to 1.  Same kernels, same results per stream -- streams never interact
(`modules/vio.py` runs one pipeline per process).

Calls that only ENQUEUE (`push_imu_batch`, `step`, `prestage`, `push_imu`, `submit_dev`) are handed to the parts' worker threads and
return at once; every call that reads something back drains the workers first."""
import queue
import threading

import numpy as np
import torch

from . import _native as N
from .frontend import FrontendEngine


class _Pipe(threading.Thread):
    """Worker of one part: runs the closures it is handed in order, on its own HIP stream."""

    def __init__(self, device):
        threading.Thread.__init__(self, daemon=True)
        self.device = device
        self.q = queue.Queue()
        self.err = None
        self.stream = torch.cuda.Stream(device=device)       # the front-end's launches
        self.fstream = torch.cuda.Stream(device=device)      # handed to the filter (a batch of one group launches from the caller's stream)
        self.start()

    def run(self):
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self.stream):
            while True:
                fn = self.q.get()
                try:
                    if fn is None:
                        return
                    if self.err is None:
                        fn()
                except BaseException as e:                   # surfaces at the next drain
                    self.err = e
                finally:
                    self.q.task_done()

    def put(self, fn):
        self.q.put(fn)

    def drain(self):
        self.q.join()
        if self.err is not None:
            e, self.err = self.err, None
            raise e

    def stop(self):
        self.q.put(None)
        self.join()


def split_parts(n_streams, n_parts):
    """Contiguous, equal (+-1) parts [(lo, hi)] of range(n_streams)."""
    n_parts = max(1, min(int(n_parts), int(n_streams)))
    edges = [n_streams * p // n_parts for p in range(n_parts + 1)]
    return [(edges[p], edges[p + 1]) for p in range(n_parts)]


class EngineSet(object):
    def __init__(self, config, n_streams, n_parts, device=0, **kw):
        self.device = device
        self.n_streams = int(n_streams)
        self.parts = split_parts(n_streams, n_parts)
        self.pipes = [_Pipe(torch.device('cuda', device)) for _ in self.parts]
        self.engs = [FrontendEngine(config, n_streams=hi - lo, device=device, **kw) for lo, hi in self.parts]
        self.max_features = self.engs[0].max_features
        self.height, self.width = self.engs[0].height, self.engs[0].width

    # ---- enqueue-only calls -------------------------------------------------------------------------------------------------------
    def _split_rows(self, idx, *cols):
        idx = np.asarray(idx)
        out = []
        for lo, hi in self.parts:
            m = (idx >= lo) & (idx < hi)
            out.append((np.ascontiguousarray(idx[m] - lo, dtype=np.int32),) + tuple(np.ascontiguousarray(np.asarray(c)[m]) for c in cols))
        return out

    def push_imu_batch(self, stream_idx, timestamps, gyro):
        for p, rows in enumerate(self._split_rows(stream_idx, timestamps, gyro)):
            if len(rows[0]):
                self.pipes[p].put(lambda e=self.engs[p], r=rows: e.push_imu_batch(*r))

    def step(self, img0, img1, timestamps):
        ts = list(timestamps)
        for p, (lo, hi) in enumerate(self.parts):
            self.pipes[p].put(lambda e=self.engs[p], a=img0[lo:hi], b=img1[lo:hi], t=ts[lo:hi]: e.step(a, b, t))

    def prestage(self, img0, img1):
        for p, (lo, hi) in enumerate(self.parts):
            self.pipes[p].put(lambda e=self.engs[p], a=img0[lo:hi], b=img1[lo:hi]: e.prestage(a, b))

    # ---- calls that read back ------------------------------------------------------------------------------------------------------
    def drain(self):
        for w in self.pipes:
            w.drain()

    def _on_pipe(self, p, fn):
        """Run fn on part p's worker (its HIP stream) and return the result."""
        box = []
        self.pipes[p].put(lambda: box.append(fn()))
        self.pipes[p].drain()
        return box[0]

    def read_features(self):
        self.drain()
        out = []
        for p, e in enumerate(self.engs):
            out.extend(self._on_pipe(p, e.read_features))
        return out

    def read_all_counters(self):
        self.drain()
        out = []
        for p, e in enumerate(self.engs):
            out.extend(self._on_pipe(p, e.read_all_counters))
        return out

    def _locate(self, stream):
        for p, (lo, hi) in enumerate(self.parts):
            if lo <= stream < hi:
                return p, stream - lo
        raise IndexError(stream)

    def read_match_counts(self, stream=0):
        self.drain()
        p, s = self._locate(stream)
        return self._on_pipe(p, lambda: self.engs[p].read_match_counts(s))

    def read_counters(self, stream=0):
        self.drain()
        p, s = self._locate(stream)
        return self._on_pipe(p, lambda: self.engs[p].read_counters(s))

    def enable_timing(self, max_spans):
        self.drain()
        for e in self.engs:
            e.enable_timing(max_spans)

    def read_timing(self):
        """{class: (total_ms, n_launch_groups)} summed over the parts (their launches run side by side: the sum is GPU time of the class,
        not wall time)."""
        self.drain()
        tot = {}
        for e in self.engs:
            for k, (ms, n) in e.read_timing().items():
                a = tot.get(k, (0.0, 0))
                tot[k] = (a[0] + ms, a[1] + n)
        return tot

    def close(self):
        try:
            self.drain()
        finally:
            for e in self.engs:
                e.close()
            for w in self.pipes:
                w.stop()


class _OutView(object):
    """The [S, 12] output of a FilterSet step: the parts' arrays side by side, filled once the step has retired."""

    def __init__(self, n_parts):
        self.parts = [None] * n_parts

    def array(self):
        return np.concatenate(self.parts, axis=0)

    def __getitem__(self, key):
        return self.array()[key]


class FilterSet(object):
    """One BatchedMSCKF per part of an EngineSet, driven by the same workers (a part's front-end step, its prestage and its filter
    hand-over stay in order on the part's thread)."""

    def __init__(self, config, engine_set, device=0, **kw):
        from .msckf_ops import BatchedMSCKF
        self.es = engine_set
        self.parts = engine_set.parts
        self.flts = [BatchedMSCKF(config, hi - lo, device=device, **kw) for lo, hi in self.parts]

    def device_resident(self):
        return all(f.device_resident() for f in self.flts)

    def push_imu(self, stream_idx, timestamps, gyro, acc):
        for p, rows in enumerate(self.es._split_rows(stream_idx, timestamps, gyro, acc)):
            if len(rows[0]):
                self.es.pipes[p].put(lambda f=self.flts[p], r=rows: f.push_imu(*r))

    def submit_dev(self, engine_set, timestamps, msg_stream=None):
        assert engine_set is self.es
        ts = np.asarray(timestamps, dtype=np.float64)
        out = _OutView(len(self.parts))
        for p, (lo, hi) in enumerate(self.parts):
            def job(p=p, t=np.ascontiguousarray(ts[lo:hi])):
                w = self.es.pipes[p]
                ms = N.current_stream()                      # the part's front-end stream: the message is copied behind its step
                with torch.cuda.stream(w.fstream):
                    out.parts[p] = self.flts[p].submit_dev(self.es.engs[p], t, msg_stream=ms)
            self.es.pipes[p].put(job)
        return out

    def wait(self, max_pending=0):
        self.es.drain()
        for f in self.flts:
            f.wait(max_pending)

    def counters(self):
        self.wait(0)
        cs = [f.counters() for f in self.flts]
        out = {}
        for k in cs[0]:
            v = [c[k] for c in cs]
            out[k] = min(v) if k.startswith('min_') else (max(v) if (k.startswith('max_') or k == 'steps') else sum(v))
        return out

    def work(self, enable=-1):
        self.wait(0)
        ws = [f.work(enable) for f in self.flts]
        return {k: sum(w[k] for w in ws) for k in ws[0]}

    def work_executed(self):
        self.wait(0)
        ws = [f.work_executed() for f in self.flts]
        return {k: sum(w[k] for w in ws) for k in ws[0]}

    def stream_status(self, s):
        p, i = self.es._locate(s)
        return self.flts[p].stream_status(i)

    def close(self):
        try:
            self.wait(0)
        finally:
            for f in self.flts:
                f.close()
