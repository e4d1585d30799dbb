"""Sequence x offset sweep runner on the batched throughput path.

The reference's `run.bat:4-12` runs 9 sequences x 7 start offsets serially through `main.py --path DIR --offset S`
(`main.py:10-34`, `streaming/dataset.py:189-214`).  Here every (sequence, offset) pair is one independent stream; the
pairs are partitioned over ranks (`shard.partition`, one process per GPU) and each rank steps ALL of its streams
together: one `FrontendEngine(n_streams=k)` (one batched kernel launch per stage and step for the k streams) feeding one
`BatchedMSCKF(k)` -- the same two objects bench.py times -- driven by the deterministic replay of SURVEY 3.5 (IMU
messages with timestamp <= t before the frame at t).  Sequences end at different frames; a finished stream idles
(blank images, frame timestamp -1 for the filter) until the longest one is done.  Trajectories are gathered at the end
(`shard.gather_trajectories`); there is no per-frame communication.

    python -m uav_airvision_amd.sweep --root /data/euroc --sequences MH_01_easy MH_03_medium --offsets 0 10 20
    python -m torch.distributed.run --nproc-per-node 8 -m uav_airvision_amd.sweep ...
    python -m uav_airvision_amd.sweep --make-synthetic /tmp/syn --sequences SYN_01 SYN_02 --frames 200 ...   (no dataset at hand)
"""
import argparse
import json
import os
import sys

import numpy as np


def run_stream(config, dataset_path, offset, device=0, max_frames=None):
    """ONE (sequence, offset) stream through the single-stream drop-in classes (`ImageProcessor` + `MSCKF`, the objects
    `modules/vio.py` constructs); returns (float64[n, 8] trajectory `t px py pz qx qy qz qw`, dataset).  The sweep itself
    uses `run_batched`; this is the reference-shaped path kept for comparison and for `vio.py`-style callers."""
    here = os.path.dirname(os.path.abspath(__file__))
    dropin = os.path.join(here, 'dropin')
    if dropin not in sys.path:
        sys.path.insert(0, dropin)
    from image_processing import ImageProcessor
    from msckf import MSCKF
    from .euroc import EuRoCDataset, replay
    ds = EuRoCDataset(dataset_path)
    ds.set_starttime(offset)
    ip = ImageProcessor(config, device=device)
    flt = MSCKF(config, device=device, write_trajectory=False)
    traj = []

    def on_stereo(msg):
        feat = ip.stereo_callback(msg)
        if feat and flt.feature_callback(feat) is not None:
            s = flt.state_server.imu_state
            traj.append([s.timestamp, *s.position, *s.orientation])
    replay(ds, [ip.imu_callback, flt.imu_callback], on_stereo, max_frames)
    ip.close(); flt.close()
    return np.array(traj, dtype=np.float64).reshape(-1, 8), ds


class BatchedRunner(object):
    """k streams stepped together on one GPU: `FrontendEngine(n_streams=k)` -> `BatchedMSCKF(k)`.

    `datasets`: objects with `.imu` and `.stereo` iterables of the reference's message namedtuples
    (`euroc.EuRoCDataset` after `set_starttime`, or anything shaped like it).  `run()` returns one float64[n, 8]
    trajectory per stream in the reference's output-line layout (msckf.py:152-158: t = imu_state.timestamp).
    `on_step(step, timestamps, ids, uv, n_feat, out)` -- optional, called once per step with the published feature
    arrays of the front-end and the filter's float64[k, 12] output of the same step (parity tests); without it the
    filter runs one step behind the front-end (queued), the way bench.py pipelines them."""

    def __init__(self, config, n_streams, device=0, rows_cap=None):
        from .frontend import FrontendEngine
        from .msckf_ops import BatchedMSCKF
        self.config, self.S, self.device = config, int(n_streams), int(device)
        self.eng = FrontendEngine(config, n_streams=self.S, device=self.device)
        self.flt = BatchedMSCKF(config, self.S, device=self.device, rows_cap=rows_cap, max_features=self.eng.max_features)
        self.frames_done = np.zeros(self.S, np.int64)
        self._fstream = None
        self.plan, self.steps_done = None, 0

    def close(self):
        self.eng.close(); self.flt.close()

    def run(self, datasets, max_frames=None, on_step=None, host_threads=16, share_frames=True):
        """share_frames (datasets that list their files): every distinct frame is decoded, uploaded, pyramided and FAST-scanned
        once and kept in the engine's device frame store while any stream still reads it (`euroc.SharedFramePlan`); finished
        streams launch nothing.  False: round 4's per-stream staging (`euroc.FrameStager`), kept for A/B measurements."""
        from .euroc import FrameStager, SharedFramePlan, SharedFrameStager
        staged = all(hasattr(d, 'stereo_files') for d in datasets)
        self.plan = None
        stager = None
        if staged and share_frames:
            self.plan = SharedFramePlan(datasets, max_frames=max_frames)
            self.eng.frames_reserve(self.plan.n_slots)
            stager = SharedFrameStager(self.plan, self.eng.height, self.eng.width, threads=host_threads)
        elif staged:
            stager = FrameStager(datasets, self.eng.height, self.eng.width, max_frames=max_frames, threads=host_threads)
        try:
            return self._run(datasets, stager, max_frames, on_step)
        finally:
            if stager is not None:
                stager.close()

    def _run(self, datasets, stager, max_frames, on_step):
        S = self.S
        assert len(datasets) == S
        eng, flt = self.eng, self.flt
        plan = self.plan
        staged = stager is not None
        # frames: decoded ahead on host threads when the datasets list their files (reference: the reader threads of
        # streaming/dataset.py:93-158); any other dataset-shaped object is read through `.stereo`
        its_img = None if staged else [iter(d.stereo) for d in datasets]
        # IMU: datasets that hold their samples as one array (EuRoCDataset) are sliced per frame; others are iterated
        arr = [getattr(d, '_imu', None) if hasattr(d, 'starttime') else None for d in datasets]
        pos = [0 if a is None else int(np.searchsorted(a[:, 0], d.starttime, 'left')) for a, d in zip(arr, datasets)]
        its_imu = [iter(d.imu) if a is None else None for a, d in zip(arr, datasets)]
        pend = [None if it is None else next(it, None) for it in its_imu]
        img0 = np.zeros((S, eng.height, eng.width), np.uint8) if plan is None else None
        img1 = np.zeros_like(img0) if plan is None else None
        traj = [[] for _ in range(S)]
        last_t = np.zeros(S)
        inflight = []                                     # outputs of submitted steps that have not been recorded yet
        dev = flt.device_resident()

        def record(out):
            for s in np.nonzero(out[:, 0] > 0.5)[0]:
                traj[s].append(out[s, 1:9].copy())

        if plan is not None and plan.n_steps:
            eng.frames_upload(*stager.get(0))
        step = 0
        while True:
            if plan is not None:
                if step >= plan.n_steps:
                    break
                if step + 1 < plan.n_steps:               # the NEXT step's new frames go up before this step is enqueued: copy,
                    eng.frames_upload(*stager.get(step + 1))      # pyramids and FAST of step k+1 run beside the kernels of step k
                frame_ts = plan.ts[step]
            elif staged:
                nxt = stager.next()
                if nxt is None:
                    break
                frame_ts, img0, img1 = nxt
            else:
                msgs = [next(it, None) for it in its_img] if (max_frames is None or step < max_frames) else [None] * S
                if all(m is None for m in msgs):
                    break
                frame_ts = np.array([-1.0 if m is None else m.timestamp for m in msgs])
                for s, m in enumerate(msgs):
                    if m is None:
                        its_img[s] = iter(())
                        img0[s] = 0; img1[s] = 0
                    else:
                        img0[s] = m.cam0_image; img1[s] = m.cam1_image
            ts_f = np.full(S, -1.0)                       # the filter's frame times: < 0 = no frame for this stream
            ts_e = np.empty(S)
            idx, tt, gy, ac = [], [], [], []
            for s in range(S):
                if frame_ts[s] < 0:                       # finished: idles (shared store: launches nothing) until the longest stream ends
                    last_t[s] += 0.05
                    ts_e[s] = last_t[s]
                    continue
                t = float(frame_ts[s])
                ts_f[s] = ts_e[s] = last_t[s] = t
                self.frames_done[s] += 1
                a = arr[s]
                if a is not None:                         # SURVEY 3.5 / vio.py:43-44: every IMU message with timestamp <= t first
                    e = int(np.searchsorted(a[:, 0], t, 'right'))
                    if e > pos[s]:
                        idx.append(np.full(e - pos[s], s, np.int32)); tt.append(a[pos[s]:e, 0]); gy.append(a[pos[s]:e, 1:4]); ac.append(a[pos[s]:e, 4:7])
                        pos[s] = e
                else:
                    n0 = len(tt)
                    while pend[s] is not None and pend[s].timestamp <= t:
                        tt.append(np.array([pend[s].timestamp])); gy.append(np.asarray(pend[s].angular_velocity, np.float64).reshape(1, 3))
                        ac.append(np.asarray(pend[s].linear_acceleration, np.float64).reshape(1, 3))
                        pend[s] = next(its_imu[s], None)
                    if len(tt) > n0:
                        idx.append(np.full(len(tt) - n0, s, np.int32))
            if idx:
                idx, tt, gy, ac = np.concatenate(idx), np.concatenate(tt), np.concatenate(gy), np.concatenate(ac)
                eng.push_imu_batch(idx, tt, gy)
                flt.push_imu(idx, tt, gy, ac)
            if plan is not None:
                eng.step_frames(plan.slots[step], ts_e)
            else:
                eng.step_host(img0, img1, ts_e)
            if on_step is not None:
                ids, uv, n = eng.read_features_raw()
                n[ts_f < 0] = 0
                out = flt.step(ids, uv, n, ts_f)
                record(out)
                on_step(step, ts_f, ids, uv, n, out)
            elif dev:
                # the filter reads the published message where it lies on the device; nothing of this step is waited for here:
                # the next frames are decoded, uploaded and tracked while the filter's groups work through their queues
                # (a stream without a frame carries timestamp -1: its message is ignored)
                import torch
                from . import _native as N
                if self._fstream is None:
                    self._fstream = torch.cuda.Stream(device=self.device)
                ms = N.current_stream()                   # the stream the front-end's step was enqueued on
                with torch.cuda.stream(self._fstream):    # the filter's kernels (one group: here) overlap the next frame's front-end
                    inflight.append(flt.submit_dev(eng, ts_f, msg_stream=ms))
                flt.wait(2)
                while len(inflight) > 2:
                    record(inflight.pop(0))
            else:
                eng.read_features_begin(step & 1)
                ids, uv, n = eng.read_features_end(step & 1)
                n[ts_f < 0] = 0
                inflight.append(flt.submit(ids, uv, n, ts_f))
                flt.wait(1)                               # at most one step behind: the next frames are decoded meanwhile
                while len(inflight) > 1:
                    record(inflight.pop(0))
            step += 1
        flt.wait(0)
        for out in inflight:
            record(out)
        self.steps_done = step
        return [np.array(t, dtype=np.float64).reshape(-1, 8) for t in traj]


def run_batched(config, dataset_paths, offsets, device=0, max_frames=None, on_step=None, rows_cap=None, share_frames=True, stats=None):
    """Open (path, offset) pairs as EuRoC datasets and run them as one batch; returns (trajectories, datasets).
    stats (a dict) receives the batch's stream-frames, steps, distinct frames decoded and the seconds its stepping loop took."""
    import time
    from .euroc import EuRoCDataset
    dss = []
    for p, o in zip(dataset_paths, offsets):
        ds = EuRoCDataset(p)
        ds.set_starttime(o)
        dss.append(ds)
    r = BatchedRunner(config, len(dss), device=device, rows_cap=rows_cap)
    try:
        t0 = time.perf_counter()
        trajs = r.run(dss, max_frames=max_frames, on_step=on_step, share_frames=share_frames)
        if stats is not None:
            stats['seconds'] = stats.get('seconds', 0.0) + time.perf_counter() - t0
            stats['stream_frames'] = stats.get('stream_frames', 0) + int(r.frames_done.sum())
            stats['steps'] = stats.get('steps', 0) + int(r.steps_done)
            stats['frames_decoded'] = stats.get('frames_decoded', 0) + (r.plan.n_frames_distinct if r.plan is not None else int(r.frames_done.sum()))
            stats['store_entries'] = max(stats.get('store_entries', 0), r.plan.n_slots if r.plan is not None else 0)
    finally:
        r.close()
    return trajs, dss


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--root', help='directory holding the EuRoC sequences')
    ap.add_argument('--make-synthetic', metavar='DIR', help='first write EuRoC-layout synthetic sequences (one per --sequences name) into DIR and use it as --root')
    ap.add_argument('--frames', type=int, default=200, help='frames per synthetic sequence (--make-synthetic)')
    ap.add_argument('--sequences', nargs='+', required=True)
    ap.add_argument('--offsets', nargs='+', type=float, default=[0.0])
    ap.add_argument('--max-frames', type=int, default=None)
    ap.add_argument('--batch', type=int, default=64, help='streams stepped together per GPU (larger sweeps run in several batches)')
    ap.add_argument('--grid', nargs=3, type=int, default=None, metavar=('ROWS', 'COLS', 'MAX'), help='feature grid (default 4 5 5; 10 15 10 = the 1500-feature sweep)')
    ap.add_argument('--out', default='results/txts')
    ap.add_argument('--no-share-frames', action='store_true', help="round 4's per-stream staging instead of the shared frame store (A/B)")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    from . import shard
    from .config import ConfigEuRoC
    from .evaluate import ate, format_state_line, rte
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    cfg = ConfigEuRoC(grid_row=args.grid[0], grid_col=args.grid[1], grid_max_feature_num=args.grid[2]) if args.grid else ConfigEuRoC()
    root = args.root
    if args.make_synthetic:
        root = args.make_synthetic
        if rank == 0:
            from .euroc import write_euroc_layout
            from .synth import SyntheticStream
            for i, seq in enumerate(args.sequences):
                if not os.path.isdir(os.path.join(root, seq, 'mav0')):
                    write_euroc_layout(os.path.join(root, seq), SyntheticStream(cfg, seed=1000 + i, n_frames=args.frames, motion_scale=1.5, t0=1403636580.0 + 1000 * i, rest=1.0))
        if world > 1:
            dist.barrier()
    if not root:
        ap.error('--root or --make-synthetic is required')
    jobs = shard.broadcast_object([(s, o) for s in args.sequences for o in args.offsets] if rank == 0 else None)
    mine = shard.partition(len(jobs), world, rank)
    local_traj, report, stats = {}, {}, {}
    import time
    if world > 1:
        dist.barrier()
    t_all = time.perf_counter()
    for b0 in range(0, len(mine), args.batch):
        chunk = mine[b0:b0 + args.batch]
        trajs, dss = run_batched(cfg, [os.path.join(root, jobs[j][0]) for j in chunk], [jobs[j][1] for j in chunk], device=local, max_frames=args.max_frames,
                                 share_frames=not args.no_share_frames, stats=stats)
        for j, traj, ds in zip(chunk, trajs, dss):
            local_traj[j] = traj
            gt = ds.groundtruth_array()
            if len(gt) and len(traj) > 20:
                a, r = ate(traj, gt), rte(traj, gt)
                report[j] = dict(sequence=jobs[j][0], offset=jobs[j][1], frames=len(traj), ate_rmse=a['rmse'], ate_mean=a['mean'], rte_rmse=r['rmse'])
    elapsed = shard.max_over_ranks(time.perf_counter() - t_all)          # barrier before, max over ranks after: the bench contract's clock
    allt = shard.gather_trajectories(local_traj, len(jobs), world, rank)
    per_rank = shard.gather_objects({'rank': rank, 'streams': [jobs[j] for j in mine], 'report': report, 'stats': stats})
    if rank == 0:
        os.makedirs(args.out, exist_ok=True)
        for j, (seq, off) in enumerate(jobs):
            with open(os.path.join(args.out, 'output_%s_offset%d.txt' % (seq, int(off))), 'w') as f:
                for row in allt[j]:
                    f.write(format_state_line(row[0], row[1:4], row[4:8]))
        print(json.dumps(sweep_report(jobs, per_rank, elapsed, world)))
    if world > 1:
        dist.destroy_process_group()


def sweep_report(jobs, per_rank, elapsed, world):
    """ONE line for the whole sweep (BASELINE configs[3] / [4]: "aggregate frames/sec + per-seq ATE"): the stream-frames of all
    ranks over the slowest rank's wall time, and every (sequence, offset) stream's ATE / RTE in job order."""
    rep = {}
    for r in per_rank:
        rep.update(r['report'])
    tot = {k: sum(r['stats'].get(k, 0) for r in per_rank) for k in ('stream_frames', 'steps', 'frames_decoded')}
    return {'metric': 'stream-frames/s of the sequence x offset sweep (decode + front-end + MSCKF, end to end)',
            'value': tot['stream_frames'] / elapsed if elapsed > 0 else None, 'unit': 'stereo frames/s', 'n_gpus': world,
            'streams': [list(j) for j in jobs], 'stream_frames': tot['stream_frames'], 'frames_decoded': tot['frames_decoded'], 'seconds': elapsed,
            'streams_per_rank': [len(r['streams']) for r in per_rank],
            'stepping_seconds_per_rank': [r['stats'].get('seconds', 0.0) for r in per_rank],
            'report': [dict(rep[j]) if j in rep else dict(sequence=jobs[j][0], offset=jobs[j][1], frames=0) for j in range(len(jobs))]}


if __name__ == '__main__':
    main()
