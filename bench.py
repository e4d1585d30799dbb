#!/usr/bin/env python3
"""bench.py -- stereo frames/s of the MI355X image front-end on synthetic 752x480 streams.

Workload = BASELINE.json configs[1]: "Synthetic 752x480 stereo stream, 300 tracked features,
LK+stereo kernels on 1 MI355X" -- the reference front-end (ImageProcessingPipeline.stereo_callback:
temporal LK, stereo LK fwd/bwd + gates, FAST + grid add/prune, publish) with grid 4x5 x 15 = 300
features per frame, run for S independent streams per GPU (frames of one stream are sequential,
streams are the parallel axis).  One "step" = one stereo frame of every stream.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Without a torchrun environment `--gpus N` (N > 1) starts the N ranks itself as fresh child processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) BEFORE anything touches the GPU, relays rank 0's line and exits non-zero
if any rank failed.  Prints ONE JSON line on rank 0 (contract in the task statement).  Inputs are resident in HBM
before the timed region; IMU samples are pushed from the host inside it (they are part of the path).

The timed region is always the steady state: before the `--warmup` steps the bench pre-rolls, un-timed, until every
filter holds its full camera-state window and has run the camera pruning (PREROLL frames), so the result does not depend
on how small `--warmup` is.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_IMG, H_IMG = 752, 480
LK_BYTES_PER_POINT_PASS = 4 * 2 * 289 + 25          # L=4 levels x (I + J window of 17x17) + point I/O
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: 8 TB/s spec
# rocprofv3 PMC passes of `bench.py --frontend-only --streams 64|512` (three separate passes: SQ block, FETCH_SIZE, WRITE_SIZE; collected
# by profiles/r03/collect_pmc.sh, per-kernel means in the committed summary read below).  Per lk_track_g16_kernel launch, mean over
# the launch mix of a step (temporal, stereo forward / backward of the tracked points: 300 point passes per stream each; candidates
# round 1 forward / backward: 100 each; round 2: ~10 each), 64 streams per launch; bench.py's avg_launch_ms is the mean over the same
# mix, so per-launch means scale by streams / 64.
# FETCH_SIZE correction: MI355X_MICROARCH.md prescribes x2 on gfx950 for 16-byte-per-lane streams and asks for a calibration of
# other access patterns.  profiles/r03/fetch_calib.{hip,json} (run on the GPU box under the same --pmc pass) reads KNOWN byte counts
# with 16-, 4- and 1-byte lane loads and with one dword per 128-byte / per 64-byte line: FETCH_SIZE reports 0.50000 of the bytes in
# every case, and a sparse read costs the time of the whole line (a request is 128 B, tallied as 64) -- so the x2 applies to this
# kernel's 4-byte staging loads as well (round 2 argued it might not; it does).  WRITE_SIZE is exact.
def _load_pmc():
    # round 5: all three counter families measured at 512 streams per launch with the round's kernel (profiles/r05/final_pmc.sh: the
    # largest batch rocprofv3 --pmc collects on this image), the 64-stream pass beside it for the per-stream scaling check
    p512 = os.path.join(ROOT, 'profiles', 'r05', 'pmc_frontend_s512_summary.json')
    if os.path.exists(p512):
        e = json.load(open(p512))['lk_track_g16_kernel<15>']
        return dict(path=os.path.relpath(p512, ROOT), streams=512, fetch_kb=float(e['FETCH_SIZE']), write_kb=float(e['WRITE_SIZE']),
                    valu=float(e['SQ_INSTS_VALU']), waves=float(e['SQ_WAVES']), fetch_streams=512, fetch_path=os.path.relpath(p512, ROOT),
                    wait_inst_share=float(e['wait_inst_share']), wait_any_share=float(e['wait_any_share']))
    for d in ('r03', 'r02'):
        path = os.path.join(ROOT, 'profiles', d, 'pmc_frontend_s64_summary.json')
        if os.path.exists(path):
            e = json.load(open(path))['lk_track_g16_kernel<15>']
            out = dict(path=os.path.relpath(path, ROOT), streams=64, fetch_kb=float(e['FETCH_SIZE']), write_kb=float(e['WRITE_SIZE']),
                       valu=float(e['SQ_INSTS_VALU']), waves=float(e['SQ_WAVES']), fetch_streams=64)
            # round 4: FETCH_SIZE measured at the largest batch rocprofv3 --pmc collects on this image (512 streams; it segfaults at
            # 1,024+), where the pyramids no longer fit the Infinity Cache: per-stream traffic +2.8 % over the 64-stream figure
            sc = os.path.join(ROOT, 'profiles', 'r04', 'pmc_fetch_scaling.json')
            if os.path.exists(sc):
                f = json.load(open(sc))['streams']
                big = max(f, key=int)
                out.update(fetch_kb=float(f[big]['lk_track_g16_kernel<15>']['FETCH_SIZE_KB_mean']), fetch_streams=int(big),
                           fetch_path=os.path.relpath(sc, ROOT))
            return out
    raise RuntimeError('bench.py: no committed PMC summary under profiles/')


FETCH_SIZE_FACTOR = 2.0                             # measured: profiles/r03/fetch_calib.json (raw_over_known = 0.5 for every lane width)
LK_PMC = _load_pmc()
LK_PMC_STREAMS = LK_PMC['streams']
LK_TRAFFIC_BYTES_PER_LAUNCH_S64 = (FETCH_SIZE_FACTOR * LK_PMC['fetch_kb'] * LK_PMC_STREAMS / LK_PMC['fetch_streams'] + LK_PMC['write_kb']) * 1024      # per 64 streams
LK_VALU_INSTS_PER_LAUNCH_S64 = LK_PMC['valu']
LK_PMC_POINT_PASSES_PER_STREAM_LAUNCH = 1113.915 / 7.0      # lk_point_passes_per_frame / LK launches per step of the counter runs' workload (grid 4x5x15)
FP64_PEAK_TFLOPS = 78.6                             # MI355X_MICROARCH.md: fp64 vector = fp64 matrix peak
# Issue cost of one VALU wave-instruction of lk_track_g16_kernel's mix (round 5).  Rounds 2-4 priced every instruction at the guide's
# 2 cycles (a v_fma_f32).  profiles/r05/valu_issue_microbench.json times ~65 instructions apart (wall clock of whole launches, one
# workgroup per CU, 1-4 waves per SIMD; cross-check: v_fma_f64 comes out at 61 TFLOP/s chip-wide, round 2 measured 60): there are two
# classes -- FULL rate (v_add_u32, shifts, v_and, v_fma_f32, v_mul_f32, v_mov: 1.24 ns per wave-instruction and SIMD) and HALF rate
# (v_dot2_i32_i16 and its DPP form, v_mad_i32_i16 / _i24, v_perm, v_alignbyte, packed 16-bit, DPP, SDWA, min / max, bfe, every
# three-operand integer op, conversions, compares, all of fp64: 2.07 ns = 1.66x) -- and a census of the kernel's hot path finds 70.3 %
# of its instructions in the half-rate class (lk_dma_experiment.md).  The guide's 2 cycles x (0.703 x 1.66 + 0.297) = 2.93.
# SQ_INSTS_VALU itself is exact (valu_counter_calib.txt: 64,022 per wave for every class against 64,000 + prologue executed).
VALU_CYCLES_PER_WAVE_INST = 2.93
N_SIMD, CLOCK_HZ = 1024, 2.4e9


PREROLL_FULL = 24          # frames until every filter has 20 camera states and has pruned at least once (19 + margin)
PREROLL_FE = 8             # front-end only: the feature grid is full after a few frames
REFERENCE_FILTER_FPS_300 = 10.0   # the genuine reference MSCKF.feature_callback at 300 features/frame,
                                  # imported unmodified, 1 thread of an 8-vCPU Xeon 2.1 GHz (filter half only; no cv2 there)


def frame_bytes(n_t, n_trk, n_cand):
    """Algorithmic bytes of one stereo frame: 2 images + 2x3 pyramid levels + LK point passes."""
    p = n_t + 2 * n_trk + 2 * n_cand
    return 2 * W_IMG * H_IMG + 2 * 118440 + p * LK_BYTES_PER_POINT_PASS, p


def make_config(grid=(4, 5, 15)):
    from uav_airvision_amd.config import ConfigEuRoC
    return ConfigEuRoC(grid_row=grid[0], grid_col=grid[1], grid_min_feature_num=3, grid_max_feature_num=grid[2])


def cpu_baseline(cfg, with_msckf, budget_s=10.0, max_frames=400, seed=0, traj_frames=0):
    """The CPU oracle (oracle/: scalar C image ops + Python glue + numpy MSCKF = a port of the reference's
    CPU path) on ONE stream of the same workload, single thread, bounded sample.  With traj_frames > 0 the first
    traj_frames frames are always processed (whatever the budget) and their filter poses returned, so that the GPU
    trajectory of the same stream can be compared with the CPU path's (ATE vs CPU ref)."""
    from oracle.frontend import OracleFrontend
    from oracle.msckf_np import OracleMSCKF
    from uav_airvision_amd.synth import SyntheticStream
    try:                                       # "cores": 1 must be true: pin numpy's BLAS to one thread
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    st = SyntheticStream(cfg, seed=seed, n_frames=max_frames)
    frames = []
    fe = OracleFrontend(cfg, cache_pyramids=True)
    flt = OracleMSCKF(cfg) if with_msckf else None
    it = iter(st.imu)
    pend = next(it, None)
    n = 0
    spent = 0.0
    warm = PREROLL_FULL if with_msckf else PREROLL_FE      # grid full (~300 features) and 20 camera states before timing
    traj = []
    for k in range(max_frames):
        m = st.frame(k)
        while pend is not None and pend.timestamp <= m.timestamp:
            fe.imu_callback(pend)
            if flt is not None:
                flt.imu_callback(pend)
            pend = next(it, None)
        t0 = time.perf_counter()
        msg = fe.stereo_callback(m)
        res = flt.feature_callback(msg) if flt is not None else None
        dt = time.perf_counter() - t0
        if res is not None and k < traj_frames:
            s_ = flt.imu_state
            traj.append([m.timestamp] + [float(v) for v in s_.position] + [float(v) for v in s_.orientation])
        if k >= warm:
            spent += dt
            n += 1
            frames.append(len(msg.features))
            if spent >= budget_s and k + 1 >= traj_frames:
                break
    out = dict(value=n / spent, unit='stereo frames/s', cores=1, kind='port',
               sample='%d frames of 1 synthetic stream after %d warm-up frames, %.1f s of CPU, mean %d features/frame, %s; for context, the '
                      'genuine reference filter half alone (MSCKF.feature_callback imported unmodified, SURVEY.md section 6) runs %.1f frames/s '
                      'at 300 features/frame on one 2.1 GHz Xeon thread -- its image half needs cv2, which is not installed'
                      % (n, warm, spent, int(np.mean(frames)), 'front-end + MSCKF' if with_msckf else 'front-end only', REFERENCE_FILTER_FPS_300))
    if traj_frames:
        out['_traj'] = traj
        out['_truth'] = [[st.frame_time(k)] + [float(v) for v in st.position(st.frame_time(k))] for k in range(min(traj_frames, max_frames))]
    return out


def cpu_baseline_all_cores(with_msckf, budget_s=6.0, grid=(4, 5, 15)):
    """The same single-thread port, one independent stream per host core in parallel child processes (they never touch the
    GPU): the CPU box's aggregate rate on this workload, for context next to the one-core figure."""
    import subprocess
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    n = max(1, min(16, n))                     # a GPU box gives one GPU's job a share of 16 cores
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--cpu-budget', str(budget_s), '--grid'] + [str(v) for v in grid]
    if not with_msckf:
        cmd.append('--frontend-only')
    procs = [subprocess.Popen(cmd + ['--cpu-seed', str(100 + i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(n)]
    vals = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        if p.returncode == 0 and out.strip():
            vals.append(json.loads(out.strip().splitlines()[-1])['value'])
    if not vals:
        return None
    return dict(value=float(sum(vals)), unit='stereo frames/s', cores=len(vals), kind='port',
                sample='%d concurrent single-thread ports, one synthetic stream each, %.0f s of CPU per process' % (len(vals), budget_s))


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a torchrun environment: start one fresh child per rank (this process has not
    imported torch and never touches the GPU), relay rank 0's stdout, fail if any rank fails."""
    import subprocess
    env0 = dict(os.environ)
    env0.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env0['MASTER_ADDR'] = '127.0.0.1'
    env0['MASTER_PORT'] = str(_free_port())
    env0['WORLD_SIZE'] = str(n_ranks)
    env0['LOCAL_WORLD_SIZE'] = str(n_ranks)
    procs = []
    for r in range(n_ranks):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for line in (out0 or '').splitlines():          # the contract is ONE JSON line on stdout: library chatter goes to stderr
        (sys.stdout if line.lstrip().startswith('{') else sys.stderr).write(line + '\n')
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        sys.stderr.write('bench.py: ranks failed (rank, exit code): %s\n' % bad)
        return 1
    return 0


def dry_run(args):
    """The multi-rank plumbing without a GPU (gloo): config broadcast from rank 0 -> stream partition -> barrier-bracketed
    timed loop of no-op steps -> max-over-ranks time and sum-over-ranks counters -> ONE JSON line on rank 0.  Used by the
    CPU test of the launcher; never a measurement (`dry_run: true`, value counts no-op steps)."""
    import torch.distributed as dist
    from uav_airvision_amd import shard
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    run_cfg = shard.broadcast_object({'seed': 1234, 'streams': args.streams, 'steps': args.steps, 'warmup': args.warmup} if rank == 0 else None)
    S, K = int(run_cfg['streams']), int(run_cfg['steps'])
    mine = shard.partition(S * world, world, rank)            # weak scaling: S streams per rank
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    done = 0
    for _ in range(K):
        time.sleep(0.001 * (1 + rank))                         # ranks differ: the max over ranks must win
        done += len(mine)
    if world > 1:
        dist.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0)
    total = float(shard.sum_over_ranks([done])[0])
    if rank == 0:
        print(json.dumps({'metric': 'stereo frames/sec (LK+stereo+MSCKF-update) at 752x480', 'value': total / elapsed, 'unit': 'stereo frames/s',
                          'n_gpus': world, 'steps': K, 'warmup': int(run_cfg['warmup']), 'ms_per_step': 1e3 * elapsed / K, 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': 'none', 'data': 'none', 'dry_run': True,
                          'config': {'workload': 'dry run of the launcher and the rank plumbing (no GPU work)', 'streams_per_gpu': S,
                                     'streams_total': int(total / K), 'seed': int(run_cfg['seed'])}}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def measure_regime(torch, dev, local_rank, name, grid, S_r, img0, img1, imu_steps, frame_ts, n_pre, n_steps, host=False, what=''):
    """The complete path (front-end + batched filter, pipelined as in the headline) at a small batch: the first S_r streams of the
    resident synthetic frames, own engine pair, n_pre un-timed frames into the steady state, then n_steps timed ones.
    These are the batch sizes the BASELINE EuRoC configs have per GPU (configs[2]: 1 stream; configs[4]: 8 streams at 1,500
    features; run.bat:4-12: 63 streams); the headline batch (thousands of replicas) is the throughput regime, these are the
    latency-bound ones.  host=True: frames handed over as pageable host arrays (H2D inside the timed region)."""
    from uav_airvision_amd.frontend import FrontendEngine
    from uav_airvision_amd.msckf_ops import BatchedMSCKF
    cfg_r = make_config(grid)
    eng = FrontendEngine(cfg_r, n_streams=S_r, device=local_rank, inputs_persist=not host)
    flt = BatchedMSCKF(cfg_r, S_r, device=local_rank, max_features=eng.max_features)
    try:
        imu_r = []
        for k in range(n_pre + n_steps):
            i, t, gy, ac = imu_steps[k]
            m = i < S_r
            imu_r.append((i[m], t[m], gy[m], ac[m]))
        ts_r = [frame_ts[k][:S_r] for k in range(n_pre + n_steps)]
        if host:
            h0 = img0[:n_pre + n_steps, :S_r].cpu().numpy()
            h1 = img1[:n_pre + n_steps, :S_r].cpu().numpy()
            frames = lambda k: (h0[k], h1[k])
        else:
            frames = lambda k: (img0[k][:S_r], img1[k][:S_r])
        path = CompletePath(torch, dev, eng, flt, torch.cuda.Stream(device=dev), frames, host, imu_r, imu_r, ts_r)
        path.run_pipelined(0, n_pre)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        path.run_pipelined(n_pre, n_pre + n_steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        feats = eng.read_features()
        c = flt.counters()
        return dict(name=name, what=what, streams=S_r, grid='%dx%dx%d' % tuple(grid), features_per_frame=float(np.mean([len(f[0]) for f in feats])),
                    steps=n_steps, prerolled_frames=n_pre, ms_per_step=1e3 * dt / n_steps, stream_frames_per_s=S_r * n_steps / dt,
                    frames_per_s_per_stream=n_steps / dt, cam_states=[c['min_cam_states'], c['max_cam_states']],
                    inputs='host numpy arrays, H2D inside the timed region (PCIe-inclusive)' if host else 'resident in HBM')
    finally:
        eng.close(); flt.close()


class CompletePath(object):
    """The complete path for one engine pair, organised like the reference's VIO (vio.py:24-76: image thread -> feature
    queue -> filter thread): the calling thread drives the front-end and hands every frame's feature message to a filter
    thread through a bounded queue, so the front-end of later frames overlaps the filter step of earlier ones.
    frames(k) -> (cam0, cam1) of step k for all streams (cuda tensors, or host arrays with host=True); imu_fe / imu_flt: per
    step (stream index, t, gyro, acc) arrays for the two consumers; frame_ts[k][s]."""

    def __init__(self, torch, dev, eng, flt, filt_stream, frames, host, imu_fe, imu_flt, frame_ts):
        self.torch, self.dev, self.eng, self.flt, self.filt_stream = torch, dev, eng, flt, filt_stream
        self.frames, self.host, self.imu_fe, self.imu_flt, self.frame_ts = frames, host, imu_fe, imu_flt, frame_ts
        self.msckf_s, self.push_s, self.poses = [0.0], [0.0], []      # poses: (frame index, out[S,12]) of every filter step
        self.step_times = None
        self.n_frames = None                                           # set by the caller that knows how many frames `frames` holds

    def run_fe(self, k):
        i, t, gy, ac = self.imu_fe[k]
        self.eng.push_imu_batch(i, t, gy)
        a, b = self.frames(k)
        if self.host:
            self.eng.step_host(a, b, self.frame_ts[k])
        else:
            self.eng.step(a, b, self.frame_ts[k])

    def run_filter(self, k, ids_h, uv_h, n_h, queued=False):
        t1 = time.perf_counter()
        i, t, gy, ac = self.imu_flt[k]
        self.flt.push_imu(i, t, gy, ac)
        self.push_s[0] += time.perf_counter() - t1
        with self.torch.cuda.stream(self.filt_stream):         # the filter's kernels overlap the next frame's front-end kernels
            if queued:
                self.poses.append((k, self.flt.submit(ids_h, uv_h, n_h, self.frame_ts[k])))   # the stream groups run behind their own queues ...
                self.flt.wait(1)                               # ... at most one frame ahead of the slowest group
            else:
                self.poses.append((k, self.flt.step(ids_h, uv_h, n_h, self.frame_ts[k])))
        self.msckf_s[0] += time.perf_counter() - t1

    def run_pipelined(self, k_begin, k_end):
        """Frames [k_begin, k_end).  The region ends when the last filter step has returned."""
        if self.flt.device_resident():
            return self.run_device_handover(k_begin, k_end)
        import queue, threading
        q = queue.Queue(maxsize=2)
        err = []
        eng, torch, dev = self.eng, self.torch, self.dev

        def filter_loop():
            torch.cuda.set_device(dev)
            while True:
                item = q.get()
                if item is None:
                    return
                try:
                    if not err:
                        self.run_filter(*item, queued=True)
                except Exception as e:          # surfaced by the main thread after the join
                    err.append(e)

        th = threading.Thread(target=filter_loop, name='msckf')
        th.start()
        try:
            # frame k+1 is enqueued before frame k's features are consumed: the read-back of k (pinned, double buffered)
            # sits between the two steps on the stream, so the GPU never waits for this thread
            self.run_fe(k_begin)
            eng.read_features_begin(k_begin & 1)
            for k in range(k_begin, k_end):
                if k + 1 < k_end:
                    self.run_fe(k + 1)
                    eng.read_features_begin((k + 1) & 1)
                ids_h, uv_h, n_h = eng.read_features_end(k & 1)      # waits for frame k's copy only; fresh host arrays
                q.put((k, ids_h, uv_h, n_h))
                if self.step_times is not None:
                    self.step_times.append((k, time.perf_counter()))
        finally:
            q.put(None)
            th.join()
        if err:
            raise err[0]
        self.flt.wait(0)                                             # the last queued step retires inside the region

    def run_device_handover(self, k_begin, k_end):
        """The same path with the filter state resident on the device: the filter reads every frame's feature message where
        the front-end left it (av_msckf_batch_submit_dev), so this thread only enqueues -- front-end step k, IMU samples, filter
        step k -- and never waits for a result inside the loop; the library's stream groups work through their queues at most
        two steps behind (vio.py:46-51 with the feature queue on the device)."""
        from uav_airvision_amd import _native as N
        torch = self.torch
        # serial: front-end and filter strictly alternate (exclusive GPU times add up) -- the warm-up steps of the default run (the filter's
        # chain timed with the GPU to itself: roofline_msckf.exclusive), or the whole run as a diagnostic (AV_BENCH_SERIAL=1)
        serial = os.environ.get('AV_BENCH_SERIAL') or getattr(self, 'force_serial', False)
        # The next frame's pyramids are enqueued BEFORE this frame's feature message is handed over (av_frontend_prestage): the filter's
        # chain then starts behind the pyramid kernels instead of beside them (its first kernel and the pyramid kernel halve each other:
        # both want most of a CU's LDS).  Same kernels per step, same results; AV_BENCH_PRESTAGE=0 is the A/B.
        prestage = (not serial) and (not self.host) and os.environ.get('AV_BENCH_PRESTAGE', '1') != '0' and self.n_frames is not None
        for k in range(k_begin, k_end):
            self.run_fe(k)
            if prestage and k + 1 < self.n_frames:
                a, b = self.frames(k + 1)
                self.eng.prestage(a, b)
            if serial:
                if hasattr(self.eng, 'drain'): self.eng.drain()
                torch.cuda.synchronize()
                self.fe_excl_s = getattr(self, 'fe_excl_s', 0.0)
            t1 = time.perf_counter()
            i, t, gy, ac = self.imu_flt[k]
            self.flt.push_imu(i, t, gy, ac)
            self.push_s[0] += time.perf_counter() - t1
            ms = N.current_stream()
            with torch.cuda.stream(self.filt_stream):
                self.poses.append((k, self.flt.submit_dev(self.eng, np.asarray(self.frame_ts[k], dtype=np.float64), msg_stream=ms)))
            if serial:
                self.flt.wait(0)
            self.msckf_s[0] += time.perf_counter() - t1
            if self.step_times is not None:
                self.step_times.append((k, time.perf_counter()))
        self.flt.wait(0)

    def run(self, k, filt=True):
        self.run_fe(k)
        if self.flt is not None and filt:
            ids_h, uv_h, n_h = self.eng.read_features_raw()
            self.run_filter(k, ids_h, uv_h, n_h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--streams', type=int, default=2048, help='independent stereo streams per GPU')
    ap.add_argument('--unique', type=int, default=64, help='distinct rendered streams (own scene, own speed and phase of the trajectory); replicated with per-stream noise')
    ap.add_argument('--grid', nargs=3, type=int, default=[4, 5, 15], metavar=('ROWS', 'COLS', 'MAX'),
                    help='feature grid: 4 5 15 = 300 features/frame (BASELINE configs[1], the default); 10 15 10 = 1500 (configs[4])')
    ap.add_argument('--no-stagger', action='store_true', help='start every replica at frame 0 (all filters then prune on the same frames)')
    ap.add_argument('--host-images', action='store_true', help='front-end only, images handed over as host numpy arrays every step '
                    '(av_frontend_step_host: the PCIe-inclusive rate quoted in DESIGN.md; never the contract value)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-regimes', action='store_true', help='skip the small-batch regime lines (S = 1, 8 x 1500 features, 64, host-fed)')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-budget', type=float, default=6.0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-seed', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--dry-run', action='store_true', help='exercise launcher + rank plumbing on CPU (gloo), no GPU work, not a measurement')
    ap.add_argument('--frontend-only', action='store_true', help='time only the image front-end (BASELINE configs[1] literally: MSCKF not on the GPU)')
    ap.add_argument('--pipelines', type=int, default=int(os.environ.get('AV_BENCH_PIPELINES', '1')),
                    help='independent pipelines per GPU: the streams are split into this many parts, each with its own front-end engine, batched filter, host thread '
                         'and HIP streams (uav_airvision_amd/pipelines.py); 1 = one engine and one filter for all streams')
    args = ap.parse_args()
    bench_rc = 0
    if args.cpu_baseline_worker:               # child of cpu_baseline_all_cores: CPU only, exits before torch is imported
        print(json.dumps(cpu_baseline(make_config(args.grid), not args.frontend_only, budget_s=args.cpu_budget, seed=args.cpu_seed)))
        return 0
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)           # before torch / HIP are touched in this process
    if args.dry_run:
        return dry_run(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: WORLD_SIZE=%d but --gpus %d\n' % (world, args.gpus))
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)

    from uav_airvision_amd.frontend import FrontendEngine
    from uav_airvision_amd.synth import SyntheticStream

    if world > 1 and 'AV_HOST_THREADS' not in os.environ:
        # the filter's bookkeeping threads of all ranks share the node's cores: keep every rank inside its share
        os.environ['AV_HOST_THREADS'] = str(max(4, min(16, (os.cpu_count() or 16) // world)))
    cfg = make_config(args.grid)
    n_grid_feat = args.grid[0] * args.grid[1] * args.grid[2]
    S, K, Wm = args.streams, args.steps, args.warmup
    with_msckf = not (args.frontend_only or args.host_images)
    PRE = PREROLL_FULL if with_msckf else PREROLL_FE
    stagger = 0 if args.no_stagger else 1
    PRE += stagger                              # the replicas that start one frame later reach the steady state one step later
    F = PRE + Wm + K + (K if with_msckf else 0)  # a second timed loop measures the front-end alone
    # config/seed broadcast from rank 0 over RCCL (config broadcast, no data-path collective)
    from uav_airvision_amd import shard
    run_cfg = shard.broadcast_object({'seed': 1234, 'streams': S, 'steps': K, 'warmup': Wm} if rank == 0 else None)
    base_seed = int(run_cfg['seed'])

    # ---- synthetic data: U rendered streams, replicated to S streams with per-stream pixel noise ----
    # Replica 0 of every rendered stream carries no extra noise: GPU stream 0 is exactly SyntheticStream(seed) -- the one the
    # CPU path replays for `ate_vs_cpu_ref`.
    U = max(1, min(args.unique, S))
    t_gen = time.time()
    stream_seed0 = base_seed + 97 * rank
    # U distinct streams.  Stream 0 is exactly SyntheticStream(seed) rendered by numpy (the CPU path replays it for `ate_vs_cpu_ref`);
    # the others are rendered on the GPU by the same renderer restated in torch, each with its own scene (its own texture up to 8,
    # then a random window of one of the 8), its own speed (motion_scale 0.7-1.6) and its own phase (0-2 s of standstill before the
    # trajectory starts), so that LK iteration counts, track loss and the filter's work per frame differ between streams instead of
    # marching in lock-step.
    from uav_airvision_amd.synth import make_texture
    prng = np.random.default_rng(base_seed + 7919 * rank)
    tex_pool = [make_texture(0xA1B0 + stream_seed0 + j) for j in range(min(U, 8))]
    streams = []
    for u in range(U):
        if u == 0:
            st = SyntheticStream(cfg, seed=stream_seed0, n_frames=F, texture=tex_pool[0])
        else:
            st = SyntheticStream(cfg, seed=stream_seed0 + u, n_frames=F, texture=tex_pool[u % len(tex_pool)], render=False,
                                 motion_scale=float(prng.uniform(0.7, 1.6)), rest=float(prng.uniform(0.0, 2.0)),
                                 tex_offset=(0.0, 0.0) if u < len(tex_pool) else (float(prng.uniform(0, 2048)), float(prng.uniform(0, 1536))))
            st.tex, st.rays0, st.rays1 = tex_pool[u % len(tex_pool)], streams[0].rays0, streams[0].rays1
        streams.append(st)
    b0 = torch.empty((U, F, H_IMG, W_IMG), dtype=torch.uint8, device=dev)
    b1 = torch.empty((U, F, H_IMG, W_IMG), dtype=torch.uint8, device=dev)
    for k in range(F):
        m = streams[0].frame(k)
        b0[0, k] = torch.from_numpy(m.cam0_image).to(dev)
        b1[0, k] = torch.from_numpy(m.cam1_image).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(base_seed + rank)
    rays_dev = (torch.from_numpy(streams[0].rays0).to(dev), torch.from_numpy(streams[0].rays1).to(dev))
    tex_dev = [torch.from_numpy(t).to(dev) for t in tex_pool]
    for u in range(1, U):
        state = dict(tex=tex_dev[u % len(tex_pool)], rays0=rays_dev[0], rays1=rays_dev[1])
        for k in range(F):
            b0[u, k], b1[u, k] = streams[u].frame_torch(k, state, g)
    img0 = torch.empty((F, S, H_IMG, W_IMG), dtype=torch.uint8, device=dev)
    img1 = torch.empty((F, S, H_IMG, W_IMG), dtype=torch.uint8, device=dev)
    for s in range(S):
        u = s % U
        for dst, src in ((img0, b0), (img1, b1)):
            if s < U:
                dst[:, s] = src[u]
            else:
                noise = torch.randint(-2, 3, (F, H_IMG, W_IMG), generator=g, device=dev, dtype=torch.int16)
                dst[:, s] = (src[u].to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    del b0, b1, tex_dev, rays_dev
    # IMU samples per step, all streams, as flat arrays for the batched push
    imu_steps = []
    its = [iter(st.imu) for st in streams]
    pend = [next(it, None) for it in its]
    for k in range(F):
        idx, ts, gy, ac = [], [], [], []
        for u, st in enumerate(streams):
            tf = st.frame_time(k)
            batch = []
            while pend[u] is not None and pend[u].timestamp <= tf:
                batch.append(pend[u])
                pend[u] = next(its[u], None)
            for s in range(u, S, U):
                for m in batch:
                    idx.append(s); ts.append(m.timestamp); gy.append(m.angular_velocity); ac.append(m.linear_acceleration)
        imu_steps.append((np.array(idx, np.int32), np.array(ts, np.float64), np.array(gy, np.float64).reshape(-1, 3),
                          np.array(ac, np.float64).reshape(-1, 3)))
    # The filter's copy of the IMU feed.  Staggered start: the filters of the odd replicas never see their first 10 IMU samples
    # (one frame period), so the sample that completes their gravity initialisation (the 200th, msckf.py:172-175) is newer than
    # frame 0: by the data-defined rule of the library they are not live at step 0 (that frame's features are dropped,
    # msckf.py:182-183) and start at step 1, whatever the thread timing.  From then on half of the filters run the two-camera
    # prune on even steps and half on odd ones -- the steady state of unsynchronised streams -- instead of all 1024 in
    # lock-step.  The front-end sees every sample.
    imu_steps_f = list(imu_steps)
    if stagger:
        i0, t0_, g0, a0 = imu_steps[0]
        drop = (((i0 // U) % 2) == 1) & (t0_ < t0_.min() + 10.0 / 200.0 - 1e-6)
        imu_steps_f[0] = (i0[~drop], t0_[~drop], g0[~drop], a0[~drop])
    frame_ts = [[streams[s % U].frame_time(k) for s in range(S)] for k in range(F)]
    gen_s = time.time() - t_gen

    # Optional CU partition (AV_FE_CUS="first:count" for the front-end's stream, AV_MSCKF_CUS for the filter's inside the
    # library): the front-end's kernels are then confined to their compute units and the filter's latency-bound kernels
    # never queue behind resident front-end workgroups.
    fe_stream = None
    if os.environ.get('AV_FE_CUS'):
        import ctypes
        first, count = [int(v) for v in os.environ['AV_FE_CUS'].split(':')]
        mask = (ctypes.c_uint32 * 8)()
        for c in range(first, min(256, first + count)):
            mask[c >> 5] |= 1 << (c & 31)
        hip = ctypes.CDLL('libamdhip64.so')
        hs = ctypes.c_void_p()
        rc_ = hip.hipExtStreamCreateWithCUMask(ctypes.byref(hs), 8, mask)
        if rc_ != 0:
            sys.stderr.write('bench.py: hipExtStreamCreateWithCUMask failed (%d)\n' % rc_)
            return 4
        fe_stream = torch.cuda.ExternalStream(hs.value, device=dev)
        torch.cuda.set_stream(fe_stream)
    elif os.environ.get('AV_FE_PRIO'):           # A/B: the front-end's launches on a stream of the highest (2) / lowest (0) HIP priority
        lo_hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
        fe_stream = torch.cuda.Stream(device=dev, priority=(min(lo_hi) if os.environ['AV_FE_PRIO'] == '2' else max(lo_hi)))
        torch.cuda.set_stream(fe_stream)
    # Several independent pipelines on the GPU (--pipelines P): P engines + P filters of S / P streams, each on its own host thread and
    # HIP streams; same streams, same kernels, same per-stream results.  Only for the device-resident hand-over and the front-end-only loop.
    n_pipes = max(1, min(args.pipelines, S))
    if args.host_images or os.environ.get('AV_MSCKF_STORE') == 'host' or fe_stream is not None:
        n_pipes = 1
    flt = None
    if n_pipes > 1:
        from uav_airvision_amd.pipelines import EngineSet, FilterSet
        eng = EngineSet(cfg, S, n_pipes, device=local_rank, inputs_persist=True)
        if with_msckf:
            flt = FilterSet(cfg, eng, device=local_rank, rows_cap=4096 if 5 * eng.max_features + 64 <= 4096 else None, max_features=eng.max_features)
            if not flt.device_resident():
                sys.stderr.write('bench.py: --pipelines needs the device-resident filter\n')
                return 4
    else:
        eng = FrontendEngine(cfg, n_streams=S, device=local_rank, inputs_persist=True)      # every frame of the run is resident in HBM
        if with_msckf:
            from uav_airvision_amd.msckf_ops import BatchedMSCKF
            flt = BatchedMSCKF(cfg, S, device=local_rank, rows_cap=4096 if 5 * eng.max_features + 64 <= 4096 else None, max_features=eng.max_features)
    filt_stream = torch.cuda.Stream(device=dev) if flt is not None else None

    host0 = host1 = None
    if args.host_images:                         # the same frames as pageable host arrays; the device copies are dropped
        host0, host1 = img0.cpu().numpy(), img1.cpu().numpy()
        del img0, img1
        torch.cuda.empty_cache()

    path = CompletePath(torch, dev, eng, flt, filt_stream,
                        (lambda k: (host0[k], host1[k])) if host0 is not None else (lambda k: (img0[k], img1[k])),
                        host0 is not None, imu_steps, imu_steps_f, frame_ts)
    path.step_times = [] if os.environ.get('AV_BENCH_STEP_TIMES') else None     # diagnostic: wall time at which every frame's features were handed to the filter
    path.n_frames = len(frame_ts)
    msckf_s, push_s, poses0, step_times = path.msckf_s, path.push_s, path.poses, path.step_times
    run, run_pipelined = path.run, path.run_pipelined

    def barrier():
        if hasattr(eng, 'drain'):
            eng.drain()                               # the pipelines' workers have enqueued everything they were handed
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- pre-roll (un-timed, whatever --warmup says): into the steady state, with the same pipelined organisation ----
    if flt is not None:
        run_pipelined(0, PRE)
    else:
        for k in range(PRE):
            run(k)
    eng.read_features()                              # sync + overflow check of the pre-roll
    c_pre = flt.counters() if flt is not None else None
    if c_pre is not None and (c_pre['min_cam_states'] < cfg.max_cam_state_size - 2 or c_pre['prune_stream_steps'] < S):
        sys.stderr.write('bench.py: pre-roll did not reach the steady state: %s\n' % c_pre)
        return 3
    # ---- the contract's W untimed warm-up steps ----
    # (device-resident filter: stepped with front-end and filter strictly alternating, the filter's phase chains under HIP events --
    #  what the chain takes with the GPU to itself, reported beside the in-path figure as roofline_msckf.exclusive)
    w_e0 = None
    if flt is not None and Wm > 0:
        if flt.device_resident():
            w_e0 = flt.work(enable=1); w_e0.update(flt.work_executed())
            path.force_serial = True
        run_pipelined(PRE, PRE + Wm)
        path.force_serial = False
    else:
        for k in range(PRE, PRE + Wm):
            run(k)
    eng.read_features()
    c_t0 = flt.counters() if flt is not None else None
    w_t0 = flt.work(enable=1) if flt is not None else None      # HIP events around the filter's phase chains from here on
    if w_t0 is not None: w_t0.update(flt.work_executed())
    eng.enable_timing(16 * (F - PRE - Wm) + 16)

    barrier()
    msckf_s[0] = 0.0
    push_s[0] = 0.0
    T0 = PRE + Wm
    t0 = time.perf_counter()
    if flt is not None:
        run_pipelined(T0, T0 + K)
    else:
        for k in range(T0, T0 + K):
            run(k)
            if step_times is not None:
                torch.cuda.synchronize()
                step_times.append((k, time.perf_counter()))
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed)
    timing = eng.read_timing()
    if step_times:
        tt = [t for k, t in step_times if k >= T0]
        sys.stderr.write('[step intervals ms] ' + ' '.join('%.1f' % ((b_ - a_) * 1e3) for a_, b_ in zip(tt[:-1], tt[1:])) + '\n')
    c_t1 = flt.counters() if flt is not None else None
    w_t1 = flt.work(enable=0) if flt is not None else None
    if w_t1 is not None: w_t1.update(flt.work_executed())
    fe_elapsed = None
    timing_fe = None
    if with_msckf:                                   # same engine state, next K frames, front-end only
        barrier()
        t1 = time.perf_counter()
        for k in range(T0 + K, F):
            run(k, filt=False)
        barrier()
        fe_elapsed = shard.max_over_ranks(time.perf_counter() - t1)
        timing_fe = eng.read_timing()               # spans of the front-end-only loop (reads reset the span list)

    feats = eng.read_features()                      # raises on any device-side overflow
    cnts = eng.read_all_counters()
    n_t = float(np.mean([c['before_tracking'] for c in cnts]))
    n_trk = float(np.mean([c['after_tracking'] for c in cnts]))
    n_cand = float(np.mean([c['n_candidates'] for c in cnts]))
    mc = [eng.read_match_counts(s) for s in range(0, S, max(1, S // 64))]          # a sample of the streams
    n_match1, n_match2 = float(np.mean([m[0] for m in mc])), float(np.mean([m[1] for m in mc]))
    n_pub = float(np.mean([len(f[0]) for f in feats]))
    # counter reduction is the only end-of-run exchange (no data-path collective)
    n_t, n_trk, n_cand, n_pub, n_match1, n_match2 = [float(v) / world for v in shard.sum_over_ranks([n_t, n_trk, n_cand, n_pub, n_match1, n_match2])]

    # ---- the batch sizes the EuRoC configs really have (latency-bound regimes), same frames, own small engine pairs ----
    regimes = None
    if with_msckf and world == 1 and not args.no_regimes and tuple(args.grid) == (4, 5, 15):
        n_pre_r = PREROLL_FULL + 1
        n_st_r = max(1, min(20, F - n_pre_r))
        regimes = []
        for name, grid_r, S_r, host_r, what in (
                ('single_stream', (4, 5, 15), 1, False, 'BASELINE configs[2] batch size: one sequence on one GPU (also configs[3] per GPU)'),
                ('offset_sweep_share_8x1500', (10, 15, 10), min(8, S), False, 'BASELINE configs[4] per-GPU share: 8 streams at grid 10x15x10 = 1,500 features'),
                ('run_bat_64', (4, 5, 15), min(64, S), False, 'run.bat:4-12 on one GPU: 9 sequences x 7 offsets = 63 streams'),
                ('host_fed_128', (4, 5, 15), min(128, S), True, 'complete path with the frames handed over as host arrays every step (PCIe-inclusive; never the contract value)')):
            try:
                regimes.append(measure_regime(torch, dev, local_rank, name, grid_r, S_r, img0, img1, imu_steps, frame_ts, n_pre_r, n_st_r, host=host_r, what=what))
            except Exception as e:          # a regime line must never take the headline down
                regimes.append(dict(name=name, error=repr(e)))
    fps = world * S * K / elapsed
    b_frame, p_frame = frame_bytes(n_t, n_trk, n_match1 + n_match2)      # LK point passes actually run (lazy candidate matching)
    lk_ms, lk_n = timing['lk']
    lk_avg_ms = lk_ms / max(lk_n, 1)
    lk_launches_per_step = lk_n / float(K) if lk_n else 5.0                 # temporal, stereo fwd/bwd, candidates round 1 fwd/bwd (+ round 2)
    lk_bytes_per_launch = S * p_frame * LK_BYTES_PER_POINT_PASS / lk_launches_per_step
    lk_gbs = lk_bytes_per_launch / (lk_avg_ms * 1e-3) / 1e9 if lk_avg_ms > 0 else 0.0

    if rank == 0:
        out = {
            'metric': 'stereo frames/sec (LK+stereo+MSCKF-update) at 752x480' if with_msckf else 'stereo frames/sec (LK+stereo front-end) at 752x480',
            'value': fps, 'unit': 'stereo frames/s', 'n_gpus': world, 'steps': K, 'warmup': Wm,
            'ms_per_step': 1e3 * elapsed / K, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u8/int32 windows, f32 normal equations', 'data': 'synthetic',
            'inputs': 'host numpy arrays, H2D inside the timed region (PCIe-inclusive)' if args.host_images else 'resident in HBM',
            'config': {
                'workload': ('BASELINE configs[%s] shape (synthetic 752x480 stereo streams, grid %dx%dx%d = %d features/frame): temporal LK + '
                             % ('1' if n_grid_feat <= 300 else '4', args.grid[0], args.grid[1], args.grid[2], n_grid_feat) +
                             'stereo LK fwd/bwd + gates + FAST/grid add/prune/publish on device, ' +
                             ('followed in the same step by the batched HIP MSCKF (propagation, augmentation, triangulation, Jacobians + '
                              'null-space + gate, QR-compressed update, pruning) on the published features'
                              if with_msckf else 'MSCKF not in the step (configs[1] literally)')),
                'streams_per_gpu': S, 'unique_rendered_streams': U, 'parallelism': 'stream-sharded x%d' % world,
                'pipelines_per_gpu': n_pipes,
                'replication': ('%d distinct streams (own scene, trajectory speed 0.7-1.6x, 0-2 s phase; stream 0 = the stream the CPU path replays), '
                                'replicated to %d per GPU with +-2 grey levels of per-replica noise' % (U, S)) +
                               ('; the filters of the odd replicas start one step later, so half of the filters run the two-camera prune on even and half on odd steps'
                                if stagger else '; all filters start at step 0, so all of them prune on the same steps (lock-step load)'),
                'tracked_features_per_frame': n_t, 'published_features_per_frame': n_pub,
                'lk_point_passes_per_frame': p_frame, 'algorithmic_bytes_per_frame': b_frame,
                'candidates_per_frame': n_cand, 'candidates_stereo_matched_per_frame': [n_match1, n_match2],
                'frame_hbm_frac': fps / world * b_frame / 1e9 / HBM_PEAK_GBS,
                'cu_partition': {'frontend': os.environ.get('AV_FE_CUS'), 'filter': os.environ.get('AV_MSCKF_CUS')},
            },
            'steady_state': {
                'prerolled_frames': PRE,
                'cam_states_at_t0': [c_t0['min_cam_states'], c_t0['max_cam_states']] if c_t0 else None,
                'prune_steps_in_timed_region': ((c_t1['prune_stream_steps'] - c_t0['prune_stream_steps']) / float(S)) if c_t0 else None,
                'device_buffer_reallocations_in_timed_region': (c_t1['devbuf_growths'] - c_t0['devbuf_growths']) if c_t0 else None,
                'two_pass_streams_in_timed_region': (c_t1['two_pass_streams'] - c_t0['two_pass_streams']) if c_t0 else None,
            },
            'roofline': {
                'bound': 'hbm', 'kernel': 'lk_track_g16_kernel<15>',
                'achieved': lk_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': lk_gbs / HBM_PEAK_GBS,
                # the counter passes ran the 300-feature workload (1,113.9 point passes per stream-frame over 7 launches); another grid
                # changes the points per launch, so the per-launch traffic is scaled by point passes as well as by streams
                'traffic': LK_TRAFFIC_BYTES_PER_LAUNCH_S64 * S / LK_PMC_STREAMS * ((p_frame / lk_launches_per_step) / LK_PMC_POINT_PASSES_PER_STREAM_LAUNCH),
                'traffic_source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes): FETCH_SIZE at %d streams per launch (%s: the largest batch '
                                  'rocprofv3 --pmc collects on this image; per-stream traffic +2.8 %% from 64 to 512 streams), WRITE_SIZE at 64 (%s), FETCH x%.1f '
                                  '(MI355X_MICROARCH.md; confirmed for 4-byte lane loads by profiles/r03/fetch_calib.json); scaled linearly to %d streams'
                                  % (LK_PMC['fetch_streams'], LK_PMC.get('fetch_path', LK_PMC['path']), LK_PMC['path'], FETCH_SIZE_FACTOR, S),
                # the kernel is VALU-issue bound: instructions issued / what the chip's 1,024 SIMDs could issue in the launch's duration
                'valu_issue_frac': ((LK_VALU_INSTS_PER_LAUNCH_S64 * S / LK_PMC_STREAMS * ((p_frame / lk_launches_per_step) / LK_PMC_POINT_PASSES_PER_STREAM_LAUNCH)) * VALU_CYCLES_PER_WAVE_INST / (N_SIMD * CLOCK_HZ) / (lk_avg_ms * 1e-3)) if lk_avg_ms > 0 else None,
                'valu_issue_frac_frontend_only': ((LK_VALU_INSTS_PER_LAUNCH_S64 * S / LK_PMC_STREAMS * ((p_frame / lk_launches_per_step) / LK_PMC_POINT_PASSES_PER_STREAM_LAUNCH)) * VALU_CYCLES_PER_WAVE_INST / (N_SIMD * CLOCK_HZ) /
                                                  (timing_fe['lk'][0] / max(timing_fe['lk'][1], 1) * 1e-3)) if timing_fe else None,
                'note': 'lk_track_g16_kernel is VALU-issue bound (PMC at 64 streams: 21.4 M VALU wave-instructions per launch on the mean over a step\'s launch mix, 38% of wave '
                        'cycles waiting on an instruction, LDS 2% of instructions); its tiles come from L2/Infinity Cache. The HBM fraction is '
                        'reported because the path class is byte/integer work; valu_issue_frac (2.93 cycles per wave64 instruction of this kernel\'s mix = the guide\'s 2 x the measured relative price of its half-rate instructions: '
                        'profiles/r05/valu_issue_microbench.json and lk_dma_experiment.md; 1,024 SIMDs at 2.4 GHz) is the bound that binds. In the complete path the span also contains the higher-priority filter kernels '
                        'that preempt it.',
                'avg_launch_ms': lk_avg_ms, 'launches': lk_n, 'algorithmic_bytes_per_launch': lk_bytes_per_launch,
                # --pipelines P: a launch covers S / P streams and runs beside the kernels of the other P - 1 pipelines (their launches are summed
                # over the pipelines: `launches` = 7 P per step): `achieved` / `frac` are what ONE launch gets of the machine it shares
                'streams_per_launch': S // n_pipes, 'pipelines': n_pipes,
                # same kernel, same inputs, timed in the front-end-only loop that follows (no filter kernels sharing the GPU)
                'avg_launch_ms_frontend_only': (timing_fe['lk'][0] / max(timing_fe['lk'][1], 1)) if timing_fe else None,
            },
            'kernel_ms_per_step': {k: v[0] / K for k, v in timing.items()},
            'pmc_constants': dict(LK_PMC, fetch_size_factor=FETCH_SIZE_FACTOR, fetch_calibration='profiles/r03/fetch_calib.json'),
            'data_gen_s': gen_s,
            'filter_state': ('device-resident (state, observation map, selections and injection on the device; the feature message is read from the front-end\'s device buffers)' if (flt is not None and flt.device_resident()) else ('host bookkeeping (AV_MSCKF_STORE=host)' if flt is not None else None)),
            'msckf_in_step': with_msckf, 'msckf_wall_ms_per_step': 1e3 * msckf_s[0] / K, 'msckf_push_imu_ms_per_step': 1e3 * push_s[0] / K,
            'frontend_only_frames_per_s': (world * S * K / fe_elapsed) if fe_elapsed else None,
        }
        if regimes is not None:
            out['regimes'] = regimes
        if w_t0 is not None:
            dw = {k: w_t1[k] - w_t0[k] for k in w_t1}
            fl = dw['gate_flops'] + dw['update_flops']
            tf = fl / (dw['chain_ms'] * 1e-3) / 1e12 if dw['chain_ms'] > 0 else 0.0
            out['roofline_msckf'] = {
                'bound': 'fp64 FMA peak (vector = matrix instruction on MI355X: profiles/r05/mfma_f64_probe.json); the stage itself is bound by dependent chains and placement beside the front-end (SURVEY 8d)',
                'kernels': 'triangulate, feature_kernel<256> / feature_kernel8, upd_stack, upd_rowmap, upd_gram_mfma / upd_gram_chol_mfma, upd_gather, upd_tt_mfma / upd_s_mfma / upd_chol_mfma / upd_fsolve_mfma / upd_p_mfma (single-wavefront tasks on v_mfma_f64_16x16x4), upd_info (device-resident filter: the spans also hold dk_mid)',
                'achieved': tf, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / FP64_PEAK_TFLOPS,
                'algorithmic_flops_per_step': fl / K, 'gate_flops_per_step': dw['gate_flops'] / K, 'update_flops_per_step': dw['update_flops'] / K,
                'reference_qr_flops_per_step_not_counted': dw['reference_qr_flops'] / K,
                'features_gated_per_stream_step': dw['features_gated'] / K / S, 'rows_stacked_per_update': dw['rows_stacked'] / max(dw['updates'], 1.0),
                'updates_per_stream_step': dw['updates'] / K / S,
                'chain_ms_per_step': dw['chain_ms'] / K,
                # the same stage by the flops the kernels EXECUTE on their block-sparse shapes (av_msckf_batch_work_executed: analytic
                # per gated feature / per update; the matrix-instruction products WITH their 16 x 16 x 4 padding) -- the dense formulas above price a 21-observation gate
                # at 5.5 MFLOP where feature_kernel runs ~0.6
                'executed': {'flops_per_step': (dw['gate_flops_executed'] + dw['update_flops_executed']) / K,
                             'gate_flops_per_step': dw['gate_flops_executed'] / K, 'update_flops_per_step': dw['update_flops_executed'] / K,
                             'achieved': ((dw['gate_flops_executed'] + dw['update_flops_executed']) / (dw['chain_ms'] * 1e-3) / 1e12) if dw['chain_ms'] > 0 else 0.0,
                             'frac': ((dw['gate_flops_executed'] + dw['update_flops_executed']) / (dw['chain_ms'] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if dw['chain_ms'] > 0 else 0.0,
                             'exclusive_frac': None if w_e0 is None or w_t0['chain_ms'] <= w_e0['chain_ms'] else
                                 ((w_t0['gate_flops_executed'] - w_e0['gate_flops_executed']) + (w_t0['update_flops_executed'] - w_e0['update_flops_executed']))
                                 / ((w_t0['chain_ms'] - w_e0['chain_ms']) * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
                'exclusive': None if w_e0 is None or w_t0['chain_ms'] <= w_e0['chain_ms'] else {
                    'chain_ms_per_step': (w_t0['chain_ms'] - w_e0['chain_ms']) / Wm,
                    'achieved': ((w_t0['gate_flops'] - w_e0['gate_flops']) + (w_t0['update_flops'] - w_e0['update_flops'])) / ((w_t0['chain_ms'] - w_e0['chain_ms']) * 1e-3) / 1e12,
                    'frac': ((w_t0['gate_flops'] - w_e0['gate_flops']) + (w_t0['update_flops'] - w_e0['update_flops'])) / ((w_t0['chain_ms'] - w_e0['chain_ms']) * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                    'steps': Wm,
                    'how': 'the same counters and events over the %d warm-up steps, which this bench runs with front-end and filter strictly alternating '
                           '(a device synchronisation between them): the filter\'s chains with the GPU to themselves' % Wm},
                'how': 'flops: SURVEY 8(d) formulas on the sizes actually processed (gate: r = 4M-3 rows x n; update: k = min(m, n) rows kept), '
                       'the reference-size thin QR (2mn^2 - 2/3 n^3, msckf.py:554) listed apart because the column-compressed update never runs it; '
                       'time: HIP events on every stream group\'s stream around the launches of each phase (triangulation .. covariance update), '
                       'summed over the %d groups, inside the timed region of the complete path (so it includes waiting for CUs the front-end holds)'
                       % int(os.environ.get('AV_MSCKF_GROUPS', ('2' if S >= 1024 else '1') if flt.device_resident() else ('4' if S >= 256 else ('2' if S >= 64 else '1')))),
            }
        if world == 1 and not args.no_cpu_baseline:
            n_traj = (T0 + K) if with_msckf else 0
            cb = cpu_baseline(cfg, with_msckf, seed=stream_seed0, max_frames=max(400, n_traj), traj_frames=n_traj)
            cpu_traj, truth = cb.pop('_traj', None), cb.pop('_truth', None)
            out['cpu_baseline'] = cb
            out['cpu_baseline_all_cores'] = cpu_baseline_all_cores(with_msckf, grid=args.grid)
            if with_msckf and cpu_traj:
                from uav_airvision_amd.evaluate import ate
                gpu_traj = np.array([[frame_ts[k][0]] + [float(v) for v in o[0, 2:9]] for k, o in sorted(poses0, key=lambda e: e[0]) if o[0, 0] > 0.5])
                cpu_traj, truth = np.array(cpu_traj), np.array(truth)
                a_gc, a_gt, a_ct = ate(gpu_traj, cpu_traj, max_dt=1e-6), ate(gpu_traj, truth, max_dt=1e-6), ate(cpu_traj, truth, max_dt=1e-6)
                n_c = min(len(gpu_traj), len(cpu_traj))
                out['ate_vs_cpu_ref'] = {
                    'ate_rmse_gpu_vs_cpu_m': a_gc['rmse'], 'ate_rmse_gpu_vs_truth_m': a_gt['rmse'], 'ate_rmse_cpu_vs_truth_m': a_ct['rmse'],
                    'relative_difference_of_ate_vs_truth': abs(a_gt['rmse'] - a_ct['rmse']) / max(a_ct['rmse'], 1e-30),
                    'max_abs_position_difference_m': float(np.abs(gpu_traj[:n_c, 1:4] - cpu_traj[:n_c, 1:4]).max()),
                    'frames': int(a_gc['n']),
                    'what': 'stream 0 of the GPU batch (frames 0..%d, pre-roll included) against the CPU port of the reference path on the same '
                            'images and IMU samples; truth = the synthetic trajectory; ATE = RMSE after SE(3) alignment '
                            '(uav_airvision_amd/evaluate.py)' % (T0 + K - 1)}
        # sanity of the roofline figures: the line is printed either way (a failed check must not cost the measurement), the exit
        # code says whether it can be trusted
        r_ = out['roofline']
        failed = []
        if not (0.0 < r_['frac'] <= 1.0):
            failed.append('roofline.frac out of range: %r' % r_['frac'])
        if not (r_['traffic'] >= r_['algorithmic_bytes_per_launch']):
            failed.append('measured HBM traffic below the algorithmic bytes: %r < %r' % (r_['traffic'], r_['algorithmic_bytes_per_launch']))
        rm_ = out.get('roofline_msckf')
        if rm_ is not None and not (0.0 <= rm_['frac'] <= 1.0):
            failed.append('roofline_msckf.frac out of range: %r' % rm_['frac'])
        if rm_ is not None and rm_.get('executed') and not (rm_['executed']['flops_per_step'] <= rm_['algorithmic_flops_per_step']):
            failed.append('roofline_msckf: executed flops above the dense-formula count: %r > %r' % (rm_['executed']['flops_per_step'], rm_['algorithmic_flops_per_step']))
        if failed:
            out['roofline_check_failed'] = failed
            bench_rc = 3
        print(json.dumps(out))
    eng.close()
    if flt is not None:
        flt.close()
    if world > 1:
        dist.destroy_process_group()
    return bench_rc


if __name__ == '__main__':
    sys.exit(main())
