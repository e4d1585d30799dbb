#!/usr/bin/env python3
"""bench.py -- stereo frames/s of the MI355X image front-end on synthetic 752x480 streams.

Workload = BASELINE.json configs[1]: "Synthetic 752x480 stereo stream, 300 tracked features,
LK+stereo kernels on 1 MI355X" -- the reference front-end (ImageProcessingPipeline.stereo_callback:
temporal LK, stereo LK fwd/bwd + gates, FAST + grid add/prune, publish) with grid 4x5 x 15 = 300
features per frame, run for S independent streams per GPU (frames of one stream are sequential,
streams are the parallel axis).  One "step" = one stereo frame of every stream.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement).  Inputs are resident in HBM before
the timed region; IMU samples are pushed from the host inside it (they are part of the path).
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
# Memory-side traffic of lk_track_g16_kernel per point pass, from rocprofv3 PMC passes of this same command
# (profiles/r01/pmc_frontend_s64_lk_g16_summary.json: FETCH_SIZE 68,263 KB and WRITE_SIZE 295 KB per launch of 19,200
# point passes at 64 streams), corrected as MI355X_MICROARCH.md "HBM" prescribes for gfx950 (FETCH_SIZE x 2;
# WRITE_SIZE exact).  The x2 rule is calibrated for 16-B/lane streams; this kernel stages 32-byte row segments, so
# read it as an upper bound (uncorrected: 3.7 KB per point pass against 2.3 KB algorithmic).
LK_TRAFFIC_BYTES_PER_POINT_PASS = (2 * 68263.08 + 295.15) * 1024 / 19200


def frame_bytes(n_t, n_trk, n_cand):
    """Algorithmic bytes of one stereo frame: 2 images + 2x3 pyramid levels + LK point passes."""
    p = n_t + 2 * n_trk + 2 * n_cand
    return 2 * W_IMG * H_IMG + 2 * 118440 + p * LK_BYTES_PER_POINT_PASS, p


def make_config():
    from uav_airvision_amd.config import ConfigEuRoC
    return ConfigEuRoC(grid_row=4, grid_col=5, grid_min_feature_num=3, grid_max_feature_num=15)


def cpu_baseline(cfg, with_msckf, budget_s=10.0, max_frames=400, seed=0):
    """The CPU oracle (oracle/: scalar C image ops + Python glue + numpy MSCKF = a port of the reference's
    CPU path) on ONE stream of the same workload, single thread, bounded sample."""
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
    warm = 24 if with_msckf else 8             # grid full (~300 features) and 20 camera states before timing
    for k in range(max_frames):
        m = st.frame(k)
        while pend is not None and pend.timestamp <= m.timestamp:
            fe.imu_callback(pend)
            if flt is not None:
                flt.imu_callback(pend)
            pend = next(it, None)
        t0 = time.perf_counter()
        msg = fe.stereo_callback(m)
        if flt is not None:
            flt.feature_callback(msg)
        dt = time.perf_counter() - t0
        if k >= warm:
            spent += dt
            n += 1
            frames.append(len(msg.features))
            if spent >= budget_s:
                break
    return dict(value=n / spent, unit='stereo frames/s', cores=1, kind='port',
                sample='%d frames of 1 synthetic stream after %d warm-up frames, %.1f s of CPU, mean %d features/frame, %s'
                       % (n, warm, spent, int(np.mean(frames)), 'front-end + MSCKF' if with_msckf else 'front-end only'))


def cpu_baseline_all_cores(with_msckf, budget_s=6.0):
    """The same single-thread port, one independent stream per host core in parallel child processes (they never touch the
    GPU): the CPU box's aggregate rate on this workload, for context next to the one-core figure."""
    import subprocess
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    n = max(1, min(16, n))                     # a GPU box gives one GPU's job a share of 16 cores
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--cpu-budget', str(budget_s)]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--streams', type=int, default=1024, help='independent stereo streams per GPU')
    ap.add_argument('--unique', type=int, default=4, help='distinct rendered streams (replicated with per-stream noise)')
    ap.add_argument('--host-images', action='store_true', help='front-end only, images handed over as host numpy arrays every step '
                    '(av_frontend_step_host: the PCIe-inclusive rate quoted in DESIGN.md; never the contract value)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-budget', type=float, default=6.0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-seed', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--frontend-only', action='store_true', help='time only the image front-end (BASELINE configs[1] literally: MSCKF not on the GPU)')
    args = ap.parse_args()
    if args.cpu_baseline_worker:               # child of cpu_baseline_all_cores: CPU only, exits before torch is imported
        print(json.dumps(cpu_baseline(make_config(), not args.frontend_only, budget_s=args.cpu_budget, seed=args.cpu_seed)))
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
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
    cfg = make_config()
    S, K, Wm = args.streams, args.steps, args.warmup
    with_msckf = not (args.frontend_only or args.host_images)
    F = Wm + K + (K if with_msckf else 0)      # a second timed loop measures the front-end alone
    # config/seed broadcast from rank 0 over RCCL (config broadcast, no data-path collective)
    from uav_airvision_amd import shard
    run_cfg = shard.broadcast_object({'seed': 1234, 'streams': S, 'steps': K, 'warmup': Wm} if rank == 0 else None)
    base_seed = int(run_cfg['seed'])

    # ---- synthetic data: U rendered streams, replicated to S streams with per-stream pixel noise ----
    U = max(1, min(args.unique, S))
    t_gen = time.time()
    streams = [SyntheticStream(cfg, seed=base_seed + 97 * rank + u, n_frames=F) for u in range(U)]
    base0 = np.empty((U, F, H_IMG, W_IMG), np.uint8)
    base1 = np.empty((U, F, H_IMG, W_IMG), np.uint8)
    for u, st in enumerate(streams):
        for k in range(F):
            m = st.frame(k)
            base0[u, k] = m.cam0_image
            base1[u, k] = m.cam1_image
    g = torch.Generator(device=dev)
    g.manual_seed(base_seed + rank)
    img0 = torch.empty((F, S, H_IMG, W_IMG), dtype=torch.uint8, device=dev)
    img1 = torch.empty((F, S, H_IMG, W_IMG), dtype=torch.uint8, device=dev)
    b0 = torch.from_numpy(base0).to(dev)
    b1 = torch.from_numpy(base1).to(dev)
    for s in range(S):
        u = s % U
        for dst, src in ((img0, b0), (img1, b1)):
            noise = torch.randint(-2, 3, (F, H_IMG, W_IMG), generator=g, device=dev, dtype=torch.int16)
            dst[:, s] = (src[u].to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    del b0, b1
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
    frame_ts = [[streams[s % U].frame_time(k) for s in range(S)] for k in range(F)]
    gen_s = time.time() - t_gen

    eng = FrontendEngine(cfg, n_streams=S, device=local_rank)
    flt = None
    if with_msckf:
        from uav_airvision_amd.msckf_ops import BatchedMSCKF
        flt = BatchedMSCKF(cfg, S, device=local_rank, rows_cap=4096)
    msckf_s = [0.0]
    push_s = [0.0]

    filt_stream = torch.cuda.Stream(device=dev) if flt is not None else None

    host0 = host1 = None
    if args.host_images:                         # the same frames as pageable host arrays; the device copies are dropped
        host0, host1 = img0.cpu().numpy(), img1.cpu().numpy()
        del img0, img1
        torch.cuda.empty_cache()

    def run_fe(k):
        i, t, gy, ac = imu_steps[k]
        eng.push_imu_batch(i, t, gy)
        if host0 is not None:
            eng.step_host(host0[k], host1[k], frame_ts[k])
        else:
            eng.step(img0[k], img1[k], frame_ts[k])

    def run_filter(k, ids_h, uv_h, n_h, queued=False):
        t1 = time.perf_counter()
        i, t, gy, ac = imu_steps[k]
        flt.push_imu(i, t, gy, ac)
        push_s[0] += time.perf_counter() - t1
        with torch.cuda.stream(filt_stream):         # the filter's kernels overlap the next frame's front-end kernels
            if queued:
                flt.submit(ids_h, uv_h, n_h, frame_ts[k])   # the stream groups run behind their own queues ...
                flt.wait(1)                                 # ... at most one frame ahead of the slowest group
            else:
                flt.step(ids_h, uv_h, n_h, frame_ts[k])
        msckf_s[0] += time.perf_counter() - t1

    def run_pipelined(k_begin, k_end):
        """Full path for frames [k_begin, k_end), organised like the reference's VIO (vio.py:24-76: image thread ->
        feature queue -> filter thread): this thread drives the front-end and hands every frame's feature message to a
        filter thread through a bounded queue, so the front-end of later frames overlaps the (host-blocking) filter step
        of earlier ones.  The timed region ends when the last filter step has returned."""
        import queue, threading
        q = queue.Queue(maxsize=2)
        err = []

        def filter_loop():
            torch.cuda.set_device(dev)
            while True:
                item = q.get()
                if item is None:
                    return
                try:
                    if not err:
                        run_filter(*item, queued=True)
                except Exception as e:          # surfaced by the main thread after the join
                    err.append(e)

        th = threading.Thread(target=filter_loop, name='msckf')
        th.start()
        try:
            # frame k+1 is enqueued before frame k's features are consumed: the read-back of k (pinned, double buffered)
            # sits between the two steps on the stream, so the GPU never waits for this thread
            run_fe(k_begin)
            eng.read_features_begin(k_begin & 1)
            for k in range(k_begin, k_end):
                if k + 1 < k_end:
                    run_fe(k + 1)
                    eng.read_features_begin((k + 1) & 1)
                ids_h, uv_h, n_h = eng.read_features_end(k & 1)      # waits for frame k's copy only; fresh host arrays
                q.put((k, ids_h, uv_h, n_h))
        finally:
            q.put(None)
            th.join()
        if err:
            raise err[0]
        flt.wait(0)                                             # the last queued step retires inside the timed region

    def run(k, filt=True):
        run_fe(k)
        if flt is not None and filt:
            ids_h, uv_h, n_h = eng.read_features_raw()
            run_filter(k, ids_h, uv_h, n_h)

    for k in range(Wm):
        run(k)
    eng.read_features()                              # sync + overflow check of the warm-up
    eng.enable_timing(16 * (F - Wm) + 16)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    msckf_s[0] = 0.0
    push_s[0] = 0.0
    t0 = time.perf_counter()
    if flt is not None:
        run_pipelined(Wm, Wm + K)
    else:
        for k in range(Wm, Wm + K):
            run(k)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed)
    timing = eng.read_timing()
    fe_elapsed = None
    timing_fe = None
    if with_msckf:                                   # same engine state, next K frames, front-end only
        barrier()
        t1 = time.perf_counter()
        for k in range(Wm + K, F):
            run(k, filt=False)
        barrier()
        fe_elapsed = shard.max_over_ranks(time.perf_counter() - t1)
        timing_fe = eng.read_timing()               # spans of the front-end-only loop (reads reset the span list)

    feats = eng.read_features()                      # raises on any device-side overflow
    cnts = eng.read_all_counters()
    n_t = float(np.mean([c['before_tracking'] for c in cnts]))
    n_trk = float(np.mean([c['after_tracking'] for c in cnts]))
    n_cand = float(np.mean([c['n_candidates'] for c in cnts]))
    n_pub = float(np.mean([len(f[0]) for f in feats]))
    # counter reduction is the only end-of-run exchange (no data-path collective)
    n_t, n_trk, n_cand, n_pub = [float(v) / world for v in shard.sum_over_ranks([n_t, n_trk, n_cand, n_pub])]

    fps = world * S * K / elapsed
    b_frame, p_frame = frame_bytes(n_t, n_trk, n_cand)
    lk_ms, lk_n = timing['lk']
    lk_avg_ms = lk_ms / max(lk_n, 1)
    lk_bytes_per_launch = S * p_frame * LK_BYTES_PER_POINT_PASS / 5.0       # 5 LK launches per step
    lk_gbs = lk_bytes_per_launch / (lk_avg_ms * 1e-3) / 1e9 if lk_avg_ms > 0 else 0.0

    if rank == 0:
        out = {
            'metric': 'stereo frames/sec (LK+stereo+MSCKF-update) at 752x480' if with_msckf else 'stereo frames/sec (LK+stereo front-end) at 752x480',
            'value': fps, 'unit': 'stereo frames/s', 'n_gpus': world, 'steps': K, 'warmup': Wm,
            'ms_per_step': 1e3 * elapsed / K, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u8/int32 windows, f32 normal equations', 'data': 'synthetic',
            'inputs': 'host numpy arrays, H2D inside the timed region (PCIe-inclusive)' if args.host_images else 'resident in HBM',
            'config': {
                'workload': ('BASELINE configs[1] shape (synthetic 752x480 stereo streams, grid 4x5x15 = 300 features/frame): temporal LK + '
                             'stereo LK fwd/bwd + gates + FAST/grid add/prune/publish on device, ' +
                             ('followed in the same step by the batched HIP MSCKF (propagation, augmentation, triangulation, Jacobians + '
                              'null-space + gate, QR-compressed update, pruning) on the published features'
                              if with_msckf else 'MSCKF not in the step (configs[1] literally)')),
                'streams_per_gpu': S, 'unique_rendered_streams': U, 'parallelism': 'stream-sharded x%d' % world,
                'tracked_features_per_frame': n_t, 'published_features_per_frame': n_pub,
                'lk_point_passes_per_frame': p_frame, 'algorithmic_bytes_per_frame': b_frame,
                'frame_hbm_frac': fps / world * b_frame / 1e9 / HBM_PEAK_GBS,
            },
            'roofline': {
                'bound': 'hbm', 'kernel': 'lk_track_g16_kernel<15>',
                'achieved': lk_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': lk_gbs / HBM_PEAK_GBS,
                'traffic': LK_TRAFFIC_BYTES_PER_POINT_PASS * S * p_frame / 5.0,
                'traffic_source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), profiles/r01/pmc_frontend_s64_lk_g16_summary.json, FETCH x2 per MI355X_MICROARCH.md',
                'note': 'lk_track_g16_kernel is VALU-issue bound (PMC: ~1,400 VALU instructions per point pass, 32% of wave cycles '
                        'waiting on an instruction, LDS 2% of instructions); its tiles come from L2/Infinity Cache. The HBM fraction is '
                        'reported because the path class is byte/integer work. In the complete path the span also contains the '
                        'higher-priority filter kernels that preempt it.',
                'avg_launch_ms': lk_avg_ms, 'launches': lk_n, 'algorithmic_bytes_per_launch': lk_bytes_per_launch,
                # same kernel, same inputs, timed in the front-end-only loop that follows (no filter kernels sharing the GPU)
                'avg_launch_ms_frontend_only': (timing_fe['lk'][0] / max(timing_fe['lk'][1], 1)) if timing_fe else None,
            },
            'kernel_ms_per_step': {k: v[0] / K for k, v in timing.items()},
            'data_gen_s': gen_s,
            'msckf_in_step': with_msckf, 'msckf_wall_ms_per_step': 1e3 * msckf_s[0] / K, 'msckf_push_imu_ms_per_step': 1e3 * push_s[0] / K,
            'frontend_only_frames_per_s': (world * S * K / fe_elapsed) if fe_elapsed else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, with_msckf)
            out['cpu_baseline_all_cores'] = cpu_baseline_all_cores(with_msckf)
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
