"""EuRoC-layout sweeps end to end (PNG files -> decode -> upload -> front-end -> MSCKF) with and without the shared frame store.
Run on the GPU box:  python profiles/r05/sweep_throughput.py  -> one JSON line.

 (A) BASELINE configs[4]'s shape: ONE sequence x 64 start offsets (run.bat:4-12 sweeps offsets of a sequence; an offset only moves
     the start index, dataset.py:206-214): 64 streams of ragged length that read the same files a few steps apart.
 (B) round 4's workload for comparison (4 sequences x 16 offsets x 150 frames, `profiles/r04/sweep_throughput.py`).
Each is run through `BatchedRunner.run` twice: share_frames=True (every distinct frame decoded / uploaded / pyramided / FAST-scanned
once, finished streams launch nothing) and share_frames=False (round 4: every stream stages its own copy of every frame)."""
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _write(args):
    root, seed, n, a, b, t0 = args
    sys.path.insert(0, ROOT)
    from uav_airvision_amd.config import ConfigEuRoC
    from uav_airvision_amd.euroc import write_euroc_layout
    from uav_airvision_amd.synth import SyntheticStream
    st = SyntheticStream(ConfigEuRoC(), seed=seed, n_frames=n, motion_scale=1.5, t0=t0, rest=1.0)
    write_euroc_layout(root, st, frame_range=(a, b), write_csv=(a == 0), compress_level=1)
    return b - a


def run(cfg, roots_offsets, share, threads):
    from uav_airvision_amd.euroc import EuRoCDataset
    from uav_airvision_amd.sweep import BatchedRunner
    dss = []
    for r, o in roots_offsets:
        ds = EuRoCDataset(r); ds.set_starttime(o); dss.append(ds)
    runner = BatchedRunner(cfg, len(dss))
    t0 = time.time()
    trajs = runner.run(dss, host_threads=threads, share_frames=share)
    dt = time.time() - t0
    done = int(runner.frames_done.sum())
    out = {'share_frames': share, 'streams': len(dss), 'steps': int(runner.steps_done), 'stream_frames': done, 'seconds': dt,
           'stream_frames_per_s': done / dt, 'ms_per_step': 1e3 * dt / max(runner.steps_done, 1),
           'frames_decoded': (runner.plan.n_frames_distinct if runner.plan is not None else done),
           'store_entries': (runner.plan.n_slots if runner.plan is not None else None),
           'filter_frames_published': int(sum(len(t) for t in trajs))}
    runner.close()
    return out, trajs


def main():
    import numpy as np
    tmp = tempfile.mkdtemp(prefix='sweep_tp_')
    NT = int(os.environ.get('AV_DECODE_THREADS', '16'))
    nA = int(os.environ.get('AV_SWEEP_FRAMES', '900'))
    rootA = os.path.join(tmp, 'SYN_MH_03')
    rootsB = [os.path.join(tmp, 'SYN_%d' % i) for i in range(4)]
    with mp.get_context('spawn').Pool(min(16, os.cpu_count() or 8)) as pool:
        jobs = [(rootA, 700, nA, a, min(nA, a + 10), 1403636580.0) for a in range(0, nA, 10)]
        jobs += [(rootsB[i], 500 + i, 150, a, min(150, a + 10), 1403636580.0 + 1000 * i) for i in range(4) for a in range(0, 150, 10)]
        t0 = time.time(); pool.map(_write, jobs); t_write = time.time() - t0
    from uav_airvision_amd.config import ConfigEuRoC
    cfg = ConfigEuRoC()
    A = [(rootA, 0.25 * k) for k in range(64)]                        # 64 offsets, 5 frames apart: streams of 900 .. 585 frames
    B = [(r, 0.4 * k) for r in rootsB for k in range(16)]
    res = {}
    a1, tA1 = run(cfg, A, True, NT)
    b1, _ = run(cfg, B, True, NT)
    b0, _ = run(cfg, B, False, NT)
    # the unshared run of (A) on a prefix (it is ~10x slower): same streams, the first 150 steps
    import uav_airvision_amd.sweep as sw
    from uav_airvision_amd.euroc import EuRoCDataset
    dss = []
    for r, o in A:
        ds = EuRoCDataset(r); ds.set_starttime(o); dss.append(ds)
    runner = sw.BatchedRunner(cfg, len(dss))
    t0 = time.time(); tA0 = runner.run(dss, max_frames=150, host_threads=NT, share_frames=False); dt = time.time() - t0
    a0 = {'share_frames': False, 'streams': 64, 'steps': 150, 'stream_frames': int(runner.frames_done.sum()), 'seconds': dt,
          'stream_frames_per_s': int(runner.frames_done.sum()) / dt, 'what': 'first 150 steps only'}
    runner.close()
    # the two paths publish the same trajectories (first 150 steps of A)
    same = all(np.array_equal(t1[:len(t0_)], t0_) for t1, t0_ in zip(tA1, tA0))
    print(json.dumps({'host_cpus': os.cpu_count(), 'decode_threads': NT, 'write_s': t_write,
                      'one_sequence_x_64_offsets': {'shared': a1, 'per_stream_staging_first_150_steps': a0, 'same_trajectories_on_the_common_prefix': bool(same)},
                      'four_sequences_x_16_offsets_r04_workload': {'shared': b1, 'per_stream_staging': b0}}))


if __name__ == '__main__':
    main()
