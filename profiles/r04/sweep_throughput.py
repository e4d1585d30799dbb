"""64-stream EuRoC-layout sweep through the staged path (reader -> FrameStager -> FrontendEngine(64) -> BatchedMSCKF(64)): is it
decode-bound or GPU-bound?    Run on the GPU box:  python profiles/r04/sweep_throughput.py  -> one JSON line.
Writes 4 synthetic sequences in the dataset layout, opens each at 16 start offsets (64 streams of ragged length), then times
 (a) the staging alone (FrameStager.next() until exhausted: list + threaded native PNG decode of 128 images per step),
 (b) the same frames decoded serially with Pillow, as sweep.py did in round 2 (a sample of steps),
 (c) the complete sweep (staging one step ahead of the GPU + filter queue)."""
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


def main():
    import numpy as np
    n_seq, n_off, n_frames = 4, 16, 150
    tmp = tempfile.mkdtemp(prefix='sweep_tp_')
    roots = [os.path.join(tmp, 'SYN_%d' % i) for i in range(n_seq)]
    with mp.get_context('spawn').Pool(min(16, os.cpu_count() or 8)) as pool:
        jobs = [(roots[i], 500 + i, n_frames, a, min(n_frames, a + 10), 1403636580.0 + 1000 * i) for i in range(n_seq) for a in range(0, n_frames, 10)]
        t0 = time.time(); pool.map(_write, jobs); t_write = time.time() - t0
    from uav_airvision_amd.config import ConfigEuRoC
    from uav_airvision_amd.euroc import EuRoCDataset, FrameStager, read_image
    from uav_airvision_amd.sweep import BatchedRunner
    cfg = ConfigEuRoC()

    def datasets():
        out = []
        for r in roots:
            for k in range(n_off):
                ds = EuRoCDataset(r); ds.set_starttime(0.4 * k); out.append(ds)
        return out
    dss = datasets()
    S = len(dss)
    # (a) staging alone
    NT = int(os.environ.get('AV_DECODE_THREADS', '16'))
    stg = FrameStager(dss, 480, 752, threads=NT)
    t0 = time.time(); steps = 0; frames = 0
    while True:
        nxt = stg.next()
        if nxt is None:
            break
        steps += 1; frames += int((nxt[0] >= 0).sum())
    t_stage = time.time() - t0
    stg.close()
    # (b) serial Pillow decode of a sample
    files = list(dss[0].stereo_files)[:20]
    t0 = time.time()
    for _t, p0, p1 in files:
        read_image(p0); read_image(p1)
    t_pil = (time.time() - t0) / len(files)
    # (c) the complete sweep
    runner = BatchedRunner(cfg, S)
    runner_dev = runner.flt.device_resident()
    t0 = time.time()
    trajs = runner.run(datasets(), host_threads=NT)
    t_run = time.time() - t0
    done = int(runner.frames_done.sum())
    runner.close()
    print(json.dumps({
        'streams': S, 'steps': steps, 'stream_frames': frames, 'write_s': t_write,
        'staging_alone_frames_per_s': frames / t_stage, 'staging_alone_ms_per_step': 1e3 * t_stage / steps,
        'pillow_serial_frames_per_s': 1.0 / t_pil,
        'sweep_frames_per_s': done / t_run, 'sweep_ms_per_step': 1e3 * t_run / steps, 'sweep_stream_frames': done,
        'filter_frames_published': int(sum(len(t) for t in trajs)),
        'host_cpus': os.cpu_count(), 'decode_threads': NT, 'filter_state': 'device-resident' if runner_dev else 'host bookkeeping',
        'reading': 'sweep_frames_per_s close to staging_alone_frames_per_s = decode-bound; well below it = bound by the stepping loop / GPU'}))


if __name__ == '__main__':
    main()
