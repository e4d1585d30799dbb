"""Sequence x offset sweep runner: the reference's `run.bat:4-12` (9 sequences x
7 offsets, run serially through `main.py --path DIR --offset S`) as a sharded job: every (sequence,
offset) pair is one independent stream, pairs are partitioned over ranks with `shard.partition`, each
rank runs its pairs through the drop-in `ImageProcessor` + `MSCKF` on its GPU with the deterministic
replay, and trajectories are gathered at the end (no per-frame communication).

    python -m uav_airvision_amd.sweep --root /data/euroc --sequences MH_01_easy MH_03_medium --offsets 0 10 20
    python -m torch.distributed.run --nproc-per-node 8 -m uav_airvision_amd.sweep ...
"""
import argparse
import json
import os
import sys

import numpy as np


def run_stream(config, dataset_path, offset, device=0, max_frames=None):
    """One (sequence, offset) stream through the GPU hot path; returns float64[n, 8] trajectory
    (t px py pz qx qy qz qw: the reference's output line, msckf.py:152-158)."""
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


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--root', required=True, help='directory holding the EuRoC sequences')
    ap.add_argument('--sequences', nargs='+', required=True)
    ap.add_argument('--offsets', nargs='+', type=float, default=[0.0])
    ap.add_argument('--max-frames', type=int, default=None)
    ap.add_argument('--out', default='results/txts')
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
    jobs = shard.broadcast_object([(s, o) for s in args.sequences for o in args.offsets] if rank == 0 else None)
    cfg = ConfigEuRoC()
    mine = shard.partition(len(jobs), world, rank)
    local_traj, report = {}, {}
    for j in mine:
        seq, off = jobs[j]
        traj, ds = run_stream(cfg, os.path.join(args.root, seq), off, device=local, max_frames=args.max_frames)
        local_traj[j] = traj
        gt = ds.groundtruth_array()
        if len(gt) and len(traj) > 20:
            a, r = ate(traj, gt), rte(traj, gt)
            report[j] = dict(sequence=seq, offset=off, frames=len(traj), ate_rmse=a['rmse'], ate_mean=a['mean'], rte_rmse=r['rmse'])
    allt = shard.gather_trajectories(local_traj, len(jobs), world, rank)
    if rank == 0:
        os.makedirs(args.out, exist_ok=True)
        for j, (seq, off) in enumerate(jobs):
            with open(os.path.join(args.out, 'output_%s_offset%d.txt' % (seq, int(off))), 'w') as f:
                for row in allt[j]:
                    f.write(format_state_line(row[0], row[1:4], row[4:8]))
    print(json.dumps({'rank': rank, 'streams': [jobs[j] for j in mine], 'report': list(report.values())}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
