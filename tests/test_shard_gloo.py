"""The N>1 path on CPU: two gloo ranks shard streams, broadcast the configuration, time with a
max-over-ranks and gather per-stream trajectories -- the same helpers bench.py uses with RCCL."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from uav_airvision_amd import shard
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cfg = shard.broadcast_object({'seed': 77, 'streams': 5, 'K': np.arange(4.0)} if rank == 0 else None)
    mine = shard.partition(cfg['streams'], world, rank)
    # every stream produces a deterministic "trajectory"; streams never talk to each other
    local = {s: np.full((3 + s, 8), float(cfg['seed'] + s)) for s in mine}
    t = shard.max_over_ranks(0.1 * (rank + 1))
    tot = shard.sum_over_ranks([len(mine), sum(mine)])
    allt = shard.gather_trajectories(local, cfg['streams'], world, rank)
    q.put((rank, mine, t, tot.tolist(), {k: (v.shape, float(v[0, 0])) for k, v in allt.items()}, cfg['K'].tolist()))
    dist.destroy_process_group()


def test_partition_covers_everything_once():
    from uav_airvision_amd.shard import partition
    for n in (0, 1, 5, 8, 64, 67):
        for w in (1, 2, 3, 8):
            parts = [partition(n, w, r) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_two_rank_gloo_sharding():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, t0, tot0, all0, k0), (r1, m1, t1, tot1, all1, k1) = res
    assert m0 == [0, 1, 2] and m1 == [3, 4]
    assert abs(t0 - 0.2) < 1e-12 and abs(t1 - 0.2) < 1e-12          # max over ranks
    assert tot0 == tot1 == [5.0, 10.0]
    assert all0 == all1 and sorted(all0) == [0, 1, 2, 3, 4]
    assert all0[4] == ((7, 8), 81.0) and k0 == k1 == [0.0, 1.0, 2.0, 3.0]


REF_TXTS = '/root/reference/results/txts'


def _eight_trajectories():
    """BASELINE configs[3]'s gather: one trajectory per EuRoC sequence, ragged lengths.  In the build container these ARE the
    reference's eight result files (results/txts/output_<seq>_offset<o>.txt: 295 ... 2,862 poses); elsewhere, arrays of the same
    shape family from a seeded generator."""
    from uav_airvision_amd import evaluate
    if os.path.isdir(REF_TXTS):
        return [evaluate.load_trajectory_txt(os.path.join(REF_TXTS, f)) for f in sorted(os.listdir(REF_TXTS))]
    rng = np.random.default_rng(8)
    out = []
    for n in (2862, 1517, 2481, 295, 2592, 729, 1659, 385):
        t = 1403636580.0 + 0.05 * np.arange(n)
        q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
        out.append(np.column_stack([t, np.cumsum(rng.normal(0, 0.01, (n, 3)), 0), q]))
    return out


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from uav_airvision_amd import shard
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    allt_ref = _eight_trajectories()
    jobs = shard.broadcast_object([('SEQ_%d' % i, 0.0) for i in range(8)] if rank == 0 else None)
    mine = shard.partition(len(jobs), world, rank)
    local = {j: allt_ref[j] for j in mine}                  # what this rank's sweep produced for its own sequences
    allt = shard.gather_trajectories(local, len(jobs), world, rank)
    ok = sorted(allt) == list(range(8)) and all(a.dtype == np.float64 and a.shape == r.shape and np.array_equal(a, r) for a, r in zip([allt[j] for j in range(8)], allt_ref))
    q.put((rank, mine, ok, [allt[j].shape for j in range(8)]))
    dist.destroy_process_group()


def test_two_rank_gloo_gathers_eight_ragged_sequence_trajectories():
    """The end-of-run exchange of BASELINE configs[3] (8 sequences sharded over the ranks, trajectories gathered for the writer on
    rank 0): with the reference's own eight result files as the per-sequence arrays, every rank ends up holding all eight,
    bit-identical, whatever their lengths."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, ok0, sh0), (r1, m1, ok1, sh1) = res
    assert m0 == [0, 1, 2, 3] and m1 == [4, 5, 6, 7]
    assert ok0 and ok1 and sh0 == sh1
    assert len(set(s[0] for s in sh0)) == 8 and all(s[1] == 8 for s in sh0)        # eight different lengths


def _worker_report(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from uav_airvision_amd import shard
    from uav_airvision_amd.sweep import sweep_report
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    jobs = shard.broadcast_object([('MH_03_medium', float(o)) for o in range(5)] if rank == 0 else None)
    mine = shard.partition(len(jobs), world, rank)
    # what a rank's sweep produced for its own streams: per-stream ATE / RTE and the batch statistics
    report = {j: dict(sequence=jobs[j][0], offset=jobs[j][1], frames=100 + j, ate_rmse=0.01 * (j + 1), ate_mean=0.005, rte_rmse=0.002) for j in mine if j != 3}
    stats = {'seconds': 2.0 + rank, 'stream_frames': sum(100 + j for j in mine), 'steps': 104, 'frames_decoded': 110 + rank}
    elapsed = shard.max_over_ranks(2.5 + rank)
    per_rank = shard.gather_objects({'rank': rank, 'streams': [jobs[j] for j in mine], 'report': report, 'stats': stats})
    q.put((rank, sweep_report(jobs, per_rank, elapsed, world)))
    dist.destroy_process_group()


def test_two_rank_gloo_sweep_report_is_one_aggregate_line():
    """configs[3] / [4] are defined by "aggregate frames/sec + per-seq ATE": the per-rank reports and batch statistics are gathered
    (shard.gather_objects) and `sweep.sweep_report` builds ONE line -- stream-frames of all ranks over the slowest rank's time, every
    stream's ATE in job order (a stream too short for an ATE still has its row) -- identical on every rank; rank 0 prints it."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_report, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_r0, rep0), (_r1, rep1) = res
    assert rep0 == rep1
    assert rep0['n_gpus'] == 2 and rep0['streams'] == [['MH_03_medium', float(o)] for o in range(5)] and rep0['streams_per_rank'] == [3, 2]
    assert rep0['stream_frames'] == 100 + 101 + 102 + 103 + 104 and rep0['frames_decoded'] == 221
    assert abs(rep0['seconds'] - 3.5) < 1e-12 and abs(rep0['value'] - 510 / 3.5) < 1e-9          # the slower rank's clock
    assert [r['offset'] for r in rep0['report']] == [0.0, 1.0, 2.0, 3.0, 4.0]
    assert rep0['report'][3]['frames'] == 0 and 'ate_rmse' not in rep0['report'][3]
    assert abs(rep0['report'][4]['ate_rmse'] - 0.05) < 1e-12 and rep0['report'][4]['frames'] == 104


def test_bench_self_launches_its_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts one child per rank before anything touches
    a GPU, the ranks broadcast the run configuration, partition the streams, take the max-over-ranks time and the
    sum-over-ranks counters, and exactly ONE JSON line comes out (rank 0's).  --dry-run: gloo on CPU, no GPU work."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '4', '--warmup', '1', '--streams', '6'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d['dry_run'] is True and d['n_gpus'] == 2 and d['steps'] == 4 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['config']['streams_per_gpu'] == 6 and d['config']['streams_total'] == 12 and d['config']['seed'] == 1234
    # rank r sleeps (1 + r) ms per step: the reported time is the slower rank's
    assert d['ms_per_step'] >= 2.0
    assert abs(d['value'] - 12 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6


def test_bench_reports_a_failed_rank():
    """A rank that dies makes the self-launching parent exit non-zero (here: --gpus 2 under a WORLD_SIZE=3 mismatch is
    rejected by every rank before any GPU call)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and 'WORLD_SIZE=3' in p.stderr and not p.stdout.strip()
