"""Stream sharding across ranks: the unit of work is one stereo stream
(sequence x start offset, reference: run.bat:4-12); frames inside a stream are sequential and stay
on one GPU, so there is NO data-path collective.  torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used for exactly three things: broadcasting the run configuration from rank 0, the
barrier + max-over-ranks timing of the bench contract, and gathering per-stream results at the end.
"""
import io
import pickle

import numpy as np
import torch
import torch.distributed as dist


def partition(n_items, world, rank):
    """Contiguous block partition: the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(int(n_items), int(world))
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def _device():
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')


def broadcast_object(obj, src=0):
    """Config / calibration broadcast (< 2 KB): pickled bytes through a uint8 tensor."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return obj
    dev = _device()
    if dist.get_rank() == src:
        data = pickle.dumps(obj)
        n = torch.tensor([len(data)], dtype=torch.int64, device=dev)
    else:
        data = b''
        n = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8, device=dev)
    if dist.get_rank() == src:
        buf.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    dist.broadcast(buf, src)
    return pickle.loads(buf.cpu().numpy().tobytes())


def gather_objects(obj):
    """Small picklable per-rank results (reports, counters) to EVERY rank, in rank order: world broadcasts of a few hundred bytes at
    the end of a run."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    return [broadcast_object(obj if dist.get_rank() == r else None, src=r) for r in range(dist.get_world_size())]


def max_over_ranks(value):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values):
    a = np.asarray(values, dtype=np.float64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return a
    t = torch.from_numpy(a.copy()).to(_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_trajectories(local, n_total, world, rank):
    """End-of-run gather of per-stream trajectories (float64[frames, 8] each, reference line format
    t px py pz qx qy qz qw, msckf.py:152-158) to every rank.  `local` = {stream_index: array}."""
    if not dist.is_initialized() or world == 1:
        return dict(local)
    dev = _device()
    out = {}
    for owner in range(world):
        for s in partition(n_total, world, owner):
            if owner == rank:
                arr = np.ascontiguousarray(local[s], dtype=np.float64)
                shape = torch.tensor(list(arr.shape) + [0] * (2 - arr.ndim), dtype=torch.int64, device=dev)
            else:
                arr = None
                shape = torch.zeros(2, dtype=torch.int64, device=dev)
            dist.broadcast(shape, owner)
            buf = torch.zeros(tuple(int(v) for v in shape.tolist()), dtype=torch.float64, device=dev)
            if owner == rank:
                buf.copy_(torch.from_numpy(arr))
            dist.broadcast(buf, owner)
            out[s] = buf.cpu().numpy()
    return out
