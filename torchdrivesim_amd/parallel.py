"""
Multi-GPU execution of the hot path: scenes are independent (every tensor has a leading scene-batch axis and no operation
mixes scenes -- simulator.py:444-511 of the reference), so the batch is cut into contiguous shards, one process per GPU,
and NO collective is on the data path.  torch.distributed (RCCL on the GPU box, gloo in the CPU tests) is only used for the
barrier around a timed region and the max-over-ranks of the elapsed time / sum of processed units.
"""
from typing import Tuple

import torch


def scene_shard(total_scenes: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[start, stop) of the scenes owned by `rank`: contiguous, sizes differ by at most one, union = range(total)."""
    assert 0 <= rank < world_size and total_scenes >= 0
    base, extra = divmod(total_scenes, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_simulator(sim, rank: int, world_size: int):
    """A new Simulator holding only this rank's scenes (pure batch-axis selection, simulator.py:480-511 of the reference).  The shard's device maps
    are found again by content in the process-wide cache (`_ops.map_cache`: one map per DISTINCT mesh), so sharding on the device that already
    holds the maps creates none; a rank that shards on its own device builds one map per distinct mesh of ITS scenes."""
    start, stop = scene_shard(sim.batch_size, rank, world_size)
    return sim.select_batch_elements(list(range(start, stop)), in_place=False)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def barrier(device=None, group=None, device_ids=None):
    """dist.barrier() (if a process group exists; over `group` when given, e.g. an RCCL group beside a default gloo group) followed by a
    device synchronisation when `device` is a GPU."""
    d = _dist()
    if d is not None:
        if group is not None:
            d.barrier(group=group, device_ids=device_ids)
        else:
            d.barrier()
    if device is not None and torch.device(device).type == 'cuda':
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device='cpu') -> float:
    d = _dist()
    if d is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    d.all_reduce(t, op=d.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device='cpu') -> float:
    d = _dist()
    if d is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value: float, device='cpu'):
    """list of every rank's value, in rank order (one element without a process group)"""
    d = _dist()
    if d is None:
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(d.get_world_size())]
    d.all_gather(out, mine)
    return [float(t.item()) for t in out]


def gather_objects_over_ranks(obj):
    """list of every rank's (picklable) object, in rank order (one element without a process group) -- reporting only, never data"""
    d = _dist()
    if d is None:
        return [obj]
    out = [None] * d.get_world_size()
    d.all_gather_object(out, obj)
    return out


def aggregate_throughput(units_this_rank: float, elapsed_this_rank: float, device='cpu') -> float:
    """whole-job rate = units processed by all ranks / slowest rank's time"""
    return sum_over_ranks(units_this_rank, device) / max_over_ranks(elapsed_this_rank, device)
