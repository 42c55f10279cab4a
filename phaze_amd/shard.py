"""Multi-GPU sharding helpers (no compute): static stream -> rank partition and the max-over-ranks timing reduction.

Streams are independent processors, so the data path needs no collective; torch.distributed (RCCL on the
GPU box, gloo in the CPU tests) only carries the barrier and one all_reduce(MAX) of the timing.
"""
from typing import List, Sequence


def stream_partition(nstreams: int, world: int, rank: int) -> List[int]:
    """Stream s runs on rank s mod world."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return list(range(rank, nstreams, world))


def reduce_max(values: Sequence[float], dist=None, device=None) -> List[float]:
    """Element-wise MAX over ranks (identity without a process group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def aggregate_rate(units_per_rank: int, world: int, seconds_max: float) -> float:
    """Whole-job throughput: units all ranks processed / slowest rank's time."""
    return units_per_rank * world / seconds_max
