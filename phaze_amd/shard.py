"""Multi-GPU sharding helpers (no compute): static stream -> rank partition and the max-over-ranks timing reduction.

Streams are independent processors, so the data path needs no collective; torch.distributed (RCCL on the
GPU box, gloo in the CPU tests) only carries the barrier and one all_reduce(MAX) of the timing.  When the streams of a job arrive on
(and must return to) one rank, `scatter_streams` / `gather_streams` are the only exchange there is: one block of whole streams per rank,
moved once each way (BASELINE north_star: "RCCL over xGMI only for the trivial channel scatter/gather").
"""
from typing import List, Sequence


def stream_partition(nstreams: int, world: int, rank: int) -> List[int]:
    """THE partition rule: stream s runs on rank s mod world.  Compute placement, scatter_streams / gather_streams and the Node host
    (phaze_amd/node/sharded.js) all use it; round 2's second, contiguous-block rule is gone."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return list(range(rank, nstreams, world))


def _host_side(dist, t):
    """gloo moves host tensors; RCCL moves device tensors.  With gloo (CPU tests, and the one GPU test that puts two ranks on one device) a device
    tensor makes the round trip through host memory around the call."""
    return dist.get_backend() == "gloo" and t.is_cuda


def reduce_max(values: Sequence[float], dist=None, device=None) -> List[float]:
    """Element-wise MAX over ranks (identity without a process group; WITH one the collective runs even at world size 1, so that a one-rank
    torchrun exercises the backend -- RCCL on the GPU box)."""
    if dist is None or not dist.is_initialized():
        return [float(v) for v in values]
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=None if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def aggregate_rate(units_per_rank: int, world: int, seconds_max: float) -> float:
    """Whole-job throughput: units all ranks processed / slowest rank's time."""
    return units_per_rank * world / seconds_max


def scatter_streams(x_all, nstreams: int, dist, src: int = 0):
    """Rank `src` holds x_all = [nstreams, ...]; every rank returns the streams stream_partition gives it, in ascending order (a new tensor on
    x_all's device / dtype).  Other ranks pass a tensor of the right trailing shape, dtype and device (contents ignored, no rows required)."""
    import torch
    if dist is None or not dist.is_initialized():
        return x_all[:nstreams].clone()
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = stream_partition(nstreams, world, rank)
    host = _host_side(dist, x_all)
    wire = torch.device("cpu") if host else x_all.device
    out = torch.empty((len(mine),) + tuple(x_all.shape[1:]), dtype=x_all.dtype, device=wire)
    pieces = None
    if rank == src:
        pieces = [x_all[r:nstreams:world].contiguous().to(wire) for r in range(world)]
    if nstreams % world == 0:
        dist.scatter(out, pieces, src=src)
    else:                                                    # unequal shares: point-to-point (scatter wants equal sizes)
        if rank == src:
            for r in range(world):
                if r == src:
                    out.copy_(pieces[r])
                elif pieces[r].numel():
                    dist.send(pieces[r], dst=r)
        elif out.numel():
            dist.recv(out, src=src)
    return out.to(x_all.device) if host else out


def gather_streams(y_mine, nstreams: int, dist, dst: int = 0):
    """Inverse of scatter_streams: rank `dst` returns [nstreams, ...] in stream order, the others None."""
    import torch
    if dist is None or not dist.is_initialized():
        return y_mine
    world, rank = dist.get_world_size(), dist.get_rank()
    home = y_mine.device
    if _host_side(dist, y_mine):
        y_mine = y_mine.cpu()
    pieces = None
    if nstreams % world == 0:
        pieces = [torch.empty_like(y_mine) for _ in range(world)] if rank == dst else None
        dist.gather(y_mine.contiguous(), pieces, dst=dst)
    elif rank == dst:
        pieces = []
        for r in range(world):
            n = len(stream_partition(nstreams, world, r))
            if r == dst:
                pieces.append(y_mine)
            else:
                buf = torch.empty((n,) + tuple(y_mine.shape[1:]), dtype=y_mine.dtype, device=y_mine.device)
                if buf.numel():
                    dist.recv(buf, src=r)
                pieces.append(buf)
    elif y_mine.numel():
        dist.send(y_mine.contiguous(), dst=dst)
    if rank != dst:
        return None
    out = torch.empty((nstreams,) + tuple(y_mine.shape[1:]), dtype=y_mine.dtype, device=y_mine.device)
    for r in range(world):
        out[r:nstreams:world] = pieces[r]
    return out.to(home)
