"""world_size-2 gloo test (CPU) of the N>1 path: static stream partition, independence of shards, max-over-ranks timing.

The GPU kernels cannot run here; each rank pushes ITS streams through the CPU oracle (test infrastructure) to prove that
the union of per-rank results equals the single-process result bit for bit, i.e. that sharding needs no exchange step.
"""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nstreams, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib
    import signals as S
    from phaze_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.stream_partition(nstreams, world, rank)
    fft, hop, T, nch = 1024, 256, 12, 2
    for s in mine:
        x = np.stack([S.make_signal("tonal", c, T * hop, stream=s) for c in range(nch)])
        y = oracle_lib.Oracle(fft, hop, nch).process_planar(x, np.full(T, 1.5, np.float32))
        np.save(os.path.join(tmpdir, f"s{s}.npy"), y)
    dist.barrier()
    t = shard.reduce_max([1.0 + rank, 10.0 - rank], dist)
    assert t == [float(world), 10.0], t
    assert shard.aggregate_rate(100, world, t[0]) == 100 * world / world
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import signals as S
    from phaze_amd import shard
    nstreams, world = 5, 2
    parts = [shard.stream_partition(nstreams, world, r) for r in range(world)]
    assert sorted(sum(parts, [])) == list(range(nstreams)) and parts[0] == [0, 2, 4]
    mp.spawn(_worker, args=(world, 29613, nstreams, str(tmp_path)), nprocs=world, join=True)
    fft, hop, T, nch = 1024, 256, 12, 2
    for s in range(nstreams):
        x = np.stack([S.make_signal("tonal", c, T * hop, stream=s) for c in range(nch)])
        ref = oracle_lib.Oracle(fft, hop, nch).process_planar(x, np.full(T, 1.5, np.float32))
        assert np.array_equal(np.load(tmp_path / f"s{s}.npy"), ref)


def _sg_worker(rank, world, port, nstreams, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib
    import signals as S
    from phaze_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fft, hop, T, nch = 1024, 256, 10, 2
    if rank == 0:
        x_all = torch.from_numpy(np.stack([np.stack([S.make_signal("noise", c, T * hop, stream=s) for c in range(nch)]) for s in range(nstreams)]))
    else:
        x_all = torch.empty((0, nch, T * hop), dtype=torch.float32)
    x = shard.scatter_streams(x_all, nstreams, dist)
    assert x.shape[0] == len(shard.stream_partition(nstreams, world, rank))
    y = torch.stack([torch.from_numpy(oracle_lib.Oracle(fft, hop, nch).process_planar(x[i].numpy(), np.full(T, 0.8, np.float32)).astype(np.float32))
                     for i in range(x.shape[0])]) if x.shape[0] else torch.empty((0, nch, T * hop), dtype=torch.float32)
    y_all = shard.gather_streams(y, nstreams, dist)
    if rank == 0:
        np.save(os.path.join(tmpdir, "gathered.npy"), y_all.numpy())
    else:
        assert y_all is None
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_process_gather_equals_single_process(tmp_path):
    """The only exchange of a multi-GPU job (streams out from rank 0, results back) around rank-local processing == one process doing it all."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import signals as S
    for nstreams, port in ((4, 29633), (5, 29634)):          # equal blocks (scatter/gather collectives) and ragged blocks (send/recv)
        mp.spawn(_sg_worker, args=(2, port, nstreams, str(tmp_path)), nprocs=2, join=True)
        got = np.load(tmp_path / "gathered.npy")
        fft, hop, T, nch = 1024, 256, 10, 2
        for s in range(nstreams):
            x = np.stack([S.make_signal("noise", c, T * hop, stream=s) for c in range(nch)])
            ref = oracle_lib.Oracle(fft, hop, nch).process_planar(x, np.full(T, 0.8, np.float32)).astype(np.float32)
            assert np.array_equal(got[s], ref)
