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
