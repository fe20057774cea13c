"""Multi-GPU path on CPU: world_size-2 gloo run of the chunk sharding + result gather (no data-path collective)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import chunk_dp


def test_shard_chunks_partition_properties():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            parts = [list(chunk_dp.shard_chunks(n, r, world)) for r in range(world)]
            flat = [x for p in parts for x in p]
            assert flat == list(range(n))                                   # disjoint, complete, ordered
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        chunk_dp.shard_chunks(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = chunk_dp.shard_chunks(5, rank, world)
    local = [dict(chunk=i, text="chunk-%d" % i, rank=rank) for i in ids]     # stands for per-chunk segment lists
    allr = chunk_dp.gather_results(local, dist)
    # the timing protocol of bench.py: barrier, then MAX over ranks of the local time
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((allr, float(t.item())))
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    allr, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r["chunk"] for r in allr] == [0, 1, 2, 3, 4]
    assert [r["rank"] for r in allr] == [0, 0, 0, 1, 1]
    assert tmax == 2.0
