"""world_size-2 CPU test (gloo) of the multi-GPU plumbing: block partition of streams and the
job-level reductions bench.py uses.  The data path has no collective."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from crispy_amd.sharding import reduce_job_stats, shard_range


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 4096, 8193):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            for a, b in zip(parts, parts[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(10, rank, world)
    # each rank "processes" its own streams: per-stream result depends on the stream id only
    mine = np.arange(lo, hi) * 2.0
    dist.barrier()
    elapsed, frames = reduce_job_stats(0.5 + rank, (hi - lo) * 100)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine.tolist())
    q.put((rank, elapsed, frames, gathered))
    dist.destroy_process_group()


def test_two_rank_gloo_job_stats_and_assembly():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, elapsed, frames, gathered in res:
        assert elapsed == 1.5            # max over ranks
        assert frames == 1000            # all 10 streams x 100 frames
        assert sum(gathered, []) == [2.0 * i for i in range(10)]   # shards tile the stream ids in order
