"""The N>1 path of bench.py on CPU: camera streams are sharded over ranks with no data-path
collective; the process group (gloo here, as in bench.py) only carries the barrier and the
max-over-ranks timing reduction."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_shard_streams_partition():
    import bench
    for world in (1, 2, 4, 8):
        total = 4 * world
        shards = [bench.shard_streams(total, world, r) for r in range(world)]
        assert sorted(s for sh in shards for s in sh) == list(range(total))     # disjoint cover
        assert {len(sh) for sh in shards} == {4}                                  # weak scaling: equal share


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench.barrier(world)
    mine = bench.shard_streams(6, world, rank)
    slow = bench.reduce_max(1.0 + rank, world)            # the job's time is the slowest rank's
    t = torch.tensor([float(len(mine))])
    dist.all_reduce(t)                                     # bookkeeping only: all streams accounted for
    q.put((rank, mine, slow, float(t.item())))
    bench.barrier(world)
    dist.destroy_process_group()


def test_two_rank_timing_reduction_gloo():
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3, 5]
    assert all(r[2] == 2.0 for r in res)                   # max over ranks
    assert all(r[3] == 6.0 for r in res)


def test_single_rank_helpers_need_no_process_group():
    import bench
    assert bench.reduce_max(3.5, 1) == 3.5
    bench.barrier(1)
    assert bench.BYTES_FRAME == 11138784 + 10368000       # SURVEY 8d table, configs[1]
