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
    assert bench.gather_floats(2.5, 1) == [2.5]
    # total algorithmic bytes per frame + the colour convert, per config
    want = {"c1": 1685700 + 1536000, "c2": 11138784 + 10368000, "c3": 44472360 + 41472000}
    for name, total in want.items():
        assert sum(bench.CONFIGS[name]["bytes"].values()) == total


def _run_bench(*argv, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=120, env=e)
    assert res.returncode == 0, res.stdout + res.stderr
    return [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]


def test_launcher_spawns_one_rank_per_gpu_dry_run():
    """`python bench.py --gpus N` (the shape of the driver's N=1 command) must not silently measure one GPU: with no
    launcher around it, it starts the N ranks itself.  --dry-run prints each rank's device and camera streams instead
    of touching a GPU."""
    lines = sorted(_run_bench("--gpus", "4", "--streams", "3", "--ring", "8", "--dry-run"), key=lambda d: d["rank"])
    assert [d["rank"] for d in lines] == [0, 1, 2, 3] and {d["world"] for d in lines} == {4}
    assert [d["device"] for d in lines] == [0, 1, 2, 3]                              # one GPU per rank
    assert [d["streams"] for d in lines] == [[0, 4, 8], [1, 5, 9], [2, 6, 10], [3, 7, 11]]   # stream s -> rank s % N
    assert all(len(d["phases"]) == 3 and all(0 <= ph < 8 for ph in d["phases"]) for d in lines)
    # a rehearsal on a box with fewer GPUs than ranks: ranks share devices round-robin
    lines = sorted(_run_bench("--gpus", "4", "--streams", "32", "--dry-run", "--dry-run-devices", "1"), key=lambda d: d["rank"])
    assert [d["device"] for d in lines] == [0, 0, 0, 0] and all(len(d["streams"]) == 32 for d in lines)
    # N = 1 needs no children; the defaults are BASELINE configs[1]'s
    (one,) = _run_bench("--dry-run")
    cores = one.pop("host_cores")
    import bench
    want = bench.default_streams(bench.CONFIGS["c2"], 1, cores)          # a host core per stream's thread, at most 32
    assert cores == bench_host_cores() and one.pop("host_threads_total") == want
    assert one == {"rank": 0, "world": 1, "device": 0, "streams": list(range(want)), "phases": [(3 * s) % 64 for s in range(want)]}


def bench_host_cores():
    import bench
    return bench.host_cores()


def test_streams_per_gpu_follow_the_host_cores_when_ranks_share_a_host():
    """Every camera stream has a polling host thread: 8 ranks x 32 streams would be 256 threads on one host, while 8 streams already
    saturate a GPU (profiles/r03_stream_sweep.txt), and with more threads than cores one short run in three loses 10-15 %
    (profiles/r03_streams_short_blocks.txt): a core per stream, never more than the configuration's count."""
    import bench
    c2, c3 = bench.CONFIGS["c2"], bench.CONFIGS["c3"]
    assert bench.default_streams(c2, 1, 16) == 16 and bench.default_streams(c3, 1, 16) == 16
    assert bench.default_streams(c2, 1, 64) == 32 and bench.default_streams(c3, 1, 64) == 16 and bench.default_streams(c2, 1, 2) == 4
    assert bench.default_streams(c2, 8, 128) == 16 and bench.default_streams(c2, 8, 64) == 8
    assert bench.default_streams(c2, 8, 16) == 4 and bench.default_streams(c2, 2, 8) == 4      # never below a hardware queue each
    assert bench.default_streams(c2, 2, 256) == 32 and bench.default_streams(c3, 8, 256) == 16  # never above the configuration's
    # the launcher at --gpus 8 (dry run): the reduced count reaches every rank and the line says how many host threads that is
    cores = bench.host_cores()
    want = bench.default_streams(c2, 8, cores)
    lines = sorted(_run_bench("--gpus", "8", "--dry-run"), key=lambda d: d["rank"])
    assert len(lines) == 8 and all(len(d["streams"]) == want for d in lines) and want < 32
    assert all(d["host_cores"] == cores and d["host_threads_total"] == 8 * want for d in lines)
    assert sorted(s for d in lines for s in d["streams"]) == list(range(8 * want))


def test_a_rank_that_dies_ends_the_job_within_seconds():
    """One dead rank (busy GPU, out of memory) must not leave its siblings waiting in the process group until a timeout: the
    launcher terminates them and returns the failed rank's exit code.  Rank 1 exits 3 at start-up; rank 0 is by then waiting in
    init_process_group for it (gloo, before any GPU call, so this runs on CPU)."""
    import subprocess
    import time
    e = dict(os.environ, VSA_BENCH_TEST_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    t0 = time.monotonic()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=100, env=e)
    took = time.monotonic() - t0
    assert res.returncode == 3, (res.returncode, res.stderr[-2000:])
    assert "rank 1 ended with exit code 3" in res.stderr
    assert took < 60, took                                   # seconds (python + torch start-up), not the process group's timeout
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]      # no JSON line from a failed job


def test_a_rank_that_never_arrives_trips_the_process_group_timeout():
    """Under an external launcher (torchrun) there is no sibling watch of ours: the process group's own timeout is the backstop."""
    import subprocess
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pg-timeout", "3"],
                         capture_output=True, text=True, timeout=100, env=e)
    assert res.returncode != 0 and ("imeout" in res.stderr or "timed out" in res.stderr.lower()), res.stderr[-2000:]


def test_torchrun_style_environment_is_respected_dry_run():
    """under torch.distributed.run the ranks already exist: no children, RANK / LOCAL_RANK / WORLD_SIZE from the environment"""
    (d,) = _run_bench("--gpus", "2", "--streams", "2", "--dry-run", env=dict(RANK="1", LOCAL_RANK="1", WORLD_SIZE="2"))
    assert d["rank"] == 1 and d["device"] == 1 and d["streams"] == [1, 3]
    import subprocess
    e = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True, env=e)
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr                         # a mismatch is an error, not a guess
