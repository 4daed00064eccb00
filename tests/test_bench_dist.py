"""CPU test of bench.py's N > 1 path: 2 ranks over gloo run the timing contract (barrier + sync on both
sides, MAX over ranks, whole-job aggregate).  The hot path itself does not shard for one big tree
("replicas only"), so there is no data-path collective to test."""
import os
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sleep = 0.05 if rank == 0 else 0.25   # uneven ranks: the MAX must win

    def work():
        time.sleep(sleep)
    dt = bench.timed_region(work, dist, lambda: None, reduce_device="cpu")
    value = bench.whole_job_rate(1000, 4, world, dt)
    q.put((rank, dt, value))
    dist.destroy_process_group()


def test_two_rank_timing_contract():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, v0), (r1, dt1, v1) = out
    assert dt0 == dt1 and 0.25 <= dt0 < 1.0          # both ranks report the slowest rank's time
    assert v0 == v1 == 2 * 1000 * 4 / dt0            # whole-job aggregate, not per-rank


def test_single_rank_path():
    sys.path.insert(0, ROOT)
    import bench
    dt = bench.timed_region(lambda: time.sleep(0.01), None, lambda: None)
    assert 0.01 <= dt < 0.5
    assert bench.whole_job_rate(10, 2, 1, 2.0) == 10.0
