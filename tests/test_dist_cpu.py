"""Multi-process (world_size 2, gloo, CPU) check of the N > 1 measurement path of bench.py:
independent replicas on disjoint scenes, barrier-bracketed timing, MAX over ranks, whole-job
aggregate.  The rasterizer path itself has no collective."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():  # rank 1 is 3x slower: the job time must be rank 1's
        calls.append(1)
        time.sleep(0.01 * (1 + 2 * rank))

    el = bench.timed_region(step, steps=5, warmup=2, dist=dist)
    seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seeds, torch.tensor([bench.rank_seed(1234, rank)]))
    out.put((rank, el, len(calls), [int(s) for s in seeds], list(bench.PER_RANK_SECONDS)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_timing_and_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, el0, n0, seeds0, pr0), (r1, el1, n1, seeds1, pr1) = res
    assert pr0 == pr1 and len(pr0) == 2 and max(pr0) == el0 and pr0[1] > 2 * pr0[0]   # every rank's own time is reported too
    assert n0 == n1 == 7                      # 2 warm-up + exactly 5 timed steps on every rank
    assert el0 == el1                         # MAX over ranks is what every rank reports
    assert el0 >= 5 * 0.03 * 0.9              # ... and it is the slow rank's time
    assert seeds0 == [1234, 1235] == seeds1   # disjoint scenes per rank
    v = bench.whole_job_views_per_s(16, 5, 2, el0)
    assert v == pytest.approx(16 * 5 * 2 / el0)
