"""The N > 1 path on CPU: 2 ranks over gloo.  The data path has no collective (independent
transforms / channels); what multi-GPU adds is the partition and the timing protocol
(barrier + max over ranks), which is what runs here."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simpledsp_amd.dist import shard_range, timed_steps


def test_shard_ranges_partition_the_batch():
    for total in (0, 1, 7, 65536, 2_097_152, 1_048_577):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(2_097_152, 3, 8) == (786_432, 1_048_576)  # BASELINE config 5: 262144 per GPU
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(1000, rank, world)
    done = {"units": 0}

    def step():  # rank 1 is slower: the reported time must be ITS time
        time.sleep(0.01 * (1 + 2 * rank))
        done["units"] += hi - lo

    wall = timed_steps(step, steps=5, warmup=2, sync=lambda: None, dist=dist)
    total = torch.tensor([done["units"]], dtype=torch.int64)
    dist.all_reduce(total)
    out.put((rank, wall, int(total.item()), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_timing_protocol():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, t0, s0), (r1, w1, t1, s1) = res
    assert s0 == (0, 500) and s1 == (500, 1000)
    assert t0 == t1 == 7 * 1000            # (2 warm-up + 5 timed) steps x all units, summed over ranks
    assert abs(w0 - w1) < 1e-9              # both ranks report the max
    assert 0.14 < w0 < 1.0                  # ~5 x 30 ms: the slow rank's time, not the fast rank's 50 ms
