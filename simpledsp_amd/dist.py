"""Batch sharding across the GPUs of one node: contiguous ranges, no data-path collective.

Every transform / channel is independent, so multi-GPU is pure partitioning:
rank g of G owns units [g*B/G, (g+1)*B/G).  torch.distributed is used only to line the ranks
up (barrier) and to take the max-over-ranks wall time; backend "nccl" (= RCCL) on GPUs, "gloo"
in the CPU tests."""
from __future__ import annotations

import os
import time
from typing import Callable, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `total` units for `rank` -- same split as the C ABI's
    sdsp_hip_fft_exec_sharded / sdsp_hip_iir_process_sharded."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return total * rank // world, total * (rank + 1) // world


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def timed_steps(step: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None], dist=None,
                device=None) -> float:
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + sync on both sides.
    Returns the MAX over ranks of the wall time of the K steps (seconds).  All ranks leave the opening
    barrier together; each stops its clock when its own K steps have drained (synchronize), then joins
    the closing barrier -- so the maximum is the whole job's time without charging the barrier's own
    latency (~2 ms for RCCL) to the steps."""
    import torch
    if dist is not None:
        # not the bracketing barrier: the FIRST collective of a process costs milliseconds (kernel load,
        # channel setup); taken here, before the warm-up, it leaves the bracketing barrier below at tens of
        # microseconds -- short enough that the device does not drop out of its steady clocks before step 1
        dist.barrier()
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    wall = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    sync()
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall
