"""Sharding of independent stereo pairs over the GPUs of one node.

Every pair is independent (no state survives a BlockSearch object, BlockSearch.cpp:6-13), so
the N>1 path is: one process per GPU, each takes whole pairs, no collective on the data path.
The only communication is the barrier / max-over-ranks that brackets a timed region and the
gather of small per-rank summaries.  Works with any torch.distributed backend (RCCL on the
GPUs, gloo in the CPU tests).
"""
import time


def lpt_assign(costs, world):
    """Longest-processing-time-first assignment of pairs to ranks.

    costs: work estimate per pair (H*W*D).  Returns a list of `world` lists of pair indices;
    every index appears exactly once; deterministic (ties by index)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += costs[i]
    for s in shards:
        s.sort()
    return shards


def timed_region(fn, sync, dist=None, device=None):
    """Run fn() between two barriers; return the MAX over ranks of the elapsed seconds.

    sync(): drains the local device (torch.cuda.synchronize on a GPU, a no-op on CPU)."""
    import torch
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    fn()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def gather_objects(obj, dist=None, world=1):
    """All ranks' small summaries on every rank (list indexed by rank)."""
    if dist is None:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out
