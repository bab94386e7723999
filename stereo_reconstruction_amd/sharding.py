"""Sharding of independent stereo pairs over the GPUs of one node.

Every pair is independent (no state survives a BlockSearch object, BlockSearch.cpp:6-13), so
the N>1 path is: one process per GPU, no collective on the data path.  The only communication
is the barrier / max-over-ranks that brackets a timed region and the gather of small per-rank
summaries.  Works with any torch.distributed backend (RCCL on the GPUs, gloo in the CPU tests).

Work items are whole pairs (`lpt_assign`) or, for smoothFactor 1, ROW BANDS of pairs
(`band_items`): a row of the map only depends on the image rows under its window
(BlockSearch.cpp:46-66, :120-158), so the map rows [y0, y1) are rows of a search on the
sub-images [y0 - half, y1 + half) -- what ws_search_host's own banded path relies on, results
identical.  15 trainingH pairs as whole items balance to 1.13 x / 1.15 x the mean at 4 / 8
ranks; as bands of >= 256 rows to <= 1.03 x.  smoothFactor != 1 (a raster dependency down the
whole image) and varBlock keep whole pairs.
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


def band_items(shapes, max_d, world, block_size, min_rows=256):
    """Work items (pair, y0, y1) -- map rows [y0, y1) of pair `pair` -- for `world` ranks, and their assignment.

    shapes: (width, height) per pair.  The pairs' rows are laid end to end (a row of pair i weighs width_i * max_d)
    and cut into `world` runs of equal weight; a cut inside a pair makes two bands, and both must keep at least
    `min_rows` rows -- a cut that would not is moved to the nearest place that does (or to the pair's edge), and the
    following ranks share what is left evenly, so a moved cut does not push work down the line.  Several orders of
    the pairs are tried (rotations of the given order, largest / smallest first) and the one with the smallest
    maximum load -- a band's cost is what its search really computes: (rows + 2 * half) * width * max_d -- is kept.
    Returns (items, shards): shards[r] lists the item indices of rank r.  Every map row of every pair belongs to
    exactly one item; deterministic."""
    if world < 1:
        raise ValueError("world must be >= 1")
    half = (block_size - 1) // 2
    n = len(shapes)

    def cost(i, y0, y1):
        return float(min(shapes[i][1], y1 + half) - max(0, y0 - half)) * shapes[i][0] * max_d

    def cut(order):
        per_rank = [[] for _ in range(world)]
        remaining = float(sum(shapes[i][0] * shapes[i][1] for i in order)) * max_d
        pos = 0           # index into order
        y = 0             # first row of order[pos] not handed out yet
        for r in range(world):
            target = remaining / (world - r)
            got = 0.0
            last = r == world - 1
            while pos < n:
                i = order[pos]
                w, h = shapes[i]
                rest = float(h - y) * w * max_d
                if last or got + rest <= target:
                    if h > y:
                        per_rank[r].append((i, y, h))
                    got += rest
                    pos, y = pos + 1, 0
                    continue
                # the ideal cut lies inside this pair
                yc = y + int(round((target - got) / (w * max_d)))
                lo, hi = y + min_rows, h - min_rows      # both bands keep min_rows rows
                if lo > hi:                              # the rest of this pair cannot be cut: all of it or none of it
                    yc = h if (target - got) * 2 >= rest else y
                else:
                    if yc - y < min_rows:
                        yc = y if (yc - y) * 2 < min_rows else lo
                    elif h - yc < min_rows:
                        yc = h if (h - yc) * 2 < min_rows else hi
                if yc > y:
                    per_rank[r].append((i, y, yc))
                    got += float(yc - y) * w * max_d
                if yc >= h:
                    pos, y = pos + 1, 0
                else:
                    y = yc
                break
            remaining -= got
        return per_rank

    orders = [list(range(k, n)) + list(range(k)) for k in range(max(n, 1))]
    by_cost = sorted(range(n), key=lambda i: (-shapes[i][0] * shapes[i][1], i))
    orders += [by_cost, by_cost[::-1]]
    best, best_load = None, None
    for order in orders:
        per_rank = cut(order)
        load = max(sum(cost(*it) for it in items) for items in per_rank) if per_rank else 0.0
        if best is None or load < best_load:
            best, best_load = per_rank, load
    items, shards = [], []
    for r in range(world):
        shards.append(list(range(len(items), len(items) + len(best[r]))))
        items.extend(best[r])
    return items, shards


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
