"""Query sharding across the GPUs of one node.

Alignment problems are independent: rank r of N takes a
contiguous slice of the query list, aligns it on its own GPU, and results are
concatenated in query order.  No data-path collective; the only communication
is the final gather of fixed-size result records (and the timing reduction in
bench.py).  The reference's counterpart is its pthread master/worker queue
(src/spaln.cc:1389-1468), which hands whole queries to workers.
"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced slice of [0, n_items) owned by `rank` (first ranks take the remainder)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def balanced_shards(costs, world: int):
    """Cost-balanced shards: `costs[i]` = DP cells of query i (spdp_cells); longest-processing-time rule -- queries in
    order of falling cost, each to the rank with the least work so far (ties: the lower rank), every shard then in query
    order.  Windows of 2 kb cDNAs range from 6 to 16 kb, so equal COUNTS leave the slowest rank with 10-20 % more cells
    than the mean; equal cells is what bounds the step.  Returns `world` index lists; deterministic, the same on every
    rank (and the rule of spdp_group_* in spdp_group.cpp)."""
    if world <= 0:
        raise ValueError("bad world")
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    load = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += int(costs[i])
    return [sorted(s) for s in shards]


def scatter_in_order(n_items: int, shards, parts):
    """inverse of balanced_shards: parts[r][k] is the result of query shards[r][k]; returns the list in query order"""
    out = [None] * n_items
    for idx, res in zip(shards, parts):
        for i, x in zip(idx, res):
            out[i] = x
    return out


def gather_in_order(local_results, dist=None):
    """All ranks call this with the results of their slice; every rank gets the full list in
    query order.  `dist` is torch.distributed (None / uninitialised = single process)."""
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_results)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, list(local_results))
    out = []
    for p in parts:
        out.extend(p)
    return out
