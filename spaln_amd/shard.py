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
