"""Multi-GPU layout of the path: reads shard, the index is replicated, there is no data-path collective.

One process per GPU (torch.distributed; backend "nccl" = RCCL on the GPU box, "gloo" in CPU tests).
Rank r classifies the contiguous slice shard_bounds(n, world, r) of the batch with its own replica of the
index; results stay POD (cfr_result / cfr_match arrays).  The only communication is the optional final
gather of those POD arrays to rank 0 (<= 64 B/read, far below one xGMI link) and the max-over-ranks
reduction of the elapsed time that bench.py reports.

The classify function is injected (on the GPU box it is DeviceIndex.classify); this module holds no
compute and no fallback.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous, balanced slice [lo, hi) of n reads for `rank`; concatenating ranks restores input order."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return n * rank // world, n * (rank + 1) // world


def shard_reads(bases: np.ndarray, offsets: np.ndarray, world: int, rank: int):
    """Slice a flat read buffer (bases + n+1 offsets) for one rank; offsets are rebased to 0."""
    n = len(offsets) - 1
    lo, hi = shard_bounds(n, world, rank)
    o = offsets[lo:hi + 1]
    return bases[int(o[0]):int(o[-1])], (o - o[0]).astype(np.uint64)


def merge_results(parts):
    """parts: list over ranks of (results, matches) with rank-local match_begin.  Returns the arrays of the
    whole batch in input order, match_begin rebased into the concatenated match array."""
    res_all, mat_all, shift = [], [], 0
    for results, matches in parts:
        r = results.copy()
        r["match_begin"] += np.uint64(shift)
        res_all.append(r)
        mat_all.append(matches)
        shift += len(matches)
    return np.concatenate(res_all), np.concatenate(mat_all)


def classify_sharded(classify, bases1, offsets1, bases2=None, offsets2=None, dist=None, gather=True):
    """Run `classify(b1, o1, b2, o2) -> (results, matches)` on this rank's shard; optionally gather to rank 0.

    Returns (results, matches) for the WHOLE batch on rank 0 (None, None elsewhere) when gather=True,
    else this rank's own part."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    b1, o1 = shard_reads(bases1, offsets1, world, rank)
    b2 = o2 = None
    if bases2 is not None:
        b2, o2 = shard_reads(bases2, offsets2, world, rank)
    part = classify(b1, o1, b2, o2)
    if world == 1 or not gather:
        return part
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(part, gathered, dst=0)
    if rank != 0:
        return None, None
    return merge_results(gathered)


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """bench.py's timing rule: the step time of the job is the slowest rank's."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
