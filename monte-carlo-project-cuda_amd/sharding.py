"""Multi-GPU host logic: one process per GPU, paths partitioned by contiguous global path id.

The path shards: rank g simulates global ids [lo, lo + n_local) of one job, and because the Philox
subsequence of a path is its GLOBAL id, the union of the shards draws exactly the numbers the
single-GPU job draws.  The only exchange is one all-reduce of (sum, sumsq, n) — three doubles —
per pricing call, over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo"
in the CPU tests).  The reference has no multi-GPU code at all.
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard of [0, n_total) for `rank`: sizes differ by at most one, remainder to the
    first ranks.  Returns (first global path id, number of local paths)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} for world size {world}")
    if n_total < 0:
        raise ValueError("n_total must be >= 0")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, base + (1 if rank < rem else 0)


def allreduce_stats(sum_: float, sumsq: float, n: int, device="cpu") -> Tuple[float, float, int]:
    """One all-reduce of the shard statistics.  No-op without an initialised process group."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(sum_), float(sumsq), int(n)
    t = torch.tensor([sum_, sumsq, float(n)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    s, s2, nn = t.tolist()
    return float(s), float(s2), int(round(nn))


def allreduce_vector(values, device="cpu"):
    """One SUM all-reduce of a short list of doubles (the statistics record plus whatever diagnostics ride along:
    the nested-MC work counters).  No-op without an initialised process group."""
    import torch
    import torch.distributed as dist

    vals = [float(v) for v in values]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return vals
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def price_sharded(local_stats: Callable[[int, int], Tuple[float, float]], n_total: int, world: int, rank: int,
                  finalize: Callable[[float, float, int], object], device="cpu"):
    """Prices one job across `world` ranks.

    local_stats(lo, n_local) -> (sum, sumsq) of the undiscounted payoffs of this rank's shard (on the
    GPU box: mcamd_price_paths through the C ABI); finalize(sum, sumsq, n) -> result on the reduced
    statistics (mcamd_finalize).  Every rank returns the same finalized result."""
    lo, n_local = shard_range(n_total, world, rank)
    s, s2 = local_stats(lo, n_local) if n_local else (0.0, 0.0)
    s, s2, n = allreduce_stats(s, s2, n_local, device)
    return finalize(s, s2, n)
