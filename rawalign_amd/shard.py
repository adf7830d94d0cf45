"""Multi-GPU layout of the hot path: reads shard across ranks, the reference signal is replicated,
and nothing is exchanged on the data path.  The only collective is the reduction of the final
counters (reads, chains, DTW jobs, DP cells, mapped reads) and of the step time -- SURVEY.md 8(e).
`dist` is torch.distributed (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests)."""
from __future__ import annotations


def rank_seed(base_seed: int, rank: int) -> int:
    """Seed of rank `rank`'s synthetic reads: distinct per rank, independent of the world size,
    so that a rank's shard is the same whether it runs alone or next to others (weak scaling)."""
    return int(base_seed) + 7919 * (int(rank) + 1)


def shard_reads(n_reads: int, rank: int, world: int):
    """Index range of the reads rank `rank` owns when a mini-batch of n_reads is dealt in
    contiguous blocks (PAF lines are merged back in read order on the host)."""
    per = (n_reads + world - 1) // world
    lo = min(n_reads, rank * per)
    return lo, min(n_reads, lo + per)


def reduce_counters(dist, counters, elapsed_s: float, device=None):
    """Sum the integer counters and take the max of the elapsed time over all ranks."""
    import torch

    c = torch.tensor([int(x) for x in counters], dtype=torch.int64, device=device)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(x) for x in c.cpu()], float(t.cpu()[0])
