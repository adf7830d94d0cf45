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


def sequence_until_round(dist, su, mapped, ref_id, fragment_length, n_reads_total: int):
    """Sequence-until across GPUs (src/rmap.cpp:918-944): the reference walks a mini-batch's reads IN READ ORDER, adds every
    mapped read's fragment length to its genome's counter and tests the abundance estimates every `ttest_freq` mapped
    reads; the read at which the test passes (su_stop = k + 1) decides which PAF lines are printed (rmap.cpp:960).

    With the mini-batch's reads dealt to ranks in contiguous blocks (shard_reads), every rank contributes its block's
    per-read records -- mapped flag, genome, fragment length: 9 bytes per read -- to one all-gather; every rank then
    replays the walk on the identical, read-ordered records, so all ranks hold the same counters and find the same stop
    point as one process would.  (Summing per-rank counters instead would test at different read counts.)

    `su` is rawalign_amd.mapping.SequenceUntil (the same state object on every rank); `mapped`, `ref_id`,
    `fragment_length` describe this rank's block.  Returns su.stop (0: keep going; k + 1: stop after read k)."""
    import numpy as np
    import torch

    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_reads(n_reads_total, rank, world)
    assert len(mapped) == hi - lo == len(ref_id) == len(fragment_length)
    per = (n_reads_total + world - 1) // world
    rec = torch.zeros((per, 3), dtype=torch.int64)
    rec[:hi - lo, 0] = torch.as_tensor(np.asarray(mapped, np.int64))
    rec[:hi - lo, 1] = torch.as_tensor(np.asarray(ref_id, np.int64))
    rec[:hi - lo, 2] = torch.as_tensor(np.asarray(fragment_length, np.int64))
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        mine = rec.to(dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        allrec = torch.cat([p.cpu() for p in parts])[:n_reads_total]
    else:
        allrec = rec[:n_reads_total]
    allrec = allrec.numpy()
    for k in range(n_reads_total):
        if allrec[k, 0] and su.add_mapped_read(int(allrec[k, 1]), int(allrec[k, 2]), k):
            break
    return su.stop
