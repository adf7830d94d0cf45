"""The N>1 path on CPU: two gloo ranks each build their own shard of synthetic reads, score their
chains (host job builder + oracle costs + host replay), and reduce the final counters."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_work(rank, n_reads=24):
    import ctypes as C

    sys.path.insert(0, ROOT)
    from oracle.loader import Oracle
    from rawalign_amd import synth
    from rawalign_amd._lib import AlignOpt, load_library
    from rawalign_amd.dtw import JOB_DTYPE
    from rawalign_amd.shard import rank_seed

    ref = synth.make_reference([60000], seed=11)
    n = len(ref.forward[0])
    pad = (n + 3) & ~3
    offs = {(0, 1): 0, (0, 0): pad}
    cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=n_reads, max_chunks=2), seed=rank_seed(5, rank))
    lib = load_library()
    opt = AlignOpt(1, 1, 0.10, 0.4, 20.0, 1)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    job_off = np.zeros(cb.n_chains + 1, np.uint64)
    nj = C.c_uint64()
    lib.rawdtw_batch_build_jobs(C.byref(opt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base),
                                p(cb.read_base), p(job_off), None, 0, C.byref(nj))
    jobs = np.zeros(nj.value, JOB_DTYPE)
    lib.rawdtw_batch_build_jobs(C.byref(opt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base),
                                p(cb.read_base), p(job_off), p(jobs), len(jobs), C.byref(nj))
    arena = np.zeros(2 * pad, np.float32)
    arena[:n] = ref.forward[0]
    arena[pad:pad + n] = ref.reverse[0]
    orc = Oracle()
    costs = orc.batch_costs(jobs, cb.events, arena, 1)
    score = np.zeros(cb.n_chains, np.float32)
    keep = np.zeros(cb.n_chains, np.uint8)
    lib.rawdtw_batch_replay(C.byref(opt), cb.n_reads, p(cb.chain_off), p(cb.anchor_off), p(cb.anchors), p(job_off),
                            p(costs), p(score), p(keep))
    cells = sum(orc.banded_cells(int(j["n"]), int(j["m"]), int(j["band_radius"])) for j in jobs)
    read_of_chain = np.repeat(np.arange(cb.n_reads), np.diff(cb.chain_off.astype(np.int64)))
    mapped = len(np.unique(read_of_chain[keep.astype(bool)]))
    return [cb.n_reads, cb.n_chains, len(jobs), cells, mapped], float(score[keep.astype(bool)].sum())


def _worker(rank, world, port, q):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from rawalign_amd.shard import reduce_counters

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counters, _ = _shard_work(rank)
    total, tmax = reduce_counters(dist, counters, elapsed_s=1.0 + rank)
    q.put((rank, counters, total, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_counters_reduce():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    per_rank = [g[1] for g in got]
    want = [sum(x) for x in zip(*per_rank)]
    for g in got:
        assert g[2] == want      # every rank sees the same global counters
        assert g[3] == 2.0       # max over ranks of the step time
    # shards differ (distinct seeds) and each has work
    assert per_rank[0] != per_rank[1] and all(c[2] > 0 for c in per_rank)
    assert want[4] > 0           # some reads mapped


def test_shard_is_world_size_independent():
    """A rank's shard depends only on its rank (weak scaling): rank 1 alone == rank 1 of two."""
    a, sa = _shard_work(1, n_reads=8)
    b, sb = _shard_work(1, n_reads=8)
    assert a == b and sa == sb
    from rawalign_amd.shard import shard_reads

    assert [shard_reads(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (before anything touches a GPU), reduces
    the counters over gloo and relays ONE line with n_gpus = 2.  --dry-run: launcher and reduction plumbing only."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, env=env, timeout=600)
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0 and two.returncode == 0, two.stderr[-2000:]
    l1 = json.loads(one.stdout.strip().splitlines()[-1])
    l2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["steps"] == 3 and l2["scaling"] == "weak"
    # weak scaling: rank 0's shard is the same in both runs, rank 1 adds its own
    assert l2["totals_over_timed_steps"]["reads"] == 2 * l1["totals_over_timed_steps"]["reads"]
    assert l2["totals_over_timed_steps"]["dtw_jobs"] > l1["totals_over_timed_steps"]["dtw_jobs"]


def _su_records(n_reads, n_seq, seed):
    rng = np.random.default_rng(seed)
    mapped = rng.random(n_reads) < 0.8
    probs = np.array([0.5, 0.3, 0.2, 0.0][:n_seq], float)
    probs = probs / probs.sum()
    ref_id = rng.choice(n_seq, size=n_reads, p=probs)
    frag = rng.integers(200, 9000, n_reads)
    return mapped, ref_id, frag


def _su_worker(rank, world, port, q):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from rawalign_amd.mapping import SequenceUntil
    from rawalign_amd.shard import sequence_until_round, shard_reads

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    su = SequenceUntil(n_seq=3, tmin_reads=40, ttest_freq=10, tn_samples=3, t_threshold=1.5)
    stops = []
    for mb in range(3):  # three mini-batches; the state carries over (rmap.cpp:1018-1027 initialises it once)
        n = 101 + 7 * mb
        mapped, ref_id, frag = _su_records(n, 3, 100 + mb)
        lo, hi = shard_reads(n, rank, world)
        stops.append(sequence_until_round(dist, su, mapped[lo:hi], ref_id[lo:hi], frag[lo:hi], n))
        if stops[-1]:
            break
    q.put((rank, stops, su.c_estimations.tolist(), su.nreads))
    dist.barrier()
    dist.destroy_process_group()


def test_sequence_until_across_ranks_equals_one_process():
    """f-3: the stop point of --sequence-until (rmap.cpp:918-944) with a mini-batch's reads sharded over two ranks is the one
    a single process finds, and every rank holds the same abundance counters."""
    from rawalign_amd.mapping import SequenceUntil
    from rawalign_amd.shard import sequence_until_round

    su = SequenceUntil(n_seq=3, tmin_reads=40, ttest_freq=10, tn_samples=3, t_threshold=1.5)
    want = []
    for mb in range(3):
        n = 101 + 7 * mb
        mapped, ref_id, frag = _su_records(n, 3, 100 + mb)
        want.append(sequence_until_round(None, su, mapped, ref_id, frag, n))
        if want[-1]:
            break
    assert want[-1] > 0  # the rule fires inside these mini-batches
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_su_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, stops, cest, nreads in got:
        assert stops == want and cest == su.c_estimations.tolist() and nreads == su.nreads
