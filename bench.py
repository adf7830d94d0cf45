#!/usr/bin/env python3
"""bench.py -- DTW hot path of RawAlign on MI355X: DTW GCUPS + mapped reads/s, and % of the HBM roofline.

Workload = BASELINE.json configs[1]: E. coli K-12 (4.6 Mb), r9.4-like reads, sparse border constraint, banded=0.10
fill.  One "step" = one pass of the hot path over one batch = one chunk round of `--reads` reads per GPU: every DTW
job of every candidate chain (src/rmap.cpp:509-530), the align_chain fold and the per-read accept/cut loop.

Figures of one run (all in the one JSON line):

  value = value_resident       every step is a FRESH batch: rawdtw_batch_submit (the scan of the anchor list, the DTW launch,
                               fold and select: all enqueued by one call) and, when the context comes round again,
                               rawdtw_batch_fetch_destroy -- `--inflight` contexts deep, as the reference's kt_pipeline
                               keeps mini-batches in flight (rmap.cpp:1033).  Inputs (event arenas, anchor lists) are
                               resident in HBM when the timed region starts; nothing is cached between steps.
  value_pcie (pipeline_pcie)   the same loop with the host-side hand-over inside the step: the round's NEW events
                               (rmap.cpp:554-567 is append-only) and the anchor lists cross PCIe from pinned memory --
                               what a caller that produces its anchors on the host (rmap.cpp:396-507) gets end to end.
  kernel_replay                the launches of already submitted resident batches only (what round 1 reported as `value`).
  modes                        BASELINE.json configs[2] on the same GPU: global + full-fill scoring, and the traceback of
                               every read's best chain (--dtw-output-cigar) with the device time of its kernels alone.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...          # launches N ranks itself (torch.distributed.run) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A K-step region is a few milliseconds and starts and ends with an idle pipeline, so one timed bracket (barrier +
synchronize on both sides) holds `repeats` consecutive K-step regions of the same loop with no drain between them;
ms_per_step = bracket / (repeats * K); three brackets (at least 400 ms in total), the MEDIAN one is reported (max over
ranks per bracket) -- the figure does not depend on --steps.  Multi-GPU: reads shard across ranks, reference replicated, no data-path collective; RCCL only
reduces the final counters and the times.  Weak scaling.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
WORKLOAD = "ecoli_k12_4.6Mb_r9.4_sparse_banded0.10"
SEED = 20231005 + 2  # seed = 20231005 + config id


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=160, help="steps of one region; a timed bracket holds `repeats` regions back to back (see repeat_region)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--reads", type=int, default=16384, help="reads per GPU in one batch (chunk round)")
    ap.add_argument("--genome", type=int, default=4_600_000, help="reference length in bases (E. coli K-12)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="contexts = mini-batches in flight per GPU (the reference's kt_pipeline keeps 2, rmap.cpp:1033)")
    ap.add_argument("--stagger-us", type=float, default=100.0,
                    help="pause between the first submissions of a region's contexts (inside the timed region).  A region starts from an "
                         "idle, synchronised device; four contexts submitted in the same instant run their launches in step with each other "
                         "-- a state that lasts the whole region and costs 11 %% (one region in three) -- while a caller's contexts are out "
                         "of phase by themselves; 60-120 us apart the regions all run in the fast mode, at no cost to the region's time")
    ap.add_argument("--min-region-ms", type=float, default=400.0,
                    help="the three timed brackets of a loop hold this much in total; the median bracket counts")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="CPU baseline budget per thread count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hit-prob", type=float, default=None, help="workload sensitivity: anchor density of the true chains")
    ap.add_argument("--decoy-gap-median", type=float, default=None, help="workload sensitivity: decoy chain gap (events)")
    ap.add_argument("--decoys-per-read", type=float, default=None)
    ap.add_argument("--rounds", type=int, default=4, help="chunk rounds of the cross-round cache block (0: skip it)")
    ap.add_argument("--modes-reads", type=int, default=8192, help="reads of the configs[2] block (0: skip it)")
    ap.add_argument("--mapper-reads", type=int, default=16384, help="reads of the mapper block: mapped reads/s through the chunk-round loop (0: skip it)")
    ap.add_argument("--mapper-threads", type=int, default=0, help="host threads of the mapper block (0: the cores this process may use)")
    ap.add_argument("--trace-fresh", type=int, default=0,
                    help="profiling runs: only the fresh-batch loop (inputs resident), this many steps, no JSON line")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / reduction plumbing only: no device work, synthetic counters (CPU tests)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks before anything touches a GPU, relay rank 0's line."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in run.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if run.returncode != 0 or line is None:
        sys.stderr.write(run.stdout[-4000:] + "\n" + run.stderr[-8000:] + "\n")
        sys.exit(run.returncode or 1)
    print(line)
    sys.exit(0)


def cpu_baseline(jobs, events, ref_arena, cells, seconds):
    """The same job list on the host cores, through the REFERENCE's own compiled dtw.cpp when oracle/_ref is present
    ("reference"), else through the oracle's C restatement ("port").  Thread counts: 1, every core the process may use
    (affinity mask capped by the cgroup's CPU quota), twice that; each on a bounded sample of whole jobs (cells pro-rated
    by job count)."""
    from oracle.loader import Oracle, RefDTW, build_oracle

    if RefDTW.available():
        impl, kind = RefDTW(), "reference"
    else:
        build_oracle(march_native=True)
        impl, kind = Oracle(), "port"
    n = len(jobs)
    nproc = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = nproc
    quota = cgroup_cpu_quota()  # the box gives one GPU's job a share of the host's cores (a CFS quota, not an affinity mask)
    usable = max(1, min(affinity, int(quota + 0.999))) if quota else affinity
    t0 = time.perf_counter()
    impl.batch_costs(jobs[:100000], events, ref_arena, 1)
    per_job_1 = (time.perf_counter() - t0) / min(n, 100000)
    runs = []
    out_full = None
    # thread counts: one, every core this process may use, and twice that (SMT / oversubscription check)
    counts = {1, usable} | ({2 * usable} if 2 * usable <= affinity else set())
    if quota is None and usable > 16:
        counts.add(16)  # no quota visible: a one-GPU box's share is 16 cores, so that count is tried as well
    counts = sorted(counts)
    for threads in counts:
        # one pool of threads works through `reps` passes over a prefix of the job list: about `seconds` of wall time
        take = int(min(n, max(50000, seconds * 0.7 / max(per_job_1, 1e-9)))) if threads == 1 else n
        t0 = time.perf_counter()
        impl.batch_costs(jobs[:take], events, ref_arena, threads, 1)  # calibration pass (also warms the pool's pages)
        est = time.perf_counter() - t0
        # the box's cores are shared with other tenants: three shorter measurements, the best one counts
        tries = 1 if threads == 1 else 3
        reps = int(max(1, min(400, round(seconds / tries / max(est, 1e-6)))))
        best_dt, out = None, None
        for _ in range(tries):
            t0 = time.perf_counter()
            out = impl.batch_costs(jobs[:take], events, ref_arena, threads, reps)
            dt = time.perf_counter() - t0
            best_dt = dt if best_dt is None else min(best_dt, dt)
        dt = best_dt
        runs.append({"threads": threads, "value": cells * (take / n) * reps / dt / 1e9, "jobs": take, "passes": reps, "seconds": round(dt, 3),
                     "tries": tries})
        if out_full is None or take > len(out_full):
            out_full = out
    best = max(runs, key=lambda r: r["value"])
    return {
        "value": best["value"], "unit": "GCUPS", "cores": best["threads"], "kind": kind,
        "nproc": nproc, "affinity_cores": affinity, "cgroup_cpu_quota": quota, "usable_cores": usable,
        "sample": f"first {best['jobs']} of {n} DTW jobs of one batch of the same workload ({best['jobs'] / n * 100:.0f}% of its "
                  f"cells, pro-rated by job count), {best['passes']} passes inside one pool of {best['threads']} threads pulling job "
                  f"ranges from a shared counter (as kt_for deals reads: kthread.c:54-72), {best['seconds']} s wall (best of "
                  f"{best['tries']} such measurements: the host's cores are shared); the best of the thread counts tried is reported",
        "by_threads": runs,
    }, out_full


def cgroup_cpu_quota():
    """CPUs' worth of CFS quota of this process's cgroup (v2 cpu.max, v1 cpu.cfs_quota_us), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


# ---------------------------------------------------------------------------------------------------------------------
class Pinned:
    """numpy arrays in page-locked host memory (rawdtw_host_alloc)."""

    def __init__(self, lib):
        self.lib = lib
        self.ptrs = []

    def copy(self, arr):
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        st = self.lib.rawdtw_host_alloc(max(arr.nbytes, 8), C.byref(p))
        assert st == 0 and p.value
        self.ptrs.append(p)
        buf = (C.c_char * max(arr.nbytes, 8)).from_address(p.value)
        out = np.frombuffer(buf, dtype=arr.dtype, count=arr.size).reshape(arr.shape)
        out[...] = arr
        return out

    def empty(self, n, dtype):
        return self.copy(np.zeros(n, dtype))

    def free(self):
        for p in self.ptrs:
            self.lib.rawdtw_host_free(p)
        self.ptrs = []


def vp(a):
    return C.c_void_p(a.ctypes.data)


CARRY_DTYPE = np.dtype([("prev_src", "<u8"), ("parts", "<u4"), ("flags", "<u4"), ("start_t", "<u4"), ("start_q", "<u4")])  # rawdtw_carry_t


def rounds_block(engine, lib, copt, cb, info, n_rounds, local_rank, pin):
    """SURVEY.md 8(f-4) at the bench batch's scale: `n_rounds` chunk rounds of one batch of reads (synth.make_rounds), each
    submitted from scratch as the reference does (rmap.cpp:516-517: rawdtw_batch_submit, the whole anchor lists) and carried
    (rawdtw_batch_submit_carry: per chain the host says how many leading parts did not change and where their costs lie
    in the batch before -- rawdtw_round_match_chains, timed on its own -- only the NEW anchors and one junction a chain are
    handed over, the device's scan, planning and DTW launches run on that short list, and one gather launch lays every
    chain's costs out in full for the fold).  One context, rounds in sequence
    (a round needs the one before); wall time per round from submit to fetched scores, (a) with the hand-over arrays in
    pinned host memory -- what a mapper's round is: `scratch_ms` / `carried_ms` -- and (b) with them resident in HBM, as in the
    `value` loop (`*_resident_ms`)."""
    from rawalign_amd import synth
    import torch

    rounds = synth.make_rounds(cb, info, n_rounds)
    ident = np.arange(cb.n_reads, dtype=np.uint64)
    R = []
    for k, r in enumerate(rounds):
        d = {"chain_off": pin.copy(np.ascontiguousarray(r.chain_off, np.uint64)), "anchor_off": pin.copy(np.ascontiguousarray(r.anchor_off, np.uint64)),
             "anchors": pin.copy(np.ascontiguousarray(r.anchors)), "ref_base": pin.copy(np.ascontiguousarray(r.ref_base, np.uint64)),
             "read_base": pin.copy(np.ascontiguousarray(r.read_base, np.uint32))}
        if k:
            p = R[k - 1]
            carry = np.zeros(cb.n_chains, CARRY_DTYPE)
            new_off = np.zeros(cb.n_chains + 1, np.uint64)
            new_anchors = np.zeros(len(r.anchors) + 1, r.anchors.dtype)
            t0 = time.perf_counter()
            engine._check(lib.rawdtw_round_match_chains(cb.n_reads, vp(d["chain_off"]), vp(d["anchor_off"]), vp(d["anchors"]), vp(d["ref_base"]), vp(d["read_base"]),
                                                        vp(ident), vp(p["chain_off"]), vp(p["anchor_off"]), vp(p["anchors"]), vp(p["ref_base"]), vp(p["read_base"]),
                                                        vp(carry), vp(new_off), vp(new_anchors)))
            d["match_ms"] = (time.perf_counter() - t0) * 1e3
            d["carry"], d["new_off"] = pin.copy(carry), pin.copy(new_off)
            d["new_anchors"] = pin.copy(new_anchors[:max(int(new_off[-1]), 1)])
            d["n_new"] = int(new_off[-1])
            d["t_new"] = torch.from_numpy(d["new_anchors"].view(np.uint8).copy()).cuda(local_rank)
        d["t"] = [torch.from_numpy(d[x].view(np.uint8).copy()).cuda(local_rank) for x in ("anchors", "ref_base", "read_base")]
        R.append(d)
    torch.cuda.synchronize()
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    out = {"rounds": n_rounds, "per_round": [{"parts": int(max(int(d["anchor_off"][-1]) - cb.n_chains, 0)), "anchors": int(d["anchor_off"][-1])} for d in R]}
    same = True
    want = [None] * n_rounds
    for resident in (0, 1):
        engine.set_option("resident_arrays", resident)
        sfx = "_resident" if resident else ""
        for mode in ("warm", "warm_carried", "scratch", "carried"):  # (both forms once untimed: their workspaces come out of the context's pool afterwards)
            prev = None
            for k, d in enumerate(R):
                h = C.c_void_p()
                score, keep = np.zeros(cb.n_chains, np.float32), np.zeros(cb.n_chains, np.uint8)
                A = (P(d["t"][0]), P(d["t"][1]), P(d["t"][2])) if resident else (vp(d["anchors"]), vp(d["ref_base"]), vp(d["read_base"]))
                engine.sync()
                t0 = time.perf_counter()
                if mode.endswith("carried") and prev is not None:
                    engine._check(lib.rawdtw_batch_submit_carry(engine._ctx, C.byref(copt), cb.n_reads, vp(d["chain_off"]), vp(d["anchor_off"]), vp(d["anchors"]),
                                                                vp(d["new_off"]), P(d["t_new"]) if resident else vp(d["new_anchors"]), A[1], A[2], prev,
                                                                vp(d["carry"]), C.byref(h)))
                else:
                    engine._check(lib.rawdtw_batch_submit(engine._ctx, C.byref(copt), cb.n_reads, vp(d["chain_off"]), vp(d["anchor_off"]), A[0], A[1], A[2], C.byref(h)))
                engine._check(lib.rawdtw_batch_fetch(engine._ctx, h, vp(score), vp(keep), None))
                dt = time.perf_counter() - t0
                rec = out["per_round"][k]
                if mode == "scratch":
                    rec["scratch%s_ms" % sfx] = round(dt * 1e3, 4)
                    if want[k] is None:
                        want[k] = (score.copy(), keep.copy())
                    same = same and np.array_equal(want[k][0].view(np.uint32), score.view(np.uint32)) and np.array_equal(want[k][1], keep)
                if mode == "warm_carried":
                    if prev is not None:
                        lib.rawdtw_batch_destroy(prev)
                    prev = h
                    continue
                if mode == "carried":
                    sc, ru = C.c_uint64(), C.c_uint64()
                    engine._check(lib.rawdtw_batch_round_stats(engine._ctx, h, C.byref(sc), C.byref(ru)))
                    same = same and np.array_equal(want[k][0].view(np.uint32), score.view(np.uint32)) and np.array_equal(want[k][1], keep)
                    rec.update({"parts_scored": int(sc.value), "parts_reused": int(ru.value), "carried%s_ms" % sfx: round(dt * 1e3, 4),
                                "anchors_handed_over": int(d.get("n_new", rec["anchors"])), "match_ms_host_1_thread": round(d.get("match_ms", 0.0), 3)})
                    if prev is not None:
                        lib.rawdtw_batch_destroy(prev)
                    prev = h
                else:
                    lib.rawdtw_batch_destroy(h)
            if prev is not None:
                lib.rawdtw_batch_destroy(prev)
    engine.set_option("resident_arrays", 1)
    pr = out["per_round"]
    tot = lambda key, lo=0: round(sum(r[key] for r in pr[lo:]), 4)  # noqa: E731
    out.update({"parts_total": sum(r["parts"] for r in pr), "parts_scored": sum(r["parts_scored"] for r in pr),
                "jobs_reused": sum(r["parts_reused"] for r in pr),
                "scratch_ms_total": tot("scratch_ms"), "carried_ms_total": tot("carried_ms"),
                "scratch_resident_ms_total": tot("scratch_resident_ms"), "carried_resident_ms_total": tot("carried_resident_ms"),
                "carried_rounds": {"scratch_ms": tot("scratch_ms", 1), "carried_ms": tot("carried_ms", 1), "scratch_resident_ms": tot("scratch_resident_ms", 1),
                                   "carried_resident_ms": tot("carried_resident_ms", 1),
                                   "anchor_list_bytes_scratch": 8 * sum(r["anchors"] for r in pr[1:]),
                                   "anchor_list_bytes_carried": 8 * sum(r["anchors_handed_over"] for r in pr[1:]),
                                   "note": "rounds 2.. only (round 1 has no predecessor and is the same submission in both modes)"},
                "scores_identical_to_scratch": bool(same),
                "note": "`scratch_ms` / `carried_ms`: the hand-over arrays come from pinned host memory, as a mapper's round hands them over "
                        "(the anchors are made on the host every round: rmap.cpp:396-507); `*_resident_ms`: the same with them resident in HBM (no PCIe "
                        "in either form: what is left is the kernels and the submission's small copies -- round 3's block measured only this).  A carried round sends its new anchors and a junction a chain "
                        "(anchors_handed_over), plans and scores only the parts they bring, and gathers the rest's costs out of the batch before "
                        "(k_gather: one stretch a chain); the host's matching of the "
                        "chains (match_ms_host_1_thread, rawdtw_round_match_chains on one thread; the mapper does it per read on "
                        "its pool) is outside these times"})
    return out


def mapper_block(lib, ref, local_rank, n_reads, threads, cpu_seconds, with_cpu):
    """The metric's second half: MAPPED READS per second through the chunk-round mapping loop of the library
    (rawdtw_mapper_*, rawalign_amd/csrc/rawdtw_mapper.cpp: src/rmap.cpp:667-822 turned into rounds) -- per round and read the
    chunk's events and seed hits go in (event detection and seeding stay in RawAlign: imitated by synth.make_seed_chunks,
    outside the timed calls), the mapper re-seeds, sorts, chains (rmap.cpp:344-507) on `threads` host threads, scores every
    chain of every read on the device in one submission per read group, and finishes the round (primary chains, MAPQ, stop
    rule).  Timed: the rawdtw_mapper_round calls.  Variants: one read group / two (two contexts: one group's host phase beside
    the other's batch), costs carried from round to round or every round from scratch, the reference's stop rule and every read
    through all of its chunks (>= 4 rounds: where carried costs matter).  PAF lines must be identical across all of them.
    `cpu_baseline`: the SAME flow with the DTW block on the host cores through the reference's own dtw.cpp
    (oracle/_ref, plugged in through rawdtw_mapper_set_scorer) on a bounded sample of the reads."""
    import hashlib

    import rawalign_amd as ra
    from rawalign_amd import mapper, synth
    from rawalign_amd.mapping import StopOpt

    sc = synth.make_seed_chunks(ref, n_reads, seed=SEED + 17)
    opt = ra.MapOpt()
    names = [f"seq{s}" for s in range(ref.n_seq)]
    lens = [len(x) for x in ref.forward]
    slot = int(sc["n_ev"].max()) + 8
    first, nch = sc["chunk_first"], sc["n_chunks"]
    ev_off, hit_off = sc["ev_off"].astype(np.int64), sc["hit_off"].astype(np.int64)

    pins = Pinned(lib)
    ev_pin = pins.empty(int(np.diff(ev_off).max()) * n_reads + 1, np.float32)  # (a round's events: at most one chunk a read)

    def run(cm, reads, label, warm=True, measured=2):
        # the same reads twice untimed on the same mapper (its pinned buffers, the contexts' workspaces and the allocator's arenas reach their
        # sizes: the second pass over a fresh mapper still read 1.5 x the third), then released; then `measured` timed passes, of which the
        # faster counts (the box's host cores are shared with other boxes: a pass now and then reads half as fast) -- for every flow alike, the
        # cpu_baseline included; the passes' lines must be the same lines
        for _ in range(2 if warm is True else 0):
            w = run(cm, reads, label, warm=False)
            for i in w.pop("ids"):
                cm.release_read(int(i))
        if warm is True:
            best = None
            for _ in range(max(measured, 1)):
                r = run(cm, reads, label, warm="timed")
                for i in r.pop("ids"):
                    cm.release_read(int(i))
                assert best is None or best["paf_sha1"] == r["paf_sha1"]
                if best is None or r["seconds_in_rounds"] < best["seconds_in_rounds"]:
                    best = r
            best["timed_passes"] = measured
            return best
        ids = np.array([cm.add_read("read_%d" % r, int(sc["qlen"][r]), int(nch[r])) for r in reads], np.uint32)
        tm0, st0 = cm.timing(), cm.stats()
        done = np.zeros(len(reads), np.int64)
        active = np.ones(len(reads), bool)
        t_rounds, n_rounds, read_rounds = 0.0, 0, 0
        per_round = []
        while active.any():
            sel = np.nonzero(active)[0]
            ci = first[reads[sel]] + done[sel]                      # this round's chunk of every active read
            ecnt, hcnt = ev_off[ci + 1] - ev_off[ci], hit_off[ci + 1] - hit_off[ci]
            eo = np.concatenate([[0], np.cumsum(ecnt)]).astype(np.uint64)
            ho = np.concatenate([[0], np.cumsum(hcnt)]).astype(np.uint64)
            eidx = np.repeat(ev_off[ci], ecnt) + (np.arange(int(eo[-1])) - np.repeat(eo[:-1].astype(np.int64), ecnt))
            hidx = np.repeat(hit_off[ci], hcnt) + (np.arange(int(ho[-1])) - np.repeat(ho[:-1].astype(np.int64), hcnt))
            if len(eidx):  # the round's events in page-locked memory, as a host that allocates its event buffers with rawdtw_host_alloc has them
                ev = ev_pin[:len(eidx)]
                np.take(sc["events"], eidx, out=ev)
            else:
                ev = np.zeros(1, np.float32)
            hits = np.ascontiguousarray(sc["hits"][hidx]) if len(hidx) else np.zeros(1, sc["hits"].dtype)
            rid = np.ascontiguousarray(ids[sel])
            t0 = time.perf_counter()
            cm.round_arrays(rid, eo, ev, ho, hits)
            dt = time.perf_counter() - t0
            t_rounds += dt; n_rounds += 1; read_rounds += len(sel)
            per_round.append({"reads": int(len(sel)), "ms": round(dt * 1e3, 3)})
            done[sel] += 1
            for k in sel:  # (the mapper's own stop rule decides)
                fin, _ = cm.state(int(ids[k]))
                if fin or done[k] >= nch[reads[k]]:
                    active[k] = False
        assert cm.finish() == 0
        if not warm:
            return {"ids": ids}
        lines = [cm.paf(int(i)) for i in ids]
        mapped = sum(1 for ln in lines if ln.split("\t")[4] in "+-")
        h = hashlib.sha1("\n".join(lines).encode()).hexdigest()
        tm = {k: v - tm0[k] for k, v in cm.timing().items()}
        rounds, scored, reused = (a - b for a, b in zip(cm.stats(), st0))
        return {"ids": ids, "label": label, "reads": int(len(reads)), "mapped_reads": mapped, "rounds": n_rounds, "read_rounds": read_rounds,
                "seconds_in_rounds": round(t_rounds, 4), "reads_per_s": len(reads) / t_rounds, "mapped_reads_per_s": mapped / t_rounds,
                "read_rounds_per_s": read_rounds / t_rounds, "parts_scored": scored, "parts_reused": reused,
                "ms_per_round": {k: round(v / max(n_rounds, 1), 3) for k, v in tm.items() if k.endswith("_ms")},
                "h2d_bytes_per_round": {k: int(v / max(n_rounds, 1)) for k, v in tm.items() if k.endswith("_bytes")},
                "per_round": per_round, "paf_sha1": h}

    all_reads = np.arange(n_reads)
    never = StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6)
    out = {"workload": WORKLOAD + ", synthetic seed hits (hit probability 0.2 per event, 25 false hits a chunk), chunks of 520 events",
           "reads": n_reads, "host_threads": threads, "runs": [], "all_chunks": [],
           "events_are": "handed to rawdtw_mapper_round in page-locked memory (rawdtw_host_alloc): with the chaining on the device and one read group they go to the device as they are"}

    def device_run(stop, carry, groups, label, thr=threads, dev_chain=False):
        eng = ra.Engine(local_rank)
        eng.upload_reference(ref.forward, ref.reverse)
        cm = mapper.CMapper(eng, opt, stop, names, lens, slot_events=slot, max_reads=n_reads, carry=carry, threads=thr, groups=groups, device_chain=dev_chain)
        r = run(cm, all_reads, label)
        cm.close()
        eng.close()
        return r
    for groups in (1, 2):  # the anchor sort and the chaining DP on the device too (rawdtw_chain_round): the chains reach the DTW in device memory
        out["runs"].append(device_run(StopOpt(), 0, groups, "stop rule of the reference, chaining on the device, groups=%d" % groups, dev_chain=True))
    out["runs"].append(device_run(StopOpt(), 0, 1, "stop rule of the reference, chaining on the device, groups=1, ONE host thread", thr=1, dev_chain=True))
    for carry, groups in ((1, 2), (1, 1), (0, 2), (0, 1)):
        out["runs"].append(device_run(StopOpt(), carry, groups, "stop rule of the reference, carry=%d groups=%d" % (carry, groups)))
    out["runs"].append(device_run(StopOpt(), 1, 2, "stop rule of the reference, carry=1 groups=2, ONE host thread", thr=1))
    out["all_chunks"].append(device_run(never, 0, 1, "every read through all of its chunks, chaining on the device groups=1", dev_chain=True))
    for carry, groups in ((1, 2), (0, 2), (1, 1), (0, 1)):
        out["all_chunks"].append(device_run(never, carry, groups, "every read through all of its chunks, carry=%d groups=%d" % (carry, groups)))
    out["paf_identical_across_runs"] = len({r["paf_sha1"] for r in out["runs"]}) == 1 and len({r["paf_sha1"] for r in out["all_chunks"]}) == 1
    best = max(out["runs"], key=lambda r: r["reads_per_s"])
    out["reads_per_s"], out["mapped_reads_per_s"], out["best_run"] = best["reads_per_s"], best["mapped_reads_per_s"], best["label"]
    ac = {r["label"].split(", ")[1]: r for r in out["all_chunks"]}
    out["all_chunks_summary"] = {k: {"seconds_in_rounds": r["seconds_in_rounds"], "read_rounds_per_s": r["read_rounds_per_s"],
                                     "anchor_bytes_per_round": r["h2d_bytes_per_round"]["anchor_bytes"],
                                     "fetch_wait_ms_per_round": r["ms_per_round"]["fetch_wait_ms"]} for k, r in ac.items()}
    if with_cpu:
        from oracle.loader import RefDTW
        if RefDTW.available():
            rl = RefDTW().lib

            class ScorerCtx(C.Structure):
                _fields_ = [("fwd", C.c_void_p), ("rev", C.c_void_p), ("border_constraint", C.c_int), ("fill_method", C.c_int),
                            ("band_radius_frac", C.c_float), ("match_bonus", C.c_float), ("min_score", C.c_float), ("fused_score", C.c_int),
                            ("threads", C.c_int), ("dtw_calls", C.c_uint64)]
            fwd = (C.c_void_p * ref.n_seq)(*[x.ctypes.data for x in ref.forward])
            rev = (C.c_void_p * ref.n_seq)(*[x.ctypes.data for x in ref.reverse])
            sctx = ScorerCtx(C.cast(fwd, C.c_void_p), C.cast(rev, C.c_void_p), opt.dtw_border_constraint, opt.dtw_fill_method, opt.dtw_band_radius_frac,
                             opt.dtw_match_bonus, opt.dtw_min_score, int(opt.fused_score), threads, 0)
            fn = C.cast(rl.ref_scorer, C.c_void_p)
            # a bounded sample of the same reads: sized from a first small run
            n_s = min(n_reads, 512)
            base = None
            for _ in range(3):
                cm = mapper.CMapper(None, opt, StopOpt(), names, lens, slot_events=slot, max_reads=n_s, carry=False, threads=threads)
                cm.set_scorer_c(fn, C.cast(C.pointer(sctx), C.c_void_p))
                base = run(cm, all_reads[:n_s], "stop rule of the reference, DTW on the host cores (reference dtw.cpp)")
                cm.close()
                if base["seconds_in_rounds"] >= 0.25 * cpu_seconds or n_s >= n_reads:
                    break
                n_s = int(min(n_reads, max(n_s * 2, n_s * cpu_seconds / max(base["seconds_in_rounds"], 1e-3))))
            dev_same = next(r for r in out["runs"] if r["label"].endswith("carry=0 groups=1"))
            base.pop("per_round")
            out["cpu_baseline"] = {"value": base["reads_per_s"], "unit": "reads/s", "cores": threads, "kind": "reference",
                                   "sample": "the first %d of the %d reads through the same mapper (same host threads, same chaining) with the DTW block "
                                             "on the host cores: align_chain's score-only form around the reference's own dtw.cpp (oracle/_ref: "
                                             "ref_scorer, one task per read as kt_for deals them), %.2f s in its rounds" % (base["reads"], n_reads, base["seconds_in_rounds"]),
                                   "mapped_reads_per_s": base["mapped_reads_per_s"], "ms_per_round": base["ms_per_round"], "dtw_calls": int(sctx.dtw_calls),
                                   "note": "submit_ms holds the scorer's time here (the DTW block runs inside the submission)"}
            # the sample's lines must be the device flow's lines
            out["cpu_baseline"]["paf_prefix_identical_to_device_flow"] = None
            eng = ra.Engine(local_rank)
            eng.upload_reference(ref.forward, ref.reverse)
            cm = mapper.CMapper(eng, opt, StopOpt(), names, lens, slot_events=slot, max_reads=base["reads"], carry=True, threads=threads, groups=2)
            dev_s = run(cm, all_reads[:base["reads"]], "device, same sample")
            cm.close(); eng.close()
            out["cpu_baseline"]["paf_identical_to_device_flow_on_the_sample"] = dev_s["paf_sha1"] == base["paf_sha1"]
            out["cpu_baseline"]["device_reads_per_s_on_the_sample"] = dev_s["reads_per_s"]
            del dev_same
    for r in out["runs"] + out["all_chunks"]:
        if len(r["per_round"]) > 8:
            r["per_round"] = r["per_round"][:8] + [{"more": len(r["per_round"]) - 8}]
    pins.free()
    return out


YEAST = [230218, 813184, 316620, 1531933, 576874, 270161, 1090940, 562643, 439888, 745751, 666816, 1078177, 924431,
         784333, 1091291, 948066, 85779]  # S. cerevisiae S288C chromosome lengths (12.1 Mb): BASELINE.json configs[2]


def modes_block(local_rank, reads):
    """BASELINE.json configs[2] on this GPU, one context: (a) global border constraint + full fill, score only
    (DTW_global, dtw.cpp:37-66, one job per chain: rmap.cpp:192-237); (b) --dtw-output-cigar: the best chain of every
    read through DTW_global_tb (dtw.cpp:595-667, rmap.cpp:715-717) with the packed 2-bit direction buffer in HBM.  For
    (b) the device time of the fill and walk kernels alone (HIP events inside rawdtw_traceback_batch) next to the wall
    time of the whole call (plan, fill, walk, paths D2H, copy into the caller's arrays)."""
    import rawalign_amd as ra
    from rawalign_amd import synth
    from rawalign_amd._lib import AlignOpt

    eng = ra.Engine(local_rank)
    lib = eng.lib
    ref = synth.make_reference(YEAST, seed=20231005 + 3)
    eng.upload_reference(ref.forward, ref.reverse)
    offs = {(s_, st): eng.reference_offset(s_, st) for s_ in range(ref.n_seq) for st in (0, 1)}
    cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=reads, max_chunks=6), seed=77)
    eng.upload_events(cb.events)
    out = {"workload": "scerevisiae_12.1Mb_r9.4_global_full_cigar", "reads": reads}
    # (a) scoring
    batch = ra.Batch(eng, ra.MapOpt(dtw_border_constraint=0, dtw_fill_method=0), cb)
    info = batch.info()
    batch.run_reps(1, timed=False)
    steps = 5
    t0 = time.perf_counter()
    launches = batch.run_reps(steps, timed=True)
    dt = (time.perf_counter() - t0) / steps
    kms = sum(ms for k, _, ms in launches if k in (3,))  # the full-matrix fill launches
    out["global_full_score"] = {"jobs": info["n_jobs"], "cells": info["cells"], "ms_per_step": dt * 1e3, "gcups": info["cells"] / dt / 1e9,
                                "fill_kernels_ms": round(kms, 4), "fill_kernels_gcups": info["cells"] / max(kms, 1e-9) / 1e6,
                                "algorithmic_bytes": info["algorithmic_bytes"],
                                "hbm_frac": info["algorithmic_bytes"] / max(kms, 1e-9) / 1e6 / HBM_PEAK_GBS,
                                "note": "score only reads 4(n+m)+36 bytes a job and writes nothing else: the kernel is bound by its "
                                        "VALU recurrence (3 instructions a cell), not by HBM"}
    batch.close()
    # (b) traceback of the best (first) chain of every read, global + full (rmap.cpp:220-221)
    copt = AlignOpt(0, 0, 0.10, 0.4, 20.0, 1)
    firsts = [int(cb.chain_off[r]) for r in range(cb.n_reads) if cb.chain_off[r + 1] > cb.chain_off[r]]
    jobs = np.zeros(len(firsts), ra.JOB_DTYPE)
    one = np.zeros(1, ra.JOB_DTYPE)
    for k, c in enumerate(firsts):
        a = cb.anchors[int(cb.anchor_off[c]):int(cb.anchor_off[c + 1])]
        lib.rawdtw_chain_build_jobs(C.byref(copt), a.ctypes.data_as(C.c_void_p), len(a), int(cb.ref_base[c]), int(cb.read_base[c]), 1,
                                    one.ctypes.data_as(C.c_void_p))
        jobs[k] = one[0]
    n, m = jobs["n"].astype(np.int64), jobs["m"].astype(np.int64)
    cells = int((n * m).sum())
    poff = np.concatenate([[0], np.cumsum(n + m - 1)]).astype(np.uint64)
    cost = np.zeros(len(jobs), np.float32); plen = np.zeros(len(jobs), np.uint32)
    pi = np.zeros(int(poff[-1]), np.uint32); pj = np.zeros(int(poff[-1]), np.uint32); pd = np.zeros(int(poff[-1]), np.float32)

    tpin = Pinned(lib)   # (the steps form lands in page-locked arrays: no copy on the host)
    pstep = tpin.empty(int(poff[-1]) + 1, np.uint8)
    pd_pin = tpin.empty(int(poff[-1]) + 1, np.float32)
    ev_pin = tpin.copy(cb.events)   # (the reads' events from page-locked memory too: their upload runs beside the host's planning)

    def tb():  # the paths as steps + distances: what the library's mapper consumes (rawdtw_mapper_finish writes aln:s: from them)
        eng._check(lib.rawdtw_traceback_batch_steps(eng._ctx, vp(jobs), len(jobs), vp(ev_pin), len(ev_pin), vp(cost), vp(poff), vp(plen),
                                                    vp(pstep), vp(pd_pin)))

    def tb_ij():  # ... and expanded into (i, j, distance) arrays on the host: dtw_result as the reference returns it
        eng._check(lib.rawdtw_traceback_batch(eng._ctx, vp(jobs), len(jobs), vp(cb.events), len(cb.events), vp(cost), vp(poff), vp(plen),
                                              vp(pi), vp(pj), vp(pd)))
    tb_ij()  # first calls: code loading, pinned staging and workspace sizing
    t_ij = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        tb_ij()
        t_ij = min(t_ij, time.perf_counter() - t0)
    tb()
    best, timing = 1e9, None
    for _ in range(3):
        t0 = time.perf_counter()
        tb()
        dt = time.perf_counter() - t0
        if dt < best:
            fill, walk, db, pe = C.c_float(), C.c_float(), C.c_uint64(), C.c_uint64()
            lib.rawdtw_traceback_timing(eng._ctx, C.byref(fill), C.byref(walk), C.byref(db), C.byref(pe))
            best, timing = dt, (fill.value, walk.value, db.value, pe.value)
    fill_ms, walk_ms, dir_bytes, path_elems = timing
    # SURVEY.md 8d, per traceback job: operands + result + descriptor, ceil(nm/4) direction bytes written, ceil((n+m-1)/4)
    # read back by the walk, 12 L path bytes out
    alg = int((4 * (n + m) + 36 + (n * m + 3) // 4 + (n + m - 1 + 3) // 4).sum()) + 12 * int(path_elems)
    kms = fill_ms + walk_ms
    out["traceback"] = {"jobs": int(len(jobs)), "cells": cells, "path_elements": int(path_elems), "direction_bytes": int(dir_bytes),
                        "kernels_ms": {"fill": round(fill_ms, 4), "walk": round(walk_ms, 4)},
                        "kernels_gcups": cells / max(kms, 1e-9) / 1e6, "fill_gcups": cells / max(fill_ms, 1e-9) / 1e6,
                        "algorithmic_bytes": alg, "achieved_gbs": alg / max(kms, 1e-9) / 1e6,
                        "hbm_frac": alg / max(kms, 1e-9) / 1e6 / HBM_PEAK_GBS,
                        "call_ms_end_to_end": round(best * 1e3, 3), "gcups_end_to_end": cells / best / 1e9,
                        "call_ms_end_to_end_ij_arrays": round(t_ij * 1e3, 3),
                        "call_is": "rawdtw_traceback_batch_steps: plan, fill, walk, the paths home as one step byte + one distance an element "
                                   "(5 bytes instead of 12) into the caller's arrays; `_ij_arrays`: rawdtw_traceback_batch, which rebuilds (i, j) from "
                                   "the steps on the host while it writes three arrays",
                        "note": "kernels_ms = HIP events around the fill (k_full_wave<.,true,.>: the matrix fill writing 2-bit "
                                "directions) and the walk (k_tb_walk_wave + k_tb_finish) inside rawdtw_traceback_batch; the call "
                                "also plans on the host, brings 12 bytes a path element back over PCIe and copies them into the "
                                "caller's pageable arrays"}
    tpin.free()
    eng.close()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    backend = os.environ.get("RAWDTW_BENCH_BACKEND", "gloo" if args.dry_run else "nccl")
    if "RAWDTW_BENCH_DEVICE" in os.environ:  # rehearsal: several ranks on a one-GPU box
        local_rank = int(os.environ["RAWDTW_BENCH_DEVICE"])
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    from rawalign_amd import synth
    from rawalign_amd.shard import rank_seed, reduce_counters

    red_dev = torch.device("cuda", local_rank) if (dist is not None and backend == "nccl") else None
    sp = synth.SynthParams(n_reads=args.reads)
    if args.hit_prob is not None:
        sp.hit_prob = args.hit_prob
    if args.decoy_gap_median is not None:
        sp.decoy_gap_median = args.decoy_gap_median
    if args.decoys_per_read is not None:
        sp.decoys_per_read = args.decoys_per_read

    def barrier():
        if dist is not None:
            dist.barrier()

    if args.dry_run:
        # no device: counters from the synthetic generator only, a nominal time -- exercises launch, sharding seeds,
        # the counter reduction and the line's shape
        ref = synth.make_reference([min(args.genome, 200_000)], seed=SEED)
        offs = {(0, 1): 0, (0, 0): (len(ref.forward[0]) + 3) & ~3}
        cb, info = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=min(args.reads, 256)), seed=rank_seed(SEED, rank))
        jobs = int(np.maximum(np.diff(cb.anchor_off.astype(np.int64)) - 1, 0).sum())
        barrier()
        (reads_t, chains_t, jobs_t), T = reduce_counters(dist, [cb.n_reads * args.steps, cb.n_chains * args.steps, jobs * args.steps],
                                                         1e-3 * args.steps, device=None)
        if rank == 0:
            print(json.dumps({"metric": "DTW GCUPS", "value": 0.0, "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": T / args.steps * 1e3, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                              "config": {"workload": WORKLOAD, "reads_per_gpu": cb.n_reads},
                              "totals_over_timed_steps": {"reads": reads_t, "chains": chains_t, "dtw_jobs": jobs_t}}))
        if dist is not None:
            dist.destroy_process_group()
        return

    torch.cuda.set_device(local_rank)
    torch.cuda.init()
    import rawalign_amd as ra

    lib = ra.load_library()
    slots = max(1, args.inflight)
    # ---- contexts: ONE resident reference arena per GPU, shared by the in-flight contexts ----
    ref = synth.make_reference([args.genome], seed=SEED)
    opt = ra.MapOpt()  # sparse, banded=0.10, bonus 0.4, min score 20 (roptions.c:49-53)
    copt = opt.c_struct()
    engines = [ra.Engine(local_rank) for _ in range(slots)]
    engines[0].upload_reference(ref.forward, ref.reverse)
    for e in engines[1:]:
        e._check(lib.rawdtw_share_reference(e._ctx, engines[0]._ctx))
    offs = {(s_, st): engines[0].reference_offset(s_, st) for s_ in range(ref.n_seq) for st in (0, 1)}

    # ---- this rank's batches: one distinct shard per context, host arrays in pinned memory ----
    pin = Pinned(lib)
    B = []
    for sl in range(slots):
        cb, inf = synth.make_candidate_batch(ref, offs, sp, seed=rank_seed(SEED, rank) + 104729 * sl)
        ev_off = inf["ev_off"].astype(np.int64)
        n_ev = np.diff(ev_off)
        new_len = np.minimum(n_ev, inf["events_per_chunk"])           # the events the read's last chunk added
        seg_src = np.concatenate([[0], np.cumsum(new_len)]).astype(np.uint64)
        seg_dst = (ev_off[1:] - new_len).astype(np.uint32)
        idx = np.repeat(seg_dst.astype(np.int64), new_len) + (np.arange(int(new_len.sum())) - np.repeat(seg_src[:-1].astype(np.int64), new_len))
        d = {
            "cb": cb, "inf": inf, "n_reads": cb.n_reads, "n_chains": cb.n_chains,
            "events": pin.copy(cb.events), "chain_off": pin.copy(cb.chain_off.astype(np.uint64)),
            "anchor_off": pin.copy(cb.anchor_off.astype(np.uint64)), "anchors": pin.copy(cb.anchors),
            "ref_base": pin.copy(cb.ref_base.astype(np.uint64)), "read_base": pin.copy(cb.read_base.astype(np.uint32)),
            "new_events": pin.copy(cb.events[idx]), "seg_src": pin.copy(seg_src), "seg_dst": pin.copy(seg_dst),
            "score": pin.empty(cb.n_chains, np.float32), "keep": pin.empty(cb.n_chains, np.uint8),
        }
        # the anchor lists in the compact hand-over form (rawdtw_anchors_pack: 2-byte steps; what the `value_pcie` loop sends)
        from rawalign_amd.align import pack_anchors
        ca = pack_anchors(lib, cb.anchor_off, cb.anchors)
        d["c_heads"], d["c_unit_abs"], d["c_steps"] = pin.copy(ca.heads), pin.copy(ca.unit_abs), pin.copy(ca.steps)
        d["c_wide"], d["c_n_wide"] = pin.copy(ca.wide if len(ca.wide) else np.zeros(1, ca.wide.dtype)), len(ca.wide)
        d["compact_bytes"] = int(ca.nbytes)
        # device-resident copies of the three big arrays (the `value` loop uses them in place)
        d["t_anchors"] = torch.from_numpy(cb.anchors.view(np.uint8).copy()).cuda(local_rank)
        d["t_ref_base"] = torch.from_numpy(cb.ref_base.astype(np.uint64).view(np.uint8).copy()).cuda(local_rank)
        d["t_read_base"] = torch.from_numpy(cb.read_base.astype(np.uint32).view(np.uint8).copy()).cuda(local_rank)
        B.append(d)
        e = engines[sl]
        e._check(lib.rawdtw_events_reserve(e._ctx, len(cb.events)))
        e._check(lib.rawdtw_upload_events(e._ctx, vp(d["events"]), len(cb.events)))
        e.sync()
    torch.cuda.synchronize()

    handles = [C.c_void_p() for _ in range(slots)]
    live = [False] * slots

    def create(sl, resident):
        d, e = B[sl], engines[sl]
        if resident:
            st = lib.rawdtw_batch_create(e._ctx, C.byref(copt), d["n_reads"], vp(d["chain_off"]), vp(d["anchor_off"]),
                                         C.c_void_p(d["t_anchors"].data_ptr()), C.c_void_p(d["t_ref_base"].data_ptr()),
                                         C.c_void_p(d["t_read_base"].data_ptr()), C.byref(handles[sl]))
        else:
            st = lib.rawdtw_batch_create(e._ctx, C.byref(copt), d["n_reads"], vp(d["chain_off"]), vp(d["anchor_off"]),
                                         vp(d["anchors"]), vp(d["ref_base"]), vp(d["read_base"]), C.byref(handles[sl]))
        e._check(st)
        live[sl] = True

    def submit(sl, resident):
        """rawdtw_batch_submit: create + run, one call"""
        d, e = B[sl], engines[sl]
        if resident:
            st = lib.rawdtw_batch_submit(e._ctx, C.byref(copt), d["n_reads"], vp(d["chain_off"]), vp(d["anchor_off"]),
                                         C.c_void_p(d["t_anchors"].data_ptr()), C.c_void_p(d["t_ref_base"].data_ptr()),
                                         C.c_void_p(d["t_read_base"].data_ptr()), C.byref(handles[sl]))
        else:
            st = lib.rawdtw_batch_submit(e._ctx, C.byref(copt), d["n_reads"], vp(d["chain_off"]), vp(d["anchor_off"]),
                                         vp(d["anchors"]), vp(d["ref_base"]), vp(d["read_base"]), C.byref(handles[sl]))
        e._check(st)
        live[sl] = True

    def submit_compact(sl):
        d, e = B[sl], engines[sl]
        e._check(lib.rawdtw_batch_submit_compact(e._ctx, C.byref(copt), d["n_reads"], vp(d["chain_off"]), vp(d["anchor_off"]), vp(d["c_heads"]),
                                                 vp(d["c_unit_abs"]), vp(d["c_steps"]), vp(d["c_wide"]), d["c_n_wide"], vp(d["ref_base"]),
                                                 vp(d["read_base"]), C.byref(handles[sl])))
        live[sl] = True

    def fetch_destroy(sl, job_cost=None):
        d, e = B[sl], engines[sl]
        if job_cost is None:
            e._check(lib.rawdtw_batch_fetch_destroy(e._ctx, handles[sl], vp(d["score"]), vp(d["keep"])))
        else:
            e._check(lib.rawdtw_batch_fetch(e._ctx, handles[sl], vp(d["score"]), vp(d["keep"]), vp(job_cost)))
            lib.rawdtw_batch_destroy(handles[sl])
        live[sl] = False

    def collect(sl, sink):
        ms = np.zeros(8, np.float32); kind = np.zeros(8, np.uint32); nl = C.c_uint32(); nr = C.c_uint32()
        engines[sl]._check(lib.rawdtw_batch_collect(engines[sl]._ctx, handles[sl], vp(ms), vp(kind), 8, C.byref(nl), C.byref(nr)))
        sink.append(ms[:nl.value].copy())

    host_s = {"fetch": 0.0, "submit": 0.0, "steps": 0}
    issued = {"dtw": 0}  # DTW launches (k_runs dispatches) issued by this process so far: locates a pass in a kernel trace

    def pipeline(K, pcie, timed_launches=None, host=None):
        """K steps, every one a fresh batch; context k % slots; a context's previous batch is fetched before its next."""
        for e in engines:
            e.set_option("resident_arrays", 0 if pcie else 1)
        pc = time.perf_counter
        for k in range(K):
            sl = k % slots
            d, e = B[sl], engines[sl]
            t0 = pc()
            if live[sl]:
                if timed_launches is not None:
                    collect(sl, timed_launches)
                fetch_destroy(sl)
            t1 = pc()
            if pcie:  # the round's new events: one H2D of the packed chunk events + a scatter into the per-read arrays
                e._check(lib.rawdtw_events_append(e._ctx, vp(d["new_events"]), len(d["new_events"]), d["n_reads"],
                                                  vp(d["seg_src"]), vp(d["seg_dst"])))
            if pcie == "compact":
                submit_compact(sl)
            elif timed_launches is None:
                submit(sl, not pcie)
            else:  # HIP event pair around every launch, read when the context comes round again
                create(sl, not pcie)
                e._check(lib.rawdtw_batch_enqueue(e._ctx, handles[sl], 1))
            issued["dtw"] += 1
            if args.stagger_us > 0 and k < slots - 1:  # (see --stagger-us)
                t_s = pc()
                while (pc() - t_s) * 1e6 < args.stagger_us:
                    pass
            if host is not None:
                host["fetch"] += t1 - t0; host["submit"] += pc() - t1; host["steps"] += 1
        for sl in range(slots):
            if live[sl]:
                if timed_launches is not None:
                    collect(sl, timed_launches)
                fetch_destroy(sl)

    def timed_region(fn):
        """exactly K steps between barrier + synchronize pairs; returns seconds"""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        for e in engines:
            e.sync()
        torch.cuda.synchronize()
        barrier()
        return time.perf_counter() - t0

    def repeat_region(fn_of_steps):
        """The timed figure of one loop.  A region of exactly K steps starts and ends with an idle pipeline, and at K = 20 its
        fill, drain and the contexts' staggered start are a tenth of it (round 3: 755 GCUPS at --steps 20 against 826 at 160).
        So one BRACKET (barrier + synchronize on both sides, as ever) times `reps` consecutive K-step regions with no drain
        between them -- reps * K steps of the same loop, the stagger once at its start -- and a step's time is the bracket's
        divided by reps * K.  Three brackets; the median counts.  Returns (bracket seconds, reps)."""
        first = timed_region(lambda: fn_of_steps(K))
        first_all = first
        if dist is not None:  # every rank runs the same number of steps (each bracket has two barriers)
            f = torch.tensor([first], dtype=torch.float64, device=red_dev)
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
            first_all = float(f.item())
        n_br = 3
        reps = int(min(400, max(1, np.ceil(args.min_region_ms * 1e-3 / n_br / max(first_all, 1e-6)))))
        while (reps * K) % slots:  # whole turns of the contexts: every bracket holds every batch equally often
            reps += 1
        ts = [timed_region(lambda: fn_of_steps(reps * K)) for _ in range(n_br)]
        if os.environ.get("RAWDTW_BENCH_DUMP"):  # every bracket's time, for a look at their spread
            print("first_region_ms %.3f  brackets_ms (%d x %d steps)" % (first * 1e3, reps, K), " ".join("%.2f" % (x * 1e3) for x in ts), file=sys.stderr)
        return ts, reps

    K = args.steps
    # ---- warm-up (workspace pools, code objects), then the three timed loops ----
    pipeline(max(args.warmup, slots), pcie=False)
    if args.trace_fresh:
        t = timed_region(lambda: pipeline(args.trace_fresh, pcie=False))
        print("fresh-batch loop: %d steps, %.4f ms per step" % (args.trace_fresh, t * 1e3 / args.trace_fresh), file=sys.stderr)
        return
    pipeline(slots, pcie=True)
    pipeline(slots, pcie="compact")
    t_fresh, reps_fresh = repeat_region(lambda n: pipeline(n, pcie=False, host=host_s))
    t_pcie, reps_pcie = repeat_region(lambda n: pipeline(n, pcie="compact"))
    t_pcie_plain, reps_pcie_plain = repeat_region(lambda n: pipeline(n, pcie=True))
    # host cost of one create call, steady state (inputs resident / from pinned host memory)
    create_ms = {}
    for mode, pc in (("resident", False), ("from_host", True)):
        for e in engines:
            e.set_option("resident_arrays", 0 if pc else 1)
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            create(0, not pc)
            ts.append((time.perf_counter() - t0) * 1e3)
            engines[0]._check(lib.rawdtw_batch_run(engines[0]._ctx, handles[0]))
            issued["dtw"] += 1
            fetch_destroy(0)
        create_ms[mode] = round(float(np.median(ts)), 4)
    # launch durations inside the pipeline (HIP events on the context's stream), and the planning kernels' GPU time
    launches_in_pipeline = []
    for e in engines:
        e.set_option("time_plan", 1)
        # (the timed passes create and enqueue in two calls with nothing in between: the wide bands' launch goes out with the
        # planning launches as inside rawdtw_batch_submit, so that the brackets are those of the loop `value` times)
        e.set_option("wide_at_create", 1)
    timed_pass_first = issued["dtw"]
    pipeline(max(K, 2 * slots), pcie=False, timed_launches=launches_in_pipeline)
    timed_pass_count = issued["dtw"] - timed_pass_first
    # the scan launches of one batch alone on the chip (everything else has drained) ...
    for e in engines:
        e.sync()
    plan_alone = []
    for rep_ in range(3):
        create(0, True)
        ms = C.c_float()
        engines[0].sync()
        lib.rawdtw_batch_plan_ms(engines[0]._ctx, handles[0], C.byref(ms))
        plan_alone.append(ms.value)
        lib.rawdtw_batch_destroy(handles[0])
        live[0] = False
    # ... and with all contexts submitting at once
    plan_ms, wide_ms = [], []
    for sl in range(slots):
        create(sl, True)
        engines[sl]._check(lib.rawdtw_batch_run(engines[sl]._ctx, handles[sl]))
    for sl in range(slots):
        ms = C.c_float()
        engines[sl].sync()
        lib.rawdtw_batch_plan_ms(engines[sl]._ctx, handles[sl], C.byref(ms))
        plan_ms.append(ms.value)
        lib.rawdtw_batch_wide_ms(engines[sl]._ctx, handles[sl], C.byref(ms))
        wide_ms.append(ms.value)
    for e in engines:
        e.set_option("time_plan", 0)
        e.set_option("wide_at_create", 0)
    # ---- kernel replay: the resident planned batches' launches only ----
    infos = []
    for sl in range(slots):
        pi, nc = ra._lib.PlanInfo(), C.c_uint64()
        engines[sl]._check(lib.rawdtw_batch_info(handles[sl], C.byref(pi), C.byref(nc)))
        infos.append({k: int(getattr(pi, k)) for k, _ in ra._lib.PlanInfo._fields_})
    # the tiles' launch (k_runs) moves the tile-class jobs' bytes; the side list's are k_wide's (counter words: rawdtw_internal.h)
    cntw = (C.c_uint64 * 64)(); ncw = C.c_uint32()
    engines[0]._check(lib.rawdtw_batch_stream_counters(engines[0]._ctx, handles[0], cntw, 64, C.byref(ncw)))
    K_TILE_JOBS, K_TILE_BYTES = lib.rawdtw_batch_stream_counter_index(b"tile_jobs"), lib.rawdtw_batch_stream_counter_index(b"tile_bytes")
    assert K_TILE_JOBS >= 0 and K_TILE_BYTES >= 0
    tile_jobs0, tile_bytes0 = (int(cntw[K_TILE_JOBS]), int(cntw[K_TILE_BYTES])) if ncw.value else (infos[0]["n_jobs"], infos[0]["algorithmic_bytes"])

    def replay(n=K):
        for k in range(n):
            sl = k % slots
            engines[sl]._check(lib.rawdtw_batch_enqueue(engines[sl]._ctx, handles[sl], 0))
    replay()
    t_replay, reps_replay = repeat_region(replay)
    # one context alone, launches back to back: the dominant kernel without neighbours
    ms = np.zeros(8, np.float32); kind = np.zeros(8, np.uint32); nl = C.c_uint32()
    engines[0]._check(lib.rawdtw_batch_run_reps(engines[0]._ctx, handles[0], 10, vp(ms), vp(kind), 8, C.byref(nl)))
    alone_ms = ms[:nl.value].copy()
    # results: mapped reads per batch, and the per-job costs of batch 0 for the CPU cross-check
    job_cost = np.zeros(infos[0]["n_jobs"], np.float32)
    mapped = []
    for sl in range(slots):
        fetch_destroy(sl, job_cost if sl == 0 else None)
        cb = B[sl]["cb"]
        roc = np.repeat(np.arange(cb.n_reads), np.diff(cb.chain_off.astype(np.int64)))
        mapped.append(int(len(np.unique(roc[B[sl]["keep"].astype(bool)]))))  # reads with >= 1 surviving chain

    # ---- reduce: counters summed, each repetition's time = max over ranks ----
    n_f = reps_fresh * K  # steps of one bracket of the fresh-batch loop: what the totals below are over
    steps_of_slot = [len(range(sl, n_f, slots)) for sl in range(slots)]

    def total(vals):
        return int(sum(v * r for v, r in zip(vals, steps_of_slot)))
    counters = [args.reads * n_f, total([d["n_chains"] for d in B]), total([i["n_jobs"] for i in infos]),
                total([i["cells"] for i in infos]), total(mapped), total([i["algorithmic_bytes"] for i in infos])]

    def reduce_times(ts):
        t = torch.tensor(ts, dtype=torch.float64, device=red_dev)
        if dist is not None:
            n = torch.tensor([len(ts)], dtype=torch.int64, device=red_dev)
            dist.all_reduce(n, op=dist.ReduceOp.MIN)
            t = t[: int(n.item())].contiguous()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.cpu()]
    (reads_t, chains_t, jobs_t, cells_t, mapped_t, bytes_t), _ = reduce_counters(dist, counters, 0.0, device=red_dev)
    r_fresh, r_pcie, r_replay = reduce_times(t_fresh), reduce_times(t_pcie), reduce_times(t_replay)
    r_pcie_plain = reduce_times(t_pcie_plain)

    if rank == 0:
        T = float(np.median(r_fresh))  # one bracket of the fresh-batch loop: n_f steps
        # the other loops' brackets hold their own numbers of steps: brought to n_f steps' worth (every bracket is whole turns of the contexts)
        Tp = float(np.median(r_pcie)) * n_f / (reps_pcie * K)
        Tr = float(np.median(r_replay)) * n_f / (reps_replay * K)
        Tpp = float(np.median(r_pcie_plain)) * n_f / (reps_pcie_plain * K)
        lp = np.array(launches_in_pipeline)  # rows: steps, columns: [k_wide, k_runs, fold + select, (select: in the launch before)]
        dms = float(lp[:, 1].mean())
        plan_pipe, plan_al = float(np.median(plan_ms)), float(np.median(plan_alone))
        # (k_wide goes out with the planning launches inside rawdtw_batch_submit, in front of k_runs after a separate create --
        # the timed pass creates and enqueues separately: whichever bracket holds the launch)
        wide_pipe = max(float(np.median(wide_ms)), float(lp[:, 0].mean()))
        sum_pipe = plan_pipe + wide_pipe + float(lp[:, 1:].mean(axis=0).sum())
        sum_alone = plan_al + float(alone_ms[:4].sum())
        dbytes = tile_bytes0  # the passes' jobs: what k_runs moves (the side list's jobs are k_wide's)
        achieved = dbytes / (dms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        for tname in ("traffic_r04_large.json" if args.genome >= 10 ** 8 else "traffic_r04.json", "traffic_r03.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if not os.path.exists(tpath):
                continue
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == WORKLOAD and tj.get("reads") == args.reads:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/%s (PMC passes over the same batch, collected separately; not measured by this run)" % tname
                    break
            except Exception:
                pass
        out = {
            "metric": "DTW GCUPS", "value": cells_t / T / 1e9, "unit": "GCUPS", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": T / n_f * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD, "reads_per_gpu": args.reads, "genome_bp": args.genome,
                       "border_constraint": "sparse", "fill_method": "banded=0.10", "batches_in_flight": slots,
                       "sharding": f"reads over {world} gpu(s), one resident reference arena per gpu",
                       "synth": {"hit_prob": sp.hit_prob, "decoy_gap_median": sp.decoy_gap_median, "decoys_per_read": sp.decoys_per_read}},
            "value_is": "value_resident: fresh batch every step (rawdtw_batch_submit = scan of the anchor list + DTW launch + fold + select "
                        "enqueued by one call; rawdtw_batch_fetch_destroy of score/keep when the context comes round again), inputs "
                        "resident in HBM, median of `repeats` regions of exactly `steps` steps",
            "value_resident": cells_t / T / 1e9, "value_pcie": cells_t / Tp / 1e9,
            "value_pcie_is": "the same loop with the step's host hand-over inside (new events + anchor lists from pinned host memory over "
                             "PCIe): what a caller whose anchors are produced on the host gets end to end",
            "repeats": reps_fresh, "brackets": len(r_fresh), "timed_region_ms_total": round(sum(r_fresh) * 1e3, 3),
            "region_is": "one timed bracket (barrier + synchronize on both sides) = `repeats` consecutive regions of `steps` steps of the same "
                         "loop with no drain between them (the contexts' staggered start once per bracket); ms_per_step = bracket / (repeats * "
                         "steps); `brackets` brackets, the median counts -- so that the figure does not depend on --steps (round 3: a 20-step "
                         "region was 9 % pipeline fill, drain and stagger)",
            "region_ms": {"median": round(T * 1e3, 4), "min": round(min(r_fresh) * 1e3, 4), "max": round(max(r_fresh) * 1e3, 4), "steps": n_f},
            "reads_per_s": reads_t / T, "dtw_stage_rounds_per_s": mapped_t / T, "jobs_per_s": jobs_t / T,
            "dtw_stage_rounds_is": "chunk rounds of the DTW stage per second in which >= 1 candidate chain of the (synthetic) read survives "
                                   "dtw_min_score -- a capacity of this stage alone, not RawAlign's mapped reads/s: event detection, seeding "
                                   "and chaining are imitated by their output (rawalign_amd/synth.py)",
            "pipeline_pcie": {"gcups": cells_t / Tp / 1e9, "dtw_stage_rounds_per_s": mapped_t / Tp, "ms_per_step": Tp / n_f * 1e3,
                              "repeats": reps_pcie,
                              "h2d_bytes_per_step": int(B[0]["new_events"].nbytes + B[0]["compact_bytes"] + B[0]["ref_base"].nbytes + B[0]["read_base"].nbytes +
                                                        B[0]["anchor_off"].nbytes + B[0]["chain_off"].nbytes),
                              "h2d_bytes": {"new_events": int(B[0]["new_events"].nbytes), "anchor_lists_compact": int(B[0]["compact_bytes"]),
                                            "anchor_lists_plain": int(B[0]["anchors"].nbytes)},
                              "note": "same loop with the step's host hand-over inside: the round's new events (last chunk of "
                                      "every read, rawdtw_events_append) and the anchor lists cross PCIe from pinned memory; the lists in "
                                      "the compact form (rawdtw_batch_submit_compact: 2-byte steps, decoded inside k_scan)",
                              "plain_anchor_lists": {"gcups": cells_t / Tpp / 1e9, "ms_per_step": Tpp / n_f * 1e3,
                                                     "h2d_bytes_per_step": int(B[0]["new_events"].nbytes + B[0]["anchors"].nbytes + B[0]["ref_base"].nbytes +
                                                                               B[0]["read_base"].nbytes + B[0]["anchor_off"].nbytes + B[0]["chain_off"].nbytes)}},
            "kernel_replay": {"gcups": cells_t / Tr / 1e9, "ms_per_step": Tr / n_f * 1e3, "repeats": reps_replay,
                              "note": "launches of already submitted resident batches only (round 1's headline)"},
            "host_ms_per_step": {k: round(host_s[k] / max(host_s["steps"], 1) * 1e3, 4) for k in ("fetch", "submit")} | {
                "note": "host wall time inside the two calls of one step of the timed fresh-batch loop: `fetch` "
                        "(rawdtw_batch_fetch_destroy) includes waiting for the context's previous batch -- the only blocking "
                        "call; `submit` (rawdtw_batch_submit = create + run) only enqueues"},
            "batch_create_ms": {"steady": create_ms["resident"], "from_pinned_host": create_ms["from_host"],
                                "note": "host wall time of one rawdtw_batch_create call, steady state: O(1) host work, it only enqueues (no "
                                        "synchronisation, no allocation); `planning_gpu_ms` = its launches (k_scan + k_side + k_plan; HIP events behind the "
                                        "hand-over's copies) on the "
                                        "device: alone on the chip / with all contexts submitting at once",
                                "planning_gpu_ms": round(plan_al, 4), "planning_gpu_ms_in_pipeline": round(plan_pipe, 4)},
            "kernel_ms_sum_per_batch": {"alone": round(sum_alone, 4), "in_pipeline": round(sum_pipe, 4),
                                        "overlap_factor": round(sum_pipe / (T / n_f * 1e3), 3),
                                        "note": "sum of a batch's launch durations (planning, k_wide, k_runs, fold + select; HIP events): each "
                                                "alone on the chip, and bracketed inside the pipeline, where the batches in flight stretch "
                                                "each other -- overlap_factor = that sum / ms_per_step = batches effectively in flight"},
            "totals_over_timed_steps": {"reads": reads_t, "chains": chains_t, "dtw_jobs": jobs_t, "cells": cells_t,
                                        "mapped_reads": mapped_t, "algorithmic_bytes": bytes_t},
            "batch0": {"reads": args.reads, "chains": B[0]["n_chains"], "dtw_jobs": infos[0]["n_jobs"], "cells": infos[0]["cells"],
                       "mapped_reads": mapped[0], "algorithmic_bytes": infos[0]["algorithmic_bytes"],
                       "tile_class_jobs": infos[0]["n_lane_jobs"], "wide_band_jobs": infos[0]["n_wave_band_jobs"]},
            "whole_step_hbm_frac": bytes_t / T / 1e9 / (HBM_PEAK_GBS * world),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "k_runs", "launch_ms": dms,
                         "algorithmic_bytes_per_launch": dbytes, "jobs_per_launch": tile_jobs0,
                         "launch_window": {"first_dispatch": timed_pass_first, "count": timed_pass_count,
                                           "note": "launch_ms is over these k_runs dispatches of the process (0-based, in issue "
                                                   "order); scripts/trace_window.py averages the same ones in a rocprofv3 kernel trace"},
                         "note": "achieved = algorithmic bytes of the launch's jobs (4(n+m)+36 each, SURVEY.md 8d: the tile-class jobs, "
                                 "99.5 % of the batch's; the wide bands are k_wide's) / launch_ms, the mean HIP-event bracket of the "
                                 "batch's k_runs launch inside the fresh-batch pipeline (recorded on the launch's own stream).  That bracket is an OVERLAPPED wall bracket: the launch shares the chip with "
                                 "the other contexts' launches (kernel_ms_sum_per_batch.overlap_factor).  `alone` = the same launch "
                                 "repeated on an idle chip: the kernel's own figure; `traffic` = HBM bytes by PMC counters",
                         "alone": {"launch_ms": float(alone_ms[1]), "achieved": dbytes / (float(alone_ms[1]) * 1e-3) / 1e9,
                                   "frac": dbytes / (float(alone_ms[1]) * 1e-3) / 1e9 / HBM_PEAK_GBS}},
            "launches": {"in_pipeline_ms": {"k_scan+k_side+k_plan": round(plan_pipe, 5), "k_wide": round(wide_pipe, 5), "k_runs": round(dms, 5),
                                            "k_fold_select": round(float(lp[:, 2].mean()), 5)},
                         "alone_ms": {"k_scan+k_side+k_plan": round(plan_al, 5), "k_wide": round(float(alone_ms[0]), 5), "k_runs": round(float(alone_ms[1]), 5),
                                      "k_fold_select": round(float(alone_ms[2]), 5)},
                         "note": "a batch's launches in stream order: the planning (scan of the anchor list, side list's class order, the "
                                 "passes' records and copy orders), the side list's wide bands (k_wide: 0.5 % of the jobs), the tiles' passes "
                                 "(k_runs), fold + select"},
        }
        if world == 1 and args.rounds > 0:
            out["chunk_rounds"] = rounds_block(engines[0], lib, copt, B[0]["cb"], B[0]["inf"], args.rounds, local_rank, pin)
        if world == 1 and args.modes_reads > 0:
            out["modes"] = modes_block(local_rank, args.modes_reads)
        if world == 1 and args.mapper_reads > 0:
            for e in engines:   # (the loops above are over: their contexts' workspaces go, the mapper makes its own)
                e.close()
            engines = []
            thr = args.mapper_threads
            if thr <= 0:
                quota = cgroup_cpu_quota()
                aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
                thr = max(1, min(aff, int(quota + 0.999))) if quota else min(aff, 16)
            out["mapper"] = mapper_block(lib, ref, local_rank, args.mapper_reads, thr, args.cpu_seconds * 3, not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            cb = B[0]["cb"]
            job_off = np.zeros(cb.n_chains + 1, np.uint64)
            nj = C.c_uint64()
            jobs = np.zeros(infos[0]["n_jobs"], ra.JOB_DTYPE)
            lib.rawdtw_batch_build_jobs(C.byref(copt), cb.n_chains, vp(B[0]["anchor_off"]), vp(B[0]["anchors"]), vp(B[0]["ref_base"]),
                                        vp(B[0]["read_base"]), vp(job_off), vp(jobs), len(jobs), C.byref(nj))
            n = len(ref.forward[0])
            pad = (n + 3) & ~3
            arena = np.zeros(2 * pad, np.float32)
            arena[:n] = ref.forward[0]
            arena[pad:pad + n] = ref.reverse[0]
            base, cpu_costs = cpu_baseline(jobs, cb.events, arena, infos[0]["cells"], args.cpu_seconds)
            out["cpu_baseline"] = base
            # free check: the CPU leg and the device leg computed the same jobs
            out["cpu_gpu_costs_identical"] = bool(np.array_equal(cpu_costs.view(np.uint32), job_cost[:len(cpu_costs)].view(np.uint32)))
            out["cpu_gpu_costs_compared"] = int(len(cpu_costs))
        print(json.dumps(out))
    pin.free()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
