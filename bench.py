#!/usr/bin/env python3
"""bench.py -- DTW hot path of RawAlign on MI355X: DTW GCUPS (+ reads/s) and % of the HBM roofline.

One "step" = one pass of the hot path over one batch: every DTW job of every candidate chain of
every read of a chunk round (sparse border constraint, banded=0.10 fill: BASELINE.json configs[1]),
the align_chain fold and the per-read accept/cut loop -- all on the device, inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: reads shard across ranks (each rank holds a full replica of the reference signal and
its own reads); no data-path collective; RCCL is used only for the final counters and the
max-over-ranks time.  Weak scaling: per-GPU work is fixed.
"""
import argparse
import json
import os
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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reads", type=int, default=16384, help="reads per GPU in one batch (chunk round)")
    ap.add_argument("--genome", type=int, default=4_600_000, help="reference length in bases (E. coli K-12)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="mini-batches in flight per GPU (the reference's kt_pipeline keeps 2, rmap.cpp:1033); "
                         "step k runs batch k %% inflight, each on its own context/streams")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-merge", action="store_true",
                    help="launch the tile kernel and the two small banded kernels of a batch separately")
    ap.add_argument("--serial-launches", action="store_true",
                    help="run the launches of a step one after another (profiling: per-kernel counters without overlap)")
    return ap.parse_args()


def cpu_baseline(jobs, events, ref_arena, cells, threads, target_s=1.5):
    """The same job list on the host cores, through the REFERENCE's own compiled dtw.cpp when
    oracle/_ref is present ("reference"), else through the oracle's C restatement ("port")."""
    from oracle.loader import Oracle, RefDTW, build_oracle

    if RefDTW.available():
        impl, kind = RefDTW(), "reference"
    else:
        build_oracle(march_native=True)
        impl, kind = Oracle(), "port"
    # bounded sample: the whole batch if it is small enough, else a prefix of whole reads' jobs
    n = len(jobs)
    t0 = time.perf_counter()
    impl.batch_costs(jobs[: min(n, 200000)], events, ref_arena, threads)
    probe = time.perf_counter() - t0
    per_job = probe / min(n, 200000)
    take = int(min(n, max(200000, target_s / max(per_job, 1e-9))))
    reps = max(1, int(round(target_s / max(per_job * take, 1e-6))))
    reps = min(reps, 50)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = impl.batch_costs(jobs[:take], events, ref_arena, threads)
    dt = (time.perf_counter() - t0) / reps
    frac = take / n
    # ... and one thread on a prefix, for the per-core figure (single-threaded and all cores)
    take1 = int(min(n, max(50000, 1.0 / max(per_job * threads, 1e-9))))
    t0 = time.perf_counter()
    impl.batch_costs(jobs[:take1], events, ref_arena, 1)
    dt1 = time.perf_counter() - t0
    return {
        "value": cells * frac / dt / 1e9,
        "unit": "GCUPS",
        "cores": threads,
        "kind": kind,
        "sample": f"first {take} of {n} DTW jobs of the same batch ({frac * 100:.0f}% of its cells, cells pro-rated by job count), "
                  f"{reps} repetition(s), {threads} threads pulling jobs from a shared counter, {dt * reps:.1f} s wall",
        "jobs_per_s": take / dt,
        "single_thread": {"value": cells * (take1 / n) / dt1 / 1e9, "unit": "GCUPS",
                          "sample": f"first {take1} jobs, one thread, {dt1:.1f} s wall"},
    }, out, take


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        # rehearsal knobs (several ranks on a one-GPU box): RAWDTW_BENCH_BACKEND=gloo, RAWDTW_BENCH_DEVICE=0
        backend = os.environ.get("RAWDTW_BENCH_BACKEND", "nccl")
        if "RAWDTW_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["RAWDTW_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    import rawalign_amd as ra
    from rawalign_amd import synth
    from rawalign_amd.shard import rank_seed

    # reference signal (same on every rank), one resident replica per in-flight context
    ref = synth.make_reference([args.genome], seed=SEED)
    opt = ra.MapOpt()  # sparse, banded=0.10, bonus 0.4, min score 20 (roptions.c:49-53)
    slots = max(1, args.inflight)
    engines, batches, cbs, infos, create_ms = [], [], [], [], []
    for sl in range(slots):
        e = ra.Engine(local_rank)
        if args.serial_launches:
            e.set_option("serial_launches", 1)
        if args.no_merge:
            e.set_option("merge_small", 0)
        e.upload_reference(ref.forward, ref.reverse)
        offs = {(s_, st): e.reference_offset(s_, st) for s_ in range(ref.n_seq) for st in (0, 1)}
        # this rank's reads for this slot (distinct shards)
        cb_, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=args.reads),
                                            seed=rank_seed(SEED, rank) + 104729 * sl)
        e.upload_events(cb_.events)
        e.sync()
        t_c = time.perf_counter()
        b_ = ra.Batch(e, opt, cb_)   # uploads the anchor lists and plans the batch (on the device by default)
        create_ms.append((time.perf_counter() - t_c) * 1e3)
        engines.append(e); batches.append(b_); cbs.append(cb_); infos.append(b_.info())
        e.sync()
    eng, batch, cb, info = engines[0], batches[0], cbs[0], infos[0]

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync_all():
        for e in engines:
            e.sync()

    for k in range(max(args.warmup, 0)):
        batches[k % slots].enqueue(timed=False)
    sync_all()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):  # K steps, no host synchronisation in between, HIP events around every launch
        batches[k % slots].enqueue(timed=True)
    t_enq = time.perf_counter() - t0
    sync_all()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    launches, _ = batch.collect()
    for b_ in batches[1:]:
        b_.collect()
    runs_of_slot = [len(range(sl, args.steps, slots)) for sl in range(slots)]

    stats_timed = batch.launch_stats(with_cells=False)  # as launched in the timed region (small classes merged in)
    # a second, untimed pass with every kernel launched on its own, one after another: clean per-kernel durations
    eng.set_option("serial_launches", 1)
    eng.set_option("merge_small", 0)
    isolated = batch.run_reps(max(3, min(args.steps, 10)), timed=True)
    stats = batch.launch_stats(with_cells=False)        # per kernel
    eng.set_option("merge_small", 0 if args.no_merge else 1)
    eng.set_option("serial_launches", 1 if args.serial_launches else 0)

    mapped_slots = []
    job_cost = None
    for sl in range(slots):
        if sl == 0:
            score, keep, job_cost = batches[sl].fetch(with_job_costs=True)
        else:
            score, keep = batches[sl].fetch()
        roc = np.repeat(np.arange(cbs[sl].n_reads), np.diff(cbs[sl].chain_off.astype(np.int64)))
        mapped_slots.append(int(len(np.unique(roc[keep.astype(bool)]))))  # reads with >= 1 surviving chain

    def total(key_or_list):
        vals = key_or_list if isinstance(key_or_list, list) else [i[key_or_list] for i in infos]
        return int(sum(v * r for v, r in zip(vals, runs_of_slot)))

    from rawalign_amd.shard import reduce_counters

    # the path's only collective: final counters (sum) and the step time (max over ranks)
    # (totals over the K timed steps of this rank)
    (reads_t, chains_t, jobs_t, cells_t, mapped_t, bytes_t), T = reduce_counters(
        dist, [args.reads * args.steps, total("n_chains"), total("n_jobs"), total("cells"), total(mapped_slots),
               total("algorithmic_bytes")],
        elapsed, device=torch.device("cuda", local_rank)
        if (dist is not None and os.environ.get("RAWDTW_BENCH_BACKEND", "nccl") == "nccl") else None)

    if rank == 0:
        # dominant kernel = the launch with the largest duration when run alone; its duration inside the
        # timed region (where launches overlap on several streams) is what `achieved` is priced with
        def waves(i):  # wavefronts launch i puts on the machine
            k = isolated[i][0]
            return stats[i]["n_jobs"] / 64.0 if k in (1, 7, 9) else stats[i]["n_jobs"]
        # ... among the launches that can fill the chip (>= 256 CUs x 8 waves); a launch of three long jobs
        # has the longest duration but occupies three wavefronts
        # ... and that carries the bytes: a launch of a few long jobs can have the longest duration (it is
        # a latency pole that overlaps other work) while occupying a handful of wavefronts
        filling = [i for i in range(len(isolated)) if waves(i) >= 2048] or list(range(len(isolated)))
        dom = max(filling, key=lambda i: stats[i]["algorithmic_bytes"])
        dms = launches[dom][2]
        dkind, dparam = stats_timed[dom]["kind"], stats_timed[dom]["param"]
        dbytes = stats_timed[dom]["algorithmic_bytes"]   # everything that launch carried in the timed region
        dbytes_iso = stats[dom]["algorithmic_bytes"]     # the kernel's own jobs (isolated pass: nothing merged in)
        achieved = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0

        def kernel_name(kind, param):
            if kind == 8 and param in (-16, -8):
                return "band_grp16" if param == -16 else "band_grp8"
            return ra.Engine.KIND_NAMES.get(kind, str(kind))
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == WORKLOAD and tj.get("reads") == args.reads:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        kname = kernel_name(dkind, dparam)
        out = {
            "metric": "DTW GCUPS",
            "value": cells_t / T / 1e9,
            "unit": "GCUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": T / args.steps * 1e3,
            "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": WORKLOAD, "reads_per_gpu": args.reads, "genome_bp": args.genome,
                       "border_constraint": "sparse", "fill_method": "banded=0.10", "batches_in_flight": slots,
                       "sharding": f"reads over {world} gpu(s), reference replicated"},
            "reads_per_s": reads_t / T,
            "mapped_reads_per_s": mapped_t / T,
            "jobs_per_s": jobs_t / T,
            "totals_over_timed_steps": {"reads": reads_t, "chains": chains_t, "dtw_jobs": jobs_t, "cells": cells_t,
                                        "mapped_reads": mapped_t, "algorithmic_bytes": bytes_t},
            "batch0": {"reads": args.reads, "chains": info["n_chains"], "dtw_jobs": info["n_jobs"],
                       "cells": info["cells"], "mapped_reads": mapped_slots[0],
                       "algorithmic_bytes": info["algorithmic_bytes"]},
            # outside the timed region (inputs resident): what creating a mini-batch costs on the host side
            "batch_create_ms": {"first": round(create_ms[0], 3), "steady": round(min(create_ms[1:] or create_ms), 3),
                                "note": "rawdtw_batch_create: anchor upload + planning (device planner unless "
                                        "RAWDTW_OPTS=device_plan=0); `first` includes code loading and workspace allocation"},
            "whole_step_hbm_frac": bytes_t / T / 1e9 / (HBM_PEAK_GBS * world),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": f"{kname}(param={dparam})", "launch_ms": dms,
                         "note": "launch_ms is the HIP-event bracket inside the timed region, where this launch shares "
                                 "the chip with the other batches in flight (and, when merged, carries the batch's two "
                                 "small banded classes, whose longest job can outlast the tiles); `isolated` is the "
                                 "kernel's own jobs launched alone on an idle chip (second pass, nothing merged)",
                         "isolated": {"kernel": kernel_name(stats[dom]["kind"], stats[dom]["param"]),
                                      "launch_ms": isolated[dom][2],
                                      "achieved": dbytes_iso / (isolated[dom][2] * 1e-3) / 1e9 if isolated[dom][2] > 0 else 0.0,
                                      "frac": (dbytes_iso / (isolated[dom][2] * 1e-3) / 1e9 / HBM_PEAK_GBS) if isolated[dom][2] > 0 else 0.0,
                                      "algorithmic_bytes_per_launch": dbytes_iso, "jobs_per_launch": stats[dom]["n_jobs"]},
                         "algorithmic_bytes_per_launch": dbytes, "jobs_per_launch": stats_timed[dom]["n_jobs"]},
            "launches": [{"kernel": kernel_name(stats_timed[i]["kind"], stats_timed[i]["param"]),
                          "param": stats_timed[i]["param"], "ms": round(ms, 5),
                          "jobs": stats_timed[i]["n_jobs"], "algorithmic_bytes": stats_timed[i]["algorithmic_bytes"],
                          "alone": {"kernel": kernel_name(stats[i]["kind"], stats[i]["param"]), "ms": round(isolated[i][2], 5),
                                    "jobs": stats[i]["n_jobs"], "algorithmic_bytes": stats[i]["algorithmic_bytes"]}}
                         for i, (_, _, ms) in enumerate(launches)],
        }
        if world == 1 and not args.no_cpu_baseline:
            # rebuild the job list exactly as the device batch built it
            import ctypes as C

            lib = eng.lib
            copt = opt.c_struct()
            p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            job_off = np.zeros(cb.n_chains + 1, np.uint64)
            nj = C.c_uint64()
            jobs = np.zeros(info["n_jobs"], ra.JOB_DTYPE)
            lib.rawdtw_batch_build_jobs(C.byref(copt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base),
                                        p(cb.read_base), p(job_off), p(jobs), len(jobs), C.byref(nj))
            n = len(ref.forward[0])
            pad = (n + 3) & ~3
            arena = np.zeros(2 * pad, np.float32)
            arena[:n] = ref.forward[0]
            arena[pad:pad + n] = ref.reverse[0]
            base, cpu_costs, take = cpu_baseline(jobs, cb.events, arena, info["cells"], args.cpu_threads)
            out["cpu_baseline"] = base
            # free check: the CPU leg and the device leg computed the same jobs
            same = bool(np.array_equal(cpu_costs.view(np.uint32), job_cost[:take].view(np.uint32)))
            out["cpu_gpu_costs_identical"] = same
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
