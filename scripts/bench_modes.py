#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): the other BASELINE.json configurations on one GPU.

  --mode global_full    : config 3 scoring  (border=global, fill=full)            -> k_full_wave
  --mode global_banded  : the reference's default fill on global chains            -> k_band_wreg<C>
  --mode traceback      : --dtw-output-cigar on the best chain of every read       -> k_full_wave<.,TB> + k_tb_walk
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import rawalign_amd as ra  # noqa: E402
from rawalign_amd import synth  # noqa: E402

YEAST = [230218, 813184, 316620, 1531933, 576874, 270161, 1090940, 562643, 439888, 745751, 666816, 1078177, 924431,
         784333, 1091291, 948066, 85779]  # S. cerevisiae S288C chromosome lengths (12.1 Mb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="global_full")
    ap.add_argument("--reads", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--max-chunks", type=int, default=6)
    args = ap.parse_args()
    eng = ra.Engine(0)
    ref = synth.make_reference(YEAST, seed=20231005 + 3)
    eng.upload_reference(ref.forward, ref.reverse)
    offs = {(s, st): eng.reference_offset(s, st) for s in range(ref.n_seq) for st in (0, 1)}
    cb, sinfo = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=args.reads, max_chunks=args.max_chunks),
                                           seed=77)
    eng.upload_events(cb.events)
    out = {"mode": args.mode, "reads": args.reads, **{k: v for k, v in sinfo.items() if k != "ev_off"}}
    if args.mode in ("global_full", "global_banded"):
        opt = ra.MapOpt(dtw_border_constraint=0, dtw_fill_method=0 if args.mode == "global_full" else 1)
        batch = ra.Batch(eng, opt, cb)
        info = batch.info()
        batch.run_reps(1, timed=False)
        t0 = time.perf_counter()
        launches = batch.run_reps(args.steps, timed=True)
        dt = (time.perf_counter() - t0) / args.steps
        eng.set_option("serial_launches", 1)
        iso = batch.run_reps(2, timed=True)
        eng.set_option("serial_launches", 0)
        stats = batch.launch_stats(with_cells=True)
        out.update({"jobs": info["n_jobs"], "cells": info["cells"], "ms_per_step": dt * 1e3,
                    "GCUPS": info["cells"] / dt / 1e9,
                    "launches": [{"kernel": ra.Engine.KIND_NAMES.get(k, str(k)), "param": p, "ms": round(ms, 4),
                                  "ms_isolated": round(iso[i][2], 4), "jobs": stats[i]["n_jobs"],
                                  "cells": stats[i]["cells"],
                                  "GCUPS_isolated": round(stats[i]["cells"] / max(iso[i][2], 1e-9) / 1e6, 1)}
                                 for i, (k, p, ms) in enumerate(launches)]})
    elif args.mode == "traceback":
        # the best (first) chain of every read, global + full, as rmap.cpp:715-717 would re-align it
        import ctypes as C
        from rawalign_amd._lib import AlignOpt

        lib = eng.lib
        copt = AlignOpt(0, 0, 0.10, 0.4, 20.0, 1)
        firsts = [int(cb.chain_off[r]) for r in range(cb.n_reads) if cb.chain_off[r + 1] > cb.chain_off[r]]
        jobs = np.zeros(len(firsts), ra.JOB_DTYPE)
        for k, c in enumerate(firsts):
            a = cb.anchors[int(cb.anchor_off[c]):int(cb.anchor_off[c + 1])]
            one = np.zeros(1, ra.JOB_DTYPE)
            lib.rawdtw_chain_build_jobs(C.byref(copt), a.ctypes.data_as(C.c_void_p), len(a), int(cb.ref_base[c]),
                                        int(cb.read_base[c]), 1, one.ctypes.data_as(C.c_void_p))
            jobs[k] = one[0]
        cells = int((jobs["n"].astype(np.int64) * jobs["m"]).sum())
        eng.traceback_batch(jobs, cb.events)  # first call: code loading, pinned staging and workspace sizing
        dt = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            res = eng.traceback_batch(jobs, cb.events)
            dt = min(dt, time.perf_counter() - t0)
        out.update({"jobs": len(jobs), "cells": cells, "seconds_end_to_end": dt, "GCUPS_end_to_end": cells / dt / 1e9,
                    "path_elements": int(sum(len(r) for r in res)),
                    "direction_bytes": int(cells // 4)})
    print(json.dumps(out, default=lambda o: o.tolist() if hasattr(o, "tolist") else float(o)))


if __name__ == "__main__":
    main()
