#!/usr/bin/env python3
"""Tile-kernel cost by job class: runs the bench batch's tile jobs with one class (by longer side and radius) at a time
replaced by 1x1 jobs, two k_band_tile dispatches per variant, so that a PMC pass (rocprofv3 --pmc SQ_INSTS_VALU ...) attributes the VALU work.
Prints the subset table; the dispatch order in the counter CSV is the order printed here."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rawalign_amd as ra
from rawalign_amd import synth

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ref = synth.make_reference([4_600_000], seed=20231007)
eng = ra.Engine(0)
eng.upload_reference(ref.forward, ref.reverse)
offs = {(0, st): eng.reference_offset(0, st) for st in (0, 1)}
cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=reads), seed=20231007 + 7919)
opt = ra.MapOpt(); copt = opt.c_struct(); lib = eng.lib
p = lambda a: a.ctypes.data_as(C.c_void_p)
job_off = np.zeros(cb.n_chains + 1, np.uint64); nj = C.c_uint64()
lib.rawdtw_batch_build_jobs(C.byref(copt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base), p(cb.read_base), p(job_off), None, 0, C.byref(nj))
jobs = np.zeros(nj.value, ra.JOB_DTYPE)
lib.rawdtw_batch_build_jobs(C.byref(copt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base), p(cb.read_base), p(job_off), p(jobs), len(jobs), C.byref(nj))
eng.upload_events(cb.events)
N = np.maximum(jobs["n"], jobs["m"]).astype(np.int64); M = np.minimum(jobs["n"], jobs["m"]).astype(np.int64)
R0 = jobs["band_radius"].astype(np.int64); R = R0 + ((N - M) * R0 + N - 1) // N
tile = (R <= 3) & (N <= 73)
classes = [("none", None), ("N<=4", (0, 4, -1)), ("5<=N<=8", (5, 8, -1)), ("9<=N<=16 R1", (9, 16, 1)), ("9<=N<=16 R2", (9, 16, 2)),
           ("9<=N<=16 R3", (9, 16, 3)), ("17<=N<=32", (17, 32, -1)), ("N>=33", (33, 255, -1)), ("all", (0, 255, -1))]
rows = []
sub = np.ascontiguousarray(jobs[tile])
for name, skip in classes:
    # RAWDTW_DEBUG_SKIP (planner, profiling only): the class's jobs stay staged but unscored, everything else as in the
    # real batch, so the difference to the "none" row is what the class costs
    if skip is None: os.environ.pop("RAWDTW_DEBUG_SKIP", None)
    else: os.environ["RAWDTW_DEBUG_SKIP"] = "%d,%d,%d" % skip
    plan = eng.plan(sub)
    plan.run(); eng.sync()
    ms = plan.run_timed()
    if skip is None: sel = np.zeros(len(jobs), bool)
    else: sel = tile & (N >= skip[0]) & (N <= skip[1]) & ((R == skip[2]) | (skip[2] < 0))
    cells = int((N[sel] * np.minimum(2 * R[sel] + 1, M[sel])).sum())
    rows.append({"skipped": name, "jobs": int(sel.sum()), "approx_cells": cells, "ms": [round(x[2], 4) for x in ms]})
    plan.close()
os.environ.pop("RAWDTW_DEBUG_SKIP", None)
print(json.dumps(rows, indent=1))
