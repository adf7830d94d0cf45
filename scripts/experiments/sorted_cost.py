#!/usr/bin/env python3
"""Cost of the shape-sorted tile class alone: all jobs with longer side <= 16 are left unscored (RAWDTW_DEBUG_SKIP),
so the tile launch's time is staging + the long/rare jobs, with and without sorting them into their own tiles."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rawalign_amd as ra
from rawalign_amd import synth
ref = synth.make_reference([4_600_000], seed=20231007)
eng = ra.Engine(0)
eng.upload_reference(ref.forward, ref.reverse)
offs = {(0, st): eng.reference_offset(0, st) for st in (0, 1)}
cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=16384), seed=20231007 + 7919)
opt = ra.MapOpt(); copt = opt.c_struct(); lib = eng.lib
p = lambda a: a.ctypes.data_as(C.c_void_p)
job_off = np.zeros(cb.n_chains + 1, np.uint64); nj = C.c_uint64()
lib.rawdtw_batch_build_jobs(C.byref(copt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base), p(cb.read_base), p(job_off), None, 0, C.byref(nj))
jobs = np.zeros(nj.value, ra.JOB_DTYPE)
lib.rawdtw_batch_build_jobs(C.byref(copt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base), p(cb.read_base), p(job_off), p(jobs), len(jobs), C.byref(nj))
eng.upload_events(cb.events)
N = np.maximum(jobs["n"], jobs["m"]).astype(np.int64); M = np.minimum(jobs["n"], jobs["m"]).astype(np.int64)
R0 = jobs["band_radius"].astype(np.int64); R = R0 + ((N - M) * R0 + N - 1) // N
sub = np.ascontiguousarray(jobs[(R <= 3) & (N <= 73)])
eng.set_option("merge_small", 0)
for skip in (None, "0,16,-1", "0,255,-1"):
    for sort in ({"sort_n": 0, "sort_r1_n": 0, "sort_r3": 0}, {"sort_n": 17, "sort_r1_n": 0, "sort_r3": 0}, {"sort_n": 17, "sort_r1_n": 9, "sort_r3": 1}):
        if skip is None: os.environ.pop("RAWDTW_DEBUG_SKIP", None)
        else: os.environ["RAWDTW_DEBUG_SKIP"] = skip
        for k, v in sort.items(): eng.set_option(k, v)
        plan = eng.plan(sub); plan.run(); eng.sync()
        ms = min(plan.run_timed()[0][2] for _ in range(5))
        print(json.dumps({"skip": skip, **sort, "tile_ms": round(ms, 4)}))
        plan.close()
