#!/usr/bin/env python3
"""Host-side costs around the device path on the bench workload: job building, planning (binning + tiles),
H2D/D2H, i.e. what a caller pays per mini-batch when it hands over host buffers (PCIe-inclusive rate)."""
import ctypes as C, json, os, sys, time
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
opt = ra.MapOpt()
copt = opt.c_struct()
lib = eng.lib
p = lambda a: a.ctypes.data_as(C.c_void_p)
t0 = time.perf_counter()
job_off = np.zeros(cb.n_chains + 1, np.uint64); nj = C.c_uint64()
lib.rawdtw_batch_build_jobs(C.byref(copt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base), p(cb.read_base), p(job_off), None, 0, C.byref(nj))
jobs = np.zeros(nj.value, ra.JOB_DTYPE)
lib.rawdtw_batch_build_jobs(C.byref(copt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base), p(cb.read_base), p(job_off), p(jobs), len(jobs), C.byref(nj))
t_build = time.perf_counter() - t0
t0 = time.perf_counter(); eng.upload_events(cb.events); eng.sync(); t_h2d = time.perf_counter() - t0
t0 = time.perf_counter(); plan = eng.plan(jobs); t_plan = time.perf_counter() - t0
info = plan.info()
plan.run(); eng.sync()
t0 = time.perf_counter(); plan.run(); eng.sync(); t_run = time.perf_counter() - t0
t0 = time.perf_counter(); costs = plan.fetch(); t_d2h = time.perf_counter() - t0
plan.close()
t0 = time.perf_counter(); c2 = eng.score_batch(jobs, cb.events); t_oneshot = time.perf_counter() - t0
t0 = time.perf_counter(); b = ra.Batch(eng, opt, cb); t_batch_create_first = time.perf_counter() - t0
b.close() if hasattr(b, "close") else None
t0 = time.perf_counter(); b = ra.Batch(eng, opt, cb); t_batch_create = time.perf_counter() - t0   # steady state: code loaded, workspace allocated
b.run(); eng.sync()
t0 = time.perf_counter(); b.run(); eng.sync(); t_batch_run = time.perf_counter() - t0
eng.set_option("device_plan", 0)
t0 = time.perf_counter(); b2 = ra.Batch(eng, opt, cb); t_batch_create_host = time.perf_counter() - t0
from rawalign_amd.dtw import plan_dry_run
_, host_tiles = plan_dry_run(jobs, len(cb.events), 2 * ((4_600_000 + 3) & ~3) + 64, options={"verify": 0})
print(json.dumps({"host_planner_tiles": host_tiles, "reads": reads, "jobs": int(nj.value), "cells": info["cells"], "events_MB": cb.events.nbytes / 1e6,
                  "build_jobs_s": t_build, "h2d_events_s": t_h2d, "plan_create_s": t_plan, "run_s": t_run, "fetch_costs_s": t_d2h,
                  "score_batch_oneshot_s": t_oneshot, "GCUPS_pcie_inclusive": info["cells"] / t_oneshot / 1e9,
                  "batch_create_s": t_batch_create, "batch_create_first_s": t_batch_create_first,
                  "batch_create_host_planner_s": t_batch_create_host, "batch_run_s": t_batch_run,
                  "GCUPS_batch_create_plus_run": info["cells"] / (t_batch_create + t_batch_run) / 1e9}))
