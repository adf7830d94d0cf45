#!/usr/bin/env python3
"""Thread scaling of the CPU baseline (the reference's dtw.cpp over the bench job list)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rawalign_amd as ra
from rawalign_amd import synth
from rawalign_amd._lib import AlignOpt, load_library
from oracle.loader import RefDTW, Oracle
ref = synth.make_reference([4_600_000], seed=20231007)
n = len(ref.forward[0]); pad = (n + 3) & ~3
offs = {(0, 1): 0, (0, 0): pad}
cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=4096), seed=20231007 + 7919)
lib = load_library(); opt = AlignOpt(1, 1, 0.10, 0.4, 20.0, 1)
p = lambda a: a.ctypes.data_as(C.c_void_p)
job_off = np.zeros(cb.n_chains + 1, np.uint64); nj = C.c_uint64()
lib.rawdtw_batch_build_jobs(C.byref(opt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base), p(cb.read_base), p(job_off), None, 0, C.byref(nj))
jobs = np.zeros(nj.value, ra.JOB_DTYPE)
lib.rawdtw_batch_build_jobs(C.byref(opt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base), p(cb.read_base), p(job_off), p(jobs), len(jobs), C.byref(nj))
arena = np.zeros(2 * pad, np.float32); arena[:n] = ref.forward[0]; arena[pad:pad + n] = ref.reverse[0]
print("jobs", len(jobs), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for name, impl in (("reference", RefDTW()), ("port", Oracle())):
    for th in (1, 2, 4, 8, 16, 32):
        impl.batch_costs(jobs[:100000], cb.events, arena, th)
        t0 = time.perf_counter(); impl.batch_costs(jobs, cb.events, arena, th); dt = time.perf_counter() - t0
        print(name, "threads", th, f"{dt*1e3:.1f} ms", f"{len(jobs)/dt/1e6:.1f} Mjobs/s")
