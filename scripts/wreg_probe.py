"""Latency of single wide-band jobs through the job-list path's wave-per-job kernel, alone on the chip (GPU box).
Usage: python scripts/wreg_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import rawalign_amd as ra  # noqa: E402

rng = np.random.default_rng(1)
ref = rng.standard_normal(8192).astype(np.float32)
ev = rng.standard_normal(8192).astype(np.float32)
eng = ra.Engine(0)
eng.upload_reference([ref], [ref[::-1].copy()])
eng.upload_events(ev)
for n, m, r0, copies in [(952, 801, 85, 1), (524, 427, 45, 1), (524, 427, 45, 256), (524, 427, 45, 4096), (150, 120, 12, 1), (150, 120, 12, 4096),
                         (90, 70, 5, 1), (90, 70, 5, 8192)]:
    jobs = np.zeros(copies, ra.JOB_DTYPE)
    jobs["n"], jobs["m"], jobs["band_radius"] = n, m, r0
    p = eng.plan(jobs)
    p.run()
    t = [p.run_timed() for _ in range(5)]
    best = min(sum(x[2] for x in run) for run in t)
    kinds = [(eng.KIND_NAMES.get(k, k), par) for k, par, _ in t[0]]
    print("n %4d m %4d r0 %3d x%5d  %8.1f us  %7.1f ns/column  %s" % (n, m, r0, copies, best * 1e3, best * 1e6 / n, kinds), flush=True)
    p.close()
