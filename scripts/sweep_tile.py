#!/usr/bin/env python3
"""Parameter sweep of the tile kernel on the bench workload (isolated kernel time)."""
import sys, os, json, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rawalign_amd as ra
from rawalign_amd import synth

ref = synth.make_reference([4_600_000], seed=20231007)
eng = ra.Engine(0)
eng.upload_reference(ref.forward, ref.reverse)
offs = {(0, st): eng.reference_offset(0, st) for st in (0, 1)}
cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=16384), seed=20231007 + 7919)
eng.upload_events(cb.events)
eng.set_option("serial_launches", 1)
combos = [(5120, 1024, 3, 73, mm) for mm in (8, 4, 0)] + [(4096, 768, 3, 73, 8), (5120, 1536, 3, 73, 8), (7168, 1536, 3, 73, 8)]
for lds, mj, lmr, ln, mm in combos:
    eng.set_option("lane_max_n", ln); eng.set_option("micro_max_n", mm)
    eng.set_option("tile_lds_floats", lds); eng.set_option("tile_max_jobs", mj); eng.set_option("lane_max_radius", lmr)
    b = ra.Batch(eng, ra.MapOpt(), cb)
    b.run_reps(2, timed=False)
    L = b.run_reps(8, timed=True)
    st = b.launch_stats(with_cells=False)
    print(json.dumps({"lds_floats": lds, "max_jobs": mj, "lane_max_r": lmr, "lane_max_n": ln, "micro": mm,
                      "launches": [(ra.Engine.KIND_NAMES.get(k), p, round(ms, 4), st[i]["n_jobs"]) for i, (k, p, ms) in enumerate(L)]}))
    b.close()
