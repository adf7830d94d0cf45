"""Timing probe for the sync-free batch path: one bench-like batch, k_runs alone under the `stream_debug` masks and a
few tile sizes (GPU box).  Usage: python scripts/stream_probe.py [n_reads]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (initialise torch's HIP runtime first)

import rawalign_amd as ra  # noqa: E402
from rawalign_amd import synth  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
configs = sys.argv[2:] or [""]
ref = synth.make_reference([int(os.environ.get("RAWDTW_PROBE_GENOME", 4_600_000))], seed=20231007)  # (RAWDTW_PROBE_GENOME: a reference beyond the Infinity Cache)
eng0 = ra.Engine(0)
eng0.upload_reference(ref.forward, ref.reverse)
offs = {(0, st): eng0.reference_offset(0, st) for st in (0, 1)}
cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=n_reads), seed=20231007 + 7919)
for cfg in configs:
    eng = ra.Engine(0)
    eng._check(eng.lib.rawdtw_share_reference(eng._ctx, eng0._ctx))
    for item in filter(None, cfg.split(",")):
        k, v = item.split("=")
        eng.set_option(k, int(v))
    eng.set_option("time_plan", 1)
    eng.upload_events(cb.events)
    t0 = time.perf_counter()
    b = ra.Batch(eng, ra.MapOpt(), cb)
    t_create = (time.perf_counter() - t0) * 1e3
    b.run_reps(3, timed=False)
    ms = b.run_reps(20, timed=True)
    pm = C.c_float()
    eng.lib.rawdtw_batch_plan_ms(eng._ctx, b._h, C.byref(pm))
    cnt = (C.c_uint64 * 64)()
    ncnt = C.c_uint32()
    eng.lib.rawdtw_batch_stream_counters(eng._ctx, b._h, cnt, 64, C.byref(ncnt))
    if cfg == configs[0]:
        ix = lambda n: eng.lib.rawdtw_batch_stream_counter_index(n)  # noqa: E731
        print("counters: side list %d  classes %s" % (cnt[ix(b"side_jobs")], list(cnt[ix(b"class0"):ix(b"class0") + 21])), flush=True)
    # launches: k_wide, k_runs, fold + select (one launch), -
    print("%-28s create %.3f ms  plan(gpu) %.3f ms  launches %s" % (cfg or "default", t_create, pm.value, ["%.4f" % m[2] for m in ms]), flush=True)
    s0 = eng.lib.rawdtw_batch_stream_counter_index(b"stamp0")
    if any(cnt[s0:s0 + 10]):  # "stream_debug" 256: cycles per phase of k_runs, summed over waves and runs
        tot = float(sum(cnt[s0:s0 + 10]))
        names = ["entry", "stage issue", "stage wait", "B1", "DP + next records", "ticket + wait", "B2"]
        print("   phase shares: " + "  ".join("%s %.1f%%" % (n, 100.0 * c / tot) for n, c in zip(names, cnt[s0:s0 + 10])), flush=True)
    b.close()
    eng.close()
