"""Timing probe for the cross-round cache: the last of four chunk rounds of a bench-like batch, submitted from scratch and
with the round before to take costs over from; the planning launches (k_scan + k_side + k_plan) and the DTW launches alone on the chip.
Usage (GPU box): python scripts/rounds_probe.py [n_reads]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import rawalign_amd as ra  # noqa: E402
from rawalign_amd import synth  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ref = synth.make_reference([4_600_000], seed=20231007)
eng = ra.Engine(0)
eng.upload_reference(ref.forward, ref.reverse)
offs = {(0, st): eng.reference_offset(0, st) for st in (0, 1)}
cb, info = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=n_reads), seed=20231007 + 7919)
rounds = synth.make_rounds(cb, info, 4)
eng.upload_events(cb.events)
eng.set_option("time_plan", 1)
lib, copt = eng.lib, ra.MapOpt().c_struct()
vp = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
arrs = [[np.ascontiguousarray(r.chain_off, np.uint64), np.ascontiguousarray(r.anchor_off, np.uint64), np.ascontiguousarray(r.anchors),
         np.ascontiguousarray(r.ref_base, np.uint64), np.ascontiguousarray(r.read_base, np.uint32)] for r in rounds]
ident = np.arange(cb.n_reads, dtype=np.uint64)
CARRY_DTYPE = np.dtype([("prev_src", "<u8"), ("parts", "<u4"), ("flags", "<u4"), ("start_t", "<u4"), ("start_q", "<u4")])  # rawdtw_carry_t
keep_alive = []


def submit(k, prev):
    h = C.c_void_p()
    a = arrs[k]
    if prev is None:
        eng._check(lib.rawdtw_batch_submit(eng._ctx, C.byref(copt), cb.n_reads, vp(a[0]), vp(a[1]), vp(a[2]), vp(a[3]), vp(a[4]), C.byref(h)))
    else:
        p = arrs[k - 1]
        carry = np.zeros(cb.n_chains, CARRY_DTYPE); new_off = np.zeros(cb.n_chains + 1, np.uint64); new_anchors = np.zeros(len(a[2]) + 1, a[2].dtype)
        eng._check(lib.rawdtw_round_match_chains(cb.n_reads, vp(a[0]), vp(a[1]), vp(a[2]), vp(a[3]), vp(a[4]), vp(ident), vp(p[0]), vp(p[1]), vp(p[2]), vp(p[3]),
                                                 vp(p[4]), vp(carry), vp(new_off), vp(new_anchors)))
        keep_alive.append((carry, new_off, new_anchors))
        eng._check(lib.rawdtw_batch_submit_carry(eng._ctx, C.byref(copt), cb.n_reads, vp(a[0]), vp(a[1]), vp(a[2]), vp(new_off), vp(new_anchors), vp(a[3]), vp(a[4]),
                                                 prev, vp(carry), C.byref(h)))
    eng.sync()
    return h


for mode in ("scratch", "carried"):
    prev = submit(2, None)
    for rep in range(2):
        h = submit(3, prev if mode == "carried" else None)
        pm = C.c_float()
        lib.rawdtw_batch_plan_ms(eng._ctx, h, C.byref(pm))
        ms = np.zeros(8, np.float32); kind = np.zeros(8, np.uint32); nl = C.c_uint32()
        eng._check(lib.rawdtw_batch_run_reps(eng._ctx, h, 10, vp(ms), vp(kind), 8, C.byref(nl)))
        sc, ru = C.c_uint64(), C.c_uint64()
        lib.rawdtw_batch_round_stats(eng._ctx, h, C.byref(sc), C.byref(ru))
        if rep:
            print("%-8s parts scored %d reused %d   planning (k_scan + k_side + k_plan) %.4f ms   k_wide %.4f ms  k_runs %.4f ms  fold + select %.4f"
                  % (mode, sc.value, ru.value, pm.value, ms[0], ms[1], ms[2]), flush=True)
        lib.rawdtw_batch_destroy(h)
    lib.rawdtw_batch_destroy(prev)
