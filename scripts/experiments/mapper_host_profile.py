#!/usr/bin/env python3
"""The mapper's host side alone (no GPU): the bench's mapper block with the DTW block on the host cores (oracle/_ref's ref_scorer), every phase's
time against the wall time of the round calls.  python scripts/experiments/mapper_host_profile.py [reads] [threads]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rawalign_amd as ra  # noqa: E402
from rawalign_amd import mapper, synth  # noqa: E402
from rawalign_amd.mapping import StopOpt  # noqa: E402
from oracle.loader import RefDTW  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ref = synth.make_reference([4_600_000], seed=20231005)
sc = synth.make_seed_chunks(ref, n_reads, seed=20231005 + 17)
opt = ra.MapOpt()
names = [f"seq{s}" for s in range(ref.n_seq)]
lens = [len(x) for x in ref.forward]
slot = int(sc["n_ev"].max()) + 8
first, nch = sc["chunk_first"], sc["n_chunks"]
ev_off, hit_off = sc["ev_off"].astype(np.int64), sc["hit_off"].astype(np.int64)
rl = RefDTW().lib


class ScorerCtx(C.Structure):
    _fields_ = [("fwd", C.c_void_p), ("rev", C.c_void_p), ("border_constraint", C.c_int), ("fill_method", C.c_int), ("band_radius_frac", C.c_float),
                ("match_bonus", C.c_float), ("min_score", C.c_float), ("fused_score", C.c_int), ("threads", C.c_int), ("dtw_calls", C.c_uint64)]


fwd = (C.c_void_p * ref.n_seq)(*[x.ctypes.data for x in ref.forward])
rev = (C.c_void_p * ref.n_seq)(*[x.ctypes.data for x in ref.reverse])
sctx = ScorerCtx(C.cast(fwd, C.c_void_p), C.cast(rev, C.c_void_p), opt.dtw_border_constraint, opt.dtw_fill_method, opt.dtw_band_radius_frac, opt.dtw_match_bonus,
                 opt.dtw_min_score, int(opt.fused_score), threads, 0)
fn = C.cast(rl.ref_scorer, C.c_void_p)
for stop, label in ((StopOpt(), "stop rule"), (StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6), "all chunks")):
    for rep in range(2):
        cm = mapper.CMapper(None, opt, stop, names, lens, slot_events=slot, max_reads=n_reads, carry=False, threads=threads)
        cm.set_scorer_c(fn, C.cast(C.pointer(sctx), C.c_void_p))
        reads = np.arange(n_reads)
        ids = np.array([cm.add_read("read_%d" % r, int(sc["qlen"][r]), int(nch[r])) for r in reads], np.uint32)
        done = np.zeros(n_reads, np.int64)
        active = np.ones(n_reads, bool)
        wall, rounds = 0.0, []
        while active.any():
            sel = np.nonzero(active)[0]
            ci = first[reads[sel]] + done[sel]
            ecnt, hcnt = ev_off[ci + 1] - ev_off[ci], hit_off[ci + 1] - hit_off[ci]
            eo = np.concatenate([[0], np.cumsum(ecnt)]).astype(np.uint64)
            ho = np.concatenate([[0], np.cumsum(hcnt)]).astype(np.uint64)
            eidx = np.repeat(ev_off[ci], ecnt) + (np.arange(int(eo[-1])) - np.repeat(eo[:-1].astype(np.int64), ecnt))
            hidx = np.repeat(hit_off[ci], hcnt) + (np.arange(int(ho[-1])) - np.repeat(ho[:-1].astype(np.int64), hcnt))
            ev = np.ascontiguousarray(sc["events"][eidx]) if len(eidx) else np.zeros(1, np.float32)
            hits = np.ascontiguousarray(sc["hits"][hidx]) if len(hidx) else np.zeros(1, sc["hits"].dtype)
            t0 = time.perf_counter()
            cm.round_arrays(np.ascontiguousarray(ids[sel]), eo, ev, ho, hits)
            dt = time.perf_counter() - t0
            wall += dt
            rounds.append((len(sel), round(dt * 1e3, 2)))
            done[sel] += 1
            for k in sel:
                fin, _ = cm.state(int(ids[k]))
                if fin or done[k] >= nch[reads[k]]:
                    active[k] = False
        tm = cm.timing()
        phases = {k: round(v, 2) for k, v in tm.items() if k.endswith("_ms")}
        print(label, "rep", rep, "wall %.2f ms" % (wall * 1e3), "phases", phases, "sum %.2f" % sum(phases.values()), "rounds", rounds[:4], flush=True)
        cm.close()
