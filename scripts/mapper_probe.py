#!/usr/bin/env python3
"""The mapper block of bench.py for chosen (groups, threads, carry) combinations: wall time of the round calls and the mapper's own phase times, the
first round apart.  python scripts/mapper_probe.py [reads] [combo ...]   combo = groups,threads,carry[,all][,dev]  (all: every read through all of its chunks; dev: chaining on the device)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rawalign_amd as ra  # noqa: E402
from rawalign_amd import mapper, synth  # noqa: E402
from rawalign_amd.mapping import StopOpt  # noqa: E402

if os.environ.get("PROBE_TORCH"):  # (as bench.py's process: torch's runtime, its threads and a few launches before the mapper)
    import torch
    x = torch.randn(1 << 20, device="cuda:0")
    for _ in range(10):
        x = x * 1.0001
    torch.cuda.synchronize()
    y = torch.randn(2000, 2000) @ torch.randn(2000, 2000)
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
combos = sys.argv[2:] or ["1,16,0", "1,16,0,dev", "2,16,0,dev", "1,4,0,dev", "1,16,0,all", "1,16,0,all,dev"]
ref = synth.make_reference([4_600_000], seed=20231005)
sc = synth.make_seed_chunks(ref, n_reads, seed=20231005 + 17)
opt = ra.MapOpt()
names = [f"seq{s}" for s in range(ref.n_seq)]
lens = [len(x) for x in ref.forward]
slot = int(sc["n_ev"].max()) + 8
first, nch = sc["chunk_first"], sc["n_chunks"]
ev_off, hit_off = sc["ev_off"].astype(np.int64), sc["hit_off"].astype(np.int64)
never = StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6)


import ctypes as C  # noqa: E402



def pinned(n, dtype):
    from rawalign_amd._lib import load_library
    lib = load_library()
    p = C.c_void_p()
    nbytes = max(int(n) * np.dtype(dtype).itemsize, 8)
    assert lib.rawdtw_host_alloc(nbytes, C.byref(p)) == 0
    return np.frombuffer((C.c_char * nbytes).from_address(p.value), dtype=dtype, count=int(n))


ev_pin = pinned(int(np.diff(ev_off).max()) * n_reads + 1, np.float32) if os.environ.get("PROBE_PAGEABLE_EVENTS") is None else None


def one_pass(cm, reads):
    ids = np.array([cm.add_read("read_%d" % r, int(sc["qlen"][r]), int(nch[r])) for r in reads], np.uint32)
    done = np.zeros(len(reads), np.int64)
    active = np.ones(len(reads), bool)
    rounds = []
    while active.any():
        sel = np.nonzero(active)[0]
        ci = first[reads[sel]] + done[sel]
        ecnt, hcnt = ev_off[ci + 1] - ev_off[ci], hit_off[ci + 1] - hit_off[ci]
        eo = np.concatenate([[0], np.cumsum(ecnt)]).astype(np.uint64)
        ho = np.concatenate([[0], np.cumsum(hcnt)]).astype(np.uint64)
        eidx = np.repeat(ev_off[ci], ecnt) + (np.arange(int(eo[-1])) - np.repeat(eo[:-1].astype(np.int64), ecnt))
        hidx = np.repeat(hit_off[ci], hcnt) + (np.arange(int(ho[-1])) - np.repeat(ho[:-1].astype(np.int64), hcnt))
        if len(eidx) and ev_pin is not None:  # (page-locked, as a host that allocates its event buffers with rawdtw_host_alloc has them)
            ev = ev_pin[:len(eidx)]
            np.take(sc["events"], eidx, out=ev)
        else:
            ev = np.ascontiguousarray(sc["events"][eidx]) if len(eidx) else np.zeros(1, np.float32)
        hits = np.ascontiguousarray(sc["hits"][hidx]) if len(hidx) else np.zeros(1, sc["hits"].dtype)
        tm0 = cm.timing()
        t0 = time.perf_counter()
        cm.round_arrays(np.ascontiguousarray(ids[sel]), eo, ev, ho, hits)
        dt = time.perf_counter() - t0
        tm = cm.timing()
        rounds.append((len(sel), round(dt * 1e3, 2), {k[:-3]: round(tm[k] - tm0[k], 2) for k in tm if k.endswith("_ms")}))
        done[sel] += 1
        for k in sel:
            fin, _ = cm.state(int(ids[k]))
            if fin or done[k] >= nch[reads[k]]:
                active[k] = False
    return ids, rounds


for combo in combos:
    f = combo.split(",")
    groups, threads, carry = int(f[0]), int(f[1]), int(f[2])
    stop = never if len(f) > 3 and f[3] == "all" else StopOpt()
    dev_chain = "dev" in f[3:]
    eng = ra.Engine(0)
    eng.upload_reference(ref.forward, ref.reverse)
    cm = mapper.CMapper(eng, opt, stop, names, lens, slot_events=slot, max_reads=n_reads, carry=bool(carry), threads=threads, groups=groups, device_chain=dev_chain)
    reads = np.arange(n_reads)
    ids, _ = one_pass(cm, reads)
    for i in ids:
        cm.release_read(int(i))
    for rep in range(2):
        ids, rounds = one_pass(cm, reads)
        if os.environ.get("PROBE_FINISH"):
            cm.finish()
            [cm.paf(int(i)) for i in ids]
        print(combo, "rep", rep, "wall %.2f ms" % sum(r[1] for r in rounds), "reads/s %d" % (n_reads / sum(r[1] for r in rounds) * 1e3), flush=True)
        for r in rounds[:3]:
            print("    ", r, flush=True)
        for i in ids:
            cm.release_read(int(i))
    cm.close()
    eng.close()
