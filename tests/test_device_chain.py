"""The anchor sort and the chaining DP on the device (rawdtw_chain_round, rawalign_amd/csrc/rawdtw_chain.hip) against the host
restatement of rmap.cpp:396-401, 430-507, 130-173 and 512 (rawdtw_chain_anchors + rawdtw_sort_by_chaining_score, themselves checked against the
oracle in tests/test_chaining.py): per read the same chains in the same order -- scores bit for bit, positions, every anchor -- and the batch
arrays the DTW takes in device memory."""
import ctypes as C

import numpy as np
import pytest

import rawalign_amd as ra
from rawalign_amd import mapping as M
from rawalign_amd.dtw import ANCHOR_DTYPE

SEED_DTYPE = np.dtype([("key", "<u4"), ("target_position", "<u4"), ("query_position", "<u4")])  # rawdtw_seed_t
REC_DTYPE = np.dtype([("chaining_score", "<f4"), ("key", "<u4"), ("start_position", "<u4"), ("end_position", "<u4"), ("n_anchors", "<u4")])  # rawdtw_chain_rec_t


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def host_chains(lib, copt, seeds):
    """a read's chains as the mapper's host phase makes them: lists by key, each sorted, chained with the running maximum, then the order"""
    order = np.lexsort((seeds["query_position"], seeds["target_position"], seeds["key"]))
    s = seeds[order]
    chains, maxs = [], 0.0
    for key in np.unique(s["key"]):
        g = s[s["key"] == key]
        a = np.zeros(len(g), ANCHOR_DTYPE)
        a["target_position"], a["query_position"] = g["target_position"], g["query_position"]
        cs, maxs = M.chain_anchors(a, copt, maxs, int(key) >> 1, int(key) & 1)
        for ch in cs:
            chains.append((np.float32(ch.chaining_score), int(key), ch.start_position, ch.end_position, ch.anchors))
    if chains:
        sc = np.array([c[0] for c in chains], np.float32)
        perm = np.zeros(len(chains), np.uint32)
        assert lib.rawdtw_sort_by_chaining_score(vp(sc), len(sc), vp(perm)) == 0
        chains = [chains[p] for p in perm]
    return chains


def device_round(eng, copt, per_read, n_keys=8):
    lib = eng.lib
    n = len(per_read)
    seed_off = np.zeros(n + 1, np.uint64)
    for r, s in enumerate(per_read):
        seed_off[r + 1] = seed_off[r] + len(s)
    seeds = np.concatenate(list(per_read) + [np.zeros(1, SEED_DTYPE)])
    read_base = (np.arange(n, dtype=np.uint32) * 1000).astype(np.uint32)
    key_base = (np.arange(n_keys, dtype=np.uint64) * 100000 + 7).astype(np.uint64)
    cap = n * 32
    chain_off = np.zeros(n + 1, np.uint64)
    anchor_off = np.zeros(cap + 1, np.uint64)
    recs = np.zeros(cap, REC_DTYPE)
    anchors = np.zeros(int(seed_off[-1]) + 1, ANCHOR_DTYPE)
    d_a, d_rb, d_qb = C.c_void_p(), C.c_void_p(), C.c_void_p()
    st = lib.rawdtw_chain_round(eng._ctx, C.byref(copt), n, vp(seed_off), vp(seeds), vp(read_base), n_keys, vp(key_base), vp(chain_off), vp(anchor_off), vp(recs),
                                cap, vp(anchors), C.byref(d_a), C.byref(d_rb), C.byref(d_qb))
    return st, chain_off, anchor_off, recs, anchors, (d_a, d_rb, d_qb), read_base, key_base


def random_read(rng, n, n_keys, span, dup=0.05, lines=3):
    """seeds on a few diagonals (true chains), noise, repeated targets / queries and exact duplicates -- unsorted"""
    s = np.zeros(n, SEED_DTYPE)
    s["key"] = rng.integers(0, n_keys, n)
    kind = rng.random(n)
    t = rng.integers(0, span, n)
    q = rng.integers(0, max(span // 8, 4), n)
    for ln in range(lines):
        on = (kind > 0.25) & (rng.integers(0, lines, n) == ln)
        q0, t0, slope = int(rng.integers(0, 50)), int(rng.integers(0, span)), float(rng.uniform(0.8, 1.3))
        q[on] = q0 + np.sort(rng.integers(0, max(span // 8, 4), int(on.sum())))
        t[on] = t0 + ((q[on] - q0) * slope).astype(np.int64) + rng.integers(-3, 4, int(on.sum()))
    s["target_position"], s["query_position"] = np.clip(t, 0, 2 ** 30), q
    d = rng.random(n) < dup
    if n > 1:
        src = rng.integers(0, n, n)
        s[d] = s[src[d]]
    return s


def compare(lib, copt, per_read, out):
    st, chain_off, anchor_off, recs, anchors, _, _, _ = out
    assert st == 0
    for r, seeds in enumerate(per_read):
        want = host_chains(lib, copt, seeds)
        c0, c1 = int(chain_off[r]), int(chain_off[r + 1])
        assert c1 - c0 == len(want), (r, c1 - c0, len(want))
        for i, (score, key, start, end, an) in enumerate(want):
            rec = recs[c0 + i]
            assert np.float32(rec["chaining_score"]).view(np.uint32) == np.float32(score).view(np.uint32), (r, i, rec, score)
            assert (int(rec["key"]), int(rec["start_position"]), int(rec["end_position"]), int(rec["n_anchors"])) == (key, start, end, len(an)), (r, i, rec)
            got = anchors[int(anchor_off[c0 + i]):int(anchor_off[c0 + i + 1])]
            assert len(got) == len(an) and (got["target_position"] == an["target_position"]).all() and (got["query_position"] == an["query_position"]).all(), (r, i)
    assert int(anchor_off[int(chain_off[-1])]) == sum(int(x) for x in recs["n_anchors"][:int(chain_off[-1])])


@pytest.mark.gpu
def test_chains_of_random_reads_equal_the_host_restatement():
    rng = np.random.default_rng(11)
    eng = ra.Engine(0)
    lib = eng.lib
    copt = M.default_chain_opt(6)
    per_read = [np.zeros(0, SEED_DTYPE), random_read(rng, 1, 2, 100), random_read(rng, 2, 1, 100), random_read(rng, 3, 2, 50)]
    for n in (10, 40, 63, 64, 65, 100, 130, 200, 257, 300, 511, 512, 513, 700, 1000):
        for nk in (1, 2, 5):
            per_read.append(random_read(rng, n, nk, int(rng.integers(200, 20000))))
    for _ in range(60):
        per_read.append(random_read(rng, int(rng.integers(5, 400)), 2, int(rng.integers(300, 8000)), dup=0.15))
    # long skip-free runs (many candidates per anchor: several rounds of 64), and dense ties (all on one diagonal with unit steps)
    s = np.zeros(600, SEED_DTYPE)
    s["target_position"] = 1000 + np.arange(600) * 3
    s["query_position"] = np.arange(600) * 3
    per_read.append(s[rng.permutation(600)])
    s = np.zeros(300, SEED_DTYPE)
    s["target_position"] = 50 + np.arange(300) // 2
    s["query_position"] = np.arange(300) // 3
    per_read.append(s[rng.permutation(300)])
    compare(lib, copt, per_read, device_round(eng, copt, per_read))
    # other options: narrow band, few skips, score filtering off, one chain a list, longer minimum
    for copt2 in (M.ChainOpt(2000, 5000, 20, 25, 2, 3, 10.0, 6, 0), M.ChainOpt(2000, 5000, 5000, 3, 2, 3, 10.0, 6, 0), M.ChainOpt(500, 300, 5000, 25, 2, 5, 10.0, 6, 1),
                  M.ChainOpt(2000, 5000, 5000, 25, 4, 1, 30.0, 8, 0)):
        compare(lib, copt2, per_read[:70], device_round(eng, copt2, per_read[:70]))
    eng.close()


@pytest.mark.gpu
def test_round_arrays_in_device_memory_and_what_the_device_declines():
    rng = np.random.default_rng(12)
    eng = ra.Engine(0)
    lib = eng.lib
    copt = M.default_chain_opt(6)
    per_read = [random_read(rng, int(rng.integers(20, 300)), 4, 5000) for _ in range(50)]
    st, chain_off, anchor_off, recs, anchors, dev, read_base, key_base = device_round(eng, copt, per_read)
    assert st == 0
    nc, na = int(chain_off[-1]), int(anchor_off[int(chain_off[-1])])
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def dev_array(ptr, dtype, count):  # the arrays rawdtw_batch_submit_device takes, brought home
        out = np.zeros(count, dtype)
        assert hip.hipMemcpy(vp(out), ptr, out.nbytes, 2) == 0
        return out
    d_anch = dev_array(dev[0], ANCHOR_DTYPE, na)
    assert (d_anch["target_position"] == anchors[:na]["target_position"]).all() and (d_anch["query_position"] == anchors[:na]["query_position"]).all()
    d_rb = dev_array(dev[1], np.uint64, nc)
    d_qb = dev_array(dev[2], np.uint32, nc)
    owner = np.repeat(np.arange(len(per_read)), np.diff(chain_off).astype(np.int64))
    assert (d_qb == read_base[owner]).all() and (d_rb == key_base[recs["key"][:nc]]).all()
    # declined: more seeds than the wave's LDS holds
    big = [random_read(rng, 2049, 2, 50000)] + per_read[:3]
    assert device_round(eng, copt, big)[0] != 0
    # ... and more than 16 chains with equal scores: 20 lists with the same little chain each
    one = np.zeros(4, SEED_DTYPE)
    one["target_position"] = [100, 110, 120, 130]
    one["query_position"] = [5, 15, 25, 35]
    many = []
    for k in range(20):
        x = one.copy()
        x["key"] = k
        many.append(x)
    tied = [np.concatenate(many)]
    st = device_round(eng, copt, tied, n_keys=32)[0]
    assert st != 0
    # a seed on a key the caller gave no base for: refused (4 keys declared, a chain on key 5)
    bad = [many[5].copy()]
    assert device_round(eng, copt, bad, n_keys=4)[0] == 1  # RAWDTW_ERR_INVALID
    # 16 such lists are within what an insertion sort orders: the same order as the host's
    ok16 = [np.concatenate(many[:16])]
    compare(lib, copt, ok16, device_round(eng, copt, ok16, n_keys=32))
    eng.close()


@pytest.mark.gpu
def test_the_dtw_batch_straight_from_the_devices_chains_equals_the_batch_from_host_arrays():
    """rawdtw_chain_round -> rawdtw_batch_submit_device (anchors, ref_base, read_base used where the chaining left them, in device memory) against
    rawdtw_batch_submit of the same chains from the host copies: every chain's score and keep flag bit for bit -- sparse + banded (the device-planned
    path) and global + full (the job-list path, which brings the device arrays home first)."""
    from rawalign_amd import mapper, synth

    ref = synth.make_reference([150_000], seed=31)
    n = 300
    seeds = mapper.SyntheticSeeds(ref, n, seed=9, max_chunks=2)
    eng = ra.Engine(0)
    lib = eng.lib
    eng.upload_reference(ref.forward, ref.reverse)
    copt = M.default_chain_opt(6)
    # one round: every read's first chunk; the reads' events side by side in the arena
    evs, per_read, read_base = [], [], np.zeros(n, np.uint32)
    at = 0
    for r in range(n):
        ev, hits = seeds.chunk(r, 0)
        read_base[r] = at
        at += len(ev)
        evs.append(np.asarray(ev, np.float32))
        s = np.zeros(len(hits), SEED_DTYPE)
        for k, (sq, st, t, q) in enumerate(hits):
            s[k] = (sq * 2 + (1 if st else 0), t, q)
        per_read.append(s)
    events = np.concatenate(evs)
    eng.upload_events(events)
    key_base = np.array([eng.reference_offset(0, 0), eng.reference_offset(0, 1)], np.uint64)
    seed_off = np.zeros(n + 1, np.uint64)
    for r, s in enumerate(per_read):
        seed_off[r + 1] = seed_off[r] + len(s)
    allseeds = np.concatenate(per_read + [np.zeros(1, SEED_DTYPE)])
    cap = n * 32
    chain_off, anchor_off, recs = np.zeros(n + 1, np.uint64), np.zeros(cap + 1, np.uint64), np.zeros(cap, REC_DTYPE)
    anchors = np.zeros(int(seed_off[-1]) + 1, ANCHOR_DTYPE)
    d_a, d_rb, d_qb = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.rawdtw_chain_round(eng._ctx, C.byref(copt), n, vp(seed_off), vp(allseeds), vp(read_base), 2, vp(key_base), vp(chain_off), vp(anchor_off), vp(recs), cap,
                                  vp(anchors), C.byref(d_a), C.byref(d_rb), C.byref(d_qb)) == 0
    nc = int(chain_off[-1])
    assert nc > n // 2
    h_ref_base = key_base[recs["key"][:nc]].astype(np.uint64)
    h_read_base = np.repeat(read_base, np.diff(chain_off).astype(np.int64)).astype(np.uint32)
    for opt in (ra.MapOpt(), ra.MapOpt(dtw_border_constraint=0, dtw_fill_method=0)):
        co = opt.c_struct()
        out = {}
        for dev in (1, 0):
            h = C.c_void_p()
            if dev:
                st = lib.rawdtw_batch_submit_device(eng._ctx, C.byref(co), n, vp(chain_off), vp(anchor_off), d_a, d_rb, d_qb, C.byref(h))
            else:
                st = lib.rawdtw_batch_submit(eng._ctx, C.byref(co), n, vp(chain_off), vp(anchor_off), vp(anchors), vp(h_ref_base), vp(h_read_base), C.byref(h))
            assert st == 0, lib.rawdtw_last_error(eng._ctx)
            score, keep = np.zeros(nc + 1, np.float32), np.zeros(nc + 1, np.uint8)
            assert lib.rawdtw_batch_fetch_destroy(eng._ctx, h, vp(score), vp(keep)) == 0
            out[dev] = (score[:nc].copy(), keep[:nc].copy())
        assert (out[0][0].view(np.uint32) == out[1][0].view(np.uint32)).all() and (out[0][1] == out[1][1]).all()
        assert out[1][1].sum() > 0
    eng.close()


@pytest.mark.gpu
def test_device_chains_against_the_plain_python_restatement():
    """The device against the plain-Python restatement of rmap.cpp:430-507 / 130-173 that tests/test_mapping_host.py pins the host function with
    (test infrastructure, no product code in it): one list a read, so that the running maximum starts at 0 as in py_chain."""
    from tests.test_mapping_host import f32, py_chain

    rng = np.random.default_rng(21)
    eng = ra.Engine(0)
    copt = M.default_chain_opt(6)
    per_read = [random_read(rng, int(rng.integers(2, 220)), 1, int(rng.integers(300, 6000)), dup=0.1) for _ in range(40)]
    st, chain_off, anchor_off, recs, anchors, _, _, _ = device_round(eng, copt, per_read)
    assert st == 0
    for r, s in enumerate(per_read):
        a = np.zeros(len(s), ANCHOR_DTYPE)
        a["target_position"], a["query_position"] = s["target_position"], s["query_position"]
        a = np.sort(a, order=["target_position", "query_position"])
        want, _ = py_chain(a)
        want.sort(key=lambda c: -c[0])  # (stable: the evaluation order of rmap.cpp:512 for fewer than 17 chains)
        c0, c1 = int(chain_off[r]), int(chain_off[r + 1])
        assert c1 - c0 == len(want), (r, c1 - c0, len(want))
        for i, (score, idx) in enumerate(want):
            rec = recs[c0 + i]
            assert f32(rec["chaining_score"]) == f32(score) and int(rec["n_anchors"]) == len(idx)
            got = anchors[int(anchor_off[c0 + i]):int(anchor_off[c0 + i + 1])]
            assert [int(x) for x in got["target_position"]] == [int(a[k]["target_position"]) for k in idx]
            assert [int(x) for x in got["query_position"]] == [int(a[k]["query_position"]) for k in idx]
    eng.close()
