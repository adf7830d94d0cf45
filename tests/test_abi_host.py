"""CPU-side tests: the C-ABI library loads and exports every symbol include/rawdtw.h declares, and the
host mirror of align_chain / gen_chains' DTW block (pure host code) agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rawalign_amd as ra
from rawalign_amd._lib import SYMBOLS, AlignOpt, load_library
from rawalign_amd.dtw import ANCHOR_DTYPE, JOB_DTYPE
from oracle.loader import OrcOpt, OrcStats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rawdtw.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rawdtw_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = load_library()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rawdtw.h but not exported"
        assert n in SYMBOLS, f"{n} has no ctypes prototype"
    assert lib.rawdtw_abi_version() == 2
    assert lib.rawdtw_status_string(4) == b"job window out of range"


def test_job_struct_layout():
    assert JOB_DTYPE.itemsize == 32 and ANCHOR_DTYPE.itemsize == 8
    assert [JOB_DTYPE.fields[k][1] for k in JOB_DTYPE.names] == [0, 8, 12, 16, 20, 24, 28]


def test_no_device_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a device is present")
    with pytest.raises(ra.RawDTWError) as e:
        ra.Engine(0)
    assert e.value.status == 6


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def random_chain(rng, n_anchors, q0, t0, big=False):
    """anchors end-first with strictly increasing target and non-decreasing... (rmap.cpp:453-461: target diff > 0, query diff >= 0, not both equal)"""
    q, t = [q0], [t0]
    for _ in range(n_anchors - 1):
        dq = int(rng.integers(1, 60 if big else 14))
        dt = max(1, int(round(dq * rng.uniform(0.5, 1.25))))
        q.append(q[-1] + dq)
        t.append(t[-1] + dt)
    a = np.zeros(n_anchors, ANCHOR_DTYPE)
    a["query_position"] = q[::-1]
    a["target_position"] = t[::-1]
    return a


@pytest.mark.parametrize("border,fill", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("fused", [0, 1])
def test_host_mirror_matches_oracle(oracle, border, fill, fused):
    """build_jobs + (oracle-computed job costs) + replay == the oracle's align_chain, for every
    min_score regime (never cut / cut at part k / cut before the first part)."""
    lib = load_library()
    rng = np.random.default_rng(100 + border * 2 + fill)
    ref = rng.normal(size=6000).astype(np.float32)
    opt = AlignOpt(border, fill, 0.10, 0.4, 20.0, fused)
    oopt = OrcOpt(border, fill, 0.10, 0.4, 20.0, fused)
    for trial in range(120):
        na = int(rng.integers(2, 30))
        anchors = random_chain(rng, na, int(rng.integers(0, 50)), int(rng.integers(0, 500)))
        n_ev = int(anchors[0]["query_position"]) + 1
        # reads that partly match the reference so that scores straddle the thresholds
        events = (ref[int(anchors[-1]["target_position"]):][:n_ev] if trial % 2 else rng.normal(size=n_ev)).astype(np.float32)
        if len(events) < n_ev:
            events = np.concatenate([events, rng.normal(size=n_ev - len(events)).astype(np.float32)])
        events = events + rng.normal(scale=0.2, size=n_ev).astype(np.float32)
        nj = lib.rawdtw_chain_job_count(C.byref(opt), na)
        jobs = np.zeros(nj, JOB_DTYPE)
        assert lib.rawdtw_chain_build_jobs(C.byref(opt), _ptr(anchors), na, 0, 0, 0, _ptr(jobs)) == 0
        costs = np.array([
            oracle.dtw_global(events[j["read_off"]:j["read_off"] + j["n"]], ref[j["ref_off"]:j["ref_off"] + j["m"]], j["exclude_last"])
            if j["band_radius"] < 0 else
            oracle.dtw_banded(events[j["read_off"]:j["read_off"] + j["n"]], ref[j["ref_off"]:j["ref_off"] + j["m"]], j["band_radius"], j["exclude_last"])
            for j in jobs], np.float32)
        for min_score in (-1e10, 0.0, 5.0, 30.0, 1e6):
            want = oracle.align_chain(anchors, ref, events, oopt, min_score)
            got = np.float32(lib.rawdtw_chain_replay(C.byref(opt), _ptr(anchors), na, _ptr(costs), C.c_float(min_score)))
            assert got.view(np.uint32) == want.view(np.uint32), (trial, min_score, got, want)


def test_sort_matches_stable_insertion_for_small_inputs():
    lib = load_library()
    s = np.array([3, 1, 3, 2, 2, 5, 1], np.float32)
    perm = np.zeros(len(s), np.uint32)
    assert lib.rawdtw_sort_by_chaining_score(_ptr(s), len(s), _ptr(perm)) == 0
    # <= 16 elements: libstdc++'s std::sort is one insertion sort => ties keep input order
    assert list(perm) == [5, 0, 2, 3, 4, 1, 6]


def test_batch_build_and_replay_against_oracle_loop(oracle):
    """rawdtw_batch_build_jobs / rawdtw_batch_replay == the sequential loop of rmap.cpp:515-524."""
    lib = load_library()
    rng = np.random.default_rng(21)
    ref = rng.normal(size=20000).astype(np.float32)
    opt = AlignOpt(1, 1, 0.10, 0.4, 20.0, 1)
    oopt = OrcOpt(1, 1, 0.10, 0.4, 20.0, 1)
    n_reads = 40
    chain_off = [0]
    anchor_off = [0]
    anchors_all, ref_base, read_base, events_all = [], [], [], []
    ev_acc = 0
    per_read = []
    for r in range(n_reads):
        true_t = int(rng.integers(0, 15000))
        n_ev = int(rng.integers(300, 900))
        events = (ref[true_t:true_t + n_ev] + rng.normal(scale=0.25, size=n_ev)).astype(np.float32)
        chains = []
        for c in range(int(rng.integers(1, 6))):
            na = int(rng.integers(2, 40))
            t0 = true_t + int(rng.integers(0, 20)) if c == 0 else int(rng.integers(0, 15000))
            a = random_chain(rng, na, int(rng.integers(0, 20)), t0)
            # keep the chain inside the read and the reference
            while a[0]["query_position"] >= n_ev or a[0]["target_position"] >= len(ref):
                a = a[1:]
            if len(a) < 2:
                continue
            chains.append(a)
        per_read.append((events, chains))
        for a in chains:
            anchors_all.append(a)
            anchor_off.append(anchor_off[-1] + len(a))
            ref_base.append(0)
            read_base.append(ev_acc)
        chain_off.append(chain_off[-1] + len(chains))
        events_all.append(events)
        ev_acc += n_ev
    events_cat = np.concatenate(events_all)
    anchors_cat = np.concatenate(anchors_all)
    anchor_off = np.array(anchor_off, np.uint64)
    chain_off = np.array(chain_off, np.uint64)
    ref_base = np.array(ref_base, np.uint64)
    read_base = np.array(read_base, np.uint32)
    n_chains = len(anchors_all)
    job_off = np.zeros(n_chains + 1, np.uint64)
    nj = C.c_uint64()
    assert lib.rawdtw_batch_build_jobs(C.byref(opt), n_chains, _ptr(anchor_off), _ptr(anchors_cat), _ptr(ref_base),
                                       _ptr(read_base), _ptr(job_off), None, 0, C.byref(nj)) == 0
    jobs = np.zeros(nj.value, JOB_DTYPE)
    assert lib.rawdtw_batch_build_jobs(C.byref(opt), n_chains, _ptr(anchor_off), _ptr(anchors_cat), _ptr(ref_base),
                                       _ptr(read_base), _ptr(job_off), _ptr(jobs), len(jobs), C.byref(nj)) == 0
    costs = oracle.batch_costs(jobs, events_cat, ref, nthreads=2)
    score = np.zeros(n_chains, np.float32)
    keep = np.zeros(n_chains, np.uint8)
    assert lib.rawdtw_batch_replay(C.byref(opt), n_reads, _ptr(chain_off), _ptr(anchor_off), _ptr(anchors_cat),
                                   _ptr(job_off), _ptr(costs), _ptr(score), _ptr(keep)) == 0
    # the oracle's sequential loop, read by read
    c = 0
    cut = 0
    for events, chains in per_read:
        best = np.float32(0.0)
        for a in chains:
            want = oracle.align_chain(a, ref, events, oopt, float(best))
            assert score[c].view(np.uint32) == want.view(np.uint32)
            k = want >= np.float32(20.0)
            assert bool(keep[c]) == bool(k)
            if k and want > best:
                best = want
            cut += want == np.float32(-1e10)
            c += 1
    assert c == n_chains and cut > 0  # the early-exit path was exercised


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under rawalign_amd/ or include/ may import, load or mention it
    outside comments."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for base in ("rawalign_amd", "include"):
        for dp, _, files in os.walk(os.path.join(root, base)):
            for fn in files:
                if not fn.endswith((".py", ".h", ".cpp", ".hip")):
                    continue
                for ln in open(os.path.join(dp, fn), encoding="utf-8"):
                    code = ln.split("#")[0] if fn.endswith(".py") else ln.split("//")[0]
                    assert not re.search(r"\boracle\b|liboracle|_ref/", code), (fn, ln)


def test_compact_anchor_lists_round_trip():
    """rawdtw_anchors_pack / _unpack (include/rawdtw.h): 2-byte steps back along every chain, chains' first entries and every
    8192nd entry whole, steps of 255 or more in the escape list; a chain that does not descend is refused."""
    import ctypes as C

    from rawalign_amd.align import COMPACT_STRIDE, pack_anchors, unpack_anchors

    lib = ra.load_library()
    rng = np.random.default_rng(3)
    lens = np.concatenate([rng.integers(0, 40, 300), [1, 2, 9000, 8192, 1, 0, 17000], rng.integers(1, 6, 2000)])
    anchor_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    anchors = np.zeros(int(anchor_off[-1]), ra.ANCHOR_DTYPE)
    for c, n in enumerate(lens):
        if n == 0:
            continue
        dq = rng.integers(0, 30, n); dt = rng.integers(0, 30, n)
        big = rng.random(n) < 0.02   # steps beyond a byte, in either component or both
        dq[big] = rng.integers(200, 5000, int(big.sum())); dt[rng.random(n) < 0.01] = 254; dt[rng.random(n) < 0.01] = 255
        q = 10 + np.cumsum(dq); t = 1000 + np.cumsum(dt)
        a0 = int(anchor_off[c])
        anchors["query_position"][a0:a0 + n] = q[::-1]   # end-first (rmap.cpp:193-196)
        anchors["target_position"][a0:a0 + n] = t[::-1]
    ca = pack_anchors(lib, anchor_off, anchors)
    assert len(ca.steps) == len(anchors) and len(ca.unit_abs) == (len(anchors) + COMPACT_STRIDE - 1) // COMPACT_STRIDE
    assert len(ca.wide) > 50 and np.all(np.diff(ca.wide["index"].astype(np.int64)) > 0)
    back = unpack_anchors(lib, anchor_off, ca)
    assert np.array_equal(back.view(np.uint8), anchors.view(np.uint8))
    assert ca.nbytes < 0.4 * anchors.nbytes  # (2 300 chains of a handful of entries, 3 % escapes: the bench batch is at 0.26)
    # not descending: refused
    bad = anchors.copy()
    bad["query_position"][int(anchor_off[302]) + 5] += 100000
    with pytest.raises(ValueError):
        pack_anchors(lib, anchor_off, bad)
