"""Host consumers downstream of the DTW block (pure host code): primary-chain selection + MAPQ,
the stop rule, PAF formatting and sequence-until, against plain-Python restatements of the
reference (rmap.cpp:65-128, 594-665, 696-801, 918-965; sequence_until.c:4-18).
The reference's rmap.cpp cannot be built here (HDF5), so these are pinned by restatement only --
except find_outlier, whose translation unit compiles on its own."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import rawalign_amd as ra
from rawalign_amd import mapping as M

f32 = np.float32


def mk_chain(rng, score=None, ali=None, ref=0, strand=0, start=None, n=None):
    n = n or int(rng.integers(2, 12))
    start = int(rng.integers(0, 5000)) if start is None else start
    t = start + np.cumsum(rng.integers(1, 20, n))
    q = np.cumsum(rng.integers(1, 20, n))
    a = np.zeros(n, ra.ANCHOR_DTYPE)
    a["target_position"] = t[::-1]
    a["query_position"] = q[::-1]
    c = ra.Chain(float(score if score is not None else rng.uniform(10, 200)), ref, strand, a)
    c.alignment_score = float(ali if ali is not None else rng.uniform(-50, 300))
    c.start_position = int(t[0])
    c.end_position = int(t[-1])
    return c


def py_primary(chains, evaluate):
    key = lambda c: (f32(c.alignment_score), f32(c.chaining_score), c.n_anchors, c.strand,  # noqa: E731
                     c.reference_sequence_index, c.start_position, c.end_position)
    srt = sorted(chains, key=key, reverse=True)
    prim = [srt[0]]
    for c in srt[1:]:
        back = prim[-1]
        if evaluate:
            if f32(c.alignment_score) < f32(f32(back.alignment_score) / f32(3)):
                break
        elif f32(c.chaining_score) < f32(f32(back.chaining_score) / f32(3)):
            break
        if any(c.reference_sequence_index == p.reference_sequence_index and
               max(c.start_position, p.start_position) <= min(c.end_position, p.end_position) for p in prim):
            continue
        prim.append(c)
    if len(prim) == 1:
        mapq = 60
    else:
        s0, s1 = (f32(prim[0].alignment_score), f32(prim[1].alignment_score)) if evaluate else \
                 (f32(prim[0].chaining_score), f32(prim[1].chaining_score))
        with np.errstate(all="ignore"):
            v = f32(40) * f32(f32(1) - f32(s1 / s0))
        mapq = int(np.trunc(v)) if np.isfinite(v) else 0
        mapq = max(0, min(60, mapq)) & 0xFF
    return prim, mapq


def py_high_conf(prim, evaluate, br=1.2, mr=5.0, mca=2):
    if not prim or prim[0].n_anchors == 0:
        return False
    sc = [f32(c.alignment_score if evaluate else c.chaining_score) for c in prim]
    if len(prim) >= 2:
        with np.errstate(all="ignore"):
            if f32(sc[0] / sc[1]) >= f32(br):
                return True
        mean = f32(0)
        for s in sc:
            mean = f32(mean + s)
        mean = f32(mean / f32(len(prim)))
        return bool(sc[0] >= f32(f32(mr) * mean))
    return prim[0].n_anchors >= mca


@pytest.mark.parametrize("evaluate", [True, False])
def test_primary_chains_and_mapq(evaluate):
    rng = np.random.default_rng(5 + evaluate)
    opt = ra.MapOpt(flag=2 if evaluate else 0)
    for trial in range(300):
        n = int(rng.integers(1, 12))
        chains = [mk_chain(rng, ref=int(rng.integers(0, 3)), strand=int(rng.integers(0, 2)),
                           ali=rng.uniform(20, 300)) for _ in range(n)]
        want, mapq = py_primary(chains, evaluate)
        got = M.gen_primary_chains(chains, opt)
        assert [id(c) for c in got] == [id(c) for c in want], trial
        assert got[0].mapq == mapq
        assert M.is_mapped_with_high_confidence(got, opt) == py_high_conf(want, evaluate)


def test_paf_line_mapped_and_unmapped():
    rng = np.random.default_rng(9)
    opt = ra.MapOpt()
    c0 = mk_chain(rng, score=120, ali=210.5, ref=1, strand=1, start=4000, n=9)
    c1 = mk_chain(rng, score=40, ali=30.0, ref=0, strand=0, start=100, n=3)
    prim = M.gen_primary_chains([c1, c0], opt)
    assert prim[0] is c0
    rs = M.ReadState("read_7", qlen=12000, offset=1450, chunks_done=2, broke_early=True, primary=prim)
    line = M.paf_line(rs, ["chrA", "chrB"], [50000, 60000], opt).split("\t")
    assert line[0] == "read_7" and line[4] == "-" and line[5] == "chrB" and line[6] == "60000"
    scale = f32(f32(f32(3) * f32(4000)) / f32(1450)) / f32(f32(4000) / f32(450))
    q_end = int(c0.anchors[0]["query_position"])
    q_start = int(c0.anchors[-1]["query_position"])
    assert int(line[1]) == int(np.uint32(scale * f32(q_end))) == int(line[3])
    assert int(line[2]) == int(np.uint32(scale * f32(q_start)))
    assert int(line[7]) == 60000 + 1 - c0.end_position            # reverse strand (rmap.cpp:751)
    assert int(line[8]) - int(line[7]) == c0.end_position - c0.start_position + 1 == int(line[10])
    assert int(line[11]) == prim[0].mapq
    tags = dict(t.split(":", 2)[0::2] for t in line[12:])
    assert tags["ci"] == "3" and tags["sl"] == "12000" and tags["cm"] == "9" and tags["nc"] == str(len(prim))
    assert tags["s1"] == "120.000000"
    # unmapped read with no chains (rmap.cpp:783-801, 965)
    rs2 = M.ReadState("read_8", qlen=9000, offset=900, chunks_done=3, broke_early=False, primary=[])
    l2 = M.paf_line(rs2, ["chrA"], [50000], opt).split("\t")
    assert l2[2:11] == ["*"] * 9 and l2[11] == "0"
    assert "cm:i:0" in l2 and "s1:f:0" in l2
    # the loop ran out of signal: current_chunk steps back by one (rmap.cpp:696): 3 chunks -> ci = 3
    assert "ci:i:3" in l2


def test_find_outlier_against_compiled_reference(tmp_path):
    src = "/root/reference/src/sequence_until.c"
    rng = np.random.default_rng(2)
    xs = [rng.random((5, n)).astype(np.float32) for n in (1, 3, 8, 17, 40)]
    ours = [M.find_outlier(x) for x in xs]
    # restatement in numpy fp32, sequential accumulation
    for x, o in zip(xs, ours):
        outl, best = 0, f32(0)
        for i in range(len(x)):
            d = f32(0)
            for j in range(x.shape[1]):
                t = f32(x[i, j] - x[outl, j])
                d = f32(d + f32(t * t))
            if d > best:
                best, outl = d, i
        assert o == best
    if not os.path.exists(src):
        pytest.skip("no /root/reference here")
    # the reference's translation unit, built twice from where it lies: with contraction off (the source's
    # arithmetic) and as its Makefile builds it on an FMA host (-O3, FMA available): both matched bit for bit
    xs += [rng.random((4, n)).astype(np.float32) for n in (2, 4, 5, 7, 9, 12, 15, 16, 31, 33, 100, 257)]
    for flags, contracted in ((["-ffp-contract=off", "-march=x86-64-v3"], False), (["-march=x86-64-v3"], True)):
        so = tmp_path / f"libsu{int(contracted)}.so"
        subprocess.run(["gcc", "-O3", *flags, "-shared", "-fPIC", "-I/root/reference/src", "-o", str(so), src],
                       check=True)
        lib = C.CDLL(str(so))
        lib.find_outlier.restype = C.c_float
        for x in xs:
            rows = (C.c_void_p * len(x))(*[x[i].ctypes.data for i in range(len(x))])
            ref = np.float32(lib.find_outlier(rows, x.shape[1], len(x)))
            assert ref == M.find_outlier(x, contracted=contracted), (flags, x.shape)


def test_sequence_until_state_machine():
    su = M.SequenceUntil(n_seq=3, tmin_reads=10, ttest_freq=5, tn_samples=3, t_threshold=1.5)
    rng = np.random.default_rng(4)
    fired_at = None
    for k in range(200):
        if su.add_mapped_read(int(rng.integers(0, 3)), int(rng.integers(100, 200)), k):
            fired_at = k
            break
    # estimates are fractions in [0,1]: distances are far below 1.5, so it fires at the first eligible test:
    # tests happen at nreads = 15, 20, 25, ... and the outlier check starts once tn_samples estimates exist
    assert fired_at == 29 and su.stop == 30


def py_chain(anchors, e=6, max_gap=2000, max_tgap=5000, band=5000, max_skips=25, min_anchors=2, nbest=3,
             min_score=10.0, maxs=0.0):
    """Plain-Python restatement of the chaining DP + traceback (rmap.cpp:430-507, 130-173), fp32 scores."""
    n = len(anchors)
    score = [f32(e)] * n
    pred = list(range(n))
    used = [False] * n
    ends = []
    maxs = f32(maxs)
    T = [int(x) for x in anchors["target_position"]]
    Q = [int(x) for x in anchors["query_position"]]
    for ai in range(n):
        start = ai - band if ai > band else 0
        skips = 0
        for pi in range(ai - 1, start - 1, -1):
            if Q[pi] == Q[ai] or T[pi] == T[ai]:
                continue
            if T[pi] + max_tgap < T[ai]:
                break
            td, qd = T[ai] - T[pi], Q[ai] - Q[pi]
            if qd < 0:
                continue
            cur = f32(0)
            scale = f32(f32(qd) / f32(td)) if td > 0 else f32(1)
            if abs(td - qd) < max_gap and scale < f32(5) and float(scale) > 0.75:
                cur = f32(score[pi] + f32(min(td, qd, e)))
            if cur > score[ai]:
                score[ai], pred[ai] = cur, pi
                skips -= 1
            else:
                skips += 1
                if skips > max_skips:
                    break
        if score[ai] > maxs:
            maxs = score[ai]
        if score[ai] >= f32(min_score) and score[ai] > f32(maxs / f32(2)):
            ends.append((score[ai], ai))
    ends.sort(key=lambda x: (-float(x[0]), -x[1]))
    chains = []
    for k, (_, end) in enumerate(ends[:nbest]):
        if not used[end]:
            idx = [end]
            stop = pred[end] != end and used[pred[end]]
            used[end] = True
            cur = end
            while pred[cur] != cur and not used[pred[cur]]:
                cur = pred[cur]
                idx.append(cur)
                if pred[cur] != cur and used[pred[cur]]:
                    stop = True
                used[cur] = True
            if len(idx) >= min_anchors:
                adj = score[end]
                if stop:
                    adj = f32(adj - score[pred[cur]])
                chains.append((float(adj), idx))
        if score[end] < f32(maxs / f32(2)):
            break
    return chains, float(maxs)


def test_chaining_dp_matches_restatement():
    rng = np.random.default_rng(17)
    copt = M.default_chain_opt(6)
    for trial in range(120):
        # seed hits on a true diagonal with local stretch + random false hits
        n_true = int(rng.integers(0, 60))
        q = np.sort(rng.choice(800, size=n_true, replace=False)) if n_true else np.zeros(0, int)
        t = 3000 + (q * rng.uniform(0.7, 1.0)).astype(int) + rng.integers(0, 3, n_true)
        nf = int(rng.integers(0, 80))
        qf, tf = rng.integers(0, 800, nf), rng.integers(0, 20000, nf)
        a = np.zeros(n_true + nf, ra.ANCHOR_DTYPE)
        a["query_position"] = np.concatenate([q, qf])
        a["target_position"] = np.concatenate([t, tf])
        a = np.sort(a, order=["target_position", "query_position"])  # rmap.h:24-26 operator<
        start_max = float(rng.choice([0.0, 30.0]))
        got, gmax = M.chain_anchors(a, copt, start_max, ref_index=0, strand=1)
        want, wmax = py_chain(a, maxs=start_max)
        assert gmax == wmax
        assert len(got) == len(want)
        for g, (ws, widx) in zip(got, want):
            assert f32(g.chaining_score) == f32(ws)
            assert [int(x) for x in g.anchors["target_position"]] == [int(a[i]["target_position"]) for i in widx]
            assert [int(x) for x in g.anchors["query_position"]] == [int(a[i]["query_position"]) for i in widx]
            assert g.start_position == int(a[widx[-1]]["target_position"]) and g.end_position == int(a[widx[0]]["target_position"])
