"""`.ind` index reader: header, sequence table and signal arrays of a file in
ri_idx_dump's layout (src/rawindex.cpp:275-315)."""
import numpy as np
import pytest

import rawalign_amd as ra
from rawalign_amd.index import Index, write_index


def make_index(tmp_path, lens=(1500, 7, 32001)):
    rng = np.random.default_rng(3)
    names = [f"chr{i}" for i in range(len(lens))]
    fwd = [rng.normal(size=n).astype(np.float32) for n in lens]
    rev = [rng.normal(size=n).astype(np.float32) for n in lens]
    p = str(tmp_path / "ref.ind")
    write_index(p, names, fwd, rev, e=7, q=9, lq=3, k=6)
    return p, names, fwd, rev


def test_index_header_and_signals(tmp_path):
    p, names, fwd, rev = make_index(tmp_path)
    ix = Index(p)
    assert ix.n_seq == 3 and ix.names == names and ix.lens == [len(x) for x in fwd]
    assert (ix.e, ix.q, ix.lq, ix.k, ix.w, ix.n) == (7, 9, 3, 6, 0, 0)
    for s in range(3):
        assert np.array_equal(ix.signal(s, 1), fwd[s])  # strand==1 -> forward_signals (rmap.cpp:182-188)
        assert np.array_equal(ix.signal(s, 0), rev[s])


def test_index_rejects_other_files(tmp_path):
    p = tmp_path / "x.ind"
    p.write_bytes(b"MM" + b"\0" * 64)
    with pytest.raises(ValueError):
        Index(str(p))
    t = tmp_path / "trunc.ind"
    good, *_ = make_index(tmp_path)
    t.write_bytes(open(good, "rb").read()[:200])
    with pytest.raises(ValueError):
        Index(str(t))


@pytest.mark.gpu
def test_index_upload_feeds_the_kernels(tmp_path, oracle):
    p, names, fwd, rev = make_index(tmp_path)
    eng = ra.Engine(0)
    Index(p).upload(eng)
    rng = np.random.default_rng(1)
    events = rng.normal(size=400).astype(np.float32)
    jobs = np.zeros(6, ra.JOB_DTYPE)
    want = []
    for k, (s, strand) in enumerate([(0, 1), (0, 0), (1, 1), (1, 0), (2, 1), (2, 0)]):
        arr = fwd[s] if strand == 1 else rev[s]
        m = min(len(arr), 40 + k)
        off = len(arr) - m  # windows ending at the last element of the array
        jobs[k] = (eng.reference_offset(s, strand) + off, 10 * k, 30 + k, m, 3, k & 1, 0)
        want.append(oracle.dtw_banded(events[10 * k:10 * k + 30 + k], arr[off:off + m], 3, k & 1))
    got = eng.score_batch(jobs, events)
    assert np.array_equal(got.view(np.uint32), np.array(want, np.float32).view(np.uint32))
