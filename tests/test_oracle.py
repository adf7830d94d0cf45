"""The CPU oracle against (1) the golden vectors captured from the reference's own dtw.cpp,
(2) the known answers and visited-cell masks recorded in SURVEY.md, and (3) -- where
oracle/_ref is present -- the compiled reference itself on fresh random inputs."""
import numpy as np
import pytest

from tests.golden_util import bits

# (n, m, R0) -> DTW_global excl F/T, banded excl F/T, tb path length
KNOWN = [
    (4, 4, 1, 4.82105255, 2.23947358, 4.82105255, 2.23947358, 5),
    (10, 10, 1, 11.6842098, 8.68157864, 11.6842098, 8.68157864, 11),
    (25, 10, 2, 22.2394753, 21.7368431, 22.2394753, 21.7368431, 25),
    (10, 25, 1, 24.7263184, 23.6236877, 25.0526314, 23.9500008, 25),
    (64, 50, 6, 52.6105423, 49.244751, 52.6105423, 49.244751, 68),
    (200, 30, 20, 230.25, 225.615784, 230.25, 225.615784, 200),
    (33, 47, 3, 42.5421104, 39.3947411, 42.5421104, 39.3947411, 48),
]


def known_inputs(n, m):
    a = (((np.arange(n) * 37 + 11) % 101).astype(np.float32) / np.float32(20.0) - np.float32(2.5)).astype(np.float32)
    b = (((np.arange(m) * 53 + 7) % 97).astype(np.float32) / np.float32(19.0) - np.float32(2.5)).astype(np.float32)
    return a, b


@pytest.mark.parametrize("n,m,R0,g0,g1,b0,b1,plen", KNOWN)
def test_known_answers(oracle, n, m, R0, g0, g1, b0, b1, plen):
    a, b = known_inputs(n, m)
    assert max(1, int(np.float32(n) * np.float32(0.10))) == R0
    assert oracle.dtw_global(a, b) == np.float32(g0)
    assert oracle.dtw_global(a, b, True) == np.float32(g1)
    assert oracle.dtw_banded(a, b, R0) == np.float32(b0)
    assert oracle.dtw_banded(a, b, R0, True) == np.float32(b1)
    assert len(oracle.dtw_global_tb(a, b)[1]) == plen


def test_golden_costs_bit_exact(oracle, golden):
    for c in golden:
        assert bits(oracle.dtw_global(c.a, c.b, c.exclude_last)) == c.global_bits
        assert bits(oracle.dtw_banded(c.a, c.b, c.R0, c.exclude_last)) == c.banded_bits


def test_golden_cellset_formulation(oracle, golden):
    """The geometric restatement (plain recurrence over the band's cell set) is the same function."""
    for c in golden:
        if len(c.a) * len(c.b) > 40000:
            continue
        cost, cells, mask = oracle.dtw_banded_cellset(c.a, c.b, c.R0, c.exclude_last)
        assert bits(cost) == c.banded_bits
        assert cells == int(mask.sum()) == oracle.banded_cells(len(c.a), len(c.b), c.R0)


def test_golden_traceback(oracle, golden):
    seen = 0
    for c in golden:
        if c.tb is None:
            continue
        cost, pi, pj, pd = oracle.dtw_global_tb(c.a, c.b, c.exclude_last)
        assert bits(cost) == c.tb[0]
        assert np.array_equal(pi, c.tb[1]) and np.array_equal(pj, c.tb[2])
        assert np.array_equal(pd.view(np.uint32), c.tb[3])
        seen += 1
    assert seen > 50


def test_directions_reproduce_path(oracle, golden):
    """Walking the 2-bit direction matrix gives the reference's path (the GPU keeps only this)."""
    for c in golden:
        if c.tb is None or c.exclude_last:
            continue
        d = oracle.dtw_directions(c.a, c.b)
        i, j = len(c.a) - 1, len(c.b) - 1
        rev = [(i, j)]
        while i > 0 or j > 0:
            code = d[i, j]
            if code == 1:
                i -= 1
            elif code == 2:
                j -= 1
            else:
                i -= 1
                j -= 1
            rev.append((i, j))
        rev.reverse()
        assert [p[0] for p in rev] == list(c.tb[1]) and [p[1] for p in rev] == list(c.tb[2])


# cells per row (row = index in the shorter sequence) from the reference's DEBUG build
MASK_ROWS = {
    (12, 12, 2): [3, 4, 5, 5, 5, 5, 5, 5, 5, 5, 4, 3],
    (12, 12, 3): [4, 5, 6, 7, 7, 7, 7, 7, 7, 6, 5, 4],
    (20, 8, 2): [10, 12, 15, 16, 17, 14, 12, 9],
    (20, 8, 1): [6, 9, 9, 10, 9, 10, 8, 6],
}


@pytest.mark.parametrize("shape", sorted(MASK_ROWS))
def test_visited_cell_masks(oracle, shape):
    n, m, R0 = shape
    rng = np.random.default_rng(7)
    a = rng.normal(size=n).astype(np.float32)
    b = rng.normal(size=m).astype(np.float32)
    _, cells, mask = oracle.dtw_banded_cellset(a, b, R0)
    assert list(mask.sum(axis=0)) == MASK_ROWS[shape]
    assert cells == sum(MASK_ROWS[shape])


def test_square_band_is_diagonal_strip(oracle):
    # for square inputs the set is |i-j| <= R, n(2R+1) - R(R+1) cells
    for n, R in [(12, 2), (12, 3), (30, 5), (31, 4), (64, 6)]:
        assert oracle.banded_cells(n, n, R) == n * (2 * R + 1) - R * (R + 1)


def test_against_compiled_reference(oracle):
    from oracle.loader import RefDTW

    if not RefDTW.available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    ref = RefDTW()
    rng = np.random.default_rng(99)
    for t in range(3000):
        n, m = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        a = rng.normal(size=n).astype(np.float32)
        b = rng.normal(size=m).astype(np.float32)
        R0 = int(rng.integers(0, 14))
        ex = t & 1
        assert bits(oracle.dtw_global(a, b, ex)) == bits(ref.dtw_global(a, b, ex))
        assert bits(oracle.dtw_banded(a, b, R0, ex)) == bits(ref.dtw_banded(a, b, R0, ex))
        if t % 7 == 0:
            c1, i1, j1, d1 = oracle.dtw_global_tb(a, b, ex)
            c2, i2, j2, d2 = ref.dtw_global_tb(a, b, ex)
            assert bits(c1) == bits(c2) and np.array_equal(i1, i2) and np.array_equal(j1, j2)
            assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32))
