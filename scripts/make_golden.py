#!/usr/bin/env python3
"""Generate tests/golden/dtw_golden.npz from the REFERENCE's own compiled dtw.cpp.

Runs only where /root/reference exists (the build container): it builds
oracle/_ref/libref_dtw.so with oracle/Makefile (the reference sources are compiled
where they lie, never copied) and records, for a fixed list of seeded inputs, what the
reference returns.  The fixture holds data only: inputs and the reference's outputs.

    python scripts/make_golden.py

Layout of the .npz (all little-endian):
  vals      float32[sum(n_k + m_k)]   operands, a_k then b_k per case
  cases     int64[n_cases, 6]         n, m, R0, exclude_last, offset into vals, tb_offset (-1: no path)
  global_   uint32[n_cases]           bits of DTW_global
  banded    uint32[n_cases]           bits of DTW_global_slantedbanded_antidiagonalwise
  tb_cost   uint32[n_cases]           bits of DTW_global_tb cost (0 when no path stored)
  tb_len    int64[n_cases]
  tb_i/tb_j uint32[sum tb_len]        path positions; tb_d uint32 bits of the differences
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.loader import RefDTW, build_ref  # noqa: E402


def known_answer_inputs(n, m):
    """The closed-form inputs SURVEY.md 8(c) quotes known answers for."""
    a = (((np.arange(n) * 37 + 11) % 101).astype(np.float32) / np.float32(20.0) - np.float32(2.5)).astype(np.float32)
    b = (((np.arange(m) * 53 + 7) % 97).astype(np.float32) / np.float32(19.0) - np.float32(2.5)).astype(np.float32)
    return a, b


def default_radius(n):
    # rmap.cpp:214,276 with dtw_band_radius_frac = 0.10f
    return max(1, int(np.float32(n) * np.float32(0.10)))


def main():
    assert build_ref() is not None, "needs /root/reference"
    ref = RefDTW()
    rng = np.random.default_rng(20231005)
    cases = []  # (a, b, R0, excl, want_tb)

    for n, m in [(4, 4), (10, 10), (25, 10), (10, 25), (64, 50), (200, 30), (33, 47)]:
        a, b = known_answer_inputs(n, m)
        for ex in (0, 1):
            cases.append((a, b, default_radius(n), ex, True))

    # check_dtw.cpp:188-234 shape groups, our own seeded values in the same range
    for n, m in [(4, 4), (10, 10), (20, 10), (25, 10), (100, 100), (200, 50), (200, 30)]:
        for k in range(6):
            a = rng.uniform(-2.5, 2.5, n).astype(np.float32)
            b = rng.uniform(-2.5, 2.5, m).astype(np.float32)
            cases.append((a, b, default_radius(n), k & 1, k < 2))
            cases.append((b, a, default_radius(m), k & 1, False))  # n < m: swap path

    # every radius parity / slant combination on small shapes, band often clips the optimum
    for n in (1, 2, 3, 5, 8, 12, 20):
        for m in (1, 2, 3, 7, 12, 19):
            for R0 in (0, 1, 2, 3, 4, 7):
                a = rng.normal(0, 1, n).astype(np.float32)
                b = rng.normal(0, 1, m).astype(np.float32)
                cases.append((a, b, R0, (n + m + R0) & 1, n * m <= 64))

    # sparse-mode like segments (2..50 events), default radius
    for _ in range(200):
        n = int(rng.integers(2, 51))
        m = max(2, int(round(n * rng.uniform(0.4, 1.3))))
        a = rng.normal(0, 1, n).astype(np.float32)
        b = rng.normal(0, 1, m).astype(np.float32)
        cases.append((a, b, default_radius(n), int(rng.integers(0, 2)), False))

    # mid/large banded + full (global-mode like)
    for n, m in [(300, 260), (260, 300), (513, 512), (1000, 800), (800, 1000), (2048, 2048), (3000, 1700)]:
        a = rng.normal(0, 1, n).astype(np.float32)
        b = rng.normal(0, 1, m).astype(np.float32)
        cases.append((a, b, default_radius(n), 0, n * m <= 300 * 300))
        cases.append((a, b, default_radius(n) + 1, 1, False))

    vals, meta, g_bits, b_bits, tb_cost, tb_len, tb_i, tb_j, tb_d = [], [], [], [], [], [], [], [], []
    off = 0
    tb_off = 0
    for a, b, R0, ex, want_tb in cases:
        n, m = len(a), len(b)
        vals += [a, b]
        g_bits.append(ref.dtw_global(a, b, ex).view(np.uint32))
        b_bits.append(ref.dtw_banded(a, b, R0, ex).view(np.uint32))
        if want_tb:
            c, pi, pj, pd = ref.dtw_global_tb(a, b, ex)
            tb_cost.append(c.view(np.uint32))
            tb_len.append(len(pi))
            tb_i.append(pi)
            tb_j.append(pj)
            tb_d.append(pd.view(np.uint32))
            meta.append((n, m, R0, ex, off, tb_off))
            tb_off += len(pi)
        else:
            tb_cost.append(np.uint32(0))
            tb_len.append(0)
            meta.append((n, m, R0, ex, off, -1))
        off += n + m

    out = os.path.join(ROOT, "tests", "golden", "dtw_golden.npz")
    np.savez_compressed(
        out,
        vals=np.concatenate(vals).astype("<f4"),
        cases=np.array(meta, dtype="<i8"),
        global_=np.array(g_bits, dtype="<u4"),
        banded=np.array(b_bits, dtype="<u4"),
        tb_cost=np.array(tb_cost, dtype="<u4"),
        tb_len=np.array(tb_len, dtype="<i8"),
        tb_i=np.concatenate(tb_i).astype("<u4"),
        tb_j=np.concatenate(tb_j).astype("<u4"),
        tb_d=np.concatenate(tb_d).astype("<u4"),
    )
    print(f"{len(cases)} cases, {off} operand floats, {tb_off} path elements -> {out} "
          f"({os.path.getsize(out) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
