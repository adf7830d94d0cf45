"""The batch planner's host half (rawdtw_plan_dry_run): binning, tiles and their invariants, without a device.
The C side verifies each plan (every job exactly once, launches partition the plan, every window staged inside
its tile's LDS image at the right place); here we drive it over the job mixes the path sees and check the counts."""
import ctypes as C
import time

import numpy as np
import pytest

import rawalign_amd as ra
from rawalign_amd import synth
from rawalign_amd._lib import AlignOpt, RawDTWError, load_library
from rawalign_amd.dtw import JOB_DTYPE, plan_dry_run


def _sparse_jobs(n_reads, seed, max_chunks=3):
    ref = synth.make_reference([200_000], seed=3)
    n = len(ref.forward[0])
    pad = (n + 3) & ~3
    offs = {(0, 1): 0, (0, 0): pad}
    cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=n_reads, max_chunks=max_chunks), seed=seed)
    lib = load_library()
    opt = AlignOpt(1, 1, 0.10, 0.4, 20.0, 1)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    job_off = np.zeros(cb.n_chains + 1, np.uint64)
    nj = C.c_uint64()
    lib.rawdtw_batch_build_jobs(C.byref(opt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base),
                                p(cb.read_base), p(job_off), None, 0, C.byref(nj))
    jobs = np.zeros(nj.value, JOB_DTYPE)
    lib.rawdtw_batch_build_jobs(C.byref(opt), cb.n_chains, p(cb.anchor_off), p(cb.anchors), p(cb.ref_base),
                                p(cb.read_base), p(job_off), p(jobs), len(jobs), C.byref(nj))
    return jobs, len(cb.events), 2 * pad


def _mixed_jobs(rng, count, n_ev, n_ref):
    """every planner class: micro, tile, grp16, wreg (1..32 chunks), LDS band, full (1..8 rows per lane, multi-strip)"""
    jobs = np.zeros(count, JOB_DTYPE)
    kind = rng.integers(0, 8, count)
    n = np.where(kind == 0, rng.integers(1, 9, count),
        np.where(kind == 1, rng.integers(1, 74, count),
        np.where(kind == 2, rng.integers(60, 200, count),
        np.where(kind == 3, rng.integers(200, 3000, count),
        np.where(kind == 4, rng.integers(3000, 20000, count),
        np.where(kind == 5, rng.integers(1, 70, count),
        np.where(kind == 6, rng.integers(70, 600, count), rng.integers(600, 3000, count))))))))
    ratio = rng.uniform(0.4, 1.0, count)
    m = np.maximum(1, (n * ratio).astype(np.int64))
    swap = rng.random(count) < 0.5
    jobs["n"] = np.where(swap, m, n)
    jobs["m"] = np.where(swap, n, m)
    frac = rng.choice([0.0, 0.05, 0.1, 0.3], count)
    jobs["band_radius"] = np.where(kind >= 5, -1, (frac * np.maximum(n, 1)).astype(np.int64))
    jobs["read_off"] = (rng.random(count) * (n_ev - jobs["n"])).astype(np.int64)
    jobs["ref_off"] = (rng.random(count) * (n_ref - jobs["m"])).astype(np.int64)
    jobs["exclude_last"] = rng.integers(0, 2, count)
    return jobs


def test_dry_run_sparse_batch_threads_agree():
    jobs, n_ev, n_ref = _sparse_jobs(400, seed=5)
    assert len(jobs) > 50_000
    base, tiles1 = plan_dry_run(jobs, n_ev, n_ref, threads=1)
    assert base["n_jobs"] == len(jobs)
    assert base["n_lane_jobs"] + base["n_wave_band_jobs"] + base["n_full_jobs"] == len(jobs)
    assert base["n_lane_jobs"] > 0.95 * len(jobs)          # sparse parts are short, narrow bands
    for threads in (2, 3, 8):
        info, tiles = plan_dry_run(jobs, n_ev, n_ref, threads=threads)
        for key in ("n_jobs", "cells", "algorithmic_bytes", "n_lane_jobs", "n_wave_band_jobs", "n_full_jobs", "n_launches"):
            assert info[key] == base[key], key
        assert tiles1 <= tiles <= tiles1 + threads           # a tile never spans two threads' runs


@pytest.mark.parametrize("options", [{}, {"micro_max_n": 0}, {"grp16": 0}, {"grp8": 0}, {"full_wg": 0}, {"lane_hi": 1},
                                     {"tile_lds_floats": 2048, "tile_max_jobs": 64}, {"lane_max_radius": 1, "lane_max_n": 32},
                                     {"sort_n": 17, "sort_r1_n": 9, "sort_r3": 1}, {"sort_n": 9, "sorted_tile_jobs": 128}])
def test_dry_run_every_class(options):
    rng = np.random.default_rng(17)
    n_ev, n_ref = 400_000, 500_000
    jobs = _mixed_jobs(rng, 6000, n_ev, n_ref)
    for threads in (1, 4):
        info, tiles = plan_dry_run(jobs, n_ev, n_ref, threads=threads, options=options)
        assert info["n_jobs"] == len(jobs) and tiles > 0
        assert info["n_full_jobs"] == int((jobs["band_radius"] < 0).sum())


def test_dry_run_cells_match_oracle(oracle):
    rng = np.random.default_rng(2)
    jobs = _mixed_jobs(rng, 300, 50_000, 50_000)
    jobs = jobs[np.maximum(jobs["n"], jobs["m"]) < 2500]
    info, _ = plan_dry_run(jobs, 50_000, 50_000)
    want = 0
    for j in jobs:
        if j["band_radius"] < 0:
            want += int(j["n"]) * int(j["m"])
        else:
            want += oracle.banded_cells(int(j["n"]), int(j["m"]), int(j["band_radius"]))
    assert info["cells"] == want


def test_dry_run_errors_report_first_bad_job():
    jobs = np.zeros(100_000, JOB_DTYPE)
    jobs["n"] = 8; jobs["m"] = 8; jobs["band_radius"] = 1
    jobs["read_off"] = np.arange(len(jobs)) % 1000
    jobs["ref_off"] = np.arange(len(jobs)) % 1000
    bad = jobs.copy()
    bad["n"][[70_001, 30_000, 99_999]] = 0
    for threads in (1, 4):
        with pytest.raises(RawDTWError) as e:
            plan_dry_run(bad, 2000, 2000, threads=threads)
        assert e.value.status == 1 and "job 30000:" in str(e.value)
    far = jobs.copy()
    far["ref_off"][55_555] = 1995
    with pytest.raises(RawDTWError) as e:
        plan_dry_run(far, 2000, 2000, threads=4)
    assert e.value.status == 4 and "job 55555:" in str(e.value)
    with pytest.raises(RawDTWError):
        plan_dry_run(jobs, 2000, 2000, options={"no_such_option": 1})
    info, tiles = plan_dry_run(jobs[:0], 2000, 2000)
    assert info["n_jobs"] == 0 and tiles == 0


def test_planner_scales_with_threads():
    """not a benchmark: the threaded planner must not be slower than one thread on a large batch"""
    jobs, n_ev, n_ref = _sparse_jobs(2000, seed=9)
    t = {}
    for threads in (1, 4):
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            plan_dry_run(jobs, n_ev, n_ref, threads=threads, options={"verify": 0})
            best = min(best, time.perf_counter() - t0)
        t[threads] = best
    print(f"planner: {len(jobs)} jobs, 1 thread {t[1]*1e3:.1f} ms, 4 threads {t[4]*1e3:.1f} ms")
    assert t[4] < 1.5 * t[1]
