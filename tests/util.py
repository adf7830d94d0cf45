"""Shared helpers for the parity tests."""
import numpy as np

from rawalign_amd.dtw import JOB_DTYPE


def default_radius(n, frac=0.10):
    # rmap.cpp:214,276
    return max(1, int(np.float32(n) * np.float32(frac)))


def make_arena_jobs(cases):
    """cases: list of (a, b, band_radius, exclude_last). Lays every a into an event arena and every b
    into a reference arena; returns (jobs, events, ref)."""
    jobs = np.zeros(len(cases), JOB_DTYPE)
    ev, rf = [], []
    eo = ro = 0
    for k, (a, b, R0, ex) in enumerate(cases):
        jobs[k] = (ro, eo, len(a), len(b), R0, ex, 0)
        ev.append(np.asarray(a, np.float32))
        rf.append(np.asarray(b, np.float32))
        eo += len(a)
        ro += len(b)
    return jobs, np.concatenate(ev), np.concatenate(rf)


def oracle_costs(oracle, jobs, events, ref):
    out = np.zeros(len(jobs), np.float32)
    for k, j in enumerate(jobs):
        a = events[j["read_off"]:j["read_off"] + j["n"]]
        b = ref[j["ref_off"]:j["ref_off"] + j["m"]]
        out[k] = (oracle.dtw_global(a, b, j["exclude_last"]) if j["band_radius"] < 0
                  else oracle.dtw_banded(a, b, j["band_radius"], j["exclude_last"]))
    return out


def assert_bits_equal(got, want, what=""):
    g = np.asarray(got, np.float32).view(np.uint32)
    w = np.asarray(want, np.float32).view(np.uint32)
    bad = np.nonzero(g != w)[0]
    assert len(bad) == 0, f"{what}: {len(bad)} of {len(g)} differ, first at {bad[:5]}: got {np.asarray(got)[bad[:5]]} want {np.asarray(want)[bad[:5]]}"
