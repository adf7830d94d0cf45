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


class OracleScorer:
    """CPU scorer built on the oracle (test infrastructure): same `score(reads, opt)` interface as
    rawalign_amd.mapper.DeviceScorer, so that mapper.map_reads can run the identical control flow on the checker.
    Follows the DTW block of gen_chains, rmap.cpp:515-524."""

    def __init__(self, oracle, ref):
        self.oracle = oracle
        self.ref = ref

    def score(self, reads, opt):
        from oracle.loader import OrcOpt

        oopt = OrcOpt(opt.dtw_border_constraint, opt.dtw_fill_method, opt.dtw_band_radius_frac, opt.dtw_match_bonus,
                      opt.dtw_min_score, int(opt.fused_score))
        out = []
        for events, chains in reads:
            best = np.float32(0.0)
            kept = []
            for ch in chains:
                arr = self.ref.forward[ch.reference_sequence_index] if ch.strand == 1 else self.ref.reverse[ch.reference_sequence_index]
                s = self.oracle.align_chain(ch.anchors, arr, events, oopt, float(best))
                ch.alignment_score = float(s)
                if s >= np.float32(opt.dtw_min_score):
                    if s > best:
                        best = s
                    kept.append(ch)
            out.append(kept)
        return out

    def align_cigar(self, chain, read_events, opt):
        """rmap.cpp:715-717 on the checker: align_chain(..., cigar=true)."""
        from oracle.loader import OrcOpt
        from rawalign_amd.dtw import DtwResult

        oopt = OrcOpt(opt.dtw_border_constraint, opt.dtw_fill_method, opt.dtw_band_radius_frac, opt.dtw_match_bonus,
                      opt.dtw_min_score, int(opt.fused_score))
        arr = self.ref.forward[chain.reference_sequence_index] if chain.strand == 1 else self.ref.reverse[chain.reference_sequence_index]
        sc, cost, pi, pj, pd = self.oracle.align_chain_cigar(chain.anchors, arr, read_events, oopt)
        chain.alignment_score = float(sc)
        chain.dtw_result = DtwResult(np.float32(cost), pi, pj, pd)
        return chain


import contextlib


@contextlib.contextmanager
def planner_options(engine, **opts):
    """Set planner options on a (possibly shared) engine for the duration of a block and put the library's defaults
    back afterwards, whatever happens inside: later tests on a module-scoped engine then take the default
    (device-planned) path instead of whatever the previous test left behind."""
    defaults = {"device_plan": 1, "device_plan_min_jobs": 0}
    try:
        for k, v in opts.items():
            engine.set_option(k, v)
        yield engine
    finally:
        for k in opts:
            engine.set_option(k, defaults[k])
