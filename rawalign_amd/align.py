"""Host-side mirror of align_chain (src/rmap.cpp:181-313) and of the DTW block of gen_chains
(src/rmap.cpp:509-530), driving the GPU engine: build every chain's DTW jobs, score them in one
batch on the device, then replay the reference's sequential accept/cut logic on the host.

Names follow the reference: MapOpt fields are ri_mapopt_t's (src/roptions.h:61-65), Chain fields
are ri_chain_t's (src/rmap.h:29-46)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from ._lib import AlignOpt
from .dtw import ANCHOR_DTYPE, JOB_DTYPE, DtwResult, Engine

# src/roptions.h:13-15,21-26
RI_M_DTW_EVALUATE_CHAINS = 0x2
RI_M_DTW_OUTPUT_CIGAR = 0x4
RI_M_DTW_LOG_SCORES = 0x8
RI_M_DTW_BORDER_CONSTRAINT_GLOBAL = 0
RI_M_DTW_BORDER_CONSTRAINT_SPARSE = 1
RI_M_DTW_BORDER_CONSTRAINT_LOCAL = 2
RI_M_DTW_FILL_METHOD_FULL = 0
RI_M_DTW_FILL_METHOD_BANDED = 1


@dataclass
class MapOpt:
    """The ri_mapopt_t fields the hot path reads, with the defaults of src/roptions.c:49-53."""

    dtw_border_constraint: int = RI_M_DTW_BORDER_CONSTRAINT_SPARSE
    dtw_fill_method: int = RI_M_DTW_FILL_METHOD_BANDED
    dtw_band_radius_frac: float = 0.10
    dtw_match_bonus: float = 0.4
    dtw_min_score: float = 20.0
    flag: int = RI_M_DTW_EVALUATE_CHAINS
    # rmap.cpp:306 compiles to one fused multiply-subtract with the reference's flags on an FMA host
    fused_score: bool = True

    def c_struct(self) -> AlignOpt:
        if self.dtw_border_constraint not in (0, 1):
            # rmap.cpp:301-304: fprintf(stderr, "ERROR: invalid border constraint") + exit(EXIT_FAILURE)
            raise SystemExit("ERROR: invalid border constraint")
        return AlignOpt(self.dtw_border_constraint, self.dtw_fill_method, self.dtw_band_radius_frac,
                        self.dtw_match_bonus, self.dtw_min_score, int(self.fused_score))


@dataclass
class Chain:
    chaining_score: float
    reference_sequence_index: int
    strand: int
    anchors: np.ndarray  # ANCHOR_DTYPE, end-first: anchors[n-1] is the chain start (rmap.cpp:193-196)
    alignment_score: float = 0.0
    dtw_result: DtwResult | None = None

    @property
    def n_anchors(self) -> int:
        return len(self.anchors)


@dataclass
class ReadCandidates:
    """One read at one chunk round: its global event array and the chains gen_chains produced."""

    events: np.ndarray
    chains: list = field(default_factory=list)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def evaluation_order(engine: Engine, chaining_scores) -> np.ndarray:
    s = np.ascontiguousarray(chaining_scores, dtype=np.float32)
    perm = np.zeros(len(s), np.uint32)
    engine._check(engine.lib.rawdtw_sort_by_chaining_score(_ptr(s), len(s), _ptr(perm)))
    return perm


def evaluate_reads(engine: Engine, reads, opt: MapOpt):
    """The DTW block of gen_chains for a whole batch of reads (rmap.cpp:509-530).

    The reference reference-signal arrays must already be uploaded (engine.upload_reference).
    Returns, per read, the list of surviving chains (post_alignment_chains, in evaluation order)
    with alignment_score filled in, plus batch statistics."""
    copt = opt.c_struct()
    lib = engine.lib
    # flatten
    ev_parts, read_base = [], []
    acc = 0
    for r in reads:
        ev = np.ascontiguousarray(r.events, dtype=np.float32)
        ev_parts.append(ev)
        read_base.append(acc)
        acc += len(ev)
    events = np.concatenate(ev_parts) if ev_parts else np.zeros(0, np.float32)
    order_per_read, chain_list = [], []
    chain_off = np.zeros(len(reads) + 1, np.uint64)
    for ri, r in enumerate(reads):
        perm = evaluation_order(engine, [c.chaining_score for c in r.chains])
        order_per_read.append(perm)
        for k in perm:
            chain_list.append((ri, r.chains[int(k)]))
        chain_off[ri + 1] = len(chain_list)
    n_chains = len(chain_list)
    anchor_off = np.zeros(n_chains + 1, np.uint64)
    ref_base = np.zeros(n_chains, np.uint64)
    rbase = np.zeros(n_chains, np.uint32)
    for c, (ri, ch) in enumerate(chain_list):
        anchor_off[c + 1] = anchor_off[c] + np.uint64(ch.n_anchors)
        ref_base[c] = engine.reference_offset(ch.reference_sequence_index, ch.strand)
        rbase[c] = read_base[ri]
    anchors = (np.concatenate([np.ascontiguousarray(ch.anchors, dtype=ANCHOR_DTYPE) for _, ch in chain_list])
               if n_chains else np.zeros(0, ANCHOR_DTYPE))
    job_off = np.zeros(n_chains + 1, np.uint64)
    n_jobs = C.c_uint64()
    engine._check(lib.rawdtw_batch_build_jobs(C.byref(copt), n_chains, _ptr(anchor_off), _ptr(anchors),
                                              _ptr(ref_base), _ptr(rbase), _ptr(job_off), None, 0,
                                              C.byref(n_jobs)))
    jobs = np.zeros(n_jobs.value, JOB_DTYPE)
    engine._check(lib.rawdtw_batch_build_jobs(C.byref(copt), n_chains, _ptr(anchor_off), _ptr(anchors),
                                              _ptr(ref_base), _ptr(rbase), _ptr(job_off), _ptr(jobs),
                                              len(jobs), C.byref(n_jobs)))
    cost = engine.score_batch(jobs, events)
    score = np.zeros(n_chains, np.float32)
    keep = np.zeros(n_chains, np.uint8)
    engine._check(lib.rawdtw_batch_replay(C.byref(copt), len(reads), _ptr(chain_off), _ptr(anchor_off),
                                          _ptr(anchors), _ptr(job_off), _ptr(cost), _ptr(score), _ptr(keep)))
    out = []
    for ri in range(len(reads)):
        kept = []
        for c in range(int(chain_off[ri]), int(chain_off[ri + 1])):
            ch = chain_list[c][1]
            ch.alignment_score = float(score[c])
            if keep[c]:
                kept.append(ch)
        out.append(kept)
    stats = {"n_chains": n_chains, "n_jobs": int(n_jobs.value), "jobs": jobs, "job_cost": cost,
             "scores": score, "keep": keep}
    return out, stats


def align_chain(engine: Engine, chain: Chain, read_events, opt: MapOpt, cigar: bool = False,
                min_score: float = -1e10) -> Chain:
    """align_chain for one chain (rmap.cpp:181-313).  With cigar=True the traceback jobs run on the
    GPU and chain.dtw_result is assembled with the reference's quirks (SURVEY.md 8 a-4 i, ii)."""
    copt = opt.c_struct()
    lib = engine.lib
    anchors = np.ascontiguousarray(chain.anchors, dtype=ANCHOR_DTYPE)
    na = len(anchors)
    nj = lib.rawdtw_chain_job_count(C.byref(copt), na)
    jobs = np.zeros(max(nj, 1), JOB_DTYPE)
    ref_base = engine.reference_offset(chain.reference_sequence_index, chain.strand)
    st = lib.rawdtw_chain_build_jobs(C.byref(copt), _ptr(anchors), na, ref_base, 0, int(cigar), _ptr(jobs))
    if st == 5:
        # rmap.cpp:223-225: assert(false) //not implemented
        raise AssertionError("banded global alignment with --dtw-output-cigar is not implemented")
    engine._check(st)
    jobs = jobs[:nj]
    ev = np.ascontiguousarray(read_events, dtype=np.float32)
    if not cigar:
        cost = engine.score_batch(jobs, ev)
        chain.alignment_score = float(lib.rawdtw_chain_replay(C.byref(copt), _ptr(anchors), na, _ptr(cost),
                                                               C.c_float(min_score)))
        return chain
    results = engine.traceback_batch(jobs, ev)
    cost = np.array([r.cost for r in results], np.float32)
    # rmap.cpp:306 with the summed sub-costs; no early exit on this path (min_score defaults to -1e10)
    chain.alignment_score = float(lib.rawdtw_chain_replay(C.byref(copt), _ptr(anchors), na, _ptr(cost),
                                                           C.c_float(-1e10)))
    if opt.dtw_border_constraint == RI_M_DTW_BORDER_CONSTRAINT_GLOBAL:
        r = results[0]
        pi = r.i.astype(np.uint64)
        pj = r.j.astype(np.uint64)
        # rmap.cpp:230-233: the loop adds the anchor offsets to alignment.back() once per element
        if len(pi):
            pi[-1] += np.uint64(len(pi)) * np.uint64(anchors[na - 1]["query_position"])
            pj[-1] += np.uint64(len(pj)) * np.uint64(anchors[na - 1]["target_position"])
        chain.dtw_result = DtwResult(np.float32(r.cost), pi, pj, r.difference)
    else:
        parts = na - 1
        pis, pjs, pds = [], [], []
        total = np.float32(0.0)
        for p, r in enumerate(results):
            s = anchors[parts - p]
            pis.append(r.i.astype(np.uint64) + np.uint64(s["query_position"]))   # rmap.cpp:287
            pjs.append(r.j.astype(np.uint64) + np.uint64(s["target_position"]))  # rmap.cpp:288
            pds.append(r.difference)
            total = np.float32(total + r.cost)                                    # rmap.cpp:290
        chain.dtw_result = DtwResult(total, np.concatenate(pis) if pis else np.zeros(0, np.uint64),
                                     np.concatenate(pjs) if pjs else np.zeros(0, np.uint64),
                                     np.concatenate(pds) if pds else np.zeros(0, np.float32))
    return chain


def dtwresult_to_string(res: DtwResult) -> str:
    """rmap.cpp:580-592: "(i,j,diff)" per element, diff through ostream<<float (== %g)."""
    return "".join("(%d,%d,%s)" % (int(i), int(j), "%g" % float(d)) for i, j, d in zip(res.i, res.j, res.difference))


# ------------------------------------------------------------------------------------------------
# whole-batch form (flat arrays; what a chunk round of the mapper hands to the device)
# ------------------------------------------------------------------------------------------------
@dataclass
class CandidateBatch:
    """All candidate chains of all reads of one submission, flattened.

    Read r owns chains [chain_off[r], chain_off[r+1]) -- already in evaluation order (rmap.cpp:512);
    chain c owns anchors[anchor_off[c]:anchor_off[c+1]] (end-first); ref_base[c] / read_base[c] are
    the arena offsets of the chain's strand array and of its read's event array."""

    events: np.ndarray
    chain_off: np.ndarray
    anchor_off: np.ndarray
    anchors: np.ndarray
    ref_base: np.ndarray
    read_base: np.ndarray

    @property
    def n_reads(self):
        return len(self.chain_off) - 1

    @property
    def n_chains(self):
        return len(self.anchor_off) - 1


COMPACT_STRIDE = 8192  # RAWDTW_COMPACT_STRIDE
WIDE_STEP_DTYPE = np.dtype([("index", np.uint32), ("query_step", np.uint32), ("target_step", np.uint32)])


@dataclass
class CompactAnchors:
    """The anchor lists of a CandidateBatch in the compact hand-over form (include/rawdtw.h, rawdtw_anchors_pack): every
    chain's first entry and every COMPACT_STRIDE-th entry whole, the others as 2-byte steps back, steps >= 255 listed."""

    heads: np.ndarray
    unit_abs: np.ndarray
    steps: np.ndarray
    wide: np.ndarray

    @property
    def nbytes(self):
        return self.heads.nbytes + self.unit_abs.nbytes + self.steps.nbytes + self.wide.nbytes


def pack_anchors(lib, anchor_off, anchors) -> CompactAnchors:
    anchor_off = np.ascontiguousarray(anchor_off, np.uint64)
    anchors = np.ascontiguousarray(anchors, ANCHOR_DTYPE)
    nc, na = len(anchor_off) - 1, len(anchors)
    heads = np.zeros(max(nc, 1), ANCHOR_DTYPE)
    unit_abs = np.zeros(max((na + COMPACT_STRIDE - 1) // COMPACT_STRIDE, 1), ANCHOR_DTYPE)
    steps = np.zeros(max(na, 1), np.uint16)
    wide = np.zeros(64, WIDE_STEP_DTYPE)
    nw = C.c_uint64()
    st = lib.rawdtw_anchors_pack(nc, _ptr(anchor_off), _ptr(anchors), _ptr(heads), _ptr(unit_abs), _ptr(steps), _ptr(wide), len(wide), C.byref(nw))
    if st == 4:  # RAWDTW_ERR_RANGE: more wide steps than the list holds
        wide = np.zeros(nw.value, WIDE_STEP_DTYPE)
        st = lib.rawdtw_anchors_pack(nc, _ptr(anchor_off), _ptr(anchors), _ptr(heads), _ptr(unit_abs), _ptr(steps), _ptr(wide), len(wide), C.byref(nw))
    if st != 0:
        raise ValueError("rawdtw_anchors_pack: status %d (a chain whose positions do not descend along its list?)" % st)
    return CompactAnchors(heads[:max(nc, 0)] if nc else heads[:0], unit_abs, steps[:na], wide[:nw.value].copy())


def unpack_anchors(lib, anchor_off, ca: CompactAnchors) -> np.ndarray:
    anchor_off = np.ascontiguousarray(anchor_off, np.uint64)
    na = int(anchor_off[-1])
    out = np.zeros(max(na, 1), ANCHOR_DTYPE)
    st = lib.rawdtw_anchors_unpack(len(anchor_off) - 1, _ptr(anchor_off), _ptr(ca.heads), _ptr(ca.unit_abs), _ptr(ca.steps), _ptr(ca.wide),
                                   len(ca.wide), _ptr(out))
    if st != 0:
        raise ValueError("rawdtw_anchors_unpack: status %d" % st)
    return out[:na]


class Batch:
    """rawdtw_batch: DTW scoring + align_chain fold + per-read selection, all on the device.  compact=True hands the anchor
    lists over in the compact form (rawdtw_batch_submit_compact: the batch is then already running when this returns)."""

    def __init__(self, engine: Engine, opt: MapOpt, cb: CandidateBatch, compact: bool = False):
        self.engine = engine
        self.cb = cb
        self._copt = opt.c_struct()
        self._arrays = [
            np.ascontiguousarray(cb.chain_off, np.uint64), np.ascontiguousarray(cb.anchor_off, np.uint64),
            np.ascontiguousarray(cb.anchors, ANCHOR_DTYPE), np.ascontiguousarray(cb.ref_base, np.uint64),
            np.ascontiguousarray(cb.read_base, np.uint32),
        ]
        h = C.c_void_p()
        a = self._arrays
        self._submitted = False
        if compact:
            self.compact = ca = pack_anchors(engine.lib, a[1], a[2])
            engine._check(engine.lib.rawdtw_batch_submit_compact(engine._ctx, C.byref(self._copt), cb.n_reads, _ptr(a[0]), _ptr(a[1]),
                                                                 _ptr(ca.heads), _ptr(ca.unit_abs), _ptr(ca.steps), _ptr(ca.wide), len(ca.wide),
                                                                 _ptr(a[3]), _ptr(a[4]), C.byref(h)))
            self._submitted = True
        else:
            engine._check(engine.lib.rawdtw_batch_create(engine._ctx, C.byref(self._copt), cb.n_reads, _ptr(a[0]),
                                                         _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(a[4]), C.byref(h)))
        self._h = h
        engine._children.add(self)

    def info(self) -> dict:
        from ._lib import PlanInfo

        pi = PlanInfo()
        nc = C.c_uint64()
        self.engine._check(self.engine.lib.rawdtw_batch_info(self._h, C.byref(pi), C.byref(nc)))
        d = {k: int(getattr(pi, k)) for k, _ in PlanInfo._fields_}
        d["n_chains"] = int(nc.value)
        d["n_reads"] = self.cb.n_reads
        return d

    def build_jobs(self):
        """The batch's job list as the library builds it (rawdtw_batch_build_jobs): (jobs, job_off)."""
        from .dtw import JOB_DTYPE

        cb, lib, a = self.cb, self.engine.lib, self._arrays
        job_off = np.zeros(cb.n_chains + 1, np.uint64)
        nj = C.c_uint64()
        args = (C.byref(self._copt), cb.n_chains, _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(a[4]), _ptr(job_off))
        self.engine._check(lib.rawdtw_batch_build_jobs(*args, None, 0, C.byref(nj)))
        jobs = np.zeros(nj.value, JOB_DTYPE)
        self.engine._check(lib.rawdtw_batch_build_jobs(*args, _ptr(jobs), len(jobs), C.byref(nj)))
        return jobs, job_off

    def verify_plan(self):
        """rawdtw_batch_verify_plan: checks the tile records on the device against the job list; returns True when the
        tile class was planned on the device.  Raises RawDTWError with the first broken invariant."""
        from ._lib import RawDTWError

        jobs, _ = self.build_jobs()
        dev = C.c_int()
        msg = C.create_string_buffer(512)
        st = self.engine.lib.rawdtw_batch_verify_plan(self.engine._ctx, self._h, _ptr(jobs), len(jobs), C.byref(dev),
                                                       C.cast(msg, C.c_void_p), 512)
        if st != 0:
            raise RawDTWError(st, msg.value.decode())
        return bool(dev.value)

    def run(self):
        if self._submitted:  # (a compact batch was enqueued by its submit call)
            self._submitted = False
            return
        self.engine._check(self.engine.lib.rawdtw_batch_run(self.engine._ctx, self._h))

    def run_timed(self):
        cap = 64
        ms = np.zeros(cap, np.float32)
        kind = np.zeros(cap, np.uint32)
        n = C.c_uint32()
        self.engine._check(self.engine.lib.rawdtw_batch_run_timed(self.engine._ctx, self._h, _ptr(ms), _ptr(kind),
                                                                  cap, C.byref(n)))
        return [(int(kind[k] & 0xFF), int(kind[k] >> 8), float(ms[k])) for k in range(min(n.value, cap))]

    def run_reps(self, reps: int, timed: bool = True):
        """`reps` back-to-back runs, one sync at the end; returns [(kind, param, mean_ms)] when timed."""
        cap = 64
        ms = np.zeros(cap, np.float32)
        kind = np.zeros(cap, np.uint32)
        n = C.c_uint32()
        self.engine._check(self.engine.lib.rawdtw_batch_run_reps(
            self.engine._ctx, self._h, int(reps), _ptr(ms) if timed else None, _ptr(kind), cap, C.byref(n)))
        if not timed:
            return []
        return [(int(kind[k] & 0xFF), int(kind[k] >> 8), float(ms[k])) for k in range(min(n.value, cap))]

    def enqueue(self, timed: bool = False):
        """One run, no host synchronisation (pipelined submission)."""
        self.engine._check(self.engine.lib.rawdtw_batch_enqueue(self.engine._ctx, self._h, int(timed)))

    def collect(self):
        """Wait for the engine's stream and return [(kind, param, mean_ms)] over the timed runs enqueued."""
        cap = 64
        ms = np.zeros(cap, np.float32)
        kind = np.zeros(cap, np.uint32)
        n, runs = C.c_uint32(), C.c_uint32()
        self.engine._check(self.engine.lib.rawdtw_batch_collect(self.engine._ctx, self._h, _ptr(ms), _ptr(kind), cap,
                                                                C.byref(n), C.byref(runs)))
        return [(int(kind[k] & 0xFF), int(kind[k] >> 8), float(ms[k])) for k in range(min(n.value, cap))], runs.value

    def launch_stats(self, with_cells=True):
        out = []
        i = 0
        while True:
            kind, param = C.c_uint32(), C.c_int32()
            nj, ab, cl = C.c_uint64(), C.c_uint64(), C.c_uint64()
            st = self.engine.lib.rawdtw_batch_launch_stats(self._h, i, C.byref(kind), C.byref(param), C.byref(nj),
                                                           C.byref(ab), C.byref(cl) if with_cells else None)
            if st != 0:
                break
            out.append({"kind": int(kind.value), "param": int(param.value), "n_jobs": int(nj.value),
                        "algorithmic_bytes": int(ab.value), "cells": int(cl.value)})
            i += 1
        return out

    def fetch(self, with_job_costs=False):
        nc = self.cb.n_chains
        score = np.zeros(nc, np.float32)
        keep = np.zeros(nc, np.uint8)
        jc = np.zeros(self.info()["n_jobs"], np.float32) if with_job_costs else None
        self.engine._check(self.engine.lib.rawdtw_batch_fetch(self.engine._ctx, self._h, _ptr(score), _ptr(keep),
                                                              _ptr(jc) if jc is not None else None))
        return (score, keep, jc) if with_job_costs else (score, keep)

    def close(self):
        if getattr(self, "_h", None) is not None:
            self.engine.lib.rawdtw_batch_destroy(self._h)  # (safe in any order: rawdtw_destroy detaches live batches)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
