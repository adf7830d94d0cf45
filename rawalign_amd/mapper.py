"""Chunk-round mapper: the control flow of map_worker_for / ri_map_frag / gen_chains
(src/rmap.cpp:667-699, 545-578, 315-541) turned inside out so that every chunk round makes ONE
device submission for all active reads (option A).

Per round, for every active read (host): take the chunk's events, re-seed with the previous
chains' anchors plus the chunk's new seed hits (rmap.cpp:344-391), sort (396-401), run the chaining
DP per (sequence, strand) (430-507), order the chains by chaining score (512).  Then one
`Batch` scores every chain of every read on the device (DTW + fold + accept/cut), and the host
finishes the round: gen_primary_chains, comp_mapq, the stop rule (532-541, 692).

Event detection and seeding are NOT implemented here (they stay in RawAlign: revent.c, rsketch.c,
rawindex.cpp); a `SeedSource` supplies each chunk's events and seed hits.  `SyntheticSeeds`
imitates them for tests and demos.  The scorer is pluggable (`score(reads, opt)`); the product ships
only the device scorer -- the parity tests plug a CPU checker of their own into the same control flow
(tests/util.py) and compare PAF lines."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import mapping as M
from .align import (RI_M_DTW_EVALUATE_CHAINS, RI_M_DTW_LOG_SCORES, RI_M_DTW_OUTPUT_CIGAR, Batch, CandidateBatch, Chain,
                    MapOpt, align_chain)
from .dtw import ANCHOR_DTYPE

CARRY_DTYPE = np.dtype([("prev_src", "<u8"), ("parts", "<u4"), ("flags", "<u4"), ("start_t", "<u4"), ("start_q", "<u4")])  # rawdtw_carry_t


@dataclass
class ReadJob:
    name: str
    qlen: int                      # samples in the read
    n_chunks_available: int        # chunks the read has signal for
    events: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))  # p->events[read].values
    offset: int = 0                # reg->offset: events of the chunks that were chained (rmap.cpp:574-575)
    chains: list = field(default_factory=list)     # reg0->chains (primary chains, best first)
    chunks_done: int = 0
    finished: bool = False
    broke_early: bool = False


class SyntheticSeeds:
    """Stands in for detect_events + ri_sketch + ri_idx_get: per read and chunk, the chunk's events and
    the (sequence, strand, target, query-in-chunk) seed hits."""

    def __init__(self, ref, n_reads, seed, events_per_chunk=520, hit_prob=0.2, false_hits=25, noise_sd=0.25,
                 max_chunks=6, unmappable_frac=0.1):
        self.ref = ref
        rng = np.random.default_rng(seed)
        self.rng = rng
        self.reads = []
        lens = np.array([len(x) for x in ref.forward])
        for r in range(n_reads):
            seq = int(rng.choice(len(lens), p=lens / lens.sum()))
            strand = int(rng.integers(0, 2))
            chunks = int(rng.integers(1, max_chunks + 1))
            n_k = min(chunks * int(events_per_chunk / 1.28), lens[seq] - 1)
            start = int(rng.integers(0, lens[seq] - n_k))
            mult = rng.choice(4, size=n_k, p=(0.06, 0.66, 0.22, 0.06))
            mult[0] = max(mult[0], 1)
            t_of_ev = np.repeat(start + np.arange(n_k), mult)
            arr = ref.forward[seq] if strand == 1 else ref.reverse[seq]
            mappable = rng.random() >= unmappable_frac
            vals = arr[t_of_ev] + rng.normal(0, noise_sd, len(t_of_ev)) if mappable else rng.normal(0, 1, len(t_of_ev))
            self.reads.append(dict(seq=seq, strand=strand, t_of_ev=t_of_ev, vals=vals.astype(np.float32),
                                   mappable=mappable, n_ev=len(t_of_ev)))
        self.events_per_chunk = events_per_chunk
        self.hit_prob = hit_prob
        self.false_hits = false_hits
        self.lens = lens

    def read_job(self, r) -> ReadJob:
        rd = self.reads[r]
        n_chunks = (rd["n_ev"] + self.events_per_chunk - 1) // self.events_per_chunk
        return ReadJob(f"read_{r}", qlen=n_chunks * 4000, n_chunks_available=n_chunks)

    def chunk(self, r, c):
        """events of chunk c (z-normalised per chunk, revent.c:178-184) and its seed hits."""
        rd = self.reads[r]
        lo, hi = c * self.events_per_chunk, min(rd["n_ev"], (c + 1) * self.events_per_chunk)
        ev = rd["vals"][lo:hi].astype(np.float64)
        ev = ((ev - ev.mean()) / max(ev.std(), 1e-9)).astype(np.float32)
        rng = np.random.default_rng(hash((r, c)) & 0xFFFFFFFF)
        hits = []
        if rd["mappable"]:
            sel = np.nonzero(rng.random(hi - lo) < self.hit_prob)[0]
            for q in sel:
                hits.append((rd["seq"], rd["strand"], int(rd["t_of_ev"][lo + q]), int(q)))
        for _ in range(self.false_hits):
            s = int(rng.integers(0, len(self.lens)))
            hits.append((s, int(rng.integers(0, 2)), int(rng.integers(0, self.lens[s])), int(rng.integers(0, hi - lo))))
        return ev, hits


def score_log_line(chain) -> str:
    """rmap.cpp:308-312: sprintf("chaining_score=%f alignment_score=%f\\n", ...)."""
    return "chaining_score=%f alignment_score=%f\n" % (float(np.float32(chain.chaining_score)),
                                                        float(np.float32(chain.alignment_score)))


class DeviceScorer:
    """Scores a round's chains on the device.

    memoise=True adds the cross-chunk cache of SURVEY.md 8(f-4): the reference re-aligns every surviving
    chain from scratch in every chunk round (rmap.cpp:516-517), but a sparse part between two anchors that
    already existed in an earlier round has the same operands (events are append-only, rmap.cpp:554-567) and
    therefore the same cost.  Only unseen parts are sent to the device (always without exclude_last); the
    excluded variant is the cached cost minus the last cell's distance, which is exactly what the DTW
    functions return (dtw.cpp:514-516), so results are bit-identical to scoring everything again."""

    def __init__(self, engine, memoise: bool = False):
        self.engine = engine
        self.offs = {}
        self.memoise = memoise
        self.cache = {}       # (read key, seq, strand, t0, q0, t1, q1) -> np.float32 cost without exclusion
        self.jobs_scored = 0
        self.jobs_reused = 0

    def align_cigar(self, chain, read_events, opt: MapOpt):
        """align_chain(chains[0], ..., cigar=true) of rmap.cpp:715-717: traceback on the device."""
        return align_chain(self.engine, chain, read_events, opt, cigar=True)

    def _offset(self, ch):
        key = (ch.reference_sequence_index, ch.strand)
        if key not in self.offs:
            self.offs[key] = self.engine.reference_offset(*key)
        return self.offs[key]

    def score(self, reads, opt: MapOpt, read_keys=None):
        if self.memoise and opt.dtw_border_constraint == 1:
            return self._score_memoised(reads, opt, read_keys)
        eng = self.engine
        ev_parts, read_base, acc = [], [], 0
        for events, _ in reads:
            ev_parts.append(events)
            read_base.append(acc)
            acc += len(events)
        chain_off, anchor_off, anchors, ref_base, rbase, flat = [0], [0], [], [], [], []
        for ri, (_, chains) in enumerate(reads):
            for ch in chains:
                anchors.append(np.ascontiguousarray(ch.anchors, ANCHOR_DTYPE))
                anchor_off.append(anchor_off[-1] + len(ch.anchors))
                ref_base.append(self._offset(ch))
                rbase.append(read_base[ri])
                flat.append(ch)
            chain_off.append(len(flat))
        if not flat:
            return [[] for _ in reads]
        cb = CandidateBatch(np.concatenate(ev_parts), np.array(chain_off, np.uint64), np.array(anchor_off, np.uint64),
                            np.concatenate(anchors), np.array(ref_base, np.uint64), np.array(rbase, np.uint32))
        eng.upload_events(cb.events)
        b = Batch(eng, opt, cb)
        self.jobs_scored += b.info()["n_jobs"]
        b.run()
        score, keep = b.fetch()
        b.close()
        out = []
        for ri in range(len(reads)):
            kept = []
            for c in range(chain_off[ri], chain_off[ri + 1]):
                flat[c].alignment_score = float(score[c])
                if keep[c]:
                    kept.append(flat[c])
            out.append(kept)
        return out

    def _score_memoised(self, reads, opt: MapOpt, read_keys):
        import ctypes as C

        from .dtw import JOB_DTYPE

        eng, lib = self.engine, self.engine.lib
        copt = opt.c_struct()
        banded = opt.dtw_fill_method != 0
        ev_parts, read_base, acc = [], [], 0
        for events, _ in reads:
            ev_parts.append(events)
            read_base.append(acc)
            acc += len(events)
        events_cat = np.concatenate(ev_parts) if ev_parts else np.zeros(0, np.float32)
        new_jobs, new_keys = [], []
        plans = []   # per read: list of (chain, [part keys in align order], [exclude flags], [(a_last index, b arrays)])
        for ri, (events, chains) in enumerate(reads):
            rk = read_keys[ri] if read_keys is not None else ri
            per_read = []
            for ch in chains:
                a = ch.anchors
                parts = len(a) - 1
                keys = []
                for p in range(parts):
                    s, e = a[parts - p], a[parts - p - 1]
                    key = (rk, ch.reference_sequence_index, ch.strand, int(s["target_position"]), int(s["query_position"]),
                           int(e["target_position"]), int(e["query_position"]))
                    keys.append(key)
                    if key not in self.cache:
                        self.cache[key] = None  # pending
                        n = int(e["query_position"]) - int(s["query_position"]) + 1
                        m = int(e["target_position"]) - int(s["target_position"]) + 1
                        R0 = max(1, int(np.float32(n) * np.float32(opt.dtw_band_radius_frac))) if banded else -1
                        new_jobs.append((self._offset(ch) + int(s["target_position"]), read_base[ri] + int(s["query_position"]),
                                         n, m, R0, 0, 0))
                        new_keys.append(key)
                    else:
                        self.jobs_reused += 1
                per_read.append((ch, keys))
            plans.append(per_read)
        if new_jobs:
            jobs = np.array(new_jobs, dtype=JOB_DTYPE)
            costs = eng.score_batch(jobs, events_cat)
            self.jobs_scored += len(jobs)
            for k, c in zip(new_keys, costs):
                self.cache[k] = np.float32(c)
        # compose the per-part costs the reference would have computed and fold them on the host
        out = []
        arena_cache = {}
        for ri, (events, chains) in enumerate(reads):
            best = np.float32(0.0)
            kept = []
            for ch, keys in plans[ri]:
                a = np.ascontiguousarray(ch.anchors, ANCHOR_DTYPE)
                parts = len(a) - 1
                costs = np.zeros(max(parts, 1), np.float32)
                for p, key in enumerate(keys):
                    c = self.cache[key]
                    if p != parts - 1:  # exclude_last_element (rmap.cpp:270): minus the last cell's distance
                        q1, t1 = key[6], key[5]
                        akey = (ch.reference_sequence_index, ch.strand)
                        if akey not in arena_cache:
                            arena_cache[akey] = self._ref_array(ch)
                        d = np.float32(abs(np.float32(events[q1]) - np.float32(arena_cache[akey][t1])))
                        c = np.float32(c - d)
                    costs[p] = c
                s = np.float32(lib.rawdtw_chain_replay(C.byref(copt), a.ctypes.data_as(C.c_void_p), len(a),
                                                        costs.ctypes.data_as(C.c_void_p), C.c_float(float(best))))
                ch.alignment_score = float(s)
                if s >= np.float32(opt.dtw_min_score):
                    if s > best:
                        best = s
                    kept.append(ch)
            out.append(kept)
        return out

    def _ref_array(self, ch):
        """Host copy of a strand's signal array (for the one subtraction per excluded part)."""
        if not hasattr(self, "ref_host"):
            raise RuntimeError("memoised scoring needs DeviceScorer.ref_host = synth.Reference / index signals")
        r = self.ref_host
        return r.forward[ch.reference_sequence_index] if ch.strand == 1 else r.reverse[ch.reference_sequence_index]


class RoundScorer:
    """Chunk rounds scored on the device with the part costs carried over from round to round ON the device (SURVEY.md 8 f-4,
    rawdtw_batch_submit_carry): every read keeps a slot in the event arena and only its new events are uploaded
    (rawdtw_events_append, rmap.cpp:554-567 is append-only); the batch of the round before stays resident until the next one
    has taken what it can from it; per chain the host names the previous chain it continues and the leading parts that
    did not change (rawdtw_round_match_chains), and only the NEW anchors are handed over (rawdtw_batch_submit_carry).  Results are bit-identical to scoring every round from scratch (rmap.cpp:516-517)."""

    memoise = True

    def __init__(self, engine, slot_events: int, n_slots: int):
        self.engine = engine
        self.slot_events = int(slot_events)
        self.n_slots = int(n_slots)
        self.offs = {}
        self.slot_of = {}        # read key -> slot
        self.uploaded = {}       # read key -> events already in its slot
        self.prev = None         # (Batch handle, arrays, {read key: read index})
        self.jobs_scored = 0
        self.jobs_reused = 0
        self.anchors_sent = 0    # anchors handed to the device over all rounds (a carried round sends its new ones only)
        engine._check(engine.lib.rawdtw_events_reserve(engine._ctx, self.slot_events * self.n_slots))

    def align_cigar(self, chain, read_events, opt: MapOpt):
        return align_chain(self.engine, chain, read_events, opt, cigar=True)

    def _offset(self, ch):
        key = (ch.reference_sequence_index, ch.strand)
        if key not in self.offs:
            self.offs[key] = self.engine.reference_offset(*key)
        return self.offs[key]

    def score(self, reads, opt: MapOpt, read_keys=None):
        import ctypes as C

        eng, lib = self.engine, self.engine.lib
        keys = list(read_keys) if read_keys is not None else list(range(len(reads)))
        # ---- the round's new events into the reads' slots ----
        new_parts, seg_src, seg_dst = [], [0], []
        for key, (events, _) in zip(keys, reads):
            if key not in self.slot_of:
                assert len(self.slot_of) < self.n_slots, "more reads than slots"
                self.slot_of[key] = len(self.slot_of)
                self.uploaded[key] = 0
            assert len(events) <= self.slot_events, "a read outgrew its slot"
            done = self.uploaded[key]
            if len(events) > done:
                new_parts.append(np.ascontiguousarray(events[done:], np.float32))
                seg_src.append(seg_src[-1] + len(events) - done)
                seg_dst.append(self.slot_of[key] * self.slot_events + done)
                self.uploaded[key] = len(events)
        keep_alive = None
        if new_parts:
            h_new = np.concatenate(new_parts)
            src = np.array(seg_src, np.uint64)
            dst = np.array(seg_dst, np.uint32)
            eng._check(lib.rawdtw_events_append(eng._ctx, _vp(h_new), len(h_new), len(dst), _vp(src), _vp(dst)))
            keep_alive = (h_new, src, dst)
        # ---- the round's chains ----
        chain_off, anchor_off, anchors, ref_base, rbase, flat = [0], [0], [], [], [], []
        for key, (_, chains) in zip(keys, reads):
            for ch in chains:
                anchors.append(np.ascontiguousarray(ch.anchors, ANCHOR_DTYPE))
                anchor_off.append(anchor_off[-1] + len(ch.anchors))
                ref_base.append(self._offset(ch))
                rbase.append(self.slot_of[key] * self.slot_events)
                flat.append(ch)
            chain_off.append(len(flat))
        arrays = dict(chain_off=np.array(chain_off, np.uint64), anchor_off=np.array(anchor_off, np.uint64),
                      anchors=np.concatenate(anchors) if anchors else np.zeros(0, ANCHOR_DTYPE),
                      ref_base=np.array(ref_base, np.uint64), read_base=np.array(rbase, np.uint32))
        n_chains = len(flat)
        copt = opt.c_struct()
        h = C.c_void_p()
        carried = False
        if self.prev is not None and n_chains and lib.rawdtw_batch_can_carry(eng._ctx, self.prev[0], C.byref(copt)):
            prev_h, pa, pidx = self.prev
            prev_read = np.array([pidx.get(k, 0xFFFFFFFFFFFFFFFF) for k in keys], np.uint64)
            carry = np.zeros(n_chains, CARRY_DTYPE)
            new_off = np.zeros(n_chains + 1, np.uint64)
            new_anchors = np.zeros(max(len(arrays["anchors"]), 1), ANCHOR_DTYPE)
            eng._check(lib.rawdtw_round_match_chains(len(reads), _vp(arrays["chain_off"]), _vp(arrays["anchor_off"]), _vp(arrays["anchors"]),
                                                     _vp(arrays["ref_base"]), _vp(arrays["read_base"]), _vp(prev_read), _vp(pa["chain_off"]),
                                                     _vp(pa["anchor_off"]), _vp(pa["anchors"]), _vp(pa["ref_base"]), _vp(pa["read_base"]), _vp(carry),
                                                     _vp(new_off), _vp(new_anchors)))
            eng._check(lib.rawdtw_batch_submit_carry(eng._ctx, C.byref(copt), len(reads), _vp(arrays["chain_off"]), _vp(arrays["anchor_off"]),
                                                     _vp(arrays["anchors"]), _vp(new_off), _vp(new_anchors), _vp(arrays["ref_base"]),
                                                     _vp(arrays["read_base"]), prev_h, _vp(carry), C.byref(h)))
            keep_alive = (keep_alive, carry, new_off, new_anchors)
            self.anchors_sent += int(new_off[-1])
            carried = True
        if not carried:
            a_ = arrays["anchors"] if len(arrays["anchors"]) else np.zeros(1, ANCHOR_DTYPE)
            rb_ = arrays["ref_base"] if n_chains else np.zeros(1, np.uint64)
            qb_ = arrays["read_base"] if n_chains else np.zeros(1, np.uint32)
            eng._check(lib.rawdtw_batch_submit(eng._ctx, C.byref(copt), len(reads), _vp(arrays["chain_off"]), _vp(arrays["anchor_off"]), _vp(a_),
                                               _vp(rb_), _vp(qb_), C.byref(h)))
            keep_alive = (keep_alive, a_, rb_, qb_)
            self.anchors_sent += len(arrays["anchors"])
        score = np.zeros(max(n_chains, 1), np.float32)
        keep = np.zeros(max(n_chains, 1), np.uint8)
        eng._check(lib.rawdtw_batch_fetch(eng._ctx, h, _vp(score), _vp(keep), None))
        del keep_alive
        sc, ru = C.c_uint64(), C.c_uint64()
        eng._check(lib.rawdtw_batch_round_stats(eng._ctx, h, C.byref(sc), C.byref(ru)))
        self.jobs_scored += sc.value
        self.jobs_reused += ru.value
        if self.prev is not None:
            lib.rawdtw_batch_destroy(self.prev[0])
        self.prev = (h, arrays, {k: i for i, k in enumerate(keys)})
        out = []
        for ri in range(len(reads)):
            kept = []
            for c in range(chain_off[ri], chain_off[ri + 1]):
                flat[c].alignment_score = float(score[c])
                if keep[c]:
                    kept.append(flat[c])
            out.append(kept)
        return out

    def close(self):
        if self.prev is not None:
            self.engine.lib.rawdtw_batch_destroy(self.prev[0])
            self.prev = None


def _vp(a):
    import ctypes as C

    return C.c_void_p(a.ctypes.data)


def map_reads(seeds, read_ids, scorer, opt: MapOpt, stop: M.StopOpt = M.StopOpt(), e: int = 6, log=None):
    """Runs chunk rounds until every read stopped; returns the PAF lines in read order.

    Flags (src/roptions.h:13-15): DTW runs when RI_M_DTW_EVALUATE_CHAINS or RI_M_DTW_LOG_SCORES is set
    (rmap.cpp:509); the chain list is replaced by the surviving chains only under EVALUATE_CHAINS
    (rmap.cpp:525) -- with LOG_SCORES alone every chain stays, carrying its alignment score (which the
    sort of gen_primary_chains then sees first, rmap.h:41-45).  With RI_M_DTW_OUTPUT_CIGAR the best chain of
    a mapped read is aligned once more with traceback (rmap.cpp:715-717).  `log`, when given, receives the
    lines --dtw-log-scores writes to stderr (rmap.cpp:308-312), in order."""
    copt = M.default_chain_opt(e)
    jobs = {r: seeds.read_job(r) for r in read_ids}
    names = [f"seq{s}" for s in range(len(seeds.lens))]
    rounds = 0
    while True:
        active = [r for r in read_ids if not jobs[r].finished]
        if not active:
            break
        rounds += 1
        submission = []
        skipped = set()
        for r in active:
            rj = jobs[r]
            ev, hits = seeds.chunk(r, rj.chunks_done)
            rj.events = np.concatenate([rj.events, ev])        # rmap.cpp:554-567
            if len(ev) < stop.min_events:                      # rmap.cpp:569-572: no gen_chains, chains and offset stay
                skipped.add(r)
                submission.append((rj.events, []))
                continue
            chunk_start = rj.offset                            # reg->offset (rmap.cpp:574)
            rj.offset += len(ev)                               # rmap.cpp:575
            per = {}
            for ch in rj.chains:                               # rmap.cpp:344-357: re-seed with previous anchors
                per.setdefault((ch.reference_sequence_index, ch.strand), []).extend(
                    (int(a["target_position"]), int(a["query_position"])) for a in ch.anchors)
            for s, st, t, q in hits:                           # rmap.cpp:371-391
                per.setdefault((s, st), []).append((t, q + chunk_start))
            chains, maxs = [], 0.0
            for s in range(len(seeds.lens)):                   # rmap.cpp:432-433: sequence-major, strand 0 then 1
                for st in (0, 1):
                    lst = per.get((s, st))
                    if not lst:
                        continue
                    a = np.array(sorted(lst), dtype=[("target_position", "<u4"), ("query_position", "<u4")])
                    cs, maxs = M.chain_anchors(a.astype(ANCHOR_DTYPE), copt, maxs, s, st)
                    chains.extend(cs)
            from .align import evaluation_order

            if chains:
                order = evaluation_order(scorer.engine if hasattr(scorer, "engine") else _SortHelper.get(),
                                         [c.chaining_score for c in chains])
                chains = [chains[int(k)] for k in order]
            submission.append((rj.events, chains))
        runs_dtw = bool(opt.flag & (RI_M_DTW_EVALUATE_CHAINS | RI_M_DTW_LOG_SCORES))  # rmap.cpp:509
        if runs_dtw:
            kept = scorer.score(submission, opt, read_keys=active) if getattr(scorer, "memoise", False) else scorer.score(submission, opt)
            if log is not None and (opt.flag & RI_M_DTW_LOG_SCORES):
                for _, chains in submission:       # evaluation order; a cut chain returns before the fprintf
                    log.extend(score_log_line(c) for c in chains if np.float32(c.alignment_score) != np.float32(-1e10))
        if not (opt.flag & RI_M_DTW_EVALUATE_CHAINS):
            kept = [chains for _, chains in submission]                                # rmap.cpp:525: list not replaced
        for r, post in zip(active, kept):
            rj = jobs[r]
            if r not in skipped:
                rj.chains = M.gen_primary_chains(post, opt, stop) if post else []
            rj.chunks_done += 1
            if M.is_mapped_with_high_confidence(rj.chains, opt, stop):   # rmap.cpp:692
                rj.finished, rj.broke_early = True, True
            elif rj.chunks_done >= min(rj.n_chunks_available, stop.max_num_chunk):
                rj.finished = True
    lines = []
    for r in read_ids:
        rj = jobs[r]
        if (opt.flag & RI_M_DTW_OUTPUT_CIGAR) and M.is_mapped_with_high_confidence(rj.chains, opt, stop):
            scorer.align_cigar(rj.chains[0], rj.events, opt)                            # rmap.cpp:715-717
            if log is not None and (opt.flag & RI_M_DTW_LOG_SCORES):
                log.append(score_log_line(rj.chains[0]))
        rs = M.ReadState(rj.name, rj.qlen, rj.offset, rj.chunks_done if not rj.broke_early else rj.chunks_done - 1,
                         rj.broke_early, 0.0, rj.chains)
        lines.append(M.paf_line(rs, names, [int(x) for x in seeds.lens], opt, stop))
    return lines, rounds


class _SortHelper:
    """evaluation_order needs only the library handle."""
    _inst = None

    @classmethod
    def get(cls):
        if cls._inst is None:
            from ._lib import load_library

            class _E:
                lib = load_library()

                @staticmethod
                def _check(st):
                    assert st == 0
            cls._inst = _E()
        return cls._inst


HIT_DTYPE = np.dtype([("ref_seq", "<u4"), ("strand", "<i4"), ("target_position", "<u4"), ("query_position", "<u4")])  # rawdtw_seed_hit_t


class CMapper:
    """The chunk-round mapper inside the library (rawdtw_mapper_*, rawalign_amd/csrc/rawdtw_mapper.cpp) behind ctypes: what a
    RawAlign maintainer's binding calls.  `engine` may be None for a mapper that scores through
    `set_scorer` only (CPU harnesses)."""

    def __init__(self, engine, opt: MapOpt, stop: M.StopOpt, seq_names, seq_lens, slot_events: int, max_reads: int, carry: bool = True,
                 threads: int = 1, groups: int = 1, e: int = 6, device_chain: bool = False):
        import ctypes as C

        from ._lib import MapperOpt, load_library

        self.lib = engine.lib if engine is not None else load_library()
        self.engine = engine
        mo = MapperOpt()
        mo.flag = opt.flag
        mo.align = opt.c_struct()
        mo.chain = M.default_chain_opt(e)
        mo.min_bestmap_ratio, mo.min_meanmap_ratio, mo.min_chain_anchor = stop.min_bestmap_ratio, stop.min_meanmap_ratio, stop.min_chain_anchor
        mo.bp_per_sec, mo.sample_rate, mo.chunk_size, mo.max_num_chunk = stop.bp_per_sec, stop.sample_rate, stop.chunk_size, stop.max_num_chunk
        mo.slot_events, mo.max_reads, mo.carry, mo.min_events = int(slot_events), int(max_reads), int(bool(carry)), int(stop.min_events)
        mo.threads, mo.groups, mo.device_chain = int(threads), int(groups), int(bool(device_chain))
        names = (C.c_char_p * len(seq_names))(*[n.encode() for n in seq_names])
        lens = np.ascontiguousarray(seq_lens, np.uint32)
        self._h = C.c_void_p()
        st = self.lib.rawdtw_mapper_create(engine._ctx if engine is not None else None, C.byref(mo), len(seq_names), names, _vp(lens), C.byref(self._h))
        if st != 0:
            raise RuntimeError(f"rawdtw_mapper_create -> {st}")
        self._cb = None

    def _check(self, st):
        if st != 0:
            raise RuntimeError(f"rawdtw_mapper status {st}: {self.lib.rawdtw_mapper_last_error(self._h).decode()}")

    def add_read(self, name: str, qlen: int, n_chunks: int) -> int:
        import ctypes as C

        rid = C.c_uint32()
        self._check(self.lib.rawdtw_mapper_add_read(self._h, name.encode(), int(qlen), int(n_chunks), C.byref(rid)))
        return rid.value

    def release_read(self, rid: int):
        self._check(self.lib.rawdtw_mapper_release_read(self._h, int(rid)))

    def round_arrays(self, read_ids, event_off, events, hit_off, hits):
        """one chunk round from flat arrays (read_ids u32, event_off / hit_off u64, events f32, hits HIT_DTYPE)"""
        self._check(self.lib.rawdtw_mapper_round(self._h, len(read_ids), _vp(read_ids), _vp(event_off), _vp(events), _vp(hit_off), _vp(hits)))

    def round(self, read_ids, chunks):
        """chunks[k] = (events, hits as (seq, strand, target, query) tuples) of read read_ids[k]"""
        ids = np.ascontiguousarray(read_ids, np.uint32)
        eoff = np.zeros(len(ids) + 1, np.uint64)
        hoff = np.zeros(len(ids) + 1, np.uint64)
        for k, (ev, hits) in enumerate(chunks):
            eoff[k + 1] = eoff[k] + len(ev)
            hoff[k + 1] = hoff[k] + len(hits)
        ev = np.concatenate([np.ascontiguousarray(c[0], np.float32) for c in chunks] + [np.zeros(1, np.float32)])
        hits = np.zeros(int(hoff[-1]) + 1, HIT_DTYPE)
        at = 0
        for _, hs in chunks:
            for s, st, t, q in hs:
                hits[at] = (s, st, t, q)
                at += 1
        self.round_arrays(ids, eoff, ev, hoff, hits)

    def state(self, rid: int):
        import ctypes as C

        fin, done = C.c_int(), C.c_uint32()
        self._check(self.lib.rawdtw_mapper_read_state(self._h, int(rid), C.byref(fin), C.byref(done)))
        return bool(fin.value), done.value

    def finish(self):
        return self.lib.rawdtw_mapper_finish(self._h)

    def paf(self, rid: int) -> str:
        import ctypes as C

        n = C.c_uint32()
        buf = C.create_string_buffer(1 << 16)
        st = self.lib.rawdtw_mapper_paf(self._h, int(rid), buf, len(buf), C.byref(n))
        if st == 4:  # RAWDTW_ERR_RANGE: the line is longer
            buf = C.create_string_buffer(n.value + 1)
            st = self.lib.rawdtw_mapper_paf(self._h, int(rid), buf, len(buf), C.byref(n))
        self._check(st)
        return buf.value.decode()

    def log(self):
        import ctypes as C

        p = C.c_char_p()
        self._check(self.lib.rawdtw_mapper_log(self._h, C.byref(p)))
        return (p.value or b"").decode()

    def stats(self):
        import ctypes as C

        r, s, u = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.rawdtw_mapper_stats(self._h, C.byref(r), C.byref(s), C.byref(u)))
        return r.value, s.value, u.value

    def timing(self):
        t = np.zeros(8, np.float64)
        self._check(self.lib.rawdtw_mapper_timing(self._h, _vp(t)))
        return dict(host_phase_ms=t[0], layout_ms=t[1], submit_ms=t[2], fetch_wait_ms=t[3], round_end_ms=t[4], anchor_bytes=int(t[5]),
                    event_bytes=int(t[6]), other_bytes=int(t[7]))

    def set_scorer(self, fn):
        """fn(chain_off, anchor_off, anchors, chain_seq, chain_strand, read_events: list of arrays) -> (score f32, keep u8);
        harnesses only (bench.py's cpu_baseline, the CPU tests) -- see rawdtw_mapper_set_scorer"""
        import ctypes as C

        from ._lib import SCORER_FN

        def cb(_user, n_reads, p_coff, p_aoff, p_anch, p_seq, p_strand, p_ev, p_nev, p_score, p_keep):
            try:
                n_reads = int(n_reads)
                coff = np.ctypeslib.as_array(C.cast(p_coff, C.POINTER(C.c_uint64)), (n_reads + 1,)).copy()
                nc = int(coff[-1])
                aoff = np.ctypeslib.as_array(C.cast(p_aoff, C.POINTER(C.c_uint64)), (nc + 1,)).copy()
                na = int(aoff[-1])
                anch = np.frombuffer((C.c_char * (max(na, 1) * 8)).from_address(p_anch), ANCHOR_DTYPE, na).copy()
                seq = np.ctypeslib.as_array(C.cast(p_seq, C.POINTER(C.c_uint32)), (max(nc, 1),))[:nc].copy() if nc else np.zeros(0, np.uint32)
                strand = np.ctypeslib.as_array(C.cast(p_strand, C.POINTER(C.c_int32)), (max(nc, 1),))[:nc].copy() if nc else np.zeros(0, np.int32)
                nev = np.ctypeslib.as_array(C.cast(p_nev, C.POINTER(C.c_uint32)), (n_reads,))
                evp = C.cast(p_ev, C.POINTER(C.c_void_p))
                evs = [np.ctypeslib.as_array(C.cast(evp[r], C.POINTER(C.c_float)), (max(int(nev[r]), 1),))[:int(nev[r])] for r in range(n_reads)]
                score, keep = fn(coff, aoff, anch, seq, strand, evs)
                if nc:
                    np.ctypeslib.as_array(C.cast(p_score, C.POINTER(C.c_float)), (nc,))[:] = score
                    np.ctypeslib.as_array(C.cast(p_keep, C.POINTER(C.c_uint8)), (nc,))[:] = keep
                return 0
            except Exception:  # noqa: BLE001 -- nothing may propagate through the C frames
                import traceback

                traceback.print_exc()
                return 1
        self._cb = SCORER_FN(cb) if fn is not None else None
        self._check(self.lib.rawdtw_mapper_set_scorer(self._h, self._cb, None))

    def set_scorer_c(self, fn_ptr, user_ptr):
        """a scorer that is C code itself (address of a rawdtw_scorer_fn, its user pointer): no Python between the mapper and it"""
        self._check(self.lib.rawdtw_mapper_set_scorer(self._h, fn_ptr, user_ptr))

    def close(self):
        if self._h:
            self.lib.rawdtw_mapper_destroy(self._h)
            self._h = None


def map_reads_c(seeds, read_ids, cm: CMapper):
    """map_reads through the library's mapper: chunk rounds until every read stopped; the PAF lines in read order, the rounds."""
    jobs = {r: seeds.read_job(r) for r in read_ids}
    ids = {r: cm.add_read(jobs[r].name, jobs[r].qlen, jobs[r].n_chunks_available) for r in read_ids}
    rounds = 0
    while True:
        act, chunks = [], []
        for r in read_ids:
            fin, done = cm.state(ids[r])
            if fin or done >= jobs[r].n_chunks_available:
                continue
            act.append(ids[r])
            chunks.append(seeds.chunk(r, done))
        if not act:
            break
        cm.round(act, chunks)
        rounds += 1
    st = cm.finish()
    if st != 0:
        raise RuntimeError(f"rawdtw_mapper_finish -> {st}")
    return [cm.paf(ids[r]) for r in read_ids], rounds
