"""Seeded synthetic inputs for the DTW hot path: no real data, ONT pore model or
FAST5 reader exists in this environment, so bench.py and the larger tests build their inputs here.

* genome          : uniform random ACGT
* pore model      : 4^k level means ~ N(90, 12^2) pA clipped to [55, 135] (r9.4-like 6-mer table)
* reference signal: what ri_seq_to_sig computes (src/rsig.cpp:7-41): k-mer level lookup along each
                    strand, z-normalised per sequence in double precision, stored as float
* read events     : the reference signal along a random window, each k-mer emitting 0-3 events
                    (skips / over-segmentation), Gaussian noise, z-normalised per read
* candidate chains: a true chain whose anchors lie on the read's real path (seed hits with a given
                    hit probability) plus decoy chains at random reference positions, listed in
                    evaluation order (true chain first = highest chaining score)

Event detection, seeding and chaining themselves stay on the host in RawAlign and are out of scope
here; this module only imitates their OUTPUT so that the job-shape mix
(thousands of 2..50-event segments plus occasional long decoy segments) is realistic."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .align import CandidateBatch
from .dtw import ANCHOR_DTYPE


def make_pore_model(k: int = 6, seed: int = 20231005) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return np.clip(rng.normal(90.0, 12.0, 4 ** k), 55.0, 135.0).astype(np.float32)


def make_genome(n_bp: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 4, n_bp, dtype=np.uint8)


def seq_to_sig(codes: np.ndarray, pore: np.ndarray, k: int = 6, strand: int = 0) -> np.ndarray:
    """ri_seq_to_sig (src/rsig.cpp:7-41) for an unambiguous sequence given as 0..3 codes."""
    seq = (3 - codes[::-1]) if strand else codes  # rsig.cpp:15,22: reverse traversal, complemented base
    n = len(seq) - k + 1
    if n > (1 << 26):
        return _seq_to_sig_chunked(seq, pore, k, n)
    idx = np.zeros(n, np.int64)
    for j in range(k):
        idx = (idx << 2) | seq[j:j + n].astype(np.int64)
    vals = pore[idx].astype(np.float64)  # rsig.cpp:28: curval is a double
    mean = vals.sum() / n                # rsig.cpp:34
    std = np.sqrt((vals * vals).sum() / n - mean * mean)
    return ((pore[idx].astype(np.float32).astype(np.float64) - mean) / std).astype(np.float32)  # rsig.cpp:37-38


def _seq_to_sig_chunked(seq, pore, k, n, chunk=1 << 25):
    """The same for a human-size sequence (gigabases): two passes over chunks, so that the k-mer indices and the doubles of one
    chunk are all that is alive at a time (the sums of a chunked pass differ from one pairwise sum in their last bits: the
    whole-array form above stays the one the small fixtures use)."""
    def levels(lo, hi):
        idx = np.zeros(hi - lo, np.int32)
        for j in range(k):
            idx = (idx << 2) | seq[lo + j:hi + j].astype(np.int32)
        return pore[idx]
    s1 = s2 = 0.0
    for lo in range(0, n, chunk):
        v = levels(lo, min(n, lo + chunk)).astype(np.float64)
        s1 += float(v.sum()); s2 += float((v * v).sum())
    mean = s1 / n
    std = np.sqrt(s2 / n - mean * mean)
    out = np.empty(n, np.float32)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        out[lo:hi] = ((levels(lo, hi).astype(np.float64) - mean) / std).astype(np.float32)
    return out


@dataclass
class Reference:
    forward: list  # per sequence, float32
    reverse: list
    names: list

    @property
    def n_seq(self):
        return len(self.forward)


def make_reference(seq_lengths, seed: int, k: int = 6) -> Reference:
    """RAWDTW_SYNTH_CACHE=<dir>: keep the signal arrays of references of 10^8 bases and more there (a gigabase reference takes
    minutes to make; the profiling passes of one GPU-box call make the same one several times)."""
    import os

    pore = make_pore_model(k)
    fwd, rev, names = [], [], []
    cache = os.environ.get("RAWDTW_SYNTH_CACHE")
    for s, n in enumerate(seq_lengths):
        path = os.path.join(cache, "ref_%d_%d_%d_%d.npy" % (int(n), seed, s, k)) if cache and int(n) >= 10 ** 8 else None
        if path and os.path.exists(path):
            both = np.load(path, mmap_mode="r")
            fwd.append(np.ascontiguousarray(both[0])); rev.append(np.ascontiguousarray(both[1]))
        else:
            g = make_genome(int(n), seed + 1000 * s)
            fwd.append(seq_to_sig(g, pore, k, 0))
            rev.append(seq_to_sig(g, pore, k, 1))
            if path:
                os.makedirs(cache, exist_ok=True)
                np.save(path, np.stack([fwd[-1], rev[-1]]))
        names.append(f"synth_{s}")
    return Reference(fwd, rev, names)


def _segment_arange(lengths: np.ndarray) -> np.ndarray:
    """concatenate(arange(l) for l in lengths)"""
    lengths = np.asarray(lengths, np.int64)
    total = int(lengths.sum())
    starts = np.cumsum(lengths) - lengths
    return np.arange(total, dtype=np.int64) - np.repeat(starts, lengths)


@dataclass
class SynthParams:
    n_reads: int = 4096
    bases_per_chunk: int = 450      # 4000 samples / (4000/450 samples per base), roptions.c:9-11
    max_chunks: int = 12            # chunk rounds a read has been through (reference cap: 30, roptions.c:24)
    mean_chunks: float = 3.0
    event_mult_probs: tuple = (0.06, 0.66, 0.22, 0.06)  # events emitted per k-mer: 0,1,2,3
    noise_sd: float = 0.25
    hit_prob: float = 0.20          # fraction of read events that become anchors of the true chain
    unmappable_frac: float = 0.05
    decoys_per_read: float = 1.5
    decoy_gap_median: float = 12.0
    decoy_gap_sigma: float = 1.0
    decoy_gap_max: int = 800


def make_candidate_batch(ref: Reference, ref_offsets, params: SynthParams, seed: int):
    """ref_offsets[(seq, strand)] -> arena offset (Engine.reference_offset).  Returns
    (CandidateBatch, info dict)."""
    rng = np.random.default_rng(seed)
    P = params
    R = P.n_reads
    seq_len = np.array([len(x) for x in ref.forward], np.int64)
    seq_of_read = rng.choice(ref.n_seq, size=R, p=seq_len / seq_len.sum())
    strand = rng.integers(0, 2, R)
    chunks = np.clip(rng.geometric(1.0 / P.mean_chunks, R), 1, P.max_chunks)
    n_k = np.minimum(chunks * P.bases_per_chunk, seq_len[seq_of_read] - 1).astype(np.int64)
    start = (rng.random(R) * (seq_len[seq_of_read] - n_k)).astype(np.int64)
    mappable = rng.random(R) >= P.unmappable_frac

    # ---- events ------------------------------------------------------------------------------
    read_of_k = np.repeat(np.arange(R), n_k)
    kpos = np.repeat(start, n_k) + _segment_arange(n_k)       # position in the strand's signal array
    mult = rng.choice(4, size=len(kpos), p=P.event_mult_probs)
    # every read keeps at least its first k-mer
    first_k = np.cumsum(n_k) - n_k
    mult[first_k] = np.maximum(mult[first_k], 1)
    t_of_ev = np.repeat(kpos, mult)                             # true target position of each event
    read_of_ev = np.repeat(read_of_k, mult)
    n_ev = np.bincount(read_of_ev, minlength=R).astype(np.int64)
    ev_off = np.concatenate([[0], np.cumsum(n_ev)])
    q_of_ev = np.arange(len(t_of_ev), dtype=np.int64) - ev_off[read_of_ev]  # global query position in its read

    clean = np.empty(len(t_of_ev), np.float32)
    for s in range(ref.n_seq):
        for st in (0, 1):
            sel = (seq_of_read[read_of_ev] == s) & (strand[read_of_ev] == st)
            arr = ref.forward[s] if st == 1 else ref.reverse[s]  # strand==1 -> forward_signals (rmap.cpp:182-188)
            clean[sel] = arr[t_of_ev[sel]]
    vals = clean.astype(np.float64) + rng.normal(0.0, P.noise_sd, len(clean))
    unm = ~mappable[read_of_ev]
    vals[unm] = rng.normal(0.0, 1.0, int(unm.sum()))
    # z-normalise per read (the mapper does it per chunk, revent.c:178-184)
    s1 = np.add.reduceat(vals, ev_off[:-1])
    s2 = np.add.reduceat(vals * vals, ev_off[:-1])
    mean = s1 / n_ev
    sd = np.sqrt(np.maximum(s2 / n_ev - mean * mean, 1e-12))
    events = ((vals - mean[read_of_ev]) / sd[read_of_ev]).astype(np.float32)

    # ---- true chains: seed hits on the real path ----------------------------------------------
    hit = (rng.random(len(t_of_ev)) < P.hit_prob) & mappable[read_of_ev]
    hi = np.nonzero(hit)[0]
    # the chaining DP needs strictly increasing target positions (rmap.cpp:453-455): drop repeats
    same = np.zeros(len(hi), bool)
    same[1:] = (t_of_ev[hi[1:]] == t_of_ev[hi[:-1]]) & (read_of_ev[hi[1:]] == read_of_ev[hi[:-1]])
    hi = hi[~same]
    tr_read = read_of_ev[hi]
    tr_cnt = np.bincount(tr_read, minlength=R)
    ok = tr_cnt[tr_read] >= 2
    hi, tr_read = hi[ok], tr_read[ok]
    tr_cnt = np.bincount(tr_read, minlength=R)
    tr_t, tr_q = t_of_ev[hi], q_of_ev[hi]

    # ---- decoy chains -------------------------------------------------------------------------
    n_dec = np.minimum(rng.poisson(P.decoys_per_read, R), 6)
    dec_read = np.repeat(np.arange(R), n_dec)
    D = len(dec_read)
    dec_na = 2 + rng.geometric(0.4, D)                       # anchors per decoy chain
    dec_seq = rng.choice(ref.n_seq, size=D, p=seq_len / seq_len.sum())
    dec_strand = rng.integers(0, 2, D)
    a_chain = np.repeat(np.arange(D), dec_na)
    first_a = np.cumsum(dec_na) - dec_na
    gaps_q = np.clip(np.round(rng.lognormal(np.log(P.decoy_gap_median), P.decoy_gap_sigma, len(a_chain))),
                     1, P.decoy_gap_max).astype(np.int64)
    gaps_t = np.maximum(1, np.round(gaps_q * rng.uniform(0.8, 1.3, len(a_chain)))).astype(np.int64)
    gaps_q[first_a] = 0
    gaps_t[first_a] = 0
    cq = np.cumsum(gaps_q)
    ct = np.cumsum(gaps_t)
    cq -= np.repeat(cq[first_a], dec_na)
    ct -= np.repeat(ct[first_a], dec_na)
    q0 = (rng.random(D) * np.maximum(n_ev[dec_read] - 2, 1)).astype(np.int64)
    t0 = (rng.random(D) * np.maximum(seq_len[dec_seq] - 2, 1)).astype(np.int64)
    dq = cq + np.repeat(q0, dec_na)
    dt = ct + np.repeat(t0, dec_na)
    inside = (dq < np.repeat(n_ev[dec_read], dec_na)) & (dt < np.repeat(seq_len[dec_seq], dec_na))
    keep_cnt = np.bincount(a_chain[inside], minlength=D)     # inside is a prefix of every chain (monotone)
    good = keep_cnt >= 2
    sel_a = inside & good[a_chain]
    dq, dt, a_chain = dq[sel_a], dt[sel_a], a_chain[sel_a]
    dec_ids = np.nonzero(good)[0]
    dec_cnt = keep_cnt[dec_ids]

    # ---- assemble: per read, true chain first, then its decoys ------------------------------------
    true_reads = np.nonzero(tr_cnt >= 2)[0]
    chain_read = np.concatenate([true_reads, dec_read[dec_ids]])
    chain_rank = np.concatenate([np.zeros(len(true_reads), np.int64), 1 + np.arange(len(dec_ids))])
    chain_cnt = np.concatenate([tr_cnt[true_reads], dec_cnt]).astype(np.int64)
    chain_seq = np.concatenate([seq_of_read[true_reads], dec_seq[dec_ids]])
    chain_strand = np.concatenate([strand[true_reads], dec_strand[dec_ids]])
    # anchors currently: [all true anchors grouped by read][all decoy anchors grouped by chain], forward order
    src_first = np.concatenate([np.cumsum(tr_cnt[true_reads]) - tr_cnt[true_reads],
                                len(tr_t) + np.cumsum(dec_cnt) - dec_cnt]).astype(np.int64)
    all_t = np.concatenate([tr_t, dt])
    all_q = np.concatenate([tr_q, dq])
    order = np.lexsort((chain_rank, chain_read))
    chain_read, chain_cnt, chain_seq, chain_strand, src_first = (x[order] for x in
                                                                 (chain_read, chain_cnt, chain_seq, chain_strand, src_first))
    n_chains = len(order)
    anchor_off = np.concatenate([[0], np.cumsum(chain_cnt)]).astype(np.uint64)
    # end-first inside every chain (rmap.cpp:193-196)
    within = _segment_arange(chain_cnt)
    src = np.repeat(src_first + chain_cnt - 1, chain_cnt) - within
    anchors = np.zeros(len(src), ANCHOR_DTYPE)
    anchors["target_position"] = all_t[src]
    anchors["query_position"] = all_q[src]
    chain_off = np.concatenate([[0], np.cumsum(np.bincount(chain_read, minlength=R))]).astype(np.uint64)
    ref_base = np.array([ref_offsets[(int(s), int(st))] for s, st in zip(chain_seq, chain_strand)], np.uint64) \
        if n_chains else np.zeros(0, np.uint64)
    read_base = ev_off[chain_read].astype(np.uint32)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    info = {"n_reads": R, "n_events": int(len(events)), "n_chains": int(n_chains),
            "n_true_chains": int(len(true_reads)), "n_decoy_chains": int(len(dec_ids)),
            "n_anchors": int(len(anchors)), "mappable_reads": int(mappable.sum()),
            # per read: offset of its event array in `events` (n_reads + 1 entries) and the events its last chunk added
            "ev_off": ev_off.astype(np.uint64), "events_per_chunk": int(round(P.bases_per_chunk * 1.28))}
    return cb, info


def make_rounds(cb: CandidateBatch, info: dict, n_rounds: int):
    """Chunk rounds of one batch of reads (rmap.cpp:685-693), for the cross-round cache: `cb` is the LAST
    round; round k (1-based) sees every read without its last (n_rounds - k) chunks of events, and of every chain the
    anchors that lie in the events seen so far -- chains grow at their ends from round to round, as the mapper's do
    (rmap.cpp:344-357 re-seeds a round's chaining with the previous chains' anchors).  Every chain keeps its index and at
    least its start anchor in every round, so chain c of round k continues chain c of round k - 1.  Read slots in the event
    arena are those of the last round (a read keeps its place while it grows).  Returns a list of CandidateBatch."""
    ev_off = info["ev_off"].astype(np.int64)
    n_ev = np.diff(ev_off)
    epc = int(info["events_per_chunk"])
    ao = cb.anchor_off.astype(np.int64)
    na_chain = np.diff(ao)
    chain_read = np.repeat(np.arange(cb.n_reads), np.diff(cb.chain_off.astype(np.int64)))
    anchor_chain = np.repeat(np.arange(cb.n_chains), na_chain)
    q = cb.anchors["query_position"].astype(np.int64)
    rounds = []
    for k in range(1, n_rounds + 1):
        if k == n_rounds:
            rounds.append(cb)
            break
        vis = np.maximum(n_ev - (n_rounds - k) * epc, 0)                      # events of each read seen in round k
        seen = q < vis[chain_read[anchor_chain]]
        # positions descend along a chain's list: the anchors seen are a tail of it; a chain keeps at least its last entry (= its start)
        cnt = np.maximum(np.bincount(anchor_chain[seen], minlength=cb.n_chains), np.minimum(na_chain, 1))
        new_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        first = ao[1:] - cnt                                                  # first kept entry of each chain in the full list
        idx = np.repeat(first - new_off[:-1], cnt) + np.arange(int(new_off[-1]))
        rounds.append(CandidateBatch(cb.events, cb.chain_off, new_off.astype(np.uint64), cb.anchors[idx], cb.ref_base, cb.read_base))
    return rounds


def make_seed_chunks(ref: Reference, n_reads: int, seed: int, events_per_chunk: int = 520, hit_prob: float = 0.2, false_hits: int = 25,
                     noise_sd: float = 0.25, max_chunks: int = 6, unmappable_frac: float = 0.05):
    """What detect_events + ri_sketch + ri_idx_get hand the mapper, for many reads at once (vectorised; mapper.SyntheticSeeds is the
    small per-read form the parity tests use): per read and chunk the chunk's events (z-normalised per chunk, revent.c:178-184) and
    its seed hits -- true hits on the read's real path with probability `hit_prob` per event, `false_hits` random ones per chunk.
    Returns a dict of flat arrays: `n_chunks[r]`, `qlen[r]`; chunk (r, c) has index ci = chunk_first[r] + c, events
    events[ev_off[ci] .. ev_off[ci + 1]) and hits hits[hit_off[ci] .. hit_off[ci + 1]) (mapper.HIT_DTYPE, query positions inside
    the chunk)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(x) for x in ref.forward], np.int64)
    seq = rng.choice(len(lens), size=n_reads, p=lens / lens.sum())
    strand = rng.integers(0, 2, n_reads)
    chunks_k = rng.integers(1, max_chunks + 1, n_reads)
    n_k = np.minimum(chunks_k * int(events_per_chunk / 1.28), lens[seq] - 1).astype(np.int64)
    start = (rng.random(n_reads) * (lens[seq] - n_k)).astype(np.int64)
    mappable = rng.random(n_reads) >= unmappable_frac
    read_of_k = np.repeat(np.arange(n_reads), n_k)
    kpos = np.repeat(start, n_k) + _segment_arange(n_k)
    mult = rng.choice(4, size=len(kpos), p=(0.06, 0.66, 0.22, 0.06))
    mult[np.cumsum(n_k) - n_k] = np.maximum(mult[np.cumsum(n_k) - n_k], 1)
    t_of_ev = np.repeat(kpos, mult)
    read_of_ev = np.repeat(read_of_k, mult)
    n_ev = np.bincount(read_of_ev, minlength=n_reads).astype(np.int64)
    rd_off = np.concatenate([[0], np.cumsum(n_ev)])
    q_in_read = np.arange(len(t_of_ev), dtype=np.int64) - rd_off[read_of_ev]
    clean = np.empty(len(t_of_ev), np.float32)
    for s in range(ref.n_seq):
        for st in (0, 1):
            sel = (seq[read_of_ev] == s) & (strand[read_of_ev] == st)
            arr = ref.forward[s] if st == 1 else ref.reverse[s]
            clean[sel] = arr[t_of_ev[sel]]
    vals = clean.astype(np.float64) + rng.normal(0.0, noise_sd, len(clean))
    unm = ~mappable[read_of_ev]
    vals[unm] = rng.normal(0.0, 1.0, int(unm.sum()))
    # chunks: events_per_chunk events each, the last one what is left
    n_chunks = (n_ev + events_per_chunk - 1) // events_per_chunk
    chunk_first = np.concatenate([[0], np.cumsum(n_chunks)]).astype(np.int64)
    chunk_of_ev = chunk_first[read_of_ev] + q_in_read // events_per_chunk
    n_tot = int(chunk_first[-1])
    ev_cnt = np.bincount(chunk_of_ev, minlength=n_tot).astype(np.int64)
    ev_off = np.concatenate([[0], np.cumsum(ev_cnt)]).astype(np.uint64)
    s1 = np.add.reduceat(vals, ev_off[:-1].astype(np.int64))
    s2 = np.add.reduceat(vals * vals, ev_off[:-1].astype(np.int64))
    mean = s1 / ev_cnt
    sd = np.maximum(np.sqrt(np.maximum(s2 / ev_cnt - mean * mean, 0.0)), 1e-9)
    events = ((vals - mean[chunk_of_ev]) / sd[chunk_of_ev]).astype(np.float32)
    q_in_chunk = q_in_read % events_per_chunk
    # hits: true ones (events on the path of a mappable read), then the chunk's false ones
    hit = (rng.random(len(t_of_ev)) < hit_prob) & mappable[read_of_ev]
    hi = np.nonzero(hit)[0]
    n_false = n_tot * false_hits
    f_chunk = np.repeat(np.arange(n_tot), false_hits)
    f_seq = rng.integers(0, len(lens), n_false)
    f_hits = np.zeros(n_false, [("chunk", "<i8"), ("ref_seq", "<u4"), ("strand", "<i4"), ("target_position", "<u4"), ("query_position", "<u4")])
    f_hits["chunk"], f_hits["ref_seq"], f_hits["strand"] = f_chunk, f_seq, rng.integers(0, 2, n_false)
    f_hits["target_position"] = (rng.random(n_false) * lens[f_seq]).astype(np.int64)
    f_hits["query_position"] = (rng.random(n_false) * ev_cnt[f_chunk]).astype(np.int64)
    t_hits = np.zeros(len(hi), f_hits.dtype)
    t_hits["chunk"], t_hits["ref_seq"], t_hits["strand"] = chunk_of_ev[hi], seq[read_of_ev[hi]], strand[read_of_ev[hi]]
    t_hits["target_position"], t_hits["query_position"] = t_of_ev[hi], q_in_chunk[hi]
    allh = np.concatenate([t_hits, f_hits])
    allh = allh[np.argsort(allh["chunk"], kind="stable")]
    hit_off = np.concatenate([[0], np.cumsum(np.bincount(allh["chunk"], minlength=n_tot))]).astype(np.uint64)
    hits = np.zeros(len(allh), [("ref_seq", "<u4"), ("strand", "<i4"), ("target_position", "<u4"), ("query_position", "<u4")])
    for f in hits.dtype.names:
        hits[f] = allh[f]
    return {"n_reads": n_reads, "n_chunks": n_chunks.astype(np.int64), "qlen": (n_chunks * 4000).astype(np.int64), "chunk_first": chunk_first,
            "ev_off": ev_off, "events": events, "hit_off": hit_off, "hits": hits, "mappable": mappable, "n_ev": n_ev, "seq_lens": lens}
