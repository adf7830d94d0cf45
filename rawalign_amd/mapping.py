"""Host-side consumers of the alignment scores, downstream of the DTW block -- pure host code that
mirrors the reference so that a read's PAF line can be produced from the device results:

* gen_primary_chains + comp_mapq          src/rmap.cpp:90-128, 65-88
* is_mapped_with_high_confidence          src/rmap.cpp:594-665
* the PAF fields and tag string           src/rmap.cpp:696-801, 950-965
* sequence-until (relative abundance)     src/rmap.cpp:918-944, src/sequence_until.c:4-18

Everything here runs after `rawdtw_batch_fetch`; none of it touches the device."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from ._lib import ChainOpt, ChainOut, ChainRec, SelectOpt, load_library
from .dtw import ANCHOR_DTYPE
from .align import RI_M_DTW_EVALUATE_CHAINS, RI_M_DTW_OUTPUT_CIGAR, Chain, MapOpt, dtwresult_to_string

f32 = np.float32


@dataclass
class StopOpt:
    """ri_mapopt_t fields of the stop rule and the PAF maths (src/roptions.c:9-11, 24-31)."""

    min_bestmap_ratio: float = 1.2
    min_meanmap_ratio: float = 5.0
    min_chain_anchor: int = 2
    bp_per_sec: int = 450
    sample_rate: int = 4000
    chunk_size: int = 4000
    max_num_chunk: int = 30
    min_events: int = 50      # roptions.c:23: a chunk with fewer events is not chained (rmap.cpp:569-575)

    def c_struct(self, opt: MapOpt) -> SelectOpt:
        return SelectOpt(1 if (opt.flag & RI_M_DTW_EVALUATE_CHAINS) else 0, self.min_bestmap_ratio,
                         self.min_meanmap_ratio, self.min_chain_anchor)


def _records(chains):
    arr = (ChainRec * len(chains))()
    for k, c in enumerate(chains):
        a = c.anchors
        arr[k] = ChainRec(c.chaining_score, c.alignment_score, c.reference_sequence_index,
                          getattr(c, "start_position", int(a[-1]["target_position"])),
                          getattr(c, "end_position", int(a[0]["target_position"])), len(a), c.strand, 0, k)
    return arr


def gen_primary_chains(chains, opt: MapOpt, stop: StopOpt = StopOpt()):
    """rmap.cpp:532-536: sort by (alignment_score, chaining_score, ...) descending, keep non-overlapping
    chains within a third of the previous primary's score, compute MAPQ for the best.  Returns the
    primary chains, best first, with `.mapq` set on the first."""
    if not chains:
        return []
    lib = load_library()
    rec = _records(chains)
    kept = (C.c_uint32 * len(chains))()
    so = stop.c_struct(opt)
    nk = lib.rawdtw_gen_primary_chains(rec, len(chains), C.byref(so), kept)
    out = [chains[rec[kept[k]].tag] for k in range(nk)]
    out[0].mapq = int(rec[kept[0]].mapq)
    return out


def is_mapped_with_high_confidence(primary, opt: MapOpt, stop: StopOpt = StopOpt()) -> bool:
    lib = load_library()
    if not primary:
        return False
    rec = _records(primary)
    so = stop.c_struct(opt)
    return bool(lib.rawdtw_is_mapped_with_high_confidence(rec, len(primary), C.byref(so)))


def find_outlier(x, contracted: bool = False) -> np.float32:
    """src/sequence_until.c:4-18; x[point][dim].  `contracted` selects the arithmetic of the reference's
    default FMA build (see include/rawdtw.h) instead of the source's."""
    lib = load_library()
    x = np.ascontiguousarray(x, dtype=np.float32)
    m, n = x.shape
    rows = (C.c_void_p * m)(*[x[i].ctypes.data for i in range(m)])
    fn = lib.rawdtw_find_outlier_contracted if contracted else lib.rawdtw_find_outlier
    return np.float32(fn(rows, n, m))


def default_chain_opt(e: int = 6) -> ChainOpt:
    """src/roptions.c:13-19; e = events per seed (6 for the `sensitive` preset, main.cpp:138)."""
    return ChainOpt(2000, 5000, 5000, 25, 2, 3, 10.0, e, 0)


def chain_anchors(anchors, copt: ChainOpt, max_chaining_score: float, ref_index: int, strand: int):
    """The chaining DP + traceback of gen_chains for one (sequence, strand) (rmap.cpp:430-507, 130-173).
    `anchors` (ANCHOR_DTYPE) must be sorted by (target, query) (rmap.cpp:396-401).
    Returns (list of Chain, updated running max chaining score)."""
    lib = load_library()
    a = np.ascontiguousarray(anchors, dtype=ANCHOR_DTYPE)
    n = len(a)
    cap_c = max(1, copt.num_best_chains)
    outc = (ChainOut * cap_c)()
    off = (C.c_uint64 * (cap_c + 1))()
    outa = np.zeros(max(n, 1), ANCHOR_DTYPE)
    ms = C.c_float(max_chaining_score)
    nc = lib.rawdtw_chain_anchors(C.byref(copt), a.ctypes.data_as(C.c_void_p), n, C.byref(ms), outc, off,
                                  outa.ctypes.data_as(C.c_void_p), cap_c, len(outa))
    if nc < 0:
        raise RuntimeError("chain output buffers too small")
    chains = []
    for k in range(nc):
        ch = Chain(float(outc[k].chaining_score), ref_index, strand, outa[off[k]:off[k + 1]].copy())
        ch.start_position = int(outc[k].start_position)
        ch.end_position = int(outc[k].end_position)
        chains.append(ch)
    return chains, float(ms.value)


# ------------------------------------------------------------------------------------------------
# PAF
# ------------------------------------------------------------------------------------------------
def _to_string(x) -> str:
    """std::to_string(float/double) == printf("%f")."""
    return "%f" % float(x)


@dataclass
class ReadState:
    """What map_worker_for knows about a read when it stops mapping it (rmap.cpp:667-699)."""

    read_name: str
    qlen: int                 # sig->l_sig, samples in the read
    offset: int               # reg0->offset: events consumed so far (rmap.cpp:574)
    chunks_done: int          # the loop's current_chunk when it exits (before the -- at rmap.cpp:696)
    broke_early: bool         # left the loop through the high-confidence break
    mapping_time_s: float = 0.0
    primary: list = field(default_factory=list)  # primary chains, best first (reg0->chains)


def paf_line(rs: ReadState, seq_names, seq_lens, opt: MapOpt, stop: StopOpt = StopOpt(),
             output_chains: bool = False) -> str:
    """The PAF line of one read (rmap.cpp:696-801 for the fields/tags, 956-965 for the format).
    `mt:f:` is wall-clock in the reference and therefore not comparable."""
    l_chunk, max_chunk = stop.chunk_size, stop.max_num_chunk
    current_chunk = rs.chunks_done
    chunk_start = current_chunk * l_chunk
    # rmap.cpp:696: step back one chunk when the loop ran out of signal or chunks rather than breaking
    if not rs.broke_early and current_chunk > 0 and (chunk_start >= rs.qlen or current_chunk == max_chunk):
        current_chunk -= 1
    # rmap.cpp:698, float arithmetic throughout
    scale = f32(f32(f32(current_chunk + 1) * f32(l_chunk)) / f32(rs.offset)) / f32(f32(stop.sample_rate) / f32(stop.bp_per_sec)) \
        if rs.offset else f32(np.inf)
    chains = rs.primary
    n_chains = len(chains)
    n_anchors0 = chains[0].n_anchors if n_chains else 0
    mean_chain_score = f32(0)
    for c in chains:
        mean_chain_score = f32(mean_chain_score + f32(c.chaining_score))
    mean_chain_score = f32(mean_chain_score / f32(n_chains)) if n_chains else f32(np.nan)
    mapped = is_mapped_with_high_confidence(chains, opt, stop)

    def gaps():
        at = aq = f32(0)
        a = chains[0].anchors
        for ai in range(n_anchors0 - 1):  # rmap.cpp:719-724: uint32 differences accumulated in float
            at = f32(at + f32(np.uint32(a[ai]["target_position"]) - np.uint32(a[ai + 1]["target_position"])))
            aq = f32(aq + f32(np.uint32(a[ai]["query_position"]) - np.uint32(a[ai + 1]["query_position"])))
        if n_anchors0:
            at, aq = f32(at / f32(n_anchors0)), f32(aq / f32(n_anchors0))
        return at, aq

    tags = ["mt:f:" + _to_string(rs.mapping_time_s * 1000), "ci:i:%d" % (current_chunk + 1), "sl:i:%d" % rs.qlen]
    if mapped:
        at, aq = gaps()
        c0 = chains[0]
        tags += ["cm:i:%d" % n_anchors0, "nc:i:%d" % n_chains, "s1:f:" + _to_string(f32(c0.chaining_score)),
                 "s2:f:" + _to_string(f32(chains[1].chaining_score) if n_chains > 1 else 0),
                 "sm:f:" + _to_string(mean_chain_score), "at:f:" + _to_string(at), "aq:f:" + _to_string(aq)]
        if opt.flag & RI_M_DTW_OUTPUT_CIGAR:
            tags += ["alns:f:" + _to_string(f32(c0.alignment_score)), "aln:s:" + dtwresult_to_string(c0.dtw_result)]
        if output_chains:
            tags += ["anchors:s:" + "".join("(%d,%d)" % (int(x["target_position"]), int(x["query_position"]))
                                            for x in c0.anchors)]
        a = c0.anchors
        read_end = int(np.uint32(scale * f32(np.uint32(a[0]["query_position"]))))
        read_start = int(np.uint32(scale * f32(np.uint32(a[n_anchors0 - 1]["query_position"]))))
        read_length = read_end
        start_pos = getattr(c0, "start_position", int(a[-1]["target_position"]))
        end_pos = getattr(c0, "end_position", int(a[0]["target_position"]))
        ref_len = int(seq_lens[c0.reference_sequence_index])
        frag_start = (ref_len + 1 - end_pos) & 0xFFFFFFFF if c0.strand else start_pos  # rmap.cpp:751
        frag_len = (end_pos - start_pos + 1) & 0xFFFFFFFF
        strand = "-" if c0.strand else "+"
        # rmap.cpp:961-963
        return "%s\t%u\t%u\t%u\t%s\t%s\t%u\t%u\t%u\t%u\t%u\t%d\t%s" % (
            rs.read_name, read_length, read_start, read_end, strand, seq_names[c0.reference_sequence_index], ref_len,
            frag_start, (frag_start + frag_len) & 0xFFFFFFFF, (read_end - read_start - 1) & 0xFFFFFFFF, frag_len,
            getattr(c0, "mapq", 0), "\t".join(tags))
    if n_chains >= 1:
        at, aq = gaps()
        tags += ["cm:i:%d" % n_anchors0, "nc:i:%d" % n_chains, "s1:f:" + _to_string(f32(chains[0].chaining_score)),
                 "s2:f:" + _to_string(f32(chains[1].chaining_score) if n_chains > 1 else 0),
                 "sm:f:" + _to_string(mean_chain_score), "at:f:" + _to_string(at), "aq:f:" + _to_string(aq)]
    else:
        tags += ["cm:i:0", "nc:i:0", "s1:f:0", "s2:f:0", "sm:f:0", "at:f:0", "aq:f:0"]
    read_length = int(np.uint32(scale * f32(rs.offset))) if rs.offset else 0
    return "%s\t%u\t*\t*\t*\t*\t*\t*\t*\t*\t*\t%d\t%s" % (rs.read_name, read_length, 0, "\t".join(tags))  # rmap.cpp:965


# ------------------------------------------------------------------------------------------------
# sequence-until (rmap.cpp:918-944)
# ------------------------------------------------------------------------------------------------
@dataclass
class SequenceUntil:
    n_seq: int
    t_threshold: float = 1.5   # roptions.c:43-46
    tn_samples: int = 5
    ttest_freq: int = 500
    tmin_reads: int = 500

    def __post_init__(self):
        self.c_estimations = np.zeros(self.n_seq, np.uint32)  # rmap.h:76: uint32_t (wraps like the reference's)
        self.estimations = np.zeros((self.tn_samples, self.n_seq), np.float32)
        self.ab_count = 0
        self.nreads = 0
        self.cur = 0
        self.nestimations = 0
        self.stop = 0

    def add_mapped_read(self, ref_id: int, fragment_length: int, k: int) -> bool:
        """One mapped read, in output order; k is its index in the mini-batch.  Returns True when the
        stop signal fires (p->su_stop = k+1)."""
        if self.stop:
            return True
        self.c_estimations[ref_id] = np.uint32((int(self.c_estimations[ref_id]) + int(fragment_length)) & 0xFFFFFFFF)
        self.ab_count = (self.ab_count + int(fragment_length)) & 0xFFFFFFFF  # rmap.h:74: uint32_t
        self.nreads += 1
        if self.nreads > self.tmin_reads and self.nreads % self.ttest_freq == 0:
            self.estimations[self.cur] = (self.c_estimations.astype(np.float32) / np.float32(self.ab_count))
            self.cur += 1
            if self.cur >= self.tn_samples:
                self.cur = 0
            fire = self.nestimations >= self.tn_samples
            self.nestimations += 1
            if fire and find_outlier(self.estimations) <= np.float32(self.t_threshold):
                self.stop = k + 1
                return True
        return False
