"""The library's chunk-round mapper (rawdtw_mapper_*, rawalign_amd/csrc/rawdtw_mapper.cpp) WITHOUT a device: its host logic
-- events, --min-events, re-seeding, sort, chaining DP, evaluation order, primary chains / MAPQ / stop rule, the PAF line, the
thread pool, the checks in front of a round -- driven through the harness hook rawdtw_mapper_set_scorer with the oracle as the
scorer, against the Python mirror (rawalign_amd.mapper.map_reads) scored by the same oracle.  Line for line."""
import numpy as np
import pytest

import rawalign_amd as ra
from rawalign_amd import mapper, synth
from rawalign_amd.mapping import StopOpt
from tests.util import OracleScorer


def _oracle_scorer(oracle, ref, opt):
    from oracle.loader import OrcOpt

    oopt = OrcOpt(opt.dtw_border_constraint, opt.dtw_fill_method, opt.dtw_band_radius_frac, opt.dtw_match_bonus, opt.dtw_min_score, int(opt.fused_score))

    def score(chain_off, anchor_off, anchors, seq, strand, evs):   # the DTW block of gen_chains, rmap.cpp:515-524
        nc = int(chain_off[-1])
        sc, kp = np.zeros(nc, np.float32), np.zeros(nc, np.uint8)
        for r in range(len(chain_off) - 1):
            best = np.float32(0.0)
            for c in range(int(chain_off[r]), int(chain_off[r + 1])):
                arr = ref.forward[int(seq[c])] if strand[c] == 1 else ref.reverse[int(seq[c])]
                s = oracle.align_chain(anchors[int(anchor_off[c]):int(anchor_off[c + 1])], arr, evs[r], oopt, float(best))
                sc[c] = s
                if s >= np.float32(opt.dtw_min_score):
                    kp[c] = 1
                    if s > best:
                        best = s
        return sc, kp
    return score


@pytest.mark.parametrize("flag,never,threads,min_events", [(0x2, 0, 1, 50), (0x2, 1, 4, 50), (0x8, 0, 3, 50), (0x2, 1, 2, 300), (0x2 | 0x8, 0, 4, 300), (0x0, 0, 2, 50)])
def test_cpp_mapper_host_logic_equals_python_mirror(oracle, flag, never, threads, min_events):
    ref = synth.make_reference([29903, 12000], seed=20231005 + 1)
    n = 30
    seeds = mapper.SyntheticSeeds(ref, n, seed=11, max_chunks=4)
    opt = ra.MapOpt(flag=flag)
    stop = StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6, min_events=min_events) if never else StopOpt(min_events=min_events)
    lo = []
    want, rounds = mapper.map_reads(seeds, list(range(n)), OracleScorer(oracle, ref), opt, stop, log=lo)
    cm = mapper.CMapper(None, opt, stop, [f"seq{s}" for s in range(ref.n_seq)], [len(x) for x in ref.forward],
                        slot_events=max(rd["n_ev"] for rd in seeds.reads) + 8, max_reads=n, carry=True, threads=threads)
    cm.set_scorer(_oracle_scorer(oracle, ref, opt))
    got, rounds_c = mapper.map_reads_c(seeds, list(range(n)), cm)
    assert got == want and rounds_c == rounds
    if flag & 0x8:
        assert cm.log() == "".join(lo) and lo
    if min_events > 50:
        assert sum(1 for r in range(n) for c in range(seeds.read_job(r).n_chunks_available) if len(seeds.chunk(r, c)[0]) < min_events) > 0
    t = cm.timing()
    assert t["host_phase_ms"] > 0 and t["anchor_bytes"] == 0   # (nothing went to a device)
    cm.close()


def test_cpp_mapper_round_is_checked_before_anything_changes(oracle):
    """unknown id, a read twice, a finished read, a hit on an unknown sequence, a read over its slot: the round is refused and
    the reads stay as they were (the same round can be made again); released reads give their slot back."""
    ref = synth.make_reference([20000], seed=5)
    seeds = mapper.SyntheticSeeds(ref, 4, seed=2, max_chunks=3)
    opt, stop = ra.MapOpt(), StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6)
    slot = max(rd["n_ev"] for rd in seeds.reads) + 8
    cm = mapper.CMapper(None, opt, stop, ["seq0"], [len(ref.forward[0])], slot_events=slot, max_reads=2, threads=2)
    with pytest.raises(RuntimeError):   # no device and no scorer
        cm.round([cm.add_read("x", 4000, 1)], [seeds.chunk(0, 0)])
    cm.close()
    cm = mapper.CMapper(None, opt, stop, ["seq0"], [len(ref.forward[0])], slot_events=slot, max_reads=2, threads=2)
    cm.set_scorer(_oracle_scorer(oracle, ref, opt))
    a = cm.add_read("read_0", seeds.read_job(0).qlen, seeds.read_job(0).n_chunks_available)
    b = cm.add_read("read_1", seeds.read_job(1).qlen, seeds.read_job(1).n_chunks_available)
    with pytest.raises(RuntimeError):
        cm.add_read("read_2", 4000, 1)                      # no slot left
    c0, c1 = seeds.chunk(0, 0), seeds.chunk(1, 0)
    for ids, chunks in (([a, 7], [c0, c1]), ([a, a], [c0, c0]), ([a, b], [c0, (c1[0], [(3, 0, 5, 5)])]),
                        ([a, b], [c0, (np.zeros(slot + 1, np.float32), [])])):
        with pytest.raises(RuntimeError):
            cm.round(ids, chunks)
        assert cm.state(a) == (False, 0) and cm.state(b) == (False, 0) and cm.stats()[0] == 0
    cm.round([a, b], [c0, c1])                               # ... and the round still goes through afterwards
    assert cm.state(a)[1] == 1 and cm.stats()[0] == 1
    # run read a to its end, release it: its slot goes to the next read
    while not cm.state(a)[0]:
        cm.round([a], [seeds.chunk(0, cm.state(a)[1])])
    with pytest.raises(RuntimeError):
        cm.round([a], [c0])                                  # a finished read in a round
    line = cm.paf(a)
    cm.release_read(a)
    with pytest.raises(RuntimeError):
        cm.paf(a)
    c = cm.add_read("read_2", seeds.read_job(2).qlen, seeds.read_job(2).n_chunks_available)
    cm.round([c], [seeds.chunk(2, 0)])
    assert cm.state(c)[1] == 1 and line.startswith("read_0\t")
    cm.close()
