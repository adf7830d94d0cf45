"""End-to-end control flow (chunk rounds -> chaining -> DTW batch -> primary chains / MAPQ / stop rule ->
PAF): the device path must produce exactly the PAF lines of the same flow scored by the CPU oracle.
d1-like scale (configs[0]: SARS-CoV-2, 29,903 bp) on the CPU leg."""
import numpy as np
import pytest

import rawalign_amd as ra
from rawalign_amd import mapper, synth
from tests.util import OracleScorer


def setup(n_reads=24):
    ref = synth.make_reference([29903], seed=20231005 + 1)
    seeds = mapper.SyntheticSeeds(ref, n_reads, seed=3, max_chunks=4)
    return ref, seeds


def test_oracle_path_produces_paf(oracle):
    ref, seeds = setup(12)
    opt = ra.MapOpt()
    lines, rounds = mapper.map_reads(seeds, list(range(12)), OracleScorer(oracle, ref), opt)
    assert len(lines) == 12 and rounds >= 1
    mapped = [l for l in lines if l.split("\t")[4] in "+-"]
    assert len(mapped) >= 6  # most mappable reads map
    for l in mapped:
        f = l.split("\t")
        r = int(f[0].split("_")[1])
        rd = seeds.reads[r]
        assert f[4] == ("-" if rd["strand"] else "+")
        assert int(f[6]) == 29898  # signal length of the only sequence (29903 - k + 1)


@pytest.mark.gpu
def test_device_path_paf_identical_to_oracle_path(oracle):
    ref, seeds = setup(40)
    eng = ra.Engine(0)
    eng.upload_reference(ref.forward, ref.reverse)
    for opt in (ra.MapOpt(), ra.MapOpt(dtw_border_constraint=0, dtw_fill_method=1),
                ra.MapOpt(dtw_border_constraint=1, dtw_fill_method=0)):
        a, ra_ = mapper.map_reads(seeds, list(range(40)), OracleScorer(oracle, ref), opt)
        b, rb_ = mapper.map_reads(seeds, list(range(40)), mapper.DeviceScorer(eng), opt)
        assert ra_ == rb_ and a == b


def _abundance_stop(lines, n_seq):
    """configs[4]-style consumer: feed the mapped reads, in output order, to sequence-until."""
    from rawalign_amd.mapping import SequenceUntil

    su = SequenceUntil(n_seq=n_seq, tmin_reads=8, ttest_freq=4, tn_samples=3, t_threshold=1.5)
    for k, l in enumerate(lines):
        f = l.split("\t")
        if f[4] in "+-":
            ref_id = int(f[5][3:])              # "seq<N>"
            if su.add_mapped_read(ref_id, int(f[10]), k):
                break
    return su.stop, su.c_estimations.copy()


def test_multi_genome_sequence_until_oracle_path(oracle):
    """Metagenomic multi-genome index + real-time sequence-until on the CPU leg (plumbing of configs[4])."""
    ref = synth.make_reference([20000, 35000, 12000], seed=20231005 + 5)
    seeds = mapper.SyntheticSeeds(ref, 40, seed=11, max_chunks=3)
    lines, _ = mapper.map_reads(seeds, list(range(40)), OracleScorer(oracle, ref), ra.MapOpt())
    stop, est = _abundance_stop(lines, 3)
    assert stop > 0 and est.sum() > 0
    # every mapped read landed on the genome and strand it was drawn from
    for l in lines:
        f = l.split("\t")
        if f[4] in "+-":
            rd = seeds.reads[int(f[0].split("_")[1])]
            assert int(f[5][3:]) == rd["seq"] and f[4] == ("-" if rd["strand"] else "+")


@pytest.mark.gpu
def test_multi_genome_sequence_until_device_equals_oracle(oracle):
    ref = synth.make_reference([20000, 35000, 12000], seed=20231005 + 5)
    seeds = mapper.SyntheticSeeds(ref, 60, seed=12, max_chunks=3)
    eng = ra.Engine(0)
    eng.upload_reference(ref.forward, ref.reverse)
    a, _ = mapper.map_reads(seeds, list(range(60)), OracleScorer(oracle, ref), ra.MapOpt())
    b, _ = mapper.map_reads(seeds, list(range(60)), mapper.DeviceScorer(eng), ra.MapOpt())
    assert a == b
    sa, ea = _abundance_stop(a, 3)
    sb, eb = _abundance_stop(b, 3)
    assert sa == sb and np.array_equal(ea, eb)


@pytest.mark.gpu
def test_cross_chunk_memoisation_is_exact_and_saves_work(oracle):
    """parts already scored in an earlier chunk round are reused; PAF lines unchanged."""
    ref, seeds = setup(40)
    eng = ra.Engine(0)
    eng.upload_reference(ref.forward, ref.reverse)
    for opt in (ra.MapOpt(), ra.MapOpt(dtw_fill_method=0)):
        plain = mapper.DeviceScorer(eng)
        memo = mapper.DeviceScorer(eng, memoise=True)
        memo.ref_host = ref
        # a stop rule that never fires: every read goes through all of its chunks, chains grow round by round
        from rawalign_amd.mapping import StopOpt

        never = StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6)
        a, ra_ = mapper.map_reads(seeds, list(range(40)), plain, opt, never)
        b, rb_ = mapper.map_reads(seeds, list(range(40)), memo, opt, never)
        assert a == b and ra_ == rb_ and ra_ > 1
        assert memo.jobs_reused > 0 and memo.jobs_scored < plain.jobs_scored
        # ... and against the oracle's sequential loop under the same stop rule (not only against the plain device path)
        c, rc_ = mapper.map_reads(seeds, list(range(40)), OracleScorer(oracle, ref), opt, never)
        assert c == b and rc_ == rb_


def test_cigar_and_log_scores_flags_oracle_path(oracle):
    """a-6 on the CPU leg: with RI_M_DTW_OUTPUT_CIGAR a mapped read's line carries alns:f: and aln:s: (rmap.cpp:741-744);
    with RI_M_DTW_LOG_SCORES alone DTW runs but the chain list is not replaced (rmap.cpp:509, 525)."""
    from rawalign_amd.align import RI_M_DTW_EVALUATE_CHAINS, RI_M_DTW_LOG_SCORES, RI_M_DTW_OUTPUT_CIGAR

    ref, seeds = setup(10)
    opt = ra.MapOpt(flag=RI_M_DTW_EVALUATE_CHAINS | RI_M_DTW_OUTPUT_CIGAR)
    lines, _ = mapper.map_reads(seeds, list(range(10)), OracleScorer(oracle, ref), opt)
    mapped = [l for l in lines if l.split("\t")[4] in "+-"]
    assert mapped and all("\talns:f:" in l and "\taln:s:(" in l for l in mapped)
    assert all("aln:s:" not in l for l in lines if l.split("\t")[4] == "*")
    # log-scores only: every chain that was not cut is logged, and chains that fail dtw_min_score stay in the list
    log = []
    ev_lines, _ = mapper.map_reads(seeds, list(range(10)), OracleScorer(oracle, ref), ra.MapOpt())
    lg_lines, _ = mapper.map_reads(seeds, list(range(10)), OracleScorer(oracle, ref), ra.MapOpt(flag=RI_M_DTW_LOG_SCORES), log=log)
    assert log and all(l.startswith("chaining_score=") and " alignment_score=" in l and l.endswith("\n") for l in log)
    nc = lambda ls: sum(int(l.split("nc:i:")[1].split("\t")[0]) for l in ls)  # noqa: E731
    assert nc(lg_lines) >= nc(ev_lines)


@pytest.mark.gpu
def test_cigar_and_log_scores_flags_device_equals_oracle(oracle):
    """a-6 through the device: PAF lines with alns:f: / aln:s: and the --dtw-log-scores lines are character-identical to
    the same control flow on the checker -- sparse (banded and full) and global+full, plus log-scores alone."""
    from rawalign_amd.align import RI_M_DTW_EVALUATE_CHAINS, RI_M_DTW_LOG_SCORES, RI_M_DTW_OUTPUT_CIGAR

    ref, seeds = setup(30)
    eng = ra.Engine(0)
    eng.upload_reference(ref.forward, ref.reverse)
    ec = RI_M_DTW_EVALUATE_CHAINS | RI_M_DTW_OUTPUT_CIGAR
    for opt in (ra.MapOpt(flag=ec), ra.MapOpt(dtw_fill_method=0, flag=ec),
                ra.MapOpt(dtw_border_constraint=0, dtw_fill_method=0, flag=ec),
                ra.MapOpt(flag=RI_M_DTW_LOG_SCORES), ra.MapOpt(flag=ec | RI_M_DTW_LOG_SCORES)):
        la, lb = [], []
        a, _ = mapper.map_reads(seeds, list(range(30)), OracleScorer(oracle, ref), opt, log=la)
        b, _ = mapper.map_reads(seeds, list(range(30)), mapper.DeviceScorer(eng), opt, log=lb)
        assert a == b and la == lb
        if opt.flag & RI_M_DTW_OUTPUT_CIGAR:
            assert any("\taln:s:(" in l for l in b)
        if opt.flag & RI_M_DTW_LOG_SCORES:
            assert lb


@pytest.mark.gpu
def test_device_side_carry_over_three_rounds_and_more(oracle):
    """SURVEY.md 8(f-4) on the device (rawdtw_batch_submit_round): the part costs of a round stay in HBM and the next
    round's batch takes over every part whose anchors did not change -- the host only says which chain continues which.
    Every read runs through all of its chunks (>= 3 rounds); the PAF lines must equal the plain device path's and the
    oracle's from-scratch loop (rmap.cpp:516-517), with fewer parts scored."""
    from rawalign_amd.mapping import StopOpt

    ref = synth.make_reference([29903], seed=20231005 + 1)
    seeds = mapper.SyntheticSeeds(ref, 40, seed=3, max_chunks=5)
    eng = ra.Engine(0)
    eng.upload_reference(ref.forward, ref.reverse)
    never = StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6)
    for opt in (ra.MapOpt(), ra.MapOpt(dtw_fill_method=0)):
        plain = mapper.DeviceScorer(eng)
        a, ra_ = mapper.map_reads(seeds, list(range(40)), plain, opt, never)
        eng2 = ra.Engine(0)
        eng2.upload_reference(ref.forward, ref.reverse)
        carry = mapper.RoundScorer(eng2, slot_events=max(rd["n_ev"] for rd in seeds.reads) + 8, n_slots=40)
        b, rb_ = mapper.map_reads(seeds, list(range(40)), carry, opt, never)
        carry.close()
        assert ra_ == rb_ and ra_ >= 3
        assert a == b
        c, rc_ = mapper.map_reads(seeds, list(range(40)), OracleScorer(oracle, ref), opt, never)
        assert c == b and rc_ == rb_
        if opt.dtw_fill_method == 1:   # (the sync-free path: sparse + banded; other modes go through the job list and score everything)
            assert carry.jobs_reused > 0 and carry.jobs_scored < plain.jobs_scored
        assert carry.jobs_scored + carry.jobs_reused == plain.jobs_scored
    # the reference's own stop rule as well (reads drop out of the rounds at different times)
    a, _ = mapper.map_reads(seeds, list(range(40)), mapper.DeviceScorer(eng), ra.MapOpt())
    eng3 = ra.Engine(0)
    eng3.upload_reference(ref.forward, ref.reverse)
    carry = mapper.RoundScorer(eng3, slot_events=max(rd["n_ev"] for rd in seeds.reads) + 8, n_slots=40)
    b, _ = mapper.map_reads(seeds, list(range(40)), carry, ra.MapOpt())
    carry.close()
    assert a == b


@pytest.mark.gpu
def test_cpp_mapper_lines_do_not_depend_on_threads_groups_or_carry(oracle):
    """The library's mapper (rawdtw_mapper_*) over 600 reads of a 200 kb reference: the PAF lines' hash is the same on one host
    thread and eight, with one read group and two (two contexts), with costs carried from round to round and without, under
    the reference's stop rule and with every read through all of its chunks -- and equals the Python mirror's on the plain
    device path (the flow the rounds above pin to the oracle)."""
    import hashlib

    from rawalign_amd.mapping import StopOpt

    ref = synth.make_reference([200_000], seed=77)
    n = 600
    seeds = mapper.SyntheticSeeds(ref, n, seed=5, max_chunks=5)
    names, lens = ["seq0"], [len(ref.forward[0])]
    slot = max(rd["n_ev"] for rd in seeds.reads) + 8
    for stop in (StopOpt(), StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6)):
        hashes, reused = {}, {}
        for threads, groups, carry in ((1, 1, 1), (8, 2, 1), (8, 1, 0), (3, 2, 0), (8, 2, 1)):
            eng = ra.Engine(0)
            eng.upload_reference(ref.forward, ref.reverse)
            cm = mapper.CMapper(eng, ra.MapOpt(), stop, names, lens, slot_events=slot, max_reads=n, carry=carry, threads=threads, groups=groups)
            lines, rounds = mapper.map_reads_c(seeds, list(range(n)), cm)
            hashes[(threads, groups, carry)] = hashlib.sha1("\n".join(lines).encode()).hexdigest()
            reused[(threads, groups, carry)] = cm.stats()[2]
            tm = cm.timing()
            assert tm["anchor_bytes"] > 0 and tm["event_bytes"] > 0
            cm.close()
            eng.close()
        assert len(set(hashes.values())) == 1, hashes
        eng = ra.Engine(0)
        eng.upload_reference(ref.forward, ref.reverse)
        want, _ = mapper.map_reads(seeds, list(range(n)), mapper.DeviceScorer(eng), ra.MapOpt(), stop)
        eng.close()
        assert hashlib.sha1("\n".join(want).encode()).hexdigest() == hashes[(1, 1, 1)]
        if stop.min_chain_anchor > 2:   # every read through all of its chunks: the rounds in which costs are taken over
            assert reused[(8, 2, 1)] > 0 and reused[(1, 1, 1)] == reused[(8, 2, 1)] and reused[(8, 1, 0)] == 0


@pytest.mark.gpu
def test_cpp_mapper_with_the_chaining_on_the_device_writes_the_same_lines(oracle):
    """opt.device_chain: the anchor sort and the chaining DP of every round on the device (rawdtw_chain_round), the chains handed to the DTW in
    device memory -- the PAF lines' hash equals the host-chained mapper's (which the test above ties to the Python mirror and the rounds further
    up to the oracle), under the stop rule and with every read through all of its chunks, with one read group and two, flags for the CIGAR and the
    score log included; no anchor list crosses PCIe on the way in."""
    import hashlib

    from rawalign_amd.mapping import StopOpt

    ref = synth.make_reference([200_000], seed=77)
    n = 600
    seeds = mapper.SyntheticSeeds(ref, n, seed=5, max_chunks=5)
    names, lens = ["seq0"], [len(ref.forward[0])]
    slot = max(rd["n_ev"] for rd in seeds.reads) + 8
    for stop, opt in ((StopOpt(), ra.MapOpt()), (StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6), ra.MapOpt()),
                      (StopOpt(), ra.MapOpt(dtw_border_constraint=0, dtw_fill_method=0, flag=0x2 | 0x4 | 0x8))):
        hashes, logs = {}, {}
        for threads, groups, dev in ((8, 1, 0), (8, 1, 1), (3, 2, 1), (1, 2, 1)):
            eng = ra.Engine(0)
            eng.upload_reference(ref.forward, ref.reverse)
            cm = mapper.CMapper(eng, opt, stop, names, lens, slot_events=slot, max_reads=n, carry=True, threads=threads, groups=groups, device_chain=bool(dev))
            lines, rounds = mapper.map_reads_c(seeds, list(range(n)), cm)
            hashes[(threads, groups, dev)] = hashlib.sha1("\n".join(lines).encode()).hexdigest()
            logs[(threads, groups, dev)] = hashlib.sha1(cm.log().encode()).hexdigest()
            tm = cm.timing()
            assert (tm["anchor_bytes"] == 0) == bool(dev) and tm["event_bytes"] > 0
            assert cm.stats()[1] > 0
            cm.close()
            eng.close()
        assert len(set(hashes.values())) == 1, hashes
        assert len(set(logs.values())) == 1, logs


@pytest.mark.gpu
def test_a_round_the_device_declines_to_chain_is_chained_on_the_host(oracle, monkeypatch):
    """RAWDTW_CHAIN_MAX_SEEDS=60 makes rawdtw_chain_round decline every round in which some read has more than 60 seeds (a chunk's hits alone are
    often that many; the previous chains' anchors come on top): those rounds are chained on the host -- their anchor lists go up from there --, any
    other on the device, and the lines are the host-chained mapper's."""
    import hashlib

    from rawalign_amd.mapping import StopOpt

    monkeypatch.setenv("RAWDTW_CHAIN_MAX_SEEDS", "60")
    ref = synth.make_reference([120_000], seed=78)
    n = 200
    seeds = mapper.SyntheticSeeds(ref, n, seed=6, max_chunks=5)
    names, lens = ["seq0"], [len(ref.forward[0])]
    slot = max(rd["n_ev"] for rd in seeds.reads) + 8
    stop = StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6)
    out = {}
    for dev in (0, 1):
        eng = ra.Engine(0)
        eng.upload_reference(ref.forward, ref.reverse)
        cm = mapper.CMapper(eng, ra.MapOpt(), stop, names, lens, slot_events=slot, max_reads=n, carry=False, threads=4, groups=2, device_chain=bool(dev))
        lines, rounds = mapper.map_reads_c(seeds, list(range(n)), cm)
        out[dev] = (hashlib.sha1("\n".join(lines).encode()).hexdigest(), cm.timing()["anchor_bytes"], rounds)
        cm.close()
        eng.close()
    assert out[0][0] == out[1][0]
    assert 0 < out[1][1] <= out[0][1]  # the declined rounds' anchors went up from the host


@pytest.mark.gpu
def test_multi_genome_index_with_the_chaining_on_the_device(oracle):
    """configs[4]'s shape -- a multi-genome index: seven sequences, fourteen (sequence, strand) lists a read can have seeds on -- through the library's
    mapper with the chaining on the host and on the device: the same lines, equal to the oracle-scored Python flow's, and the same abundance stop."""
    from rawalign_amd.mapping import StopOpt

    ref = synth.make_reference([20000, 35000, 12000, 8000, 26000, 15000, 30000], seed=20231005 + 9)
    n = 150
    seeds = mapper.SyntheticSeeds(ref, n, seed=13, max_chunks=4)
    names, lens = [f"seq{s}" for s in range(ref.n_seq)], [len(x) for x in ref.forward]
    slot = max(rd["n_ev"] for rd in seeds.reads) + 8
    want, _ = mapper.map_reads(seeds, list(range(n)), OracleScorer(oracle, ref), ra.MapOpt(), StopOpt())
    out = {}
    for dev, groups in ((0, 1), (1, 1), (1, 2)):
        eng = ra.Engine(0)
        eng.upload_reference(ref.forward, ref.reverse)
        cm = mapper.CMapper(eng, ra.MapOpt(), StopOpt(), names, lens, slot_events=slot, max_reads=n, carry=False, threads=4, groups=groups, device_chain=bool(dev))
        out[(dev, groups)], _ = mapper.map_reads_c(seeds, list(range(n)), cm)
        cm.close()
        eng.close()
    assert out[(0, 1)] == want and out[(1, 1)] == want and out[(1, 2)] == want
    stop, est = _abundance_stop(want, ref.n_seq)
    assert stop > 0 and est.sum() > 0


@pytest.mark.gpu
def test_a_round_declined_after_the_device_chained_it_is_chained_on_the_host(oracle):
    """A read whose seeds make twenty equal chains on twenty (sequence, strand) lists: more than sixteen chains with equal scores is an order only
    std::sort knows (rawdtw_chain_round reports it when the round is ENDED, after the device has chained it) -- the mapper chains that round on
    the host and writes the lines of the host-chained mapper; the other reads of the round and the rounds behind it are not disturbed."""
    from rawalign_amd.mapping import StopOpt

    ref = synth.make_reference([9000] * 10, seed=99)
    rng = np.random.default_rng(5)
    names, lens = [f"seq{s}" for s in range(ref.n_seq)], [len(x) for x in ref.forward]
    n, n_chunks = 12, 3
    stop = StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6)
    chunks = {}
    for r in range(n):
        for c in range(n_chunks):
            ev = rng.normal(size=300).astype(np.float32)
            hits = []
            if r == 3 and c == 1:   # the same little diagonal on every list
                for s in range(10):
                    for st in (0, 1):
                        hits += [(s, st, 1000 + 11 * k, 20 + 10 * k) for k in range(6)]
            else:
                s, st, t0 = int(rng.integers(0, 10)), int(rng.integers(0, 2)), int(rng.integers(500, 7000))
                hits = [(s, st, t0 + 12 * k + int(rng.integers(0, 3)), 10 + 11 * k) for k in range(20)]
                hits += [(int(rng.integers(0, 10)), int(rng.integers(0, 2)), int(rng.integers(0, 8000)), int(rng.integers(0, 300))) for _ in range(10)]
            chunks[(r, c)] = (ev, hits)
    out = {}
    for dev in (0, 1):
        eng = ra.Engine(0)
        eng.upload_reference(ref.forward, ref.reverse)
        cm = mapper.CMapper(eng, ra.MapOpt(flag=0x2 | 0x8), stop, names, lens, slot_events=1000, max_reads=n, carry=False, threads=3, groups=2, device_chain=bool(dev))
        ids = [cm.add_read("read_%d" % r, 4000 * n_chunks, n_chunks) for r in range(n)]
        for c in range(n_chunks):
            cm.round(ids, [chunks[(r, c)] for r in range(n)])
        assert cm.finish() == 0
        out[dev] = ([cm.paf(i) for i in ids], cm.log(), cm.timing()["anchor_bytes"])
        cm.close()
        eng.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert 0 < out[1][2] < out[0][2]  # one round (of one group) went up as anchor lists from the host, the others did not
