"""Parity of the HIP path (through the C ABI) with the oracle and the reference's golden vectors.
Bit-exact: the DP is add/sub/abs/min in fp32 with no reassociation."""
import numpy as np
import pytest

try:  # PyTorch bundles its own HIP runtime: when both live in one process, torch has to come up first
    import torch

    torch.cuda.is_available()
except Exception:  # pragma: no cover - torch is optional for these tests
    torch = None

import rawalign_amd as ra
from rawalign_amd.dtw import JOB_DTYPE
from tests.golden_util import bits
from tests.util import assert_bits_equal, default_radius, make_arena_jobs, oracle_costs, planner_options

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    return ra.Engine(0)


def run(engine, jobs, events, ref):
    pad = np.zeros(8, np.float32)
    engine.upload_reference([np.concatenate([ref, pad])], [np.concatenate([ref, pad])])
    jobs = jobs.copy()
    jobs["ref_off"] += engine.reference_offset(0, 1)
    return engine.score_batch(jobs, events)


def test_golden_banded_and_full(engine, golden):
    cases = [(c.a, c.b, c.R0, c.exclude_last) for c in golden] + [(c.a, c.b, -1, c.exclude_last) for c in golden]
    jobs, ev, rf = make_arena_jobs(cases)
    got = run(engine, jobs, ev, rf)
    want = np.array([c.banded_bits for c in golden] + [c.global_bits for c in golden], np.uint32).view(np.float32)
    assert_bits_equal(got, want, "golden")


def test_golden_traceback(engine, golden):
    cs = [c for c in golden if c.tb is not None]
    jobs, ev, rf = make_arena_jobs([(c.a, c.b, -1, c.exclude_last) for c in cs])
    engine.upload_reference([rf], [rf])
    jobs["ref_off"] += engine.reference_offset(0, 1)
    res = engine.traceback_batch(jobs, ev)
    for c, r in zip(cs, res):
        assert bits(r.cost) == c.tb[0]
        assert np.array_equal(r.i, c.tb[1]) and np.array_equal(r.j, c.tb[2])
        assert np.array_equal(r.difference.view(np.uint32), c.tb[3])


def test_reference_function_names(engine, oracle):
    """Reads like src/check_dtw.cpp:138-181: same inputs through every DTW variant."""
    rng = np.random.default_rng(11)
    for n, m in [(4, 4), (10, 10), (20, 10), (25, 10), (100, 100), (200, 50), (200, 30), (1, 1), (1, 9), (9, 1)]:
        a = rng.uniform(-2.5, 2.5, n).astype(np.float32)
        b = rng.uniform(-2.5, 2.5, m).astype(np.float32)
        for ex in (False, True):
            assert bits(engine.DTW_global(a, b, ex)) == bits(oracle.dtw_global(a, b, ex))
            for R in (0, 1, 2, 5, 13, 40):
                assert bits(engine.DTW_global_slantedbanded_antidiagonalwise(a, b, R, ex)) == bits(
                    oracle.dtw_banded(a, b, R, ex))
            tb = engine.DTW_global_tb(a, b, ex)
            c, pi, pj, pd = oracle.dtw_global_tb(a, b, ex)
            assert bits(tb.cost) == bits(c)
            assert np.array_equal(tb.i, pi) and np.array_equal(tb.j, pj)
            assert np.array_equal(tb.difference.view(np.uint32), pd.view(np.uint32))


def test_random_sparse_like_batch(engine, oracle):
    """Thousands of tiny segments, default radius, every radius class of the lane kernel."""
    rng = np.random.default_rng(3)
    cases = []
    for t in range(6000):
        n = int(rng.integers(1, 130))
        m = max(1, int(round(n * rng.uniform(0.3, 1.6))))
        R0 = default_radius(n) if t % 3 else int(rng.integers(0, 9))
        cases.append((rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32), R0, t & 1))
    jobs, ev, rf = make_arena_jobs(cases)
    assert_bits_equal(run(engine, jobs, ev, rf), oracle_costs(oracle, jobs, ev, rf), "sparse-like")


def test_random_wave_band(engine, oracle):
    """Radii beyond the lane kernel (wave-per-job kernel), including bands that clip the optimum."""
    rng = np.random.default_rng(4)
    cases = []
    for t in range(160):
        n = int(rng.integers(100, 1500))
        m = max(1, int(round(n * rng.uniform(0.5, 1.4))))
        R0 = default_radius(n) if t % 2 else int(rng.integers(13, 90))
        cases.append((rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32), R0, t & 1))
    # radii on both sides of every registers-per-lane class boundary (K = R+1: one register a lane up to 64 slots, two up to 128, then the odd counts --
    # three up to 192, five up to 320, seven up to 448, nine up to 512 -- and the register-only body beyond), square and slanted; long enough
    # for several LDS segments
    for K in (64, 65, 128, 129, 192, 193, 256, 257, 320, 321, 448, 449, 512, 513, 1024, 1025, 2048, 2049):
        n = int(K * 1.6) + 7
        cases.append((rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32), K - 1, K & 1))
    for K in (150, 192, 193, 300, 320, 321, 400, 448, 449, 512):
        n = 2300 + K
        m = n - n // 10
        r0 = max(r for r in range(1, K) if r + ((n - m) * r + n - 1) // n + 1 <= K)  # (the radius that the slant of m = 0.9 n widens to K slots: dtw.cpp:298-300)
        cases.append((rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32), r0, K & 1))
        cases.append((rng.normal(size=m).astype(np.float32), rng.normal(size=n).astype(np.float32), r0, 1 - (K & 1)))
    for R0 in range(5, 16):  # around the lane-kernel / wave-kernel hand-over
        cases.append((rng.normal(size=150).astype(np.float32), rng.normal(size=140).astype(np.float32), R0, R0 & 1))
    cases.append((rng.normal(size=6000).astype(np.float32), rng.normal(size=5200).astype(np.float32), 600, 0))
    cases.append((rng.normal(size=3000).astype(np.float32), rng.normal(size=9000).astype(np.float32), 2100, 1))
    jobs, ev, rf = make_arena_jobs(cases)
    assert_bits_equal(run(engine, jobs, ev, rf), oracle_costs(oracle, jobs, ev, rf), "wave band")


def test_random_full(engine, oracle):
    """DTW_global through every rows-per-lane class and across strip boundaries."""
    rng = np.random.default_rng(5)
    shapes = [(1, 1), (2, 2), (1, 50), (50, 1), (63, 64), (64, 65), (65, 64), (128, 200), (129, 300), (256, 256),
              (257, 256), (511, 700), (512, 512), (513, 514), (700, 520), (1025, 1100), (1600, 1030), (40, 3000),
              (2100, 2300), (1537, 1536), (3000, 2049), (1025, 5000)]  # the last four: >= 3 strips, four waves per job
    cases = []
    for n, m in shapes:
        for ex in (0, 1):
            cases.append((rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32), -1, ex))
    for _ in range(300):
        n, m = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        cases.append((rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32), -1, 0))
    jobs, ev, rf = make_arena_jobs(cases)
    assert_bits_equal(run(engine, jobs, ev, rf), oracle_costs(oracle, jobs, ev, rf), "full")


def test_random_traceback(engine, oracle):
    rng = np.random.default_rng(6)
    shapes = [(1, 1), (1, 7), (7, 1), (2, 2), (30, 20), (64, 64), (65, 100), (130, 129), (260, 300), (513, 600),
              (600, 513), (900, 1100), (1500, 1300), (1100, 2600)]
    cases = [(rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32), -1, k & 1)
             for k, (n, m) in enumerate(shapes)]
    jobs, ev, rf = make_arena_jobs(cases)
    engine.upload_reference([rf], [rf])
    jobs["ref_off"] += engine.reference_offset(0, 1)
    res = engine.traceback_batch(jobs, ev)
    for (a, b, _, ex), r in zip(cases, res):
        c, pi, pj, pd = oracle.dtw_global_tb(a, b, ex)
        assert bits(r.cost) == bits(c), (len(a), len(b))
        assert np.array_equal(r.i, pi) and np.array_equal(r.j, pj), (len(a), len(b))
        assert np.array_equal(r.difference.view(np.uint32), pd.view(np.uint32))


def test_quantised_inputs_force_ties(engine, oracle):
    """Equal neighbours everywhere: the diagonal-wins tie rule (dtw.cpp:633-646) decides the path."""
    rng = np.random.default_rng(8)
    for n, m in [(40, 40), (70, 55), (300, 280)]:
        a = rng.integers(-2, 3, n).astype(np.float32)
        b = rng.integers(-2, 3, m).astype(np.float32)
        tb = engine.DTW_global_tb(a, b)
        c, pi, pj, _ = oracle.dtw_global_tb(a, b)
        assert bits(tb.cost) == bits(c) and np.array_equal(tb.i, pi) and np.array_equal(tb.j, pj)
        R0 = default_radius(n)
        assert bits(engine.DTW_global_slantedbanded_antidiagonalwise(a, b, R0)) == bits(oracle.dtw_banded(a, b, R0))


def test_error_behaviour(engine):
    ev = np.zeros(16, np.float32)
    engine.upload_reference([np.zeros(16, np.float32)], [np.zeros(16, np.float32)])
    bad = np.zeros(1, JOB_DTYPE)
    bad[0] = (0, 0, 0, 4, 1, 0, 0)  # zero length: dtw.cpp:274 asserts
    with pytest.raises(ra.RawDTWError) as e:
        engine.score_batch(bad, ev)
    assert e.value.status == 1
    bad[0] = (0, 0, 4, 4, -5, 0, 0)  # negative radius: dtw.cpp:277 asserts
    with pytest.raises(ra.RawDTWError):
        engine.score_batch(bad, ev)
    bad[0] = (28, 0, 4, 10, 1, 0, 0)  # window past the arena (fwd + rev = 32 floats)
    with pytest.raises(ra.RawDTWError) as e:
        engine.score_batch(bad, ev)
    assert e.value.status == 4
    bad[0] = (0, 0, 4, 4, 2, 0, 0)  # traceback of a banded job: rmap.cpp:223-225
    with pytest.raises(ra.RawDTWError) as e:
        engine.traceback_batch(bad, ev)
    assert e.value.status == 5
    assert len(engine.score_batch(np.zeros(0, JOB_DTYPE), ev)) == 0  # empty batch


def test_full_matrix_pipeline_guard(engine, oracle):
    """The four-wave pipelined full-matrix kernel publishes (strip << 21) | columns: a job with >= 2^21 columns
    (reachable through rawdtw_score_batch, not from the mapper's shapes) must take the one-wave variant and still
    match the reference (dtw.cpp:37-66) bit for bit."""
    rng = np.random.default_rng(77)
    nx, ny = (1 << 21) + 70, 1030  # >= 3 strips of 512 rows: would otherwise be class 60
    ref = rng.normal(size=nx).astype(np.float32)
    ev = rng.normal(size=ny).astype(np.float32)
    engine.upload_reference([ref], [ref])
    jobs = np.zeros(2, JOB_DTYPE)
    jobs[0] = (engine.reference_offset(0, 1), 0, ny, nx, -1, 0, 0)
    jobs[1] = (engine.reference_offset(0, 1), 0, ny, 3000, -1, 1, 0)  # an ordinary pipelined job next to it
    got = engine.score_batch(jobs, ev)
    want = np.array([oracle.dtw_global(ev, ref, 0), oracle.dtw_global(ev, ref[:3000], 1)], np.float32)
    assert_bits_equal(got, want, "full-matrix guard")


def test_properties_at_scale(engine):
    """Size-independent properties on a batch too large to check cell by cell on the CPU:
    DTW(x,x)=0, symmetry of the full DP, band never beats the full DP, identical jobs agree."""
    rng = np.random.default_rng(9)
    ref = rng.normal(size=1 << 20).astype(np.float32)
    events = ref[: 1 << 19].copy() + rng.normal(scale=0.3, size=1 << 19).astype(np.float32)
    engine.upload_reference([ref], [ref])
    base = engine.reference_offset(0, 1)
    nj = 200000
    n = rng.integers(2, 60, nj).astype(np.uint32)
    off = rng.integers(0, (1 << 19) - 64, nj).astype(np.uint32)
    jobs = np.zeros(4 * nj, JOB_DTYPE)
    R0 = np.maximum(1, (n.astype(np.float32) * np.float32(0.1)).astype(np.int32))
    for q in range(4):
        s = slice(q * nj, (q + 1) * nj)
        jobs["read_off"][s] = off
        jobs["ref_off"][s] = off.astype(np.uint64) + base
        jobs["n"][s] = n
        jobs["m"][s] = n
    jobs["band_radius"][:nj] = R0
    jobs["band_radius"][nj:2 * nj] = R0          # duplicate batch
    jobs["band_radius"][2 * nj:3 * nj] = -1      # full
    jobs["band_radius"][3 * nj:] = -1            # full, on (ref, ref): x against itself
    cost = engine.score_batch(jobs, events)
    banded, dup, full = cost[:nj], cost[nj:2 * nj], cost[2 * nj:3 * nj]
    assert np.array_equal(banded.view(np.uint32), dup.view(np.uint32))
    assert np.all(banded >= full)
    # x vs x: upload ref as events too
    self_jobs = jobs[3 * nj:].copy()
    c0 = engine.score_batch(self_jobs, ref[: 1 << 19])
    assert np.all(c0 == 0.0)
    # symmetry of the full DP: swap roles of the two arenas
    sym = engine.score_batch(jobs[2 * nj:2 * nj + 20000], events)
    engine.upload_reference([events], [events])
    b2 = engine.reference_offset(0, 1)
    j2 = jobs[2 * nj:2 * nj + 20000].copy()
    j2["ref_off"] = j2["read_off"].astype(np.uint64) + b2
    sym2 = engine.score_batch(j2, ref[: 1 << 19])
    assert np.array_equal(sym.view(np.uint32), sym2.view(np.uint32))


@pytest.mark.parametrize("border,fill", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_batch_matches_oracle_loop(engine, oracle, border, fill):
    """rawdtw_batch (DTW kernels + device fold + device select) == the reference's sequential loop
    (rmap.cpp:515-524 calling align_chain with the running best), chain by chain, bit for bit."""
    from oracle.loader import OrcOpt
    from rawalign_amd import synth

    ref = synth.make_reference([120000], seed=77)
    engine.upload_reference(ref.forward, ref.reverse)
    offs = {(0, st): engine.reference_offset(0, st) for st in (0, 1)}
    cb, info = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=96, max_chunks=3, decoys_per_read=3.0),
                                          seed=1234 + border * 2 + fill)
    engine.upload_events(cb.events)
    opt = ra.MapOpt(dtw_border_constraint=border, dtw_fill_method=fill)
    batch = ra.Batch(engine, opt, cb)
    batch.run()
    score, keep = batch.fetch()
    for mode, long_parts in ((0, 768), (1, 768), (2, 768), (3, 768), (3, 40), (3, 1), (4, 768)):
        # every fold kernel: wave per chain, lane per chain with 16 / 32 parts per round, lanes + a wave per long chain
        # (with thresholds that give the wave role none, some and all of this batch's chains), fold + select in one launch
        # out of LDS (sync-free batches; a batch keeps the form it was created under)
        engine.set_option("fold_mode", mode)
        engine.set_option("fold_long_parts", long_parts)
        b2 = ra.Batch(engine, opt, cb)
        b2.run()
        s2, k2 = b2.fetch()
        b2.close()
        assert np.array_equal(s2.view(np.uint32), score.view(np.uint32)) and np.array_equal(k2, keep), (mode, long_parts)
    engine.set_option("fold_mode", ra.DEFAULT_FOLD_MODE)
    engine.set_option("fold_long_parts", 768)
    oopt = OrcOpt(border, fill, 0.10, 0.4, 20.0, 1)
    arrays = {1: ref.forward[0], 0: ref.reverse[0]}
    strand_of = {offs[(0, 1)]: 1, offs[(0, 0)]: 0}
    n_cut = n_keep = 0
    for r in range(cb.n_reads):
        best = np.float32(0.0)
        for c in range(int(cb.chain_off[r]), int(cb.chain_off[r + 1])):
            a = cb.anchors[int(cb.anchor_off[c]):int(cb.anchor_off[c + 1])]
            ev = cb.events[int(cb.read_base[c]):]
            want = oracle.align_chain(a, arrays[strand_of[int(cb.ref_base[c])]], ev, oopt, float(best))
            assert bits(score[c]) == bits(want), (r, c, score[c], want)
            k = want >= np.float32(20.0)
            assert bool(keep[c]) == bool(k)
            if k and want > best:
                best = want
            n_cut += want == np.float32(-1e10)
            n_keep += bool(k)
    assert n_keep > 0 and n_cut > 0  # both the accept and the early-cut path were exercised
    # evaluate_reads (object API) agrees with the flat batch
    st = batch.info()
    assert st["n_chains"] == cb.n_chains and st["n_jobs"] > 0 and st["cells"] > 0


def test_align_chain_cigar_quirks(engine, oracle):
    """--dtw-output-cigar path: sparse (parts never exclude their last element, anchors counted
    twice) and global+full (offsets added to the last tuple only), SURVEY.md 8 a-4 (i)(ii)."""
    from oracle.loader import OrcOpt

    rng = np.random.default_rng(31)
    refsig = rng.normal(size=5000).astype(np.float32)
    engine.upload_reference([refsig], [refsig[::-1].copy()])
    q = np.array([3, 9, 14, 30, 31, 47, 80])
    t = np.array([100, 105, 111, 125, 126, 140, 171])
    anchors = np.zeros(len(q), ra.ANCHOR_DTYPE)
    anchors["query_position"] = q[::-1]
    anchors["target_position"] = t[::-1]
    events = (refsig[97:97 + 90] + rng.normal(scale=0.2, size=90)).astype(np.float32)
    for border, fill in [(1, 1), (1, 0), (0, 0)]:
        opt = ra.MapOpt(dtw_border_constraint=border, dtw_fill_method=fill)
        ch = ra.Chain(50.0, 0, 1, anchors)
        ra.align_chain(engine, ch, events, opt, cigar=True)
        oopt = OrcOpt(border, fill, 0.10, 0.4, 20.0, 1)
        sc, cost, pi, pj, pd = oracle.align_chain_cigar(anchors, refsig, events, oopt)
        assert bits(ch.alignment_score) == bits(sc)
        assert bits(ch.dtw_result.cost) == bits(cost)
        assert np.array_equal(ch.dtw_result.i, pi) and np.array_equal(ch.dtw_result.j, pj)
        assert np.array_equal(ch.dtw_result.difference.view(np.uint32), pd.view(np.uint32))
    with pytest.raises(AssertionError):  # rmap.cpp:223-225
        ra.align_chain(engine, ra.Chain(50.0, 0, 1, anchors), events, ra.MapOpt(dtw_border_constraint=0, dtw_fill_method=1), cigar=True)
    # score-only single chain, every min_score regime
    for ms in (-1e10, 0.0, 25.0, 1e6):
        opt = ra.MapOpt()
        ch = ra.align_chain(engine, ra.Chain(50.0, 0, 1, anchors), events, opt, cigar=False, min_score=ms)
        assert bits(ch.alignment_score) == bits(oracle.align_chain(anchors, refsig, events, OrcOpt(1, 1, 0.10, 0.4, 20.0, 1), ms))


def test_human_scale_reference_offsets(engine, oracle):
    """configs[3] (human CHM13: 2 x 3.1e9 floats = 24.8 GB of reference signal): window offsets beyond
    2^32 elements.  A 4.4e9-float arena (17.6 GB) is adopted from torch; jobs sit on both sides of 2^32."""
    if torch is None:
        pytest.skip("needs torch for the device arena")
    n_big = (1 << 32) + (1 << 27)
    try:
        arena = torch.empty(n_big, dtype=torch.float32, device="cuda:0")
    except RuntimeError as err:
        if "out of memory" not in str(err).lower():
            raise  # (a stale HIP error left behind by the library would land here too: that is a bug, not a skip)
        pytest.skip("not enough device memory for a 17.6 GB arena")
    rng = np.random.default_rng(12)
    seg = rng.normal(size=200000).astype(np.float32)
    places = [0, (1 << 32) - 100000, (1 << 32) + 300007, n_big - len(seg)]  # the second one straddles 2^32
    for p in places:
        arena[p:p + len(seg)] = torch.from_numpy(seg).cuda()
    torch.cuda.synchronize()
    engine.set_reference_device(arena.data_ptr(), n_big, keepalive=arena)
    events = (seg[:50000] + rng.normal(scale=0.2, size=50000)).astype(np.float32)
    cases, jobs = [], []
    for k in range(400):
        n = int(rng.integers(2, 70)) if k % 8 else int(rng.integers(200, 1500))
        m = max(2, int(n * rng.uniform(0.6, 1.3)))
        eo = int(rng.integers(0, len(events) - n))
        so = int(rng.integers(0, len(seg) - m))
        base = places[k % 4]
        R0 = -1 if k % 5 == 0 else max(1, int(np.float32(n) * np.float32(0.1)))
        jobs.append((base + so, eo, n, m, R0, k & 1, 0))
        cases.append((events[eo:eo + n], seg[so:so + m], R0, k & 1))
    jobs = np.array(jobs, dtype=JOB_DTYPE)
    got = engine.score_batch(jobs, events)
    want = np.array([oracle.dtw_global(a, b, ex) if R0 < 0 else oracle.dtw_banded(a, b, R0, ex)
                     for a, b, R0, ex in cases], np.float32)
    assert_bits_equal(got, want, "offsets beyond 2^32")
    # the same arena through the batch API, planned on the host and on the device (64-bit span sources, 32-bit LDS math)
    from rawalign_amd.align import CandidateBatch

    anchors, anchor_off, ref_base = [], [0], []
    for c in range(48):
        na = int(rng.integers(2, 60))
        q = np.cumsum(rng.integers(2, 14, na)) + int(rng.integers(0, 20000))
        t = np.cumsum(rng.integers(2, 12, na)) + int(rng.integers(0, 100000))
        a = np.zeros(na, ra.ANCHOR_DTYPE)
        a["query_position"] = q[::-1]; a["target_position"] = t[::-1]
        anchors.append(a); anchor_off.append(anchor_off[-1] + na); ref_base.append(places[c % 4])
    cb = CandidateBatch(events, np.array([0, 20, 48], np.uint64), np.array(anchor_off, np.uint64), np.concatenate(anchors),
                        np.array(ref_base, np.uint64), np.zeros(48, np.uint32))
    engine.upload_events(events)
    res = {}
    for dev in (0, 1):
        with planner_options(engine, device_plan=dev, device_plan_min_jobs=0):
            b = ra.Batch(engine, ra.MapOpt(), cb)
            assert b.verify_plan() == bool(dev)
            b.run()
            res[dev] = b.fetch(with_job_costs=True)
            b.close()
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8))
    assert len(res[0][2]) > 1000
    del arena


def test_yeast_like_global_full_cigar(engine, oracle):
    """configs[2]: multi-sequence reference, border=global, fill=full, --dtw-output-cigar on the best
    chain: scores through the batch path, traceback through align_chain(cigar=True), both strands."""
    from oracle.loader import OrcOpt
    from rawalign_amd import synth

    ref = synth.make_reference([30000, 80000, 12000, 55000], seed=41)
    engine.upload_reference(ref.forward, ref.reverse)
    offs = {(s, st): engine.reference_offset(s, st) for s in range(ref.n_seq) for st in (0, 1)}
    cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=40, max_chunks=3), seed=5)
    engine.upload_events(cb.events)
    opt = ra.MapOpt(dtw_border_constraint=0, dtw_fill_method=0, flag=0x2 | 0x4)
    batch = ra.Batch(engine, opt, cb)
    batch.run()
    score, keep = batch.fetch()
    oopt = OrcOpt(0, 0, 0.10, 0.4, 20.0, 1)
    inv = {v: k for k, v in offs.items()}
    checked_tb = 0
    for r in range(cb.n_reads):
        best = np.float32(0.0)
        first = True
        for c in range(int(cb.chain_off[r]), int(cb.chain_off[r + 1])):
            a = cb.anchors[int(cb.anchor_off[c]):int(cb.anchor_off[c + 1])]
            seq, st = inv[int(cb.ref_base[c])]
            arr = ref.forward[seq] if st == 1 else ref.reverse[seq]
            ev = cb.events[int(cb.read_base[c]):]
            want = oracle.align_chain(a, arr, ev, oopt, float(best))
            assert bits(score[c]) == bits(want)
            if want >= np.float32(20.0) and want > best:
                best = want
            if first and keep[c] and checked_tb < 6:
                # rmap.cpp:715-717 on chains[0]: traceback with the global-mode offset quirk
                ch = ra.Chain(1.0, seq, st, a)
                ra.align_chain(engine, ch, ev, opt, cigar=True)
                sc, cost, pi, pj, pd = oracle.align_chain_cigar(a, arr, ev, oopt)
                assert bits(ch.alignment_score) == bits(sc) and bits(ch.dtw_result.cost) == bits(cost)
                assert np.array_equal(ch.dtw_result.i, pi) and np.array_equal(ch.dtw_result.j, pj)
                assert np.array_equal(ch.dtw_result.difference.view(np.uint32), pd.view(np.uint32))
                checked_tb += 1
            first = False
    assert checked_tb > 0


def test_planner_options_do_not_change_results(oracle):
    """Every tuning knob (kernel class boundaries, tile sizes, micro paths, stream layout) must leave
    the costs bit-identical: they only move jobs between kernels."""
    rng = np.random.default_rng(21)
    cases = []
    for t in range(3000):
        n = int(rng.integers(1, 110))
        m = max(1, int(round(n * rng.uniform(0.4, 1.5))))
        R0 = default_radius(n) if t % 4 else int(rng.integers(0, 12))
        cases.append((rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32), R0, t & 1))
    jobs, ev, rf = make_arena_jobs(cases)
    want = oracle_costs(oracle, jobs, ev, rf)
    settings = [
        {},
        {"lane_hi": 1, "lane_hi_max_n": 200},
        {"micro_max_n": 0},
        {"micro_max_n": 4, "tile_lds_floats": 2048, "tile_max_jobs": 64},
        {"lane_max_radius": 0, "lane_max_n": 12},
        {"lane_max_radius": 1, "lane_hi": 1, "lane_hi_max_n": 40, "serial_launches": 1},
        {"tile_lds_floats": 30000, "tile_max_jobs": 4096},
        {"grp16": 0},
        {"grp8": 0},
        {"grp8": 0, "merge_small": 0},
        {"full_wg": 0},
        {"lane_max_radius": 0, "lane_max_n": 8, "grp16": 1},
        {"merge_small": 0},
        {"sort_n": 17, "sort_r1_n": 9, "sort_r3": 1},
        {"sort_n": 5, "sorted_tile_jobs": 128},
        {"merge_small": 0, "grp16": 0},
        {"plan_threads": 1},
        {"plan_threads": 7, "tile_max_jobs": 64},
    ]
    for st in settings:
        eng = ra.Engine(0)
        for k, v in st.items():
            eng.set_option(k, v)
        assert_bits_equal(run(eng, jobs, ev, rf), want, f"options {st}")
        eng.close()
    eng = ra.Engine(0)
    with pytest.raises(ra.RawDTWError):
        eng.set_option("no_such_option", 1)


@pytest.mark.parametrize("device_plan", [0, 1])
def test_batch_edge_cases(engine, oracle, device_plan):
    """Reads without chains, a batch without any chain, and single-anchor chains (no DTW call at all:
    align_chain returns 0*bonus - 0, which fails dtw_min_score) -- through the job list and through the sync-free path."""
    from rawalign_amd.align import CandidateBatch

    with planner_options(engine, device_plan=device_plan, device_plan_min_jobs=0):
        rng = np.random.default_rng(2)
        refsig = rng.normal(size=3000).astype(np.float32)
        engine.upload_reference([refsig], [refsig[::-1].copy()])
        events = rng.normal(size=500).astype(np.float32)
        engine.upload_events(events)
        base = engine.reference_offset(0, 1)
        # no chains at all
        cb0 = CandidateBatch(events, np.zeros(4, np.uint64), np.zeros(1, np.uint64), np.zeros(0, ra.ANCHOR_DTYPE),
                             np.zeros(0, np.uint64), np.zeros(0, np.uint32))
        b0 = ra.Batch(engine, ra.MapOpt(), cb0)
        b0.run()
        s0, k0 = b0.fetch()
        assert len(s0) == 0 and len(k0) == 0
        # read 0: nothing; read 1: a single-anchor chain and a real chain; read 2: nothing
        a1 = np.zeros(1, ra.ANCHOR_DTYPE); a1[0] = (100, 7)
        q = np.array([5, 12, 30, 31, 60]); t = np.array([200, 206, 221, 223, 250])
        a2 = np.zeros(5, ra.ANCHOR_DTYPE); a2["query_position"] = q[::-1]; a2["target_position"] = t[::-1]
        ev2 = events.copy()
        ev2[5:61] = refsig[200:256]
        engine.upload_events(ev2)
        cb = CandidateBatch(ev2, np.array([0, 0, 2, 2], np.uint64), np.array([0, 1, 6], np.uint64),
                            np.concatenate([a1, a2]), np.array([base, base], np.uint64), np.zeros(2, np.uint32))
        b = ra.Batch(engine, ra.MapOpt(dtw_min_score=5.0), cb)
        b.run()
        score, keep = b.fetch()
        assert score[0] == 0.0 and keep[0] == 0
        from oracle.loader import OrcOpt

        want = oracle.align_chain(a2, refsig, ev2, OrcOpt(1, 1, 0.10, 0.4, 5.0, 1), 0.0)
        assert bits(score[1]) == bits(want) and bool(keep[1]) == bool(want >= np.float32(5.0))
        assert b.verify_plan() == bool(device_plan)
        # anchors that do not ascend (a chain the mapper could never produce): both paths refuse the batch, with the
        # host planner's wording (the sync-free path hands such batches over to it)
        bad = a2.copy(); bad["query_position"][1] = 200
        cbb = CandidateBatch(ev2, np.array([0, 1], np.uint64), np.array([0, 5], np.uint64), bad, np.array([base], np.uint64),
                             np.zeros(1, np.uint32))
        with pytest.raises(ra.RawDTWError) as e:   # (the sync-free path reports it when the results are fetched)
            bb = ra.Batch(engine, ra.MapOpt(), cbb)
            bb.run()
            bb.fetch()
        assert "job " in str(e.value)


@pytest.mark.gpu
def test_device_planned_batch_at_scale():
    """A million jobs (several tiles per chain, thousands of tiles, the persistent grid's queue): plan self-check, batch
    totals and every per-job cost of the sync-free path against the host-planned batch."""
    from rawalign_amd import synth

    ref = synth.make_reference([600000], seed=93)
    results = {}
    for dev in (0, 1):
        eng = ra.Engine(0)
        eng.set_option("device_plan", dev)
        eng.upload_reference(ref.forward, ref.reverse)
        offs = {(0, st): eng.reference_offset(0, st) for st in (0, 1)}
        cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=4500), seed=99)
        eng.upload_events(cb.events)
        batch = ra.Batch(eng, ra.MapOpt(), cb)
        assert batch.verify_plan() == bool(dev)
        info = batch.info()
        assert info["n_jobs"] > 1_100_000
        batch.run()
        results[dev] = batch.fetch(with_job_costs=True) + (info,)
        eng.close()
    (s0, k0, c0, i0), (s1, k1, c1, i1) = results[0], results[1]
    assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32))
    assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32)) and np.array_equal(k0, k1)
    for key in ("n_jobs", "cells", "algorithmic_bytes", "n_lane_jobs", "n_wave_band_jobs", "n_full_jobs"):
        assert i0[key] == i1[key], key


@pytest.mark.gpu
@pytest.mark.parametrize("border,fill", [(1, 1), (0, 1), (1, 0)])
def test_device_planned_batch_matches_host_planned(oracle, border, fill):
    """rawdtw_batch_create through the sync-free path (rawdtw_stream.hip: planning on the device, tiles laid out in LDS):
    its job records and tile boundaries pass the self-check, and scores / keeps / per-job costs equal the host-planned
    batch bit for bit."""
    from rawalign_amd import synth

    ref = synth.make_reference([150000], seed=91)
    results = {}
    for dev in (0, 1):
        eng = ra.Engine(0)
        eng.set_option("device_plan", dev)
        eng.set_option("device_plan_min_jobs", 0)
        eng.upload_reference(ref.forward, ref.reverse)
        offs = {(0, st): eng.reference_offset(0, st) for st in (0, 1)}
        cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=700, max_chunks=4, decoys_per_read=2.0),
                                           seed=4321 + border)
        eng.upload_events(cb.events)
        batch = ra.Batch(eng, ra.MapOpt(dtw_border_constraint=border, dtw_fill_method=fill), cb)
        planned_on_device = batch.verify_plan()            # raises on the first broken invariant
        # the sync-free path takes sparse + banded batches; everything else is planned on the host from the job list
        assert planned_on_device == bool(dev and fill == 1 and border == 1)
        info = batch.info()
        batch.run()
        results[dev] = batch.fetch(with_job_costs=True) + (info,)
        eng.close()
    (s0, k0, c0, i0), (s1, k1, c1, i1) = results[0], results[1]
    assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32))
    assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32)) and np.array_equal(k0, k1)
    for key in ("n_jobs", "cells", "algorithmic_bytes", "n_lane_jobs", "n_wave_band_jobs", "n_full_jobs"):
        assert i0[key] == i1[key], key


def test_traceback_steps_form_equals_ij_arrays(engine):
    """rawdtw_traceback_batch_steps: a path as one step byte and one distance an element (what leaves the device) must expand
    to exactly the (i, j, distance) arrays of rawdtw_traceback_batch (which the golden paths pin to the reference) -- with the
    caller's offsets dense (the stretch comes home whole: through the landing zone, and in place when the arrays are page-
    locked) and with gaps between the paths (job by job), exclude_last pops included."""
    import ctypes as C

    rng = np.random.default_rng(41)
    lib = engine.lib
    cases = []
    for k in range(60):
        n, m = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        cases.append((rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32), -1, k % 3 == 0))
    jobs, ev, rf = make_arena_jobs(cases)
    engine.upload_reference([rf], [rf])
    jobs["ref_off"] += engine.reference_offset(0, 1)
    want = engine.traceback_batch(jobs, ev)
    caps = jobs["n"].astype(np.uint64) + jobs["m"].astype(np.uint64) - 1
    vp = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
    for gaps, pinned in ((0, False), (0, True), (7, False)):
        off = np.zeros(len(jobs), np.uint64)
        off[1:] = np.cumsum(caps + np.uint64(gaps))[:-1]
        total = int(off[-1] + caps[-1]) + 8
        ptrs = []
        if pinned:
            def alloc(nbytes):
                p = C.c_void_p()
                assert lib.rawdtw_host_alloc(nbytes, C.byref(p)) == 0
                ptrs.append(p)
                return p
            step = np.frombuffer((C.c_char * total).from_address(alloc(total).value), np.uint8, total)
            dist = np.frombuffer((C.c_char * (4 * total)).from_address(alloc(4 * total).value), np.float32, total)
        else:
            step, dist = np.zeros(total, np.uint8), np.zeros(total, np.float32)
        step[:] = 0xEE
        cost, plen = np.zeros(len(jobs), np.float32), np.zeros(len(jobs), np.uint32)
        engine._check(lib.rawdtw_traceback_batch_steps(engine._ctx, vp(jobs), len(jobs), vp(ev), len(ev), vp(cost), vp(off), vp(plen), vp(step), vp(dist)))
        for k, w in enumerate(want):
            s, n = int(off[k]), int(plen[k])
            assert n == len(w) and bits(cost[k]) == bits(w.cost)
            st = step[s:s + n]
            assert st[0] == 0 and np.array_equal(np.cumsum(st & 1), w.i) and np.array_equal(np.cumsum(st >> 1), w.j)
            assert np.array_equal(dist[s:s + n].view(np.uint32), w.difference.view(np.uint32))
            if gaps:
                assert np.all(step[s + int(caps[k]):s + int(caps[k]) + gaps] == 0xEE)   # (nothing written between the paths)
        for p in ptrs:
            lib.rawdtw_host_free(p)
