"""The sync-free batch path (rawdtw_runs.hip: k_scan, k_side, k_plan, k_wide, k_runs, k_fold_select) against the oracle: every shape class the tiles' passes and the side list take
(radii 1..3 with any slant, shorter sides down to 1-2 events, wide bands, bands it declines), its tuning knobs, the
incremental event upload and the shared / device-resident inputs.  Parts are the DTW sub-problems align_chain issues
between consecutive anchors (src/rmap.cpp:248-293) through DTW_global_slantedbanded_antidiagonalwise (src/dtw.cpp:273-520)."""
import ctypes as C

import numpy as np
import pytest

try:  # PyTorch bundles its own HIP runtime: when both live in one process, torch has to come up first
    import torch

    torch.cuda.is_available()
except Exception:  # pragma: no cover
    torch = None

import rawalign_amd as ra
from rawalign_amd.align import CandidateBatch
from tests.golden_util import bits

pytestmark = pytest.mark.gpu


def _chains(rng, n_reads, ref_len, shapes, parts_range=(1, 40)):
    """One read per chain group; every chain's parts draw (dq, dt) from `shapes`(rng) -> anchors end-first."""
    events, chain_off, anchor_off, anchors, ref_base, read_base = [], [0], [0], [], [], []
    ev_at = 0
    for r in range(n_reads):
        n_chains = int(rng.integers(1, 4))
        read_len = 0
        per = []
        for _ in range(n_chains):
            parts = int(rng.integers(*parts_range))
            dq, dt = zip(*[shapes(rng) for _ in range(parts)]) if parts else ((), ())
            q = np.concatenate([[int(rng.integers(0, 5))], np.cumsum(dq) + 0]).astype(np.int64)
            q[1:] += q[0]
            t0 = int(rng.integers(0, ref_len - int(np.sum(dt)) - 2))
            t = np.concatenate([[t0], t0 + np.cumsum(dt)]).astype(np.int64)
            per.append((q, t))
            read_len = max(read_len, int(q[-1]) + 1)
        events.append(rng.normal(size=read_len).astype(np.float32))
        for q, t in per:
            a = np.zeros(len(q), ra.ANCHOR_DTYPE)
            a["query_position"] = q[::-1]
            a["target_position"] = t[::-1]
            anchors.append(a)
            anchor_off.append(anchor_off[-1] + len(a))
            read_base.append(ev_at)
            ref_base.append(int(rng.integers(0, 2)))  # strand slot, resolved by the caller
        chain_off.append(len(anchor_off) - 1)
        ev_at += read_len
    return (np.concatenate(events), np.array(chain_off, np.uint64), np.array(anchor_off, np.uint64), np.concatenate(anchors),
            np.array(ref_base), np.array(read_base, np.uint32))


def _oracle_check(oracle, cb, ref_arrays, strand_of, score, keep, job_cost, opt):
    from oracle.loader import OrcOpt

    oopt = OrcOpt(1, 1, opt.dtw_band_radius_frac, opt.dtw_match_bonus, opt.dtw_min_score, 1)
    j = 0
    for r in range(cb.n_reads):
        best = np.float32(0.0)
        for c in range(int(cb.chain_off[r]), int(cb.chain_off[r + 1])):
            a = cb.anchors[int(cb.anchor_off[c]):int(cb.anchor_off[c + 1])]
            arr = ref_arrays[strand_of[c]]
            ev = cb.events[int(cb.read_base[c]):]
            parts = len(a) - 1
            for p in range(parts):  # every part's cost, bit for bit (dtw.cpp:273-520 with rmap.cpp:270,276)
                s, e = a[parts - p], a[parts - p - 1]
                n = int(e["query_position"]) - int(s["query_position"]) + 1
                m = int(e["target_position"]) - int(s["target_position"]) + 1
                R0 = max(1, int(np.float32(n) * np.float32(opt.dtw_band_radius_frac)))
                want = oracle.dtw_banded(ev[int(s["query_position"]):int(s["query_position"]) + n],
                                         arr[int(s["target_position"]):int(s["target_position"]) + m], R0, p != parts - 1)
                assert bits(job_cost[j]) == bits(want), (r, c, p, n, m, R0, job_cost[j], want)
                j += 1
            want_s = oracle.align_chain(a, arr, ev, oopt, float(best))
            assert bits(score[c]) == bits(want_s), (r, c, score[c], want_s)
            k = want_s >= np.float32(opt.dtw_min_score)
            assert bool(keep[c]) == bool(k)
            if k and want_s > best:
                best = want_s
    assert j == len(job_cost)


def _tiny(rng):     # the bulk of sparse mode: 2..8 events, slanted or square
    dq = int(rng.integers(1, 8))
    return dq, max(1, dq + int(rng.integers(-2, 3)))


def _medium(rng):   # radii 1..3, shorter side down to 2, up to the tile class's longest side and a little beyond
    dq = int(rng.integers(1, 40))
    return dq, max(1, int(round(dq * rng.uniform(0.3, 2.2))))


def _sprinkled(rng):  # the bench batch's mix: small parts with a radius-3 part now and then (a few a tile: partly filled
    if rng.random() < 0.03:  # waves of quad_dp_r3, four lanes a job) -- read side 20..39, the other side close to it
        dq = int(rng.integers(20, 40))
        return dq, max(1, dq + int(rng.integers(-9, 10)))
    return _tiny(rng)


def _degenerate(rng):  # windows of one or two elements on either side (anchors that share a position), next to small ones
    return int(rng.integers(0, 3)), int(rng.integers(0, 4))


def _wide(rng):     # side list: 8-lane, 16-lane and wave-per-job classes
    dq = int(rng.choice([3, 12, 45, 90, 160, 400]))
    return dq, max(1, int(round(dq * rng.uniform(0.6, 1.6))))


@pytest.mark.parametrize("shapes,opts", [
    (_tiny, {}), (_medium, {}), (_wide, {}), (_degenerate, {}), (_sprinkled, {}), (_sprinkled, {"tile_lds_floats": 2048}), (_degenerate, {"stream_tile_radius": 1}),
    (_medium, {"stream_threads": 512, "tile_lds_floats": 9000}),
    (_medium, {"tile_lds_floats": 2048}), (_tiny, {"micro_max_n": 0}), (_medium, {"micro_max_n": 4, "lane_max_n": 20}),
    (_medium, {"lane_max_radius": 1}), (_wide, {"stream_threads": 512}),
    # tiles of radius <= 2 / <= 1: the wider lane radii go to the side list's lane classes (lane_global_wave, lane_dp_r12)
    (_medium, {"stream_tile_radius": 2}), (_medium, {"stream_tile_radius": 1}), (_wide, {"stream_tile_radius": 2}),
    # the diagnostic instance of k_runs (profiling runs: scripts/pmc_debug_masks.sh, stream_probe.py) -- no mask set (128), with
    # the phase stamps (256), with the tiles dealt by block index (8): it has to score like the production instance
    (_medium, {"stream_debug": 128}), (_wide, {"stream_debug": 256}), (_tiny, {"stream_debug": 256 | 8}),
    # the side list's launch forked onto the context's second stream and joined before the fold, instead of in line
    (_wide, {"wide_beside": 1}), (_medium, {"wide_beside": 1, "wide_blocks": 7}),
])
def test_stream_path_shapes_against_oracle(oracle, shapes, opts):
    rng = np.random.default_rng(hash((shapes.__name__, tuple(sorted(opts)))) & 0xFFFF)
    ref = [rng.normal(size=60000).astype(np.float32), rng.normal(size=60000).astype(np.float32)]
    eng = ra.Engine(0)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.upload_reference([ref[0]], [ref[1]])
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, 120, 60000, shapes)
    strand_of = [1 if s == 0 else 0 for s in slot]  # slot 0 = forward array (strand 1, rmap.cpp:182-188)
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    eng.upload_events(events)
    opt = ra.MapOpt(dtw_min_score=5.0)
    b = ra.Batch(eng, opt, cb)
    assert b.verify_plan() is True  # the sync-free path took the batch, and its records pass the self-check
    b.run()
    score, keep, jc = b.fetch(with_job_costs=True)
    _oracle_check(oracle, cb, {1: ref[0], 0: ref[1]}, strand_of, score, keep, jc, opt)
    info = b.info()
    assert info["n_jobs"] == len(jc) and info["n_lane_jobs"] + info["n_wave_band_jobs"] == len(jc)


@pytest.mark.parametrize("n_reads,parts_range", [(900, (1, 4)), (12, (300, 500)), (200, (1, 120)), (700, (0, 3))])
def test_stream_path_chain_lengths_against_oracle(oracle, n_reads, parts_range):
    """The scan and the planner find an anchor's chain from a chain-start bit mask of its unit / tile (k_scan: 8192 anchors,
    k_plan: 512), built from the chains' offsets in as many rounds of loads as the range has chains to 64: batches of very
    short chains (hundreds a tile, several passes a tile for the 32-entry run table), very long ones (a chain over several
    units), a mix, and chains of a single anchor (no part, no job: rmap.cpp:251 does not enter its loop) between the others."""
    rng = np.random.default_rng(n_reads)
    ref = [rng.normal(size=60000).astype(np.float32), rng.normal(size=60000).astype(np.float32)]
    eng = ra.Engine(0)
    eng.upload_reference([ref[0]], [ref[1]])
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, n_reads, 60000, _tiny, parts_range)
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    eng.upload_events(events)
    opt = ra.MapOpt(dtw_min_score=5.0)
    b = ra.Batch(eng, opt, cb)
    assert b.verify_plan() is True
    b.run()
    score, keep, jc = b.fetch(with_job_costs=True)
    _oracle_check(oracle, cb, {1: ref[0], 0: ref[1]}, strand_of, score, keep, jc, opt)


@pytest.mark.parametrize("chains_per_read", [12, 45])
def test_reads_with_many_chains(oracle, chains_per_read):
    """k_fold_select takes 8 reads a wave, a chain a lane: with more than 64 chains among them a lane folds several chains
    one after the other (the sweep down the anchor list meets them last to first), and beyond 256 the accept/cut loop reads
    the scores back from memory instead of LDS.  (--num-best-chains bounds the chains of a read in the mapper: roptions.c:19;
    the boundary takes any number.)"""
    rng = np.random.default_rng(chains_per_read)
    ref = [rng.normal(size=60000).astype(np.float32), rng.normal(size=60000).astype(np.float32)]
    eng = ra.Engine(0)
    eng.upload_reference([ref[0]], [ref[1]])
    events, chain_off, anchor_off, anchors, read_base, slot = [], [0], [0], [], [], []
    ev_at = 0
    for r in range(40):
        read_len = 0
        for _ in range(chains_per_read):
            parts = int(rng.integers(0, 12))
            dq = rng.integers(1, 7, size=parts); dt = np.maximum(1, dq + rng.integers(-1, 2, size=parts))
            q = np.concatenate([[int(rng.integers(0, 5))], np.cumsum(dq)]).astype(np.int64); q[1:] += q[0]
            t0 = int(rng.integers(0, 60000 - int(dt.sum()) - 2))
            t = np.concatenate([[t0], t0 + np.cumsum(dt)]).astype(np.int64)
            a = np.zeros(len(q), ra.ANCHOR_DTYPE)
            a["query_position"] = q[::-1]; a["target_position"] = t[::-1]
            anchors.append(a); anchor_off.append(anchor_off[-1] + len(a)); read_base.append(ev_at); slot.append(int(rng.integers(0, 2)))
            read_len = max(read_len, int(q[-1]) + 1)
        events.append(rng.normal(size=read_len).astype(np.float32))
        chain_off.append(len(anchor_off) - 1)
        ev_at += read_len
    events = np.concatenate(events)
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    cb = CandidateBatch(events, np.array(chain_off, np.uint64), np.array(anchor_off, np.uint64), np.concatenate(anchors), ref_base,
                        np.array(read_base, np.uint32))
    eng.upload_events(events)
    opt = ra.MapOpt(dtw_min_score=-12.0)  # (random signals: scores around -10: some chains pass, some are cut by the running best)
    b = ra.Batch(eng, opt, cb)
    assert b.verify_plan() is True
    b.run()
    score, keep, jc = b.fetch(with_job_costs=True)
    _oracle_check(oracle, cb, {1: ref[0], 0: ref[1]}, strand_of, score, keep, jc, opt)
    assert keep.any()


def test_run_reads_the_events_uploaded_after_create(oracle):
    """A run reads the context's arenas as they are when it is enqueued (include/rawdtw.h): events uploaded between
    rawdtw_batch_create and rawdtw_batch_run count -- for the tiles' passes and for the side list's wide bands alike (the
    wide bands' launch goes out with the planning launches only inside rawdtw_batch_submit, where nothing can come between)."""
    rng = np.random.default_rng(5)
    ref = [rng.normal(size=60000).astype(np.float32), rng.normal(size=60000).astype(np.float32)]
    eng = ra.Engine(0)
    eng.upload_reference([ref[0]], [ref[1]])
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, 120, 60000, _wide)
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    eng.upload_events(events)
    opt = ra.MapOpt(dtw_min_score=5.0)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    b = ra.Batch(eng, opt, cb)
    events2 = rng.normal(size=len(events)).astype(np.float32)
    eng.upload_events(events2)
    b.run()
    score, keep, jc = b.fetch(with_job_costs=True)
    cb2 = CandidateBatch(events2, chain_off, anchor_off, anchors, ref_base, read_base)
    _oracle_check(oracle, cb2, {1: ref[0], 0: ref[1]}, strand_of, score, keep, jc, opt)


def test_batch_that_runs_out_of_pass_slots_is_redone_through_the_job_list(oracle):
    """A tile whose parts do not fit one pass (image budget, or more than 32 runs) takes further passes, each with a slot of
    copy orders from a pool; a batch that runs out of slots is declined by the planner -- the later launches return at
    once -- and rawdtw_batch_fetch redoes it through the job list: same results as ever."""
    rng = np.random.default_rng(77)
    ref = [rng.normal(size=60000).astype(np.float32), rng.normal(size=60000).astype(np.float32)]
    eng = ra.Engine(0)
    eng.set_option("tile_lds_floats", 2048)
    eng.set_option("pass_pool", 1)
    eng.upload_reference([ref[0]], [ref[1]])
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, 300, 60000, _medium, (1, 30))
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    eng.upload_events(events)
    opt = ra.MapOpt(dtw_min_score=5.0)
    b = ra.Batch(eng, opt, cb)
    assert b.verify_plan() is False  # declined: more passes than slots
    b.run()
    score, keep, jc = b.fetch(with_job_costs=True)
    _oracle_check(oracle, cb, {1: ref[0], 0: ref[1]}, strand_of, score, keep, jc, opt)
    eng.set_option("pass_pool", -1)
    b2 = ra.Batch(eng, opt, cb)
    assert b2.verify_plan() is True
    b2.run()
    s2, k2, j2 = b2.fetch(with_job_costs=True)
    assert np.array_equal(s2.view(np.uint32), score.view(np.uint32)) and np.array_equal(k2, keep) and np.array_equal(j2.view(np.uint32), jc.view(np.uint32))


@pytest.mark.parametrize("ref_len,tail", [(60001, 0), (60003, 0), (60002, 1)])
def test_windows_that_end_with_the_arenas(oracle, ref_len, tail):
    """The last part of the last chain ends on the last reference element and the last event (arena lengths that are
    not multiples of four: the staging copies whole 16-byte pieces), `tail` elements short of it in the second case."""
    rng = np.random.default_rng(ref_len)
    ref = [rng.normal(size=ref_len).astype(np.float32), rng.normal(size=ref_len).astype(np.float32)]
    eng = ra.Engine(0)
    eng.upload_reference([ref[0]], [ref[1]])
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, 40, ref_len, _tiny)
    # the event array ends with the last read's last used event already (_chains); move the last read's chains to the end
    # of their strand arrays
    for c in range(int(chain_off[-2]), int(chain_off[-1])):
        a = anchors[int(anchor_off[c]):int(anchor_off[c + 1])]
        a["target_position"] += np.uint32(ref_len - 1 - tail - int(a["target_position"].max()))
    if tail:
        events = np.concatenate([events, rng.normal(size=tail).astype(np.float32)])
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    eng.upload_events(events)
    opt = ra.MapOpt(dtw_min_score=5.0)
    b = ra.Batch(eng, opt, cb)
    assert b.verify_plan() is True
    b.run()
    score, keep, jc = b.fetch(with_job_costs=True)
    _oracle_check(oracle, cb, {1: ref[0], 0: ref[1]}, strand_of, score, keep, jc, opt)


def test_wave_per_job_bands_at_the_register_layouts_edges(oracle):
    """k_wide's wave-per-job bodies hold a band of K slots in C = 1, 2, 3 or 5 registers a lane (wband_gen<C>: K <= 64 C; odd beyond two, for the
    LDS banks).  Parts whose radius puts K on both sides of every change of layout -- radius 63 / 64 (K = 64 fills the one-register wave: its last
    lane's neighbour is the DPP shift's fill; K = 65 is the first two-register band), 127 / 128, 191 / 192, and 255 (K = 256, the widest
    band the sync-free path takes) -- square and slanted, with the longer side on either arena, between small parts: every part cost and every
    chain score bit for bit against the oracle (dtw.cpp:298-303 for P and S, rmap.cpp:276 for the radius)."""
    rng = np.random.default_rng(4242)
    ref = [rng.normal(size=60000).astype(np.float32), rng.normal(size=60000).astype(np.float32)]
    eng = ra.Engine(0)
    eng.upload_reference([ref[0]], [ref[1]])
    # (dq, dt): n = dq + 1, m = dt + 1, r0 = int(0.1 n), R = r0 + ceil((N - M) r0 / N) with N the longer side (dtw.cpp:298-300)
    edges = [(629, 629), (639, 639), (571, 630), (1269, 1269), (1279, 1279), (1160, 1280), (2549, 2549), (2320, 2570), (700, 630), (1915, 1915), (1925, 1925), (1740, 1926)]

    def radius(dq, dt):
        n, m = dq + 1, dt + 1
        r0, N, M = max(1, int(np.float32(n) * np.float32(0.1))), max(n, m), min(n, m)
        return r0 + ((N - M) * r0 + N - 1) // N
    assert [radius(*e) for e in edges][:8] == [63, 64, 63, 127, 128, 127, 255, 255] and [radius(*e) for e in edges][9:11] == [191, 192]
    state = {"k": 0}

    def shapes(r):
        state["k"] += 1
        return edges[(state["k"] // 9) % len(edges)] if state["k"] % 9 == 4 else _tiny(r)
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, 40, 60000, shapes, (1, 14))
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    eng.upload_events(events)
    opt = ra.MapOpt(dtw_min_score=5.0)
    b = ra.Batch(eng, opt, cb)
    assert b.verify_plan() is True
    b.run()
    score, keep, jc = b.fetch(with_job_costs=True)
    _oracle_check(oracle, cb, {1: ref[0], 0: ref[1]}, strand_of, score, keep, jc, opt)
    assert b.info()["n_wave_band_jobs"] >= len(edges)


def test_band_too_wide_for_the_stream_path_is_redone_through_the_job_list(oracle):
    """A part whose band needs more than 256 offsets: the sync-free path declines the batch at fetch and the job-list
    path (register-resident wave kernel with more chunks) produces the same answers the oracle gives."""
    rng = np.random.default_rng(77)
    ref = [rng.normal(size=40000).astype(np.float32), rng.normal(size=40000).astype(np.float32)]
    eng = ra.Engine(0)
    eng.upload_reference([ref[0]], [ref[1]])
    state = {"k": 0}

    def shapes(r):
        state["k"] += 1
        return (3000, 2800) if state["k"] == 17 else _tiny(r)  # n = 3001 -> band_radius 300 (rmap.cpp:276)
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, 30, 40000, shapes)
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    eng.upload_events(events)
    opt = ra.MapOpt(dtw_min_score=5.0)
    b = ra.Batch(eng, opt, cb)
    b.run()
    score, keep, jc = b.fetch(with_job_costs=True)
    assert b.verify_plan() is False  # now a job-list batch
    _oracle_check(oracle, cb, {1: ref[0], 0: ref[1]}, strand_of, score, keep, jc, opt)


def test_incremental_events_shared_reference_and_resident_arrays(oracle):
    """rawdtw_events_reserve / rawdtw_events_append (a round uploads only the new events, rmap.cpp:554-567),
    rawdtw_share_reference (one resident arena per GPU) and "resident_arrays" (anchors already on the device): same
    scores as the plain upload."""
    if torch is None:
        pytest.skip("torch not importable")
    rng = np.random.default_rng(5)
    ref = [rng.normal(size=40000).astype(np.float32), rng.normal(size=40000).astype(np.float32)]
    owner = ra.Engine(0)
    owner.upload_reference([ref[0]], [ref[1]])
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, 200, 40000, _medium)
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([owner.reference_offset(0, st) for st in strand_of], np.uint64)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    opt = ra.MapOpt(dtw_min_score=5.0)
    owner.upload_events(events)
    b = ra.Batch(owner, opt, cb)
    b.run()
    want = b.fetch(with_job_costs=True)
    # second context: shares the arena, uploads the events in two rounds (a prefix, then every read's last 30 events)
    eng = ra.Engine(0)
    lib = eng.lib
    eng._check(lib.rawdtw_share_reference(eng._ctx, owner._ctx))
    assert eng.reference_offset(0, 1) == owner.reference_offset(0, 1)
    starts = np.unique(read_base.astype(np.int64))
    ends = np.concatenate([starts[1:], [len(events)]])
    new_len = np.minimum(ends - starts, 30)
    old = events.copy()
    for s, e, nl in zip(starts, ends, new_len):
        old[e - nl:e] = np.float32(7777.0)  # not there yet
    eng._check(lib.rawdtw_events_reserve(eng._ctx, len(events)))
    eng._check(lib.rawdtw_upload_events(eng._ctx, old.ctypes.data_as(C.c_void_p), len(old)))
    seg_src = np.concatenate([[0], np.cumsum(new_len)]).astype(np.uint64)
    seg_dst = (ends - new_len).astype(np.uint32)
    new_events = np.concatenate([events[e - nl:e] for e, nl in zip(ends, new_len)]).astype(np.float32)
    eng._check(lib.rawdtw_events_append(eng._ctx, new_events.ctypes.data_as(C.c_void_p), len(new_events), len(new_len),
                                        seg_src.ctypes.data_as(C.c_void_p), seg_dst.ctypes.data_as(C.c_void_p)))
    # ... and takes the three big arrays from device memory
    eng.set_option("resident_arrays", 1)
    t_anchors = torch.from_numpy(anchors.view(np.uint8).copy()).cuda()
    t_ref_base = torch.from_numpy(ref_base.view(np.uint8).copy()).cuda()
    t_read_base = torch.from_numpy(read_base.view(np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    copt = opt.c_struct()
    h = C.c_void_p()
    co, ao = np.ascontiguousarray(chain_off, np.uint64), np.ascontiguousarray(anchor_off, np.uint64)
    eng._check(lib.rawdtw_batch_create(eng._ctx, C.byref(copt), cb.n_reads, co.ctypes.data_as(C.c_void_p), ao.ctypes.data_as(C.c_void_p),
                                       C.c_void_p(t_anchors.data_ptr()), C.c_void_p(t_ref_base.data_ptr()),
                                       C.c_void_p(t_read_base.data_ptr()), C.byref(h)))
    eng._check(lib.rawdtw_batch_run(eng._ctx, h))
    score = np.zeros(cb.n_chains, np.float32)
    keep = np.zeros(cb.n_chains, np.uint8)
    jc = np.zeros(len(want[2]), np.float32)
    eng._check(lib.rawdtw_batch_fetch(eng._ctx, h, score.ctypes.data_as(C.c_void_p), keep.ctypes.data_as(C.c_void_p), jc.ctypes.data_as(C.c_void_p)))
    lib.rawdtw_batch_destroy(h)
    assert np.array_equal(score.view(np.uint32), want[0].view(np.uint32)) and np.array_equal(keep, want[1])
    assert np.array_equal(jc.view(np.uint32), want[2].view(np.uint32))
    # out-of-range segment: refused
    bad_dst = seg_dst.copy(); bad_dst[0] = len(events)
    assert lib.rawdtw_events_append(eng._ctx, new_events.ctypes.data_as(C.c_void_p), len(new_events), len(new_len),
                                    seg_src.ctypes.data_as(C.c_void_p), bad_dst.ctypes.data_as(C.c_void_p)) == 4
    # pinned host memory round trip
    p = C.c_void_p()
    assert lib.rawdtw_host_alloc(4096, C.byref(p)) == 0 and p.value
    assert lib.rawdtw_host_free(p) == 0


@pytest.mark.parametrize("shapes,n_reads,parts_range", [(_tiny, 400, (1, 120)), (_medium, 150, (1, 60)), (_wide, 120, (1, 40)), (_tiny, 2500, (0, 4)),
                                                        (_tiny, 6, (4000, 5200))])
def test_compact_hand_over_equals_plain(oracle, shapes, n_reads, parts_range):
    """rawdtw_batch_submit_compact: the anchor lists cross as 2-byte steps and are decoded on the device, unit by unit of
    8192 entries, inside the scan (chains over several units, units of hundreds of tiny chains, escapes for steps of 255 or
    more).  Scores, keeps and every part's cost must equal the plain hand-over's, which the cases above pin to the oracle;
    the first case is checked against the oracle directly as well."""
    rng = np.random.default_rng(n_reads + parts_range[1])
    ref = [rng.normal(size=90000).astype(np.float32), rng.normal(size=90000).astype(np.float32)]
    eng = ra.Engine(0)
    eng.upload_reference([ref[0]], [ref[1]])
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, n_reads, 90000, shapes, parts_range)
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
    eng.upload_events(events)
    opt = ra.MapOpt(dtw_min_score=5.0)
    plain = ra.Batch(eng, opt, cb)
    plain.run()
    want = plain.fetch(with_job_costs=True)
    plain.close()
    b = ra.Batch(eng, opt, cb, compact=True)
    assert b.verify_plan() is True
    got = b.fetch(with_job_costs=True)
    for x, y in zip(got, want):
        assert np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8))
    if shapes is _tiny and n_reads == 400:
        _oracle_check(oracle, cb, {1: ref[0], 0: ref[1]}, strand_of, got[0], got[1], got[2], opt)
    b.close()


CARRY_DTYPE = np.dtype([("prev_src", "<u8"), ("parts", "<u4"), ("flags", "<u4"), ("start_t", "<u4"), ("start_q", "<u4")])  # rawdtw_carry_t


def _two_rounds(rng, eng, n_reads=300):
    """Round 2 = round 1's chains: 0 grown at the end (the usual case), 1 unchanged, 2 an interior anchor moved, 3 a chain round 1
    did not have, 4 cut back at the end (its last part was not the last one then).  Returns (cb1, cb2, prev_read, expected
    parts taken over per chain of round 2, events)."""
    events, chain_off, anchor_off, anchors, slot, read_base = _chains(rng, n_reads, 90000, _medium, (3, 90))
    strand_of = [1 if s == 0 else 0 for s in slot]
    ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
    nc = len(anchor_off) - 1
    kinds = rng.integers(0, 5, nc)
    a1, off1, a2, off2, expect, sel = [], [0], [], [0], np.zeros(nc, np.int64), []
    for c in range(nc):
        a = anchors[int(anchor_off[c]):int(anchor_off[c + 1])].copy()
        parts = len(a) - 1
        r2 = a
        if kinds[c] == 3:
            r1 = None
        elif kinds[c] == 0 and parts >= 2:
            cut = int(rng.integers(1, parts))        # round 1 lacks the last `cut` parts (end-first: the list's first entries)
            r1 = a[cut:]
            expect[c] = parts - cut
        elif kinds[c] == 2 and parts >= 3:
            k = int(rng.integers(1, len(a) - 1))     # an interior anchor that differs: the parts from the chain's start up to it stay
            r1 = a.copy()
            r1[k]["query_position"] -= 0 if r1[k]["query_position"] == r1[k + 1]["query_position"] else 1
            moved = r1[k]["query_position"] != a[k]["query_position"]
            same = len(a) - 1 - k if moved else len(a)   # anchors of the common tail
            expect[c] = same - 1 if same >= 2 else 0
        elif kinds[c] == 4 and parts >= 3:
            r1 = a
            r2 = a[1:]                                # round 2 lacks round 1's last part: its own last part is scored again
            expect[c] = len(r2) - 2
        else:
            r1 = a
            expect[c] = parts
        a2.append(r2); off2.append(off2[-1] + len(r2))
        if r1 is not None:
            sel.append(c); a1.append(r1); off1.append(off1[-1] + len(r1))
    # round 1 keeps the reads' structure: read r holds the chains of round 2's read r that existed then
    keep1 = np.zeros(nc, bool); keep1[sel] = True
    ch_read = np.repeat(np.arange(n_reads), np.diff(chain_off.astype(np.int64)))
    chain_off1 = np.concatenate([[0], np.cumsum(np.bincount(ch_read[keep1], minlength=n_reads))]).astype(np.uint64)
    cb1 = CandidateBatch(events, chain_off1, np.array(off1, np.uint64), np.concatenate(a1), ref_base[sel], read_base[sel])
    cb2 = CandidateBatch(events, chain_off, np.array(off2, np.uint64), np.concatenate(a2), ref_base, read_base)
    return cb1, cb2, np.arange(n_reads, dtype=np.uint64), expect


def _match(lib, cb2, cb1, prev_read):
    import ctypes as C

    vp = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
    a2 = [np.ascontiguousarray(cb2.chain_off, np.uint64), np.ascontiguousarray(cb2.anchor_off, np.uint64), np.ascontiguousarray(cb2.anchors),
          np.ascontiguousarray(cb2.ref_base, np.uint64), np.ascontiguousarray(cb2.read_base, np.uint32)]
    a1 = [np.ascontiguousarray(cb1.chain_off, np.uint64), np.ascontiguousarray(cb1.anchor_off, np.uint64), np.ascontiguousarray(cb1.anchors),
          np.ascontiguousarray(cb1.ref_base, np.uint64), np.ascontiguousarray(cb1.read_base, np.uint32)]
    carry = np.zeros(cb2.n_chains, CARRY_DTYPE)
    new_off = np.zeros(cb2.n_chains + 1, np.uint64)
    new_anchors = np.zeros(len(cb2.anchors) + 1, ra.ANCHOR_DTYPE)
    st = lib.rawdtw_round_match_chains(cb2.n_reads, vp(a2[0]), vp(a2[1]), vp(a2[2]), vp(a2[3]), vp(a2[4]), vp(prev_read), vp(a1[0]), vp(a1[1]), vp(a1[2]),
                                       vp(a1[3]), vp(a1[4]), vp(carry), vp(new_off), vp(new_anchors))
    assert st == 0
    return a2, carry, new_off, new_anchors


def test_round_carry_takes_over_unchanged_parts(oracle):
    """rawdtw_batch_submit_carry: every leading part (from the chain's start) whose anchors were there the round before is taken
    over -- the host's matcher says how many (checked against an independent count), only the new anchors are handed over, the
    device assembles the list and copies the costs; a chain's former last part loses its last cell's distance when it is no
    longer last; a part that is the last now and was not then is scored again.  All scores, keeps and part costs equal a
    from-scratch batch of round 2 (which the cases above pin to the oracle)."""
    import ctypes as C

    rng = np.random.default_rng(99)
    ref = [rng.normal(size=90000).astype(np.float32), rng.normal(size=90000).astype(np.float32)]
    eng = ra.Engine(0)
    eng.upload_reference([ref[0]], [ref[1]])
    lib = eng.lib
    cb1, cb2, prev_read, expect = _two_rounds(rng, eng)
    nc = cb2.n_chains
    eng.upload_events(cb2.events)
    opt = ra.MapOpt(dtw_min_score=5.0)
    copt = opt.c_struct()
    vp = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
    b1 = ra.Batch(eng, opt, cb1)
    assert lib.rawdtw_batch_can_carry(eng._ctx, b1._h, C.byref(copt)) == 0   # never run: nothing to take over
    h = C.c_void_p()
    arr, carry, new_off, new_anchors = _match(lib, cb2, cb1, prev_read)
    assert np.array_equal(carry["parts"].astype(np.int64), expect)
    assert int(new_off[-1]) == len(cb2.anchors) - int(expect.sum())   # (the new entries and, where a stretch is taken over, the junction)
    st = lib.rawdtw_batch_submit_carry(eng._ctx, C.byref(copt), cb2.n_reads, vp(arr[0]), vp(arr[1]), vp(arr[2]), vp(new_off), vp(new_anchors), vp(arr[3]),
                                       vp(arr[4]), b1._h, vp(carry), C.byref(h))
    assert st == 5 and not h.value                                            # RAWDTW_ERR_UNSUPPORTED, nothing enqueued
    b1.run()
    b1.fetch()
    other = ra.MapOpt(dtw_min_score=5.0, dtw_band_radius_frac=0.2).c_struct()
    assert lib.rawdtw_batch_can_carry(eng._ctx, b1._h, C.byref(other)) == 0   # another radius: other costs
    assert lib.rawdtw_batch_submit_carry(eng._ctx, C.byref(other), cb2.n_reads, vp(arr[0]), vp(arr[1]), vp(arr[2]), vp(new_off), vp(new_anchors), vp(arr[3]),
                                         vp(arr[4]), b1._h, vp(carry), C.byref(h)) == 5
    assert lib.rawdtw_batch_can_carry(eng._ctx, b1._h, C.byref(copt)) == 1
    plain = ra.Batch(eng, opt, cb2)
    plain.run()
    want = plain.fetch(with_job_costs=True)
    plain.close()
    eng._check(lib.rawdtw_batch_submit_carry(eng._ctx, C.byref(copt), cb2.n_reads, vp(arr[0]), vp(arr[1]), vp(arr[2]), vp(new_off), vp(new_anchors), vp(arr[3]),
                                             vp(arr[4]), b1._h, vp(carry), C.byref(h)))
    score, keep = np.zeros(nc, np.float32), np.zeros(nc, np.uint8)
    jc = np.zeros(len(want[2]), np.float32)
    eng._check(lib.rawdtw_batch_fetch(eng._ctx, h, vp(score), vp(keep), vp(jc)))
    b1.close()
    sc, ru = C.c_uint64(), C.c_uint64()
    eng._check(lib.rawdtw_batch_round_stats(eng._ctx, h, C.byref(sc), C.byref(ru)))
    assert np.array_equal(jc.view(np.uint32), want[2].view(np.uint32))
    assert np.array_equal(score.view(np.uint32), want[0].view(np.uint32)) and np.array_equal(keep, want[1])
    assert ru.value == int(expect.sum()) and sc.value + ru.value == len(jc) and ru.value > 1000
    # a third round on top of the carried one (its costs were copied, not computed): round 2 again, everything taken over
    arr3, carry3, new_off3, new_anchors3 = _match(lib, cb2, cb2, prev_read)
    assert int(carry3["parts"].sum()) == len(jc) and int(new_off3[-1]) == cb2.n_chains   # (nothing new: a junction a chain)
    h3 = C.c_void_p()
    eng._check(lib.rawdtw_batch_submit_carry(eng._ctx, C.byref(copt), cb2.n_reads, vp(arr3[0]), vp(arr3[1]), vp(arr3[2]), vp(new_off3), vp(new_anchors3),
                                             vp(arr3[3]), vp(arr3[4]), h, vp(carry3), C.byref(h3)))
    score3, keep3, jc3 = np.zeros(nc, np.float32), np.zeros(nc, np.uint8), np.zeros(len(jc), np.float32)
    eng._check(lib.rawdtw_batch_fetch(eng._ctx, h3, vp(score3), vp(keep3), vp(jc3)))
    eng._check(lib.rawdtw_batch_round_stats(eng._ctx, h3, C.byref(sc), C.byref(ru)))
    assert ru.value == len(jc) and sc.value == 0
    assert np.array_equal(jc3.view(np.uint32), want[2].view(np.uint32)) and np.array_equal(score3.view(np.uint32), want[0].view(np.uint32))
    lib.rawdtw_batch_destroy(h3)
    # an invented record (one part more than there is room for): the counts do not add up on the device -- the batch is not
    # scored from it: it is redone from the full lists through the job-list path (never a wrong cost)
    bad = carry.copy()
    c_bad = int(np.argmax(bad["parts"] > 0))
    bad["parts"][c_bad] += 1
    hb = C.c_void_p()
    st = lib.rawdtw_batch_submit_carry(eng._ctx, C.byref(copt), cb2.n_reads, vp(arr[0]), vp(arr[1]), vp(arr[2]), vp(new_off), vp(new_anchors), vp(arr[3]),
                                       vp(arr[4]), h, vp(bad), C.byref(hb))
    if st == 0:
        st = lib.rawdtw_batch_fetch(eng._ctx, hb, vp(score3), vp(keep3), None)
        assert st != 0 or np.array_equal(score3.view(np.uint32), want[0].view(np.uint32))
        lib.rawdtw_batch_destroy(hb)
    lib.rawdtw_batch_destroy(h)
    eng.close()
