// rawdtw_planner.cpp -- the host planner of the job-list path (any list of DTW jobs: global mode, traceback, batches the
// stream path declines, the tests), its launch sequences, and the rawdtw_plan_* / rawdtw_score_batch entry points
// (include/rawdtw.h).  Host code only; kernels live in rawdtw_kernels.hip.
#include "rawdtw_capi.h"

using namespace rawdtw;
using namespace rawdtw::capi;

namespace rawdtw {
namespace capi {

// ---- planner ---------------------------------------------------------------------------------
// Everything up to the device upload is host work on plain arrays (plan_host), so that it can be
// checked without a GPU (rawdtw_plan_dry_run) and spread over threads: at the bench's 5 M jobs per
// mini-batch a one-thread planner costs a thousand times the kernels it feeds.

struct PlanCfg {
    uint64_t n_ev = 0, n_ref = 0;
    int lane_max_radius = kMaxLaneRadius;
    uint32_t lane_max_n = kLaneMaxN, lane_hi_max_n = 96;
    bool lane_hi = false, grp16 = true, grp8 = true, full_wg = true;
    int micro_max_n = 8;
    uint32_t tile_lds_floats = kTileLdsFloats, tile_max_jobs = kTileMaxJobs, tile_max_spans = kTileMaxSpans;
    int threads = 0; // 0: pick from the job count and the machine
    // Optional: tile-eligible jobs that are rare inside a tile (long, or of a radius few neighbours share) leave the job
    // order and are tiled by shape instead: longer side >= sort_n, radius 1 with longer side >= sort_r1_n, radius 3
    // (0 = never).  Measured on the bench workload with sort_n = 17: the tile kernel's VALU work drops 40 % (full waves of
    // one shape), but every such job then fetches its own cache lines (+130 MB of scattered reads per batch); alone the
    // kernel breaks even, with several batches in flight throughput falls 5-15 %.  Off by default.
    uint32_t sort_n = 0, sort_r1_n = 0, sort_r3 = 0, sorted_tile_jobs = 64;
};

// tile records built on the host (uploaded by build_plan)
struct HostTiles {
    std::vector<TileDesc> tiles;
    std::vector<TileSpan> spans;
    RawVec<TileJob> tjobs;
    std::vector<unsigned long long> masks;
};

PlanCfg cfg_of(const rawdtw_ctx *ctx)
{
    PlanCfg c;
    c.n_ev = ctx->n_ev; c.n_ref = ctx->n_ref;
    c.lane_max_radius = ctx->lane_max_radius; c.lane_max_n = ctx->lane_max_n; c.lane_hi_max_n = ctx->lane_hi_max_n;
    c.lane_hi = ctx->lane_hi; c.grp16 = ctx->grp16; c.grp8 = ctx->grp8; c.full_wg = ctx->full_wg; c.micro_max_n = ctx->micro_max_n;
    c.tile_lds_floats = ctx->tile_lds_floats; c.tile_max_jobs = ctx->tile_max_jobs; c.threads = ctx->plan_threads;
    c.tile_max_spans = ctx->tile_max_spans;
    c.sort_n = ctx->sort_n; c.sort_r1_n = ctx->sort_r1_n; c.sort_r3 = ctx->sort_r3; c.sorted_tile_jobs = ctx->sorted_tile_jobs;
    return c;
}


// band masks of every micro shape: index ((N-1)*8 + (M-1)) * (kMaxLaneRadius+1) + R, N >= M
const std::vector<unsigned long long> &micro_masks()
{
    static const std::vector<unsigned long long> table = [] {
        std::vector<unsigned long long> t(8 * 8 * (kMaxLaneRadius + 1), 0ull);
        for (uint32_t N = 1; N <= 8; N++)
            for (uint32_t M = 1; M <= N; M++)
                for (int R = 0; R <= kMaxLaneRadius; R++)
                    t[((N - 1) * 8 + (M - 1)) * (kMaxLaneRadius + 1) + R] = band_mask8(N, M, R);
        return t;
    }();
    return table;
}

// Tiles for plan positions [p0, p1) of the tile class `hi` (consecutive jobs in job order).  Appends
// to tiles/spans (span_first relative to `spans`), fills tjobs[p] in place; returns the largest LDS image.
uint32_t build_tiles(const PlanCfg &cfg, bool hi, bool by_shape, uint32_t max_jobs, uint32_t max_spans, const RawVec<DevJob> &h_jobs, uint64_t p0, uint64_t p1,
                     std::vector<TileDesc> &tiles, std::vector<TileSpan> &spans, TileJob *tjobs)
{
    const uint32_t lds_budget = hi ? kTileHiLdsFloats : cfg.tile_lds_floats;
    uint32_t tile_lds_max = 0;
    // profiling aid (scripts/valu_by_class.py), never set in production: RAWDTW_DEBUG_SKIP="lo,hi,r" leaves the jobs with
    // lo <= longer side <= hi (and radius r, -1 = any) staged but unscored (their cost reads 0), to attribute kernel time
    int dbg_lo = 0, dbg_hi = -1, dbg_r = -1;
    const char *dbg_env = getenv("RAWDTW_DEBUG_SKIP");
    const bool dbg_skip = dbg_env && sscanf(dbg_env, "%d,%d,%d", &dbg_lo, &dbg_hi, &dbg_r) == 3;
    struct Sp { uint64_t start, end; bool is_ref; uint32_t lds; }; // [start,end) in floats, start 4-aligned
    std::vector<Sp> cur;
    struct Pend { uint32_t spA, spB; uint64_t a0, b0; };
    std::vector<Pend> pend;
    std::vector<uint32_t> ia, ib;
    std::vector<TileJob> tmp;
    uint64_t t_first = p0;
    uint32_t lds_used = 0;
    auto span_cost = [](const Sp &s) { return (uint32_t)(((s.end - s.start) + 3) & ~3ull); };
    auto close_tile = [&](uint64_t t_end) {
        if (t_end == t_first) return;
        uint32_t off = 0;
        const uint32_t span_first = (uint32_t)spans.size();
        for (Sp &s : cur) {
            s.lds = off;
            const uint32_t len4 = span_cost(s);
            spans.push_back(TileSpan{s.start, off, (len4 / 4) | (s.is_ref ? 0x80000000u : 0u)});
            off += len4;
        }
        tile_lds_max = std::max(tile_lds_max, off);
        for (uint64_t p = t_first; p < t_end; p++) {
            const Pend &pe = pend[p - t_first];
            TileJob &tj = tjobs[p];
            tj.offA = (uint16_t)(cur[pe.spA].lds + (pe.a0 - cur[pe.spA].start));
            tj.offB = (uint16_t)(cur[pe.spB].lds + (pe.b0 - cur[pe.spB].start));
        }
        // order the tile's records by (dispatch kind, longer side desc, shorter side desc, job): waves get one
        // shape.  Stable LSD radix sort over the three bytes (the records start in job order).
        const uint32_t cnt = (uint32_t)(t_end - t_first);
        tmp.assign(tjobs + t_first, tjobs + t_end);
        ia.resize(cnt); ib.resize(cnt);
        for (uint32_t q = 0; q < cnt; q++) ia[q] = q;
        for (int pass = 0; pass < 3; pass++) {
            uint32_t count[257] = {0};
            auto digit = [&](uint32_t q) -> uint32_t {
                const TileJob &x = tmp[q];
                return pass == 0 ? 255u - x.M : pass == 1 ? 255u - x.N : x.R;
            };
            for (uint32_t q = 0; q < cnt; q++) count[digit(ia[q]) + 1]++;
            for (int b = 0; b < 256; b++) count[b + 1] += count[b];
            for (uint32_t q = 0; q < cnt; q++) ib[count[digit(ia[q])]++] = ia[q];
            ia.swap(ib);
        }
        for (uint32_t q = 0; q < cnt; q++) tjobs[t_first + q] = tmp[ia[q]];
        tiles.push_back(TileDesc{(uint32_t)t_first, (uint32_t)(t_end - t_first), span_first,
                                 (uint32_t)cur.size() | (by_shape && !hi ? 0x80000000u : 0u)});
        cur.clear(); pend.clear(); lds_used = 0; t_first = t_end;
    };
    // find the span that holds (or can be grown to hold) window [w0, w0+len) of the given arena; -1: a new one
    auto place = [&](uint64_t w0, uint32_t len, bool is_ref, uint32_t &extra) -> int {
        extra = 0;
        for (int q = (int)cur.size() - 1; q >= 0 && q >= (int)cur.size() - 8; q--) {
            Sp &s = cur[q];
            // a window that starts a little past the span still extends it (a part that left for another class leaves
            // a hole in its chain's windows; staging the hole is cheaper than another span)
            if (s.is_ref != is_ref || w0 < s.start || w0 > s.end + kSpanGapFloats) continue;
            if (w0 + len <= s.end) return q; // already covered
            const uint32_t before = span_cost(s);
            Sp grown = s; grown.end = w0 + len;
            extra = span_cost(grown) - before;
            return q; // caller extends after the budget check
        }
        extra = (uint32_t)((((w0 & 3ull) + len) + 3) & ~3ull);
        return -1;
    };
    for (uint64_t p = p0; p < p1; p++) {
        const DevJob &d = h_jobs[p];
        const bool swap = d.n < d.m; // dtw.cpp:284-292: A is the longer sequence
        const uint64_t a0 = swap ? d.ref_off : d.read_off, b0 = swap ? d.read_off : d.ref_off;
        const uint32_t NA = swap ? d.m : d.n, NB = swap ? d.n : d.m;
        const bool a_ref = swap, b_ref = !swap;
        for (int attempt = 0; attempt < 2; attempt++) {
            uint32_t ea = 0, eb = 0;
            int qa = place(a0, NA, a_ref, ea);
            int qb = place(b0, NB, b_ref, eb); // a fresh span for A cannot serve B: other arena
            const uint32_t new_spans = (qa < 0) + (qb < 0);
            if (attempt == 0 && (lds_used + ea + eb > lds_budget || cur.size() + new_spans > max_spans ||
                                 p - t_first >= max_jobs)) {
                close_tile(p);
                continue; // retry in the fresh tile
            }
            if (qa < 0) { cur.push_back(Sp{a0 & ~3ull, a0 + NA, a_ref, 0}); qa = (int)cur.size() - 1; }
            else cur[qa].end = std::max(cur[qa].end, a0 + NA);
            if (qb < 0) { cur.push_back(Sp{b0 & ~3ull, b0 + NB, b_ref, 0}); qb = (int)cur.size() - 1; }
            else cur[qb].end = std::max(cur[qb].end, b0 + NB);
            lds_used += ea + eb;
            pend.push_back(Pend{(uint32_t)qa, (uint32_t)qb, a0, b0});
            TileJob &tj = tjobs[p];
            tj.N = (uint8_t)NA; tj.M = (uint8_t)NB; tj.flags = (uint8_t)d.flags;
            tj.aux = d.aux; tj.pad = 0; tj.offA = tj.offB = 0;
            if (!hi && NA <= (uint32_t)cfg.micro_max_n) { // micro path: band membership from a per-shape bitmask
                tj.pad = ((NA - 1) * 8 + (NB - 1)) * (kMaxLaneRadius + 1) + (uint32_t)d.R;
                tj.R = NA <= 4 ? 0 : 1;
            } else {
                tj.R = (uint8_t)(2 + d.R);
            }
            if (dbg_skip && (int)NA >= dbg_lo && (int)NA <= dbg_hi && (dbg_r < 0 || dbg_r == d.R)) tj.R = 255; // no kernel path: cost 0
            break;
        }
    }
    close_tile(p1);
    return tile_lds_max;
}

// Host half of plan creation.  traceback=true: every job must be a full-matrix job and gets a direction buffer.
// Sort key of the jobs outside the tile kernel: class in the top bits, then descending length so long jobs start first
//   banded tile (lane DP): class 0 (and 1 for the optional wide instance), kept in JOB order
//   banded 16-lane rows  : class 39
//   banded wave, register: class 40 (<= 4 chunks, merged) / 40 + log2(chunks)      (radius+1 <= 64*chunks, chunks <= 32)
//   banded wave, LDS     : class 48 + lds bucket
//   full                 : class 56 + log2(rpl), 60 = four waves per job
int plan_host(const PlanCfg &cfg, const rawdtw_job_t *jobs, uint64_t n_jobs, bool traceback, rawdtw_plan *pl,
              HostTiles &ht, std::string &err)
{
    pl->n_jobs = n_jobs;
    int T = cfg.threads;
    if (T <= 0) {
        const unsigned hc = std::thread::hardware_concurrency();
        T = (int)std::min<uint64_t>(std::min<unsigned>(hc ? hc : 1, 16), n_jobs / 32768 + 1);
    }
    T = std::max(1, std::min(T, 64));
    auto seg = [&](int t, uint64_t n) { return n * (uint64_t)t / (uint64_t)T; };
    static const bool timing = getenv("RAWDTW_PLAN_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = now();
        fprintf(stderr, "[plan] %-10s %8.2f ms (T=%d)\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count(), T);
        t_prev = t;
    };

    // ---- pass A: validate + classify -----------------------------------------------------------
    RawVec<uint8_t> cls;
    RawVec<int32_t> Rv;
    cls.resize(n_jobs);
    Rv.resize(n_jobs);
    struct PerThread {
        uint64_t n0 = 0, n1 = 0, n2 = 0, nother = 0, alg_bytes = 0;
        uint64_t bad = ~0ull; int bad_status = RAWDTW_OK; const char *bad_msg = nullptr;
        double work0 = 0, work1 = 0;
        char pad[64];
    };
    std::vector<PerThread> pt(T);
    parallel_for(T, [&](int t) {
        PerThread &P = pt[t];
        for (uint64_t k = seg(t, n_jobs); k < seg(t + 1, n_jobs); k++) {
            const rawdtw_job_t &j = jobs[k];
            auto bad = [&](int status, const char *msg) { if (P.bad == ~0ull) { P.bad = k; P.bad_status = status; P.bad_msg = msg; } };
            if (j.n == 0 || j.m == 0 || j.band_radius < RAWDTW_FULL || j.n >= 0x7fffffffu || j.m >= 0x7fffffffu) {
                bad(RAWDTW_ERR_INVALID, "zero length or negative band radius (dtw.cpp:274-277 asserts)");
                continue;
            }
            if ((uint64_t)j.read_off + j.n > cfg.n_ev || j.ref_off + j.m > cfg.n_ref) {
                bad(RAWDTW_ERR_RANGE, "window outside the uploaded arenas");
                continue;
            }
            P.alg_bytes += 4ull * ((uint64_t)j.n + j.m) + 4 + 32;
            const uint32_t N = std::max(j.n, j.m), NY = std::min(j.n, j.m);
            uint32_t c;
            int32_t R = -1;
            if (j.band_radius == RAWDTW_FULL) {
                const int rpl = full_rpl(NY);
                c = 56 + (rpl == 1 ? 0 : rpl == 2 ? 1 : rpl == 4 ? 2 : 3);
                // >= 3 strips: four waves per job.  The pipelined kernel's progress word packs (strip << 21) | columns
                // (k_full_wave): shapes beyond 2^21 columns or 2^11 strips stay on the one-wave variant.
                if (rpl == 8 && NY > 2 * 512u && cfg.full_wg && N < (1u << 21) && (NY + 511u) / 512u < (1u << 11)) c = 60;
            } else {
                if (traceback) {
                    bad(RAWDTW_ERR_UNSUPPORTED, "traceback of a banded job is not implemented (rmap.cpp:223-225 assert(false))");
                    continue;
                }
                R = slanted_radius(j.n, j.m, j.band_radius);
                if (R < 0 || R + 1 > kMaxWaveBandK) {
                    bad(RAWDTW_ERR_UNSUPPORTED, "band radius too large for the LDS-resident band kernel");
                    continue;
                }
                const uint32_t K = (uint32_t)R + 1;
                if (R <= cfg.lane_max_radius && N <= cfg.lane_max_n) {
                    c = 0;
                    if ((cfg.sort_n && N >= cfg.sort_n) || (cfg.sort_r1_n && R == 1 && N >= cfg.sort_r1_n) || (cfg.sort_r3 && R == 3))
                        c = 2; // rare inside a tile: tiled by shape, full waves of one shape
                }
                else if (R <= kMaxLaneRadiusHi && cfg.lane_hi && N <= cfg.lane_hi_max_n) c = 1; // the wide instance covers radii 0..8
                else if (K <= 8 && cfg.grp16 && cfg.grp8) c = 38; // eight jobs per wave (8-lane groups)
                else if (K <= 16 && cfg.grp16) c = 39; // four jobs per wave (16-lane rows)
                else if (K <= 64u * kMaxWregChunks) {
                    uint32_t chunks = 1, lg = 0;
                    while (64u * chunks < K) { chunks <<= 1; lg++; }
                    c = chunks <= 4 ? 40 : 40 + lg; // one merged launch for radius+1 <= 256 (param 0)
                } else {
                    c = 48 + (K <= 8192 ? 0 : 1); // LDS buckets: 3K floats
                }
            }
            cls[k] = (uint8_t)c;
            Rv[k] = R;
            if (c == 0) { P.n0++; P.work0 += (double)N * std::min<double>(2.0 * R + 1.0, NY); }
            else {
                if (c == 1) { P.n1++; P.work1 += (double)N * std::min<double>(2.0 * R + 1.0, NY); }
                if (c == 2) { P.n2++; P.work0 += (double)N * std::min<double>(2.0 * R + 1.0, NY); }
                P.nother++; // classes 1 and 2 are sorted with the rest (by shape); they only share the tile kernels
            }
        }
    });
    lap("classify");
    {   // first offending job, as the one-thread planner would report it
        uint64_t bad = ~0ull; int t_bad = -1;
        for (int t = 0; t < T; t++) if (pt[t].bad < bad) { bad = pt[t].bad; t_bad = t; }
        if (t_bad >= 0) {
            err = "job " + std::to_string(bad) + ": " + pt[t_bad].bad_msg;
            return pt[t_bad].bad_status;
        }
    }
    uint64_t n0 = 0, n1 = 0, n2 = 0, nother = 0, alg_bytes = 0;
    double work0 = 0, work1 = 0;
    std::vector<uint64_t> base0(T), baseo(T);
    for (int t = 0; t < T; t++) {
        base0[t] = n0; baseo[t] = nother;
        n0 += pt[t].n0; n1 += pt[t].n1; n2 += pt[t].n2; nother += pt[t].nother; alg_bytes += pt[t].alg_bytes;
        work0 += pt[t].work0; work1 += pt[t].work1;
    }
    pl->n_tile_jobs = n0 + n1 + n2; // plan order: [class 0, job order][class 2, by shape][class 1, by shape][the rest]

    // ---- pass B: plan positions.  The bulk tile class keeps job order (consecutive parts share their spans); the
    // rest is sorted by (class, shape).  Class 1 (wide-band tile instance) sorts first, by (radius, longer side,
    // shorter side): its jobs are rare and far apart, so nothing is shared anyway, and a wave of one radius and
    // similar lengths runs one pass of the lane DP instead of one per radius present ----
    struct Keyed { uint64_t key; uint32_t idx; };
    std::vector<Keyed> keyed(nother);
    pl->order.resize(n_jobs);
    pl->h_jobs.resize(n_jobs);
    auto put = [&](uint64_t p, uint64_t k) {
        const rawdtw_job_t &j = jobs[k];
        pl->order[p] = (uint32_t)k;
        DevJob &d = pl->h_jobs[p];
        d.ref_off = j.ref_off; d.read_off = j.read_off; d.n = j.n; d.m = j.m;
        d.R = Rv[k]; d.flags = j.exclude_last ? kFlagExcludeLast : 0u; d.aux = (uint32_t)k;
    };
    parallel_for(T, [&](int t) {
        uint64_t q0 = base0[t], qo = baseo[t];
        for (uint64_t k = seg(t, n_jobs); k < seg(t + 1, n_jobs); k++) {
            const uint32_t c = cls[k];
            if (c == 0) put(q0++, k);
            else {
                const rawdtw_job_t &j = jobs[k];
                const uint64_t N = std::max(j.n, j.m), NY = std::min(j.n, j.m), lim = (1ull << 28) - 1;
                uint64_t key = ((uint64_t)c << 56) | ((lim - std::min(N, lim)) << 28) | (lim - std::min(NY, lim));
                if (c == 1 || c == 2) // top byte: class 2 sorts before class 1
                    key = ((uint64_t)(c == 2 ? 1 : 2) << 56) | ((uint64_t)Rv[k] << 40) | ((255 - std::min<uint64_t>(N, 255)) << 20) | (255 - std::min<uint64_t>(NY, 255));
                keyed[qo++] = Keyed{key, (uint32_t)k};
            }
        }
    });
    lap("scatter");
    std::sort(keyed.begin(), keyed.end(), [](const Keyed &x, const Keyed &y) {
        return x.key != y.key ? x.key < y.key : x.idx < y.idx;
    });
    for (uint64_t q = 0; q < nother; q++) put(n0 + q, keyed[q].idx); // classes 2 and 1 first: positions [n0, n0 + n2 + n1)
    lap("sort-rest");

    // ---- launches: the two tile classes, then maximal runs of equal class; workspace of the full-matrix jobs ----
    const uint64_t n_dev = nother - n1 - n2; // jobs with a device record (all but the tile classes)
    const uint64_t n12 = n1 + n2;
    pl->h_aux.assign(n_dev, FullAux{0, 0}); // indexed like d_jobs: plan position - n_tile_jobs
    if (n0 + n2) pl->launches.push_back(Launch{kKindBandLane, 0, 0, n0 + n2});
    if (n1) pl->launches.push_back(Launch{kKindBandLaneHi, 0, n0 + n2, n1});
    uint64_t bnd = 0, dirb = 0;
    for (uint64_t q = 0; q < n_dev; q++) {
        const uint64_t p = pl->n_tile_jobs + q;
        const uint64_t c = keyed[n12 + q].key >> 56;
        const DevJob &j = pl->h_jobs[p];
        if (c >= 56) {
            const int rpl = c == 60 ? 8 : 1 << (c - 56);
            const uint64_t rows = c == 60 ? kFullWgWaves : 1; // boundary rows: a ring for the pipelined variant
            const uint32_t NX = std::max(j.n, j.m), NY = std::min(j.n, j.m);
            if (NY > 64u * rpl) { // multi-strip: needs a boundary row
                pl->h_aux[q].bnd_off = bnd;
                bnd += rows * (((uint64_t)NX + 63) & ~63ull);
            }
            if (traceback) {
                pl->h_aux[q].dir_off = dirb;
                dirb += (dir_bytes_for(j.n, j.m, rpl) + 255) & ~255ull;
            }
        }
        if (q == 0 || (keyed[n12 + q - 1].key >> 56) != c) {
            Launch L{};
            L.first = p; L.count = 0;
            if (c == 38) { L.kind = kKindBandWreg; L.param = -8; }
            else if (c == 39) { L.kind = kKindBandWreg; L.param = -16; }
            else if (c < 48) { L.kind = kKindBandWreg; L.param = c == 40 ? 0 : 1 << (c - 40); }
            else if (c < 56) { L.kind = kKindBandWave; L.param = 3 * kMaxWaveBandK; }
            else { L.kind = traceback ? kKindFullTb : kKindFullWave; L.param = c == 60 ? 8 + 256 : 1 << (c - 56); }
            pl->launches.push_back(L);
        }
        pl->launches.back().count++;
    }

    // ---- tiles: each thread tiles a contiguous run of the job-ordered positions (a tile never spans two runs) ----
    ht.tjobs.resize(pl->n_tile_jobs);
    if ((n0 || n2) && cfg.micro_max_n > 0) ht.masks = micro_masks();
    for (Launch &TL : pl->launches) {
        if (TL.kind != kKindBandLane && TL.kind != kKindBandLaneHi) continue;
        const bool hi = TL.kind == kKindBandLaneHi;
        const size_t tiles_before = ht.tiles.size();
        uint32_t tile_lds_max = 0;
        // the bulk launch has two runs: class 0 in job order, then class 2 by shape (small tiles of whole waves)
        // (a by-shape tile holds two spans per job: nothing is shared)
        struct Run { uint64_t first, count; uint32_t max_jobs, max_spans; bool by_shape; };
        std::vector<Run> runs;
        if (hi) runs.push_back(Run{TL.first, TL.count, kTileHiMaxJobs, 2 * kTileHiMaxJobs, true});
        else {
            if (n0) runs.push_back(Run{0, n0, cfg.tile_max_jobs, cfg.tile_max_spans, false});
            if (n2) runs.push_back(Run{n0, n2, cfg.sorted_tile_jobs, std::max(cfg.tile_max_spans, 2 * cfg.sorted_tile_jobs), true});
        }
        for (const Run &run : runs) {
            const int TT = (int)std::min<uint64_t>(T, run.count / 8192 + 1);
            std::vector<std::vector<TileDesc>> tl(TT);
            std::vector<std::vector<TileSpan>> sp(TT);
            std::vector<uint32_t> lmax(TT, 0);
            // thread boundaries at multiples of 64 jobs: a sorted run is cut into whole waves
            auto cut = [&](int t) { return t >= TT ? run.count : (run.count * (uint64_t)t / TT) & ~63ull; };
            parallel_for(TT, [&](int t) {
                lmax[t] = build_tiles(cfg, hi, run.by_shape, run.max_jobs, run.max_spans, pl->h_jobs, run.first + cut(t), run.first + cut(t + 1), tl[t], sp[t],
                                      ht.tjobs.data());
            });
            for (int t = 0; t < TT; t++) {
                const uint32_t span_base = (uint32_t)ht.spans.size();
                for (TileDesc d : tl[t]) { d.span_first += span_base; ht.tiles.push_back(d); }
                ht.spans.insert(ht.spans.end(), sp[t].begin(), sp[t].end());
                tile_lds_max = std::max(tile_lds_max, lmax[t]);
            }
        }
        if (hi) { pl->n_tiles_hi = ht.tiles.size() - tiles_before; pl->tile_hi_lds_floats = tile_lds_max; }
        else { pl->n_tiles = ht.tiles.size() - tiles_before; pl->tile_lds_floats = tile_lds_max; }
        TL.param = (int32_t)tile_lds_max;
    }
    lap("tiles");
    // a banded-wave launch only needs LDS for its own largest K (jobs are sorted by N, not K)
    for (Launch &L : pl->launches)
        if (L.kind == kKindBandWave) {
            int32_t kmax = 0;
            for (uint64_t p = L.first; p < L.first + L.count; p++) kmax = std::max(kmax, pl->h_jobs[p].R + 1);
            L.param = 3 * kmax;
        }
    pl->bnd_floats = bnd;
    pl->dir_bytes = dirb;
    {   // rough work per launch: sum over jobs of (longer side) x (band width or shorter side)
        std::vector<double> work(pl->launches.size(), 0.0);
        for (size_t i = 0; i < pl->launches.size(); i++) {
            const Launch &L = pl->launches[i];
            if (L.kind == kKindBandLane) { work[i] = work0; continue; }
            if (L.kind == kKindBandLaneHi) { work[i] = work1; continue; }
            for (uint64_t p = L.first; p < L.first + L.count; p++) {
                const DevJob &d = pl->h_jobs[p];
                const double N = std::max(d.n, d.m), M = std::min(d.n, d.m);
                const double w = d.R < 0 ? M : std::min<double>(2.0 * d.R + 1.0, M);
                work[i] += N * std::max(w, 64.0); // wave-per-job kernels spend a whole wave on one job
            }
        }
        pl->run_order.resize(pl->launches.size());
        for (uint32_t i = 0; i < pl->run_order.size(); i++) pl->run_order[i] = i;
        std::stable_sort(pl->run_order.begin(), pl->run_order.end(),
                         [&](uint32_t x, uint32_t y) { return work[x] > work[y]; });
    }

    rawdtw_plan_info_t &I = pl->info;
    I.n_jobs = n_jobs;
    I.algorithmic_bytes = alg_bytes;
    I.n_launches = (uint32_t)pl->launches.size();
    for (const Launch &L : pl->launches) {
        if (L.kind == kKindBandLane || L.kind == kKindBandLaneHi) I.n_lane_jobs += L.count;
        else if (L.kind == kKindBandWave || L.kind == kKindBandWreg) I.n_wave_band_jobs += L.count;
        else I.n_full_jobs += L.count;
    }
    I.workspace_bytes = bnd * 4 + dirb + n_dev * (sizeof(DevJob) + sizeof(FullAux)) + n_jobs * 4 +
                        ht.tiles.size() * sizeof(TileDesc) + ht.spans.size() * sizeof(TileSpan) +
                        ht.tjobs.size() * sizeof(TileJob) + ht.masks.size() * 8;
    pl->plan_threads_used = T;
    lap("finish");
    return RAWDTW_OK;
}

// The tile records against the jobs they were built from: every window staged inside its tile's LDS image at the right
// place, every record's shape / radius / mask right, records in dispatch order.  tseen[k] = job k has a tile record.
std::string verify_tile_arrays(const PlanCfg &cfg, const rawdtw_job_t *jobs, uint64_t n_jobs, const rawdtw_plan *pl,
                               const TileDesc *tiles, size_t n_tiles_all, const TileSpan *spans, size_t n_spans_all,
                               const TileJob *tjobs, size_t n_tjobs, const std::vector<unsigned long long> &masks,
                               std::vector<uint8_t> &tseen)
{
    auto S = [](uint64_t v) { return std::to_string(v); };
    if (n_tiles_all != pl->n_tiles + pl->n_tiles_hi || n_tjobs != pl->n_tile_jobs) return "tile counts";
    tseen.assign(n_jobs, 0);
    uint64_t next_job = 0;
    for (size_t ti = 0; ti < n_tiles_all; ti++) {
        TileDesc t = tiles[ti];
        t.n_spans &= 0x7fffffffu;
        const bool hi = ti >= pl->n_tiles;
        const uint32_t budget = hi ? kTileHiLdsFloats : cfg.tile_lds_floats;
        if (t.job_first != next_job || t.n_jobs == 0) return "tile " + S(ti) + ": jobs not consecutive";
        next_job += t.n_jobs;
        if (t.n_jobs > (hi ? kTileHiMaxJobs : std::max(cfg.tile_max_jobs, cfg.sorted_tile_jobs)) || t.n_spans == 0 ||
            t.n_spans > (hi ? 2 * kTileHiMaxJobs : std::max(cfg.tile_max_spans, 2 * cfg.sorted_tile_jobs)))
            return "tile " + S(ti) + ": too many jobs or spans";
        if ((uint64_t)t.span_first + t.n_spans > n_spans_all) return "tile " + S(ti) + ": spans out of range";
        uint32_t off = 0;
        for (uint32_t s = 0; s < t.n_spans; s++) {
            const TileSpan &sp = spans[t.span_first + s];
            const uint32_t len = 4 * (sp.chunks_arena & 0x7fffffffu);
            const bool is_ref = sp.chunks_arena >> 31;
            if (sp.lds_off != off || (sp.src & 3) || len == 0) return "tile " + S(ti) + ": span layout";
            // the copy reads whole 16-byte chunks: the arenas are allocated with that slack (see upload_*), the
            // span itself must start inside the arena
            if (sp.src >= (is_ref ? cfg.n_ref : cfg.n_ev)) return "tile " + S(ti) + ": span outside its arena";
            off += len;
        }
        if (off > budget || off > (hi ? pl->tile_hi_lds_floats : pl->tile_lds_floats)) return "tile " + S(ti) + ": LDS image over budget";
        for (uint32_t q = 0; q < t.n_jobs; q++) {
            const TileJob &tj = tjobs[t.job_first + q];
            const uint32_t k = tj.aux;
            if (k >= n_jobs || tseen[k]) return "tile job " + S(k) + " duplicated";
            tseen[k] = 1;
            const rawdtw_job_t &j = jobs[k];
            const bool swap = j.n < j.m;
            const uint64_t a0 = swap ? j.ref_off : j.read_off, b0 = swap ? j.read_off : j.ref_off;
            const uint32_t NA = swap ? j.m : j.n, NB = swap ? j.n : j.m;
            if (tj.N != NA || tj.M != NB || ((tj.flags & kFlagExcludeLast) != 0) != (j.exclude_last != 0))
                return "tile job " + S(k) + ": shape or flags";
            const int R = slanted_radius(j.n, j.m, j.band_radius);
            if (tj.R >= 2) { if ((int)tj.R - 2 != R) return "tile job " + S(k) + ": radius"; }
            else {
                if (NA > (tj.R == 0 ? 4u : 8u) || tj.pad >= masks.size() || masks[tj.pad] != band_mask8(NA, NB, R))
                    return "tile job " + S(k) + ": micro mask";
            }
            // both windows must lie inside one staged span of the right arena, at the right place
            for (int w = 0; w < 2; w++) {
                const uint32_t o = w ? tj.offB : tj.offA, len = w ? NB : NA;
                const uint64_t g0 = w ? b0 : a0;
                const bool want_ref = w ? !swap : swap;
                bool ok = false;
                for (uint32_t s = 0; s < t.n_spans && !ok; s++) {
                    const TileSpan &sp = spans[t.span_first + s];
                    const uint32_t slen = 4 * (sp.chunks_arena & 0x7fffffffu);
                    if ((bool)(sp.chunks_arena >> 31) != want_ref) continue;
                    if (o >= sp.lds_off && o + len <= sp.lds_off + slen && sp.src + (o - sp.lds_off) == g0) ok = true;
                }
                if (!ok) return "tile job " + S(k) + ": window " + (w ? "B" : "A") + " not staged";
            }
        }
        for (uint32_t q = 1; q < t.n_jobs; q++) { // dispatch order inside the tile
            const TileJob &x = tjobs[t.job_first + q - 1], &y = tjobs[t.job_first + q];
            if (x.R > y.R || (x.R == y.R && x.N < y.N)) return "tile " + S(ti) + ": records not sorted";
        }
    }
    if (next_job != pl->n_tile_jobs) return "tiles cover " + S(next_job) + " of " + S(pl->n_tile_jobs) + " tile jobs";
    return "";
}

// Self-check of a host plan against the jobs it was built from (rawdtw_plan_dry_run; tests).  Returns an
// empty string when every invariant the kernels rely on holds.
std::string verify_tile_arrays(const PlanCfg &cfg, const rawdtw_job_t *jobs, uint64_t n_jobs, const rawdtw_plan *pl, const TileDesc *tiles, size_t n_tiles,
                               const TileSpan *spans, size_t n_spans, const TileJob *tjobs, size_t n_tjobs, const std::vector<unsigned long long> &masks,
                               std::vector<uint8_t> &seen);
std::string verify_uploaded_tiles(const rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, const rawdtw_plan *pl, const TileDesc *tiles,
                                  size_t n_tiles, const TileSpan *spans, size_t n_spans, const TileJob *tjobs, size_t n_tjobs, std::vector<uint8_t> &seen)
{
    return verify_tile_arrays(cfg_of(ctx), jobs, n_jobs, pl, tiles, n_tiles, spans, n_spans, tjobs, n_tjobs, micro_masks(), seen);
}

std::string verify_host_plan(const PlanCfg &cfg, const rawdtw_job_t *jobs, uint64_t n_jobs, const rawdtw_plan *pl,
                             const HostTiles &ht)
{
    auto S = [](uint64_t v) { return std::to_string(v); };
    if (pl->order.size() != n_jobs || pl->h_jobs.size() != n_jobs) return "order/h_jobs size";
    std::vector<uint8_t> seen(n_jobs, 0);
    for (uint64_t p = 0; p < n_jobs; p++) {
        const uint32_t k = pl->order[p];
        if (k >= n_jobs || seen[k]) return "job " + S(k) + " missing or planned twice";
        seen[k] = 1;
        const DevJob &d = pl->h_jobs[p];
        if (d.aux != k || d.n != jobs[k].n || d.m != jobs[k].m || d.ref_off != jobs[k].ref_off || d.read_off != jobs[k].read_off)
            return "record of job " + S(k) + " differs from the job";
    }
    uint64_t covered = 0;
    for (const Launch &L : pl->launches) {
        if (L.first != covered || L.count == 0) return "launches do not partition the plan";
        covered += L.count;
    }
    if (covered != n_jobs) return "launches cover " + S(covered) + " of " + S(n_jobs) + " jobs";
    std::vector<uint8_t> tseen;
    {
        const std::string e = verify_tile_arrays(cfg, jobs, n_jobs, pl, ht.tiles.data(), ht.tiles.size(), ht.spans.data(),
                                                 ht.spans.size(), ht.tjobs.data(), ht.tjobs.size(), ht.masks, tseen);
        if (!e.empty()) return e;
    }
    for (uint64_t p = 0; p < pl->n_tile_jobs; p++) if (!tseen[pl->order[p]]) return "tile-class job without a tile";
    return "";
}

int build_plan(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, bool traceback, rawdtw_plan **out)
{
    *out = nullptr;
    if (!ctx) return RAWDTW_ERR_INVALID;
    if (n_jobs > 0 && !jobs) return fail(ctx, RAWDTW_ERR_INVALID, "jobs is NULL");
    if (n_jobs >= (1ull << 32)) return fail(ctx, RAWDTW_ERR_INVALID, "more than 2^32-1 jobs in one batch");
    rawdtw_plan *pl = new (std::nothrow) rawdtw_plan;
    if (!pl) return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed");
    pl->ctx = ctx;
    HostTiles ht;
    std::string err;
    int st;
    try {
        st = plan_host(cfg_of(ctx), jobs, n_jobs, traceback, pl, ht, err);
    } catch (const std::bad_alloc &) {
        st = RAWDTW_ERR_OOM; err = "host allocation failed";
    }
    if (st != RAWDTW_OK) { delete pl; return fail(ctx, st, err); }
    ctx->live_plans.push_back(pl);

    const uint64_t n_dev_jobs = n_jobs - pl->n_tile_jobs;
    if ((st = dev_alloc(ctx, &pl->d_jobs, n_dev_jobs)) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_tiles, (uint64_t)ht.tiles.size())) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_spans, (uint64_t)ht.spans.size())) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_tjobs, (uint64_t)ht.tjobs.size())) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_masks, (uint64_t)ht.masks.size())) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_aux, n_dev_jobs)) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_cost, n_jobs)) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_bnd, pl->bnd_floats)) != RAWDTW_OK) {
        rawdtw_plan_destroy(pl);
        return st;
    }
    if (pl->dir_bytes) { // the context's direction workspace (one traceback batch at a time per context)
        if (ctx->tb_dir_bytes < pl->dir_bytes) {
            if (ctx->d_tb_dir) (void)hipFree(ctx->d_tb_dir);
            ctx->d_tb_dir = nullptr; ctx->tb_dir_bytes = 0;
            const uint64_t want = pl->dir_bytes + pl->dir_bytes / 8;
            if ((st = dev_alloc(ctx, &ctx->d_tb_dir, want)) != RAWDTW_OK) { rawdtw_plan_destroy(pl); return st; }
            ctx->tb_dir_bytes = want;
        }
        pl->d_dir = ctx->d_tb_dir;
        pl->dir_borrowed = true;
    }
    if (n_jobs) {
        hipError_t e = hipSuccess;
        auto up = [&](void *dst, const void *src, size_t bytes) {
            if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
        };
        up(pl->d_jobs, pl->h_jobs.data() + pl->n_tile_jobs, n_dev_jobs * sizeof(DevJob));
        up(pl->d_aux, pl->h_aux.data(), n_dev_jobs * sizeof(FullAux));
        up(pl->d_tiles, ht.tiles.data(), ht.tiles.size() * sizeof(TileDesc));
        up(pl->d_spans, ht.spans.data(), ht.spans.size() * sizeof(TileSpan));
        up(pl->d_tjobs, ht.tjobs.data(), ht.tjobs.size() * sizeof(TileJob));
        up(pl->d_masks, ht.masks.data(), ht.masks.size() * 8);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            rawdtw_plan_destroy(pl);
            return hip_fail(ctx, e, "uploading job descriptors");
        }
    }
    *out = pl;
    return RAWDTW_OK;
}

// cells evaluated by plan positions [p0, p1) (exact band cell sets; reporting only)
uint64_t count_cells(const rawdtw_plan *pl, uint64_t p0, uint64_t p1)
{
    if (p1 <= p0) return 0;
    const int T = (int)std::min<uint64_t>(std::max(pl->plan_threads_used, 1), (p1 - p0) / 32768 + 1);
    std::vector<uint64_t> part(T, 0);
    parallel_for(T, [&](int t) {
        uint64_t c = 0;
        for (uint64_t p = p0 + (p1 - p0) * (uint64_t)t / T; p < p0 + (p1 - p0) * (uint64_t)(t + 1) / T; p++) {
            const DevJob &d = pl->h_jobs[p];
            c += d.R < 0 ? (uint64_t)d.n * d.m : banded_cells(d.n, d.m, d.R);
        }
        part[t] = c;
    });
    uint64_t cells = 0;
    for (uint64_t c : part) cells += c;
    return cells;
}

int run_launch(rawdtw_ctx *ctx, rawdtw_plan *pl, const Launch &L, hipStream_t stream)
{
    // device job records exist only for the non-tile jobs (plan positions >= n_tile_jobs)
    const DevJob *jobs = pl->d_jobs + (L.first >= pl->n_tile_jobs ? L.first - pl->n_tile_jobs : 0);
    const FullAux *aux = pl->d_aux + (L.first >= pl->n_tile_jobs ? L.first - pl->n_tile_jobs : 0);
    float *out = pl->d_cost; // job order: every kernel stores at out[job.aux]
    hipError_t e = hipSuccess;
    {   // timing experiments only (RAWDTW_OPTS=debug_skip_kinds=mask, bit = launch kind, the 16-lane-row kernel = bit 15)
        const bool grp = L.kind == kKindBandWreg && (L.param == -16 || L.param == -8);
        if (ctx->debug_skip_kinds & (1u << (grp ? 15 : L.kind))) return RAWDTW_OK;
    }
    switch (L.kind) {
    case kKindBandLane:
        e = launch_band_tile(false, ctx->tile_threads, pl->d_tiles, pl->n_tiles, pl->d_spans, pl->d_tjobs, pl->d_masks, pl->tile_lds_floats,
                             ctx->d_ev, ctx->d_ref, out, stream);
        break;
    case kKindBandLaneHi:
        e = launch_band_tile(true, 64, pl->d_tiles + pl->n_tiles, pl->n_tiles_hi, pl->d_spans, pl->d_tjobs, pl->d_masks,
                             pl->tile_hi_lds_floats, ctx->d_ev, ctx->d_ref, out, stream);
        break;
    case kKindBandWreg:
        e = launch_band_wreg(L.param, jobs, L.count, ctx->d_ev, ctx->d_ref, out, stream);
        break;
    case kKindBandWave:
        e = launch_band_wave(jobs, L.count, (uint32_t)L.param, ctx->d_ev, ctx->d_ref, out, stream);
        break;
    case kKindFullWave:
    case kKindFullTb:
        e = launch_full_wave(L.param, L.kind == kKindFullTb, jobs, L.count, aux, ctx->d_ev,
                             ctx->d_ref, out, pl->d_bnd, pl->d_dir, stream);
        break;
    default:
        return fail(ctx, RAWDTW_ERR_INVALID, "unknown launch kind");
    }
    if (e != hipSuccess) return hip_fail(ctx, e, "kernel launch");
    return RAWDTW_OK;
}

// Which launches of a plan travel as one k_band_merged launch (indices into pl->launches, -1 = none).

MergeSel merge_of(const rawdtw_ctx *ctx, const rawdtw_plan *pl)
{
    MergeSel m;
    if (!ctx->merge_small || ctx->tile_threads != 256 || (ctx->n_side > 0 && !ctx->serial_launches)) return m;
    int tile = -1;
    for (size_t i = 0; i < pl->launches.size(); i++) {
        const Launch &L = pl->launches[i];
        if (L.kind == kKindBandLane) tile = (int)i;
        else if (L.kind == kKindBandWreg && L.param == -16) m.grp16 = (int)i;
        else if (L.kind == kKindBandWreg && L.param == -8) m.grp8 = (int)i;
        else if (L.kind == kKindBandWreg && L.param == 0) m.wreg = (int)i;
    }
    if (tile >= 0 && (m.grp16 >= 0 || m.grp8 >= 0 || m.wreg >= 0)) m.tile = tile;
    else m = MergeSel{};
    return m;
}

int run_merged(rawdtw_ctx *ctx, rawdtw_plan *pl, const MergeSel &m, hipStream_t stream)
{
    auto recs = [&](int i, uint64_t &n) -> const DevJob * {
        n = 0;
        if (i < 0) return nullptr;
        const Launch &L = pl->launches[i];
        n = L.count;
        return pl->d_jobs + (L.first - pl->n_tile_jobs);
    };
    uint64_t n_w = 0, n_g = 0, n_h = 0;
    const DevJob *wj = recs(m.wreg, n_w), *gj = recs(m.grp16, n_g), *hj = recs(m.grp8, n_h);
    hipError_t e = launch_band_merged(pl->d_tiles, pl->n_tiles, pl->d_spans, pl->d_tjobs, pl->d_masks, pl->tile_lds_floats,
                                      wj, n_w, gj, n_g, hj, n_h, ctx->d_ev, ctx->d_ref, pl->d_cost, stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "kernel launch");
    return RAWDTW_OK;
}

// All launches of a plan are independent: fork them over the main and side streams (heaviest
// first), join back on the main stream.  `ev`, when given, receives a start/stop event pair per
// launch (2*n entries), recorded on the stream that launch runs on.
int run_all_launches(rawdtw_ctx *ctx, rawdtw_plan *pl, hipEvent_t *ev)
{
    const size_t nl = pl->launches.size();
    if (nl == 0) return RAWDTW_OK;
    // hipGetLastError is sticky per thread: an error a library left behind while probing (rocPRIM's device queries
    // during planning do) would otherwise be reported as the first kernel launch's
    (void)hipGetLastError();
    const bool fork = nl > 1 && !ctx->serial_launches && ctx->n_side > 0;
    if (fork) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        for (int k = 0; k < ctx->n_side; k++) HIP_TRY(ctx, hipStreamWaitEvent(ctx->side[k], ctx->ev_fork, 0));
    }
    int st = RAWDTW_OK;
    const MergeSel mg = merge_of(ctx, pl);
    for (size_t q = 0; q < nl && st == RAWDTW_OK; q++) {
        const size_t i = pl->run_order[q];
        const int sl = fork ? (int)(q % (ctx->n_side + 1)) : 0;
        hipStream_t s = sl == 0 ? ctx->stream : ctx->side[sl - 1];
        if (ev && hipEventRecord(ev[2 * i], s) != hipSuccess) st = RAWDTW_ERR_DEVICE;
        if (mg.on() && ((int)i == mg.grp16 || (int)i == mg.grp8 || (int)i == mg.wreg)) { /* travels inside the tile launch */ }
        else if (mg.on() && (int)i == mg.tile) { if (st == RAWDTW_OK) st = run_merged(ctx, pl, mg, s); }
        else if (st == RAWDTW_OK) st = run_launch(ctx, pl, pl->launches[i], s);
        if (st == RAWDTW_OK && ev && hipEventRecord(ev[2 * i + 1], s) != hipSuccess) st = RAWDTW_ERR_DEVICE;
    }
    if (fork)
        for (int k = 0; k < ctx->n_side; k++) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_join[k], ctx->side[k]));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[k], 0));
        }
    return st;
}

} // namespace capi
} // namespace rawdtw

extern "C" {

int rawdtw_plan_create(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, rawdtw_plan **out)
{
    if (!out) return RAWDTW_ERR_INVALID;
    if (ctx) { hipError_t e = hipSetDevice(ctx->device); if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice"); }
    return build_plan(ctx, jobs, n_jobs, false, out);
}

int rawdtw_plan_dry_run(uint64_t n_events, uint64_t n_reference, const rawdtw_job_t *jobs, uint64_t n_jobs, int threads,
                        const char *const *option_names, const int64_t *option_values, uint32_t n_options,
                        rawdtw_plan_info_t *info, uint64_t *n_tiles, char *message, uint32_t message_cap)
{
    auto say = [&](const std::string &s) {
        if (message && message_cap) { snprintf(message, message_cap, "%s", s.c_str()); }
    };
    say("");
    if ((n_jobs && !jobs) || n_jobs >= (1ull << 32) || (n_options && (!option_names || !option_values))) return RAWDTW_ERR_INVALID;
    PlanCfg cfg;
    cfg.n_ev = n_events; cfg.n_ref = n_reference; cfg.threads = threads;
    bool verify = true;
    for (uint32_t i = 0; i < n_options; i++) {
        const char *nm = option_names[i];
        const int64_t v = option_values[i];
        if (!strcmp(nm, "verify")) verify = v != 0; // dry run only: skip the self-check (to time the planner alone)
        else if (!strcmp(nm, "tile_lds_floats")) cfg.tile_lds_floats = (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 1024), 40000);
        else if (!strcmp(nm, "tile_max_jobs")) cfg.tile_max_jobs = (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 64), 65535);
        else if (!strcmp(nm, "tile_max_spans")) cfg.tile_max_spans = (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 8), 4096);
        else if (!strcmp(nm, "sort_n")) cfg.sort_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 0), 255);
        else if (!strcmp(nm, "sort_r1_n")) cfg.sort_r1_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 0), 255);
        else if (!strcmp(nm, "sort_r3")) cfg.sort_r3 = v != 0;
        else if (!strcmp(nm, "sorted_tile_jobs")) cfg.sorted_tile_jobs = (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 16), 1024);
        else if (!strcmp(nm, "full_wg")) cfg.full_wg = v != 0;
        else if (!strcmp(nm, "grp16")) cfg.grp16 = v != 0;
        else if (!strcmp(nm, "grp8")) cfg.grp8 = v != 0;
        else if (!strcmp(nm, "micro_max_n")) cfg.micro_max_n = v >= 8 ? 8 : (v >= 4 ? 4 : 0);
        else if (!strcmp(nm, "lane_hi")) cfg.lane_hi = v != 0;
        else if (!strcmp(nm, "lane_hi_max_n")) cfg.lane_hi_max_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 8), 200);
        else if (!strcmp(nm, "lane_max_n")) cfg.lane_max_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 8), kLaneMaxN);
        else if (!strcmp(nm, "lane_max_radius")) cfg.lane_max_radius = v < 0 ? 0 : (v > kMaxLaneRadius ? kMaxLaneRadius : (int)v);
        else { say(std::string("unknown option ") + nm); return RAWDTW_ERR_INVALID; }
    }
    rawdtw_plan pl;
    HostTiles ht;
    std::string err;
    int st;
    try {
        st = plan_host(cfg, jobs, n_jobs, false, &pl, ht, err);
        if (st == RAWDTW_OK && verify) {
            err = verify_host_plan(cfg, jobs, n_jobs, &pl, ht);
            if (!err.empty()) st = RAWDTW_ERR_DEVICE + 100; // never returned for a correct planner
        }
    } catch (const std::bad_alloc &) {
        st = RAWDTW_ERR_OOM; err = "host allocation failed";
    }
    say(err);
    if (st != RAWDTW_OK) return st;
    if (info) { pl.info.cells = verify ? count_cells(&pl, 0, n_jobs) : 0; *info = pl.info; }
    if (n_tiles) *n_tiles = ht.tiles.size();
    return RAWDTW_OK;
}

int rawdtw_plan_info(const rawdtw_plan *plan, rawdtw_plan_info_t *info)
{
    if (!plan || !info) return RAWDTW_ERR_INVALID;
    rawdtw_plan *pl = const_cast<rawdtw_plan *>(plan);
    if (!pl->cells_counted) {
        pl->info.cells = count_cells(pl, 0, pl->n_jobs);
        pl->cells_counted = true;
    }
    *info = pl->info;
    return RAWDTW_OK;
}

int rawdtw_plan_run(rawdtw_ctx *ctx, rawdtw_plan *plan)
{
    if (!ctx || !plan || plan->ctx != ctx) return fail(ctx, RAWDTW_ERR_INVALID, "plan does not belong to this context");
    return run_all_launches(ctx, plan, nullptr);
}

int rawdtw_plan_run_timed(rawdtw_ctx *ctx, rawdtw_plan *plan, float *launch_ms, uint32_t *launch_kind,
                          uint32_t cap)
{
    if (!ctx || !plan || plan->ctx != ctx) return fail(ctx, RAWDTW_ERR_INVALID, "plan does not belong to this context");
    const size_t nl = plan->launches.size();
    std::vector<hipEvent_t> ev(2 * nl, nullptr);
    for (auto &e : ev) HIP_TRY(ctx, hipEventCreate(&e));
    int st = run_all_launches(ctx, plan, ev.data());
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && st == RAWDTW_OK) st = RAWDTW_ERR_DEVICE;
    for (size_t i = 0; i < nl && st == RAWDTW_OK; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) { st = RAWDTW_ERR_DEVICE; break; }
        if (i < cap) {
            if (launch_ms) launch_ms[i] = ms;
            if (launch_kind) launch_kind[i] = plan->launches[i].kind | ((uint32_t)plan->launches[i].param << 8);
        }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    if (st != RAWDTW_OK && ctx->err.empty()) ctx->err = "timed run failed";
    return st;
}

int rawdtw_plan_fetch(rawdtw_ctx *ctx, rawdtw_plan *plan, float *out_cost)
{
    if (!ctx || !plan || plan->ctx != ctx || (!out_cost && plan->n_jobs))
        return fail(ctx, RAWDTW_ERR_INVALID, "bad arguments to plan_fetch");
    if (plan->n_jobs == 0) return RAWDTW_OK;
    HIP_TRY(ctx, hipMemcpyAsync(out_cost, plan->d_cost, plan->n_jobs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RAWDTW_OK;
}

int rawdtw_plan_device_costs(const rawdtw_plan *plan, const float **d_cost, const uint32_t **h_order)
{
    if (!plan) return RAWDTW_ERR_INVALID;
    if (d_cost) *d_cost = plan->d_cost;
    if (h_order) *h_order = plan->order.data();
    return RAWDTW_OK;
}

// a plan's device arrays (the host record stays: rawdtw_plan_info still answers)
} // extern "C"
namespace rawdtw { namespace capi {
void plan_release_device(rawdtw_plan *plan)
{
    auto drop = [](auto *&p) { if (p) (void)hipFree(p); p = nullptr; };
    drop(plan->d_jobs); drop(plan->d_aux); drop(plan->d_tiles); drop(plan->d_spans); drop(plan->d_tjobs);
    drop(plan->d_masks); drop(plan->d_cost); drop(plan->d_bnd);
    if (plan->d_dir && !plan->dir_borrowed) (void)hipFree(plan->d_dir);
    plan->d_dir = nullptr;
}
} } // namespace rawdtw::capi
extern "C" {

int rawdtw_plan_destroy(rawdtw_plan *plan)
{
    if (!plan) return RAWDTW_OK;
    if (rawdtw_ctx *ctx = plan->ctx) { // (null: rawdtw_destroy came first and took the device arrays with it)
        (void)hipSetDevice(ctx->device);
        plan_release_device(plan);
        unregister(ctx->live_plans, plan);
    }
    delete plan;
    return RAWDTW_OK;
}

int rawdtw_score_batch(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, const float *h_events,
                       uint64_t n_events, float *out_cost)
{
    int st = rawdtw_upload_events(ctx, h_events, n_events);
    if (st != RAWDTW_OK) return st;
    rawdtw_plan *pl = nullptr;
    st = rawdtw_plan_create(ctx, jobs, n_jobs, &pl);
    if (st != RAWDTW_OK) return st;
    st = rawdtw_plan_run(ctx, pl);
    if (st == RAWDTW_OK) st = rawdtw_plan_fetch(ctx, pl, out_cost);
    rawdtw_plan_destroy(pl);
    return st;
}

} // extern "C"
