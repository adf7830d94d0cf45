// rawdtw_capi.cpp -- the C ABI of librawdtw.so (include/rawdtw.h): context, arenas, the
// batch planner and the launch sequences.  Host code only; kernels live in rawdtw_kernels.hip.
//
// There is NO CPU fallback in here: every scoring entry point runs the HIP kernels or
// returns an error status.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rawdtw_internal.h"

using namespace rawdtw;

struct rawdtw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // side streams: independent launches of one batch run concurrently (fork/join around the main stream)
    static constexpr int kSide = 3;
    hipStream_t side[kSide] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kSide] = {nullptr, nullptr, nullptr};
    bool serial_launches = false;
    int n_side = 0; // side streams used to fork the launches of one batch (RAWDTW_SIDE_STREAMS, 0..kSide). 0: the
                    // launches of a batch run in sequence on its one stream and overlap comes from several batches in
                    // flight on several contexts (swept: best throughput and cleaner per-kernel timings)
    uint32_t lane_hi_max_n = 96;
    int micro_max_n = 8; // shapes with longer side <= this use the micro paths (0: none, 4: micro4 only)
    bool grp16 = true; // bands of at most 16 offsets: four jobs per wave (else one job per wave)
    bool full_wg = true; // full-matrix jobs with >= 3 strips: four waves per job, pipelined strips
    bool lane_hi = false; // radii 4..8 on the second tile-kernel instance (else on k_band_wreg<1>)
    uint32_t tile_lds_floats = kTileLdsFloats, tile_max_jobs = kTileMaxJobs;
    uint32_t lane_max_n = kLaneMaxN;
    int lane_max_radius = kMaxLaneRadius; // radii above this go to the register-resident wave kernel (RAWDTW_LANE_MAX_R)
    // reference arena
    float *d_ref = nullptr;
    uint64_t n_ref = 0;
    bool own_ref = false;
    std::vector<uint64_t> ref_off; // 2*n_seq entries: [seq*2 + 0] = forward (strand 1), [seq*2 + 1] = reverse
    std::vector<uint32_t> ref_len;
    // event arena
    float *d_ev = nullptr;
    uint64_t n_ev = 0, cap_ev = 0;
    bool own_ev = false;
    std::string err;
};

struct rawdtw_plan {
    rawdtw_ctx *ctx = nullptr;
    uint64_t n_jobs = 0;
    std::vector<uint32_t> order;   // plan position -> job index
    std::vector<Launch> launches;
    std::vector<uint32_t> run_order; // launch indices, heaviest first
    std::vector<int32_t> launch_rpl;
    DevJob *d_jobs = nullptr;      // records of the jobs NOT handled by the tile kernel (plan order, after the tile jobs)
    uint64_t n_tile_jobs = 0;      // plan positions [0, n_tile_jobs) are tile-kernel jobs, in job order
    TileDesc *d_tiles = nullptr;
    TileSpan *d_spans = nullptr;
    TileJob *d_tjobs = nullptr;
    unsigned long long *d_masks = nullptr; // band bitmasks of the micro-path shapes
    uint64_t n_tiles = 0, n_tiles_hi = 0;   // d_tiles = [bulk tiles][wide-band tiles]
    uint32_t tile_lds_floats = 0, tile_hi_lds_floats = 0;
    FullAux *d_aux = nullptr;      // indexed like d_jobs (only meaningful for full-matrix jobs)
    float *d_cost = nullptr;
    float *d_bnd = nullptr;
    uint8_t *d_dir = nullptr;
    uint64_t bnd_floats = 0, dir_bytes = 0;
    std::vector<DevJob> h_jobs;    // plan order (kept for traceback + info)
    std::vector<FullAux> h_aux;
    rawdtw_plan_info_t info{};
    bool cells_counted = false;
};

struct rawdtw_index {
    std::string path;
    uint32_t pars[8] = {0};
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    std::vector<uint64_t> fwd_pos; // file offset of each sequence's forward array (reverse follows it)
};

struct rawdtw_batch {
    rawdtw_ctx *ctx = nullptr;
    rawdtw_plan *plan = nullptr;
    rawdtw_align_opt_t opt{};
    uint64_t n_reads = 0, n_chains = 0;
    ChainDesc *d_chains = nullptr;
    uint64_t *d_chain_off = nullptr;
    uint32_t *d_fold_order = nullptr; // chain ids, longest chain first
    float *d_full = nullptr, *d_gate = nullptr, *d_score = nullptr;
    uint8_t *d_keep = nullptr;
    std::vector<hipEvent_t> ev; // event pairs of the runs enqueued since the last collect
    uint32_t ev_runs = 0;
};

namespace {

int fail(rawdtw_ctx *ctx, int status, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return status;
}

int hip_fail(rawdtw_ctx *ctx, hipError_t e, const char *what)
{
    int st = (e == hipErrorOutOfMemory) ? RAWDTW_ERR_OOM : RAWDTW_ERR_DEVICE;
    return fail(ctx, st, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(ctx, expr)                                                                            \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return hip_fail((ctx), e_, #expr);                                      \
    } while (0)

// post-slant radius, dtw.cpp:298-300 (unsigned 32-bit arithmetic for the correction)
inline int slanted_radius(uint32_t n, uint32_t m, int r0)
{
    uint32_t N = n > m ? n : m, M = n > m ? m : n;
    uint32_t extra = ((N - M) * (uint32_t)r0 + N - 1u) / N;
    return r0 + (int)extra;
}

// exact size of the band's cell set (same walk as the kernels; host side, for reporting)
uint64_t banded_cells(uint32_t n, uint32_t m, int R)
{
    const uint32_t N = n > m ? n : m, M = n > m ? m : n;
    const int P = R + ((R % 2 == 0) ? 1 : 0), S = R + ((R % 2 == 1) ? 1 : 0);
    uint64_t cells = 1;
    int row = 0;
    uint32_t rem = 0;
    for (uint32_t col = 1; col < N; col++) {
        rem += M;
        const bool adv = rem >= N;
        if (adv) { rem -= N; row++; }
        for (int pass = adv ? 0 : 1; pass < 2; pass++) {
            const int len = pass == 0 ? S : P;
            const int si = pass == 0 ? (int)col + S / 2 - 1 : (int)col + P / 2;
            const int sj = pass == 0 ? row - S / 2 : row - P / 2;
            int lo = 0, hi = len;
            lo = std::max(lo, si - (int)N + 1);
            lo = std::max(lo, -sj);
            hi = std::min(hi, si + 1);
            hi = std::min(hi, (int)M - sj);
            if (hi > lo) cells += (uint64_t)(hi - lo);
        }
    }
    return cells;
}

// bitmask of the band's cell set for a shape whose longer side is <= 8: bit 8*j + i  <=>  cell
// (i over the longer sequence, j over the shorter) is evaluated (same walk as banded_cells)
uint64_t band_mask8(uint32_t N, uint32_t M, int R)
{
    const int P = R + ((R % 2 == 0) ? 1 : 0), S = R + ((R % 2 == 1) ? 1 : 0);
    uint64_t mask = 1; // (0,0)
    int row = 0;
    uint32_t rem = 0;
    for (uint32_t col = 1; col < N; col++) {
        rem += M;
        const bool adv = rem >= N;
        if (adv) { rem -= N; row++; }
        for (int pass = adv ? 0 : 1; pass < 2; pass++) {
            const int len = pass == 0 ? S : P;
            const int si = pass == 0 ? (int)col + S / 2 - 1 : (int)col + P / 2;
            const int sj = pass == 0 ? row - S / 2 : row - P / 2;
            int lo = 0, hi = len;
            lo = std::max(lo, si - (int)N + 1);
            lo = std::max(lo, -sj);
            hi = std::min(hi, si + 1);
            hi = std::min(hi, (int)M - sj);
            for (int o = lo; o < hi; o++) mask |= 1ull << (8 * (sj + o) + (si - o));
        }
    }
    return mask;
}

inline int full_rpl(uint32_t ny)
{
    return ny <= 64 ? 1 : ny <= 128 ? 2 : ny <= 256 ? 4 : 8;
}

inline uint64_t dir_bytes_for(uint32_t n, uint32_t m, int rpl)
{
    const uint32_t NX = n > m ? n : m, NY = n > m ? m : n;
    const uint64_t strips = (NY + 64ull * rpl - 1) / (64ull * rpl);
    const uint64_t spb = rpl == 8 ? 8 : 16; // steps per 16-byte block (k_full_wave)
    return strips * (((uint64_t)NX + 63 + spb - 1) / spb) * 64 * 16;
}

template <typename T> int dev_alloc(rawdtw_ctx *ctx, T **p, uint64_t count)
{
    *p = nullptr;
    if (count == 0) return RAWDTW_OK;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return RAWDTW_OK;
}

int ensure_events_capacity(rawdtw_ctx *ctx, uint64_t n)
{
    if (ctx->own_ev && ctx->cap_ev >= n) return RAWDTW_OK;
    if (ctx->own_ev && ctx->d_ev) (void)hipFree(ctx->d_ev);
    ctx->d_ev = nullptr;
    ctx->own_ev = true;
    uint64_t cap = std::max<uint64_t>(n + (n >> 2), 1024);
    cap = (cap + 63) & ~63ull;
    int st = dev_alloc(ctx, &ctx->d_ev, cap);
    if (st != RAWDTW_OK) { ctx->cap_ev = 0; return st; }
    ctx->cap_ev = cap;
    return RAWDTW_OK;
}

// Build a plan. traceback=true: every job must be a full-matrix job and gets a direction buffer.
int build_plan(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, bool traceback,
               rawdtw_plan **out)
{
    *out = nullptr;
    if (!ctx) return RAWDTW_ERR_INVALID;
    if (n_jobs > 0 && !jobs) return fail(ctx, RAWDTW_ERR_INVALID, "jobs is NULL");
    if (n_jobs >= (1ull << 32)) return fail(ctx, RAWDTW_ERR_INVALID, "more than 2^32-1 jobs in one batch");
    rawdtw_plan *pl = new (std::nothrow) rawdtw_plan;
    if (!pl) return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed");
    pl->ctx = ctx;
    pl->n_jobs = n_jobs;

    // sort key: class in the top bits, then descending length so long jobs start first
    //   banded tile (lane DP): class = 0, kept in JOB order (R in 0..lane_max_radius, longer side <= 73)
    //   banded wave, register: class = 40 + log2(chunks)      (radius+1 <= 64*chunks, chunks <= 32)
    //   banded wave, LDS     : class = 48 + lds bucket
    //   full                 : class = 56 + log2(rpl)
    // inside a class: longer side descending, then shorter side descending (waves share one shape)
    struct Keyed { uint64_t key; uint32_t idx; int32_t R; };
    std::vector<Keyed> keyed(n_jobs);
    uint64_t alg_bytes = 0;
    for (uint64_t k = 0; k < n_jobs; k++) {
        const rawdtw_job_t &j = jobs[k];
        if (j.n == 0 || j.m == 0 || j.band_radius < RAWDTW_FULL ||
            j.n >= 0x7fffffffu || j.m >= 0x7fffffffu) {
            delete pl;
            return fail(ctx, RAWDTW_ERR_INVALID,
                        "job " + std::to_string(k) + ": zero length or negative band radius (dtw.cpp:274-277 asserts)");
        }
        if ((uint64_t)j.read_off + j.n > ctx->n_ev || j.ref_off + j.m > ctx->n_ref) {
            delete pl;
            return fail(ctx, RAWDTW_ERR_RANGE, "job " + std::to_string(k) + ": window outside the uploaded arenas");
        }
        alg_bytes += 4ull * ((uint64_t)j.n + j.m) + 4 + 32;
        const uint32_t N = std::max(j.n, j.m), NY = std::min(j.n, j.m);
        uint64_t cls;
        int32_t R = -1;
        if (j.band_radius == RAWDTW_FULL) {
            const int rpl = full_rpl(NY);
            cls = 56 + (rpl == 1 ? 0 : rpl == 2 ? 1 : rpl == 4 ? 2 : 3);
            if (rpl == 8 && NY > 2 * 512u && ctx->full_wg) cls = 60; // >= 3 strips: four waves per job
        } else {
            if (traceback) {
                delete pl;
                return fail(ctx, RAWDTW_ERR_UNSUPPORTED,
                            "traceback of a banded job is not implemented (rmap.cpp:223-225 assert(false))");
            }
            R = slanted_radius(j.n, j.m, j.band_radius);
            if (R < 0 || R + 1 > kMaxWaveBandK) {
                delete pl;
                return fail(ctx, RAWDTW_ERR_UNSUPPORTED, "band radius too large for the LDS-resident band kernel");
            }
            const uint32_t K = (uint32_t)R + 1;
            if (R <= ctx->lane_max_radius && N <= ctx->lane_max_n) cls = 0;
            else if (R <= kMaxLaneRadiusHi && ctx->lane_hi && N <= ctx->lane_hi_max_n) cls = 1; // any radius 0..8 (the instance covers all)
            else if (K <= 16 && ctx->grp16) cls = 39; // four jobs per wave (16-lane rows)
            else if (K <= 64u * kMaxWregChunks) {
                uint32_t chunks = 1, lg = 0;
                while (64u * chunks < K) { chunks <<= 1; lg++; }
                cls = chunks <= 4 ? 40 : 40 + lg; // one merged launch for radius+1 <= 256 (param 0)
            } else {
                cls = 48 + (K <= 8192 ? 0 : 1); // LDS buckets: 3K floats
            }
        }
        if (cls <= 1) keyed[k].key = (cls << 56) | k; // tile jobs stay in job order: consecutive parts share their spans
        else {
            const uint64_t lim = (1ull << 28) - 1;
            keyed[k].key = (cls << 56) | ((lim - std::min<uint64_t>(N, lim)) << 28) | (lim - std::min<uint64_t>(NY, lim));
        }
        keyed[k].idx = (uint32_t)k;
        keyed[k].R = R;
    }
    std::sort(keyed.begin(), keyed.end(), [](const Keyed &x, const Keyed &y) {
        return x.key != y.key ? x.key < y.key : x.idx < y.idx;
    });

    pl->order.resize(n_jobs);
    pl->h_jobs.resize(n_jobs);
    pl->h_aux.assign(n_jobs, FullAux{0, 0});
    uint64_t bnd = 0, dirb = 0;
    for (uint64_t p = 0; p < n_jobs; p++) {
        const rawdtw_job_t &j = jobs[keyed[p].idx];
        pl->order[p] = keyed[p].idx;
        DevJob &d = pl->h_jobs[p];
        d.ref_off = j.ref_off; d.read_off = j.read_off; d.n = j.n; d.m = j.m;
        d.R = keyed[p].R; d.flags = j.exclude_last ? kFlagExcludeLast : 0u; d.aux = keyed[p].idx;
        const uint64_t cls = keyed[p].key >> 56;
        if (cls >= 56) {
            const int rpl = cls == 60 ? 8 : 1 << (cls - 56);
            const uint64_t rows = cls == 60 ? kFullWgWaves : 1; // boundary rows: a ring for the pipelined variant
            const uint32_t NX = std::max(j.n, j.m), NY = std::min(j.n, j.m);
            if (NY > 64u * rpl) { // multi-strip: needs a boundary row
                pl->h_aux[p].bnd_off = bnd;
                bnd += rows * (((uint64_t)NX + 63) & ~63ull);
            }
            if (traceback) {
                pl->h_aux[p].dir_off = dirb;
                dirb += (dir_bytes_for(j.n, j.m, rpl) + 255) & ~255ull;
            }
        }
        // launches: maximal runs of equal class
        if (pl->launches.empty() || (keyed[p - 1].key >> 56) != cls) {
            Launch L{};
            L.first = p; L.count = 0;
            if (cls == 0) { L.kind = kKindBandLane; L.param = 0; }
            else if (cls == 1) { L.kind = kKindBandLaneHi; L.param = 0; }
            else if (cls == 39) { L.kind = kKindBandWreg; L.param = -16; }
            else if (cls < 48) { L.kind = kKindBandWreg; L.param = cls == 40 ? 0 : 1 << (cls - 40); }
            else if (cls < 56) { L.kind = kKindBandWave; L.param = 3 * kMaxWaveBandK; }
            else { L.kind = traceback ? kKindFullTb : kKindFullWave; L.param = cls == 60 ? 8 + 256 : 1 << (cls - 56); }
            pl->launches.push_back(L);
        }
        pl->launches.back().count++;
    }
    // ---- tiles for the lane-eligible jobs (plan positions [0, n_tile_jobs), job order) ----
    std::vector<TileDesc> tiles;
    std::vector<TileSpan> spans;
    std::vector<TileJob> tjobs;
    std::vector<unsigned long long> masks;
    std::vector<int32_t> mask_index(8 * 8 * (kMaxLaneRadius + 1), -1); // (N-1, M-1, R) -> index into masks
    uint32_t tile_lds_max = 0;
    for (const Launch &TL : pl->launches) if (TL.kind == kKindBandLane || TL.kind == kKindBandLaneHi) pl->n_tile_jobs += TL.count;
    tjobs.resize(pl->n_tile_jobs);
    for (Launch &TL : pl->launches) {
        if (TL.kind != kKindBandLane && TL.kind != kKindBandLaneHi) continue;
        const bool hi = TL.kind == kKindBandLaneHi;
        const uint64_t p0 = TL.first, nt = TL.first + TL.count;
        const uint32_t lds_budget = hi ? kTileHiLdsFloats : ctx->tile_lds_floats;
        const uint32_t max_jobs = hi ? kTileHiMaxJobs : ctx->tile_max_jobs;
        const size_t tiles_before = tiles.size();
        tile_lds_max = 0;
        struct Sp { uint64_t start, end; bool is_ref; uint32_t lds; }; // [start,end) in floats, start 4-aligned
        std::vector<Sp> cur;
        struct Pend { uint32_t spA, spB; uint64_t a0, b0; };
        std::vector<Pend> pend;
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
            // order the tile's records by (radius, longer side, shorter side): waves get one shape
            std::sort(tjobs.begin() + t_first, tjobs.begin() + t_end, [](const TileJob &x, const TileJob &y) {
                if (x.R != y.R) return x.R < y.R; // dispatch kind
                if (x.N != y.N) return x.N > y.N;
                if (x.M != y.M) return x.M > y.M;
                return x.aux < y.aux;
            });
            tiles.push_back(TileDesc{(uint32_t)t_first, (uint32_t)(t_end - t_first), span_first, (uint32_t)cur.size()});
            cur.clear(); pend.clear(); lds_used = 0; t_first = t_end;
        };
        // find (or make) the span that holds window [w0, w0+len) of the given arena; returns its index or -1
        auto place = [&](uint64_t w0, uint32_t len, bool is_ref, uint32_t &extra) -> int {
            extra = 0;
            for (int q = (int)cur.size() - 1; q >= 0 && q >= (int)cur.size() - 8; q--) {
                Sp &s = cur[q];
                if (s.is_ref != is_ref || w0 < s.start || w0 > s.end) continue;
                if (w0 + len <= s.end) return q; // already covered
                const uint32_t before = span_cost(s);
                Sp grown = s; grown.end = w0 + len;
                extra = span_cost(grown) - before;
                return q; // caller extends after the budget check
            }
            extra = (uint32_t)((((w0 & 3ull) + len) + 3) & ~3ull);
            return -1;
        };
        for (uint64_t p = p0; p < nt; p++) {
            const DevJob &d = pl->h_jobs[p];
            const bool swap = d.n < d.m; // dtw.cpp:284-292: A is the longer sequence
            const uint64_t a0 = swap ? d.ref_off : d.read_off, b0 = swap ? d.read_off : d.ref_off;
            const uint32_t NA = swap ? d.m : d.n, NB = swap ? d.n : d.m;
            const bool a_ref = swap, b_ref = !swap;
            for (int attempt = 0; attempt < 2; attempt++) {
                uint32_t ea = 0, eb = 0;
                int qa = place(a0, NA, a_ref, ea);
                // place B after tentatively accounting for A (a fresh span for A cannot serve B: other arena)
                int qb = place(b0, NB, b_ref, eb);
                const uint32_t new_spans = (qa < 0) + (qb < 0);
                if (attempt == 0 && (lds_used + ea + eb > lds_budget || cur.size() + new_spans > kTileMaxSpans ||
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
                if (!hi && NA <= (uint32_t)ctx->micro_max_n) { // micro path: band membership from a per-shape bitmask
                    int32_t &mi = mask_index[((NA - 1) * 8 + (NB - 1)) * (kMaxLaneRadius + 1) + d.R];
                    if (mi < 0) { mi = (int32_t)masks.size(); masks.push_back(band_mask8(NA, NB, d.R)); }
                    tj.pad = (uint32_t)mi;
                    tj.R = NA <= 4 ? 0 : 1;
                } else {
                    tj.R = (uint8_t)(2 + d.R);
                }
                break;
            }
        }
        close_tile(nt);
        if (hi) { pl->n_tiles_hi = tiles.size() - tiles_before; pl->tile_hi_lds_floats = tile_lds_max; }
        else { pl->n_tiles = tiles.size() - tiles_before; pl->tile_lds_floats = tile_lds_max; }
        TL.param = (int32_t)tile_lds_max;
    }
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
            for (uint64_t p = L.first; p < L.first + L.count; p++) {
                const DevJob &d = pl->h_jobs[p];
                const double N = std::max(d.n, d.m), M = std::min(d.n, d.m);
                const double w = d.R < 0 ? M : std::min<double>(2.0 * d.R + 1.0, M);
                // wave-per-job kernels spend a whole wave on one job
                work[i] += N * ((L.kind == kKindBandLane || L.kind == kKindBandLaneHi) ? w : std::max(w, 64.0));
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
    I.workspace_bytes = bnd * 4 + dirb + (n_jobs - pl->n_tile_jobs) * (sizeof(DevJob) + sizeof(FullAux)) + n_jobs * 4 +
                        tiles.size() * sizeof(TileDesc) + spans.size() * sizeof(TileSpan) + tjobs.size() * sizeof(TileJob);

    int st;
    const uint64_t n_dev_jobs = n_jobs - pl->n_tile_jobs;
    if ((st = dev_alloc(ctx, &pl->d_jobs, n_dev_jobs)) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_tiles, (uint64_t)tiles.size())) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_spans, (uint64_t)spans.size())) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_tjobs, (uint64_t)tjobs.size())) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_masks, (uint64_t)masks.size())) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_aux, n_dev_jobs)) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_cost, n_jobs)) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_bnd, bnd)) != RAWDTW_OK ||
        (st = dev_alloc(ctx, &pl->d_dir, dirb)) != RAWDTW_OK) {
        rawdtw_plan_destroy(pl);
        return st;
    }
    if (n_jobs) {
        hipError_t e = hipSuccess;
        if (n_dev_jobs)
            e = hipMemcpyAsync(pl->d_jobs, pl->h_jobs.data() + pl->n_tile_jobs, n_dev_jobs * sizeof(DevJob),
                               hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && n_dev_jobs)
            e = hipMemcpyAsync(pl->d_aux, pl->h_aux.data() + pl->n_tile_jobs, n_dev_jobs * sizeof(FullAux),
                               hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && !tiles.empty())
            e = hipMemcpyAsync(pl->d_tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && !spans.empty())
            e = hipMemcpyAsync(pl->d_spans, spans.data(), spans.size() * sizeof(TileSpan), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && !tjobs.empty())
            e = hipMemcpyAsync(pl->d_tjobs, tjobs.data(), tjobs.size() * sizeof(TileJob), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && !masks.empty())
            e = hipMemcpyAsync(pl->d_masks, masks.data(), masks.size() * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            rawdtw_plan_destroy(pl);
            return hip_fail(ctx, e, "uploading job descriptors");
        }
    }
    *out = pl;
    return RAWDTW_OK;
}

int run_launch(rawdtw_ctx *ctx, rawdtw_plan *pl, const Launch &L, hipStream_t stream)
{
    // device job records exist only for the non-tile jobs (plan positions >= n_tile_jobs)
    const DevJob *jobs = pl->d_jobs + (L.first >= pl->n_tile_jobs ? L.first - pl->n_tile_jobs : 0);
    const FullAux *aux = pl->d_aux + (L.first >= pl->n_tile_jobs ? L.first - pl->n_tile_jobs : 0);
    float *out = pl->d_cost; // job order: every kernel stores at out[job.aux]
    hipError_t e = hipSuccess;
    switch (L.kind) {
    case kKindBandLane:
        e = launch_band_tile(false, pl->d_tiles, pl->n_tiles, pl->d_spans, pl->d_tjobs, pl->d_masks, pl->tile_lds_floats,
                             ctx->d_ev, ctx->d_ref, out, stream);
        break;
    case kKindBandLaneHi:
        e = launch_band_tile(true, pl->d_tiles + pl->n_tiles, pl->n_tiles_hi, pl->d_spans, pl->d_tjobs, pl->d_masks,
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

// All launches of a plan are independent: fork them over the main and side streams (heaviest
// first), join back on the main stream.  `ev`, when given, receives a start/stop event pair per
// launch (2*n entries), recorded on the stream that launch runs on.
int run_all_launches(rawdtw_ctx *ctx, rawdtw_plan *pl, hipEvent_t *ev)
{
    const size_t nl = pl->launches.size();
    if (nl == 0) return RAWDTW_OK;
    const bool fork = nl > 1 && !ctx->serial_launches && ctx->n_side > 0;
    if (fork) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        for (int k = 0; k < ctx->n_side; k++) HIP_TRY(ctx, hipStreamWaitEvent(ctx->side[k], ctx->ev_fork, 0));
    }
    int st = RAWDTW_OK;
    for (size_t q = 0; q < nl && st == RAWDTW_OK; q++) {
        const size_t i = pl->run_order[q];
        const int sl = fork ? (int)(q % (ctx->n_side + 1)) : 0;
        hipStream_t s = sl == 0 ? ctx->stream : ctx->side[sl - 1];
        if (ev && hipEventRecord(ev[2 * i], s) != hipSuccess) st = RAWDTW_ERR_DEVICE;
        if (st == RAWDTW_OK) st = run_launch(ctx, pl, pl->launches[i], s);
        if (st == RAWDTW_OK && ev && hipEventRecord(ev[2 * i + 1], s) != hipSuccess) st = RAWDTW_ERR_DEVICE;
    }
    if (fork)
        for (int k = 0; k < ctx->n_side; k++) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_join[k], ctx->side[k]));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[k], 0));
        }
    return st;
}

} // namespace

extern "C" {

int rawdtw_abi_version(void) { return RAWDTW_ABI_VERSION; }

const char *rawdtw_status_string(int status)
{
    switch (status) {
    case RAWDTW_OK: return "ok";
    case RAWDTW_ERR_INVALID: return "invalid argument";
    case RAWDTW_ERR_DEVICE: return "HIP runtime error";
    case RAWDTW_ERR_OOM: return "out of memory";
    case RAWDTW_ERR_RANGE: return "job window out of range";
    case RAWDTW_ERR_UNSUPPORTED: return "unsupported";
    case RAWDTW_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
    }
}

int rawdtw_device_count(int *count)
{
    if (!count) return RAWDTW_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return RAWDTW_OK;
}

int rawdtw_create(int device_ordinal, rawdtw_ctx **out)
{
    if (!out) return RAWDTW_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RAWDTW_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return RAWDTW_ERR_INVALID;
    rawdtw_ctx *ctx = new (std::nothrow) rawdtw_ctx;
    if (!ctx) return RAWDTW_ERR_OOM;
    ctx->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return RAWDTW_ERR_DEVICE;
    }
    if (const char *e = getenv("RAWDTW_SIDE_STREAMS")) ctx->n_side = std::min(std::max(atoi(e), 0), (int)rawdtw_ctx::kSide);
    // only the side streams that will be used: HIP maps streams onto a handful of hardware queues, and an
    // idle stream still takes a slot in that rotation
    bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < ctx->n_side && ok; k++)
        ok = hipStreamCreateWithFlags(&ctx->side[k], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) { rawdtw_destroy(ctx); return RAWDTW_ERR_DEVICE; }
    if (const char *e = getenv("RAWDTW_LANE_HI")) ctx->lane_hi = atoi(e) != 0;
    if (const char *e = getenv("RAWDTW_LANE_HI_MAX_N")) ctx->lane_hi_max_n = (uint32_t)std::min(std::max(atoi(e), 8), 200);
    if (const char *e = getenv("RAWDTW_LANE_MAX_R")) {
        int v = atoi(e);
        ctx->lane_max_radius = v < 0 ? 0 : (v > kMaxLaneRadius ? kMaxLaneRadius : v);
    }
    *out = ctx;
    return RAWDTW_OK;
}

int rawdtw_destroy(rawdtw_ctx *ctx)
{
    if (!ctx) return RAWDTW_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
    for (int k = 0; k < rawdtw_ctx::kSide; k++) {
        if (ctx->side[k]) { (void)hipStreamSynchronize(ctx->side[k]); (void)hipStreamDestroy(ctx->side[k]); }
        if (ctx->ev_join[k]) (void)hipEventDestroy(ctx->ev_join[k]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->own_ref && ctx->d_ref) (void)hipFree(ctx->d_ref);
    if (ctx->own_ev && ctx->d_ev) (void)hipFree(ctx->d_ev);
    delete ctx;
    return RAWDTW_OK;
}

const char *rawdtw_last_error(const rawdtw_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rawdtw_sync(rawdtw_ctx *ctx)
{
    if (!ctx) return RAWDTW_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RAWDTW_OK;
}

int rawdtw_set_option(rawdtw_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return RAWDTW_ERR_INVALID;
    if (!strcmp(name, "serial_launches")) { ctx->serial_launches = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "tile_lds_floats")) { ctx->tile_lds_floats = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1024), 40000); return RAWDTW_OK; }
    if (!strcmp(name, "tile_max_jobs")) { ctx->tile_max_jobs = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 64), 65535); return RAWDTW_OK; }
    if (!strcmp(name, "full_wg")) { ctx->full_wg = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "grp16")) { ctx->grp16 = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "micro_max_n")) { ctx->micro_max_n = value >= 8 ? 8 : (value >= 4 ? 4 : 0); return RAWDTW_OK; }
    if (!strcmp(name, "lane_hi_max_n")) { ctx->lane_hi_max_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 8), 200); return RAWDTW_OK; }
    if (!strcmp(name, "lane_hi")) { ctx->lane_hi = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "lane_max_n")) { ctx->lane_max_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 8), kLaneMaxN); return RAWDTW_OK; }
    if (!strcmp(name, "lane_max_radius")) {
        ctx->lane_max_radius = value < 0 ? 0 : (value > kMaxLaneRadius ? kMaxLaneRadius : (int)value);
        return RAWDTW_OK;
    }
    return fail(ctx, RAWDTW_ERR_INVALID, std::string("unknown option ") + name);
}

int rawdtw_stream(rawdtw_ctx *ctx, void **stream)
{
    if (!ctx || !stream) return RAWDTW_ERR_INVALID;
    *stream = (void *)ctx->stream;
    return RAWDTW_OK;
}

int rawdtw_upload_reference(rawdtw_ctx *ctx, uint32_t n_seq, const float *const *fwd,
                            const float *const *rev, const uint32_t *len)
{
    if (!ctx || (n_seq && (!fwd || !rev || !len))) return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->own_ref && ctx->d_ref) (void)hipFree(ctx->d_ref);
    ctx->d_ref = nullptr; ctx->own_ref = true; ctx->n_ref = 0;
    ctx->ref_off.assign(2ull * n_seq, 0);
    ctx->ref_len.assign(len, len + n_seq);
    uint64_t total = 0;
    for (uint32_t s = 0; s < n_seq; s++) {
        // every array starts on a 16-byte boundary so 128-bit loads of window chunks stay aligned
        ctx->ref_off[2 * s] = total; total += ((uint64_t)len[s] + 3) & ~3ull;
        ctx->ref_off[2 * s + 1] = total; total += ((uint64_t)len[s] + 3) & ~3ull;
    }
    int st = dev_alloc(ctx, &ctx->d_ref, std::max<uint64_t>(total, 4));
    if (st != RAWDTW_OK) return st;
    ctx->n_ref = total;
    for (uint32_t s = 0; s < n_seq; s++) {
        if (len[s] == 0) continue;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ref + ctx->ref_off[2 * s], fwd[s], (size_t)len[s] * 4,
                                    hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ref + ctx->ref_off[2 * s + 1], rev[s], (size_t)len[s] * 4,
                                    hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RAWDTW_OK;
}

int rawdtw_reference_offset(const rawdtw_ctx *ctx, uint32_t seq, int strand, uint64_t *off)
{
    if (!ctx || !off || 2ull * seq + 1 >= ctx->ref_off.size() + 0ull) return RAWDTW_ERR_INVALID;
    // rmap.cpp:182-188: strand==1 -> forward_signals, otherwise reverse_signals
    *off = ctx->ref_off[2ull * seq + (strand == 1 ? 0 : 1)];
    return RAWDTW_OK;
}

int rawdtw_set_reference_device(rawdtw_ctx *ctx, const float *d_ref, uint64_t n_floats)
{
    if (!ctx || (!d_ref && n_floats)) return fail(ctx, RAWDTW_ERR_INVALID, "null reference arena");
    if (((uintptr_t)d_ref & 15u) != 0) return fail(ctx, RAWDTW_ERR_INVALID, "reference arena must be 16-byte aligned");
    if (ctx->own_ref && ctx->d_ref) (void)hipFree(ctx->d_ref);
    ctx->d_ref = const_cast<float *>(d_ref);
    ctx->n_ref = n_floats;
    ctx->own_ref = false;
    ctx->ref_off.clear();
    ctx->ref_len.clear();
    return RAWDTW_OK;
}

int rawdtw_upload_events(rawdtw_ctx *ctx, const float *h_events, uint64_t n_floats)
{
    if (!ctx || (!h_events && n_floats)) return fail(ctx, RAWDTW_ERR_INVALID, "null events");
    if (n_floats >= (1ull << 32)) return fail(ctx, RAWDTW_ERR_INVALID, "event arena limited to 2^32-1 floats per batch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->own_ev) { ctx->d_ev = nullptr; ctx->cap_ev = 0; ctx->own_ev = true; }
    int st = ensure_events_capacity(ctx, n_floats);
    if (st != RAWDTW_OK) return st;
    if (n_floats)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ev, h_events, n_floats * 4, hipMemcpyHostToDevice, ctx->stream));
    ctx->n_ev = n_floats;
    return RAWDTW_OK;
}

int rawdtw_set_events_device(rawdtw_ctx *ctx, const float *d_events, uint64_t n_floats)
{
    if (!ctx || (!d_events && n_floats)) return fail(ctx, RAWDTW_ERR_INVALID, "null event arena");
    if (((uintptr_t)d_events & 15u) != 0) return fail(ctx, RAWDTW_ERR_INVALID, "event arena must be 16-byte aligned");
    if (n_floats >= (1ull << 32)) return fail(ctx, RAWDTW_ERR_INVALID, "event arena limited to 2^32-1 floats per batch");
    if (ctx->own_ev && ctx->d_ev) (void)hipFree(ctx->d_ev);
    ctx->d_ev = const_cast<float *>(d_events);
    ctx->n_ev = n_floats;
    ctx->cap_ev = 0;
    ctx->own_ev = false;
    return RAWDTW_OK;
}

int rawdtw_plan_create(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, rawdtw_plan **out)
{
    if (!out) return RAWDTW_ERR_INVALID;
    if (ctx) { hipError_t e = hipSetDevice(ctx->device); if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice"); }
    return build_plan(ctx, jobs, n_jobs, false, out);
}

int rawdtw_plan_info(const rawdtw_plan *plan, rawdtw_plan_info_t *info)
{
    if (!plan || !info) return RAWDTW_ERR_INVALID;
    rawdtw_plan *pl = const_cast<rawdtw_plan *>(plan);
    if (!pl->cells_counted) {
        uint64_t cells = 0;
        for (const DevJob &d : pl->h_jobs)
            cells += d.R < 0 ? (uint64_t)d.n * d.m : banded_cells(d.n, d.m, d.R);
        pl->info.cells = cells;
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

int rawdtw_plan_destroy(rawdtw_plan *plan)
{
    if (!plan) return RAWDTW_OK;
    if (plan->ctx) (void)hipSetDevice(plan->ctx->device);
    if (plan->d_jobs) (void)hipFree(plan->d_jobs);
    if (plan->d_aux) (void)hipFree(plan->d_aux);
    if (plan->d_tiles) (void)hipFree(plan->d_tiles);
    if (plan->d_spans) (void)hipFree(plan->d_spans);
    if (plan->d_tjobs) (void)hipFree(plan->d_tjobs);
    if (plan->d_masks) (void)hipFree(plan->d_masks);
    if (plan->d_cost) (void)hipFree(plan->d_cost);
    if (plan->d_bnd) (void)hipFree(plan->d_bnd);
    if (plan->d_dir) (void)hipFree(plan->d_dir);
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

int rawdtw_traceback_batch(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, const float *h_events,
                           uint64_t n_events, float *out_cost, const uint64_t *path_off, uint32_t *path_len,
                           uint32_t *path_i, uint32_t *path_j, float *path_d)
{
    if (!ctx) return RAWDTW_ERR_INVALID;
    if (n_jobs && (!jobs || !out_cost || !path_off || !path_len || !path_i || !path_j || !path_d))
        return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    int st = rawdtw_upload_events(ctx, h_events, n_events);
    if (st != RAWDTW_OK) return st;

    // sub-batches bounded by the direction-buffer budget
    uint64_t budget = 16ull << 30;
    if (const char *e = getenv("RAWDTW_TB_WORKSPACE_MB")) budget = std::max<uint64_t>(1, strtoull(e, nullptr, 10)) << 20;
    uint64_t begin = 0;
    while (begin < n_jobs) {
        uint64_t end = begin, bytes = 0, path_elems = 0;
        while (end < n_jobs) {
            const rawdtw_job_t &j = jobs[end];
            if (j.n == 0 || j.m == 0) return fail(ctx, RAWDTW_ERR_INVALID, "zero-length traceback job");
            const uint64_t b = dir_bytes_for(j.n, j.m, full_rpl(std::min(j.n, j.m))) + 256;
            if (end > begin && bytes + b > budget) break;
            bytes += b;
            path_elems += (uint64_t)j.n + j.m - 1;
            end++;
        }
        const uint64_t cnt = end - begin;
        rawdtw_plan *pl = nullptr;
        st = build_plan(ctx, jobs + begin, cnt, true, &pl);
        if (st != RAWDTW_OK) return st;
        // device path buffers in plan order
        std::vector<uint64_t> h_poff(cnt);
        uint64_t acc = 0;
        for (uint64_t p = 0; p < cnt; p++) {
            h_poff[p] = acc;
            acc += (uint64_t)pl->h_jobs[p].n + pl->h_jobs[p].m - 1;
        }
        uint64_t *d_poff = nullptr;
        uint32_t *d_plen = nullptr, *d_pi = nullptr, *d_pj = nullptr;
        float *d_pd = nullptr;
        auto cleanup = [&]() {
            if (d_poff) (void)hipFree(d_poff);
            if (d_plen) (void)hipFree(d_plen);
            if (d_pi) (void)hipFree(d_pi);
            if (d_pj) (void)hipFree(d_pj);
            if (d_pd) (void)hipFree(d_pd);
            rawdtw_plan_destroy(pl);
        };
        if ((st = dev_alloc(ctx, &d_poff, cnt)) != RAWDTW_OK || (st = dev_alloc(ctx, &d_plen, cnt)) != RAWDTW_OK ||
            (st = dev_alloc(ctx, &d_pi, acc)) != RAWDTW_OK || (st = dev_alloc(ctx, &d_pj, acc)) != RAWDTW_OK ||
            (st = dev_alloc(ctx, &d_pd, acc)) != RAWDTW_OK) {
            cleanup();
            return st;
        }
        hipError_t e = hipMemcpyAsync(d_poff, h_poff.data(), cnt * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { cleanup(); return hip_fail(ctx, e, "path offsets upload"); }
        st = rawdtw_plan_run(ctx, pl);
        if (st != RAWDTW_OK) { cleanup(); return st; }
        for (const Launch &L : pl->launches) {
            e = launch_tb_walk(pl->d_jobs + L.first, L.count, pl->d_aux + L.first, L.param & 255, ctx->d_ev, ctx->d_ref,
                               pl->d_dir, d_poff + L.first, d_plen + L.first, d_pi, d_pj, d_pd, ctx->stream);
            if (e != hipSuccess) { cleanup(); return hip_fail(ctx, e, "traceback walk launch"); }
        }
        std::vector<float> h_cost(cnt), h_pd(acc);
        std::vector<uint32_t> h_plen(cnt), h_pi(acc), h_pj(acc);
        e = hipMemcpyAsync(h_cost.data(), pl->d_cost, cnt * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h_plen.data(), d_plen, cnt * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && acc) e = hipMemcpyAsync(h_pi.data(), d_pi, acc * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && acc) e = hipMemcpyAsync(h_pj.data(), d_pj, acc * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && acc) e = hipMemcpyAsync(h_pd.data(), d_pd, acc * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { cleanup(); return hip_fail(ctx, e, "traceback download"); }
        for (uint64_t p = 0; p < cnt; p++) {
            const uint64_t k = begin + pl->order[p];
            out_cost[k] = h_cost[pl->order[p]];
            const uint32_t len = h_plen[p];
            // device paths are end-first; the reference returns them start-first (dtw.cpp:656-657)
            // and pops the last element when exclude_last_element is set (dtw.cpp:659-663)
            const uint32_t outlen = jobs[k].exclude_last ? len - 1 : len;
            const uint64_t src = h_poff[p], dst = path_off[k];
            for (uint32_t q = 0; q < outlen; q++) {
                path_i[dst + q] = h_pi[src + len - 1 - q];
                path_j[dst + q] = h_pj[src + len - 1 - q];
                path_d[dst + q] = h_pd[src + len - 1 - q];
            }
            path_len[k] = outlen;
        }
        cleanup();
        begin = end;
    }
    return RAWDTW_OK;
}

// ---- single-call drop-ins ----------------------------------------------------------------------
static int single_call(rawdtw_ctx *ctx, const float *a, uint32_t n, const float *b, uint32_t m, int radius,
                       int excl, float *cost)
{
    if (!ctx || !a || !b || !cost) return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    if (n == 0 || m == 0 || radius < RAWDTW_FULL) return fail(ctx, RAWDTW_ERR_INVALID, "zero length or negative radius");
    // b goes to a private reference arena for the duration of the call
    const float *saved_ref = ctx->d_ref; uint64_t saved_n = ctx->n_ref; bool saved_own = ctx->own_ref;
    float *d_b = nullptr;
    int st = dev_alloc(ctx, &d_b, ((uint64_t)m + 3) & ~3ull);
    if (st != RAWDTW_OK) return st;
    hipError_t e = hipMemcpyAsync(d_b, b, (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d_b); return hip_fail(ctx, e, "operand upload"); }
    ctx->d_ref = d_b; ctx->n_ref = m; ctx->own_ref = false;
    rawdtw_job_t j{0, 0, n, m, radius, excl ? 1u : 0u, 0};
    st = rawdtw_score_batch(ctx, &j, 1, a, n, cost);
    ctx->d_ref = const_cast<float *>(saved_ref); ctx->n_ref = saved_n; ctx->own_ref = saved_own;
    (void)hipFree(d_b);
    return st;
}

int rawdtw_dtw_global(rawdtw_ctx *ctx, const float *a, uint32_t n, const float *b, uint32_t m, int exclude_last,
                      float *cost)
{
    return single_call(ctx, a, n, b, m, RAWDTW_FULL, exclude_last, cost);
}

int rawdtw_dtw_global_slantedbanded_antidiagonalwise(rawdtw_ctx *ctx, const float *a, uint32_t n, const float *b,
                                                     uint32_t m, int band_radius, int exclude_last, float *cost)
{
    if (band_radius < 0) return fail(ctx, RAWDTW_ERR_INVALID, "negative band radius (dtw.cpp:277 asserts)");
    return single_call(ctx, a, n, b, m, band_radius, exclude_last, cost);
}

int rawdtw_dtw_global_tb(rawdtw_ctx *ctx, const float *a, uint32_t n, const float *b, uint32_t m, int exclude_last,
                         float *cost, uint32_t *path_len, uint32_t *path_i, uint32_t *path_j, float *path_d)
{
    if (!ctx || !a || !b || !cost || !path_len || !path_i || !path_j || !path_d)
        return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    if (n == 0 || m == 0) return fail(ctx, RAWDTW_ERR_INVALID, "zero length (dtw.cpp:596 asserts)");
    const float *saved_ref = ctx->d_ref; uint64_t saved_n = ctx->n_ref; bool saved_own = ctx->own_ref;
    float *d_b = nullptr;
    int st = dev_alloc(ctx, &d_b, ((uint64_t)m + 3) & ~3ull);
    if (st != RAWDTW_OK) return st;
    hipError_t e = hipMemcpyAsync(d_b, b, (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d_b); return hip_fail(ctx, e, "operand upload"); }
    ctx->d_ref = d_b; ctx->n_ref = m; ctx->own_ref = false;
    rawdtw_job_t j{0, 0, n, m, RAWDTW_FULL, exclude_last ? 1u : 0u, 0};
    uint64_t off = 0;
    st = rawdtw_traceback_batch(ctx, &j, 1, a, n, cost, &off, path_len, path_i, path_j, path_d);
    ctx->d_ref = const_cast<float *>(saved_ref); ctx->n_ref = saved_n; ctx->own_ref = saved_own;
    (void)hipFree(d_b);
    return st;
}

// ---- whole-batch form ----------------------------------------------------------------------------
int rawdtw_batch_create(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                        const uint64_t *anchor_off, const rawdtw_anchor_t *anchors, const uint64_t *ref_base,
                        const uint32_t *read_base, rawdtw_batch **out)
{
    if (!out) return RAWDTW_ERR_INVALID;
    *out = nullptr;
    if (!ctx || !opt || !chain_off || !anchor_off || (!anchors && n_reads) || !ref_base || !read_base)
        return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    if (opt->border_constraint != 0 && opt->border_constraint != 1)
        return fail(ctx, RAWDTW_ERR_INVALID, "invalid border constraint (rmap.cpp:301-304)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n_chains = chain_off[n_reads];
    std::vector<uint64_t> job_off(n_chains + 1);
    uint64_t n_jobs = 0;
    int st = rawdtw_batch_build_jobs(opt, n_chains, anchor_off, anchors, ref_base, read_base, job_off.data(), nullptr,
                                     0, &n_jobs);
    if (st != RAWDTW_OK) return fail(ctx, st, "job counting failed");
    std::vector<rawdtw_job_t> jobs(n_jobs);
    st = rawdtw_batch_build_jobs(opt, n_chains, anchor_off, anchors, ref_base, read_base, job_off.data(), jobs.data(),
                                 n_jobs, &n_jobs);
    if (st != RAWDTW_OK) return fail(ctx, st, "job building failed");
    std::vector<ChainDesc> desc(n_chains);
    for (uint64_t c = 0; c < n_chains; c++) {
        const uint64_t a0 = anchor_off[c], a1 = anchor_off[c + 1];
        ChainDesc &d = desc[c];
        d.job_first = job_off[c];
        d.n_jobs = (uint32_t)(job_off[c + 1] - job_off[c]);
        d.reserved = 0;
        if (a1 == a0) { d.span = 0; d.num_aligned = 0; continue; }
        const rawdtw_anchor_t &first = anchors[a1 - 1], &last = anchors[a0];
        d.span = last.query_position - first.query_position + 1; // rmap.cpp:202,245
        uint32_t na = 0;
        for (uint64_t k = job_off[c]; k < job_off[c + 1]; k++) na += jobs[k].n; // rmap.cpp:236,292
        d.num_aligned = na;
    }
    rawdtw_batch *b = new (std::nothrow) rawdtw_batch;
    if (!b) return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed");
    b->ctx = ctx; b->opt = *opt; b->n_reads = n_reads; b->n_chains = n_chains;
    st = build_plan(ctx, jobs.data(), n_jobs, false, &b->plan);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_chains, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_chain_off, n_reads + 1);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_fold_order, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_full, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_gate, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_score, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_keep, n_chains);
    if (st == RAWDTW_OK) {
        hipError_t e = hipSuccess;
        std::vector<uint32_t> fold_order(n_chains);
        for (uint64_t c = 0; c < n_chains; c++) fold_order[c] = (uint32_t)c;
        std::stable_sort(fold_order.begin(), fold_order.end(),
                         [&](uint32_t x, uint32_t y) { return desc[x].n_jobs > desc[y].n_jobs; });
        if (n_chains) e = hipMemcpyAsync(b->d_chains, desc.data(), n_chains * sizeof(ChainDesc), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && n_chains)
            e = hipMemcpyAsync(b->d_fold_order, fold_order.data(), n_chains * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(b->d_chain_off, chain_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = hip_fail(ctx, e, "uploading chain descriptors");
    }
    if (st != RAWDTW_OK) { rawdtw_batch_destroy(b); return st; }
    *out = b;
    return RAWDTW_OK;
}

int rawdtw_batch_info(const rawdtw_batch *batch, rawdtw_plan_info_t *info, uint64_t *n_chains)
{
    if (!batch) return RAWDTW_ERR_INVALID;
    if (n_chains) *n_chains = batch->n_chains;
    if (info) return rawdtw_plan_info(batch->plan, info);
    return RAWDTW_OK;
}

static int batch_tail(rawdtw_ctx *ctx, rawdtw_batch *b, int which)
{
    hipError_t e;
    if (which == 0)
        e = launch_chain_fold(b->d_chains, b->d_fold_order, b->n_chains, b->plan->d_cost, b->opt.match_bonus, b->opt.fused_score,
                              b->d_full, b->d_gate, ctx->stream);
    else
        e = launch_read_select(b->d_chain_off, b->n_reads, b->d_full, b->d_gate, b->opt.min_score, b->d_score,
                               b->d_keep, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, which == 0 ? "chain fold launch" : "read select launch");
    return RAWDTW_OK;
}

int rawdtw_batch_run(rawdtw_ctx *ctx, rawdtw_batch *batch)
{
    if (!ctx || !batch || batch->ctx != ctx) return fail(ctx, RAWDTW_ERR_INVALID, "batch does not belong to this context");
    int st = rawdtw_plan_run(ctx, batch->plan);
    if (st == RAWDTW_OK) st = batch_tail(ctx, batch, 0);
    if (st == RAWDTW_OK) st = batch_tail(ctx, batch, 1);
    return st;
}

int rawdtw_batch_run_timed(rawdtw_ctx *ctx, rawdtw_batch *batch, float *launch_ms, uint32_t *launch_kind, uint32_t cap,
                           uint32_t *n_launches)
{
    std::vector<float> tmp(64, 0.f);
    int st = rawdtw_batch_run_reps(ctx, batch, 1, launch_ms ? launch_ms : tmp.data(), launch_kind,
                                   launch_ms ? cap : 64, n_launches);
    return st;
}

static int batch_enqueue_one(rawdtw_ctx *ctx, rawdtw_batch *batch, hipEvent_t *e)
{
    rawdtw_plan *pl = batch->plan;
    const uint32_t np = (uint32_t)pl->launches.size();
    int st = run_all_launches(ctx, pl, e);
    for (int k = 0; k < 2 && st == RAWDTW_OK; k++) {
        if (e && hipEventRecord(e[2 * (np + k)], ctx->stream) != hipSuccess) st = RAWDTW_ERR_DEVICE;
        if (st == RAWDTW_OK) st = batch_tail(ctx, batch, k);
        if (st == RAWDTW_OK && e && hipEventRecord(e[2 * (np + k) + 1], ctx->stream) != hipSuccess) st = RAWDTW_ERR_DEVICE;
    }
    return st;
}

int rawdtw_batch_enqueue(rawdtw_ctx *ctx, rawdtw_batch *batch, int timed)
{
    if (!ctx || !batch || batch->ctx != ctx) return fail(ctx, RAWDTW_ERR_INVALID, "batch does not belong to this context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t nl = (uint32_t)batch->plan->launches.size() + 2;
    hipEvent_t *e = nullptr;
    if (timed) {
        const size_t base = batch->ev.size();
        batch->ev.resize(base + 2 * nl, nullptr);
        for (size_t k = base; k < batch->ev.size(); k++) HIP_TRY(ctx, hipEventCreate(&batch->ev[k]));
        e = &batch->ev[base];
        batch->ev_runs++;
    }
    return batch_enqueue_one(ctx, batch, e);
}

int rawdtw_batch_collect(rawdtw_ctx *ctx, rawdtw_batch *batch, float *launch_ms, uint32_t *launch_kind, uint32_t cap,
                         uint32_t *n_launches, uint32_t *n_runs)
{
    if (!ctx || !batch || batch->ctx != ctx) return fail(ctx, RAWDTW_ERR_INVALID, "batch does not belong to this context");
    rawdtw_plan *pl = batch->plan;
    const uint32_t np = (uint32_t)pl->launches.size(), nl = np + 2;
    if (n_launches) *n_launches = nl;
    if (n_runs) *n_runs = batch->ev_runs;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int st = RAWDTW_OK;
    for (uint32_t i = 0; i < nl && i < cap; i++) {
        double acc = 0;
        for (uint32_t r = 0; r < batch->ev_runs; r++) {
            float ms = 0.f;
            const size_t b = (size_t)r * 2 * nl + 2 * i;
            if (hipEventElapsedTime(&ms, batch->ev[b], batch->ev[b + 1]) != hipSuccess) st = RAWDTW_ERR_DEVICE;
            acc += ms;
        }
        if (launch_ms) launch_ms[i] = batch->ev_runs ? (float)(acc / batch->ev_runs) : 0.f;
        if (launch_kind)
            launch_kind[i] = i < np ? (pl->launches[i].kind | ((uint32_t)pl->launches[i].param << 8))
                                    : (i == np ? kKindChainFold : kKindReadSelect);
    }
    for (auto &e : batch->ev) if (e) (void)hipEventDestroy(e);
    batch->ev.clear();
    batch->ev_runs = 0;
    if (st != RAWDTW_OK && ctx->err.empty()) ctx->err = "collect failed";
    return st;
}

int rawdtw_batch_run_reps(rawdtw_ctx *ctx, rawdtw_batch *batch, uint32_t reps, float *launch_ms, uint32_t *launch_kind,
                          uint32_t cap, uint32_t *n_launches)
{
    if (!ctx || !batch || batch->ctx != ctx) return fail(ctx, RAWDTW_ERR_INVALID, "batch does not belong to this context");
    if (n_launches) *n_launches = (uint32_t)batch->plan->launches.size() + 2;
    int st = RAWDTW_OK;
    for (uint32_t r = 0; r < reps && st == RAWDTW_OK; r++) st = rawdtw_batch_enqueue(ctx, batch, launch_ms != nullptr);
    if (launch_ms) {
        int st2 = rawdtw_batch_collect(ctx, batch, launch_ms, launch_kind, cap, nullptr, nullptr);
        if (st == RAWDTW_OK) st = st2;
    } else if (hipStreamSynchronize(ctx->stream) != hipSuccess && st == RAWDTW_OK) st = RAWDTW_ERR_DEVICE;
    return st;
}

int rawdtw_batch_launch_stats(const rawdtw_batch *batch, uint32_t i, uint32_t *kind, int32_t *param, uint64_t *n_jobs,
                              uint64_t *algorithmic_bytes, uint64_t *cells)
{
    if (!batch) return RAWDTW_ERR_INVALID;
    const rawdtw_plan *pl = batch->plan;
    const uint32_t nl = (uint32_t)pl->launches.size();
    if (i >= nl + 2) return RAWDTW_ERR_INVALID;
    if (i >= nl) {
        if (kind) *kind = i == nl ? kKindChainFold : kKindReadSelect;
        if (param) *param = 0;
        if (n_jobs) *n_jobs = i == nl ? batch->n_chains : batch->n_reads;
        // fold: one 4-byte cost per job + a 24-byte descriptor and two 4-byte results per chain;
        // select: 8 bytes read and 5 written per chain
        if (algorithmic_bytes)
            *algorithmic_bytes = i == nl ? pl->n_jobs * 4 + batch->n_chains * 32 : batch->n_chains * 13 + batch->n_reads * 8;
        if (cells) *cells = 0;
        return RAWDTW_OK;
    }
    const Launch &L = pl->launches[i];
    uint64_t bytes = 0, cl = 0;
    for (uint64_t p = L.first; p < L.first + L.count; p++) {
        const DevJob &d = pl->h_jobs[p];
        bytes += 4ull * ((uint64_t)d.n + d.m) + 4 + 32;
        if (cells) cl += d.R < 0 ? (uint64_t)d.n * d.m : banded_cells(d.n, d.m, d.R);
    }
    if (kind) *kind = L.kind;
    if (param) *param = L.param;
    if (n_jobs) *n_jobs = L.count;
    if (algorithmic_bytes) *algorithmic_bytes = bytes;
    if (cells) *cells = cl;
    return RAWDTW_OK;
}

int rawdtw_batch_fetch(rawdtw_ctx *ctx, rawdtw_batch *batch, float *score, uint8_t *keep, float *job_cost)
{
    if (!ctx || !batch || batch->ctx != ctx) return fail(ctx, RAWDTW_ERR_INVALID, "batch does not belong to this context");
    if (batch->n_chains) {
        if (score) HIP_TRY(ctx, hipMemcpyAsync(score, batch->d_score, batch->n_chains * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (keep) HIP_TRY(ctx, hipMemcpyAsync(keep, batch->d_keep, batch->n_chains, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (job_cost && batch->plan->n_jobs)
        HIP_TRY(ctx, hipMemcpyAsync(job_cost, batch->plan->d_cost, batch->plan->n_jobs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RAWDTW_OK;
}

int rawdtw_batch_destroy(rawdtw_batch *b)
{
    if (!b) return RAWDTW_OK;
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    rawdtw_plan_destroy(b->plan);
    if (b->d_chains) (void)hipFree(b->d_chains);
    if (b->d_chain_off) (void)hipFree(b->d_chain_off);
    if (b->d_fold_order) (void)hipFree(b->d_fold_order);
    if (b->d_full) (void)hipFree(b->d_full);
    if (b->d_gate) (void)hipFree(b->d_gate);
    if (b->d_score) (void)hipFree(b->d_score);
    if (b->d_keep) (void)hipFree(b->d_keep);
    delete b;
    return RAWDTW_OK;
}

// ---- index reader --------------------------------------------------------------------------------
int rawdtw_index_open(const char *path, rawdtw_index **out)
{
    if (!path || !out) return RAWDTW_ERR_INVALID;
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return RAWDTW_ERR_INVALID;
    rawdtw_index *ix = new (std::nothrow) rawdtw_index;
    if (!ix) { fclose(f); return RAWDTW_ERR_OOM; }
    ix->path = path;
    char magic[2];
    bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'R' && magic[1] == 'I'; // rawindex.h:7-8 RI_IDX_MAGIC, 2 bytes
    ok = ok && fread(ix->pars, 4, 8, f) == 8;
    const uint32_t n_seq = ok ? ix->pars[6] : 0;
    for (uint32_t i = 0; ok && i < n_seq; i++) {
        uint8_t l = 0;
        ok = fread(&l, 1, 1, f) == 1;
        std::string name(l, '\0');
        if (ok && l) ok = fread(&name[0], 1, l, f) == l;
        uint32_t len = 0;
        ok = ok && fread(&len, 4, 1, f) == 1;
        if (!ok) break;
        ix->names.push_back(name);
        ix->lens.push_back(len);
        ix->fwd_pos.push_back((uint64_t)ftello(f));
        ok = fseeko(f, (off_t)len * 8, SEEK_CUR) == 0; // skip forward + reverse arrays
    }
    fclose(f);
    if (!ok) { delete ix; return RAWDTW_ERR_INVALID; }
    *out = ix;
    return RAWDTW_OK;
}

int rawdtw_index_info(const rawdtw_index *idx, uint32_t *n_seq, uint32_t pars[8])
{
    if (!idx) return RAWDTW_ERR_INVALID;
    if (n_seq) *n_seq = (uint32_t)idx->lens.size();
    if (pars) memcpy(pars, idx->pars, sizeof(idx->pars));
    return RAWDTW_OK;
}

int rawdtw_index_seq(const rawdtw_index *idx, uint32_t i, const char **name, uint32_t *len)
{
    if (!idx || i >= idx->lens.size()) return RAWDTW_ERR_INVALID;
    if (name) *name = idx->names[i].c_str();
    if (len) *len = idx->lens[i];
    return RAWDTW_OK;
}

int rawdtw_index_read_signal(const rawdtw_index *idx, uint32_t i, int strand, float *out)
{
    if (!idx || i >= idx->lens.size() || !out) return RAWDTW_ERR_INVALID;
    FILE *f = fopen(idx->path.c_str(), "rb");
    if (!f) return RAWDTW_ERR_INVALID;
    // file order: forward_signals[i] then reverse_signals[i]; strand==1 selects forward (rmap.cpp:182-188)
    const uint64_t pos = idx->fwd_pos[i] + (strand == 1 ? 0 : (uint64_t)idx->lens[i] * 4);
    bool ok = fseeko(f, (off_t)pos, SEEK_SET) == 0 && fread(out, 4, idx->lens[i], f) == idx->lens[i];
    fclose(f);
    return ok ? RAWDTW_OK : RAWDTW_ERR_INVALID;
}

int rawdtw_index_upload(rawdtw_ctx *ctx, const rawdtw_index *idx)
{
    if (!ctx || !idx) return RAWDTW_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n_seq = (uint32_t)idx->lens.size();
    if (ctx->own_ref && ctx->d_ref) (void)hipFree(ctx->d_ref);
    ctx->d_ref = nullptr; ctx->own_ref = true; ctx->n_ref = 0;
    ctx->ref_off.assign(2ull * n_seq, 0);
    ctx->ref_len = idx->lens;
    uint64_t total = 0;
    for (uint32_t s = 0; s < n_seq; s++) {
        ctx->ref_off[2 * s] = total; total += ((uint64_t)idx->lens[s] + 3) & ~3ull;
        ctx->ref_off[2 * s + 1] = total; total += ((uint64_t)idx->lens[s] + 3) & ~3ull;
    }
    int st = dev_alloc(ctx, &ctx->d_ref, std::max<uint64_t>(total, 4));
    if (st != RAWDTW_OK) return st;
    ctx->n_ref = total;
    FILE *f = fopen(idx->path.c_str(), "rb");
    if (!f) return fail(ctx, RAWDTW_ERR_INVALID, "cannot reopen index file");
    // stream through two pinned staging buffers so that the file read overlaps the H2D copy
    const size_t CH = 16u << 20; // floats per staging buffer (64 MiB)
    float *stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipError_t e = hipHostMalloc((void **)&stage[0], CH * 4, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&stage[1], CH * 4, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreate(&done[0]);
    if (e == hipSuccess) e = hipEventCreate(&done[1]);
    bool ok = e == hipSuccess;
    int which = 0;
    bool used[2] = {false, false};
    for (uint32_t s = 0; ok && s < n_seq; s++) {
        ok = fseeko(f, (off_t)idx->fwd_pos[s], SEEK_SET) == 0;
        for (int strand_slot = 0; ok && strand_slot < 2; strand_slot++) {
            uint64_t left = idx->lens[s], at = ctx->ref_off[2 * s + strand_slot];
            while (ok && left) {
                const size_t take = (size_t)std::min<uint64_t>(left, CH);
                if (used[which]) ok = hipEventSynchronize(done[which]) == hipSuccess;
                ok = ok && fread(stage[which], 4, take, f) == take;
                ok = ok && hipMemcpyAsync(ctx->d_ref + at, stage[which], take * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
                ok = ok && hipEventRecord(done[which], ctx->stream) == hipSuccess;
                used[which] = true;
                which ^= 1; left -= take; at += take;
            }
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) ok = false;
    fclose(f);
    for (int k = 0; k < 2; k++) { if (stage[k]) (void)hipHostFree(stage[k]); if (done[k]) (void)hipEventDestroy(done[k]); }
    if (!ok) return fail(ctx, RAWDTW_ERR_DEVICE, "index upload failed (short file or HIP error)");
    return RAWDTW_OK;
}

int rawdtw_index_close(rawdtw_index *idx)
{
    delete idx;
    return RAWDTW_OK;
}

} // extern "C"
