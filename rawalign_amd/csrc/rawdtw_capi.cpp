// rawdtw_capi.cpp -- the C ABI of librawdtw.so (include/rawdtw.h): context, arenas, the
// batch planner and the launch sequences.  Host code only; kernels live in rawdtw_kernels.hip.
//
// There is NO CPU fallback in here: every scoring entry point runs the HIP kernels or
// returns an error status.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rawdtw_internal.h"

using namespace rawdtw;

// Array of trivially-copyable records whose resize leaves the elements uninitialised: the planner's
// large outputs are written once, in parallel, and a value-initialising resize would first sweep them
// on one thread (page faults included).
template <typename T> struct RawVec {
    T *p = nullptr;
    size_t n = 0;
    RawVec() = default;
    RawVec(const RawVec &) = delete;
    RawVec &operator=(const RawVec &) = delete;
    ~RawVec() { free(p); }
    void resize(size_t count)
    {
        free(p);
        p = count ? static_cast<T *>(malloc(count * sizeof(T))) : nullptr;
        if (count && !p) { n = 0; throw std::bad_alloc(); }
        n = count;
    }
    size_t size() const { return n; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    const T *begin() const { return p; }
    const T *end() const { return p + n; }
};

// one pooled workspace of the stream path: a device block and a pinned host block (rawdtw_batch_create carves them up)
struct StreamWs {
    char *d = nullptr; size_t d_bytes = 0;
    char *h = nullptr; size_t h_bytes = 0;
};

// a reference arena the library allocated, alive while any context uses it
struct RefHold {
    float *d = nullptr;
    std::atomic<int> refs{1};
};

struct rawdtw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // side streams: independent launches of one batch run concurrently (fork/join around the main stream)
    static constexpr int kSide = 3;
    hipStream_t side[kSide] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kSide] = {nullptr, nullptr, nullptr};
    hipStream_t wide = nullptr;                 // sync-free batches: the side list's launch runs here, beside the tiles' launch
    hipEvent_t ev_wide_fork = nullptr, ev_wide_join = nullptr;
    bool serial_launches = false;
    int n_side = 0; // side streams used to fork the launches of one batch (RAWDTW_SIDE_STREAMS, 0..kSide). 0: the
                    // launches of a batch run in sequence on its one stream and overlap comes from several batches in
                    // flight on several contexts (swept: best throughput and cleaner per-kernel timings)
    uint32_t lane_hi_max_n = 96;
    int micro_max_n = 8; // shapes with longer side <= this use the micro paths (0: none, 4: micro4 only)
    bool grp16 = true; // bands of at most 16 offsets: four jobs per wave (else one job per wave)
    bool grp8 = true;  // ... and of at most 8 offsets: eight jobs per wave
    bool full_wg = true; // full-matrix jobs with >= 3 strips: four waves per job, pipelined strips
    bool lane_hi = false; // radii 4..8 on the second tile-kernel instance (else on k_band_wreg<1>)
    uint32_t tile_lds_floats = kTileLdsFloats, tile_max_jobs = kTileMaxJobs;
    bool tile_lds_set = false;                  // "tile_lds_floats" was given: it also sizes the device-planned batches' tiles
    uint32_t lane_max_n = kLaneMaxN;
    uint32_t debug_skip_kinds = 0; // timing experiments: launches of these kinds are not issued (results are then wrong)
    uint32_t debug_skip_tail = 0;  // timing experiments on the sync-free path: 1 no fold launch, 2 no select launch (results are then wrong)
    uint32_t sort_n = 0, sort_r1_n = 0, sort_r3 = 0, sorted_tile_jobs = 64; // see PlanCfg
    bool device_plan = true;  // rawdtw_batch_create takes the sync-free stream path (rawdtw_stream.hip) for sparse + banded batches
    uint64_t device_plan_min_jobs = 0; // smaller batches go through the job list
    std::vector<StreamWs> ws_free;     // workspaces of destroyed batches, reused by the next ones (no hipMalloc in the steady state)
    uint32_t stream_lds = 0, stream_blocks = 0; // persistent grid of k_stream at the current tile size
    int stream_threads = 256;                   // workgroup size of k_runs (256 or 512)
    uint32_t wide_blocks = 256;                 // workgroups (four waves each) of the side list's launch
    int pass_pool = -1;                         // copy-order slots beyond one a tile (tests: a batch that runs out is redone through the job list); -1: 3 a tile + 64
    int wide_at_create = 0;                     // 1: also after a plain rawdtw_batch_create (the caller leaves the arenas alone until the run)
    bool in_submit = false;                     // inside rawdtw_batch_submit*: create and run are one call
    int wide_order = 0;                         // 0: k_wide between the scan and the pass planning (first run), 1: in front of k_runs, 2: behind it
    int wide_beside = 0;                        // 1: that launch on the context's second stream, beside the tiles' launch; 0: in line (measured:
                                                // the fork and join cost the fresh-batch pipeline 8 % and the PCIe loop 17 %)
    int stream_threads_cached = 0;
    int stream_blocks_per_cu = 4;               // 0: what the occupancy query gives; else at most this many (leaves room for other streams' kernels)
    int stream_bpc_cached = -1;
    uint32_t stream_debug = 0;         // StreamArgs::debug
    bool resident_arrays = false;      // rawdtw_batch_create: anchors / ref_base / read_base are DEVICE pointers (used in place)
    bool time_plan = false;            // record an event pair around a batch's planning kernels (rawdtw_batch_plan_ms)
    std::vector<uint64_t> job_off_scratch;
    void *d_append = nullptr;          // rawdtw_events_append staging, grow-only
    size_t append_bytes = 0;
    uint8_t *d_tb_dir = nullptr; // traceback direction workspace, grow-only (hipFree of 600 MB per call costs 1 ms)
    uint64_t tb_dir_bytes = 0;
    void *d_tb_paths = nullptr;  // traceback path buffers (offsets, lengths, i/j end-first, i/j/d start-first), grow-only
    size_t tb_paths_bytes = 0;
    void *h_pinned = nullptr;  // pinned host staging (traceback paths), grow-only
    size_t pinned_bytes = 0;
    hipEvent_t tb_ev[3] = {nullptr, nullptr, nullptr}; // (kept for ABI of the struct's users; a traceback sub-batch has events of its own)
    hipStream_t tb_copy = nullptr;                      // traceback: the paths' way home, beside the next sub-batch's kernels
    float tb_fill_ms = 0.f, tb_walk_ms = 0.f;          // device time of the most recent rawdtw_traceback_batch
    uint64_t tb_dir_written = 0, tb_path_elems = 0;
    bool merge_small = true; // tile + 16-lane-row + register-wave launches of a batch as one launch (k_band_merged)
    int fold_mode = 4; // 0: wave per chain, 1/2: lane per chain (16/32 parts per round; 30x less VALU work), 3: lanes + a wave for each long chain,
                       // 4: sync-free batches fold and select in one launch out of LDS (k_fold_select), job-list batches as 3
    uint32_t fold_long_parts = 768; // fold_mode 3: chains of at least this many parts are folded a wave each
    int tile_threads = 256; // workgroup size of the tile kernel (256, 512, 1024)
    uint32_t tile_max_spans = kTileMaxSpans;
    int plan_threads = 0; // planner threads (0: from the job count and the machine, at most 16)
    int lane_max_radius = kMaxLaneRadius; // radii above this go to the register-resident wave kernel (RAWDTW_LANE_MAX_R)
    int stream_tile_radius = 3;           // device-planned batches: the tiles' radius limit ("stream_tile_radius")
    // reference arena.  An arena the library allocated (rawdtw_upload_reference, rawdtw_index_upload) is held through a
    // counted RefHold, shared by every context that adopted it with rawdtw_share_reference: it is freed when the last of
    // them lets go, so the owner may upload another reference or be destroyed while sharers still run on the old one.
    float *d_ref = nullptr;
    uint64_t n_ref = 0;
    struct RefHold *ref_hold = nullptr; // null: no arena, or the caller's own device memory (rawdtw_set_reference_device)
    // plans and batches created on this context and not destroyed yet: rawdtw_destroy detaches them (frees their device
    // memory, clears their back pointer), after which rawdtw_plan_destroy / rawdtw_batch_destroy only delete the host record
    std::vector<rawdtw_plan *> live_plans;
    std::vector<rawdtw_batch *> live_batches;
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
    RawVec<uint32_t> order;        // plan position -> job index
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
    RawVec<DevJob> h_jobs;         // plan order (kept for traceback + info)
    std::vector<FullAux> h_aux;    // of the non-tile jobs: index = plan position - n_tile_jobs
    rawdtw_plan_info_t info{};
    bool cells_counted = false;
    int plan_threads_used = 1;
    bool dir_borrowed = false; // d_dir is the context's workspace, not the plan's
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
    rawdtw_plan *plan = nullptr;   // job-list path
    rawdtw_align_opt_t opt{};
    uint64_t n_reads = 0, n_chains = 0, n_jobs = 0; // (n_jobs of a sync-free batch: counted on first use, see batch_count_jobs)
    bool jobs_counted = false;
    ChainDesc *d_chains = nullptr;
    uint64_t *d_chain_off = nullptr;
    uint32_t *d_fold_order = nullptr; // chain ids, longest chain first
    bool fold_fused = false;          // sync-free batch: fold and select are one launch (k_fold_select), no fold order was built
    float *d_full = nullptr, *d_gate = nullptr, *d_score = nullptr;
    uint8_t *d_keep = nullptr;
    bool own_chain_arrays = false;  // the arrays above are hipMalloc'd (job-list path) rather than carved from `ws`
    std::vector<hipEvent_t> ev; // event pairs of the runs enqueued since the last collect
    uint32_t ev_runs = 0;
    // stream path
    bool stream = false;
    StreamWs ws;
    StreamArgs sa{};
    uint32_t stream_lds = 0;
    int stream_threads = 256;
    unsigned long long *h_cnt = nullptr; // pinned landing zone of the counter block ...
    float *h_score = nullptr;            // ... and, behind it at the device block's offsets, of the scores and the keep flags: counters, scores
    uint8_t *h_keep = nullptr;           // and flags lie one behind the other in the workspace and come home in ONE copy (rawdtw_batch_fetch:
    size_t res_bytes = 0;                // every operation on a batch's stream is a step of its latency through the pipeline: three copies -> one, + 2 %)
    bool cnt_valid = false, cells_counted = false;
    uint32_t stream_runs = 0;                // DTW launches issued for this batch (the tile queue needs a reset from the second on)
    bool dirty = false;                  // work enqueued since the last host synchronisation
    size_t ws_bytes = 0;
    hipEvent_t ev_plan[4] = {nullptr, nullptr, nullptr, nullptr}; // ("time_plan") around scan + side list order, the side list's launch, the pass planning
    bool wide_out = false;            // the side list's launch for the next run went out with the planning launches
    // the caller's arrays (valid until fetch: a declined batch is redone from them through the job list)
    const uint64_t *in_chain_off = nullptr, *in_anchor_off = nullptr;
    const rawdtw_anchor_t *in_anchors = nullptr;
    const uint64_t *in_ref_base = nullptr;
    const uint32_t *in_read_base = nullptr;
    // compact hand-over (rawdtw_batch_submit_compact): the lists in packed form instead of in_anchors
    const rawdtw_anchor_t *in_heads = nullptr, *in_unit_abs = nullptr;
    const uint16_t *in_steps = nullptr;
    const rawdtw_wide_step_t *in_wide = nullptr;
    uint64_t in_n_wide = 0;
    // chunk rounds (rawdtw_batch_submit_carry): the batch of the round before, the per-chain carry records and the round's SHORT
    // lists (new entries + junction); in_anchor_off / in_anchors stay the FULL lists' (in_anchors: the fallback's, may be null)
    const rawdtw_batch *in_prev = nullptr;
    const rawdtw_carry_t *in_carry = nullptr;
    const uint64_t *in_new_off = nullptr;
    const rawdtw_anchor_t *in_new_anchors = nullptr;
    bool in_carried = false;             // a chunk round: the device works on the short lists
    uint64_t parts_carried = 0;          // (summed from the carry records at create)
    // "resident_arrays": the three big arrays are device pointers; host copies are made only if the job list is needed
    bool in_resident = false;
    std::vector<rawdtw_anchor_t> host_anchors;
    std::vector<uint64_t> host_ref_base;
    std::vector<uint32_t> host_read_base;
};

namespace {

// the next `count` elements of a 256-byte aligned block
template <typename T> T *carve(char *&p, uint64_t count)
{
    T *q = reinterpret_cast<T *>(p);
    p += (count * sizeof(T) + 255) & ~(size_t)255;
    return q;
}

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

// let go of the context's reference arena (the allocation dies with its last user)
void drop_reference(rawdtw_ctx *ctx)
{
    if (RefHold *h = ctx->ref_hold) {
        if (h->refs.fetch_sub(1) == 1) { if (h->d) (void)hipFree(h->d); delete h; }
    }
    ctx->ref_hold = nullptr; ctx->d_ref = nullptr; ctx->n_ref = 0;
}

template <typename T> void unregister(std::vector<T *> &v, T *x)
{
    for (size_t i = 0; i < v.size(); i++)
        if (v[i] == x) { v[i] = v.back(); v.pop_back(); return; }
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

// run fn(t) for t in [0, T) on T threads (the caller's thread takes t = 0)
template <typename F> void parallel_for(int T, F fn)
{
    if (T <= 1) { fn(0); return; }
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (int t = 1; t < T; t++) th.emplace_back([&fn, t] { fn(t); });
    fn(0);
    for (auto &x : th) x.join();
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
struct MergeSel { int tile = -1, grp16 = -1, grp8 = -1, wreg = -1; bool on() const { return tile >= 0; } };

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
    bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->wide, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->ev_wide_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->ev_wide_join, hipEventDisableTiming) == hipSuccess;
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
    if (const char *e = getenv("RAWDTW_OPTS")) { // "name=value,name=value": rawdtw_set_option for each (tuning runs)
        std::string all(e);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(pos, end - pos);
            const size_t eq = item.find('=');
            if (eq != std::string::npos) (void)rawdtw_set_option(ctx, item.substr(0, eq).c_str(), atoll(item.c_str() + eq + 1));
            pos = end + 1;
        }
    }
    *out = ctx;
    return RAWDTW_OK;
}

static void batch_detach(rawdtw_ctx *ctx, rawdtw_batch *b);
static void plan_release_device(rawdtw_plan *plan);

int rawdtw_destroy(rawdtw_ctx *ctx)
{
    if (!ctx) return RAWDTW_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
    // Plans and batches that outlive their context (a caller tearing down in the "wrong" order): their device memory and
    // pooled workspaces go now, their back pointers are cleared, and the later *_destroy calls only delete host records.
    while (!ctx->live_batches.empty()) {
        rawdtw_batch *b = ctx->live_batches.back();
        ctx->live_batches.pop_back();
        batch_detach(ctx, b);
        b->ctx = nullptr;
    }
    while (!ctx->live_plans.empty()) {
        rawdtw_plan *pl = ctx->live_plans.back();
        ctx->live_plans.pop_back();
        plan_release_device(pl);
        pl->ctx = nullptr;
    }
    for (int k = 0; k < rawdtw_ctx::kSide; k++) {
        if (ctx->side[k]) { (void)hipStreamSynchronize(ctx->side[k]); (void)hipStreamDestroy(ctx->side[k]); }
        if (ctx->ev_join[k]) (void)hipEventDestroy(ctx->ev_join[k]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->wide) { (void)hipStreamSynchronize(ctx->wide); (void)hipStreamDestroy(ctx->wide); }
    if (ctx->ev_wide_fork) (void)hipEventDestroy(ctx->ev_wide_fork);
    if (ctx->ev_wide_join) (void)hipEventDestroy(ctx->ev_wide_join);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    drop_reference(ctx);
    if (ctx->own_ev && ctx->d_ev) (void)hipFree(ctx->d_ev);
    for (StreamWs &w : ctx->ws_free) { if (w.d) (void)hipFree(w.d); if (w.h) (void)hipHostFree(w.h); }
    if (ctx->d_append) (void)hipFree(ctx->d_append);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    for (hipEvent_t &e : ctx->tb_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->tb_copy) (void)hipStreamDestroy(ctx->tb_copy);
    if (ctx->d_tb_dir) (void)hipFree(ctx->d_tb_dir);
    if (ctx->d_tb_paths) (void)hipFree(ctx->d_tb_paths);
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
    if (!strcmp(name, "tile_lds_floats")) { ctx->tile_lds_floats = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1024), 40000); ctx->tile_lds_set = true; return RAWDTW_OK; }
    if (!strcmp(name, "tile_max_jobs")) { ctx->tile_max_jobs = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 64), 65535); return RAWDTW_OK; }
    if (!strcmp(name, "plan_threads")) { ctx->plan_threads = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 64); return RAWDTW_OK; }
    if (!strcmp(name, "debug_skip_kinds")) { ctx->debug_skip_kinds = (uint32_t)value; return RAWDTW_OK; }
    if (!strcmp(name, "pass_pool")) { ctx->pass_pool = (int)std::min<int64_t>(std::max<int64_t>(value, -1), 1 << 24); return RAWDTW_OK; }
    if (!strcmp(name, "wide_at_create")) { ctx->wide_at_create = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "wide_order")) { ctx->wide_order = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 2); return RAWDTW_OK; }
    if (!strcmp(name, "wide_beside")) { ctx->wide_beside = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "wide_blocks")) { ctx->wide_blocks = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 65535); return RAWDTW_OK; }
    if (!strcmp(name, "debug_skip_tail")) { ctx->debug_skip_tail = (uint32_t)value; return RAWDTW_OK; }
    if (!strcmp(name, "sort_n")) { ctx->sort_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 255); return RAWDTW_OK; }
    if (!strcmp(name, "sort_r1_n")) { ctx->sort_r1_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 255); return RAWDTW_OK; }
    if (!strcmp(name, "sort_r3")) { ctx->sort_r3 = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "sorted_tile_jobs")) { ctx->sorted_tile_jobs = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 16), 1024); return RAWDTW_OK; }
    if (!strcmp(name, "device_plan")) { ctx->device_plan = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "device_plan_min_jobs")) { ctx->device_plan_min_jobs = (uint64_t)std::max<int64_t>(value, 0); return RAWDTW_OK; }
    if (!strcmp(name, "stream_blocks_per_cu")) { ctx->stream_blocks_per_cu = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 16); return RAWDTW_OK; }
    if (!strcmp(name, "stream_threads")) { ctx->stream_threads = value >= 512 ? 512 : 256; return RAWDTW_OK; }
    if (!strcmp(name, "stream_debug")) { ctx->stream_debug = (uint32_t)value; return RAWDTW_OK; }
    if (!strcmp(name, "resident_arrays")) { ctx->resident_arrays = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "time_plan")) { ctx->time_plan = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "merge_small")) { ctx->merge_small = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "fold_mode")) { ctx->fold_mode = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 4); return RAWDTW_OK; }
    if (!strcmp(name, "fold_long_parts")) { ctx->fold_long_parts = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 1 << 30); return RAWDTW_OK; }
    if (!strcmp(name, "tile_threads")) { ctx->tile_threads = value >= 1024 ? 1024 : (value >= 512 ? 512 : 256); return RAWDTW_OK; }
    if (!strcmp(name, "tile_max_spans")) { ctx->tile_max_spans = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 8), 4096); return RAWDTW_OK; }
    if (!strcmp(name, "full_wg")) { ctx->full_wg = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "grp16")) { ctx->grp16 = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "grp8")) { ctx->grp8 = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "micro_max_n")) { ctx->micro_max_n = value >= 8 ? 8 : (value >= 4 ? 4 : 0); return RAWDTW_OK; }
    if (!strcmp(name, "lane_hi_max_n")) { ctx->lane_hi_max_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 8), 200); return RAWDTW_OK; }
    if (!strcmp(name, "lane_hi")) { ctx->lane_hi = value != 0; return RAWDTW_OK; }
    if (!strcmp(name, "lane_max_n")) { ctx->lane_max_n = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 8), kLaneMaxN); return RAWDTW_OK; }
    if (!strcmp(name, "stream_tile_radius")) { ctx->stream_tile_radius = value < 1 ? 1 : (value > kMaxLaneRadius ? kMaxLaneRadius : (int)value); return RAWDTW_OK; }
    if (!strcmp(name, "lane_max_radius")) {
        ctx->lane_max_radius = value < 0 ? 0 : (value > kMaxLaneRadius ? kMaxLaneRadius : (int)value);
        return RAWDTW_OK;
    }
    return fail(ctx, RAWDTW_ERR_INVALID, std::string("unknown option ") + name);
}

int rawdtw_context_device(const rawdtw_ctx *ctx, int *device_ordinal)
{
    if (!ctx || !device_ordinal) return RAWDTW_ERR_INVALID;
    *device_ordinal = ctx->device;
    return RAWDTW_OK;
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
    drop_reference(ctx);
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
    ctx->ref_hold = new (std::nothrow) RefHold;
    if (!ctx->ref_hold) { (void)hipFree(ctx->d_ref); ctx->d_ref = nullptr; return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed"); }
    ctx->ref_hold->d = ctx->d_ref;
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
    drop_reference(ctx);
    ctx->d_ref = const_cast<float *>(d_ref);
    ctx->n_ref = n_floats;
    ctx->ref_off.clear();
    ctx->ref_len.clear();
    return RAWDTW_OK;
}

int rawdtw_share_reference(rawdtw_ctx *ctx, const rawdtw_ctx *owner)
{
    if (!ctx || !owner || ctx == owner) return fail(ctx, RAWDTW_ERR_INVALID, "bad arguments to share_reference");
    if (ctx->device != owner->device) return fail(ctx, RAWDTW_ERR_INVALID, "contexts on different devices cannot share an arena");
    drop_reference(ctx);
    if (owner->ref_hold) { owner->ref_hold->refs.fetch_add(1); ctx->ref_hold = owner->ref_hold; } // (else: the caller's memory, its to keep alive)
    ctx->d_ref = owner->d_ref; ctx->n_ref = owner->n_ref;
    ctx->ref_off = owner->ref_off; ctx->ref_len = owner->ref_len;
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
static void plan_release_device(rawdtw_plan *plan)
{
    auto drop = [](auto *&p) { if (p) (void)hipFree(p); p = nullptr; };
    drop(plan->d_jobs); drop(plan->d_aux); drop(plan->d_tiles); drop(plan->d_spans); drop(plan->d_tjobs);
    drop(plan->d_masks); drop(plan->d_cost); drop(plan->d_bnd);
    if (plan->d_dir && !plan->dir_borrowed) (void)hipFree(plan->d_dir);
    plan->d_dir = nullptr;
}

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

// What leaves the device per path element is its distance and ONE byte, the step from the element before it (k_tb_finish):
// 5 bytes over the bus instead of 12.  rawdtw_traceback_batch_steps hands exactly that to the caller (the mapper's aln:s:
// writer walks the steps while it formats); rawdtw_traceback_batch rebuilds (i, j) from the steps while it writes the
// caller's three arrays.  Sub-batches (only a batch whose direction buffers exceed the budget has several) run as a two-deep
// pipeline: sub-batch k's paths come home on a second stream and are written out by a few host threads while sub-batch
// k + 1 fills and walks.  (Cutting a batch that fits into quarters to hide the download behind the kernels was tried and
// cost more than it hid: a fill or walk launch takes as long as its longest job -- fill 9.0 -> 18.8 ms, walk 1.9 -> 7.4 ms
// for 8 080 paths in five launches each.)
static int traceback_core(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, const float *h_events,
                          uint64_t n_events, float *out_cost, const uint64_t *path_off, uint32_t *path_len,
                          uint32_t *path_i, uint32_t *path_j, uint8_t *path_step, float *path_d)
{
    if (!ctx) return RAWDTW_ERR_INVALID;
    if (n_jobs && (!jobs || !out_cost || !path_off || !path_len || !path_d || (!path_step && (!path_i || !path_j))))
        return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    static const bool timing = getenv("RAWDTW_PLAN_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[traceback] %-14s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    int st = rawdtw_upload_events(ctx, h_events, n_events);
    if (st != RAWDTW_OK) return st;
    lap("events H2D");
    for (hipEvent_t &e : ctx->tb_ev) if (!e) HIP_TRY(ctx, hipEventCreate(&e));
    if (!ctx->tb_copy) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->tb_copy, hipStreamNonBlocking));
    ctx->tb_fill_ms = ctx->tb_walk_ms = 0.f; ctx->tb_dir_written = 0; ctx->tb_path_elems = 0;

    uint64_t budget = 16ull << 30;
    if (const char *e = getenv("RAWDTW_TB_WORKSPACE_MB")) budget = std::max<uint64_t>(1, strtoull(e, nullptr, 10)) << 20;
    for (uint64_t k = 0; k < n_jobs; k++)
        if (jobs[k].n == 0 || jobs[k].m == 0) return fail(ctx, RAWDTW_ERR_INVALID, "zero-length traceback job");
    struct Sub {
        rawdtw_plan *pl = nullptr;
        uint64_t begin = 0, cnt = 0, acc = 0;
        std::vector<uint64_t> poff;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; // fill start, fill end, walk end, download end
        int slot = 0;
        bool in_flight = false;
        bool dense = false;   // steps form, and the caller's offsets of these jobs are one dense ascending stretch: the device writes the paths
        uint64_t lo = 0;      // in the CALLER's layout (from element lo on) and they come home as two copies, into the caller's arrays when
        bool direct = false;  // those are page-locked (rawdtw_host_alloc), else through the pinned landing zone and one memcpy a thread
    };
    auto page_locked = [](const void *p) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    };
    const bool caller_pinned = path_step && n_jobs && page_locked(path_step) && page_locked(path_d);
    std::vector<Sub> subs;
    struct Guard { // plans and events go when the call ends (a plan's hipFree waits for the device: not in the middle of the pipeline)
        std::vector<Sub> &v; rawdtw_ctx *c;
        ~Guard()
        {
            (void)hipStreamSynchronize(c->stream);
            if (c->tb_copy) (void)hipStreamSynchronize(c->tb_copy);
            for (Sub &sb : v) { if (sb.pl) rawdtw_plan_destroy(sb.pl); for (hipEvent_t &e : sb.ev) if (e) (void)hipEventDestroy(e); }
        }
    } guard{subs, ctx};
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // per slot: device path buffers {offsets, lengths, i/j end-first, distances, steps} and the pinned landing zone {distances, steps, costs, lengths}
    auto dev_need = [&](uint64_t cnt, uint64_t acc) { return al(cnt * 8) + al(cnt * 4) + 3 * al((size_t)acc * 4) + al((size_t)acc); };
    auto host_need = [&](uint64_t cnt, uint64_t acc) { return al((size_t)acc * 4) + al((size_t)acc) + 2 * al(cnt * 4); };

    // ---- all sub-batches planned first (host planner, device allocations, job records' upload: while nothing is in flight) ----
    size_t dn = 0, hn = 0;
    for (uint64_t begin = 0; begin < n_jobs;) {
        uint64_t end = begin, bytes = 0;
        while (end < n_jobs) {
            const rawdtw_job_t &j = jobs[end];
            const uint64_t b = dir_bytes_for(j.n, j.m, full_rpl(std::min(j.n, j.m))) + 256;
            if (end > begin && bytes + b > budget) break;
            bytes += b;
            end++;
        }
        subs.emplace_back();
        Sub &sb = subs.back();
        sb.begin = begin; sb.cnt = end - begin; sb.slot = (int)((subs.size() - 1) & 1);
        st = build_plan(ctx, jobs + begin, sb.cnt, true, &sb.pl);
        if (st != RAWDTW_OK) { sb.pl = nullptr; return st; }
        for (hipEvent_t &e : sb.ev) HIP_TRY(ctx, hipEventCreate(&e));
        sb.poff.resize(sb.cnt);
        uint64_t acc = 0;
        sb.dense = path_step != nullptr;
        for (uint64_t k = begin; sb.dense && k + 1 < end; k++) sb.dense = path_off[k + 1] == path_off[k] + jobs[k].n + jobs[k].m - 1;
        if (sb.dense) {
            sb.lo = path_off[begin];
            for (uint64_t p = 0; p < sb.cnt; p++) sb.poff[p] = path_off[begin + sb.pl->order[p]] - sb.lo;
            acc = path_off[end - 1] + jobs[end - 1].n + jobs[end - 1].m - 1 - sb.lo;
            sb.direct = caller_pinned;
        } else
            for (uint64_t p = 0; p < sb.cnt; p++) { sb.poff[p] = acc; acc += (uint64_t)sb.pl->h_jobs[p].n + sb.pl->h_jobs[p].m - 1; }
        sb.acc = acc;
        dn = std::max(dn, dev_need(sb.cnt, acc)); hn = std::max(hn, host_need(sb.cnt, acc));
        begin = end;
    }
    for (Sub &sb : subs) sb.pl->d_dir = ctx->d_tb_dir; // (the context's direction workspace may have grown while the later ones were planned)
    // grow-only buffers of the context, two slots each
    if (ctx->tb_paths_bytes < 2 * dn) {
        if (ctx->d_tb_paths) (void)hipFree(ctx->d_tb_paths);
        ctx->d_tb_paths = nullptr; ctx->tb_paths_bytes = 0;
        const size_t want = 2 * (dn + dn / 8);
        if (hipMalloc(&ctx->d_tb_paths, want) != hipSuccess) return fail(ctx, RAWDTW_ERR_OOM, "path buffer allocation failed");
        ctx->tb_paths_bytes = want;
    }
    if (ctx->pinned_bytes < 2 * hn) {
        if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
        ctx->h_pinned = nullptr; ctx->pinned_bytes = 0;
        const size_t want = 2 * (hn + hn / 8);
        if (hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault) != hipSuccess) return fail(ctx, RAWDTW_ERR_OOM, "pinned host allocation failed");
        ctx->pinned_bytes = want;
    }
    lap("plans + alloc");

    // the second half of a sub-batch: wait for its download, write the caller's arrays
    auto finish = [&](Sub &sb) -> int {
        if (!sb.in_flight) return RAWDTW_OK;
        sb.in_flight = false;
        hipError_t e = hipEventSynchronize(sb.ev[3]);
        if (e != hipSuccess) return hip_fail(ctx, e, "traceback download");
        lap("wait download");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sb.ev[0], sb.ev[1]) == hipSuccess) ctx->tb_fill_ms += ms;
        if (hipEventElapsedTime(&ms, sb.ev[1], sb.ev[2]) == hipSuccess) ctx->tb_walk_ms += ms;
        ctx->tb_dir_written += sb.pl->dir_bytes;
        const char *hp = static_cast<const char *>(ctx->h_pinned) + (size_t)sb.slot * (ctx->pinned_bytes / 2);
        const float *h_pd = reinterpret_cast<const float *>(hp);
        const uint8_t *h_mv = reinterpret_cast<const uint8_t *>(hp + al((size_t)sb.acc * 4));
        const float *h_cost = reinterpret_cast<const float *>(hp + al((size_t)sb.acc * 4) + al((size_t)sb.acc));
        const uint32_t *h_plen = reinterpret_cast<const uint32_t *>(hp + al((size_t)sb.acc * 4) + al((size_t)sb.acc) + al(sb.cnt * 4));
        // into the caller's arrays (pageable memory: spread over a few threads); (i, j) from the steps
        int T = ctx->plan_threads > 0 ? ctx->plan_threads : (int)std::min<uint64_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u), sb.acc / (1u << 20) + 1);
        T = std::max(1, std::min(T, 16));
        std::vector<uint64_t> elems(T, 0);
        const rawdtw_plan *pl = sb.pl;
        parallel_for(T, [&](int t) {
            for (uint64_t p = sb.cnt * (uint64_t)t / T; p < sb.cnt * (uint64_t)(t + 1) / T; p++) {
                const uint64_t k = sb.begin + pl->order[p];
                out_cost[k] = h_cost[pl->order[p]];
                const uint32_t len = h_plen[p];
                // device paths are start-first already (k_tb_finish, dtw.cpp:656-657); the reference pops the last
                // element when exclude_last_element is set (dtw.cpp:659-663)
                const uint32_t outlen = jobs[k].exclude_last ? len - 1 : len;
                const uint64_t src = sb.poff[p], dst = path_off[k];
                const uint8_t *mv = h_mv + src;
                if (sb.dense) { /* (the stretch is copied whole below, or came home in place) */ }
                else if (path_step) memcpy(path_step + dst, mv, outlen);
                else {
                    uint32_t i = 0, j = 0;
                    uint32_t *pi = path_i + dst, *pj = path_j + dst;
                    for (uint32_t q = 0; q < outlen; q++) { i += mv[q] & 1u; j += mv[q] >> 1; pi[q] = i; pj[q] = j; }
                }
                if (!sb.dense) memcpy(path_d + dst, h_pd + src, (size_t)outlen * 4);
                path_len[k] = outlen;
                elems[t] += outlen;
            }
            if (sb.dense && !sb.direct) { // this thread's share of the stretch
                const uint64_t a0 = sb.acc * (uint64_t)t / T, a1 = sb.acc * (uint64_t)(t + 1) / T;
                memcpy(path_step + sb.lo + a0, h_mv + a0, a1 - a0);
                memcpy(path_d + sb.lo + a0, h_pd + a0, (a1 - a0) * 4);
            }
        });
        for (int t = 0; t < T; t++) ctx->tb_path_elems += elems[t];
        lap("copy out");
        return RAWDTW_OK;
    };

    for (size_t k = 0; k < subs.size(); k++) {
        Sub &sb = subs[k];
        if (k >= 2) { st = finish(subs[k - 2]); if (st != RAWDTW_OK) return st; } // (its slot's buffers are this sub-batch's now)
        rawdtw_plan *pl = sb.pl;
        const uint64_t acc = sb.acc;
        char *pb = static_cast<char *>(ctx->d_tb_paths) + (size_t)sb.slot * (ctx->tb_paths_bytes / 2);
        uint64_t *d_poff = reinterpret_cast<uint64_t *>(pb); pb += al(sb.cnt * 8);
        uint32_t *d_plen = reinterpret_cast<uint32_t *>(pb); pb += al(sb.cnt * 4);
        uint32_t *d_ti = reinterpret_cast<uint32_t *>(pb); pb += al((size_t)acc * 4);
        uint32_t *d_tj = reinterpret_cast<uint32_t *>(pb); pb += al((size_t)acc * 4);
        float *d_pd = reinterpret_cast<float *>(pb); pb += al((size_t)acc * 4);
        uint8_t *d_mv = reinterpret_cast<uint8_t *>(pb);
        hipError_t e = hipMemcpyAsync(d_poff, sb.poff.data(), sb.cnt * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return hip_fail(ctx, e, "path offsets upload");
        (void)hipEventRecord(sb.ev[0], ctx->stream);
        st = rawdtw_plan_run(ctx, pl);
        if (st != RAWDTW_OK) return st;
        (void)hipEventRecord(sb.ev[1], ctx->stream);
        for (const Launch &L : pl->launches) {
            // one wave per job over the direction buffer, then start-first order, steps and distances (k_tb_finish)
            e = launch_tb_walk_wave(pl->d_jobs + L.first, L.count, pl->d_aux + L.first, L.param & 255, ctx->d_ev, ctx->d_ref,
                                    pl->d_dir, d_poff + L.first, d_plen + L.first, d_ti, d_tj, d_mv, d_pd, ctx->stream);
            if (e != hipSuccess) return hip_fail(ctx, e, "traceback walk launch");
        }
        (void)hipEventRecord(sb.ev[2], ctx->stream);
        // the download: on the second stream, behind the walk; everything lands in pinned memory (a download into pageable
        // memory makes the call wait for the kernels in front of it)
        char *hp = static_cast<char *>(ctx->h_pinned) + (size_t)sb.slot * (ctx->pinned_bytes / 2);
        char *hc = hp + al((size_t)acc * 4) + al((size_t)acc);
        e = hipStreamWaitEvent(ctx->tb_copy, sb.ev[2], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(hc, pl->d_cost, sb.cnt * 4, hipMemcpyDeviceToHost, ctx->tb_copy);
        if (e == hipSuccess) e = hipMemcpyAsync(hc + al(sb.cnt * 4), d_plen, sb.cnt * 4, hipMemcpyDeviceToHost, ctx->tb_copy);
        if (e == hipSuccess && acc) e = hipMemcpyAsync(sb.direct ? static_cast<void *>(path_d + sb.lo) : static_cast<void *>(hp), d_pd, acc * 4, hipMemcpyDeviceToHost, ctx->tb_copy);
        if (e == hipSuccess && acc) e = hipMemcpyAsync(sb.direct ? static_cast<void *>(path_step + sb.lo) : static_cast<void *>(hp + al((size_t)acc * 4)), d_mv, acc, hipMemcpyDeviceToHost, ctx->tb_copy);
        if (e == hipSuccess) e = hipEventRecord(sb.ev[3], ctx->tb_copy);
        if (e != hipSuccess) return hip_fail(ctx, e, "traceback download");
        sb.in_flight = true;
        lap("enqueue");
        // ... and while all that runs: the sub-batch before this one
        if (k >= 1) { st = finish(subs[k - 1]); if (st != RAWDTW_OK) return st; }
    }
    for (size_t k = subs.size() >= 2 ? subs.size() - 2 : 0; k < subs.size(); k++) { st = finish(subs[k]); if (st != RAWDTW_OK) return st; }
    return RAWDTW_OK;
}

int rawdtw_traceback_batch(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, const float *h_events,
                           uint64_t n_events, float *out_cost, const uint64_t *path_off, uint32_t *path_len,
                           uint32_t *path_i, uint32_t *path_j, float *path_d)
{
    if (n_jobs && (!path_i || !path_j)) return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    return traceback_core(ctx, jobs, n_jobs, h_events, n_events, out_cost, path_off, path_len, path_i, path_j, nullptr, path_d);
}

int rawdtw_traceback_batch_steps(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, const float *h_events,
                                 uint64_t n_events, float *out_cost, const uint64_t *path_off, uint32_t *path_len,
                                 uint8_t *path_step, float *path_d)
{
    if (n_jobs && !path_step) return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    return traceback_core(ctx, jobs, n_jobs, h_events, n_events, out_cost, path_off, path_len, nullptr, nullptr, path_step, path_d);
}

// ---- single-call drop-ins ----------------------------------------------------------------------
static int single_call(rawdtw_ctx *ctx, const float *a, uint32_t n, const float *b, uint32_t m, int radius,
                       int excl, float *cost)
{
    if (!ctx || !a || !b || !cost) return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    if (n == 0 || m == 0 || radius < RAWDTW_FULL) return fail(ctx, RAWDTW_ERR_INVALID, "zero length or negative radius");
    // b goes to a private reference arena for the duration of the call
    const float *saved_ref = ctx->d_ref; uint64_t saved_n = ctx->n_ref; // (the hold on the context's own arena stays)
    float *d_b = nullptr;
    int st = dev_alloc(ctx, &d_b, ((uint64_t)m + 3) & ~3ull);
    if (st != RAWDTW_OK) return st;
    hipError_t e = hipMemcpyAsync(d_b, b, (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d_b); return hip_fail(ctx, e, "operand upload"); }
    ctx->d_ref = d_b; ctx->n_ref = m;
    rawdtw_job_t j{0, 0, n, m, radius, excl ? 1u : 0u, 0};
    st = rawdtw_score_batch(ctx, &j, 1, a, n, cost);
    ctx->d_ref = const_cast<float *>(saved_ref); ctx->n_ref = saved_n;
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
    const float *saved_ref = ctx->d_ref; uint64_t saved_n = ctx->n_ref; // (the hold on the context's own arena stays)
    float *d_b = nullptr;
    int st = dev_alloc(ctx, &d_b, ((uint64_t)m + 3) & ~3ull);
    if (st != RAWDTW_OK) return st;
    hipError_t e = hipMemcpyAsync(d_b, b, (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d_b); return hip_fail(ctx, e, "operand upload"); }
    ctx->d_ref = d_b; ctx->n_ref = m;
    rawdtw_job_t j{0, 0, n, m, RAWDTW_FULL, exclude_last ? 1u : 0u, 0};
    uint64_t off = 0;
    st = rawdtw_traceback_batch(ctx, &j, 1, a, n, cost, &off, path_len, path_i, path_j, path_d);
    ctx->d_ref = const_cast<float *>(saved_ref); ctx->n_ref = saved_n;
    (void)hipFree(d_b);
    return st;
}

// ---- whole-batch form ----------------------------------------------------------------------------
// Two implementations behind rawdtw_batch_*:
//   stream  (sparse + banded batches, the default): rawdtw_stream.hip -- rawdtw_batch_create only enqueues copies and
//           planning kernels on the context's stream, no host synchronisation, no allocation in the steady state
//           (workspaces are pooled per context); every count stays on the device.
//   job list (everything else, and the rare batch the stream path declines): the jobs are built on the host and go
//           through plan_host / build_plan like any rawdtw_plan.
namespace {

struct StreamLayout { // sizes in bytes of one batch's device workspace and pinned host block
    size_t dev = 0, host = 0, tmp = 0;
    uint64_t others_cap = 0;
    uint32_t tiles_cap = 0;
};

// LDS image of a device-planned batch's tiles, in floats.  With four workgroups a CU (stream_blocks_per_cu) a SIMD keeps
// 128 registers free beside the DTW launch's waves -- room for a wave of the next batches' planning kernels or of the
// batch before's fold -- and the LDS that a fifth workgroup would take goes into larger tiles (fewer tiles, fuller sorted
// waves).  Measured on the bench pipeline (4 batches in flight): 5 x 4800 floats 425 GCUPS, 4 x 7200 453, 3 x 6400 450;
// later, with k_pre's chain table in LDS (2 KB a workgroup): 4 x 7200 508, 4 x 7000 524, 4 x 6800 523, 4 x 6000 511 -- the planning
// kernels of the batches behind need LDS beside the four resident workgroups.
static uint32_t stream_tile_floats(const rawdtw_ctx *ctx) { return (ctx->tile_lds_set ? ctx->tile_lds_floats : kStreamTileFloats) & ~3u; } // (16-byte multiples: the records and the sort table sit behind the image)

bool stream_eligible(const rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_anchors)
{
    // (the job-list path takes batches below "device_plan_min_jobs" anchors: a chain of n anchors has n - 1 jobs)
    if (!ctx->device_plan || n_anchors < ctx->device_plan_min_jobs || n_anchors == 0 || n_anchors >= (1ull << 31)) return false;
    if (opt->border_constraint != 1 || opt->fill_method == 0) return false; // sparse + banded only
    if (ctx->lane_max_radius < 0 || ctx->sort_n || ctx->sort_r1_n || ctx->sort_r3 || ctx->lane_hi || !ctx->merge_small) return false;
    if (ctx->tile_threads != 256 || ctx->debug_skip_kinds) return false;
    const uint32_t worst_part = 2u * ctx->lane_max_n + 16u; // image floats of the largest tile-class part alone
    return stream_tile_floats(ctx) >= 2u * worst_part && stream_tile_floats(ctx) <= 16384u;
}

int ws_acquire(rawdtw_ctx *ctx, size_t dev_bytes, size_t host_bytes, StreamWs *out)
{
    int best = -1;
    for (size_t i = 0; i < ctx->ws_free.size(); i++) {
        const StreamWs &w = ctx->ws_free[i];
        if (w.d_bytes >= dev_bytes && w.h_bytes >= host_bytes && (best < 0 || w.d_bytes < ctx->ws_free[best].d_bytes)) best = (int)i;
    }
    if (best >= 0) {
        *out = ctx->ws_free[best];
        ctx->ws_free.erase(ctx->ws_free.begin() + best);
        return RAWDTW_OK;
    }
    // nothing fits: drop the smallest pooled workspace when the pool is full, then allocate with head room
    if (ctx->ws_free.size() >= 8) {
        size_t smallest = 0;
        for (size_t i = 1; i < ctx->ws_free.size(); i++) if (ctx->ws_free[i].d_bytes < ctx->ws_free[smallest].d_bytes) smallest = i;
        (void)hipFree(ctx->ws_free[smallest].d); (void)hipHostFree(ctx->ws_free[smallest].h);
        ctx->ws_free.erase(ctx->ws_free.begin() + smallest);
    }
    StreamWs w;
    w.d_bytes = dev_bytes + dev_bytes / 4; w.h_bytes = host_bytes + host_bytes / 4;
    if (hipMalloc(reinterpret_cast<void **>(&w.d), w.d_bytes) != hipSuccess) return fail(ctx, RAWDTW_ERR_OOM, "batch workspace allocation failed");
    if (hipHostMalloc(reinterpret_cast<void **>(&w.h), w.h_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(w.d);
        return fail(ctx, RAWDTW_ERR_OOM, "pinned batch staging allocation failed");
    }
    *out = w;
    return RAWDTW_OK;
}

void ws_release(rawdtw_ctx *ctx, StreamWs &w)
{
    if (w.d) ctx->ws_free.push_back(w);
    w = StreamWs{};
}

bool stream_declined(const rawdtw_batch *b);

// the side list's launch on the context's second stream: behind everything enqueued on the main stream so far; the main
// stream joins it (ev_wide_join) before the fold
static hipError_t stream_wide_fork(rawdtw_ctx *ctx, const StreamArgs &a)
{
    if (!ctx->wide_beside) return stream_wide(a, ctx->wide_blocks, ctx->stream); // (in line: nothing to join)
    hipError_t he = hipEventRecord(ctx->ev_wide_fork, ctx->stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(ctx->wide, ctx->ev_wide_fork, 0);
    if (he == hipSuccess) he = stream_wide(a, ctx->wide_blocks, ctx->wide);
    if (he == hipSuccess) he = hipEventRecord(ctx->ev_wide_join, ctx->wide);
    return he;
}

// the stream path: everything rawdtw_batch_create does for a sparse + banded batch -- O(1) host work: a workspace from the
// pool, five copies and three launches enqueued
int batch_create_stream(rawdtw_ctx *ctx, rawdtw_batch *b, const uint64_t *chain_off, const uint64_t *anchor_off,
                        const rawdtw_anchor_t *anchors, const uint64_t *ref_base, const uint32_t *read_base)
{
    // (a chunk round: the device's lists are the SHORT ones -- new entries + junction -- and `na` their length; the full lists only
    // give the fold its offsets)
    const rawdtw_batch *prev = b->in_prev;
    const bool round = prev != nullptr; // (rawdtw_batch_submit_carry checked that it can serve: rawdtw_batch_can_carry)
    const uint64_t nc = b->n_chains, nr = b->n_reads, n_full = anchor_off[nc], na = round ? b->in_new_off[nc] : n_full;
    const uint32_t lds_floats = stream_tile_floats(ctx);
    StreamArgs &a = b->sa;
    a = StreamArgs{};
    a.n_anchors = na; a.n_chains = nc; a.n_reads = nr; a.n_ev = ctx->n_ev; a.n_ref = ctx->n_ref;
    a.frac = b->opt.band_radius_frac;
    // tiles take radius <= stream_tile_radius; the radii between that and lane_max_radius (none by default) go to the side
    // list's lane classes
    a.lane_max_radius = std::min(ctx->lane_max_radius, ctx->stream_tile_radius); a.side_lane_radius = ctx->lane_max_radius;
    a.lane_max_n = ctx->lane_max_n;
    a.tile_anchors = kStreamTile;
    a.n_tiles = (uint32_t)((na + a.tile_anchors - 1) / a.tile_anchors);
    // (a tile over the image budget or the run table takes further passes, a slot of copy orders each: rare in a mapper's
    // batch, the rule for tiles of very short chains; a batch that runs out of slots is redone through the job list)
    a.n_slots = ctx->pass_pool >= 0 ? a.n_tiles + (uint32_t)ctx->pass_pool : 4 * a.n_tiles + 64;
    a.lds_floats = lds_floats;
    a.others_cap = std::min<uint64_t>(na, na / 4 + 4096);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const bool compact = b->in_steps != nullptr;
    const uint64_t n_units = (na + RAWDTW_COMPACT_STRIDE - 1) / RAWDTW_COMPACT_STRIDE;
    const size_t compact_bytes = compact ? al(nc * 8) + al(n_units * 8) + al(n_units * RAWDTW_COMPACT_STRIDE * 2) + al(b->in_n_wide * sizeof(rawdtw_wide_step_t)) : 0;
    const size_t round_bytes = round ? al(nc * sizeof(rawdtw_carry_t)) + al((nc + 1) * 8) + al(n_full * 4) : 0;
    const size_t dev_bytes = compact_bytes + round_bytes +
                             al(kStreamCounters * 8) + al((nc + 1) * 8) + al(na * 8) + al(nc * 8) + al(nc * 4) + al((nr + 1) * 8) + // counters, inputs
                             al((size_t)a.n_tiles * 8) + al((size_t)a.n_slots * 16) + al((size_t)a.n_tiles * 24) +               // tile list, work list, statistics
                             al((size_t)a.n_tiles * kStreamRecStride * 8) + al((size_t)a.n_slots * 2 * kStreamMaxSeg * 16) +         // job records, copy orders
                             2 * al(a.others_cap * sizeof(DevJob)) + al(a.others_cap) +                                           // side list
                             al(nc * sizeof(ChainDesc)) + 4 * al(nc * 4) + al(nc) + al(na * 4);                                   // fold, results
    const size_t host_bytes = al(kStreamCounters * 8) + al(nc * 4) + al(nc);
    int st = ws_acquire(ctx, dev_bytes, host_bytes, &b->ws);
    if (st != RAWDTW_OK) return st;
    char *p = b->ws.d;
    a.cnt = carve<unsigned long long>(p, kStreamCounters);
    b->d_score = carve<float>(p, nc); b->d_keep = carve<uint8_t>(p, nc); // (right behind the counters: one copy brings all three home)
    b->res_bytes = (size_t)(reinterpret_cast<char *>(b->d_keep) - reinterpret_cast<char *>(a.cnt)) + nc;
    uint64_t *d_anchor_off = carve<uint64_t>(p, nc + 1);
    rawdtw_anchor_t *d_anchors = carve<rawdtw_anchor_t>(p, na);
    uint64_t *d_ref_base = carve<uint64_t>(p, nc);
    uint32_t *d_read_base = carve<uint32_t>(p, nc);
    b->d_chain_off = carve<uint64_t>(p, nr + 1);
    a.tlist = carve<uint2>(p, a.n_tiles);
    a.todo = carve<uint4>(p, a.n_slots);
    a.recs = carve<uint2>(p, (uint64_t)a.n_tiles * kStreamRecStride);
    a.runtab = carve<uint4>(p, (uint64_t)a.n_slots * 2 * kStreamMaxSeg);
    a.tile_stats = carve<unsigned long long>(p, 3ull * a.n_tiles);
    a.omix = carve<DevJob>(p, a.others_cap); a.ojobs = carve<DevJob>(p, a.others_cap); a.ocls = carve<uint8_t>(p, a.others_cap);
    b->d_chains = carve<ChainDesc>(p, nc);
    b->d_fold_order = carve<uint32_t>(p, nc);
    b->d_full = carve<float>(p, nc); b->d_gate = carve<float>(p, nc);
    a.out = carve<float>(p, na);
    a.debug = ctx->stream_debug;
    a.anchor_off = d_anchor_off; a.anchors = d_anchors; a.ref_base = d_ref_base; a.read_base = d_read_base;
    if (b->in_resident) { a.anchors = round ? b->in_new_anchors : anchors; a.ref_base = ref_base; a.read_base = read_base; } // used in place
    rawdtw_anchor_t *d_heads = nullptr, *d_unit_abs = nullptr;
    uint16_t *d_steps = nullptr;
    rawdtw_wide_step_t *d_wide = nullptr;
    if (compact) { // the packed lists; k_scan decodes them into d_anchors
        d_heads = carve<rawdtw_anchor_t>(p, nc); d_unit_abs = carve<rawdtw_anchor_t>(p, n_units);
        d_steps = carve<uint16_t>(p, n_units * RAWDTW_COMPACT_STRIDE); d_wide = carve<rawdtw_wide_step_t>(p, b->in_n_wide);
        a.heads = d_heads; a.unit_abs = d_unit_abs; a.steps = d_steps; a.wide = d_wide; a.n_wide = b->in_n_wide; a.anchors_w = d_anchors;
    }
    a.ev = ctx->d_ev; a.ref = ctx->d_ref;
    rawdtw_carry_t *d_carry = nullptr;
    uint64_t *d_full_off = nullptr;
    a.full_off = d_anchor_off; a.n_full = n_full; a.out_full = a.out; // (no predecessor: the lists are the full ones)
    if (round) { // (the previous batch's cost array is read by this batch's k_gather: stream order keeps it alive that long)
        d_carry = carve<rawdtw_carry_t>(p, nc); d_full_off = carve<uint64_t>(p, nc + 1);
        a.out_full = carve<float>(p, n_full);
        a.carry = d_carry; a.full_off = d_full_off;
        a.prev_out_full = prev->sa.out_full; a.prev_n_full = prev->sa.n_full;
        a.prev_cnt = prev->sa.cnt; a.prev_others_cap = prev->sa.others_cap;
    }
    char *hp = b->ws.h;
    b->h_cnt = carve<unsigned long long>(hp, kStreamCounters);
    b->h_score = carve<float>(hp, nc); b->h_keep = carve<uint8_t>(hp, nc); // (same offsets as on the device)
    unsigned long long *h_init = b->h_cnt; // the counters' initial values travel from the pinned block
    for (int i = 0; i < kStreamCounters; i++) h_init[i] = 0;
    h_init[kCntBad] = h_init[kCntOverflow] = ~0ull;
    hipStream_t s = ctx->stream;
    if (ctx->time_plan) for (hipEvent_t &pe : b->ev_plan) if (!pe) HIP_TRY(ctx, hipEventCreate(&pe));
    HIP_TRY(ctx, hipMemcpyAsync(a.cnt, h_init, kStreamCounters * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_anchor_off, round ? b->in_new_off : anchor_off, (nc + 1) * 8, hipMemcpyHostToDevice, s));
    if (compact) {
        HIP_TRY(ctx, hipMemcpyAsync(d_heads, b->in_heads, nc * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(d_unit_abs, b->in_unit_abs, n_units * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(d_steps, b->in_steps, na * 2, hipMemcpyHostToDevice, s));
        if (b->in_n_wide) HIP_TRY(ctx, hipMemcpyAsync(d_wide, b->in_wide, b->in_n_wide * sizeof(rawdtw_wide_step_t), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(d_ref_base, ref_base, nc * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(d_read_base, read_base, nc * 4, hipMemcpyHostToDevice, s));
    } else if (round) { // only the round's NEW anchors (and the junctions) cross the bus
        HIP_TRY(ctx, hipMemcpyAsync(d_carry, b->in_carry, nc * sizeof(rawdtw_carry_t), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(d_full_off, anchor_off, (nc + 1) * 8, hipMemcpyHostToDevice, s));
        if (b->in_resident) a.anchors = b->in_new_anchors; // ("resident_arrays": the three big arrays are device pointers, used in place)
        else {
            if (na) HIP_TRY(ctx, hipMemcpyAsync(d_anchors, b->in_new_anchors, na * sizeof(rawdtw_anchor_t), hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, hipMemcpyAsync(d_ref_base, ref_base, nc * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, hipMemcpyAsync(d_read_base, read_base, nc * 4, hipMemcpyHostToDevice, s));
        }
    } else if (!b->in_resident) {
        HIP_TRY(ctx, hipMemcpyAsync(d_anchors, anchors, na * sizeof(rawdtw_anchor_t), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(d_ref_base, ref_base, nc * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(d_read_base, read_base, nc * 4, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(ctx, hipMemcpyAsync(b->d_chain_off, chain_off, (nr + 1) * 8, hipMemcpyHostToDevice, s));
    b->fold_fused = ctx->fold_mode == 4; // (no fold order then: the one-workgroup sort stays off the scan's critical path)
    if (ctx->time_plan) HIP_TRY(ctx, hipEventRecord(b->ev_plan[0], s)); // ("time_plan": the planning LAUNCHES, behind the hand-over's copies)
    hipError_t e = stream_plan(a, b->d_chains, b->fold_fused ? nullptr : b->d_fold_order, s);
    // The side list's launch goes out here, between the scan and the pass planning, for the batch's first run (a batch that
    // runs again launches it again in front of the tiles' launch): measured, the fresh-batch pipeline runs 6 % faster with
    // the wide bands' long tail in front of the planning launch than behind it.
    if (e == hipSuccess && ctx->time_plan) e = hipEventRecord(b->ev_plan[1], s);
    b->wide_out = false;
    // (only inside rawdtw_batch_submit*: between a separate create and run the caller may upload new events, and a run reads
    // the arenas as they are then)
    if (e == hipSuccess && !(ctx->stream_debug & 4u) && ctx->wide_order == 0 && (ctx->in_submit || ctx->wide_at_create)) { e = stream_wide_fork(ctx, a); b->wide_out = e == hipSuccess; }
    if (e == hipSuccess && ctx->time_plan) e = hipEventRecord(b->ev_plan[2], s);
    if (e == hipSuccess) e = stream_plan_passes(a, s);
    if (e != hipSuccess) return hip_fail(ctx, e, "batch planning launches");
    if (ctx->time_plan) HIP_TRY(ctx, hipEventRecord(b->ev_plan[3], s));
    // the persistent grid: what the device holds at this LDS size
    if (ctx->stream_lds != lds_floats || ctx->stream_threads_cached != ctx->stream_threads || ctx->stream_bpc_cached != ctx->stream_blocks_per_cu) {
        hipDeviceProp_t prop;
        HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        const int per_cu = stream_blocks_per_cu(lds_floats, ctx->stream_threads);
        if (per_cu <= 0) return fail(ctx, RAWDTW_ERR_DEVICE, "occupancy query failed for the batch kernel");
        const int use = ctx->stream_blocks_per_cu > 0 ? std::min(per_cu, ctx->stream_blocks_per_cu) : per_cu;
        ctx->stream_blocks = (uint32_t)(use * prop.multiProcessorCount);
        ctx->stream_bpc_cached = ctx->stream_blocks_per_cu;
        ctx->stream_lds = lds_floats;
        ctx->stream_threads_cached = ctx->stream_threads;
    }
    b->stream = true;
    b->stream_lds = lds_floats;
    b->stream_threads = ctx->stream_threads;
    b->n_jobs = 0; b->jobs_counted = false;
    b->cnt_valid = false;
    b->dirty = true;
    b->ws_bytes = dev_bytes;
    return RAWDTW_OK;
}

// DTW jobs of a sync-free batch (align_chain issues n_anchors - 1 per chain, rmap.cpp:248): counted from the caller's
// chain offsets the first time somebody asks -- rawdtw_batch_create itself does not walk the chains
void batch_count_jobs(rawdtw_batch *b)
{
    if (b->jobs_counted || !b->stream) return;
    uint64_t n = 0;
    for (uint64_t c = 0; c < b->n_chains; c++) {
        const uint64_t k = b->in_anchor_off[c + 1] - b->in_anchor_off[c];
        n += k ? k - 1 : 0;
    }
    b->n_jobs = n;
    b->jobs_counted = true;
}

// "resident_arrays": bring the three device-resident arrays to the host (the job-list path reads them there)
int materialise_host_arrays(rawdtw_ctx *ctx, rawdtw_batch *b)
{
    if (b->in_steps) { // the compact form: the job list is built from plain anchors
        const uint64_t nc = b->n_chains, na = b->in_anchor_off[nc];
        try { b->host_anchors.resize(na); } catch (const std::bad_alloc &) { return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed"); }
        if (rawdtw_anchors_unpack(nc, b->in_anchor_off, b->in_heads, b->in_unit_abs, b->in_steps, b->in_wide, b->in_n_wide, b->host_anchors.data()) != RAWDTW_OK)
            return fail(ctx, RAWDTW_ERR_INVALID, "malformed compact anchor lists");
        b->in_anchors = b->host_anchors.data();
        b->in_steps = nullptr;
        return RAWDTW_OK;
    }
    if (b->in_carried) { // a chunk round: the device only has the short lists; the full ones are the caller's, for exactly this
        if (!b->in_anchors) return fail(ctx, RAWDTW_ERR_UNSUPPORTED, "a carried round the device-planned path declined, and no full anchor lists to redo it from: submit the round whole");
        b->in_carried = false;
        if (b->in_resident) { // (the bases are the caller's device arrays)
            const uint64_t nc = b->n_chains;
            try { b->host_ref_base.resize(nc); b->host_read_base.resize(nc); } catch (const std::bad_alloc &) { return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed"); }
            if (nc) HIP_TRY(ctx, hipMemcpy(b->host_ref_base.data(), b->in_ref_base, nc * 8, hipMemcpyDeviceToHost));
            if (nc) HIP_TRY(ctx, hipMemcpy(b->host_read_base.data(), b->in_read_base, nc * 4, hipMemcpyDeviceToHost));
            b->in_ref_base = b->host_ref_base.data(); b->in_read_base = b->host_read_base.data();
            b->in_resident = false;
        }
        return RAWDTW_OK;
    }
    if (!b->in_resident) return RAWDTW_OK;
    const uint64_t nc = b->n_chains, na = b->in_anchor_off[nc];
    try { b->host_anchors.resize(na); b->host_ref_base.resize(nc); b->host_read_base.resize(nc); }
    catch (const std::bad_alloc &) { return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed"); }
    if (na) HIP_TRY(ctx, hipMemcpy(b->host_anchors.data(), b->in_anchors, na * sizeof(rawdtw_anchor_t), hipMemcpyDeviceToHost));
    if (nc) HIP_TRY(ctx, hipMemcpy(b->host_ref_base.data(), b->in_ref_base, nc * 8, hipMemcpyDeviceToHost));
    if (nc) HIP_TRY(ctx, hipMemcpy(b->host_read_base.data(), b->in_read_base, nc * 4, hipMemcpyDeviceToHost));
    b->in_anchors = b->host_anchors.data(); b->in_ref_base = b->host_ref_base.data(); b->in_read_base = b->host_read_base.data();
    b->in_resident = false;
    return RAWDTW_OK;
}

// the job-list path: jobs built on the host (chain ranges spread over the planner's threads), plan_host, chain records
int batch_create_joblist(rawdtw_ctx *ctx, rawdtw_batch *b, const uint64_t *chain_off, const uint64_t *anchor_off,
                         const rawdtw_anchor_t *anchors, const uint64_t *ref_base, const uint32_t *read_base,
                         const std::vector<uint64_t> &job_off, uint64_t n_jobs)
{
    const uint64_t n_chains = b->n_chains, n_reads = b->n_reads;
    const rawdtw_align_opt_t *opt = &b->opt;
    // chain descriptors: from the anchors alone.  The parts' read regions telescope (consecutive parts share their
    // anchor event), so sum(n) = (last.q - first.q) + parts in the reference's uint32 arithmetic (rmap.cpp:236,292).
    std::vector<ChainDesc> desc(n_chains);
    for (uint64_t c = 0; c < n_chains; c++) {
        const uint64_t a0 = anchor_off[c], a1 = anchor_off[c + 1];
        ChainDesc &d = desc[c];
        d.job_first = job_off[c];
        d.n_jobs = (uint32_t)(job_off[c + 1] - job_off[c]);
        d.descending = 0;
        if (a1 == a0) { d.span = 0; d.num_aligned = 0; continue; }
        const rawdtw_anchor_t &first = anchors[a1 - 1], &last = anchors[a0];
        d.span = last.query_position - first.query_position + 1; // rmap.cpp:202,245
        d.num_aligned = opt->border_constraint == 0 ? d.span : (last.query_position - first.query_position) + d.n_jobs;
    }
    RawVec<rawdtw_job_t> jobs;
    try { jobs.resize(n_jobs); } catch (const std::bad_alloc &) { return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed"); }
    int T = ctx->plan_threads;
    if (T <= 0) {
        const unsigned hc = std::thread::hardware_concurrency();
        T = (int)std::min<uint64_t>(std::min<unsigned>(hc ? hc : 1, 16), n_jobs / 32768 + 1);
    }
    T = std::max(1, std::min(T, 64));
    std::vector<int> status(T, RAWDTW_OK);
    parallel_for(T, [&](int t) {
        // split by jobs, not chains: chain lengths are skewed
        const uint64_t j_lo = n_jobs * (uint64_t)t / T, j_hi = n_jobs * (uint64_t)(t + 1) / T;
        const uint64_t c_lo = std::lower_bound(job_off.begin(), job_off.begin() + n_chains, j_lo) - job_off.begin();
        const uint64_t c_hi = t + 1 == T ? n_chains
                                         : std::lower_bound(job_off.begin(), job_off.begin() + n_chains, j_hi) - job_off.begin();
        for (uint64_t c = c_lo; c < c_hi; c++) {
            const uint64_t a0 = anchor_off[c], a1 = anchor_off[c + 1];
            const uint32_t nj = (uint32_t)(job_off[c + 1] - job_off[c]);
            if (!nj) continue;
            int s2 = rawdtw_chain_build_jobs(opt, anchors + a0, (uint32_t)(a1 - a0), ref_base[c], read_base[c], 0,
                                             jobs.data() + job_off[c]);
            if (s2 != RAWDTW_OK) { status[t] = s2; return; }
        }
    });
    for (int t = 0; t < T; t++) if (status[t] != RAWDTW_OK) return fail(ctx, status[t], "job building failed");
    int st = build_plan(ctx, jobs.data(), n_jobs, false, &b->plan);
    if (st != RAWDTW_OK) return st;
    b->n_jobs = n_jobs;
    st = dev_alloc(ctx, &b->d_chains, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_chain_off, n_reads + 1);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_fold_order, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_full, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_gate, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_score, n_chains);
    if (st == RAWDTW_OK) st = dev_alloc(ctx, &b->d_keep, n_chains);
    if (st != RAWDTW_OK) return st;
    b->own_chain_arrays = true;
    hipError_t e = hipSuccess;
    // fold order: longest chain first (stable counting sort on the part count)
    std::vector<uint32_t> fold_order(n_chains);
    {
        constexpr uint32_t kB = 65536;
        std::vector<uint64_t> start(kB + 1, 0);
        auto bucket = [&](uint64_t c) { return kB - 1 - std::min<uint32_t>(desc[c].n_jobs, kB - 1); };
        for (uint64_t c = 0; c < n_chains; c++) start[bucket(c) + 1]++;
        for (uint32_t q = 0; q < kB; q++) start[q + 1] += start[q];
        for (uint64_t c = 0; c < n_chains; c++) fold_order[start[bucket(c)]++] = (uint32_t)c;
    }
    if (n_chains) e = hipMemcpyAsync(b->d_chains, desc.data(), n_chains * sizeof(ChainDesc), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n_chains)
        e = hipMemcpyAsync(b->d_fold_order, fold_order.data(), n_chains * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b->d_chain_off, chain_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); // (the staging vectors above die with this scope)
    if (e != hipSuccess) return hip_fail(ctx, e, "uploading chain descriptors");
    return RAWDTW_OK;
}

void batch_release_device(rawdtw_batch *b)
{
    if (b->plan) { rawdtw_plan_destroy(b->plan); b->plan = nullptr; }
    if (b->own_chain_arrays) {
        if (b->d_chains) (void)hipFree(b->d_chains);
        if (b->d_chain_off) (void)hipFree(b->d_chain_off);
        if (b->d_fold_order) (void)hipFree(b->d_fold_order);
        if (b->d_full) (void)hipFree(b->d_full);
        if (b->d_gate) (void)hipFree(b->d_gate);
        if (b->d_score) (void)hipFree(b->d_score);
        if (b->d_keep) (void)hipFree(b->d_keep);
    }
    b->d_chains = nullptr; b->d_chain_off = nullptr; b->d_fold_order = nullptr;
    b->d_full = b->d_gate = b->d_score = nullptr; b->d_keep = nullptr;
    b->own_chain_arrays = false;
}

// The counters of a stream batch, read once (after its planning kernels have run).
int stream_counters(rawdtw_ctx *ctx, rawdtw_batch *b)
{
    if (b->cnt_valid) return RAWDTW_OK;
    HIP_TRY(ctx, hipMemcpyAsync(b->h_cnt, b->sa.cnt, kStreamCounters * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    b->cnt_valid = true;
    b->dirty = false;
    return RAWDTW_OK;
}

// does the stream path's result stand?  (no invalid job, nothing over a capacity, no band it does not take)
bool stream_declined(const rawdtw_batch *b)
{
    const unsigned long long *c = b->h_cnt;
    return c[kCntBad] != ~0ull || c[kCntOverflow] != ~0ull || c[kCntUnsupported] != 0 || c[kCntOthers] > b->sa.others_cap;
}

// Redo a stream batch through the job-list path (which also words the error of an invalid batch).
int stream_fallback(rawdtw_ctx *ctx, rawdtw_batch *b)
{
    const uint64_t nc = b->n_chains;
    std::vector<uint64_t> job_off(nc + 1);
    uint64_t n_jobs = 0;
    int st = materialise_host_arrays(ctx, b);
    if (st != RAWDTW_OK) return st;
    st = rawdtw_batch_build_jobs(&b->opt, nc, b->in_anchor_off, b->in_anchors, b->in_ref_base, b->in_read_base,
                                     job_off.data(), nullptr, 0, &n_jobs);
    if (st != RAWDTW_OK) return fail(ctx, st, "job counting failed");
    b->stream = false; b->jobs_counted = true; // (batch_create_joblist sets n_jobs)
    ws_release(ctx, b->ws);
    b->h_cnt = nullptr; b->h_score = nullptr; b->h_keep = nullptr; b->res_bytes = 0; // (they lay in the workspace)
    b->d_chains = nullptr; b->d_chain_off = nullptr; b->d_fold_order = nullptr;
    b->d_full = b->d_gate = b->d_score = nullptr; b->d_keep = nullptr;
    st = batch_create_joblist(ctx, b, b->in_chain_off, b->in_anchor_off, b->in_anchors, b->in_ref_base, b->in_read_base, job_off, n_jobs);
    if (st != RAWDTW_OK) { batch_release_device(b); return st; }
    return rawdtw_batch_run(ctx, b);
}

} // namespace

// a batch whose (deferred) planning failed has neither form left: every entry point but destroy refuses it
static bool batch_dead(const rawdtw_batch *b) { return !b->stream && !b->plan; }

struct CompactIn {
    const rawdtw_anchor_t *heads, *unit_abs;
    const uint16_t *steps;
    const rawdtw_wide_step_t *wide;
    uint64_t n_wide;
};

static int batch_create_any(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                            const uint64_t *anchor_off, const rawdtw_anchor_t *anchors, const CompactIn *cin, const uint64_t *ref_base,
                            const uint32_t *read_base, rawdtw_batch **out, const rawdtw_batch *prev = nullptr, const rawdtw_carry_t *carry = nullptr,
                            const uint64_t *new_off = nullptr, const rawdtw_anchor_t *new_anchors = nullptr)
{
    if (!out) return RAWDTW_ERR_INVALID;
    *out = nullptr;
    if (!ctx || !opt || !chain_off || !anchor_off || (!anchors && !cin && !prev && n_reads) || !ref_base || !read_base)
        return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    if (cin && (!cin->heads || !cin->unit_abs || !cin->steps || (!cin->wide && cin->n_wide)))
        return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    if (opt->border_constraint != 0 && opt->border_constraint != 1)
        return fail(ctx, RAWDTW_ERR_INVALID, "invalid border constraint (rmap.cpp:301-304)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n_chains = chain_off[n_reads];
    int st = RAWDTW_OK;
    rawdtw_batch *b = new (std::nothrow) rawdtw_batch;
    if (!b) return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed");
    b->ctx = ctx; b->opt = *opt; b->n_reads = n_reads; b->n_chains = n_chains;
    ctx->live_batches.push_back(b);
    b->in_chain_off = chain_off; b->in_anchor_off = anchor_off; b->in_anchors = anchors; b->in_ref_base = ref_base; b->in_read_base = read_base;
    b->in_resident = ctx->resident_arrays && !cin;
    b->in_prev = prev; b->in_carry = carry; b->in_new_off = new_off;
    if (prev) {
        b->in_new_anchors = new_anchors; b->in_carried = true;
        for (uint64_t c = 0; c < n_chains; c++) b->parts_carried += carry[c].parts;
    }
    if (cin) { b->in_heads = cin->heads; b->in_unit_abs = cin->unit_abs; b->in_steps = cin->steps; b->in_wide = cin->wide; b->in_n_wide = cin->n_wide; }
    if (stream_eligible(ctx, opt, anchor_off[n_chains]))
        st = batch_create_stream(ctx, b, chain_off, anchor_off, anchors, ref_base, read_base);
    else {
        st = materialise_host_arrays(ctx, b);
        std::vector<uint64_t> &job_off = ctx->job_off_scratch; // (a context is not re-entrant)
        job_off.resize(n_chains + 1);
        uint64_t n_jobs = 0;
        if (st == RAWDTW_OK && rawdtw_batch_build_jobs(opt, n_chains, anchor_off, b->in_anchors, b->in_ref_base, b->in_read_base, job_off.data(),
                                                        nullptr, 0, &n_jobs) != RAWDTW_OK)
            st = fail(ctx, RAWDTW_ERR_INVALID, "job counting failed");
        if (st == RAWDTW_OK)
            st = batch_create_joblist(ctx, b, chain_off, anchor_off, b->in_anchors, b->in_ref_base, b->in_read_base, job_off, n_jobs);
    }
    b->in_prev = nullptr; // (read at create only)
    if (st != RAWDTW_OK) { rawdtw_batch_destroy(b); return st; }
    *out = b;
    return RAWDTW_OK;
}

int rawdtw_batch_create(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                        const uint64_t *anchor_off, const rawdtw_anchor_t *anchors, const uint64_t *ref_base,
                        const uint32_t *read_base, rawdtw_batch **out)
{
    return batch_create_any(ctx, opt, n_reads, chain_off, anchor_off, anchors, nullptr, ref_base, read_base, out);
}

static int batch_enqueue_one(rawdtw_ctx *ctx, rawdtw_batch *batch, hipEvent_t *e);

int rawdtw_batch_submit_compact(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                                const uint64_t *anchor_off, const rawdtw_anchor_t *heads, const rawdtw_anchor_t *unit_abs,
                                const uint16_t *steps, const rawdtw_wide_step_t *wide, uint64_t n_wide, const uint64_t *ref_base,
                                const uint32_t *read_base, rawdtw_batch **out)
{
    const CompactIn cin{heads, unit_abs, steps, wide, n_wide};
    if (ctx) ctx->in_submit = true;
    int st = batch_create_any(ctx, opt, n_reads, chain_off, anchor_off, nullptr, &cin, ref_base, read_base, out);
    if (ctx) ctx->in_submit = false;
    if (st != RAWDTW_OK) return st;
    st = batch_enqueue_one(ctx, *out, nullptr);
    if (st != RAWDTW_OK) { rawdtw_batch_destroy(*out); *out = nullptr; }
    return st;
}

// can `prev` serve as the previous batch of a chunk round with options `opt`?  (include/rawdtw.h)
int rawdtw_batch_can_carry(const rawdtw_ctx *ctx, const rawdtw_batch *prev, const rawdtw_align_opt_t *opt)
{
    if (!ctx || !prev || !opt || prev->ctx != ctx || !prev->stream || prev->stream_runs == 0) return 0;
    if (prev->cnt_valid && stream_declined(prev)) return 0;
    // (a part's radius, and with it its cost, follows from these; the fold's options may differ)
    if (prev->opt.border_constraint != opt->border_constraint || prev->opt.fill_method != opt->fill_method ||
        memcmp(&prev->opt.band_radius_frac, &opt->band_radius_frac, sizeof(float)) != 0)
        return 0;
    return 1; // (the kernel-selection options only decide which body scores a part: costs do not depend on them)
}

int rawdtw_batch_submit_carry(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                              const uint64_t *anchor_off, const rawdtw_anchor_t *anchors, const uint64_t *new_off, const rawdtw_anchor_t *new_anchors,
                              const uint64_t *ref_base, const uint32_t *read_base, const rawdtw_batch *prev, const rawdtw_carry_t *carry, rawdtw_batch **out)
{
    if (out) *out = nullptr;
    if (!ctx || !opt || !out || !chain_off || !anchor_off || !new_off || !carry || !prev) return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    const uint64_t nc = chain_off[n_reads];
    if (!rawdtw_batch_can_carry(ctx, prev, opt) || !stream_eligible(ctx, opt, new_off[nc]))
        return fail(ctx, RAWDTW_ERR_UNSUPPORTED, "the previous batch cannot serve this round (another context or options, never run, or not on the device-planned path): submit the round whole");
    if (new_off[nc] > anchor_off[nc] || (!new_anchors && new_off[nc])) return fail(ctx, RAWDTW_ERR_INVALID, "more new anchors than anchors");
    ctx->in_submit = true;
    int st = batch_create_any(ctx, opt, n_reads, chain_off, anchor_off, anchors, nullptr, ref_base, read_base, out, prev, carry, new_off, new_anchors);
    ctx->in_submit = false;
    if (st != RAWDTW_OK) return st;
    st = batch_enqueue_one(ctx, *out, nullptr);
    if (st != RAWDTW_OK) { rawdtw_batch_destroy(*out); *out = nullptr; }
    return st;
}

int rawdtw_batch_round_stats(rawdtw_ctx *ctx, rawdtw_batch *batch, uint64_t *parts_scored, uint64_t *parts_reused)
{
    if (!ctx || !batch || batch->ctx != ctx || batch_dead(batch)) return fail(ctx, RAWDTW_ERR_INVALID, "batch does not belong to this context");
    batch_count_jobs(batch);
    uint64_t reused = 0;
    if (batch->stream) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        const int st = stream_counters(ctx, batch);
        if (st != RAWDTW_OK) return st;
        if (!stream_declined(batch)) reused = batch->parts_carried;
    }
    if (parts_reused) *parts_reused = reused;
    if (parts_scored) *parts_scored = batch->n_jobs - reused;
    return RAWDTW_OK;
}

int rawdtw_batch_verify_plan(rawdtw_ctx *ctx, const rawdtw_batch *batch, const rawdtw_job_t *jobs, uint64_t n_jobs,
                             int *device_planned, char *message, uint32_t message_cap)
{
    auto say = [&](const std::string &m) { if (message && message_cap) snprintf(message, message_cap, "%s", m.c_str()); };
    say("");
    if (!ctx || !batch || batch->ctx != ctx || (n_jobs && !jobs) || batch_dead(batch)) return RAWDTW_ERR_INVALID;
    if (device_planned) *device_planned = batch->stream ? 1 : 0;
    batch_count_jobs(const_cast<rawdtw_batch *>(batch));
    if (n_jobs != batch->n_jobs) { say("job count differs from the batch's"); return RAWDTW_ERR_INVALID; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    auto S = [](uint64_t v) { return std::to_string(v); };
    std::string e;
    if (batch->stream) {
        // What the scan left behind for the DTW launch, against the job list the host builds from the same chains
        // (rawdtw_batch_build_jobs): every job either of the tile class by the class rule, then in exactly one pass's
        // records with its shape, radius, flag and windows, or in the side list exactly once with the job's windows,
        // shape, slanted radius and flag; the statistics.
        rawdtw_batch *mb = const_cast<rawdtw_batch *>(batch);
        int st = stream_counters(ctx, mb);
        if (st != RAWDTW_OK) return st;
        const StreamArgs &a = batch->sa;
        const unsigned long long *cnt = batch->h_cnt;
        if (stream_declined(batch)) {
            if (device_planned) *device_planned = 0;
            say("the stream path declined this batch (it is redone through the job list at fetch)");
            return RAWDTW_OK;
        }
        const uint64_t nc = batch->n_chains, na = a.n_anchors;
        const uint64_t *aoff = batch->in_anchor_off;
        // the work list: one entry a pass (checked below, once the jobs' classes are known)
        const uint64_t n_first = cnt[kCntTodo], n_pool = cnt[kCntPool], n_todo = n_first + n_pool;
        std::vector<uint4> todo(n_todo);
        if (n_first > a.n_tiles || n_pool > a.n_slots - a.n_tiles) e = "work list longer than the slots";
        else {
            if (n_first) HIP_TRY(ctx, hipMemcpy(todo.data(), a.todo, n_first * sizeof(uint4), hipMemcpyDeviceToHost));
            if (n_pool) HIP_TRY(ctx, hipMemcpy(todo.data() + n_first, a.todo + a.n_tiles, n_pool * sizeof(uint4), hipMemcpyDeviceToHost));
        }
        const uint64_t n_other = cnt[kCntOthers];
        std::vector<DevJob> oj(n_other);
        if (n_other) HIP_TRY(ctx, hipMemcpy(oj.data(), a.ojobs, n_other * sizeof(DevJob), hipMemcpyDeviceToHost));
        // job k of chain c's part p lives at anchor index a1 - 2 - p
        std::vector<uint64_t> slot_job(na, ~0ull);
        {
            uint64_t k = 0;
            for (uint64_t c = 0; c < nc; c++) {
                const uint64_t a0 = aoff[c], a1 = aoff[c + 1];
                for (uint64_t pidx = 0; a1 > a0 && pidx + 1 < a1 - a0; pidx++) slot_job[a1 - 2 - pidx] = k++;
            }
            if (k != n_jobs) e = "job count";
        }
        uint64_t tile_jobs = 0, tile_bytes = 0, other_bytes = 0;
        std::vector<uint8_t> is_tile(n_jobs, 0);
        for (uint64_t k = 0; k < n_jobs && e.empty(); k++) {
            const rawdtw_job_t &j = jobs[k];
            const int R = slanted_radius(j.n, j.m, j.band_radius);
            const uint32_t N = std::max(j.n, j.m);
            is_tile[k] = R <= a.lane_max_radius && N <= a.lane_max_n;
            tile_jobs += is_tile[k];
            (is_tile[k] ? tile_bytes : other_bytes) += 4ull * ((uint64_t)j.n + j.m) + 36ull;
        }
        std::vector<uint8_t> oseen(n_jobs, 0);
        for (uint64_t q = 0; q < n_other && e.empty(); q++) {
            const DevJob &d = oj[q];
            const uint64_t k = d.aux < na ? slot_job[d.aux] : ~0ull;
            if (k == ~0ull || oseen[k] || is_tile[k]) e = "side-list entry " + S(q) + " (anchor " + S(d.aux) + ") duplicated, of the tile class or no job at all";
            else if (d.n != jobs[k].n || d.m != jobs[k].m || d.ref_off != jobs[k].ref_off || d.read_off != jobs[k].read_off ||
                     d.R != slanted_radius(d.n, d.m, jobs[k].band_radius) || ((d.flags & kFlagExcludeLast) != 0) != (jobs[k].exclude_last != 0))
                e = "side-list record of job " + S(k) + " differs from the job";
            else oseen[k] = 1;
        }
        for (uint64_t k = 0; k < n_jobs && e.empty(); k++)
            if (!is_tile[k] && !oseen[k]) e = "job " + S(k) + " is in no launch";
        // Every pass: its records name tile-class jobs of its tile, each job once over all passes, with the job's shape,
        // slanted radius and flag, in the order the lanes take them (radius class, longer side); a record's windows lie in the
        // image, inside one of the pass's copy orders, and that order maps them onto the job's windows in the arenas.
        if (e.empty()) {
            std::vector<uint8_t> tseen(n_jobs, 0), slot_used(a.n_slots, 0);
            std::vector<uint2> recs(kStreamTile);
            std::vector<uint4> ords(2 * kStreamMaxSeg);
            for (uint64_t q = 0; q < n_todo && e.empty(); q++) {
                const uint4 t = todo[q];
                const uint32_t nj = t.z & 0xffffu, nr = t.z >> 16, region = t.w & 0xffffu, rec0 = t.w >> 16;
                if (t.x >= a.n_tiles || t.y >= a.n_slots || slot_used[t.y] || nj > kStreamTile || nr > kStreamMaxSeg || (nj && !nr) || (rec0 & 1u) || rec0 + nj > kStreamRecStride) {
                    e = "work list entry " + S(q) + ": tile " + S(t.x) + ", slot " + S(t.y) + ", " + S(nj) + " jobs, " + S(nr) + " runs"; break;
                }
                slot_used[t.y] = 1;
                if (!nj) continue;
                HIP_TRY(ctx, hipMemcpy(recs.data(), a.recs + (uint64_t)t.x * kStreamRecStride + rec0, nj * sizeof(uint2), hipMemcpyDeviceToHost));
                HIP_TRY(ctx, hipMemcpy(ords.data(), a.runtab + (uint64_t)t.y * 2 * kStreamMaxSeg, 2 * nr * sizeof(uint4), hipMemcpyDeviceToHost));
                for (uint32_t o = 0; o < 2 * nr && e.empty(); o++) {
                    const uint4 &od = ords[o];
                    const bool evs = (o & 1u) == 0;
                    if (od.x >= od.y || 4ull * od.y > a.lds_floats || (evs ? 4ull * od.y > region : 4ull * od.x < region))
                        e = "pass " + S(q) + " (tile " + S(t.x) + ", " + S(nj) + " jobs, " + S(nr) + " runs, event region " + S(region) + " of " + S(a.lds_floats) +
                            " floats): copy order " + S(o) + " = pieces [" + S(od.x) + ", " + S(od.y) + ") outside its region of the image";
                }
                uint32_t prev_bin = 0;
                for (uint32_t r = 0; r < nj && e.empty(); r++) {
                    const uint2 rc = recs[r];
                    const uint32_t N = rc.y & 127u, M = (rc.y >> 7) & 127u, R = (rc.y >> 14) & 3u, ex = (rc.y >> 16) & 1u, u = (rc.y >> 17) & (kStreamTile - 1u);
                    const uint64_t i = ((uint64_t)t.x + 1) * kStreamTile - 1 - u;
                    const uint64_t k = i < na ? slot_job[i] : ~0ull;
                    const std::string who = "pass " + S(q) + " record " + S(r) + " (anchor " + S(i) + ")";
                    if (k == ~0ull || !is_tile[k] || tseen[k]) { e = who + ": no job, not of the tile class, or in two passes"; break; }
                    const rawdtw_job_t &j = jobs[k];
                    const bool swap = j.n < j.m;
                    if (N != std::max(j.n, j.m) || M != std::min(j.n, j.m) || (int)R != slanted_radius(j.n, j.m, j.band_radius) || (ex != 0) != (j.exclude_last != 0)) {
                        e = who + ": shape, radius or flag differ from job " + S(k); break;
                    }
                    const uint32_t bin = (3u - R) * 64u + (63u - std::min(N, 63u));
                    if (bin < prev_bin) { e = who + ": out of the lanes' order"; break; }
                    prev_bin = bin;
                    const uint32_t p_long = rc.x & 0xffffu, p_short = rc.x >> 16;
                    const uint32_t p_ev = swap ? p_short : p_long, p_rf = swap ? p_long : p_short;
                    for (int w = 0; w < 2 && e.empty(); w++) {
                        const uint32_t pw = w ? p_rf : p_ev, len = w ? j.m : j.n;
                        const uint64_t want = w ? j.ref_off : (uint64_t)j.read_off;
                        bool ok = false;
                        for (uint32_t g = 0; g < nr && !ok; g++) {
                            const uint4 &od = ords[2 * g + w];
                            const long long src = (long long)((unsigned long long)od.z | ((unsigned long long)od.w << 32));
                            ok = 4ull * od.x <= pw && (uint64_t)pw + len <= 4ull * od.y && (long long)pw + src == (long long)want;
                        }
                        if (!ok) e = who + ": its " + (w ? "reference" : "event") + " window is in no copy order of the pass";
                    }
                    tseen[k] = 1;
                }
            }
            if (e.empty() && cnt[kCntReused] == 0) // (a round that took costs over leaves the carried parts out)
                for (uint64_t k = 0; k < n_jobs && e.empty(); k++)
                    if (is_tile[k] && !tseen[k]) e = "tile-class job " + S(k) + " is in no pass";
        }
        if (e.empty()) {
            HIP_TRY(ctx, stream_sum_stats(a, ctx->stream));
            unsigned long long st3[3];
            HIP_TRY(ctx, hipMemcpyAsync(st3, a.cnt + kCntTileJobs, 24, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (st3[0] != tile_jobs || st3[1] != tile_bytes || st3[2] != other_bytes) e = "tile statistics";
        }
        say(e);
        return e.empty() ? RAWDTW_OK : RAWDTW_ERR_DEVICE + 100;
    }
    const rawdtw_plan *pl = batch->plan;
    // the tile records as the kernels will read them
    const size_t n_tiles = pl->n_tiles + pl->n_tiles_hi;
    std::vector<TileDesc> tiles(n_tiles);
    std::vector<TileJob> tjobs(pl->n_tile_jobs);
    if (n_tiles) HIP_TRY(ctx, hipMemcpy(tiles.data(), pl->d_tiles, n_tiles * sizeof(TileDesc), hipMemcpyDeviceToHost));
    if (pl->n_tile_jobs) HIP_TRY(ctx, hipMemcpy(tjobs.data(), pl->d_tjobs, pl->n_tile_jobs * sizeof(TileJob), hipMemcpyDeviceToHost));
    size_t n_spans = 0;
    for (const TileDesc &t : tiles) n_spans = std::max<size_t>(n_spans, (size_t)t.span_first + (t.n_spans & 0x7fffffffu));
    std::vector<TileSpan> spans(n_spans);
    if (n_spans) HIP_TRY(ctx, hipMemcpy(spans.data(), pl->d_spans, n_spans * sizeof(TileSpan), hipMemcpyDeviceToHost));
    const PlanCfg cfg = cfg_of(ctx);
    std::vector<uint8_t> tseen;
    e = verify_tile_arrays(cfg, jobs, n_jobs, pl, tiles.data(), n_tiles, spans.data(), n_spans, tjobs.data(), tjobs.size(),
                           micro_masks(), tseen);
    // every job has exactly one home: a tile record or a record of another class
    std::vector<uint8_t> oseen(n_jobs, 0);
    if (e.empty()) {
        const uint64_t n_other = n_jobs - pl->n_tile_jobs;
        for (uint64_t q = 0; q < n_other && e.empty(); q++) {
            const DevJob &d = pl->h_jobs[pl->n_tile_jobs + q];
            const uint32_t k = d.aux;
            if (k >= n_jobs || oseen[k] || tseen[k]) e = "job " + S(k) + " planned twice";
            else if (d.n != jobs[k].n || d.m != jobs[k].m || d.ref_off != jobs[k].ref_off || d.read_off != jobs[k].read_off ||
                     ((d.flags & kFlagExcludeLast) != 0) != (jobs[k].exclude_last != 0))
                e = "record of job " + S(k) + " differs from the job";
            else oseen[k] = 1;
        }
        for (uint64_t k = 0; k < n_jobs && e.empty(); k++)
            if (!tseen[k] && !oseen[k]) e = "job " + S(k) + " is in no launch";
    }
    say(e);
    return e.empty() ? RAWDTW_OK : RAWDTW_ERR_DEVICE + 100;
}

int rawdtw_batch_info(const rawdtw_batch *batch, rawdtw_plan_info_t *info, uint64_t *n_chains)
{
    if (!batch || batch_dead(batch)) return RAWDTW_ERR_INVALID;
    if (n_chains) *n_chains = batch->n_chains;
    if (!info) return RAWDTW_OK;
    if (!batch->stream) return rawdtw_plan_info(batch->plan, info);
    rawdtw_batch *b = const_cast<rawdtw_batch *>(batch);
    rawdtw_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int st = stream_counters(ctx, b);
    if (st != RAWDTW_OK) return st;
    if (stream_declined(b)) { // what the job-list path will run
        st = stream_fallback(ctx, b);
        if (st != RAWDTW_OK) return st;
        return rawdtw_plan_info(b->plan, info);
    }
    if (!b->cells_counted) {
        HIP_TRY(ctx, stream_count_cells(b->sa, b->sa.cnt + kCntCells, ctx->stream));
        HIP_TRY(ctx, stream_sum_stats(b->sa, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&b->h_cnt[kCntCells], b->sa.cnt + kCntCells, 4 * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        b->cells_counted = true;
    }
    batch_count_jobs(b);
    const unsigned long long *c = b->h_cnt;
    rawdtw_plan_info_t I{};
    I.n_jobs = b->n_jobs;
    I.cells = c[kCntCells];
    I.algorithmic_bytes = c[kCntTileBytes] + c[kCntOtherBytes];
    unsigned long long side_lane = 0; // the side list's lane-per-job classes count with the tiles' jobs: same body, same class rule
    for (uint32_t q = kClsL0; q < kClsL0 + kClsLCount; q++) side_lane += c[kCntCls0 + q]; // (the 8-slot lane classes stay with the wide bands)
    I.n_lane_jobs = c[kCntTileJobs] + side_lane;
    I.n_wave_band_jobs = c[kCntOthers] - side_lane;
    I.n_full_jobs = 0;
    I.workspace_bytes = b->ws_bytes;
    I.n_launches = 1;
    *info = I;
    return RAWDTW_OK;
}

static int batch_tail(rawdtw_ctx *ctx, rawdtw_batch *b, int which)
{
    hipError_t e;
    if (ctx->debug_skip_kinds & (1u << (which == 0 ? kKindChainFold : kKindReadSelect))) return RAWDTW_OK;
    if (ctx->debug_skip_tail & (1u << which)) return RAWDTW_OK;
    const float *job_cost = b->stream ? b->sa.out_full : b->plan->d_cost;
    if (b->stream && b->fold_fused) {
        if (which == 1) return RAWDTW_OK; // (done by the launch before)
        StreamArgs f = b->sa; // (the fold walks the FULL lists: a chunk round's costs were gathered into out_full)
        f.anchor_off = b->sa.full_off; f.out = b->sa.out_full; f.n_anchors = b->sa.n_full;
        e = stream_gather(b->sa, b->d_chains, ctx->stream);
        if (e == hipSuccess) e = stream_fold_select(f, b->d_chains, b->d_chain_off, b->n_reads, b->opt.match_bonus, b->opt.fused_score, b->opt.min_score, b->d_full,
                               b->d_gate, b->d_score, b->d_keep, ctx->stream);
    } else if (which == 0)
        e = launch_chain_fold(std::min(ctx->fold_mode, 3), b->d_chains, b->d_fold_order, b->n_chains, job_cost, b->opt.match_bonus, b->opt.fused_score,
                              b->d_full, b->d_gate, ctx->fold_long_parts, ctx->stream);
    else
        e = launch_read_select(b->d_chain_off, b->n_reads, b->d_full, b->d_gate, b->opt.min_score, b->d_score,
                               b->d_keep, ctx->stream);
    if (e != hipSuccess) return hip_fail(ctx, e, which == 0 ? "chain fold launch" : "read select launch");
    return RAWDTW_OK;
}

// launches of a batch's DTW part (before fold and select): the job-list plan's, or the stream path's one
static uint32_t batch_dtw_launches(const rawdtw_batch *b) { return b->stream ? 2u : b->plan ? (uint32_t)b->plan->launches.size() : 0u; }


static int batch_enqueue_one(rawdtw_ctx *ctx, rawdtw_batch *batch, hipEvent_t *e)
{
    const uint32_t np = batch_dtw_launches(batch);
    int st = RAWDTW_OK;
    batch->dirty = true;
    if (batch->stream) {
        // The arenas may have been re-uploaded, grown or swapped since the batch was planned (rawdtw_upload_events,
        // rawdtw_events_reserve, rawdtw_upload_reference ... free and reallocate them): the launch reads the context's
        // CURRENT arrays, and the windows -- checked against the sizes at planning time -- must still lie inside them.
        if (ctx->n_ev < batch->sa.n_ev || ctx->n_ref < batch->sa.n_ref)
            return fail(ctx, RAWDTW_ERR_INVALID, "an arena shrank after the batch was created: create the batch again");
        batch->sa.ev = ctx->d_ev; batch->sa.ref = ctx->d_ref;
        // launch 0: the side list (k_wide) -- in line, or (option "wide_beside") forked onto the context's second stream and
        // joined before the fold; launch 1: the tiles' passes (k_runs)
        const bool wide = !(ctx->stream_debug & 4u);
        if (e && hipEventRecord(e[0], ctx->stream) != hipSuccess) st = RAWDTW_ERR_DEVICE;
        const bool wide_now = wide && !batch->wide_out; // (the first run's went out with the planning launches)
        if (st == RAWDTW_OK && wide_now && ctx->wide_order != 2) {
            const hipError_t he = stream_wide_fork(ctx, batch->sa);
            if (he != hipSuccess) st = hip_fail(ctx, he, "side list launch");
        }
        batch->wide_out = false;
        if (st == RAWDTW_OK && e && hipEventRecord(e[1], ctx->stream) != hipSuccess) st = RAWDTW_ERR_DEVICE;
        if (st == RAWDTW_OK && e && hipEventRecord(e[2], ctx->stream) != hipSuccess) st = RAWDTW_ERR_DEVICE;
        if (st == RAWDTW_OK) {
            hipError_t he = stream_run(batch->sa, ctx->stream_blocks, batch->stream_lds, batch->stream_threads, batch->stream_runs++ > 0, ctx->stream);
            if (he != hipSuccess) st = hip_fail(ctx, he, "batch kernel launch");
        }
        if (st == RAWDTW_OK && e && hipEventRecord(e[3], ctx->stream) != hipSuccess) st = RAWDTW_ERR_DEVICE;
        if (st == RAWDTW_OK && wide_now && ctx->wide_order == 2) { // (timing experiments: the side list behind the tiles)
            const hipError_t he = stream_wide_fork(ctx, batch->sa);
            if (he != hipSuccess) st = hip_fail(ctx, he, "side list launch");
        }
        if (st == RAWDTW_OK && wide && ctx->wide_beside && hipStreamWaitEvent(ctx->stream, ctx->ev_wide_join, 0) != hipSuccess) st = RAWDTW_ERR_DEVICE;
    } else st = run_all_launches(ctx, batch->plan, e);
    for (int k = 0; k < 2 && st == RAWDTW_OK; k++) {
        if (e && hipEventRecord(e[2 * (np + k)], ctx->stream) != hipSuccess) st = RAWDTW_ERR_DEVICE;
        if (st == RAWDTW_OK) st = batch_tail(ctx, batch, k);
        if (st == RAWDTW_OK && e && hipEventRecord(e[2 * (np + k) + 1], ctx->stream) != hipSuccess) st = RAWDTW_ERR_DEVICE;
    }
    return st;
}

int rawdtw_batch_run(rawdtw_ctx *ctx, rawdtw_batch *batch)
{
    if (!ctx || !batch || batch->ctx != ctx || batch_dead(batch)) return fail(ctx, RAWDTW_ERR_INVALID, "batch does not belong to this context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return batch_enqueue_one(ctx, batch, nullptr);
}

int rawdtw_batch_run_timed(rawdtw_ctx *ctx, rawdtw_batch *batch, float *launch_ms, uint32_t *launch_kind, uint32_t cap,
                           uint32_t *n_launches)
{
    std::vector<float> tmp(64, 0.f);
    int st = rawdtw_batch_run_reps(ctx, batch, 1, launch_ms ? launch_ms : tmp.data(), launch_kind,
                                   launch_ms ? cap : 64, n_launches);
    return st;
}

int rawdtw_batch_enqueue(rawdtw_ctx *ctx, rawdtw_batch *batch, int timed)
{
    if (!ctx || !batch || batch->ctx != ctx || batch_dead(batch)) return fail(ctx, RAWDTW_ERR_INVALID, "batch does not belong to this context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t nl = batch_dtw_launches(batch) + 2;
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
    const uint32_t np = batch_dtw_launches(batch), nl = np + 2;
    if (n_launches) *n_launches = nl;
    if (n_runs) *n_runs = batch->ev_runs;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    batch->dirty = false;
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
        if (launch_kind) {
            if (i >= np) launch_kind[i] = i == np ? kKindChainFold : kKindReadSelect;
            else if (batch->stream) launch_kind[i] = i == 0 ? (uint32_t)kKindBandWreg : (kKindBandMerged | ((uint32_t)batch->stream_lds << 8));
            else launch_kind[i] = batch->plan->launches[i].kind | ((uint32_t)batch->plan->launches[i].param << 8);
        }
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
    if (n_launches) *n_launches = batch_dtw_launches(batch) + 2;
    int st = RAWDTW_OK;
    for (uint32_t r = 0; r < reps && st == RAWDTW_OK; r++) st = rawdtw_batch_enqueue(ctx, batch, launch_ms != nullptr);
    if (launch_ms) {
        int st2 = rawdtw_batch_collect(ctx, batch, launch_ms, launch_kind, cap, nullptr, nullptr);
        if (st == RAWDTW_OK) st = st2;
    } else {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && st == RAWDTW_OK) st = RAWDTW_ERR_DEVICE;
        batch->dirty = false;
    }
    return st;
}

int rawdtw_batch_launch_stats(const rawdtw_batch *batch, uint32_t i, uint32_t *kind, int32_t *param, uint64_t *n_jobs,
                              uint64_t *algorithmic_bytes, uint64_t *cells)
{
    if (!batch || batch_dead(batch)) return RAWDTW_ERR_INVALID;
    const uint32_t nl = batch_dtw_launches(batch);
    if (i >= nl + 2) return RAWDTW_ERR_INVALID;
    if (i >= nl) {
        if (kind) *kind = i == nl ? kKindChainFold : kKindReadSelect;
        if (param) *param = 0;
        batch_count_jobs(const_cast<rawdtw_batch *>(batch));
        if (n_jobs) *n_jobs = i == nl ? batch->n_chains : batch->n_reads;
        // fold: one 4-byte cost per job + a 24-byte descriptor and two 4-byte results per chain;
        // select: 8 bytes read and 5 written per chain
        if (algorithmic_bytes)
            *algorithmic_bytes = i == nl ? batch->n_jobs * 4 + batch->n_chains * 32 : batch->n_chains * 13 + batch->n_reads * 8;
        if (cells) *cells = 0;
        return RAWDTW_OK;
    }
    if (batch->stream) {
        rawdtw_plan_info_t I{};
        if (cells) { int st = rawdtw_batch_info(batch, &I, nullptr); if (st != RAWDTW_OK) return st; }
        else {
            rawdtw_batch *b = const_cast<rawdtw_batch *>(batch);
            int st = stream_counters(b->ctx, b);
            if (st != RAWDTW_OK) return st;
            if (!b->cells_counted) {
                if (stream_sum_stats(b->sa, b->ctx->stream) != hipSuccess ||
                    hipMemcpyAsync(&b->h_cnt[kCntTileJobs], b->sa.cnt + kCntTileJobs, 3 * 8, hipMemcpyDeviceToHost, b->ctx->stream) != hipSuccess ||
                    hipStreamSynchronize(b->ctx->stream) != hipSuccess) return RAWDTW_ERR_DEVICE;
            }
            I.algorithmic_bytes = b->h_cnt[kCntTileBytes] + b->h_cnt[kCntOtherBytes];
        }
        if (batch->stream) { // (rawdtw_batch_info may have moved the batch to the job-list path)
            if (kind) *kind = kKindBandMerged;
            if (param) *param = (int32_t)batch->stream_lds;
            batch_count_jobs(const_cast<rawdtw_batch *>(batch));
            if (n_jobs) *n_jobs = batch->n_jobs;
            if (algorithmic_bytes) *algorithmic_bytes = I.algorithmic_bytes;
            if (cells) *cells = I.cells;
            return RAWDTW_OK;
        }
        if (i >= batch_dtw_launches(batch)) return RAWDTW_ERR_INVALID;
    }
    const rawdtw_plan *pl = batch->plan;
    const Launch &L = pl->launches[i];
    const MergeSel mg = merge_of(batch->ctx, pl);
    uint64_t bytes = 0, cl = 0, nj = 0;
    auto add = [&](const Launch &X) {
        for (uint64_t p = X.first; p < X.first + X.count; p++) {
            const DevJob &d = pl->h_jobs[p];
            bytes += 4ull * ((uint64_t)d.n + d.m) + 4 + 32;
        }
        if (cells) cl += count_cells(pl, X.first, X.first + X.count);
        nj += X.count;
    };
    uint32_t k = L.kind;
    if (mg.on() && (int)i == mg.tile) { // the merged launch reports the three classes it carries
        k = kKindBandMerged;
        add(L);
        if (mg.grp16 >= 0) add(pl->launches[mg.grp16]);
        if (mg.grp8 >= 0) add(pl->launches[mg.grp8]);
        if (mg.wreg >= 0) add(pl->launches[mg.wreg]);
    } else if (mg.on() && ((int)i == mg.grp16 || (int)i == mg.grp8 || (int)i == mg.wreg)) {
        /* folded into the merged launch: nothing of its own */
    } else add(L);
    if (kind) *kind = k;
    if (param) *param = L.param;
    if (n_jobs) *n_jobs = nj;
    if (algorithmic_bytes) *algorithmic_bytes = bytes;
    if (cells) *cells = cl;
    return RAWDTW_OK;
}

int rawdtw_batch_fetch(rawdtw_ctx *ctx, rawdtw_batch *batch, float *score, uint8_t *keep, float *job_cost)
{
    if (!ctx || !batch || batch->ctx != ctx || batch_dead(batch)) return fail(ctx, RAWDTW_ERR_INVALID, "batch does not belong to this context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int attempt = 0; attempt < 2; attempt++) {
        const float *d_cost = batch->stream ? batch->sa.out_full : batch->plan->d_cost;
        // a sync-free batch: counters, scores and keep flags in one copy into the batch's pinned block, and from there into the
        // caller's arrays (200 KB of host copying against two more operations on the stream)
        const bool block = batch->stream && batch->n_chains && (score || keep);
        if (block) HIP_TRY(ctx, hipMemcpyAsync(batch->h_cnt, batch->sa.cnt, batch->res_bytes, hipMemcpyDeviceToHost, ctx->stream));
        else if (batch->n_chains) {
            if (score) HIP_TRY(ctx, hipMemcpyAsync(score, batch->d_score, batch->n_chains * 4, hipMemcpyDeviceToHost, ctx->stream));
            if (keep) HIP_TRY(ctx, hipMemcpyAsync(keep, batch->d_keep, batch->n_chains, hipMemcpyDeviceToHost, ctx->stream));
        }
        // (a sync-free batch keeps one cost per ANCHOR: the part that ends there; they are put into job order below)
        std::vector<float> per_anchor;
        if (job_cost && batch->stream && batch->sa.n_full) {
            try { per_anchor.resize(batch->sa.n_full); } catch (const std::bad_alloc &) { return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed"); }
            HIP_TRY(ctx, hipMemcpyAsync(per_anchor.data(), d_cost, batch->sa.n_full * 4, hipMemcpyDeviceToHost, ctx->stream));
        } else if (job_cost && !batch->stream && batch->n_jobs)
            HIP_TRY(ctx, hipMemcpyAsync(job_cost, d_cost, batch->n_jobs * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (batch->stream && !batch->cnt_valid && !block)
            HIP_TRY(ctx, hipMemcpyAsync(batch->h_cnt, batch->sa.cnt, kStreamCounters * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        batch->dirty = false;
        if (!batch->stream) return RAWDTW_OK;
        batch->cnt_valid = true;
        if (!stream_declined(batch)) {
            if (block) {
                if (score) memcpy(score, batch->h_score, batch->n_chains * 4);
                if (keep) memcpy(keep, batch->h_keep, batch->n_chains);
            }
            if (job_cost) { // chain c's part p (rmap.cpp:248-293) ends at anchor a1 - 2 - p
                const uint64_t *aoff = batch->in_anchor_off;
                uint64_t k = 0;
                for (uint64_t c = 0; c < batch->n_chains; c++) {
                    const uint64_t a0 = aoff[c], a1 = aoff[c + 1];
                    for (uint64_t pidx = 0; a1 > a0 && pidx + 1 < a1 - a0; pidx++) job_cost[k++] = per_anchor[a1 - 2 - pidx];
                }
            }
            return RAWDTW_OK;
        }
        int st = stream_fallback(ctx, batch); // invalid anchors (the job-list path words the error) or a shape it does not take
        if (st != RAWDTW_OK) return st;
    }
    return RAWDTW_OK;
}

int rawdtw_batch_plan_ms(rawdtw_ctx *ctx, rawdtw_batch *batch, float *ms)
{
    if (!ctx || !batch || batch->ctx != ctx || !ms) return fail(ctx, RAWDTW_ERR_INVALID, "bad arguments to batch_plan_ms");
    *ms = 0.0f;
    if (!batch->stream || !batch->ev_plan[0] || !batch->ev_plan[3]) return RAWDTW_OK;
    HIP_TRY(ctx, hipEventSynchronize(batch->ev_plan[3]));
    float scan = 0.f, plan = 0.f; // (the side list's launch between them is DTW work: rawdtw_batch_wide_ms)
    HIP_TRY(ctx, hipEventElapsedTime(&scan, batch->ev_plan[0], batch->ev_plan[1]));
    HIP_TRY(ctx, hipEventElapsedTime(&plan, batch->ev_plan[2], batch->ev_plan[3]));
    *ms = scan + plan;
    return RAWDTW_OK;
}

int rawdtw_batch_wide_ms(rawdtw_ctx *ctx, rawdtw_batch *batch, float *ms)
{
    if (!ctx || !batch || batch->ctx != ctx || !ms) return fail(ctx, RAWDTW_ERR_INVALID, "bad arguments to batch_wide_ms");
    *ms = 0.0f;
    if (!batch->stream || !batch->ev_plan[1] || !batch->ev_plan[2]) return RAWDTW_OK;
    HIP_TRY(ctx, hipEventSynchronize(batch->ev_plan[2]));
    HIP_TRY(ctx, hipEventElapsedTime(ms, batch->ev_plan[1], batch->ev_plan[2]));
    return RAWDTW_OK;
}

int rawdtw_batch_stream_counter_index(const char *name)
{
    static const struct { const char *name; int index; } table[] = {
        {"bad", kCntBad}, {"overflow", kCntOverflow}, {"unsupported", kCntUnsupported}, {"side_jobs", kCntOthers}, {"class0", kCntCls0},
        {"cells", kCntCells}, {"tile_jobs", kCntTileJobs}, {"tile_bytes", kCntTileBytes}, {"side_bytes", kCntOtherBytes}, {"todo", kCntTodo},
        {"reused", kCntReused}, {"pool", kCntPool}, {"stamp0", kCntStamp0}};
    if (!name) return -1;
    for (const auto &t : table) if (strcmp(name, t.name) == 0) return t.index;
    return -1;
}

int rawdtw_batch_stream_counters(rawdtw_ctx *ctx, rawdtw_batch *batch, uint64_t *out, uint32_t cap, uint32_t *n_out)
{
    if (!ctx || !batch || batch->ctx != ctx || !n_out) return fail(ctx, RAWDTW_ERR_INVALID, "bad arguments to batch_stream_counters");
    *n_out = 0;
    if (!batch->stream) return RAWDTW_OK;
    batch->cnt_valid = false; // (a diagnostic call: runs since the last look have moved the phase stamps on)
    const int st = stream_counters(ctx, batch);
    if (st != RAWDTW_OK) return st;
    const uint32_t n = (uint32_t)kCntHeads; // (the queue heads behind them are the kernel's scratch)
    *n_out = n;
    for (uint32_t i = 0; i < n && i < cap && out; i++) out[i] = batch->h_cnt[i];
    return RAWDTW_OK;
}

int rawdtw_traceback_timing(const rawdtw_ctx *ctx, float *fill_ms, float *walk_ms, uint64_t *direction_bytes, uint64_t *path_elements)
{
    if (!ctx) return RAWDTW_ERR_INVALID;
    if (fill_ms) *fill_ms = ctx->tb_fill_ms;
    if (walk_ms) *walk_ms = ctx->tb_walk_ms;
    if (direction_bytes) *direction_bytes = ctx->tb_dir_written;
    if (path_elements) *path_elements = ctx->tb_path_elems;
    return RAWDTW_OK;
}

int rawdtw_batch_submit(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                        const uint64_t *anchor_off, const rawdtw_anchor_t *anchors, const uint64_t *ref_base,
                        const uint32_t *read_base, rawdtw_batch **out)
{
    if (ctx) ctx->in_submit = true;
    int st = rawdtw_batch_create(ctx, opt, n_reads, chain_off, anchor_off, anchors, ref_base, read_base, out);
    if (ctx) ctx->in_submit = false;
    if (st != RAWDTW_OK) return st;
    st = batch_enqueue_one(ctx, *out, nullptr);
    if (st != RAWDTW_OK) { rawdtw_batch_destroy(*out); *out = nullptr; }
    return st;
}

int rawdtw_batch_fetch_destroy(rawdtw_ctx *ctx, rawdtw_batch *batch, float *score, uint8_t *keep)
{
    const int st = rawdtw_batch_fetch(ctx, batch, score, keep, nullptr);
    if (batch && batch->ctx == ctx) rawdtw_batch_destroy(batch); // (a batch of another context is the caller's mistake, not ours to free)
    return st;
}

// everything of a batch that lives on the device or in its context's pools; `ctx` = the batch's context
static void batch_detach(rawdtw_ctx *ctx, rawdtw_batch *b)
{
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : b->ev_plan) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    b->ev.clear(); b->ev_runs = 0;
    batch_release_device(b); // (also destroys the job-list plan, which unregisters itself)
    ws_release(ctx, b->ws);
    b->stream = false;       // neither form left: every entry point but destroy refuses the batch (batch_dead)
}

int rawdtw_batch_destroy(rawdtw_batch *b)
{
    if (!b) return RAWDTW_OK;
    if (rawdtw_ctx *ctx = b->ctx) { // (null: rawdtw_destroy came first and took the device side with it)
        (void)hipSetDevice(ctx->device);
        if (b->dirty) (void)hipStreamSynchronize(ctx->stream); // its workspace goes back to the pool
        batch_detach(ctx, b);
        unregister(ctx->live_batches, b);
    }
    delete b;
    return RAWDTW_OK;
}

// ---- pinned host memory and incremental event upload ------------------------------------------------
int rawdtw_host_alloc(uint64_t bytes, void **out)
{
    if (!out) return RAWDTW_ERR_INVALID;
    *out = nullptr;
    if (bytes == 0) return RAWDTW_OK;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? RAWDTW_OK : RAWDTW_ERR_OOM;
}

int rawdtw_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
    return RAWDTW_OK;
}

int rawdtw_events_reserve(rawdtw_ctx *ctx, uint64_t n_floats)
{
    if (!ctx) return RAWDTW_ERR_INVALID;
    if (n_floats >= (1ull << 32)) return fail(ctx, RAWDTW_ERR_INVALID, "event arena limited to 2^32-1 floats per batch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->own_ev) { ctx->d_ev = nullptr; ctx->cap_ev = 0; ctx->n_ev = 0; ctx->own_ev = true; }
    if (ctx->cap_ev < n_floats) { // grow, keeping what is there
        uint64_t cap = std::max<uint64_t>(n_floats + (n_floats >> 2), 1024);
        cap = (cap + 63) & ~63ull;
        float *nw = nullptr;
        int st = dev_alloc(ctx, &nw, cap);
        if (st != RAWDTW_OK) return st;
        if (ctx->d_ev && ctx->n_ev) HIP_TRY(ctx, hipMemcpyAsync(nw, ctx->d_ev, ctx->n_ev * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_ev) (void)hipFree(ctx->d_ev);
        ctx->d_ev = nw;
        ctx->cap_ev = cap;
    }
    ctx->n_ev = std::max(ctx->n_ev, n_floats);
    return RAWDTW_OK;
}

int rawdtw_events_append(rawdtw_ctx *ctx, const float *h_new, uint64_t n_new, uint32_t n_segments,
                         const uint64_t *seg_src_off, const uint32_t *seg_dst_off)
{
    if (!ctx || (n_new && !h_new) || (n_segments && (!seg_src_off || !seg_dst_off))) return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    if (n_segments == 0) return RAWDTW_OK;
    if (!ctx->own_ev || !ctx->d_ev) return fail(ctx, RAWDTW_ERR_INVALID, "rawdtw_events_reserve first");
    if (seg_src_off[n_segments] > n_new) return fail(ctx, RAWDTW_ERR_RANGE, "segment sources beyond the new events");
    for (uint32_t q = 0; q < n_segments; q++)
        if (seg_src_off[q + 1] < seg_src_off[q] || (uint64_t)seg_dst_off[q] + (seg_src_off[q + 1] - seg_src_off[q]) > ctx->n_ev)
            return fail(ctx, RAWDTW_ERR_RANGE, "segment outside the reserved event arena");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // staging on the device: the round's events and the two segment tables (grow-only)
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t need = al(n_new * 4) + al(((size_t)n_segments + 1) * 8) + al((size_t)n_segments * 4);
    if (ctx->append_bytes < need) {
        if (ctx->d_append) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_append); }
        ctx->d_append = nullptr; ctx->append_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_append, need + need / 4));
        ctx->append_bytes = need + need / 4;
    }
    char *p = static_cast<char *>(ctx->d_append);
    float *d_new = carve<float>(p, n_new);
    uint64_t *d_src = carve<uint64_t>(p, (uint64_t)n_segments + 1);
    uint32_t *d_dst = carve<uint32_t>(p, n_segments);
    if (n_new) HIP_TRY(ctx, hipMemcpyAsync(d_new, h_new, n_new * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_src, seg_src_off, ((size_t)n_segments + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, seg_dst_off, (size_t)n_segments * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, launch_events_scatter(d_new, ctx->d_ev, d_src, d_dst, n_segments, ctx->stream));
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
    drop_reference(ctx);
    ctx->ref_off.assign(2ull * n_seq, 0);
    ctx->ref_len = idx->lens;
    uint64_t total = 0;
    for (uint32_t s = 0; s < n_seq; s++) {
        ctx->ref_off[2 * s] = total; total += ((uint64_t)idx->lens[s] + 3) & ~3ull;
        ctx->ref_off[2 * s + 1] = total; total += ((uint64_t)idx->lens[s] + 3) & ~3ull;
    }
    int st = dev_alloc(ctx, &ctx->d_ref, std::max<uint64_t>(total, 4));
    if (st != RAWDTW_OK) return st;
    ctx->ref_hold = new (std::nothrow) RefHold;
    if (!ctx->ref_hold) { (void)hipFree(ctx->d_ref); ctx->d_ref = nullptr; return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed"); }
    ctx->ref_hold->d = ctx->d_ref;
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
