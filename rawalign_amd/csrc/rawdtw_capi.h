// rawdtw_capi.h -- what the translation units behind the C ABI (include/rawdtw.h) share: the records behind the opaque
// handles, small helpers, and the planner's entry points.  Internal: nothing here is part of the ABI.
//   rawdtw_capi.cpp       contexts, options, arenas, pinned memory, incremental event upload
//   rawdtw_planner.cpp    the host planner of the job-list path, its launch sequences, rawdtw_plan_*, rawdtw_score_batch
//   rawdtw_batch.cpp      candidate batches: the stream path's set-up (sync-free, planned on the device), the job-list form,
//                         run / fetch / diagnostics, chunk rounds
//   rawdtw_traceback.cpp  rawdtw_traceback_batch*, the single-call drop-ins
//   rawdtw_index.cpp      the .ind reader
// There is NO CPU fallback in any of them: every scoring entry point runs the HIP kernels or returns an error status.
#pragma once
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
    struct rawdtw_chain_ws *chain_ws = nullptr; // rawdtw_chain_round's device block (rawdtw_chain.hip), grow-only
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

namespace rawdtw {
namespace capi {


// the next `count` elements of a 256-byte aligned block
template <typename T> T *carve(char *&p, uint64_t count)
{
    T *q = reinterpret_cast<T *>(p);
    p += (count * sizeof(T) + 255) & ~(size_t)255;
    return q;
}

inline int fail(rawdtw_ctx *ctx, int status, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return status;
}

inline int hip_fail(rawdtw_ctx *ctx, hipError_t e, const char *what)
{
    int st = (e == hipErrorOutOfMemory) ? RAWDTW_ERR_OOM : RAWDTW_ERR_DEVICE;
    return fail(ctx, st, std::string(what) + ": " + hipGetErrorString(e));
}

void chain_ws_free(rawdtw_ctx *ctx); // rawdtw_chain.hip

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
inline uint64_t banded_cells(uint32_t n, uint32_t m, int R)
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
inline uint64_t band_mask8(uint32_t N, uint32_t M, int R)
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
inline void drop_reference(rawdtw_ctx *ctx)
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

inline int ensure_events_capacity(rawdtw_ctx *ctx, uint64_t n)
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

// ---- the planner's entry points (rawdtw_planner.cpp) ----
struct PlanCfg;
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
int build_plan(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, bool traceback, rawdtw_plan **out);
uint64_t count_cells(const rawdtw_plan *pl, uint64_t p0, uint64_t p1);
struct MergeSel { int tile = -1, grp16 = -1, grp8 = -1, wreg = -1; bool on() const { return tile >= 0; } };
MergeSel merge_of(const rawdtw_ctx *ctx, const rawdtw_plan *pl);
int run_all_launches(rawdtw_ctx *ctx, rawdtw_plan *pl, hipEvent_t *ev);
void plan_release_device(rawdtw_plan *plan);
// the tile records as downloaded from the device against the job list (rawdtw_batch_verify_plan): "" or what is wrong
std::string verify_uploaded_tiles(const rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs, const rawdtw_plan *pl, const TileDesc *tiles,
                                  size_t n_tiles, const TileSpan *spans, size_t n_spans, const TileJob *tjobs, size_t n_tjobs, std::vector<uint8_t> &seen);
// ---- rawdtw_batch.cpp ----
void batch_detach(rawdtw_ctx *ctx, rawdtw_batch *b);

} // namespace capi
} // namespace rawdtw
