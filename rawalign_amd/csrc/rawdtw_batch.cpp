// rawdtw_batch.cpp -- candidate batches behind rawdtw_batch_* (include/rawdtw.h): the DTW block of gen_chains
// (src/rmap.cpp:509-530) for every read of a mini-batch in one submission.  The stream path's set-up (sparse + banded
// batches: everything enqueued, planned on the device -- kernels in rawdtw_runs.hip), the job-list form for the other
// modes and for what the stream path declines, run / fetch / diagnostics, and chunk rounds (rawdtw_batch_submit_carry).
// Host code only.
#include "rawdtw_capi.h"

using namespace rawdtw;
using namespace rawdtw::capi;

extern "C" {

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
    try { b->host_anchors.resize(na + 1); b->host_ref_base.resize(nc + 1); b->host_read_base.resize(nc + 1); } // (+ 1: non-null arrays for a round without chains)
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
    std::vector<uint8_t> tseen;
    e = verify_uploaded_tiles(ctx, jobs, n_jobs, pl, tiles.data(), n_tiles, spans.data(), n_spans, tjobs.data(), tjobs.size(), tseen);
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

int rawdtw_batch_submit_device(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                               const uint64_t *anchor_off, const rawdtw_anchor_t *d_anchors, const uint64_t *d_ref_base,
                               const uint32_t *d_read_base, rawdtw_batch **out)
{
    if (!ctx) return RAWDTW_ERR_INVALID;
    const bool was = ctx->resident_arrays; // (the option's meaning, for this one batch: a batch keeps the form it was created under)
    ctx->resident_arrays = true;
    const int st = rawdtw_batch_submit(ctx, opt, n_reads, chain_off, anchor_off, d_anchors, d_ref_base, d_read_base, out);
    ctx->resident_arrays = was;
    return st;
}

int rawdtw_batch_fetch_destroy(rawdtw_ctx *ctx, rawdtw_batch *batch, float *score, uint8_t *keep)
{
    const int st = rawdtw_batch_fetch(ctx, batch, score, keep, nullptr);
    if (batch && batch->ctx == ctx) rawdtw_batch_destroy(batch); // (a batch of another context is the caller's mistake, not ours to free)
    return st;
}

// everything of a batch that lives on the device or in its context's pools; `ctx` = the batch's context
} // extern "C"
namespace rawdtw { namespace capi {
void batch_detach(rawdtw_ctx *ctx, rawdtw_batch *b)
{
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : b->ev_plan) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    b->ev.clear(); b->ev_runs = 0;
    batch_release_device(b); // (also destroys the job-list plan, which unregisters itself)
    ws_release(ctx, b->ws);
    b->stream = false;       // neither form left: every entry point but destroy refuses the batch (batch_dead)
}
} } // namespace rawdtw::capi
extern "C" {

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

} // extern "C"
