// rawdtw_capi.cpp -- contexts behind the C ABI of librawdtw.so (include/rawdtw.h): lifetime, options, the reference and event
// arenas, pinned host memory, the incremental event upload.  Host code only; see rawdtw_capi.h for the other units.
//
// There is NO CPU fallback in here: every scoring entry point runs the HIP kernels or returns an error status.
#include "rawdtw_capi.h"

using namespace rawdtw;
using namespace rawdtw::capi;

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
    chain_ws_free(ctx);
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

// ---- pinned host memory and incremental event upload ------------------------------------------------
int rawdtw_host_alloc(uint64_t bytes, void **out)
{
    if (!out) return RAWDTW_ERR_INVALID;
    *out = nullptr;
    if (bytes == 0) return RAWDTW_OK;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? RAWDTW_OK : RAWDTW_ERR_OOM;
}

int rawdtw_host_is_page_locked(const void *p)
{
    if (!p) return 0;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return at.type == hipMemoryTypeHost ? 1 : 0;
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

} // extern "C"
