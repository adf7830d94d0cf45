// rawdtw_traceback.cpp -- DTW_global_tb for batches (rawdtw_traceback_batch*: dtw.hpp:28 at rmap.cpp:221,284) and the
// single-call drop-ins with the reference's own signatures (dtw.hpp:21,25,28).  Host code only.
#include "rawdtw_capi.h"

using namespace rawdtw;
using namespace rawdtw::capi;

extern "C" {

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

int rawdtw_traceback_timing(const rawdtw_ctx *ctx, float *fill_ms, float *walk_ms, uint64_t *direction_bytes, uint64_t *path_elements)
{
    if (!ctx) return RAWDTW_ERR_INVALID;
    if (fill_ms) *fill_ms = ctx->tb_fill_ms;
    if (walk_ms) *walk_ms = ctx->tb_walk_ms;
    if (direction_bytes) *direction_bytes = ctx->tb_dir_written;
    if (path_elements) *path_elements = ctx->tb_path_elems;
    return RAWDTW_OK;
}

} // extern "C"
