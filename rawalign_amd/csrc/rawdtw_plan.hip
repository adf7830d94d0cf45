// rawdtw_plan.hip -- device-side planning of a candidate batch (rawdtw_batch_create, "device_plan").
//
// The host planner (rawdtw_capi.cpp: plan_host) walks every job; at 5 M jobs per mini-batch that is tens of
// milliseconds against 0.15 ms of kernels.  Everything the bulk (tile) class needs is a pure function of the
// chains' anchor lists, so it is computed here, on the device, from the uploaded anchors:
//
//   k_plan_jobs     one thread per job: window, radius after the slant correction (dtw.cpp:298-300), class, and the
//                   LDS floats the job adds to its tile (a part that continues its chain's run shares the anchor
//                   element with its predecessor: n + m - 2; a run start pays both windows plus alignment)
//   scans           tile-class rank, running LDS cost -> tile number (a tile = the jobs whose running cost falls in one
//                   budget-sized bracket), run number (a run = consecutive tile jobs of one chain inside one tile;
//                   it becomes one span of each arena)
//   k_plan_scatter  rank -> job index, tile starts, run starts; the few jobs of the other classes are compacted for
//                   the host planner (they need sorting by shape, and there are ~25 k of them)
//   k_plan_tiles    one workgroup per tile: spans (two per run), LDS offsets, the tile's 16-byte job records sorted by
//                   (kind, longer side, shorter side) exactly as the host planner stores them
//
// The records written are the ones k_band_tile / k_band_merged read (TileDesc, TileSpan, TileJob); costs do not depend
// on how jobs are tiled, so host- and device-planned batches give bit-identical results (tests/test_gpu_parity.py).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <type_traits>

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "rawdtw_internal.h"

namespace rawdtw {

namespace {

constexpr int kPlanThreads = 256;
constexpr int kTileItems = 8;                              // records per thread in k_plan_tiles
constexpr uint32_t kPlanMaxTileJobs = kPlanThreads * kTileItems; // 2048
constexpr uint32_t kPlanMaxRuns = 768;                     // runs (chains or chain fragments) per tile.  A run start costs at
                                                           // least 8 floats of the tile's budget, so a 5120-float tile holds
                                                           // at most 640 (the host side refuses larger tile budgets)

__device__ __forceinline__ int d_slanted_radius(uint32_t n, uint32_t m, int r0)
{
    const uint32_t N = n > m ? n : m, M = n > m ? m : n;
    const uint32_t extra = ((N - M) * (uint32_t)r0 + N - 1u) / N; // dtw.cpp:298-300, unsigned 32-bit
    return r0 + (int)extra;
}

// exact size of the band's cell set (same walk as the kernels; reporting only)
__device__ uint32_t d_banded_cells(uint32_t n, uint32_t m, int R)
{
    const uint32_t N = n > m ? n : m, M = n > m ? m : n;
    const int P = R + ((R % 2 == 0) ? 1 : 0), S = R + ((R % 2 == 1) ? 1 : 0);
    uint32_t cells = 1;
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
            lo = max(lo, si - (int)N + 1);
            lo = max(lo, -sj);
            hi = min(hi, si + 1);
            hi = min(hi, (int)M - sj);
            if (hi > lo) cells += (uint32_t)(hi - lo);
        }
    }
    return cells;
}

struct JobGeom { uint64_t ref_off; uint32_t read_off, n, m; int32_t r0; uint32_t excl; bool ok; };

// job p of a chain (rawdtw_chain_build_jobs, rmap.cpp:195-196, 253-254, 270)
__device__ __forceinline__ JobGeom job_geom(const DevPlanArgs &a, const rawdtw_anchor_t *an, uint32_t na, uint32_t p,
                                            uint64_t ref_base, uint32_t read_base)
{
    JobGeom g;
    rawdtw_anchor_t s, e;
    if (a.border == 0) { s = an[na - 1]; e = an[0]; g.excl = 0; }
    else {
        const uint32_t parts = na - 1;
        s = an[parts - p]; e = an[parts - p - 1];
        g.excl = (p != parts - 1) ? 1u : 0u;
    }
    g.ok = e.target_position >= s.target_position && e.query_position >= s.query_position;
    g.ref_off = ref_base + s.target_position;
    g.read_off = read_base + s.query_position;
    g.m = e.target_position - s.target_position + 1;
    g.n = e.query_position - s.query_position + 1;
    if (a.banded) {
        const int r = (int)((float)g.n * a.frac); // rmap.cpp:214,276, fp32 product
        g.r0 = r > 1 ? r : 1;
    } else g.r0 = RAWDTW_FULL;
    // the windows must lie inside the arenas (same test as the host planner)
    if ((uint64_t)g.read_off + g.n > a.n_ev || g.ref_off + g.m > a.n_ref || g.n >= 0x7fffffffu || g.m >= 0x7fffffffu) g.ok = false;
    return g;
}

__device__ __forceinline__ bool tile_class(const DevPlanArgs &a, const JobGeom &g, int &R)
{
    R = -1;
    if (g.r0 == RAWDTW_FULL) return false;
    R = d_slanted_radius(g.n, g.m, g.r0);
    const uint32_t N = g.n > g.m ? g.n : g.m;
    return R >= 0 && R <= a.lane_max_radius && N <= a.lane_max_n;
}

} // namespace

// chain numbers per job: every chain with jobs marks its first job with (chain + 1); an inclusive running maximum
// then carries the mark over the chain's jobs
__global__ __launch_bounds__(kPlanThreads) void k_mark_chains(uint64_t n_chains, const uint64_t *__restrict__ job_off,
                                                              uint32_t *__restrict__ mark)
{
    const uint64_t c = (uint64_t)blockIdx.x * kPlanThreads + threadIdx.x;
    if (c < n_chains && job_off[c + 1] > job_off[c]) mark[job_off[c]] = (uint32_t)c + 1u;
}

// ---- k_plan_jobs -------------------------------------------------------------------------------
__global__ __launch_bounds__(kPlanThreads) void k_plan_jobs(DevPlanArgs a, const uint64_t *__restrict__ job_off,
                                                            const uint64_t *__restrict__ anchor_off,
                                                            const rawdtw_anchor_t *__restrict__ anchors,
                                                            const uint64_t *__restrict__ ref_base,
                                                            const uint32_t *__restrict__ read_base,
                                                            DevJob *__restrict__ pjobs, const uint32_t *__restrict__ chain_of,
                                                            uint32_t *__restrict__ cost, uint32_t *__restrict__ is_tile,
                                                            uint8_t *__restrict__ run_start,
                                                            unsigned long long *__restrict__ counters)
{
    unsigned long long my_cells = 0, my_bytes = 0;
    uint32_t my_max = 0;
    // grid-stride: a few thousand workgroups, so that the totals below cost a few thousand same-address atomics
    for (uint64_t j = (uint64_t)blockIdx.x * kPlanThreads + threadIdx.x; j < a.n_jobs; j += (uint64_t)gridDim.x * kPlanThreads) {
        const uint32_t c = chain_of[j] - 1u; // (k_mark_chains + a running maximum: no search per job)
        const uint32_t p = (uint32_t)(j - job_off[c]);
        const uint64_t a0 = anchor_off[c];
        const uint32_t na = (uint32_t)(anchor_off[c + 1] - a0);
        const JobGeom g = job_geom(a, anchors + a0, na, p, ref_base[c], read_base[c]);
        int R;
        bool tile = tile_class(a, g, R);
        if (!g.ok) { atomicMin(&counters[kPlanBad], (unsigned long long)j); tile = false; }
        if (g.ok && g.r0 != RAWDTW_FULL && (R < 0 || R + 1 > kMaxWaveBandK)) atomicMin(&counters[kPlanBadRadius], (unsigned long long)j);
        // A tile job continues its chain's run if the previous part is a tile job too -- or if only a small hole (parts
        // of another class, at most kSpanGapFloats of either arena) separates it from one: staging the hole is cheaper
        // than two more spans (the host planner bridges the same holes).
        bool starts = true;
        uint32_t hole = 0;
        if (tile && p > 0 && a.border != 0) {
            uint32_t gap_read = 0, gap_ref = 0;
            for (uint32_t back = 1; back <= 4 && back <= p; back++) {
                const JobGeom gp = job_geom(a, anchors + a0, na, p - back, ref_base[c], read_base[c]);
                int Rp;
                if (!gp.ok) break;
                if (tile_class(a, gp, Rp)) { starts = false; break; }
                gap_read += gp.n - 1; gap_ref += gp.m - 1;
                if (gap_read > kSpanGapFloats || gap_ref > kSpanGapFloats) break;
            }
            if (!starts) hole = gap_read + gap_ref;
        }
        DevJob d;
        d.ref_off = g.ref_off; d.read_off = g.read_off; d.n = g.n; d.m = g.m;
        d.R = tile ? R : g.r0; // tile jobs: after the slant correction; others: as the caller's job list has it
        d.flags = g.excl ? kFlagExcludeLast : 0u;
        d.aux = (uint32_t)j;
        pjobs[j] = d;
        uint32_t cst = 0;
        if (tile) {
            // in eighths of a float, with a floor that bounds the jobs of a tile (k_plan_tiles sorts at most 2048 records)
            // a run start pays both windows, their exact start alignment and up to 3 floats of padding per span end
            cst = 8u * (starts ? g.n + g.m + (g.read_off & 3u) + (uint32_t)(g.ref_off & 3ull) + 6u : g.n + g.m - 2u + hole);
            if (cst < a.min_cost8) cst = a.min_cost8;
            my_bytes += 4ull * ((unsigned long long)g.n + g.m) + 36ull; // (cells are counted on demand: dev_count_tile_cells)
        }
        cost[j] = cst;
        if (tile) my_max = max(my_max, cst);
        is_tile[j] = tile ? 1u : 0u;
        run_start[j] = (tile && starts) ? 1 : 0;
    }
    // tile-class totals (reporting) and the largest cost: reduced per workgroup, then one atomic each (same-address
    // atomics serialise: one per wave cost 1.2 ms at 5 M jobs)
    for (int off = 32; off > 0; off >>= 1) {
        my_cells += __shfl_down(my_cells, off);
        my_bytes += __shfl_down(my_bytes, off);
        my_max = max(my_max, (uint32_t)__shfl_down((int)my_max, off));
    }
    __shared__ unsigned long long s_cells[kPlanThreads / 64], s_bytes[kPlanThreads / 64];
    __shared__ uint32_t s_max[kPlanThreads / 64];
    if ((threadIdx.x & 63) == 0) { s_cells[threadIdx.x >> 6] = my_cells; s_bytes[threadIdx.x >> 6] = my_bytes; s_max[threadIdx.x >> 6] = my_max; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long c = 0, b = 0;
        uint32_t m = 0;
        for (int w = 0; w < kPlanThreads / 64; w++) { c += s_cells[w]; b += s_bytes[w]; m = max(m, s_max[w]); }
        if (c | b) {
            atomicAdd(&counters[kPlanTileCells], c);
            atomicAdd(&counters[kPlanTileBytes], b);
            atomicMax(&counters[kPlanMaxCost8], (unsigned long long)m);
        }
    }
}

// ---- k_plan_scatter ----------------------------------------------------------------------------
// rank[j]: exclusive count of tile jobs before j; cum[j]: inclusive running cost.  Writes per tile-job position:
// the job index, the tile number and whether a run starts there (chain run start, or the first job of a tile).
__global__ __launch_bounds__(kPlanThreads) void k_plan_scatter(DevPlanArgs a, const DevJob *__restrict__ pjobs,
                                                               const uint32_t *__restrict__ cost,
                                                               const uint32_t *__restrict__ is_tile,
                                                               const uint8_t *__restrict__ run_start,
                                                               const uint32_t *__restrict__ rank,
                                                               const uint64_t *__restrict__ cum,
                                                               uint32_t *__restrict__ order_tile,
                                                               uint32_t *__restrict__ tile_no,
                                                               uint32_t *__restrict__ run_flag,
                                                               rawdtw_job_t *__restrict__ other_jobs,
                                                               uint32_t *__restrict__ other_aux,
                                                               const unsigned long long *__restrict__ counters)
{
    const uint64_t j = (uint64_t)blockIdx.x * kPlanThreads + threadIdx.x;
    if (j >= a.n_jobs) return;
    const uint32_t r = rank[j];
    if (is_tile[j]) {
        order_tile[r] = (uint32_t)j;
        // bracket width: the LDS budget minus what the bracket's last job can overshoot (the batch's largest tile job)
        // and what a tile's first job pays on top of its counted cost when it continues a chain (a run start: 14)
        const uint64_t width8 = 8ull * a.tile_budget - counters[kPlanMaxCost8];
        tile_no[r] = (uint32_t)((cum[j] - cost[j]) / width8);
        run_flag[r] = run_start[j];
    } else {
        const uint64_t q = j - r; // rank among the other jobs
        const DevJob d = pjobs[j];
        rawdtw_job_t o;
        o.ref_off = d.ref_off; o.read_off = d.read_off; o.n = d.n; o.m = d.m; o.band_radius = d.R;
        o.exclude_last = (d.flags & kFlagExcludeLast) ? 1u : 0u; o.reserved = 0;
        other_jobs[q] = o;
        other_aux[q] = (uint32_t)j;
    }
}

// n_tile_jobs for the kernels that follow (their grids are sized by n_jobs; positions past it idle)
__global__ void k_plan_count(uint64_t n_jobs, const uint32_t *__restrict__ rank, const uint32_t *__restrict__ is_tile,
                             unsigned long long *__restrict__ counters)
{
    counters[kPlanTileJobs] = n_jobs ? (unsigned long long)rank[n_jobs - 1] + is_tile[n_jobs - 1] : 0ull;
}

// tile starts and run starts per tile-job position (a tile boundary also starts a run); zero past the last tile job
__global__ __launch_bounds__(kPlanThreads) void k_plan_flags(uint64_t n_jobs, const unsigned long long *__restrict__ counters,
                                                             const uint32_t *__restrict__ tile_no,
                                                             uint32_t *__restrict__ run_flag, uint32_t *__restrict__ tile_flag)
{
    const uint64_t r = (uint64_t)blockIdx.x * kPlanThreads + threadIdx.x;
    if (r >= n_jobs) return;
    if (r >= counters[kPlanTileJobs]) { tile_flag[r] = 0u; run_flag[r] = 0u; return; }
    const uint32_t ts = (r == 0 || tile_no[r] != tile_no[r - 1]) ? 1u : 0u;
    tile_flag[r] = ts;
    if (ts) run_flag[r] = 1u;
}

// tile_first[t] = position of the first job of tile t (dense tile index = exclusive sum of the tile flags); totals
__global__ __launch_bounds__(kPlanThreads) void k_plan_tile_first(uint64_t n_jobs, unsigned long long *__restrict__ counters,
                                                                  const uint32_t *__restrict__ tile_flag,
                                                                  const uint32_t *__restrict__ tile_idx,
                                                                  const uint32_t *__restrict__ run_flag,
                                                                  const uint32_t *__restrict__ run_idx,
                                                                  uint32_t *__restrict__ tile_first)
{
    const uint64_t r = (uint64_t)blockIdx.x * kPlanThreads + threadIdx.x;
    const uint64_t nt = counters[kPlanTileJobs];
    if (r >= nt) return;
    if (tile_flag[r]) tile_first[tile_idx[r]] = (uint32_t)r;
    if (r + 1 == nt) {
        counters[kPlanTiles] = (unsigned long long)tile_idx[r] + tile_flag[r];
        counters[kPlanRuns] = (unsigned long long)run_idx[r] + run_flag[r];
    }
}

// ---- k_plan_tiles ------------------------------------------------------------------------------
__global__ __launch_bounds__(kPlanThreads) void k_plan_tiles(DevPlanArgs a, uint32_t n_tiles,
                                                             const uint32_t *__restrict__ tile_first,
                                                             const uint32_t *__restrict__ order_tile,
                                                             const uint32_t *__restrict__ run_flag,
                                                             const uint32_t *__restrict__ run_idx,
                                                             const DevJob *__restrict__ pjobs,
                                                             TileDesc *__restrict__ tiles, TileSpan *__restrict__ spans,
                                                             TileJob *__restrict__ tjobs,
                                                             unsigned long long *__restrict__ counters)
{
    using Scan = hipcub::BlockScan<uint32_t, kPlanThreads>;
    // LDS is used in three phases that do not overlap: the run table (pass 1 .. record building), the sort of
    // (key, tile-local index) pairs, and the exchange that brings each record to its sorted place
    struct RunTable {
        uint64_t read_start[kPlanMaxRuns], ref_start[kPlanMaxRuns]; // aligned (4-float) span starts
        uint32_t last[kPlanMaxRuns];                                // tile-local index of the run's last job
        uint32_t read_lds[kPlanMaxRuns], ref_lds[kPlanMaxRuns];     // LDS float offsets of the two spans
    };
    __shared__ union Lds {
        RunTable runs;
        typename hipcub::BlockRadixSort<uint32_t, kPlanThreads, 2, uint16_t>::TempStorage sort2;
        typename hipcub::BlockRadixSort<uint32_t, kPlanThreads, 4, uint16_t>::TempStorage sort4;
        typename hipcub::BlockRadixSort<uint32_t, kPlanThreads, kTileItems, uint16_t>::TempStorage sort8;
        TileJob recs[kPlanMaxTileJobs];
        __device__ Lds() {}
    } lds;
    __shared__ typename Scan::TempStorage scan_tmp;
    uint64_t *const s_read_start = lds.runs.read_start, *const s_ref_start = lds.runs.ref_start;
    uint32_t *const s_last = lds.runs.last, *const s_read_lds = lds.runs.read_lds, *const s_ref_lds = lds.runs.ref_lds;

    const uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    const uint32_t n_tile_jobs = (uint32_t)counters[kPlanTileJobs];
    const uint32_t first = tile_first[t];
    const uint32_t end = (t + 1 < n_tiles) ? tile_first[t + 1] : n_tile_jobs;
    const uint32_t n = end - first;
    // run_idx is the EXCLUSIVE count of run starts: the run of position pos is run_idx[pos] + run_flag[pos] - 1
    auto run_of = [&](uint32_t pos) { return run_idx[pos] + run_flag[pos] - 1u; };
    const uint32_t run0 = run_idx[first]; // (a tile's first job starts a run)
    const uint32_t n_runs = n ? run_of(end - 1) - run0 + 1u : 0u;
    if (n > kPlanMaxTileJobs || n_runs > kPlanMaxRuns || n_runs == 0) { // cannot happen with the cost floor; keep the batch safe
        if (threadIdx.x == 0) atomicMin(&counters[kPlanTileOverflow], (unsigned long long)t);
        return;
    }
    // pass 1: run extents.  The jobs of a run are consecutive parts of one chain: windows ascend and touch.
    for (uint32_t i = threadIdx.x; i < n; i += kPlanThreads) {
        const uint32_t pos = first + i;
        const uint32_t rid = run_of(pos) - run0;
        if (run_flag[pos]) {
            const DevJob d = pjobs[order_tile[pos]];
            s_read_start[rid] = (uint64_t)(d.read_off & ~3u);
            s_ref_start[rid] = d.ref_off & ~3ull;
        }
        if ((i + 1 == n) || run_flag[pos + 1]) s_last[rid] = i;
    }
    __syncthreads();
    // per run: lengths (floats, multiples of 4) of its two spans, then their LDS offsets by a block scan
    constexpr int RPT = (kPlanMaxRuns + kPlanThreads - 1) / kPlanThreads; // runs per thread
    uint32_t len2[2 * RPT];
    uint32_t my_total = 0;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const uint32_t r = threadIdx.x * RPT + k;
        uint32_t lr = 0, lf = 0;
        if (r < n_runs) {
            const DevJob dl = pjobs[order_tile[first + s_last[r]]];
            lr = (uint32_t)(((uint64_t)dl.read_off + dl.n - s_read_start[r] + 3ull) & ~3ull);
            lf = (uint32_t)((dl.ref_off + dl.m - s_ref_start[r] + 3ull) & ~3ull);
        }
        len2[2 * k] = lr; len2[2 * k + 1] = lf;
        my_total += lr + lf;
    }
    uint32_t my_base, tile_total;
    Scan(scan_tmp).ExclusiveSum(my_total, my_base, tile_total);
    __syncthreads();
    const uint32_t span_first = 2u * run0;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const uint32_t r = threadIdx.x * RPT + k;
        if (r < n_runs) {
            s_read_lds[r] = my_base;
            s_ref_lds[r] = my_base + len2[2 * k];
            spans[span_first + 2 * r] = TileSpan{s_read_start[r], my_base, len2[2 * k] / 4};
            spans[span_first + 2 * r + 1] = TileSpan{s_ref_start[r], my_base + len2[2 * k], (len2[2 * k + 1] / 4) | 0x80000000u};
            my_base += len2[2 * k] + len2[2 * k + 1];
        }
    }
    __syncthreads();
    // pass 2: records (blocked arrangement, in registers), then a stable sort of (key, tile-local index) by
    // (kind, longer side desc, shorter side desc) -- job order breaks ties -- and an exchange through LDS
    auto records = [&](auto items_tag, auto &sort_storage) {
        constexpr int ITEMS = decltype(items_tag)::value;
        using SortT = hipcub::BlockRadixSort<uint32_t, kPlanThreads, ITEMS, uint16_t>;
        uint32_t keys[ITEMS];
        uint16_t idx[ITEMS];
        TileJob recs[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t i = threadIdx.x * ITEMS + k; // blocked arrangement
            keys[k] = 0xffffffffu;
            idx[k] = (uint16_t)i;
            recs[k] = TileJob{0, 0, 0, 0, 255, 0, 0, 0};
            if (i < n) {
                const uint32_t pos = first + i;
                const DevJob d = pjobs[order_tile[pos]];
                const uint32_t rid = run_of(pos) - run0;
                const bool swap = d.n < d.m; // dtw.cpp:284-292: A is the longer sequence
                const uint32_t off_read = s_read_lds[rid] + (uint32_t)((uint64_t)d.read_off - s_read_start[rid]);
                const uint32_t off_ref = s_ref_lds[rid] + (uint32_t)(d.ref_off - s_ref_start[rid]);
                const uint32_t NA = swap ? d.m : d.n, NB = swap ? d.n : d.m;
                TileJob tj;
                tj.offA = (uint16_t)(swap ? off_ref : off_read);
                tj.offB = (uint16_t)(swap ? off_read : off_ref);
                tj.N = (uint8_t)NA; tj.M = (uint8_t)NB; tj.flags = (uint8_t)d.flags;
                tj.aux = d.aux; tj.pad = 0;
                if (NA <= a.micro_max_n) { // micro path: per-shape band bitmask (host table, same index)
                    tj.pad = ((NA - 1) * 8 + (NB - 1)) * (kMaxLaneRadius + 1) + (uint32_t)d.R;
                    tj.R = NA <= 4 ? 0 : 1;
                } else tj.R = (uint8_t)(2 + d.R);
                recs[k] = tj;
                keys[k] = ((uint32_t)tj.R << 16) | ((255u - NA) << 8) | (255u - NB);
            }
        }
        __syncthreads(); // the run table is dead from here on
        SortT(sort_storage).Sort(keys, idx, 0, 20); // 4 + 8 + 8 key bits (padding keys are all ones: they sort last)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t i = threadIdx.x * ITEMS + k;
            if (i < n) lds.recs[i] = recs[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t i = threadIdx.x * ITEMS + k; // sorted rank (blocked)
            if (i < n) tjobs[first + i] = lds.recs[idx[k]];
        }
    };
    if (n <= 2u * kPlanThreads) records(std::integral_constant<int, 2>{}, lds.sort2);
    else if (n <= 4u * kPlanThreads) records(std::integral_constant<int, 4>{}, lds.sort4);
    else records(std::integral_constant<int, kTileItems>{}, lds.sort8);
    if (threadIdx.x == 0) {
        tiles[t] = TileDesc{first, n, span_first, 2u * n_runs};
        atomicMax(&counters[kPlanLdsMax], (unsigned long long)tile_total);
    }
}

// Cells of the tile class, from the tile records (reporting only; the walk costs as much as scoring the jobs, so it
// runs when somebody asks -- rawdtw_plan_info / launch stats -- not while planning).
__global__ __launch_bounds__(kPlanThreads) void k_count_tile_cells(const TileJob *__restrict__ tjobs, uint64_t n,
                                                                   unsigned long long *__restrict__ total)
{
    const uint64_t i = (uint64_t)blockIdx.x * kPlanThreads + threadIdx.x;
    unsigned long long c = 0;
    if (i < n) {
        const TileJob tj = tjobs[i];
        const int R = tj.R >= 2 ? (int)tj.R - 2 : (int)(tj.pad % (kMaxLaneRadius + 1));
        c = d_banded_cells(tj.N, tj.M, R);
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    __shared__ unsigned long long s_c[kPlanThreads / 64];
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kPlanThreads / 64; w++) t += s_c[w];
        if (t) atomicAdd(total, t);
    }
}

hipError_t dev_count_tile_cells(const TileJob *d_tjobs, uint64_t n, unsigned long long *d_total, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(k_count_tile_cells, dim3((uint32_t)((n + kPlanThreads - 1) / kPlanThreads)), dim3(kPlanThreads), 0, s,
                       d_tjobs, n, d_total);
    return hipGetLastError();
}

// ---- host-callable driver ----------------------------------------------------------------------
namespace {
// bring-up aid: RAWDTW_PLAN_DEBUG=1 synchronises after every planning step and names it on stderr
inline bool plan_debug() { static const bool on = getenv("RAWDTW_PLAN_DEBUG") != nullptr; return on; }
inline hipError_t step(const char *what, hipStream_t s)
{
    if (!plan_debug()) return hipSuccess;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    const hipError_t e = hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[devplan] %-18s %s  (%.3f ms since the previous step)\n", what, e == hipSuccess ? "ok" : hipGetErrorString(e),
            std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
    return e;
}
inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kPlanThreads - 1) / kPlanThreads); }
inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
}

size_t dev_plan_scratch_bytes(uint64_t n_jobs)
{
    size_t scan_tmp = 0, t2 = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_tmp, (uint32_t *)nullptr, (uint64_t *)nullptr, (int)n_jobs);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t2, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_jobs);
    if (t2 > scan_tmp) scan_tmp = t2;
    (void)hipcub::DeviceScan::InclusiveScan(nullptr, t2, (uint32_t *)nullptr, (uint32_t *)nullptr, hipcub::Max(), (int)n_jobs);
    if (t2 > scan_tmp) scan_tmp = t2;
    size_t b = 0;
    b += align_up(n_jobs * sizeof(DevJob));  // pjobs
    b += align_up(n_jobs * 4) * 7;           // chain_of, cost, is_tile, rank, order_tile, tile_no, run_flag
    b += align_up(n_jobs * 4) * 3;           // tile_flag, tile_idx, run_idx
    b += align_up(n_jobs);                   // run_start
    b += align_up(n_jobs * 8);               // cum
    b += align_up((n_jobs + 1) * 4);         // tile_first
    b += align_up(scan_tmp) + 4096;
    return b;
}

// Phase 1: everything up to the counts the host needs to size the outputs (counters: tile jobs, tiles, runs, errors).  `scratch` must hold
// dev_plan_scratch_bytes(n_jobs); `other_jobs` / `other_aux` hold up to n_jobs entries (the caller may size them smaller
// if it knows a bound; they are written only for non-tile jobs, densely).
hipError_t dev_plan_phase1(const DevPlanArgs &a, const uint64_t *d_job_off, const uint64_t *d_anchor_off,
                           const rawdtw_anchor_t *d_anchors, const uint64_t *d_ref_base, const uint32_t *d_read_base,
                           void *scratch, DevPlanBuffers *buf, rawdtw_job_t *d_other_jobs, uint32_t *d_other_aux,
                           unsigned long long *d_counters, hipStream_t s)
{
    const uint64_t n = a.n_jobs;
    char *p = static_cast<char *>(scratch);
    auto take = [&](size_t bytes) { void *q = p; p += align_up(bytes); return q; };
    buf->pjobs = static_cast<DevJob *>(take(n * sizeof(DevJob)));
    buf->chain_of = static_cast<uint32_t *>(take(n * 4));
    buf->cost = static_cast<uint32_t *>(take(n * 4));
    buf->is_tile = static_cast<uint32_t *>(take(n * 4));
    buf->rank = static_cast<uint32_t *>(take(n * 4));
    buf->order_tile = static_cast<uint32_t *>(take(n * 4));
    buf->tile_no = static_cast<uint32_t *>(take(n * 4));
    buf->run_flag = static_cast<uint32_t *>(take(n * 4));
    buf->tile_flag = static_cast<uint32_t *>(take(n * 4));
    buf->tile_idx = static_cast<uint32_t *>(take(n * 4));
    buf->run_idx = static_cast<uint32_t *>(take(n * 4));
    buf->run_start = static_cast<uint8_t *>(take(n));
    buf->cum = static_cast<uint64_t *>(take(n * 8));
    buf->tile_first = static_cast<uint32_t *>(take((n + 1) * 4));
    size_t scan_tmp = 0, t2 = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_tmp, buf->cost, buf->cum, (int)n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t2, buf->is_tile, buf->rank, (int)n);
    if (t2 > scan_tmp) scan_tmp = t2;
    (void)hipcub::DeviceScan::InclusiveScan(nullptr, t2, buf->rank, buf->chain_of, hipcub::Max(), (int)n);
    if (t2 > scan_tmp) scan_tmp = t2;
    buf->scan_tmp = take(scan_tmp);
    buf->scan_tmp_bytes = scan_tmp;

    hipError_t e;
    // counters: minima start at ~0, sums and maxima at 0
    unsigned long long init[kPlanCounters];
    for (int i = 0; i < kPlanCounters; i++) init[i] = 0;
    init[kPlanBad] = init[kPlanBadRadius] = init[kPlanTileOverflow] = ~0ull;
    if ((e = hipMemcpyAsync(d_counters, init, sizeof(init), hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    (void)hipGetLastError(); // (a stale error of an unrelated call must not be blamed on these launches)
    {   // chain of every job (buf->rank serves as the mark array until the rank scan overwrites it)
        if ((e = hipMemsetAsync(buf->rank, 0, n * 4, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_mark_chains, dim3(blocks_for(a.n_chains)), dim3(kPlanThreads), 0, s, a.n_chains, d_job_off, buf->rank);
        size_t tbm = buf->scan_tmp_bytes;
        if ((e = hipcub::DeviceScan::InclusiveScan(buf->scan_tmp, tbm, buf->rank, buf->chain_of, hipcub::Max(), (int)n, s)) != hipSuccess) return e;
        if ((e = step("chain marks", s)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_plan_jobs, dim3(std::min<uint32_t>(blocks_for(n), 4096u)), dim3(kPlanThreads), 0, s, a, d_job_off, d_anchor_off, d_anchors,
                       d_ref_base, d_read_base, buf->pjobs, buf->chain_of, buf->cost, buf->is_tile, buf->run_start, d_counters);
    if ((e = step("k_plan_jobs", s)) != hipSuccess) return e;
    size_t tb = buf->scan_tmp_bytes;
    if ((e = hipcub::DeviceScan::InclusiveSum(buf->scan_tmp, tb, buf->cost, buf->cum, (int)n, s)) != hipSuccess) return e;
    tb = buf->scan_tmp_bytes;
    if ((e = hipcub::DeviceScan::ExclusiveSum(buf->scan_tmp, tb, buf->is_tile, buf->rank, (int)n, s)) != hipSuccess) return e;
    if ((e = step("scans cost/rank", s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_plan_scatter, dim3(blocks_for(n)), dim3(kPlanThreads), 0, s, a, buf->pjobs, buf->cost, buf->is_tile,
                       buf->run_start, buf->rank, buf->cum, buf->order_tile, buf->tile_no, buf->run_flag, d_other_jobs,
                       d_other_aux, d_counters);
    if ((e = step("k_plan_scatter", s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_plan_count, dim3(1), dim3(1), 0, s, n, buf->rank, buf->is_tile, d_counters);
    // tile starts, dense tile numbers, run numbers (grids and scans sized by n_jobs: no host round trip in between)
    hipLaunchKernelGGL(k_plan_flags, dim3(blocks_for(n)), dim3(kPlanThreads), 0, s, n, d_counters, buf->tile_no,
                       buf->run_flag, buf->tile_flag);
    if ((e = step("k_plan_flags", s)) != hipSuccess) return e;
    tb = buf->scan_tmp_bytes;
    if ((e = hipcub::DeviceScan::ExclusiveSum(buf->scan_tmp, tb, buf->tile_flag, buf->tile_idx, (int)n, s)) != hipSuccess) return e;
    tb = buf->scan_tmp_bytes;
    if ((e = hipcub::DeviceScan::ExclusiveSum(buf->scan_tmp, tb, buf->run_flag, buf->run_idx, (int)n, s)) != hipSuccess) return e;
    if ((e = step("scans tile/run", s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_plan_tile_first, dim3(blocks_for(n)), dim3(kPlanThreads), 0, s, n, d_counters, buf->tile_flag,
                       buf->tile_idx, buf->run_flag, buf->run_idx, buf->tile_first);
    if ((e = step("k_plan_tile_first", s)) != hipSuccess) return e;
    return hipGetLastError();
}

// Phase 2: the tile records themselves (the host has read the counters and allocated the outputs).
hipError_t dev_plan_phase2(const DevPlanArgs &a, uint32_t n_tiles, DevPlanBuffers *buf, TileDesc *d_tiles,
                           TileSpan *d_spans, TileJob *d_tjobs, unsigned long long *d_counters, hipStream_t s)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_plan_tiles, dim3(n_tiles), dim3(kPlanThreads), 0, s, a, n_tiles, buf->tile_first,
                       buf->order_tile, buf->run_flag, buf->run_idx, buf->pjobs, d_tiles, d_spans, d_tjobs, d_counters);
    { const hipError_t e = step("k_plan_tiles", s); if (e != hipSuccess) return e; }
    return hipGetLastError();
}

} // namespace rawdtw
