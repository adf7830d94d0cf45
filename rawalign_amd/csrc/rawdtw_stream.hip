// rawdtw_stream.hip -- the sync-free candidate-batch pipeline (rawdtw_batch_create/run for sparse + banded batches).
//
// What it replaces: the DTW block of gen_chains (src/rmap.cpp:509-530) for every read of a mini-batch, i.e. all the calls
// of DTW_global_slantedbanded_antidiagonalwise (src/dtw.cpp:273-520) that align_chain (src/rmap.cpp:238-300) issues.
//
// Round 1 planned a batch in eleven launches with two host round trips and wrote 16-byte tile records for every job to
// HBM, which the DTW kernel then read back: 0.6 ms of GPU time and 4 ms of host time per 0.15 ms of DTW.  Here nothing
// about a tile ever reaches HBM:
//
//   k_pre          one thread per job: the job's windows from its chain's anchors (rmap.cpp:251-254, 270, 276), the
//                  slant-corrected radius (dtw.cpp:298-300), its class, and what it adds to a tile's LDS image.  Writes
//                  one 16-byte record per job (offsets + packed shape) and the cost; the rare jobs the lane-per-job DP
//                  does not take (radius > 3 or longer side > 73) are appended to a side list as full job records.
//   scan           running cost (hipcub) -> a tile = the jobs whose running cost starts inside one budget-wide bracket
//   k_tile_first   one binary search per tile boundary
//   k_others       the side list ordered by class and length (wave-per-job first, longest first)
//   k_stream       ONE persistent launch per batch: workgroups first take the side list's wave-cooperative jobs, then pull
//                  tiles from a queue.  Per tile, in LDS: the runs of consecutive parts (one span of each arena per
//                  run), their layout, the job records, a counting sort by (kind, longer side), the staged windows --
//                  then the lane-per-job DP of rawdtw_dp.h.
//
// No step needs a number on the host: grids are sized by the job count (known from the anchor offsets) or are
// persistent, every count lives in a device counter block.  rawdtw_batch_create only enqueues; errors and the rare
// shapes this path does not take (band wider than 256 offsets) surface in the counters, which rawdtw_batch_fetch reads
// together with the results.
#include <hipcub/hipcub.hpp>

#include "rawdtw_dp.h"

namespace rawdtw {

namespace {

constexpr int kT = 256;                 // threads per workgroup, everywhere in this file
constexpr uint32_t kPreUnit = 1024;     // jobs per k_pre workgroup
constexpr uint32_t kItems = kStreamMaxTileJobs / kT; // jobs per thread in the tile prologue (blocked)
static_assert(kStreamMaxTileJobs % kT == 0 && kItems >= 1, "tile job capacity");

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

// inclusive scan of one value per thread over the workgroup (kT threads); `tmp` holds kT/64 words.  *excl (optional)
// receives the exclusive value, *total the reduction over all threads.
template <typename Op>
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t *tmp, Op op, uint32_t identity, uint32_t *total,
                                                    uint32_t *excl = nullptr)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d);
        if (lane >= d) v = op(v, o);
    }
    const uint32_t up = (uint32_t)__shfl_up((int)v, 1);
    if (lane == 63) tmp[wv] = v;
    __syncthreads();
    uint32_t pre = identity, all = identity;
#pragma unroll
    for (int w = 0; w < kT / 64; w++) {
        const uint32_t x = tmp[w];
        if (w < wv) pre = op(pre, x);
        all = op(all, x);
    }
    __syncthreads(); // tmp may be reused by the caller
    if (total) *total = all;
    if (excl) *excl = lane ? op(pre, up) : pre;
    return op(pre, v);
}

struct OpAdd { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } };
struct OpMax { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

// first index in [0, n) with a[i] > key (n when none)
__device__ __forceinline__ uint64_t upper_bound_u64(const uint64_t *a, uint64_t n, uint64_t key)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] > key) hi = mid; else lo = mid + 1;
    }
    return lo;
}
// first index in [0, n) with a[i] >= key (n when none)
__device__ __forceinline__ uint64_t lower_bound_u64(const uint64_t *a, uint64_t n, uint64_t key)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] >= key) hi = mid; else lo = mid + 1;
    }
    return lo;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// k_pre
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void k_pre(const StreamArgs a)
{
    __shared__ uint32_t s_cidx[kPreUnit]; // chain (relative, +1) that starts at this job of the unit; then: chain of every job
    __shared__ uint32_t s_tmp[kT / 64];
    __shared__ uint64_t s_c[2];
    __shared__ uint32_t s_ocnt, s_obase, s_cls[kStreamClasses];
    __shared__ unsigned long long s_bytes, s_obytes;
    __shared__ uint32_t s_tiles, s_maxc;
    const int tid = threadIdx.x;
    const uint64_t j0 = (uint64_t)blockIdx.x * kPreUnit;
    const uint32_t cnt = (uint32_t)min<uint64_t>(kPreUnit, a.n_jobs - j0);
    if (tid == 0) {
        // chains [c_lo, c_hi) own the unit's jobs: job_off[c_lo] <= j0 < job_off[c_lo + 1]; job_off[c_hi] >= j0 + cnt
        s_c[0] = upper_bound_u64(a.job_off, a.n_chains + 1, j0) - 1;
        s_c[1] = lower_bound_u64(a.job_off, a.n_chains + 1, j0 + cnt);
        s_ocnt = 0; s_bytes = 0; s_obytes = 0; s_tiles = 0; s_maxc = 0;
    }
    if (tid < (int)kStreamClasses) s_cls[tid] = 0;
    for (uint32_t i = tid; i < kPreUnit; i += kT) s_cidx[i] = 0;
    __syncthreads();
    const uint64_t c_lo = s_c[0], c_hi = s_c[1];
    for (uint64_t c = c_lo + tid; c < c_hi; c += kT) {
        const uint64_t b = a.job_off[c], e = a.job_off[c + 1];
        if (e > b) atomicMax(&s_cidx[(uint32_t)((b > j0 ? b : j0) - j0)], (uint32_t)(c - c_lo) + 1u);
    }
    __syncthreads();
    {   // running maximum over the unit: thread t owns entries [4t, 4t + 4)
        constexpr uint32_t PER = kPreUnit / kT;
        uint32_t v[PER];
        uint32_t run = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) { run = max(run, s_cidx[tid * PER + k]); v[k] = run; }
        uint32_t excl = 0;
        (void)block_scan_incl(run, s_tmp, OpMax(), 0u, nullptr, &excl);
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) s_cidx[tid * PER + k] = max(v[k], excl);
    }
    __syncthreads();

    uint32_t my_tiles = 0, my_max = 0;
    unsigned long long my_bytes = 0, my_obytes = 0;
    DevJob oj[kPreUnit / kT];
    uint32_t oslot[kPreUnit / kT];
    uint8_t ocl[kPreUnit / kT];
#pragma unroll
    for (uint32_t k = 0; k < kPreUnit / kT; k++) {
        const uint32_t i = k * kT + tid; // strided: coalesced stores
        oslot[k] = 0xffffffffu;
        if (i >= cnt) continue;
        const uint64_t j = j0 + i;
        const uint64_t c = c_lo + s_cidx[i] - 1u;
        const uint32_t p = (uint32_t)(j - a.job_off[c]);
        const uint64_t a0 = a.anchor_off[c];
        const uint32_t parts = (uint32_t)(a.anchor_off[c + 1] - a0) - 1u;
        const rawdtw_anchor_t s = a.anchors[a0 + parts - p];     // rmap.cpp:253
        const rawdtw_anchor_t e = a.anchors[a0 + parts - p - 1]; // rmap.cpp:254
        bool ok = e.target_position >= s.target_position && e.query_position >= s.query_position;
        const uint64_t ref_off = a.ref_base[c] + s.target_position;
        const uint32_t read_off = a.read_base[c] + s.query_position;
        const uint32_t m = e.target_position - s.target_position + 1;
        const uint32_t n = e.query_position - s.query_position + 1;
        if ((uint64_t)a.read_base[c] + s.query_position + n > a.n_ev || ref_off + m > a.n_ref || n >= 0x7fffffffu || m >= 0x7fffffffu)
            ok = false;
        const bool excl = p != parts - 1; // rmap.cpp:270
        int r0 = (int)((float)n * a.frac); // rmap.cpp:276, fp32 product
        r0 = r0 > 1 ? r0 : 1;
        const int R = ok ? d_slanted_radius(n, m, r0) : 0;
        const uint32_t N = n > m ? n : m, M = n > m ? m : n;
        const bool tile = ok && R <= a.lane_max_radius && N <= a.lane_max_n;
        uint32_t meta = 0, cost8 = a.min_cost8;
        if (!ok) atomicMin(&a.cnt[kCntBad], (unsigned long long)j);
        if (tile) {
            // a tile job continues its chain's run when the part before it is a tile job too: consecutive parts share their
            // anchor element, so the run is one contiguous span of each arena
            bool starts = true;
            if (p > 0) {
                const rawdtw_anchor_t sp = a.anchors[a0 + parts - p + 1];
                const bool okp = s.target_position >= sp.target_position && s.query_position >= sp.query_position;
                const uint32_t mp = s.target_position - sp.target_position + 1, np = s.query_position - sp.query_position + 1;
                int rp = (int)((float)np * a.frac);
                rp = rp > 1 ? rp : 1;
                const int Rp = okp ? d_slanted_radius(np, mp, rp) : 99;
                if (okp && Rp <= a.lane_max_radius && max(np, mp) <= a.lane_max_n) starts = false;
            }
            meta = N | (M << 7) | ((uint32_t)R << 14) | ((excl ? 1u : 0u) << 16) | ((n < m ? 1u : 0u) << 17) |
                   ((starts ? 1u : 0u) << 18) | kMetaTile;
            // in eighths of a float; a run start pays both windows, their start alignment and the padding of two span ends
            cost8 = 8u * (starts ? n + m + (read_off & 3u) + (uint32_t)(ref_off & 3ull) + 6u : n + m - 2u);
            if (starts && cost8 < a.run_cost8) cost8 = a.run_cost8; // bounds the runs of a tile
            if (cost8 < a.min_cost8) cost8 = a.min_cost8;           // bounds the jobs of a tile
            my_tiles++;
            my_max = max(my_max, cost8);
            my_bytes += 4ull * ((unsigned long long)n + m) + 36ull;
        } else if (ok) {
            const uint32_t K = (uint32_t)R + 1u;
            uint32_t cls;
            if (K <= 8) cls = kClsG8;
            else if (K <= 16) cls = kClsG16;
            else if (K <= 256) cls = N >= 1024 ? kClsW0 : N >= 256 ? kClsW0 + 1 : N >= 64 ? kClsW0 + 2 : kClsW0 + 3;
            else cls = 0xff;
            if (cls == 0xff) atomicAdd(&a.cnt[kCntUnsupported], 1ull);
            else {
                oslot[k] = atomicAdd(&s_ocnt, 1u);
                atomicAdd(&s_cls[cls], 1u);
                ocl[k] = (uint8_t)cls;
                oj[k].ref_off = ref_off; oj[k].read_off = read_off; oj[k].n = n; oj[k].m = m; oj[k].R = R;
                oj[k].flags = excl ? kFlagExcludeLast : 0u; oj[k].aux = (uint32_t)j;
                my_obytes += 4ull * ((unsigned long long)n + m) + 36ull;
            }
        }
        a.jrec[j] = JobRec{ref_off, read_off, meta};
        a.lds_cost[j] = cost8;
    }
    // totals: one atomic per workgroup and counter
    for (int off = 32; off > 0; off >>= 1) {
        my_tiles += __shfl_down((int)my_tiles, off);
        my_max = max(my_max, (uint32_t)__shfl_down((int)my_max, off));
        my_bytes += __shfl_down(my_bytes, off);
        my_obytes += __shfl_down(my_obytes, off);
    }
    if ((tid & 63) == 0) {
        atomicAdd(&s_tiles, my_tiles); atomicMax(&s_maxc, my_max);
        atomicAdd(&s_bytes, my_bytes); atomicAdd(&s_obytes, my_obytes);
    }
    __syncthreads();
    if (tid == 0) {
        if (s_tiles) { atomicAdd(&a.cnt[kCntTileJobs], (unsigned long long)s_tiles); atomicAdd(&a.cnt[kCntTileBytes], s_bytes); }
        if (s_maxc) atomicMax(&a.cnt[kCntMaxCost8], (unsigned long long)s_maxc);
        if (s_obytes) atomicAdd(&a.cnt[kCntOtherBytes], s_obytes);
        uint32_t base = 0;
        if (s_ocnt) base = (uint32_t)atomicAdd(&a.cnt[kCntOthers], (unsigned long long)s_ocnt);
        s_obase = base;
    }
    if (tid < (int)kStreamClasses && s_cls[tid]) atomicAdd(&a.cnt[kCntCls0 + tid], (unsigned long long)s_cls[tid]);
    __syncthreads();
    if (s_ocnt) {
        const uint64_t base = s_obase;
#pragma unroll
        for (uint32_t k = 0; k < kPreUnit / kT; k++) {
            if (oslot[k] == 0xffffffffu) continue;
            const uint64_t q = base + oslot[k];
            if (q < a.others_cap) { a.omix[q] = oj[k]; a.ocls[q] = ocl[k]; }
            // (beyond the capacity: kCntOthers > others_cap tells rawdtw_batch_fetch to take the job-list path)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_tile_first: tile k = the jobs whose exclusive running cost lies in [k * width, (k + 1) * width).  Every job has a
// positive cost, so tile_first[k] = 1 + (first job whose inclusive running cost reaches k * width).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void k_tile_first(const StreamArgs a)
{
    const uint64_t k = (uint64_t)blockIdx.x * kT + threadIdx.x;
    const uint64_t width8 = 8ull * a.tile_budget - a.cnt[kCntMaxCost8];
    const uint64_t n = a.n_jobs;
    const uint64_t n_tiles = (n >= 2 ? a.cum[n - 2] / width8 : 0ull) + 1ull;
    if (k == 0) {
        a.cnt[kCntTiles] = n_tiles;
        if (n_tiles > a.tiles_cap) atomicMin(&a.cnt[kCntOverflow], 0ull);
    }
    if (k > n_tiles || k > a.tiles_cap) return;
    a.tile_first[k] = k == 0 ? 0u : k == n_tiles ? (uint32_t)n : (uint32_t)(1ull + lower_bound_u64(a.cum, n, k * width8));
}

// ---------------------------------------------------------------------------------------------------------------------
// k_others: the side list in class order (wave-per-job classes first, longest first; then 16-lane groups, 8-lane groups)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void k_others(const StreamArgs a)
{
    const uint64_t n_other = min<uint64_t>(a.cnt[kCntOthers], a.others_cap);
    uint64_t base[kStreamClasses];
    uint64_t acc = 0;
#pragma unroll
    for (uint32_t c = 0; c < kStreamClasses; c++) { base[c] = acc; acc += a.cnt[kCntCls0 + c]; }
    const int lane = threadIdx.x & 63;
    for (uint64_t i0 = ((uint64_t)blockIdx.x * kT + threadIdx.x) & ~63ull; i0 < n_other; i0 += (uint64_t)gridDim.x * kT) {
        const uint64_t i = i0 + lane;
        const uint32_t cls = i < n_other ? a.ocls[i] : 0xffu;
        uint64_t pos = ~0ull;
#pragma unroll
        for (uint32_t c = 0; c < kStreamClasses; c++) {
            const unsigned long long mask = __ballot(cls == c);
            if (!mask) continue;
            const int leader = __ffsll((long long)mask) - 1;
            unsigned long long got = 0;
            if (lane == leader) got = atomicAdd(&a.cnt[kCntCur0 + c], (unsigned long long)__popcll(mask));
            got = (unsigned long long)__shfl((long long)got, leader);
            if (cls == c) pos = base[c] + got + (uint64_t)__popcll(mask & ((1ull << lane) - 1ull));
        }
        if (pos != ~0ull && pos < a.others_cap) a.ojobs[pos] = a.omix[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// chain records for the fold (ChainDesc) and the key of the fold order (part count, clamped)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void k_chain_desc(const StreamArgs a, ChainDesc *__restrict__ chains,
                                                   uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const uint64_t c = (uint64_t)blockIdx.x * kT + threadIdx.x;
    if (c >= a.n_chains) return;
    const uint64_t a0 = a.anchor_off[c], a1 = a.anchor_off[c + 1];
    ChainDesc d;
    d.job_first = a.job_off[c];
    d.n_jobs = (uint32_t)(a.job_off[c + 1] - a.job_off[c]);
    d.reserved = 0; d.span = 0; d.num_aligned = 0;
    if (a1 > a0) {
        const rawdtw_anchor_t first = a.anchors[a1 - 1], last = a.anchors[a0];
        d.span = last.query_position - first.query_position + 1;                     // rmap.cpp:245
        d.num_aligned = (last.query_position - first.query_position) + d.n_jobs;       // sum of the parts' read regions (rmap.cpp:292)
    }
    chains[c] = d;
    key[c] = min(d.n_jobs, 65535u);
    val[c] = (uint32_t)c;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_stream: the batch's one DTW launch
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// LDS scratch: the run table while a tile is laid out, then the sort's histogram and permutation
struct RunTable {
    uint64_t ref_start[kStreamMaxRuns];
    uint32_t read_start[kStreamMaxRuns];
    uint32_t read_len[kStreamMaxRuns], ref_len[kStreamMaxRuns];   // floats, then rounded up to multiples of 4
    uint32_t span_off[2 * kStreamMaxRuns + 1];                    // LDS float offset of span 2r (events) and 2r + 1 (reference)
};
constexpr uint32_t kSortBins = 512; // bin = kind * 80 + (79 - longer side): 6 kinds x 80
struct SortTable {
    uint32_t hist[kSortBins];
    uint16_t perm[kStreamMaxTileJobs];
};
union Scratch { RunTable runs; SortTable sort; };

__device__ __forceinline__ float stream_lane_job(const float *LA, const float *LB, uint32_t N, uint32_t M, int kind,
                                                 uint32_t R, bool excl, const unsigned long long *__restrict__ masks, bool act)
{
    float res = 0.0f;
    const int k = act ? kind : -1;
    if (__any(k == 0)) {
        if (k == 0) {
            const unsigned long long mask = masks[((N - 1) * 8 + (M - 1)) * (kMaxLaneRadius + 1) + R];
            const uint32_t Nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)N); // sorted by longer side, descending
            if (Nw <= 2) res = micro_job_cols<4, 2>(LA, LB, N, M, mask);
            else if (Nw == 3) res = micro_job_cols<4, 3>(LA, LB, N, M, mask);
            else res = micro_job_cols<4, 4>(LA, LB, N, M, mask);
        }
    }
    if (__any(k == 1)) {
        if (k == 1) {
            const unsigned long long mask = masks[((N - 1) * 8 + (M - 1)) * (kMaxLaneRadius + 1) + R];
            const uint32_t Nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)N);
            if (Nw <= 5) res = micro_job_cols<8, 5>(LA, LB, N, M, mask);
            else if (Nw == 6) res = micro_job_cols<8, 6>(LA, LB, N, M, mask);
            else if (Nw == 7) res = micro_job_cols<8, 7>(LA, LB, N, M, mask);
            else res = micro_job_cols<8, 8>(LA, LB, N, M, mask);
        }
    }
#define RAWDTW_LANE_KIND(RR)                                                                                                   \
    if (__any(k == 2 + RR)) {                                                                                                  \
        if (k == 2 + RR) {                                                                                                     \
            const uint32_t N0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)N), M0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)M); \
            if (__all(N == N0)) {                                                                                              \
                if (__all(M == M0)) res = lane_dp<RR>(LA, LB, N0, M0);                                                         \
                else res = lane_dp_sel<RR>(LA, LB, N0, M);                                                                     \
            } else res = lane_dp_sel<RR>(LA, LB, N, M);                                                                        \
        }                                                                                                                      \
    }
    RAWDTW_LANE_KIND(0)
    RAWDTW_LANE_KIND(1)
    RAWDTW_LANE_KIND(2)
    RAWDTW_LANE_KIND(3)
#undef RAWDTW_LANE_KIND
    if (act && excl) res = res - dist(LA[N - 1], LB[M - 1]);
    return res;
}

} // namespace

__global__ __launch_bounds__(kT) void k_stream(const StreamArgs a, const uint32_t others_blocks, const uint32_t lds_floats)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *win = smem;                                                            // the tile's LDS image
    uint2 *rec = reinterpret_cast<uint2 *>(smem + lds_floats);                    // one record per job of the tile's range
    Scratch &sc = *reinterpret_cast<Scratch *>(smem + lds_floats + 2 * kStreamMaxTileJobs);
    __shared__ uint32_t s_tmp[kT / 64];
    __shared__ uint32_t s_next, s_first_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = (uint32_t)tid >> 6;

    // ---- the side list: wave-cooperative jobs, wave-per-job classes first ----
    if (blockIdx.x < others_blocks) {
        uint64_t n_w = 0;
#pragma unroll
        for (uint32_t c = kClsW0; c < kClsW0 + 4; c++) n_w += a.cnt[kCntCls0 + c];
        const uint64_t n_g16 = a.cnt[kCntCls0 + kClsG16], n_g8 = a.cnt[kCntCls0 + kClsG8];
        const uint64_t cap = a.others_cap;
        if (n_w + n_g16 + n_g8 <= cap) { // (otherwise the batch is redone through the job-list path)
            const uint64_t it_g16 = (n_g16 + 3) / 4, it_g8 = (n_g8 + 7) / 8, items = n_w + it_g16 + it_g8;
            for (uint64_t it = (uint64_t)blockIdx.x * (kT / 64) + wv; it < items; it += (uint64_t)others_blocks * (kT / 64)) {
                if (it < n_w) wreg_small_job(a.ojobs[it], lane, a.ev, a.ref, a.out);
                else if (it < n_w + it_g16) grp_wave<16>(a.ojobs + n_w, (uint32_t)n_g16, (uint32_t)(it - n_w), lane, a.ev, a.ref, a.out);
                else grp_wave<8>(a.ojobs + n_w + n_g16, (uint32_t)n_g8, (uint32_t)(it - n_w - it_g16), lane, a.ev, a.ref, a.out);
            }
        }
    }

    // ---- tiles: the first (gridDim - others_blocks) tiles are dealt by block index, the rest pulled from the queue ----
    const uint32_t n_tiles = (uint32_t)min<unsigned long long>(a.cnt[kCntTiles], (unsigned long long)a.tiles_cap);
    const uint32_t n_static = gridDim.x - others_blocks;
    uint32_t t;
    if (blockIdx.x >= others_blocks) t = blockIdx.x - others_blocks;
    else {
        if (tid == 0) s_next = n_static + (uint32_t)atomicAdd(&a.cnt[kCntQueue], 1ull);
        __syncthreads();
        t = s_next;
        __syncthreads();
    }
    while (t < n_tiles) {
        if (tid == 0) s_next = n_static + (uint32_t)atomicAdd(&a.cnt[kCntQueue], 1ull); // the next tile's number travels meanwhile
        const uint32_t first = a.tile_first[t];
        const uint32_t n = a.tile_first[t + 1] - first; // jobs in the tile's range (tile-class or not)
        if (n > kStreamMaxTileJobs) { // cannot happen with the cost floor; keep the batch safe
            if (tid == 0) atomicMin(&a.cnt[kCntOverflow], (unsigned long long)t);
            __syncthreads();
            t = s_next;
            __syncthreads();
            continue;
        }
        // ---- 1. the range's records (blocked: thread t owns jobs [kItems * t, kItems * t + kItems)) and its runs ----
        JobRec jr[kItems];
        uint32_t packed = 0; // low half: tile jobs, high half: run starts (by the chain rule) among this thread's jobs
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++) {
            const uint32_t i = tid * kItems + k;
            jr[k] = JobRec{0, 0, 0};
            if (i < n) jr[k] = a.jrec[first + i];
            if (jr[k].meta & kMetaTile) packed += 1u + ((jr[k].meta & kMetaStarts) ? 0x10000u : 0u);
        }
        uint32_t total = 0;
        const uint32_t incl = block_scan_incl(packed, s_tmp, OpAdd(), 0u, &total);
        uint32_t run = incl - packed; // exclusive counts before this thread's first job
        const uint32_t n_tile_jobs = total & 0xffffu;
        // the tile's first tile job opens a run whether or not its chain's run began in the previous tile
        if (tid == 0) s_first_flag = 0;
        __syncthreads();
        uint32_t rid[kItems];
        bool opens[kItems];
        {
            uint32_t tiles_before = run & 0xffffu, starts_incl = run >> 16;
#pragma unroll
            for (uint32_t k = 0; k < kItems; k++) {
                rid[k] = 0;
                opens[k] = false;
                if (jr[k].meta & kMetaTile) {
                    opens[k] = (jr[k].meta & kMetaStarts) || tiles_before == 0;
                    if (jr[k].meta & kMetaStarts) starts_incl++;
                    else if (tiles_before == 0) s_first_flag = 1; // (one thread at most: the tile's first tile job)
                    rid[k] = starts_incl; // run number + 1 - F, fixed up below
                    tiles_before++;
                }
            }
        }
        __syncthreads();
        const uint32_t F = s_first_flag;
        const uint32_t n_runs = (total >> 16) + F;
        if (n_runs > kStreamMaxRuns) {
            if (tid == 0) atomicMin(&a.cnt[kCntOverflow], (unsigned long long)t);
            __syncthreads();
            t = s_next;
            __syncthreads();
            continue;
        }
        for (uint32_t r = tid; r < n_runs; r += kT) { sc.runs.read_len[r] = 0; sc.runs.ref_len[r] = 0; }
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++) {
            if (!(jr[k].meta & kMetaTile)) continue;
            rid[k] = rid[k] - 1u + F;
            if (opens[k]) {
                sc.runs.read_start[rid[k]] = jr[k].read_off & ~3u;
                sc.runs.ref_start[rid[k]] = jr[k].ref_off & ~3ull;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++) {
            if (!(jr[k].meta & kMetaTile)) continue;
            const uint32_t N = jr[k].meta & 127u, M = (jr[k].meta >> 7) & 127u;
            const bool swap = (jr[k].meta >> 17) & 1u; // the reference window is the longer one
            const uint32_t n_read = swap ? M : N, n_ref = swap ? N : M;
            atomicMax(&sc.runs.read_len[rid[k]], jr[k].read_off + n_read - sc.runs.read_start[rid[k]]);
            atomicMax(&sc.runs.ref_len[rid[k]], (uint32_t)(jr[k].ref_off + n_ref - sc.runs.ref_start[rid[k]]));
        }
        __syncthreads();
        {   // span offsets: one run per thread (kStreamMaxRuns <= kT)
            uint32_t lr = 0, lf = 0;
            if ((uint32_t)tid < n_runs) {
                lr = (sc.runs.read_len[tid] + 3u) & ~3u;
                lf = (sc.runs.ref_len[tid] + 3u) & ~3u;
                sc.runs.read_len[tid] = lr; sc.runs.ref_len[tid] = lf;
            }
            uint32_t image = 0;
            const uint32_t end = block_scan_incl(lr + lf, s_tmp, OpAdd(), 0u, &image);
            if ((uint32_t)tid < n_runs) {
                sc.runs.span_off[2 * tid] = end - lr - lf;
                sc.runs.span_off[2 * tid + 1] = end - lf;
            }
            if (tid == 0) sc.runs.span_off[2 * n_runs] = image;
            if (image > lds_floats) { // cannot happen with the bracket rule; keep the batch safe
                if (tid == 0) atomicMin(&a.cnt[kCntOverflow], (unsigned long long)t);
                __syncthreads();
                t = s_next;
                __syncthreads();
                continue;
            }
            if (tid == 0) atomicMax(&a.cnt[kCntLdsMax], (unsigned long long)image);
        }
        __syncthreads();
        // ---- 2. job records: LDS offsets of the two windows, shape, kind ----
        uint32_t bin[kItems];
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++) {
            bin[k] = 0xffffffffu;
            if (!(jr[k].meta & kMetaTile)) continue;
            const uint32_t i = tid * kItems + k;
            const uint32_t N = jr[k].meta & 127u, M = (jr[k].meta >> 7) & 127u, R = (jr[k].meta >> 14) & 3u;
            const bool swap = (jr[k].meta >> 17) & 1u;
            const uint32_t off_read = sc.runs.span_off[2 * rid[k]] + (jr[k].read_off - sc.runs.read_start[rid[k]]);
            const uint32_t off_ref = sc.runs.span_off[2 * rid[k] + 1] + (uint32_t)(jr[k].ref_off - sc.runs.ref_start[rid[k]]);
            const uint32_t kind = N <= a.micro_max_n ? (N <= 4 ? 0u : 1u) : 2u + R;
            rec[i] = make_uint2((swap ? off_ref : off_read) | ((swap ? off_read : off_ref) << 16),
                                N | (M << 7) | (R << 14) | (((jr[k].meta >> 16) & 1u) << 16) | (kind << 17));
            bin[k] = kind * 80u + (79u - N);
        }
        // ---- 3. stage the spans: one 16-byte chunk per thread and step, the chunk's span by binary search ----
        {
            const uint32_t chunks = sc.runs.span_off[2 * n_runs] >> 2, n_spans = 2 * n_runs;
            __builtin_amdgcn_s_setprio(3); // a fresh tile's loads must not queue behind the DP of the older workgroups
            for (uint32_t q = tid; q < chunks; q += kT) {
                uint32_t lo = 0, hi = n_spans; // largest s with span_off[s] <= 4q
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sc.runs.span_off[mid] <= 4 * q) lo = mid; else hi = mid;
                }
                const uint32_t r = lo >> 1, within = 4 * q - sc.runs.span_off[lo];
                const float *src = (lo & 1u) ? a.ref + sc.runs.ref_start[r] + within : a.ev + sc.runs.read_start[r] + within;
                reinterpret_cast<float4 *>(win)[q] = *reinterpret_cast<const float4 *>(src);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads(); // the run table is dead, the image and the records are complete
        // ---- 4. counting sort of the tile jobs by (kind, longer side descending) ----
        for (uint32_t b = tid; b < kSortBins; b += kT) sc.sort.hist[b] = 0;
        __syncthreads();
        uint32_t rank[kItems];
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++)
            if (bin[k] != 0xffffffffu) rank[k] = atomicAdd(&sc.sort.hist[bin[k]], 1u);
        __syncthreads();
        {
            static_assert(kSortBins == 2 * kT, "two bins per thread");
            const uint32_t h0 = sc.sort.hist[2 * tid], h1 = sc.sort.hist[2 * tid + 1];
            const uint32_t end = block_scan_incl(h0 + h1, s_tmp, OpAdd(), 0u, nullptr);
            sc.sort.hist[2 * tid] = end - h0 - h1;
            sc.sort.hist[2 * tid + 1] = end - h1;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++)
            if (bin[k] != 0xffffffffu) sc.sort.perm[sc.sort.hist[bin[k]] + rank[k]] = (uint16_t)(tid * kItems + k);
        __syncthreads();
        // ---- 5. the DP: one lane per job, 64 jobs of (nearly) one shape per wave ----
        for (uint32_t r0 = 0; r0 < n_tile_jobs; r0 += kT) {
            const uint32_t r = r0 + tid;
            const bool act = r < n_tile_jobs;
            const uint32_t i = sc.sort.perm[act ? r : n_tile_jobs - 1];
            const uint2 rc = rec[i];
            const uint32_t N = rc.y & 127u, M = (rc.y >> 7) & 127u, R = (rc.y >> 14) & 3u;
            const float res = stream_lane_job(win + (rc.x & 0xffffu), win + (rc.x >> 16), N, M, (int)((rc.y >> 17) & 7u), R,
                                              (rc.y >> 16) & 1u, a.masks, act);
            if (act) a.out[first + i] = res;
        }
        __syncthreads();
        t = s_next;
        __syncthreads();
    }
}

// cells of a batch (reporting only; the walk costs as much as scoring the jobs): tile class from the job records, the
// side list from its job records
__global__ __launch_bounds__(kT) void k_stream_cells(const StreamArgs a, unsigned long long *__restrict__ total)
{
    const uint64_t i = (uint64_t)blockIdx.x * kT + threadIdx.x;
    unsigned long long c = 0;
    if (i < a.n_jobs) {
        const JobRec r = a.jrec[i];
        if (r.meta & kMetaTile) c = d_banded_cells(r.meta & 127u, (r.meta >> 7) & 127u, (int)((r.meta >> 14) & 3u));
    } else if (i - a.n_jobs < min<uint64_t>(a.cnt[kCntOthers], a.others_cap)) {
        const DevJob d = a.ojobs[i - a.n_jobs];
        c = d_banded_cells(d.n, d.m, d.R);
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    __shared__ unsigned long long s_c[kT / 64];
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kT / 64; w++) t += s_c[w];
        if (t) atomicAdd(total, t);
    }
}

// scatter of the round's new events into the per-read event arrays (rawdtw_events_append): segment s copies
// src[seg_src[s] .. seg_src[s + 1]) to dst[seg_dst[s] ..]
__global__ __launch_bounds__(kT) void k_events_scatter(const float *__restrict__ src, float *__restrict__ dst,
                                                       const uint64_t *__restrict__ seg_src, const uint32_t *__restrict__ seg_dst,
                                                       uint32_t n_seg)
{
    // one wave per segment and step: segments are a chunk's worth of events (hundreds of floats)
    const uint32_t wave = (blockIdx.x * kT + threadIdx.x) >> 6, lane = threadIdx.x & 63, n_waves = gridDim.x * (kT / 64);
    for (uint32_t s = wave; s < n_seg; s += n_waves) {
        const uint64_t b = seg_src[s], e = seg_src[s + 1];
        const uint32_t d = seg_dst[s];
        for (uint64_t k = b + lane; k < e; k += 64) dst[d + (k - b)] = src[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host-callable drivers
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct CastU64 {
    __host__ __device__ uint64_t operator()(const uint32_t &x) const { return (uint64_t)x; }
};
using CostIter = hipcub::TransformInputIterator<uint64_t, CastU64, const uint32_t *>;
inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kT - 1) / kT); }
} // namespace

size_t stream_scan_bytes(uint64_t n_jobs)
{
    size_t b = 0;
    CostIter it(nullptr, CastU64());
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, b, it, (uint64_t *)nullptr, (int)n_jobs);
    return b + 256;
}

size_t stream_sort_bytes(uint64_t n_chains)
{
    size_t b = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                                       (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_chains, 0, 16);
    return b + 256;
}

uint32_t stream_lds_bytes(uint32_t lds_floats)
{
    return (uint32_t)(lds_floats * 4u + kStreamMaxTileJobs * 8u + sizeof(Scratch));
}

// everything rawdtw_batch_create enqueues for a sparse + banded batch: planning of the DTW launch and the chain records
hipError_t stream_plan(const StreamArgs &a, ChainDesc *d_chains, uint32_t *d_key, uint32_t *d_val, uint32_t *d_key_out,
                       uint32_t *d_fold_order, void *d_tmp, size_t tmp_bytes, hipStream_t s)
{
    (void)hipGetLastError();
    if (a.n_jobs) {
        hipLaunchKernelGGL(k_pre, dim3((uint32_t)((a.n_jobs + kPreUnit - 1) / kPreUnit)), dim3(kT), 0, s, a);
        size_t tb = tmp_bytes;
        CostIter it(a.lds_cost, CastU64());
        hipError_t e = hipcub::DeviceScan::InclusiveSum(d_tmp, tb, it, a.cum, (int)a.n_jobs, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_tile_first, dim3(blocks_for((uint64_t)a.tiles_cap + 1)), dim3(kT), 0, s, a);
        hipLaunchKernelGGL(k_others, dim3(64), dim3(kT), 0, s, a);
    }
    if (a.n_chains) {
        hipLaunchKernelGGL(k_chain_desc, dim3(blocks_for(a.n_chains)), dim3(kT), 0, s, a, d_chains, d_key, d_val);
        size_t tb = tmp_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(d_tmp, tb, d_key, d_key_out, d_val, d_fold_order,
                                                                    (int)a.n_chains, 0, 16, s);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

hipError_t stream_run(const StreamArgs &a, uint32_t others_blocks, uint32_t tile_blocks, uint32_t lds_floats, hipStream_t s)
{
    if (a.n_jobs == 0) return hipSuccess;
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(&a.cnt[kCntQueue], 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const uint32_t lds_bytes = stream_lds_bytes(lds_floats);
    if (lds_bytes > 64 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stream), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_stream, dim3(others_blocks + tile_blocks), dim3(kT), lds_bytes, s, a, others_blocks, lds_floats);
    return hipGetLastError();
}

// workgroups of k_stream one compute unit holds at this LDS size (for the persistent grid)
int stream_blocks_per_cu(uint32_t lds_floats)
{
    int n = 0;
    const uint32_t lds_bytes = stream_lds_bytes(lds_floats);
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_stream), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
        return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(k_stream), kT, lds_bytes) != hipSuccess) return 0;
    return n;
}

hipError_t stream_count_cells(const StreamArgs &a, unsigned long long *d_total, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || a.n_jobs == 0) return e;
    hipLaunchKernelGGL(k_stream_cells, dim3(blocks_for(a.n_jobs + a.others_cap)), dim3(kT), 0, s, a, d_total);
    return hipGetLastError();
}

hipError_t launch_events_scatter(const float *d_src, float *d_dst, const uint64_t *d_seg_src, const uint32_t *d_seg_dst,
                                 uint32_t n_seg, hipStream_t s)
{
    if (n_seg == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n_seg * 64 + kT - 1) / kT, 4096);
    hipLaunchKernelGGL(k_events_scatter, dim3(blocks), dim3(kT), 0, s, d_src, d_dst, d_seg_src, d_seg_dst, n_seg);
    return hipGetLastError();
}

} // namespace rawdtw
