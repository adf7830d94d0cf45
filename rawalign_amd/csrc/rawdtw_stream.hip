// rawdtw_stream.hip -- the sync-free candidate-batch pipeline (rawdtw_batch_create/run for sparse + banded batches).
//
// What it replaces: the DTW block of gen_chains (src/rmap.cpp:509-530) for every read of a mini-batch, i.e. all the calls
// of DTW_global_slantedbanded_antidiagonalwise (src/dtw.cpp:273-520) that align_chain (src/rmap.cpp:238-300) issues.
//
// Round 1 planned a batch in eleven launches with two host round trips and wrote 16-byte tile records for every job to
// HBM, which the DTW kernel then read back: 0.6 ms of GPU time and 4 ms of host time per 0.15 ms of DTW.  Here a batch is
// six launches (k_pre, k_mid, k_tile_first, k_stream, fold, select), none of which the host waits for:
//
//   k_pre          1024 jobs per workgroup, four consecutive jobs per thread: a job's windows from its chain's anchors
//                  (rmap.cpp:251-254, 270, 276), the slant-corrected radius (dtw.cpp:298-300) and its class.  Writes one
//                  16-byte record per job (arena offsets + packed shape) and the running sums of the tile layout INSIDE
//                  the workgroup's unit (rawdtw_internal.h: Cum); the jobs the tiles do not take (radius > 2, longer side
//                  > 73) are appended to a side list as full job records.
//   k_mid          the roles between k_pre and k_tile_first in one launch: workgroup 0 scans the units' totals (a job's
//                  place in its tile's LDS image is then a closed form of its own sums); workgroup 1 sorts the chains by
//                  part count for the fold; 65 workgroups put the side list into class order (wave-per-job first, longest
//                  first) and add up the statistics; the rest write the fold's chain records
//   k_tile_first   one thread per tile: the first job of cost bracket k by a two-level binary search (unit, then job), the
//                  image's exact region bounds
//   k_stream       ONE persistent launch per batch: every wave first takes its share of the side list's items, then
//                  workgroups pull tiles from an eight-headed queue.  Per tile: each thread turns its jobs' running sums
//                  into LDS offsets and marks the image's 16-byte chunks with their source; a flat, coalesced copy of the
//                  marked chunks; a counting sort by (radius class, longer side) in LDS; the lane-per-job DP of
//                  rawdtw_dp.h.  Nothing about a tile is ever written to HBM.
//
// No step needs a number on the host: grids are sized by the job count (known from the anchor offsets) or are
// persistent, every count lives in a device counter block.  rawdtw_batch_create only enqueues; errors and the rare
// shapes this path does not take (band wider than 256 offsets) surface in the counters, which rawdtw_batch_fetch reads
// together with the results.

#include "rawdtw_dp.h"

namespace rawdtw {

namespace {

constexpr int kT = 256;                 // threads per workgroup, everywhere in this file but k_pre
#ifndef PRE_THREADS
#define PRE_THREADS 256
#endif
#ifndef PRE_WAVES
#define PRE_WAVES 4
#endif
constexpr int kPreT = PRE_THREADS;      // k_pre: 1024 jobs per workgroup, kPreUnit / kPreT consecutive jobs per thread
constexpr uint32_t kPreUnit = 1024;     // jobs per k_pre workgroup
constexpr uint32_t kPreChains = 64;     // chain entries of a unit k_pre keeps in LDS (1.8 KB)
constexpr uint32_t kItems = kStreamItems; // jobs per thread in the tile prologue (strided)
static_assert(kStreamMaxTileJobs == kItems * 512, "tile job capacity of the 512-thread instance");

__device__ __forceinline__ int d_slanted_radius(uint32_t n, uint32_t m, int r0)
{
    const uint32_t N = n > m ? n : m, M = n > m ? m : n;
    const uint32_t x = (N - M) * (uint32_t)r0 + N - 1u; // dtw.cpp:298-300, unsigned 32-bit: extra = x / N
    uint32_t q;
    if (N < (1u << 11) && (uint32_t)r0 < (1u << 11)) {
        // x < 2^23 is exact in a float and the quotient is below 2^12: the product with the hardware reciprocal (1 ulp)
        // is within 2^-10 of it, so the truncation is the quotient or one beside it -- one remainder check settles it
        // (an integer division is some 25 instructions, three of them per job)
        q = (uint32_t)((float)x * __builtin_amdgcn_rcpf((float)N));
        const int r = (int)x - (int)(q * N);
        q = r < 0 ? q - 1u : (r >= (int)N ? q + 1u : q);
    } else q = x / N;
    return r0 + (int)q;
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

// inclusive scan of one value per thread over the workgroup (NT threads); `tmp` holds NT/64 words.  *excl (optional)
// receives the exclusive value, *total the reduction over all threads.
template <int NT = kT, typename Op>
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
    for (int w = 0; w < NT / 64; w++) {
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

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// k_pre
// ---------------------------------------------------------------------------------------------------------------------
// Latency is what this kernel costs (its arithmetic is ~19 M wave instructions, its traffic ~190 MB): every job needs
// its chain, the chain's offsets, then four anchors -- dependent loads.  So each thread takes kPer CONSECUTIVE jobs,
// every stage below issues the loads of all of them before anything waits, and the chains' offsets come from a table
// the workgroup loads into LDS once (the one barrier of the job path).
// Register budget: 99 VGPRs (104 allocated) with four jobs per thread (256 threads a workgroup; PRE_THREADS=512: two jobs,
// fewer registers, 15 % more instructions, same throughput).  In a pipeline of batches this kernel shares the SIMDs with
// the k_stream waves (96 VGPRs, four a SIMD) of the batches before it, and what it needs decides how many of its waves
// fit beside them: at 116 registers (the first version of the class-rule reuse below, before the side list's record
// stopped being parked in registers) the whole pipeline ran 10 % SLOWER, every kernel in it included; at 104 it is level.
__global__ __launch_bounds__(kPreT, PRE_WAVES) void k_pre(const StreamArgs a)
{
    constexpr uint32_t kPer = kPreUnit / kPreT;
    __shared__ uint32_t s_ocnt, s_obase, s_cls[kStreamClasses];
    __shared__ unsigned long long s_stats[3]; // tile jobs, tile bytes, side-list bytes
    const int tid = threadIdx.x;
    const uint64_t j0 = (uint64_t)blockIdx.x * kPreUnit;
    if (tid == 0) s_ocnt = 0;
    if (tid < 3) s_stats[tid] = 0;
    if (tid < (int)kStreamClasses) s_cls[tid] = 0;
    // chains [c_lo, c_hi] own the unit's jobs: job_off[c_lo] <= j0 < job_off[c_lo + 1], c_hi = the chain of the next unit's
    // first job (the host tabulates the unit -> chain map while it counts the jobs; the table ends with n_chains - 1)
    const uint64_t c_lo = a.unit_chain[blockIdx.x], c_hi = a.unit_chain[blockIdx.x + 1];
    const uint64_t jf = j0 + (uint64_t)tid * kPer; // this thread's first job
    // ---- stages 1 + 2: each job's chain and the chain's offsets.  A unit spans few chains (a chain has one job per part:
    // some 130 in the bench batch), so the workgroup loads the offsets of chains c_lo .. c_hi + 1 into LDS in ONE round trip
    // and every thread searches there: the job path then has three dependent trips to memory (unit -> chains, the
    // chains' offsets, the anchors) instead of a dozen (binary search, one probe per further job, offsets, anchors).
    // Units of more than kPreChains - 2 chains (batches of very short chains) search the arrays in memory instead. ----
    __shared__ uint64_t s_jo[kPreChains], s_ao[kPreChains], s_rb[kPreChains];
    __shared__ uint32_t s_qb[kPreChains];
    const uint32_t span = (uint32_t)min<uint64_t>(c_hi - c_lo, 0xfffffff0ull);
    uint32_t ce[kPer]; // the job's chain, counted from c_lo
    bool have[kPer];
    uint64_t jo[kPer], a0[kPer], a1[kPer], rb[kPer];
    uint32_t qb[kPer];
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) have[k] = jf + k < a.n_jobs;
    if (span + 2u <= kPreChains) {
        if ((uint32_t)tid < span + 2u) { s_jo[tid] = a.job_off[c_lo + tid]; s_ao[tid] = a.anchor_off[c_lo + tid]; }
        if ((uint32_t)tid <= span) { s_rb[tid] = a.ref_base[c_lo + tid]; s_qb[tid] = a.read_base[c_lo + tid]; }
        __syncthreads();
        // the last entry e <= span with job_off <= jf (chains without jobs share their successor's offset: the LAST of them
        // owns the job), by doubling steps
        uint32_t e = 0;
        for (uint32_t step = kPreChains / 2; step > 0; step >>= 1)
            if (e + step <= span && s_jo[e + step] <= jf) e += step;
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            if (have[k]) while (s_jo[e + 1] <= jf + k) e++; // (job_off[c_hi + 1] is beyond the unit: e stays <= span)
            ce[k] = e;
            jo[k] = s_jo[e]; a0[k] = s_ao[e]; a1[k] = s_ao[e + 1]; rb[k] = s_rb[e]; qb[k] = s_qb[e];
        }
    } else {
        uint64_t c;
        {
            uint64_t lo = c_lo, hi = c_hi; // invariant: job_off[lo] <= jf, answer in [lo, hi]
            for (uint64_t sp = c_hi - c_lo; sp > 0; sp >>= 1) {
                const uint64_t mid = lo + ((hi - lo + 1) >> 1);
                if (hi > lo) { if (a.job_off[mid] <= jf) lo = mid; else hi = mid - 1; }
            }
            c = lo;
        }
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            if (have[k]) while (a.job_off[c + 1] <= jf + k) c++;
            ce[k] = (uint32_t)(c - c_lo);
            const uint64_t cc = have[k] ? c : c_lo;
            jo[k] = a.job_off[cc]; a0[k] = a.anchor_off[cc]; a1[k] = a.anchor_off[cc + 1]; rb[k] = a.ref_base[cc]; qb[k] = a.read_base[cc];
        }
    }
    // ---- stage 3: the anchors: this part's (s, e), the end of the part before (sp) and of the part after (en) ----
    rawdtw_anchor_t S[kPer], E[kPer], SP[kPer], EN[kPer];
    uint32_t P[kPer], parts[kPer];
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        parts[k] = (uint32_t)(a1[k] - a0[k]) - 1u;
        P[k] = (uint32_t)(jf + k - jo[k]);
        const uint64_t g = a0[k] + parts[k] - P[k]; // rmap.cpp:253-254: part p runs from anchors[parts - p] to anchors[parts - p - 1]
        S[k] = E[k] = SP[k] = EN[k] = rawdtw_anchor_t{0, 0};
        if (have[k]) {
            S[k] = a.anchors[g]; E[k] = a.anchors[g - 1];
            if (P[k] > 0) SP[k] = a.anchors[g + 1];
            if (P[k] + 1 < parts[k]) EN[k] = a.anchors[g - 2];
        }
    }
    // ---- stage 4: geometry, class, record ----
    uint32_t my_tiles = 0;
    unsigned long long my_bytes = 0, my_obytes = 0;
    uint64_t lpos[kPer];  // running sums over this thread's jobs: event floats | reference floats << 32
    uint32_t lcost[kPer]; // ... and cost
    uint64_t run_pos = 0;
    uint32_t run_cost = 0;
    // Tile class without the division of dtw.cpp:298-300: with d = N - M the radius is R = r0 + ceil(d r0 / N), so
    //     R <= Rm  <=>  r0 <= Rm and d r0 <= (Rm - r0) N,
    // and for a tile job (Rm <= 3 and r0 >= 1: the ceiling is 0, 1 or 2)  R = r0 + (d r0 > 0) + (d r0 > N).
    // (32-bit products: N <= lane_max_n < 128 and r0 <= 3 wherever the result counts.)  The side list's jobs -- one in two
    // hundred -- get their radius from d_slanted_radius.
    auto shape_class = [&](const uint32_t n, const uint32_t m, int &R) {
        int r0 = (int)((float)n * a.frac); // rmap.cpp:276, fp32 product
        r0 = r0 > 1 ? r0 : 1;
        const uint32_t N = n > m ? n : m, dr = (N - (n > m ? m : n)) * (uint32_t)r0;
        R = r0 + (dr > 0u ? 1 : 0) + (dr > N ? 1 : 0);
        return N <= a.lane_max_n && r0 <= a.lane_max_radius && dr <= (uint32_t)(a.lane_max_radius - r0) * N;
    };
    auto tile_class = [&](const rawdtw_anchor_t &s, const rawdtw_anchor_t &e) { // would the part s -> e be a tile job?
        int R;
        return e.target_position >= s.target_position && e.query_position >= s.query_position &&
               shape_class(e.query_position - s.query_position + 1, e.target_position - s.target_position + 1, R);
    };
    // the class rule of every job of the thread first: consecutive jobs are consecutive parts of one chain (or the chain
    // ends), so the part before job k and the part after it are the thread's own jobs k - 1 and k + 1 -- only the part
    // before the first job and the one after the last need a rule evaluation of their own (6 instead of 12 a thread)
    bool tc[kPer], okk[kPer];
    int Rk[kPer];
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const rawdtw_anchor_t s = S[k], e = E[k];
        const bool asc = have[k] && e.target_position >= s.target_position && e.query_position >= s.query_position;
        const uint32_t m = e.target_position - s.target_position + 1, n = e.query_position - s.query_position + 1;
        tc[k] = shape_class(n, m, Rk[k]) && asc;
        if (asc && !tc[k]) { // the side list's jobs: the radius by the reference's formula
            int r0 = (int)((float)n * a.frac);
            r0 = r0 > 1 ? r0 : 1;
            Rk[k] = d_slanted_radius(n, m, r0);
        }
        okk[k] = asc && !((uint64_t)qb[k] + s.query_position + n > a.n_ev || rb[k] + s.target_position + m > a.n_ref ||
                          n >= 0x7fffffffu || m >= 0x7fffffffu);
    }
    bool prev_tc[kPer], next_tc[kPer];
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        if (k == 0) prev_tc[k] = have[0] && P[0] > 0 && tile_class(SP[0], S[0]);
        else prev_tc[k] = have[k] && ce[k] == ce[k - 1] && tc[k - 1]; // (another chain: job k is its first part)
        if (k + 1 < kPer) next_tc[k] = have[k + 1 < kPer ? k + 1 : k] && ce[k + 1 < kPer ? k + 1 : k] == ce[k] && tc[k + 1 < kPer ? k + 1 : k];
        else next_tc[k] = have[k] && P[k] + 1 < parts[k] && tile_class(E[k], EN[k]);
    }
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        if (!have[k]) continue;
        const uint64_t j = jf + k;
        const rawdtw_anchor_t s = S[k], e = E[k];
        const bool ok = okk[k];
        const uint64_t ref_off = rb[k] + s.target_position;
        const uint32_t read_off = qb[k] + s.query_position;
        const uint32_t m = e.target_position - s.target_position + 1;
        const uint32_t n = e.query_position - s.query_position + 1;
        const bool excl = P[k] != parts[k] - 1; // rmap.cpp:270
        const int R = Rk[k];
        const uint32_t N = n > m ? n : m, M = n > m ? m : n;
        const bool tile = ok && tc[k]; // (its radius is >= 1: rmap.cpp:276)
        uint32_t meta = 0;
        if (!ok) atomicMin(&a.cnt[kCntBad], (unsigned long long)j);
        if (tile) {
            // a tile job continues its chain's run when the part before it is a tile job too (consecutive parts share their
            // anchor element: the run is one contiguous piece of each arena), and ends the run when the part after it is
            // not one: the run's end is then padded to a 16-byte boundary, so that two runs never share a chunk of the image
            const bool starts = !prev_tc[k];
            const bool ends = !next_tc[k];
            meta = N | (M << 7) | ((uint32_t)R << 14) | ((excl ? 1u : 0u) << 16) | ((n < m ? 1u : 0u) << 17) |
                   ((starts ? 1u : 0u) << 18) | kMetaTile | (ends ? kMetaEnds : 0u);
            my_tiles++;
            my_bytes += 4ull * ((unsigned long long)n + m) + 36ull;
        } else if (ok) {
            const uint32_t K = (uint32_t)R + 1u;
            uint32_t cls;
            if (R >= 1 && R <= a.side_lane_radius && N <= a.lane_max_n) cls = kClsL0 + side_lane_bucket(N);
            else if (K <= 8) cls = kClsM0 + side_lane_bucket(N);
            else if (K <= 16) cls = kClsG16;
            else if (K <= 256) cls = N >= 1024 ? kClsW0 : N >= 256 ? kClsW0 + 1 : N >= 64 ? kClsW0 + 2 : kClsW0 + 3;
            else cls = 0xff;
            if (cls == 0xff) atomicAdd(&a.cnt[kCntUnsupported], 1ull);
            else meta = (cls + 1u) | ((uint32_t)R << 8); // (side list: class and radius travel in the record until the append below)
            my_obytes += 4ull * ((unsigned long long)n + m) + 36ull;
        }
        const JobRec rec{ref_off, read_off, (meta & kMetaTile) ? meta : 0u};
        a.jrec[j] = rec;
        {
            const Cum d = job_cum(rec, a.min_cost8);
            run_pos += (uint64_t)d.read | ((uint64_t)d.ref << 32);
            run_cost += (uint32_t)d.cost;
            lpos[k] = run_pos; lcost[k] = run_cost;
        }
        // the side list: rare (a few jobs per workgroup)
        if (!(meta & kMetaTile) && meta) {
            const uint32_t cls = (meta & 0xffu) - 1u;
            const uint32_t slot = atomicAdd(&s_ocnt, 1u);
            atomicAdd(&s_cls[cls], 1u);
            // parked in registers-free form: recomputed from the record at append time (below) would need the shape again,
            // so stage it in the side list's own staging slot right away: the slot index is fixed, only the base is not
            P[k] = slot; parts[k] = cls | 0x80000000u;
            S[k] = rawdtw_anchor_t{n, m}; E[k] = rawdtw_anchor_t{(uint32_t)R, excl ? kFlagExcludeLast : 0u};
            SP[k] = rawdtw_anchor_t{read_off, (uint32_t)j};
        } else parts[k] = 0;
    }
    unsigned long long obase = 0; // (thread 0) the side list's base for this workgroup's jobs
    // The running sums of the tile layout (rawdtw_internal.h: Cum), local to the unit: a scan over the workgroup's threads;
    // k_mid's first workgroup turns the units' totals into their offsets.  (No pass over the batch for a global scan.)
    {
        __shared__ uint64_t s_wpos[kPreT / 64];
        __shared__ uint32_t s_wcost[kPreT / 64];
        uint64_t ipos = run_pos;
        uint32_t icost = run_cost;
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t op = (uint64_t)__shfl_up((long long)ipos, d);
            const uint32_t oc = (uint32_t)__shfl_up((int)icost, d);
            if (lane >= d) { ipos += op; icost += oc; }
        }
        if (lane == 63) { s_wpos[wv] = ipos; s_wcost[wv] = icost; }
        __syncthreads();
        // (every thread is through its jobs: the side list's count is final.  Its base is one returning atomic on one word
        // per workgroup -- the workgroups of a round arrive together and the word takes ~88 a microsecond -- so it is
        // issued here and waited for at the end, behind the sums' stores)
        if (tid == 0 && s_ocnt) obase = atomicAdd(&a.cnt[kCntOthers], (unsigned long long)s_ocnt);
        uint64_t ppos = 0, tpos = 0;
        uint32_t pcost = 0, tcost = 0;
#pragma unroll
        for (int w = 0; w < kPreT / 64; w++) {
            if (w < wv) { ppos += s_wpos[w]; pcost += s_wcost[w]; }
            tpos += s_wpos[w]; tcost += s_wcost[w];
        }
        const uint64_t epos = ppos + ipos - run_pos;   // sums before this thread's first job, inside the unit
        const uint32_t ecost = pcost + icost - run_cost;
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++)
            if (have[k]) { a.cpos[jf + k] = epos + lpos[k]; a.ccost[jf + k] = ecost + lcost[k]; }
        if (tid == 0) { a.unit_pos[blockIdx.x] = tpos; a.unit_cost[blockIdx.x] = (uint64_t)tcost; }
    }
    // totals: per workgroup, then one record per unit (reduced by k_mid); the side list's base: one returning atomic per
    // workgroup that has side-list jobs
    {
        // tile jobs (<= 4 a thread) and their bytes (< 2^12 a thread) share one 32-bit word through the wave's reduction;
        // the side list's bytes are rare and unbounded: an atomic of their own from the threads that have any
        uint32_t packed = my_tiles | ((uint32_t)my_bytes << 10);
        for (int off = 32; off > 0; off >>= 1) packed += (uint32_t)__shfl_down((int)packed, off);
        if ((tid & 63) == 0) {
            atomicAdd(&s_stats[0], (unsigned long long)(packed & 1023u)); atomicAdd(&s_stats[1], (unsigned long long)(packed >> 10));
        }
        if (my_obytes) atomicAdd(&s_stats[2], my_obytes);
    }
    __syncthreads();
    if (tid == 0) s_obase = (uint32_t)obase;
    if (tid < 3) a.unit_stats[3ull * blockIdx.x + tid] = s_stats[tid];
    if (tid < (int)kStreamClasses && s_cls[tid]) atomicAdd(&a.cnt[kCntCls0 + tid], (unsigned long long)s_cls[tid]);
    __syncthreads();
    if (s_ocnt) {
        const uint64_t base = s_obase;
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            if (!have[k] || !(parts[k] & 0x80000000u)) continue;
            const uint64_t q = base + P[k];
            if (q < a.others_cap) { // (beyond the capacity: kCntOthers > others_cap tells rawdtw_batch_fetch to take the job-list path)
                DevJob d;
                d.ref_off = a.jrec[jf + k].ref_off; // (this thread's own record, re-read: a rare path, and two registers less in the common one)
                d.read_off = SP[k].target_position; d.n = S[k].target_position; d.m = S[k].query_position;
                d.R = (int32_t)E[k].target_position; d.flags = E[k].query_position; d.aux = SP[k].query_position;
                a.omix[q] = d; a.ocls[q] = (uint8_t)(parts[k] & 0xffu);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// unit scan (a role of k_mid): the units' totals -> the sums BEFORE each unit (exclusive, in place; entry n_units = the batch totals).
// One workgroup: a few thousand entries.
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void unit_scan_body(const StreamArgs &a)
{
    constexpr uint32_t kU = 8; // consecutive units per thread and round (NT * kU units a round: two or three rounds a bench batch)
    const uint64_t n_units = (a.n_jobs + kPreUnit - 1) / kPreUnit;
    __shared__ uint64_t s_p[NT / 64], s_c[NT / 64];
    __shared__ uint64_t s_carry[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) { s_carry[0] = 0; s_carry[1] = 0; }
    __syncthreads();
    for (uint64_t base = 0; base < n_units; base += (uint64_t)NT * kU) {
        const uint64_t u0 = base + (uint64_t)tid * kU;
        uint64_t vp[kU], vc[kU];
#pragma unroll
        for (uint32_t q = 0; q < kU; q++) {
            vp[q] = u0 + q < n_units ? a.unit_pos[u0 + q] : 0ull;
            vc[q] = u0 + q < n_units ? a.unit_cost[u0 + q] : 0ull;
        }
        uint64_t tp = 0, tc = 0; // this thread's units
#pragma unroll
        for (uint32_t q = 0; q < kU; q++) { tp += vp[q]; tc += vc[q]; }
        uint64_t ip = tp, ic = tc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t op = (uint64_t)__shfl_up((long long)ip, d), oc = (uint64_t)__shfl_up((long long)ic, d);
            if (lane >= (uint32_t)d) { ip += op; ic += oc; }
        }
        if (lane == 63) { s_p[wv] = ip; s_c[wv] = ic; }
        __syncthreads();
        uint64_t pp = s_carry[0] + ip - tp, pc = s_carry[1] + ic - tc, ap = 0, ac = 0; // sums before this thread's first unit
#pragma unroll
        for (uint32_t w = 0; w < NT / 64; w++) {
            if (w < wv) { pp += s_p[w]; pc += s_c[w]; }
            ap += s_p[w]; ac += s_c[w];
        }
#pragma unroll
        for (uint32_t q = 0; q < kU; q++) {
            if (u0 + q < n_units) { a.unit_pos[u0 + q] = pp; a.unit_cost[u0 + q] = pc; }
            pp += vp[q]; pc += vc[q];
        }
        __syncthreads();
        if (tid == 0) { s_carry[0] += ap; s_carry[1] += ac; }
        __syncthreads();
    }
    if (tid == 0) { a.unit_pos[n_units] = s_carry[0]; a.unit_cost[n_units] = s_carry[1]; }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_tile_first: tile k = the jobs whose exclusive running cost lies in [k * width, (k + 1) * width).  Every job has a
// positive cost, so tile_first[k] = 1 + (first job whose inclusive running cost reaches k * width).
// ---------------------------------------------------------------------------------------------------------------------
// (one-wave workgroups: every thread chases ~25 dependent, scattered loads -- spread over as many CUs as there are waves)
__global__ __launch_bounds__(64) void k_tile_first(const StreamArgs a)
{
    const uint64_t k = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    const uint64_t n = a.n_jobs;
    const uint64_t n_units = (n + kPreUnit - 1) / kPreUnit;
    auto cost_incl = [&](uint64_t j) { return a.unit_cost[j / kPreUnit] + a.ccost[j]; }; // inclusive running cost of job j
    const uint64_t n_tiles = (n >= 2 ? cost_incl(n - 2) / a.width8 : 0ull) + 1ull;
    if (k == 0) {
        a.cnt[kCntTiles] = n_tiles;
        if (n_tiles > a.tiles_cap) atomicMin(&a.cnt[kCntOverflow], 0ull);
    }
    if (k > n_tiles || k > a.tiles_cap) return;
    auto first_of = [&](uint64_t q) -> uint64_t {
        if (q == 0) return 0;
        if (q >= n_tiles) return n;
        const uint64_t key = q * a.width8;
        // first job whose inclusive cost reaches the key: the unit first (its end = the next unit's offset), then the job
        uint64_t lo = 0, hi = n_units;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (a.unit_cost[mid + 1] >= key) hi = mid; else lo = mid + 1;
        }
        const uint64_t u = lo < n_units ? lo : n_units - 1, rel = key - a.unit_cost[u];
        uint64_t jl = u * kPreUnit, jh = jl + kPreUnit < n ? jl + kPreUnit : n;
        while (jl < jh) {
            const uint64_t mid = (jl + jh) >> 1;
            if ((uint64_t)a.ccost[mid] >= rel) jh = mid; else jl = mid + 1;
        }
        return jl + 1;
    };
    auto pos_before = [&](uint64_t j) -> Cum { // sums before job j
        if (j == 0) return Cum{0u, 0u, 0ull};
        return cum_of(a.unit_pos[(j - 1) / kPreUnit] + a.cpos[j - 1]);
    };
    TileInfo t{};
    const uint64_t first = first_of(k), end = first_of(k + 1);
    t.first = (uint32_t)first;
    t.n = (uint32_t)(end - first);
    t.first_tile = t.n;
    if (t.n) {
        // The regions run from the first tile-class job's first chunk to the last one's last element: with these exact
        // bounds every chunk of the image belongs to a job (k_stream needs no clearing pass).  Jobs of other classes
        // inside the range are rare: the walks below are one or two steps.
        uint64_t f = first, l = end;
        while (f < end && !(a.jrec[f].meta & kMetaTile)) f++;
        while (l > f && !(a.jrec[l - 1].meta & kMetaTile)) l--;
        if (f < end) {
            const JobRec rf = a.jrec[f], rl = a.jrec[l - 1];
            const Cum cf = pos_before(f);
            const Cum cl = pos_before(l - 1);
            const bool sf = (rf.meta >> 18) & 1u, sl = (rl.meta >> 18) & 1u, swl = (rl.meta >> 17) & 1u;
            const uint32_t Nl = rl.meta & 127u, Ml = (rl.meta >> 7) & 127u;
            t.base_read = image_pos(cf.read, rf.read_off, sf) & ~3u;
            t.base_ref = image_pos(cf.ref, rf.ref_off, sf) & ~3u;
            const uint32_t read_end = image_pos(cl.read, rl.read_off, sl) + (swl ? Ml : Nl);
            const uint32_t ref_end = image_pos(cl.ref, rl.ref_off, sl) + (swl ? Nl : Ml);
            t.ref_region = (read_end - t.base_read + 3u) & ~3u; // LDS float offset of the reference region
            t.image = t.ref_region + ((ref_end - t.base_ref + 3u) & ~3u);
            t.first_tile = (uint32_t)(f - first);
        }
    }
    a.tiles[k] = t;
}

// ---------------------------------------------------------------------------------------------------------------------
// side list in class order (a role of k_mid): wave-per-job classes first, longest first; then 16-lane groups, the lane classes
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void others_body(const StreamArgs &a, const uint32_t group, const uint32_t groups_all)
{
    if (group == groups_all - 1) { // the last workgroup adds up k_pre's per-unit totals instead (a few thousand records)
        const uint64_t n_units = (a.n_jobs + kPreUnit - 1) / kPreUnit;
        unsigned long long t[3] = {0, 0, 0};
        for (uint64_t u = threadIdx.x; u < n_units; u += NT)
            for (int q = 0; q < 3; q++) t[q] += a.unit_stats[3 * u + q];
        __shared__ unsigned long long s_t[3];
        if (threadIdx.x < 3) s_t[threadIdx.x] = 0;
        __syncthreads();
        for (int q = 0; q < 3; q++) {
            for (int off = 32; off > 0; off >>= 1) t[q] += __shfl_down(t[q], off);
            if ((threadIdx.x & 63) == 0) atomicAdd(&s_t[q], t[q]);
        }
        __syncthreads();
        if (threadIdx.x == 0) { a.cnt[kCntTileJobs] = s_t[0]; a.cnt[kCntTileBytes] = s_t[1]; a.cnt[kCntOtherBytes] = s_t[2]; }
        return;
    }
    // Each workgroup orders one contiguous slice of the unordered list: class counts of the slice in LDS, ONE returning
    // atomic per class and workgroup for the slice's places (same-address atomics run near 88 per microsecond: a wave-level
    // scheme spends the kernel there once the list has 10^5 entries), then the scatter through LDS cursors.
    const uint64_t n_other = min<uint64_t>(a.cnt[kCntOthers], a.others_cap);
    const uint64_t groups = groups_all - 1, per = (n_other + groups - 1) / groups;
    const uint64_t lo = min(n_other, (uint64_t)group * per), hi = min(n_other, lo + per);
    __shared__ uint32_t s_n[kStreamClasses];
    __shared__ uint64_t s_at[kStreamClasses];
    if (threadIdx.x < kStreamClasses) s_n[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = lo + threadIdx.x; i < hi; i += NT) atomicAdd(&s_n[a.ocls[i]], 1u);
    __syncthreads();
    if (threadIdx.x < kStreamClasses) {
        uint64_t base = 0;
        for (uint32_t c = 0; c < threadIdx.x; c++) base += a.cnt[kCntCls0 + c];
        const uint32_t mine = s_n[threadIdx.x];
        s_at[threadIdx.x] = base + (mine ? atomicAdd(&a.cnt[kCntCur0 + threadIdx.x], (unsigned long long)mine) : 0ull);
        s_n[threadIdx.x] = 0;
    }
    __syncthreads();
    for (uint64_t i = lo + threadIdx.x; i < hi; i += NT) {
        const uint32_t cls = a.ocls[i];
        const uint64_t pos = s_at[cls] + atomicAdd(&s_n[cls], 1u);
        if (pos < a.others_cap) a.ojobs[pos] = a.omix[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// chain records for the fold (ChainDesc) and the key of the fold order (part count, clamped)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void chain_desc_body(const StreamArgs &a, ChainDesc *__restrict__ chains, const uint64_t c)
{
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
}

// Fold order: chains by part count, longest first (the lane-per-chain fold gives a wave 64 chains of similar length).
// One workgroup: a counting sort over 1024 length buckets in LDS (the order inside a bucket does not matter).
template <int NT>
__device__ __forceinline__ void fold_order_body(const uint64_t n_chains, const uint64_t *__restrict__ job_off, uint32_t *__restrict__ order)
{
    constexpr uint32_t kBins = 1024, kB = kBins / NT; // consecutive buckets per thread in the scan
    constexpr uint32_t kC = 4;                        // chains per thread and round: their offsets are all requested before any is used
    __shared__ uint32_t hist[kBins];
    __shared__ uint32_t wsum[NT / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    auto bucket = [](uint64_t parts) { return 1023u - (uint32_t)min<uint64_t>(parts, 1023ull); }; // a chain's part count, clamped
    auto keys = [&](uint64_t c0, uint32_t (&b)[kC]) { // buckets of chains c0 + q * NT
        uint64_t lo[kC], hi[kC];
#pragma unroll
        for (uint32_t q = 0; q < kC; q++) {
            const uint64_t c = min(c0 + (uint64_t)q * NT, n_chains - 1);
            lo[q] = job_off[c]; hi[q] = job_off[c + 1];
        }
#pragma unroll
        for (uint32_t q = 0; q < kC; q++) b[q] = bucket(hi[q] - lo[q]);
    };
    for (uint32_t i = tid; i < kBins; i += NT) hist[i] = 0;
    __syncthreads();
    for (uint64_t c0 = tid; c0 < n_chains; c0 += (uint64_t)NT * kC) {
        uint32_t b[kC];
        keys(c0, b);
#pragma unroll
        for (uint32_t q = 0; q < kC; q++)
            if (c0 + (uint64_t)q * NT < n_chains) atomicAdd(&hist[b[q]], 1u);
    }
    __syncthreads();
    uint32_t v[kB], mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < kB; q++) { v[q] = hist[tid * kB + q]; mine += v[q]; }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t at = incl - mine;
    for (uint32_t w = 0; w < wv; w++) at += wsum[w];
#pragma unroll
    for (uint32_t q = 0; q < kB; q++) { hist[tid * kB + q] = at; at += v[q]; } // exclusive start of the bucket
    __syncthreads();
    for (uint64_t c0 = tid; c0 < n_chains; c0 += (uint64_t)NT * kC) {
        uint32_t b[kC];
        keys(c0, b);
#pragma unroll
        for (uint32_t q = 0; q < kC; q++)
            if (c0 + (uint64_t)q * NT < n_chains) order[atomicAdd(&hist[b[q]], 1u)] = (uint32_t)(c0 + (uint64_t)q * NT);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_mid: everything between k_pre and k_tile_first as the roles of ONE launch -- they are independent of each other and
// each is a latency chain of one or a few workgroups: workgroup 0 scans the units' totals, workgroup 1 sorts the chains
// for the fold, the next kMidOthers + 1 order the side list (the last of them adds up the per-unit statistics), the rest
// write the chain records.  (As four launches they were 60 us of mostly idle chip in a batch's critical path, and three
// more launches for the host to issue.)
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kMidOthers = 64;
#ifndef MID_THREADS
#define MID_THREADS 1024
#endif
constexpr int kMidT = MID_THREADS;
__global__ __launch_bounds__(kMidT) void k_mid(const StreamArgs a, ChainDesc *__restrict__ chains, uint32_t *__restrict__ order)
{
    const uint32_t b = blockIdx.x;
    if (b == 0) { if (a.n_jobs) unit_scan_body<kMidT>(a); }
    else if (b == 1) { if (a.n_chains) fold_order_body<kMidT>(a.n_chains, a.job_off, order); }
    else if (b < 2 + kMidOthers + 1) { if (a.n_jobs) others_body<kMidT>(a, b - 2, kMidOthers + 1); }
    else chain_desc_body(a, chains, (uint64_t)(b - (2 + kMidOthers + 1)) * kMidT + threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_stream: the batch's one DTW launch
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kSortBins = 192; // bin = (radius <= 2 ? 80 : 0) + (79 - longer side): radius 3 first, each run longest first
template <int TT> struct SortTable {
    uint32_t hist[kSortBins];
    uint16_t perm[kItems * TT];
};

// One sorted chunk of a tile: 64 lanes, one job each.  The chunks of a tile are cut from one order -- the jobs of radius 3
// first (none unless "stream_tile_radius" is 3), then the others, both runs by longer side, descending; a wave takes the
// shortest body that covers its radii.
__device__ __forceinline__ float stream_lane_job(const float *LA, const float *LB, uint32_t N, uint32_t M, uint32_t R, bool excl, bool act)
{
    const unsigned long long r12 = __ballot(R <= 2u), r1 = __ballot(R == 1u);
    uint32_t n_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)N); // lane 0 leads the wave's first run
    if (r12) {                                                         // ... and the first lane of radius <= 2 the second
        const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)r12) - 1);
        n_max = max(n_max, (uint32_t)__builtin_amdgcn_readlane((int)N, l));
    }
    float res;
    if (~r1 == 0ull) res = lane_dp_r1(LA, LB, N, M, n_max);
    else if (~r12 == 0ull) res = lane_dp_r12(LA, LB, N, M, R, n_max);
    else res = lane_dp_gen(LA, LB, N, M, R, n_max);
    if (act && excl) res = res - dist(LA[N - 1], LB[M - 1]);
    return res;
}

// 64 jobs of the side list's lane classes: one lane per job, operands straight from the arenas (the jobs of a wave come
// from all over the batch: nothing to stage together; they are 2 % of a sparse batch's jobs)
template <int SLOTS>
__device__ __forceinline__ void lane_global_wave(const DevJob *__restrict__ jobs, uint32_t count, uint32_t wave, int lane,
                                                 const float *__restrict__ ev, const float *__restrict__ ref, float *__restrict__ out)
{
    const uint32_t idx = wave * 64u + (uint32_t)lane;
    const bool have = idx < count;
    const DevJob jb = jobs[have ? idx : count - 1u];
    const float *A = ev + jb.read_off;
    const float *B = ref + jb.ref_off;
    uint32_t N = jb.n, M = jb.m;
    if (N < M) {
        const float *tp = A; A = B; B = tp;
        const uint32_t tn = N; N = M; M = tn;
    }
    uint32_t n_max = N;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, d));
    n_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_max);
    float res = SLOTS == 4 ? lane_dp_gen<true>(A, B, N, M, (uint32_t)jb.R, n_max) : lane_dp_k8(A, B, N, M, jb.R, n_max);
    if (have) {
        if (jb.flags & kFlagExcludeLast) res = res - dist(A[N - 1], B[M - 1]);
        out[jb.aux] = res;
    }
}

} // namespace

// Tile queue: one returning atomic on a single word saturates near 88 dequeues per microsecond (MI355X_MICROARCH.md),
// which a batch's ten thousand tiles would reach; eight heads on lines of their own, each dealing every eighth tile.
// A workgroup starts on the head of its block index and moves on when a head runs dry.
__device__ __forceinline__ uint32_t next_tile(const StreamArgs &a, uint32_t &head, uint32_t n_tiles)
{
    if (a.debug & 8u) { // timing experiments: tiles dealt by block index, no queue
        const uint32_t t = head;
        head += gridDim.x;
        return t < n_tiles ? t : 0xffffffffu;
    }
    for (uint32_t tries = 0; tries < 8; tries++) {
        const uint32_t h = (head + tries) & 7u;
        const uint32_t k = (uint32_t)atomicAdd(&a.cnt[kCntHeads + 16 * h], 1ull);
        const uint64_t t = (uint64_t)k * 8u + h;
        if (t < n_tiles) { head = h; return (uint32_t)t; }
    }
    return 0xffffffffu;
}

// TT threads per workgroup (256 or 512): a tile's range holds up to 4 * TT jobs.  More jobs per tile = fuller and more
// uniform waves after the sort (the lane class's first wave carries the tile's few long jobs) and half as many tiles.
template <int TT>
__global__ __launch_bounds__(TT, TT == 256 ? 5 : 4) void k_stream(const StreamArgs a, const uint32_t lds_floats)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *win = smem;                                                            // the tile's LDS image
    uint2 *rec = reinterpret_cast<uint2 *>(smem + lds_floats);                    // one record per job of the tile's range
    constexpr uint32_t kMaxJobs = kItems * TT;
    SortTable<TT> &sc = *reinterpret_cast<SortTable<TT> *>(smem + lds_floats + 2 * kMaxJobs);
    __shared__ uint32_t s_tmp[2];
    __shared__ TileInfo s_tile[2];
    __shared__ uint32_t s_seq;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = (uint32_t)tid >> 6;

    // ---- the side list: wave-cooperative jobs dealt over ALL waves of the grid, wave-per-job classes (longest first) to the
    // first waves: a long job starts at once and runs next to the tiles instead of behind them ----
    if (!(a.debug & 4u)) {
        uint64_t n_w = 0;
#pragma unroll
        for (uint32_t c = kClsW0; c < kClsW0 + 4; c++) n_w += a.cnt[kCntCls0 + c];
        const uint64_t n_g16 = a.cnt[kCntCls0 + kClsG16];
        uint64_t n_l = 0, n_m = 0;
#pragma unroll
        for (uint32_t c = kClsL0; c < kClsL0 + kClsLCount; c++) n_l += a.cnt[kCntCls0 + c];
#pragma unroll
        for (uint32_t c = kClsM0; c < kClsM0 + kClsMCount; c++) n_m += a.cnt[kCntCls0 + c];
        if (n_w + n_g16 + n_l + n_m <= a.others_cap) { // (otherwise the batch is redone through the job-list path)
            // item order = list order: wave-per-job (longest first), 16-lane groups, lane-per-job (4 slots, then 8; by length)
            const uint64_t it_g16 = (n_g16 + 3) / 4, it_l = (n_l + 63) / 64, it_m = (n_m + 63) / 64, items = n_w + it_g16 + it_l + it_m;
            // The waves that draw the long items (a wave-per-job or 16-lane item: 30..150 us) take the short ones as well, in
            // further rounds: their workgroups start on tiles late anyway, and every other workgroup starts at once.
            // (debug 2048: all waves of the grid share the items, one round.)
            const uint32_t all_waves = gridDim.x * (TT / 64), widx = __builtin_amdgcn_readfirstlane(blockIdx.x * (TT / 64) + wv);
            const uint32_t side_waves = (a.debug & 2048u) ? all_waves : (uint32_t)min<uint64_t>(all_waves, max<uint64_t>((n_w + it_g16 + 3) & ~3ull, 64));
            const uint32_t n_items = (uint32_t)min<uint64_t>(items, 0xffffffffull); // (the list's capacity is a quarter of the jobs)
            // dealt like a snake: the wave that drew the longest item of a round draws the shortest of the next
            // (item r * S + w in even rounds, r * S + S - 1 - w in odd ones: the stride alternates between 2S - 1 - 2w and 1 + 2w)
            for (uint32_t it = widx < side_waves ? widx : n_items, step = (a.debug & 4096u) ? side_waves : 2u * side_waves - 1u - 2u * widx; it < n_items;
                 it += step, step = 2u * side_waves - step) {
                if (it < n_w) {
                    if (a.debug & 32u) continue;
                    // the longest jobs bound the launch: a job of hundreds of columns is one dependent chain, and shares its
                    // SIMD with the waves around it -- it goes first in the issue order
                    const DevJob jb = a.ojobs[it];
                    const uint32_t len = max(jb.n, jb.m);
                    if (len >= 256u) __builtin_amdgcn_s_setprio(3);
                    else if (len >= 96u) __builtin_amdgcn_s_setprio(2);
                    else __builtin_amdgcn_s_setprio(1);
                    wreg_small_job(jb, lane, a.ev, a.ref, a.out);
                    __builtin_amdgcn_s_setprio(0);
                }
                else if (a.debug & 64u) continue;
                else if (it < n_w + it_g16) grp_wave<16>(a.ojobs + n_w, (uint32_t)n_g16, (uint32_t)(it - n_w), lane, a.ev, a.ref, a.out);
                else if (it < n_w + it_g16 + it_l) lane_global_wave<4>(a.ojobs + n_w + n_g16, (uint32_t)n_l, (uint32_t)(it - n_w - it_g16), lane, a.ev, a.ref, a.out);
                else lane_global_wave<8>(a.ojobs + n_w + n_g16 + n_l, (uint32_t)n_m, (uint32_t)(it - n_w - it_g16 - it_l), lane, a.ev, a.ref, a.out);
            }
        }
    }

    // ---- tiles, pulled from the queue two ahead: while tile i is computed, tile i + 1's records are already on their way
    // into registers and tile i + 2's number is being dequeued -- no global round trip sits between two tiles ----
    const uint32_t n_tiles = (uint32_t)min<unsigned long long>(a.cnt[kCntTiles], (unsigned long long)a.tiles_cap);
    // thread 0's queue state: `head`, the resolved number of the next tile, the raw ticket of the one after it
    uint32_t head = (a.debug & 8u) ? blockIdx.x : (blockIdx.x & 7u), t_next = 0xffffffffu;
    if (tid == 0) {
        const uint32_t t0 = next_tile(a, head, n_tiles);
        t_next = t0 != 0xffffffffu ? next_tile(a, head, n_tiles) : 0xffffffffu;
        s_tile[0] = t0 != 0xffffffffu ? a.tiles[t0] : TileInfo{};
        s_seq = 0;
    }
    for (uint32_t b = tid; b < kSortBins; b += TT) sc.hist[b] = 0;
    __syncthreads();
    uint32_t slot = 0, seq = 1; // seq: tag of the image's chunk marks (stale marks of an earlier tile never match)
    TileInfo D = s_tile[0];
    // thread t owns jobs t, t + TT, ... of the tile's range (a tile of 500 jobs keeps every wave busy with two of them; its
    // records load coalesced); the sums BEFORE each job come from k_pre's running sums: the unit's offset + the job before
    JobRec jr[kItems];
    uint64_t ce[kItems];
    auto fetch_records = [&](const TileInfo &T, JobRec (&r)[kItems], uint64_t (&c)[kItems]) {
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++) {
            const uint32_t i = k * TT + tid;
            r[k] = JobRec{0, 0, 0};
            c[k] = 0;
            if (i < T.n) {
                const uint64_t j = (uint64_t)T.first + i;
                r[k] = a.jrec[j];
                if (j > 0) c[k] = a.unit_pos[(j - 1) / kPreUnit] + a.cpos[j - 1];
            }
        }
    };
    fetch_records(D, jr, ce);
    while (D.n) {
        // thread 0: the next tile's geometry and the ticket of the one after it -- issued now, consumed at the publish
        // below (nothing here waits for them)
        uint4 d1_lo = make_uint4(0, 0, 0, 0), d1_hi = d1_lo; // (a TileInfo as two 16-byte halves: stays in registers)
        unsigned long long ticket = 0;
        if (tid == 0 && t_next != 0xffffffffu) {
            const uint4 *tp = reinterpret_cast<const uint4 *>(a.tiles + t_next);
            d1_lo = tp[0];
            d1_hi = tp[1];
            if (!(a.debug & 8u)) ticket = atomicAdd(&a.cnt[kCntHeads + 16 * head], 1ull);
        }
        const uint32_t first = D.first, n = D.n;
        const bool bad = n > kMaxJobs || D.image > lds_floats; // cannot happen with the cost floor and the bracket rule
        if (bad && tid == 0) atomicMin(&a.cnt[kCntOverflow], (unsigned long long)first);
        const uint32_t tag = seq++;
        // ---- 1. records, chunk marks, histogram: every job's place in the image from its own running sums ----
        uint32_t bin[kItems], rank[kItems];
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++) { bin[k] = 0xffffffffu; rank[k] = 0; }
        {
#pragma unroll
            for (uint32_t k = 0; k < kItems; k++) {
                if (k * TT >= n) break; // (uniform: no job of this round exists)
                if ((jr[k].meta & kMetaTile) && !bad) {
                    const uint32_t c_read = (uint32_t)ce[k], c_ref = (uint32_t)(ce[k] >> 32);
                    const uint32_t i = k * TT + tid;
                    const uint32_t N = jr[k].meta & 127u, M = (jr[k].meta >> 7) & 127u, R = (jr[k].meta >> 14) & 3u;
                    const bool swap = (jr[k].meta >> 17) & 1u, starts = (jr[k].meta >> 18) & 1u;
                    const uint32_t p_read = image_pos(c_read, jr[k].read_off, starts) - D.base_read;
                    const uint32_t p_ref = image_pos(c_ref, jr[k].ref_off, starts) - D.base_ref + D.ref_region;
                    rec[i] = make_uint2((swap ? p_ref : p_read) | ((swap ? p_read : p_ref) << 16),
                                        N | (M << 7) | (R << 14) | (((jr[k].meta >> 16) & 1u) << 16));
                    bin[k] = (R <= 2u ? 80u : 0u) + (79u - min(N, 79u));
                    rank[k] = atomicAdd(&sc.hist[bin[k]], 1u);
                    // Marks: every 16-byte chunk the two windows touch gets (arena index - image offset) and the tile's tag:
                    // the same value for all chunks of a run, and two runs never share a chunk.  A run start also owns the
                    // (at most one) chunk of slack between the previous run's last chunk and its own first one.
                    if (!(a.debug & 16u)) {
#pragma unroll
                        for (uint32_t w = 0; w < 2; w++) {
                            const uint32_t P = w ? p_ref : p_read, L = w ? (swap ? N : M) : (swap ? M : N);
                            const long long delta = (w ? (long long)jr[k].ref_off : (long long)jr[k].read_off) - (long long)P;
                            const uint4 mark = make_uint4((uint32_t)delta, (uint32_t)((unsigned long long)delta >> 32), 0x5eed0000u ^ tag, tag);
                            // (a window of up to nine floats -- most of a sparse batch's -- lies in three chunks: those
                            // without a loop, whose trip count would be the wave's longest window's)
                            const uint32_t c0 = P & ~3u, ce_ = P + L;
                            *reinterpret_cast<uint4 *>(win + c0) = mark;
                            if (c0 + 4u < ce_) *reinterpret_cast<uint4 *>(win + c0 + 4u) = mark;
                            if (c0 + 8u < ce_) *reinterpret_cast<uint4 *>(win + c0 + 8u) = mark;
                            for (uint32_t c = c0 + 12u; c < ce_; c += 4u) *reinterpret_cast<uint4 *>(win + c) = mark;
                            if (starts && i != D.first_tile) {
                                const uint32_t gap = ((w ? c_ref - D.base_ref + D.ref_region : c_read - D.base_read)) & ~3u;
                                if (gap < (P & ~3u)) *reinterpret_cast<uint4 *>(win + gap) = make_uint4(0u, 0u, 0u, tag);
                            }
                        }
                    }
                }
            }
        }
        if (tid == 0) { // publish the next tile; resolve the ticket into the tile after it
            uint4 *sp = reinterpret_cast<uint4 *>(&s_tile[slot ^ 1u]);
            sp[0] = d1_lo;
            sp[1] = d1_hi;
            uint32_t t2 = 0xffffffffu;
            if (t_next != 0xffffffffu) {
                if (a.debug & 8u) t2 = next_tile(a, head, n_tiles);
                else {
                    const unsigned long long t = ticket * 8ull + head;
                    t2 = t < n_tiles ? (uint32_t)t : next_tile(a, head, n_tiles); // (this head is dry: try the others)
                }
            }
            t_next = t2;
            s_seq = 0; // the DP's chunk counter
        }
        __syncthreads(); // marks, records, histogram complete; next tile's geometry visible
        // the next tile's records: on their way while this tile is staged, sorted and computed
        const TileInfo Dn = s_tile[slot ^ 1u];
        JobRec jr_n[kItems];
        uint64_t ce_n[kItems];
        fetch_records(Dn, jr_n, ce_n);
        // ---- 2. stage: a flat copy of the marked chunks, 16 bytes per lane, consecutive lanes consecutive chunks; the
        // histogram's scan is the first wave's, ahead of its share of the copy ----
        const uint32_t chunks = bad ? 0u : D.image >> 2;
        {
            __builtin_amdgcn_s_setprio(2); // a fresh tile's loads must not queue behind the DP of the older workgroups
            if (wv == 0) {
                // the sort's scan, by the first wave alone (the others are already staging): three bins per lane, the bins'
                // first places back into the table, the tile's job count next to it
                const uint32_t h0 = sc.hist[3 * lane], h1 = sc.hist[3 * lane + 1], h2 = sc.hist[3 * lane + 2];
                const uint32_t sum = h0 + h1 + h2;
                uint32_t incl = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
                    if (lane >= d) incl += o;
                }
                const uint32_t ex = incl - sum;
                sc.hist[3 * lane] = ex; sc.hist[3 * lane + 1] = ex + h0; sc.hist[3 * lane + 2] = ex + h0 + h1;
                if (lane == 63) s_tmp[0] = incl;
            }
            for (uint32_t q0 = 0; q0 < chunks || q0 == 0; q0 += 4 * TT) { // four chunks per thread and round
                uint32_t ok[4];
                const float *src[4];
                float4 v[4];
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t q = q0 + tid + u * TT;
                    ok[u] = 0; src[u] = a.ev;
                    if (q < chunks && !(a.debug & 2u)) {
                        const uint4 hd = reinterpret_cast<const uint4 *>(win)[q];
                        const long long delta = (long long)((unsigned long long)hd.x | ((unsigned long long)hd.y << 32));
                        ok[u] = (hd.z == (0x5eed0000u ^ tag) && hd.w == tag) ? 1u : 0u;
                        src[u] = ((4u * q < D.ref_region) ? a.ev : a.ref) + ((long long)(4u * q) + delta);
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4 *>(src[u]); // (idle lanes read ev[0..3])
#pragma unroll
                for (uint32_t u = 0; u < 4; u++) {
                    const uint32_t q = q0 + tid + u * TT;
                    if (ok[u]) reinterpret_cast<float4 *>(win)[q] = v[u];
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads(); // image staged; bin starts and job count written
        const uint32_t n_tile_jobs = s_tmp[0];
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++)
            if (bin[k] != 0xffffffffu) sc.perm[sc.hist[bin[k]] + rank[k]] = (uint16_t)(k * TT + tid);
        __syncthreads(); // permutation complete
        for (uint32_t b = tid; b < kSortBins; b += TT) sc.hist[b] = 0; // (for the next tile; nobody reads it any more)
        // ---- 3. the DP: one lane per job; waves pull 64 sorted jobs at a time (the heavy class first), so the waves of the
        // workgroup finish together whatever the mix ----
        while (!(a.debug & 1u)) {
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(&s_seq, 1u);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (c * 64u >= n_tile_jobs) break;
            const uint32_t r = c * 64u + lane;
            const bool act = r < n_tile_jobs;
            const uint32_t i = min((uint32_t)sc.perm[act ? r : n_tile_jobs - 1], n - 1u); // (never past the tile's range, whatever the table holds)
            const uint2 rc = rec[i];
            const uint32_t N = rc.y & 127u, M = (rc.y >> 7) & 127u, R = (rc.y >> 14) & 3u;
            const float res = stream_lane_job(win + (rc.x & 0xffffu), win + (rc.x >> 16), N, M, R, (rc.y >> 16) & 1u, act);
            if (act) rec[i].x = __float_as_uint(res); // (the job's record is done with; its slot carries the result out)
        }
        __syncthreads();
        // results out in job order, 256 bytes a wave: the sorted order would scatter every wave's 64 results over the tile's
        // range (one memory write request per job).  Thread t reads back the slots of the jobs t, t + TT, ... -- the same
        // slots it fills with the next tile's records afterwards, so no barrier stands between the two.
        if (!bad && !(a.debug & 1u)) {
#pragma unroll
            for (uint32_t k = 0; k < kItems; k++) {
                const uint32_t i = k * TT + tid;
                if (i < n && (jr[k].meta & kMetaTile)) a.out[first + i] = __uint_as_float(rec[i].x);
            }
        }
        slot ^= 1u;
        D = Dn;
#pragma unroll
        for (uint32_t k = 0; k < kItems; k++) { jr[k] = jr_n[k]; ce[k] = ce_n[k]; }
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
inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kT - 1) / kT); }
} // namespace

static uint32_t stream_lds_bytes_t(uint32_t lds_floats, int threads)
{
    return lds_floats * 4u + kItems * (uint32_t)threads * 8u + (uint32_t)(threads == 512 ? sizeof(SortTable<512>) : sizeof(SortTable<256>));
}

// everything rawdtw_batch_create enqueues for a sparse + banded batch: planning of the DTW launch and the chain records
hipError_t stream_plan(const StreamArgs &a, ChainDesc *d_chains, uint32_t *d_fold_order, hipStream_t s)
{
    (void)hipGetLastError();
    if (a.n_jobs) {
        hipLaunchKernelGGL(k_pre, dim3((uint32_t)((a.n_jobs + kPreUnit - 1) / kPreUnit)), dim3(kPreT), 0, s, a);
    }
    if (a.n_jobs || a.n_chains)
        hipLaunchKernelGGL(k_mid, dim3(2 + kMidOthers + 1 + (uint32_t)((a.n_chains + kMidT - 1) / kMidT)), dim3(kMidT), 0, s, a, d_chains, d_fold_order);
    if (a.n_jobs) hipLaunchKernelGGL(k_tile_first, dim3((uint32_t)(((uint64_t)a.tiles_cap + 1 + 63) / 64)), dim3(64), 0, s, a);
    return hipGetLastError();
}

// reset_queue: the batch has run before (the tile queue's heads start at zero with the counters' initial values)
hipError_t stream_run(const StreamArgs &a, uint32_t blocks, uint32_t lds_floats, int threads, bool reset_queue, hipStream_t s)
{
    if (a.n_jobs == 0) return hipSuccess;
    (void)hipGetLastError();
    hipError_t e = reset_queue ? hipMemsetAsync(&a.cnt[kCntHeads], 0, 8 * 16 * sizeof(unsigned long long), s) : hipSuccess;
    if (e != hipSuccess) return e;
    const uint32_t lds_bytes = stream_lds_bytes_t(lds_floats, threads);
    const void *fn = threads == 512 ? reinterpret_cast<const void *>(k_stream<512>) : reinterpret_cast<const void *>(k_stream<256>);
    if (lds_bytes > 64 * 1024) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (threads == 512) hipLaunchKernelGGL(k_stream<512>, dim3(blocks), dim3(512), lds_bytes, s, a, lds_floats);
    else hipLaunchKernelGGL(k_stream<256>, dim3(blocks), dim3(256), lds_bytes, s, a, lds_floats);
    return hipGetLastError();
}

// workgroups of k_stream one compute unit holds at this LDS size (for the persistent grid)
int stream_blocks_per_cu(uint32_t lds_floats, int threads)
{
    int n = 0;
    const uint32_t lds_bytes = stream_lds_bytes_t(lds_floats, threads);
    const void *fn = threads == 512 ? reinterpret_cast<const void *>(k_stream<512>) : reinterpret_cast<const void *>(k_stream<256>);
    if (lds_bytes > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds_bytes) != hipSuccess) return 0;
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
