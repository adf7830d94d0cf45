// rawdtw_runs.hip -- the sync-free candidate-batch pipeline (rawdtw_batch_create / submit / run for sparse + banded batches).
//
// What it replaces: the DTW block of gen_chains (src/rmap.cpp:509-530) for every read of a mini-batch, i.e. all the calls
// of DTW_global_slantedbanded_antidiagonalwise (src/dtw.cpp:273-520) that align_chain (src/rmap.cpp:238-300) issues, the
// fold of their costs (rmap.cpp:279-280, 306) and the per-read accept / cut loop (rmap.cpp:515-524).
//
// The unit is a RUN: a chain's consecutive parts are ONE contiguous piece of each arena (part p runs from anchors[parts - p]
// to anchors[parts - p - 1], rmap.cpp:248-293: it starts on the element the part before it ends on), and the anchors'
// positions ARE the running sums of the window lengths.  A batch is these launches on the context's stream, none of which
// the host waits for (rawdtw_batch.cpp: batch_create_stream, batch_enqueue_one):
//
//   k_scan         one workgroup per 8192 anchors: finds the range's first chain (a 64-way search of the chains' offsets by one
//                  wave), checks every part's anchors, classifies it (radius rmap.cpp:276 + slant dtw.cpp:298-300), appends the
//                  one part in two hundred the lane-per-job bodies do not take (radius > 3, longer side > 73) to a side list and
//                  writes the tile list (tile, its first chain) of every tile that has a part for the lane bodies; its other
//                  workgroups write the fold's chain records.  k_scan_compact: the same with the anchor lists in the compact
//                  hand-over form, decoded unit by unit first (k_scan_desc then writes the chain records).
//   k_side         the side list into class order (wave-per-job first, longest first)
//   k_wide         the side list's jobs: wave-cooperative bodies (rawdtw_dp.h: wband_gen, grp_wave, lane_dp_k8), dealt to the
//                  workgroups like a snake, pulled by a workgroup's waves from an LDS counter
//   k_plan         a wave a listed tile (512 consecutive anchors): lays the tile's LDS image out (runs: one contiguous piece of
//                  each arena a run), sorts its jobs into the order the lanes take them, and leaves per PASS the job records
//                  (8 bytes a job), the copy orders (16 bytes a run and arena) and the list entry in memory
//   k_runs         a persistent grid over the passes: stage the pass's image by LDS-DMA, one lane per job (four lanes for the
//                  radius-3 jobs), the cost of part i to out[i]; two barriers a pass
//   k_gather       chunk rounds only (rawdtw_batch_submit_carry): the lists above are then the round's SHORT lists (new entries +
//                  one junction a chain); this launch lays every chain's costs out in full -- the stretch taken over from the
//                  previous batch's cost array, then the new parts' -- for the fold
//   k_fold_select  fold (rmap.cpp:279-280, 306) and accept / cut loop (rmap.cpp:515-524) out of LDS, a wave per 8 reads
//
// No step needs a number on the host: grids are sized by the anchor count or are persistent, every count lives in a
// device counter block.  rawdtw_batch_create only enqueues; errors and the rare shapes this path does not take (band wider
// than 256 slots, chains without anchors, a list over a capacity) surface in the counters, which rawdtw_batch_fetch reads
// together with the results and which make the later launches leave at once: such a batch is redone through the job list.

#include "rawdtw_dp.h"

namespace rawdtw {

namespace {

constexpr int kT = 256;

__device__ __forceinline__ int d_slanted_radius(uint32_t n, uint32_t m, int r0)
{
    const uint32_t N = n > m ? n : m, M = n > m ? m : n;
    const uint32_t x = (N - M) * (uint32_t)r0 + N - 1u; // dtw.cpp:298-300, unsigned 32-bit: extra = x / N
    uint32_t q;
    if (N < (1u << 11) && (uint32_t)r0 < (1u << 11)) {
        // x < 2^23 is exact in a float and the quotient is below 2^12: the product with the hardware reciprocal (1 ulp)
        // is within 2^-10 of it, so the truncation is the quotient or one beside it -- one remainder check settles it
        // (an integer division is some 25 instructions)
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

// One candidate part: the DTW job between two consecutive anchors of a chain (rmap.cpp:251-254, 270, 276), its shape and
// class.  `s` = the anchor the part starts on, `e` = the one it ends on (s lies one entry BEHIND e in the end-first list).
struct Part {
    uint32_t n, m;   // read events, reference signals in the window (rmap.cpp:255-262)
    int R;           // band radius after the slant correction (dtw.cpp:298-300)
    bool asc;        // the anchors ascend (a chain the mapper could produce)
    bool tile;       // the lane-per-job bodies take it
};

__device__ __forceinline__ Part classify(const StreamArgs &a, const rawdtw_anchor_t s, const rawdtw_anchor_t e)
{
    Part p;
    p.asc = e.target_position >= s.target_position && e.query_position >= s.query_position;
    p.m = e.target_position - s.target_position + 1;
    p.n = e.query_position - s.query_position + 1;
    int r0 = (int)((float)p.n * a.frac); // rmap.cpp:276, fp32 product
    r0 = r0 > 1 ? r0 : 1;
    // Tile class without the division of dtw.cpp:298-300: with d = N - M the radius is R = r0 + ceil(d r0 / N), so
    //     R <= Rm  <=>  r0 <= Rm and d r0 <= (Rm - r0) N,
    // and for a tile part (Rm <= 3 and r0 >= 1: the ceiling is 0, 1 or 2)  R = r0 + (d r0 > 0) + (d r0 > N).
    // (32-bit products: N <= lane_max_n < 128 and r0 <= 3 wherever the result counts.)
    const uint32_t N = p.n > p.m ? p.n : p.m, dr = (N - (p.n > p.m ? p.m : p.n)) * (uint32_t)r0;
    p.R = r0 + (dr > 0u ? 1 : 0) + (dr > N ? 1 : 0);
    p.tile = p.asc && p.n < 0x7fffffffu && p.m < 0x7fffffffu && N <= a.lane_max_n && r0 <= a.lane_max_radius &&
             dr <= (uint32_t)(a.lane_max_radius - r0) * N;
    return p;
}

// The chain that owns anchor x: the LAST chain c with anchor_off[c] <= x (chains without anchors share their successor's
// offset).  A 64-way search by one wave: three rounds for 2^18 chains.  All lanes return the chain.
__device__ __forceinline__ uint64_t find_chain(const uint64_t *__restrict__ anchor_off, uint64_t n_chains, uint64_t x, int lane)
{
    uint64_t lo = 0, hi = n_chains; // invariant: anchor_off[lo] <= x, answer in [lo, hi)
    while (hi - lo > 1) {
        const uint64_t step = (hi - lo + 63) / 64;
        const uint64_t p = lo + (uint64_t)lane * step;
        const bool le = p < hi && anchor_off[p] <= x;
        const int cnt = __popcll(__ballot(le)); // the probes are ascending: the predicate is true on a prefix (lane 0 always)
        const uint64_t nlo = lo + (uint64_t)(cnt - 1) * step;
        hi = min(hi, nlo + step);
        lo = nlo;
    }
    return lo;
}

// Lanes of one wave that hand values to each other through LDS: the hardware runs a wave's LDS instructions in order, but
// the compiler orders a thread's loads and stores by what THAT thread can observe (it may read a word before another
// lane's store to it is issued, and forward a lane's own store) -- a fence at wave scope on both sides of the hand-over.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive scan of one value per lane over the wave by DPP: shifts inside the rows of 16 lanes, then the rows' last lanes
// broadcast into the rows behind them (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  Six dependent
// VALU steps of a few cycles each; through __shfl_up the same scan is six round trips through the LDS crossbar
// (ds_bpermute, ~70 cycles a step), and a tile has four such scans on its critical path.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return x;
}

// Decoding a compact anchor list (include/rawdtw.h) is a scan of maps: an entry sent whole is the constant map x -> v, an
// entry sent as a step back is x -> x - step, and entry i's anchor is the composition of the maps of its unit's entries
// 0 .. i applied to anything (the unit's first entry is sent whole).  `set` = constant map.
struct StepMap { uint32_t set, q, t; };
__device__ __forceinline__ StepMap compose(const StepMap later, const StepMap earlier)
{
    if (later.set) return later;
    return StepMap{earlier.set, earlier.set ? earlier.q - later.q : earlier.q + later.q, earlier.set ? earlier.t - later.t : earlier.t + later.t};
}
#define RAWDTW_DPP_MAP(m, ctrl, rowmask)                                                                                      \
    StepMap{(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(m).set, ctrl, rowmask, 0xf, false),                                  \
            (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(m).q, ctrl, rowmask, 0xf, false),                                    \
            (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(m).t, ctrl, rowmask, 0xf, false)}
// inclusive scan of the lanes' maps over the wave (see wave_scan_incl; a lane without a source takes the identity map)
__device__ __forceinline__ StepMap wave_scan_maps(StepMap m)
{
    m = compose(m, RAWDTW_DPP_MAP(m, 0x111, 0xf));
    m = compose(m, RAWDTW_DPP_MAP(m, 0x112, 0xf));
    m = compose(m, RAWDTW_DPP_MAP(m, 0x114, 0xf));
    m = compose(m, RAWDTW_DPP_MAP(m, 0x118, 0xf));
    m = compose(m, RAWDTW_DPP_MAP(m, 0x142, 0xa));
    m = compose(m, RAWDTW_DPP_MAP(m, 0x143, 0xc));
    return m;
}

// bit p of a tile's chain-start mask: anchor (tile base + p) is the first entry of a chain (bit AT may be set as well:
// the anchor behind the tile's last; the end of the anchor list counts as a chain start)
__device__ __forceinline__ bool mask_bit(const uint32_t *mask, uint32_t p) { return (mask[p >> 5] >> (p & 31u)) & 1u; }

// Marks the chain starts inside [base, base + AT] in `mask` (zeroed by the caller, LDS) from the chains' offsets,
// starting at chain c0 = the chain that owns anchor `base`.  Chains are few per tile (a handful in a sparse batch): one
// round of loads; tiles of very short chains take more.  Ends with the workgroup synchronised and the mask complete.
// Returns false when a chain without anchors was seen (the popcount-based chain lookup would be off: the batch is redone
// through the job list).
template <int NT>
__device__ __forceinline__ bool mark_chain_starts(const StreamArgs &a, uint64_t c0, uint64_t base, uint32_t at, uint32_t *mask)
{
    bool ok = true;
    for (uint64_t c = c0 + threadIdx.x;; c += NT) {
        uint64_t s = ~0ull;
        if (c <= a.n_chains) {
            s = a.anchor_off[c];
            if (s >= base && s <= base + at) atomicOr(&mask[(uint32_t)(s - base) >> 5], 1u << ((uint32_t)(s - base) & 31u));
            if (c < a.n_chains && s < base + at && a.anchor_off[c + 1] == s) ok = false; // (c0 itself owns an anchor)
        }
        // the round's last thread tells whether chains starting inside the tile may be left
        if (!__syncthreads_or(threadIdx.x == NT - 1 && c < a.n_chains && s < base + at)) break;
    }
    return ok;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// fold support: chain records (ChainDesc) and the fold order
// ---------------------------------------------------------------------------------------------------------------------
// Also the one bounds check a batch needs: a chain's parts tile the span between its first and last anchor, so when the
// anchors ascend (checked part by part in k_scan) every window lies inside the arenas iff the chain's span does.
__device__ __forceinline__ void chain_desc_body(const StreamArgs &a, ChainDesc *__restrict__ chains, const uint64_t c)
{
    if (c >= a.n_chains) return;
    const uint64_t a0 = a.anchor_off[c], a1 = a.anchor_off[c + 1];
    ChainDesc d;
    d.job_first = a1 >= 2 ? a1 - 2 : 0;                      // part p = out[a1 - 2 - p]
    d.n_jobs = a1 > a0 ? (uint32_t)(a1 - a0 - 1) : 0u;
    d.descending = 1; d.span = 0; d.num_aligned = 0;
    if (a1 > a0) {
        const rawdtw_anchor_t first = a.anchors[a1 - 1], last = a.anchors[a0];
        d.span = last.query_position - first.query_position + 1;                     // rmap.cpp:245
        d.num_aligned = (last.query_position - first.query_position) + d.n_jobs;       // sum of the parts' read regions (rmap.cpp:292)
        if (d.n_jobs && ((uint64_t)a.read_base[c] + last.query_position + 1ull > a.n_ev || a.ref_base[c] + last.target_position + 1ull > a.n_ref))
            atomicMin(&a.cnt[kCntBad], (unsigned long long)a0);
    } else atomicAdd(&a.cnt[kCntUnsupported], 1ull); // a chain without anchors: align_chain would read anchors[-1]
    chains[c] = d;
}

// Fold order: chains by part count, longest first (the lane-per-chain fold gives a wave 64 chains of similar length).
// One workgroup: a counting sort over 1024 length buckets in LDS (the order inside a bucket does not matter).
template <int NT>
__device__ __forceinline__ void fold_order_body(const uint64_t n_chains, const uint64_t *__restrict__ anchor_off, uint32_t *__restrict__ order)
{
    constexpr uint32_t kBins = 1024, kB = kBins / NT; // consecutive buckets per thread in the scan
    constexpr uint32_t kC = 4;                        // chains per thread and round: their offsets are all requested before any is used
    __shared__ uint32_t hist[kBins];
    __shared__ uint32_t wsum[NT / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    auto bucket = [](uint64_t anchors) { return 1023u - (uint32_t)min<uint64_t>(anchors ? anchors - 1 : 0, 1023ull); }; // a chain's part count, clamped
    auto keys = [&](uint64_t c0, uint32_t (&b)[kC]) { // buckets of chains c0 + q * NT
        uint64_t lo[kC], hi[kC];
#pragma unroll
        for (uint32_t q = 0; q < kC; q++) {
            const uint64_t c = min(c0 + (uint64_t)q * NT, n_chains - 1);
            lo[q] = anchor_off[c]; hi[q] = anchor_off[c + 1];
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
// k_scan: the pass over the anchor list ahead of the DTW launch.  Roles by workgroup: [0, n_tiles) one tile each,
// n_tiles the fold order, the rest the chain records.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kSortBins = 192; // bin = (3 - radius) * 64 + (63 - longer side): radius 3 first, then 2, then 1 (waves of one radius take the
                                     // shortest body), each run longest first (sides of 63 and more share a bin)

// A tile's items in the order the image is laid out in: item u is the part that ends at anchor (tile end - 1 - u), so that
// along a chain (stored end-first) u ascends with the positions -- a run's first part is the one with the lowest addresses,
// and the scan that places the runs meets it first.
//
// Layout of a tile's LDS image.  The image has an event region and a reference region; consecutive tile-class parts of a
// chain (a "run") share their anchor elements, so a run is ONE contiguous piece of each arena and of each region.  Every
// part adds floats to the two regions' running sums: a run's first part its whole window plus 3 floats of slack, a
// continuing part its window minus the shared first element, a run's last part 3 more (the run's END rounded up to a
// 16-byte boundary never reaches the next run).  With c = the running sum BEFORE a run's first part the run starts at
//     c + ((off - c) & 3)                 (off = the window's arena offset),
// congruent to the arena offset modulo 4 -- 16-byte chunks of the image are 16-byte chunks of the arena -- and a part's
// window starts at (its start anchor's position + D), D = the run's image start minus its first position.  So a tile is
// staged by copying each run's chunk range, fully coalesced, and two runs never share a chunk.
struct RunTab {
    uint32_t lo[2][kStreamMaxSeg];    // first float of the run's first 16-byte chunk in the image [arena: 0 events, 1 reference]
    uint32_t end[2][kStreamMaxSeg];   // the run's last position + 1 (query / target coordinates)
    int32_t D[2][kStreamMaxSeg];      // image index = position + D
    long long src[2][kStreamMaxSeg];  // arena index = image index + src (a multiple of 4)
};

// A scan unit = kScanUnit consecutive anchors = a whole number of tiles, one workgroup, eight consecutive anchors a thread
// and pass (forward order: the part of anchor i runs from anchors[i + 1] to anchors[i]).  It writes the first
// chain of each of its tiles (4 bytes a tile), three statistics, and -- for the one part in two hundred the tiles do not
// take -- a side-list record.  Units are large because every unit ends with one returning atomic per side-list class on
// a counter the whole grid shares: such a word takes ~88 atomics a microsecond, and at one workgroup per TILE (ten
// thousand a batch) the launch spent 150 of its 176 us queueing there.
constexpr uint32_t kScanUnit = 8192, kScanKI = 8;
// Workgroups of kScanT threads take a unit in two halves (eight consecutive anchors a thread and half): three such
// workgroups fit a compute unit -- all of a bench batch's 638 units are resident at once, where one 1024-thread workgroup
// a compute unit took them in three rounds of the same latency chain -- and one fits beside the DTW launch's workgroups.
// The compact form (kScanTC threads, one pass) decodes a unit with one scan over all of its entries.
constexpr uint32_t kScanT = 512, kScanTC = 1024;
static_assert(kScanUnit == RAWDTW_COMPACT_STRIDE, "the compact hand-over is decoded unit by unit");
template <bool COMPACT>
__device__ __forceinline__ void scan_unit_body(const StreamArgs &a, const uint32_t unit)
{
    constexpr int NT = COMPACT ? (int)kScanTC : (int)kScanT, KI = (int)kScanKI;
    constexpr uint32_t AT = kScanUnit, kWords = AT / 32 + 1, kHalf = (uint32_t)NT * KI, kHalves = AT / kHalf;
    static_assert(kHalves * kHalf == AT && (!COMPACT || kHalves == 1), "a unit in whole passes; the compact form in one");
    __shared__ uint32_t s_mask[kWords], s_pre[kWords];
    __shared__ uint32_t s_ocnt, s_obase, s_cls[kStreamClasses];
    __shared__ unsigned long long s_stats[3];
    __shared__ uint64_t s_c0;
    __shared__ uint32_t s_todo; // bit t: tile t of the unit has a part for the lane-per-job bodies to score
    constexpr uint32_t kOList = 2048;
    __shared__ uint32_t s_olist[kOList]; // the unit's side-list parts by slot: position | class << 13 | radius << 18
    const int tid = threadIdx.x, lane = tid & 63;
    const uint64_t base = (uint64_t)unit * AT;
    for (uint32_t w = tid; w < kWords; w += NT) s_mask[w] = 0;
    if (tid == 0) { s_ocnt = 0; s_todo = 0; }
    if (tid < 3) s_stats[tid] = 0;
    if (tid < (int)kStreamClasses) s_cls[tid] = 0;
    // the anchors: KI + 1 consecutive entries a thread (the last one is the next thread's first: the start of this thread's
    // last part), requested before the chain search waits for anything
    const uint64_t i0 = base + (uint64_t)tid * KI; // the thread's first anchor (of the first half)
    rawdtw_anchor_t an[KI + 1];
    uint4 steps8 = make_uint4(0u, 0u, 0u, 0u); // compact form: this thread's eight 2-byte steps
    if (COMPACT) { if (i0 < a.n_anchors) steps8 = *reinterpret_cast<const uint4 *>(a.steps + i0); } // (the array is padded to whole units)
    else {
#pragma unroll
        for (int k = 0; k <= KI; k++) an[k] = i0 + k < a.n_anchors ? a.anchors[i0 + k] : rawdtw_anchor_t{0, 0};
    }
    const rawdtw_anchor_t *const anchors_rd = COMPACT ? a.anchors_w : a.anchors; // (the decoded list, once this unit has written it)
    if (tid < 64) {
        const uint64_t c = find_chain(a.anchor_off, a.n_chains, base, lane);
        if (tid == 0) s_c0 = c;
    }
    __syncthreads();
    const uint64_t c0 = s_c0;
    if (!mark_chain_starts<NT>(a, c0, base, AT, s_mask)) atomicAdd(&a.cnt[kCntUnsupported], 1ull);
    {   // chain starts before each word of the mask: an exclusive scan over the mask's words, one word a thread
        static_assert(kWords <= (uint32_t)NT, "one mask word a thread");
        __shared__ uint32_t s_wsum[NT / 64];
        const uint32_t pc = (uint32_t)tid < kWords ? __popc(s_mask[tid]) : 0u;
        uint32_t incl = pc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t pre = 0;
        for (int w = 0; w < (tid >> 6); w++) pre += s_wsum[w];
        if ((uint32_t)tid < kWords) s_pre[tid] = pre + incl - pc;
    }
    __syncthreads();
    // the chain of the anchor at position p of the unit: c0 + the chain starts in positions 1 .. p
    auto chain_at = [&](uint32_t p) {
        return c0 + (s_pre[p >> 5] + __popc(s_mask[p >> 5] & (0xffffffffu >> (31u - (p & 31u)))) - (s_mask[0] & 1u));
    };
    if (COMPACT) {
        // ---- decode the unit: every entry's map, the thread's eight composed in order, a scan over the workgroup's threads ----
        static_assert(KI == 8, "eight 2-byte steps a thread");
        __shared__ StepMap s_wagg[NT / 64];
        __shared__ uint2 s_first[NT];
        const uint32_t sv[8] = {steps8.x & 0xffffu, steps8.x >> 16, steps8.y & 0xffffu, steps8.y >> 16,
                                steps8.z & 0xffffu, steps8.z >> 16, steps8.w & 0xffffu, steps8.w >> 16};
        StepMap part[KI]; // the composition of this thread's entries 0 .. k
        StepMap run{0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < KI; k++) {
            const uint32_t p = (uint32_t)tid * KI + k;
            const uint64_t i = base + p;
            StepMap m{0u, 0u, 0u};
            if (i < a.n_anchors) {
                if (mask_bit(s_mask, p)) { const rawdtw_anchor_t h = a.heads[chain_at(p)]; m = StepMap{1u, h.query_position, h.target_position}; }
                else if (p == 0) { const rawdtw_anchor_t h = a.unit_abs[unit]; m = StepMap{1u, h.query_position, h.target_position}; }
                else if (sv[k] == 0xffffu) { // an escape: the entry's steps are in the list of wide steps, ascending by index
                    uint64_t lo = 0, hi = a.n_wide;
                    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (a.wide[mid].index < (uint32_t)i) lo = mid + 1; else hi = mid; }
                    if (lo < a.n_wide && a.wide[lo].index == (uint32_t)i) m = StepMap{0u, a.wide[lo].query_step, a.wide[lo].target_step};
                    else atomicMin(&a.cnt[kCntBad], (unsigned long long)i); // (a malformed hand-over)
                } else m = StepMap{0u, sv[k] & 0xffu, sv[k] >> 8};
            }
            run = compose(m, run);
            part[k] = run;
        }
        const StepMap incl = wave_scan_maps(run);
        if (lane == 63) s_wagg[tid >> 6] = incl;
        __syncthreads();
        StepMap pre = RAWDTW_DPP_MAP(incl, 0x138, 0xf); // the lanes before this one (lane 0: the identity)
        {
            StepMap wpre{0u, 0u, 0u}; // the waves before this one
            for (int w = 0; w < (tid >> 6); w++) wpre = compose(s_wagg[w], wpre);
            pre = compose(pre, wpre);
        }
#pragma unroll
        for (int k = 0; k < KI; k++) {
            const StepMap f = compose(part[k], pre); // (set: the unit's first entry is)
            an[k] = rawdtw_anchor_t{f.t, f.q};
            if (i0 + k < a.n_anchors) a.anchors_w[i0 + k] = an[k];
        }
        s_first[tid] = make_uint2(an[0].target_position, an[0].query_position);
        __syncthreads();
        if (tid + 1 < NT) { const uint2 x = s_first[tid + 1]; an[KI] = rawdtw_anchor_t{x.x, x.y}; }
        else an[KI] = base + AT < a.n_anchors ? a.unit_abs[unit + 1] : rawdtw_anchor_t{0, 0}; // (the next unit's first entry travels whole)
    }

    uint32_t my_tiles = 0, my_bytes = 0;
    unsigned long long my_obytes = 0;
    uint32_t o_rec[kHalves][KI]; // the thread's side-list parts: slot | class << 16 | radius << 21 (0xffffffff: none)
#pragma unroll
    for (uint32_t h = 0; h < kHalves; h++) {
    const uint64_t ih = i0 + (uint64_t)h * kHalf;
    if (h) { // (the half's anchors: KI + 1 consecutive entries a thread)
#pragma unroll
        for (int k = 0; k <= KI; k++) an[k] = ih + k < a.n_anchors ? a.anchors[ih + k] : rawdtw_anchor_t{0, 0};
    }
    uint32_t half_tiles = 0;
#pragma unroll
    for (int k = 0; k < KI; k++) {
        o_rec[h][k] = 0xffffffffu;
        const uint32_t p = h * kHalf + (uint32_t)tid * KI + k; // position in the unit
        const uint64_t i = base + p;
        if (i >= a.n_anchors || mask_bit(s_mask, p + 1)) continue; // the chain's first entry (or the list's end): no part ends here
        const Part pt = classify(a, an[k + 1], an[k]);
        if (!pt.asc || pt.n >= 0x7fffffffu || pt.m >= 0x7fffffffu) { atomicMin(&a.cnt[kCntBad], (unsigned long long)i); continue; }
        if (pt.tile) { my_tiles++; half_tiles++; my_bytes += 4u * (pt.n + pt.m) + 36u; continue; }
        // the side list: rare.  The radius by the reference's formula, the class, a slot in the workgroup's share of the list
        int r0 = (int)((float)pt.n * a.frac);
        r0 = r0 > 1 ? r0 : 1;
        const int R = d_slanted_radius(pt.n, pt.m, r0);
        const uint32_t N = pt.n > pt.m ? pt.n : pt.m, K = (uint32_t)R + 1u;
        uint32_t cls;
        if (R >= 1 && R <= a.side_lane_radius && N <= a.lane_max_n) cls = kClsL0 + side_lane_bucket(N);
        else if (K <= 8) cls = kClsM0 + side_lane_bucket(N);
        else if (K <= 16) cls = kClsG16;
        else if (K <= 256) cls = N >= 1024 ? kClsW0 : N >= 256 ? kClsW0 + 1 : N >= 64 ? kClsW0 + 2 : kClsW0 + 3;
        else { atomicAdd(&a.cnt[kCntUnsupported], 1ull); continue; }
        my_obytes += 4ull * ((unsigned long long)pt.n + pt.m) + 36ull;
        {   // a slot in the unit's share of the side list; what the record is made of goes to the slot's word in LDS, where
            // the unit's threads make the records one each, side by side (below) -- or, beyond that list's length, stays with
            // the thread (position < 2^13, class < 2^5, R < 2^8)
            const uint32_t slot = atomicAdd(&s_ocnt, 1u);
            if (slot < kOList) s_olist[slot] = p | (cls << 13) | ((uint32_t)R << 18);
            else o_rec[h][k] = slot | (cls << 16) | ((uint32_t)R << 21);
        }
        atomicAdd(&s_cls[cls], 1u);
    }
    {
        const unsigned long long any_tile = __ballot(half_tiles != 0u); // (every lane votes: taken before the branch on the lane)
        if (lane == 0 && any_tile) atomicOr(&s_todo, 1u << ((h * kHalf + (uint32_t)tid * KI) / a.tile_anchors)); // (a wave's anchors lie in one tile)
    }
    }
    {   // statistics: tile parts (<= KI a thread) and their bytes (< 2^13 a thread) through one wave reduction
        unsigned long long packed = (unsigned long long)my_tiles | ((unsigned long long)my_bytes << 20);
        for (int off = 32; off > 0; off >>= 1) packed += __shfl_down(packed, off);
        if (lane == 0) { atomicAdd(&s_stats[0], packed & 0xfffffull); atomicAdd(&s_stats[1], packed >> 20); }
        if (my_obytes) atomicAdd(&s_stats[2], my_obytes);
    }
    __syncthreads();
    if (tid == 0 && s_ocnt) s_obase = (uint32_t)atomicAdd(&a.cnt[kCntOthers], (unsigned long long)s_ocnt);
    // The DTW launch's work list: the unit's tiles that have something to score, each with its first chain.  (Tiles without a
    // tile-class part -- all of their parts on the side list -- are never touched.)
    if (tid == 0 && s_todo) {
        const uint32_t tiles_per_unit = AT / a.tile_anchors, todo = s_todo;
        uint64_t at = atomicAdd(&a.cnt[kCntTodo], (unsigned long long)__popc(todo));
        for (uint32_t t = 0; t < tiles_per_unit; t++)
            if ((todo >> t) & 1u) a.tlist[at++] = make_uint2(unit * tiles_per_unit + t, (uint32_t)chain_at(t * a.tile_anchors));
    }
    if (tid < 3) a.tile_stats[3ull * unit + tid] = s_stats[tid];
    if (tid < (int)kStreamClasses && s_cls[tid]) atomicAdd(&a.cnt[kCntCls0 + tid], (unsigned long long)s_cls[tid]);
    __syncthreads();
    auto side_record = [&](const uint32_t p, const uint32_t cls, const uint32_t R, const uint64_t q) {
        if (q >= a.others_cap) return; // (beyond the capacity: kCntOthers > others_cap tells rawdtw_batch_fetch to take the job-list path)
        const uint64_t c = chain_at(p);
        // (rare: read again rather than kept; the compact form's next unit writes its first entry itself: that one travels whole)
        const rawdtw_anchor_t e = anchors_rd[base + p];
        const rawdtw_anchor_t s = (COMPACT && p + 1u == AT) ? a.unit_abs[unit + 1] : anchors_rd[base + p + 1];
        DevJob d;
        d.ref_off = a.ref_base[c] + s.target_position;
        d.read_off = a.read_base[c] + s.query_position;
        d.n = e.query_position - s.query_position + 1;
        d.m = e.target_position - s.target_position + 1;
        d.R = (int32_t)R;
        d.flags = mask_bit(s_mask, p) ? 0u : kFlagExcludeLast; // rmap.cpp:270: every part but the chain's last (= its first entry)
        d.aux = (uint32_t)(base + p);
        a.omix[q] = d; a.ocls[q] = (uint8_t)cls;
    };
    // the side list's records: one a thread out of the LDS list (a unit has some forty: one round of loads instead of one per
    // item of a thread that holds one) ...
    for (uint32_t t = (uint32_t)tid; t < min(s_ocnt, kOList); t += NT) {
        const uint32_t w = s_olist[t];
        side_record(w & 0x1fffu, (w >> 13) & 0x1fu, w >> 18, (uint64_t)s_obase + t);
    }
    if (s_ocnt > kOList) { // ... and those beyond the list's length from the threads that found them
        const uint64_t obase = s_obase;
#pragma unroll
        for (uint32_t h = 0; h < kHalves; h++)
#pragma unroll
            for (int k = 0; k < KI; k++) {
                if (o_rec[h][k] == 0xffffffffu) continue;
                side_record(h * kHalf + (uint32_t)tid * KI + k, (o_rec[h][k] >> 16) & 0x1fu, o_rec[h][k] >> 21, obase + (o_rec[h][k] & 0xffffu));
            }
    }
}

// Chunk rounds (rawdtw_batch_submit_carry): the lists the launches above and below work on are the round's SHORT lists -- per
// chain its new entries and the junction.  Two things remain to be said about the FULL chains:
//   chain_desc_carry  the fold's chain records: parts and span of the full chain (its start anchor comes with the carry record,
//                     its end anchor is the short list's first entry), the one bounds check a batch needs (chain_desc_body);
//   k_gather          every chain's costs in full, where the fold looks for them (out_full by the full lists' offsets): the new
//                     parts' from `out` (short-list order), the stretch taken over from the previous batch's full cost array --
//                     one contiguous copy a chain, a wave a chain.  A part that was its chain's last then and is not now loses
//                     its last cell's distance exactly as the DTW functions take it off (dtw.cpp:514-519): the stretch's first.
__device__ __forceinline__ void chain_desc_carry(const StreamArgs &a, ChainDesc *__restrict__ chains, const uint64_t c)
{
    if (c >= a.n_chains) return;
    const uint64_t f0 = a.full_off[c], f1 = a.full_off[c + 1], r0 = a.anchor_off[c], r1 = a.anchor_off[c + 1];
    const rawdtw_carry_t rec = a.carry[c];
    ChainDesc d;
    d.job_first = f1 >= 2 ? f1 - 2 : 0;
    d.n_jobs = f1 > f0 ? (uint32_t)(f1 - f0 - 1) : 0u;
    d.descending = 1; d.span = 0; d.num_aligned = 0;
    // (the short list: the new entries, then the junction when a stretch is taken over)
    const bool counts_ok = (r1 - r0) + rec.parts == f1 - f0 && (rec.parts == 0 || (rec.prev_src != ~0ull && rec.prev_src + rec.parts <= a.prev_n_full));
    if (f1 > f0 && r1 > r0 && counts_ok) {
        const rawdtw_anchor_t last = a.anchors[r0], first = rec.parts ? rec.start : a.anchors[r1 - 1];
        d.span = last.query_position - first.query_position + 1;                       // rmap.cpp:245
        d.num_aligned = (last.query_position - first.query_position) + d.n_jobs;         // rmap.cpp:292
        if (d.n_jobs && ((uint64_t)a.read_base[c] + last.query_position + 1ull > a.n_ev || a.ref_base[c] + last.target_position + 1ull > a.n_ref ||
                         first.query_position > last.query_position || first.target_position > last.target_position))
            atomicMin(&a.cnt[kCntBad], (unsigned long long)r0);
    } else if (f1 > f0) atomicMin(&a.cnt[kCntBad], (unsigned long long)r0); // (a record that does not add up)
    else atomicAdd(&a.cnt[kCntUnsupported], 1ull);                          // a chain without anchors: align_chain would read anchors[-1]
    chains[c] = d;
}

constexpr int kGatherT = 256;
__global__ __launch_bounds__(kGatherT) void k_gather(const StreamArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint64_t c = (uint64_t)blockIdx.x * (kGatherT / 64) + (threadIdx.x >> 6);
    if (c >= a.n_chains) return;
    // (a batch the scan declined is redone through the job list)
    if (a.cnt[kCntBad] != ~0ull || a.cnt[kCntOverflow] != ~0ull || a.cnt[kCntUnsupported] != 0ull || a.cnt[kCntOthers] > a.others_cap) return;
    const uint64_t f0 = a.full_off[c], r0 = a.anchor_off[c], r1 = a.anchor_off[c + 1];
    const rawdtw_carry_t rec = a.carry[c];
    const uint64_t L = rec.parts, n_new = r1 > r0 ? r1 - r0 - 1 : 0; // the short list's parts: all of them new
    for (uint64_t k = (uint64_t)lane; k < n_new; k += 64) a.out_full[f0 + k] = a.out[r0 + k];
    if (!L) return;
    // (the round before must have stood: a batch the scan declined has no costs to take over)
    const bool prev_ok = a.prev_cnt[kCntBad] == ~0ull && a.prev_cnt[kCntUnsupported] == 0ull && a.prev_cnt[kCntOthers] <= a.prev_others_cap &&
                         a.prev_cnt[kCntOverflow] == ~0ull;
    if (!prev_ok) { if (lane == 0) atomicMin(&a.cnt[kCntBad], (unsigned long long)r0); return; }
    const float *src = a.prev_out_full + rec.prev_src;
    float *dst = a.out_full + f0 + n_new;
    for (uint64_t k = (uint64_t)lane; k < L; k += 64) {
        float cost = src[k];
        if (k == 0 && (rec.flags & 1u)) { // its last cell's distance: the junction is where it ends
            const rawdtw_anchor_t e = a.anchors[r1 - 1];
            cost = cost - dist(a.ev[(uint64_t)a.read_base[c] + e.query_position], a.ref[a.ref_base[c] + e.target_position]);
        }
        dst[k] = cost;
    }
}

// grid: workgroup 0 the fold order (one workgroup's latency chain, ~30 us: dispatched first, it runs beside everything
// else instead of behind it), then the chain records, then the scan units
__global__ __launch_bounds__(kScanT, 8) void k_scan(const StreamArgs a, ChainDesc *__restrict__ chains, uint32_t *__restrict__ order)
{
    const uint32_t b = blockIdx.x, n_desc = (uint32_t)((a.n_chains + kScanT - 1) / kScanT);
    if (b == 0) { if (a.n_chains && order) fold_order_body<(int)kScanT>(a.n_chains, a.carry ? a.full_off : a.anchor_off, order); }
    else if (b <= n_desc) {
        if (a.carry) chain_desc_carry(a, chains, (uint64_t)(b - 1) * kScanT + threadIdx.x); // (a chunk round: the records describe the FULL chains)
        else chain_desc_body(a, chains, (uint64_t)(b - 1) * kScanT + threadIdx.x);
    }
    else scan_unit_body<false>(a, b - 1 - n_desc);
}

// The compact hand-over: the units decode their anchors first (rawdtw_batch_submit_compact).  The chain records read the
// chains' first and last anchors: the first travels whole (`heads`), the last they cannot know before the units have
// written it -- so they run as a launch of their own behind this one (k_scan_desc).
__global__ __launch_bounds__(kScanTC) void k_scan_compact(const StreamArgs a, uint32_t *__restrict__ order)
{
    const uint32_t b = blockIdx.x;
    if (b == 0) { if (a.n_chains && order) fold_order_body<(int)kScanTC>(a.n_chains, a.anchor_off, order); }
    else scan_unit_body<true>(a, b - 1);
}
__global__ __launch_bounds__(kScanT) void k_scan_desc(const StreamArgs a, ChainDesc *__restrict__ chains)
{
    chain_desc_body(a, chains, (uint64_t)blockIdx.x * kScanT + threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_plan: the DTW launch's passes, a wave a tile of the scan's tile list.  A tile = 512 consecutive anchors; lane L holds its
// positions 8 L .. 8 L + 7; the item at position t ends at anchor t, starts at anchor t + 1 and is item u = 511 - t of the
// layout order: along a chain (stored end-first) u ascends with the addresses.  A run = consecutive tile parts of a chain;
// its first part (`starts`) is the one whose predecessor in u (position + 1) is no tile part.  The wave lays the tile's
// image out (RunTab), sorts its jobs into the order the lanes of k_runs take them, and leaves in memory, per PASS (all of the
// tile's parts, or as many as fit the image budget and the run table): the job records, the copy orders, the list entry.
// The waves of a workgroup share nothing: no workgroup barrier.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPlanT = 256;
// (three waves a SIMD: 168 registers, 7 of them spilled; at four the spills cost more than the occupancy brings, at two the
// launch is 15 % slower; a version with the items in LDS and rolled loops -- 71 registers, 35 KB of LDS a workgroup -- ran as
// fast alone and 5 % slower inside the pipeline, as did one-wave workgroups)
__global__ __launch_bounds__(kPlanT, 3) void k_plan(const StreamArgs a)
{
    constexpr int KI = 8;
    constexpr uint32_t kWords = kStreamTile / 32 + 1;
    static_assert(kStreamTile == 64 * KI, "a wave a tile, eight items a lane");
    __shared__ uint32_t s_hist[kPlanT / 64][kSortBins];
    __shared__ RunTab s_rtab[kPlanT / 64];
    __shared__ uint32_t s_tmp[kPlanT / 64][8];
    __shared__ uint32_t s_maskw[kPlanT / 64][kWords + 1], s_prew[kPlanT / 64][kWords + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = (uint32_t)tid >> 6;
    const uint32_t li = blockIdx.x * (kPlanT / 64) + wv; // the wave's entry of the tile list; its first pass takes the same index
    const uint2 te = a.tlist[min(li, a.n_tiles - 1u)];  // (asked for with the counters: one round trip, not two; used if it exists)
    // (a batch the scan declined is redone through the job list: nothing may be derived from its anchors)
    if (a.cnt[kCntBad] != ~0ull || a.cnt[kCntUnsupported] != 0ull || a.cnt[kCntOthers] > a.others_cap) return;
    const uint32_t n_list = (uint32_t)min<unsigned long long>(a.cnt[kCntTodo], (unsigned long long)a.n_tiles);
    if (li >= n_list) return;
    const uint32_t tile = te.x;
    const uint64_t c0 = te.y, base = (uint64_t)tile * kStreamTile;
    uint32_t *hist = s_hist[wv], *tmp = s_tmp[wv], *mask = s_maskw[wv], *pre = s_prew[wv];
    RunTab &rt = s_rtab[wv];
    // the tile's anchors: KI + 1 consecutive entries a lane (the last one is the next lane's first)
    rawdtw_anchor_t an[KI + 1];
    const uint64_t i0 = base + (uint64_t)lane * KI;
#pragma unroll
    for (int k = 0; k <= KI; k++) an[k] = i0 + k < a.n_anchors ? a.anchors[i0 + k] : rawdtw_anchor_t{0, 0};
    // chain starts inside [base, base + 512] from the chains' offsets, starting at the chain that owns the tile's first anchor
    // (the end of the anchor list counts as a chain start); chains are few a tile: mostly one round of loads
    if ((uint32_t)lane < kWords) mask[lane] = 0;
    wave_lds_sync();
    for (uint64_t c = c0 + (uint32_t)lane;; c += 64) { // (the exit is wave-uniform)
        const bool in = c <= a.n_chains;
        const uint64_t st = in ? a.anchor_off[c] : ~0ull;
        if (in && st >= base && st <= base + kStreamTile) atomicOr(&mask[(uint32_t)(st - base) >> 5], 1u << ((uint32_t)(st - base) & 31u));
        const unsigned long long more = __ballot(in && st < base + kStreamTile);
        if (!(more >> 63)) break; // (the offsets ascend: the last lane's is the largest)
    }
    wave_lds_sync();
    {   // chain starts before each word of the mask
        const uint32_t pc = (uint32_t)lane < kWords ? __popc(mask[lane]) : 0u;
        const uint32_t incl = wave_scan_incl(pc);
        if ((uint32_t)lane < kWords) pre[lane] = incl - pc;
    }
    wave_lds_sync();
    // the chain of the anchor at position p of the tile: c0 + the chain starts in positions 1 .. p
    auto chain_at = [&](uint32_t p) { return c0 + (pre[p >> 5] + __popc(mask[p >> 5] & (0xffffffffu >> (31u - (p & 31u)))) - (mask[0] & 1u)); };
    // the tile parts: N | M << 7 | R << 14 | exclude_last << 16 | swapped << 17 | 1 << 20 (0: the lane bodies do not take it)
    uint32_t tm[KI];
#pragma unroll
    for (int k = 0; k < KI; k++) {
        const uint32_t p = (uint32_t)lane * KI + k;
        tm[k] = 0u;
        if (base + p >= a.n_anchors || mask_bit(mask, p + 1)) continue; // no part ends here
        const Part pt = classify(a, an[k + 1], an[k]);
        if (!pt.tile) continue;
        const uint32_t N = pt.n > pt.m ? pt.n : pt.m, M = pt.n > pt.m ? pt.m : pt.n;
        tm[k] = N | (M << 7) | ((uint32_t)pt.R << 14) | ((mask_bit(mask, p) ? 0u : 1u) << 16) | ((pt.n < pt.m ? 1u : 0u) << 17) | (1u << 20);
    }
    {
        const int t_first = (int)((tm[0] >> 20) & 1u), t_last = (int)((tm[KI - 1] >> 20) & 1u);
        const bool below = __builtin_amdgcn_update_dpp(0, t_last, 0x138, 0xf, 0xf, false) != 0;  // lane - 1's last item (lane 0: none)
        const bool above = __builtin_amdgcn_update_dpp(0, t_first, 0x130, 0xf, 0xf, false) != 0; // lane + 1's first item (lane 63: none)
        // an item's contributions to the regions' running sums (events | run start << 20; reference), from its record
        // (kept short on registers: recomputed where needed rather than held for the eight items)
        auto contrib = [](const uint32_t t, uint32_t &c_r, uint32_t &c_f) {
            const bool tl = (t >> 20) & 1u, starts = (t >> 18) & 1u, ends = (t >> 19) & 1u, swap = (t >> 17) & 1u;
            const uint32_t N = t & 127u, M = (t >> 7) & 127u, n = swap ? M : N, m = swap ? N : M;
            c_r = tl ? ((starts ? n + 3u : n - 1u) + (ends ? 3u : 0u)) | ((starts ? 1u : 0u) << 20) : 0u;
            c_f = tl ? (starts ? m + 3u : m - 1u) + (ends ? 3u : 0u) : 0u;
        };
        uint32_t lr = 0, lf = 0;
#pragma unroll
        for (int k = 0; k < KI; k++) {
            const bool t = (tm[k] >> 20) & 1u;
            const bool pred = k + 1 < KI ? ((tm[k + 1 < KI ? k + 1 : k] >> 20) & 1u) != 0u : above;
            const bool succ = k > 0 ? ((tm[k > 0 ? k - 1 : 0] >> 20) & 1u) != 0u : below;
            tm[k] |= ((t && !pred ? 1u : 0u) << 18) | ((t && !succ ? 1u : 0u) << 19);
            uint32_t c_r, c_f;
            contrib(tm[k], c_r, c_f);
            lr += c_r; lf += c_f;
        }
        // sums in layout order from sums in position order: before item u lie the items at higher positions,
        //     sum over v <= u  =  total - (sum over positions <= t) + own
        const uint32_t ir = wave_scan_incl(lr), jf = wave_scan_incl(lf);
        const uint32_t tot_r = (uint32_t)__builtin_amdgcn_readlane((int)ir, 63), tot_f = (uint32_t)__builtin_amdgcn_readlane((int)jf, 63);
        const uint32_t base_r = tot_r - (ir - lr), base_f = tot_f - (jf - lf); // total - (the lanes below this one)
        const uint32_t budget = a.lds_floats & ~3u;
        uint32_t u0 = 0, b0r = 0, b0f = 0, b0s = 0; // the pass's first item, the sums before it
        uint32_t rec_off = 0;                       // the pass's first record in the tile's stretch of the record array
        for (bool first = true;; first = false) { // (wave-uniform)
            // the items that fit: while the image of the parts so far stays inside the budget and their runs in the table (+ 4 + 4
            // floats and one run when the first item continues a run); the sums ascend with u: the fitting items are a prefix
            // (the rule: everything that is left fits -- the totals say so, and no item has to be looked at)
            const bool all_fit = ((((tot_r & 0xfffffu) - b0r + 7u) & ~3u) + ((tot_f - b0f + 7u) & ~3u) <= budget) && ((tot_r >> 20) - b0s + 1u <= kStreamMaxSeg);
            uint32_t u1 = kStreamTile, e_r = tot_r & 0xfffffu, e_f = tot_f, e_s = tot_r >> 20;
            bool cut_run = false;
            if (!all_fit || u0 != 0u) {
                uint32_t fits = 0;
                {
                    uint32_t xr = base_r, xf = base_f; // inclusive layout-order sums: own + everything at higher positions
#pragma unroll
                    for (int k = 0; k < KI; k++) {
                        uint32_t c_r, c_f;
                        contrib(tm[k], c_r, c_f);
                        const uint32_t u = kStreamTile - 1u - ((uint32_t)lane * KI + k), sr = xr & 0xfffffu, ss = xr >> 20;
                        if (u >= u0 && ((sr - b0r + 7u) & ~3u) + ((xf - b0f + 7u) & ~3u) <= budget && ss - b0s + 1u <= kStreamMaxSeg) fits++;
                        xr -= c_r; xf -= c_f;
                    }
                }
                fits = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(fits), 63);
                u1 = min(u0 + fits, kStreamTile);
                if (u1 <= u0) { if (lane == 0) atomicMin(&a.cnt[kCntOverflow], (unsigned long long)tile); break; } // (cannot happen: one part always fits)
                // the owner of the pass's last item publishes the sums behind it; the owner of its first item, whether it cuts a run
                {
                    uint32_t xr = base_r, xf = base_f;
#pragma unroll
                    for (int k = 0; k < KI; k++) {
                        uint32_t c_r, c_f;
                        contrib(tm[k], c_r, c_f);
                        const uint32_t u = kStreamTile - 1u - ((uint32_t)lane * KI + k);
                        if (u == u1 - 1u) { tmp[0] = xr & 0xfffffu; tmp[1] = xf; tmp[2] = xr >> 20; }
                        if (u == u0) tmp[3] = ((tm[k] >> 20) & 1u) && !((tm[k] >> 18) & 1u) ? 1u : 0u;
                        xr -= c_r; xf -= c_f;
                    }
                }
                wave_lds_sync();
                e_r = tmp[0]; e_f = tmp[1]; e_s = tmp[2];
                cut_run = tmp[3] != 0u; // the pass's first item continues a run of the pass before: it starts one here
            }
            const uint32_t region = (e_r - b0r + (cut_run ? 4u : 0u) + 3u) & ~3u;
            const uint32_t n_runs = e_s - b0s + (cut_run ? 1u : 0u);
            const bool last = u1 >= kStreamTile;
            // the pass's slot of copy orders: the tile's own for its first pass, one of the pool's for the others
            uint32_t slot = li;
            if (!first) {
                uint32_t sl = 0;
                if (lane == 0) sl = a.n_tiles + (uint32_t)atomicAdd(&a.cnt[kCntPool], 1ull);
                slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl);
                if (slot >= a.n_slots) { if (lane == 0) atomicMin(&a.cnt[kCntOverflow], (unsigned long long)tile); break; }
            }
            for (uint32_t b = (uint32_t)lane; b < kSortBins; b += 64u) hist[b] = 0;
            wave_lds_sync(); // (the words above are read, the bins are clear)
            // the bins' sizes, the run table
            {
                uint32_t xr = base_r, xf = base_f;
#pragma unroll
                for (int k = 0; k < KI; k++) {
                    uint32_t c_r, c_f;
                    contrib(tm[k], c_r, c_f);
                    const uint32_t u = kStreamTile - 1u - ((uint32_t)lane * KI + k);
                    const uint32_t sr = xr & 0xfffffu, sf = xf, ss = xr >> 20;
                    xr -= c_r; xf -= c_f;
                    if (!((tm[k] >> 20) & 1u) || u < u0 || u >= u1) continue;
                    const uint32_t N = tm[k] & 127u, R = (tm[k] >> 14) & 3u;
                    const bool starts = ((tm[k] >> 18) & 1u) || u == u0, ends = ((tm[k] >> 19) & 1u) || u == u1 - 1u;
                    atomicAdd(&hist[(3u - R) * 64u + (63u - min(N, 63u))], 1u);
                    const uint32_t adj = (cut_run && u > u0) ? 1u : 0u;
                    const uint32_t g = ss - b0s + (cut_run ? 1u : 0u) - 1u; // the item's run in this pass
                    if (starts) {
                        // the run's first part leaves what its run's table entry is made of -- the sums before it (its own
                        // contribution off; a cut run's first part counts as a start: + 4 behind it), its start anchor, its
                        // position -- in the entry's words; lane g makes the entry of it below, once for all runs (here the
                        // chain's bases would be fetched in eight divergent rounds, one per item of a lane)
                        const rawdtw_anchor_t sa = an[k + 1];
                        rt.lo[0][g] = sr - (c_r & 0xfffffu) - b0r + 4u * adj; rt.lo[1][g] = sf - c_f - b0f + 4u * adj;
                        rt.D[0][g] = (int32_t)sa.query_position; rt.D[1][g] = (int32_t)sa.target_position;
                        rt.src[0][g] = (long long)((uint32_t)lane * KI + k);
                    }
                    if (ends) { rt.end[0][g] = an[k].query_position + 1u; rt.end[1][g] = an[k].target_position + 1u; }
                }
            }
            wave_lds_sync(); // (the runs' first parts and the counts are complete)
            if ((uint32_t)lane < n_runs) { // run g = lane: its chain (a popcount over the chain-start mask), its bases, its place in the image
                const uint32_t g = (uint32_t)lane;
                const uint32_t q_r = rt.lo[0][g], q_f = rt.lo[1][g], sq = (uint32_t)rt.D[0][g], st = (uint32_t)rt.D[1][g];
                const uint64_t c = chain_at((uint32_t)rt.src[0][g]);
                const uint64_t rb = a.ref_base[c];
                const uint32_t qb = a.read_base[c];
                const uint32_t off_r = qb + sq;
                const uint64_t off_f = rb + st;
                const uint32_t p_r = q_r + ((off_r - q_r) & 3u), p_f = region + q_f + (((uint32_t)off_f - q_f) & 3u);
                rt.lo[0][g] = p_r & ~3u; rt.lo[1][g] = p_f & ~3u;
                rt.D[0][g] = (int32_t)(p_r - sq); rt.D[1][g] = (int32_t)(p_f - st);
                rt.src[0][g] = (long long)off_r - (long long)p_r; rt.src[1][g] = (long long)off_f - (long long)p_f;
            }
            wave_lds_sync(); // (the run table is complete)
            // the bins' first places (three a lane), written back over the counts
            const uint32_t h0 = hist[3 * lane], h1 = hist[3 * lane + 1], h2 = hist[3 * lane + 2];
            const uint32_t hsum = h0 + h1 + h2, hincl = wave_scan_incl(hsum);
            const uint32_t n_jobs = (uint32_t)__builtin_amdgcn_readlane((int)hincl, 63);
            hist[3 * lane] = hincl - hsum; hist[3 * lane + 1] = hincl - hsum + h0; hist[3 * lane + 2] = hincl - hsum + h0 + h1;
            wave_lds_sync();
            // the records, in the order the lanes of the DTW launch take them (a bin's jobs in any order: they are alike)
            {
                uint32_t xs = base_r >> 20;
#pragma unroll
                for (int k = 0; k < KI; k++) {
                    const uint32_t u = kStreamTile - 1u - ((uint32_t)lane * KI + k), ss = xs;
                    xs -= (tm[k] >> 18) & (tm[k] >> 20) & 1u; // (a run start counts once)
                    if (!((tm[k] >> 20) & 1u) || u < u0 || u >= u1) continue;
                    const uint32_t N = tm[k] & 127u, R = (tm[k] >> 14) & 3u;
                    const uint32_t place = atomicAdd(&hist[(3u - R) * 64u + (63u - min(N, 63u))], 1u);
                    const uint32_t g = ss - b0s + (cut_run ? 1u : 0u) - 1u;
                    const rawdtw_anchor_t sa = an[k + 1];
                    const uint32_t p_r = sa.query_position + (uint32_t)rt.D[0][g], p_f = sa.target_position + (uint32_t)rt.D[1][g];
                    const bool swap = (tm[k] >> 17) & 1u;
                    a.recs[(uint64_t)tile * kStreamRecStride + rec_off + place] = make_uint2((swap ? p_f : p_r) | ((swap ? p_r : p_f) << 16), (tm[k] & 0x1ffffu) | (u << 17));
                }
            }
            // the copy orders: a run's range of 16-byte pieces, per arena
            if ((uint32_t)lane < 2u * n_runs) {
                const uint32_t g = (uint32_t)lane >> 1, w = (uint32_t)lane & 1u;
                const long long src = rt.src[w][g];
                a.runtab[(uint64_t)slot * (2u * kStreamMaxSeg) + lane] =
                    make_uint4(rt.lo[w][g] >> 2, (rt.end[w][g] + (uint32_t)rt.D[w][g] + 3u) >> 2, (uint32_t)(unsigned long long)src, (uint32_t)((unsigned long long)src >> 32));
            }
            wave_lds_sync(); // (the run table and the bins are read: the next pass writes them again)
            if (lane == 0) a.todo[slot] = make_uint4(tile, slot, n_jobs | (n_runs << 16), region | (rec_off << 16)); // (a pass's list entry sits at its slot)
            if (last) break;
            rec_off += (n_jobs + 1u) & ~1u; // (passes start on 16-byte boundaries)
            if (rec_off + (kStreamTile - u1) > kStreamRecStride) { if (lane == 0) atomicMin(&a.cnt[kCntOverflow], (unsigned long long)tile); break; } // (> 64 passes)
            u0 = u1; b0r = e_r; b0f = e_f; b0s = e_s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_side: the side list in class order: wave-per-job classes first, longest first; then 16-lane groups, the lane classes
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kSideGroups = 64;
__global__ __launch_bounds__(1024) void k_side(const StreamArgs a)
{
    constexpr int NT = 1024;
    const uint32_t group = blockIdx.x;
    // Each workgroup orders one contiguous slice of the unordered list: class counts of the slice in LDS, ONE returning
    // atomic per class and workgroup for the slice's places (same-address atomics run near 88 per microsecond: a wave-level
    // scheme spends the kernel there once the list has 10^5 entries), then the scatter through LDS cursors.
    const uint64_t n_other = min<uint64_t>(a.cnt[kCntOthers], a.others_cap);
    const uint64_t per = (n_other + kSideGroups - 1) / kSideGroups;
    const uint64_t lo = min(n_other, (uint64_t)group * per), hi = min(n_other, lo + per);
    if (lo >= hi) return;
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
// k_runs: the batch's one DTW launch
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kStamps = 10;

// the largest value of the wave, in every lane
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t n_max)
{
    n_max = max(n_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n_max, 0x111, 0xf, 0xf, false)); // row_shr:1
    n_max = max(n_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n_max, 0x112, 0xf, 0xf, false)); // row_shr:2
    n_max = max(n_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n_max, 0x114, 0xf, 0xf, false)); // row_shr:4
    n_max = max(n_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n_max, 0x118, 0xf, 0xf, false)); // row_shr:8
    n_max = max(n_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n_max, 0x142, 0xa, 0xf, false)); // row_bcast:15
    n_max = max(n_max, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n_max, 0x143, 0xc, 0xf, false)); // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)n_max, 63);
}

// One sorted chunk of a tile: 64 lanes, one job each.  The chunks of a tile are cut from one order -- the jobs of radius 3
// first, then 2, then 1, each run by longer side, descending; a wave takes the shortest body that covers its radii.
__device__ __forceinline__ float stream_lane_job(const float *LA, const float *LB, uint32_t N, uint32_t M, uint32_t R, bool excl, bool act)
{
    const unsigned long long r12 = __ballot(R <= 2u), r1 = __ballot(R == 1u), r2 = __ballot(R == 2u);
    // the wave's longest side (its jobs come in up to three runs, each longest first; sides of 63 and more share a bin)
    const uint32_t n_max = wave_max_u32(N);
    float res;
    if (~r1 == 0ull) res = lane_dp_r1(LA, LB, N, n_max); // (radius 1: N == M)
    else if (~r2 == 0ull) res = lane_dp_r2(LA, LB, N, M, n_max);
    else if (~r12 == 0ull) res = lane_dp_r12(LA, LB, N, M, R, n_max);
    else res = lane_dp_gen(LA, LB, N, M, R, n_max);
    if (act && excl) res = res - dist(LA[N - 1], LB[M - 1]);
    return res;
}

// 64 jobs of the side list's lane classes: one lane per job, operands straight from the arenas (the jobs of a wave come
// from all over the batch: nothing to stage together)
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

// Tile queue: one returning atomic on a single word saturates near 88 dequeues per microsecond (MI355X_MICROARCH.md),
// which a batch's ten thousand tiles would reach; eight heads on lines of their own, each dealing every eighth tile.
// A workgroup starts on the head of its block index and moves on when a head runs dry.
__device__ __forceinline__ uint32_t next_tile(const StreamArgs &a, const uint32_t dbg, uint32_t &head, uint32_t n_tiles, const uint32_t base = 0u)
{
    if (dbg & 8u) { // timing experiments: tiles dealt by block index, no queue
        const uint32_t t = head;
        head += gridDim.x;
        return t < n_tiles ? t : 0xffffffffu;
    }
    for (uint32_t tries = 0; tries < 8; tries++) {
        const uint32_t h = (head + tries) & 7u;
        const uint32_t k = (uint32_t)atomicAdd(&a.cnt[kCntHeads + 16 * h], 1ull);
        const uint64_t t = (uint64_t)base + (uint64_t)k * 8u + h; // (the list's first `base` entries are dealt by block index: k_runs)
        if (t < n_tiles) { head = h; return (uint32_t)t; }
    }
    return 0xffffffffu;
}


} // namespace

// k_wide: the side list -- the one part in two hundred whose band the lane bodies of the tiles do not take -- as a launch of
// its own (inside k_runs its waves kept a third of the workgroups from the tiles for a third of the launch): wave-
// cooperative jobs, longest first, dealt over the waves of the grid.  It needs no LDS and ends with its longest job
// (a wave per job: 163 ns a column).  In line on the batch's stream, between the scan and the pass planning
// (rawdtw_batch.cpp: stream_wide_fork; on a second stream beside the tiles it cost more in fork / join than it hid).
__global__ __launch_bounds__(256) void k_wide(const StreamArgs a)
{
    const uint32_t dbg = a.debug;
    constexpr uint32_t kWaves = 4;
    __shared__ __attribute__((aligned(16))) float s_w[kWaves][kWbandLdsFloats]; // the wave-per-job bands' operand windows (wband_gen)
    __shared__ uint32_t s_next;                                                 // the workgroup's next item
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = (uint32_t)tid >> 6;
    // (a batch the scan declined is redone through the job list)
    if (a.cnt[kCntBad] != ~0ull || a.cnt[kCntUnsupported] != 0ull || a.cnt[kCntOthers] > a.others_cap) return;
    // ---- the side list: wave-cooperative jobs dealt over ALL waves of the grid, wave-per-job classes (longest first) to the
    // first waves: a long job starts at once and runs next to the tiles instead of behind them ----
    if (!(dbg & 4u)) {
        uint64_t n_w = 0;
#pragma unroll
        for (uint32_t c = kClsW0; c < kClsW0 + 4; c++) n_w += a.cnt[kCntCls0 + c];
        const uint64_t n_g16 = a.cnt[kCntCls0 + kClsG16];
        uint64_t n_l = 0, n_m = 0;
#pragma unroll
        for (uint32_t c = kClsL0; c < kClsL0 + kClsLCount; c++) n_l += a.cnt[kCntCls0 + c];
#pragma unroll
        for (uint32_t c = kClsM0; c < kClsM0 + kClsMCount; c++) n_m += a.cnt[kCntCls0 + c];
        // item order = list order: wave-per-job (longest first), 16-lane groups, lane-per-job (4 slots, then 8; by length)
        const uint64_t it_g16 = (n_g16 + 3) / 4, it_l = (n_l + 63) / 64, it_m = (n_m + 63) / 64, items = n_w + it_g16 + it_l + it_m;
        // Items are dealt to the workgroups like a snake (workgroup b: items b, 2 G - 1 - b, 2 G + b, ...: the one that drew the
        // longest item of a turn draws the shortest of the next) and inside a workgroup its four waves pull the workgroup's items
        // from a counter in LDS: the wave that sits on a long job (one job of 950 columns is 70 us, a short item 1-2 us) takes no
        // other, its three neighbours share what the workgroup was dealt.  (Dealt to the WAVES by index, the wave with the
        // batch's longest job also owned some twenty-five short items: 35 us behind a 70 us job.)
        const uint32_t G = gridDim.x;
        const uint32_t n_items = (uint32_t)min<uint64_t>(items, 0xffffffffull);
        if (tid == 0) s_next = 0;
        __syncthreads();
        for (;;) {
            uint32_t kq = 0;
            if (lane == 0) kq = atomicAdd(&s_next, 1u);
            kq = (uint32_t)__builtin_amdgcn_readfirstlane((int)kq);
            const uint64_t first = (uint64_t)kq * G;
            if (first >= n_items) break;
            const uint64_t it = first + ((kq & 1u) ? G - 1u - blockIdx.x : blockIdx.x);
            if (it >= n_items) continue; // (the last, partial turn)
            if (it < n_w) {
                if (dbg & 32u) continue;
                // the longest jobs bound the launch: a job of hundreds of columns is one dependent chain, and shares its
                // SIMD with the waves around it -- it goes first in the issue order
                const DevJob jb = a.ojobs[it];
                const uint32_t len = max(jb.n, jb.m);
                if (len >= 256u) __builtin_amdgcn_s_setprio(3);
                else if (len >= 96u) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(1);
                wreg_small_job(jb, lane, a.ev, a.ref, a.out, s_w[wv]);
                __builtin_amdgcn_s_setprio(0);
            }
            else if (dbg & 64u) continue;
            else if (it < n_w + it_g16) grp_wave<16>(a.ojobs + n_w, (uint32_t)n_g16, (uint32_t)(it - n_w), lane, a.ev, a.ref, a.out);
            else if (it < n_w + it_g16 + it_l) lane_global_wave<4>(a.ojobs + n_w + n_g16, (uint32_t)n_l, (uint32_t)(it - n_w - it_g16), lane, a.ev, a.ref, a.out);
            else lane_global_wave<8>(a.ojobs + n_w + n_g16 + n_l, (uint32_t)n_m, (uint32_t)(it - n_w - it_g16 - it_l), lane, a.ev, a.ref, a.out);
        }
    }

}

// k_runs: TT threads per workgroup (256 or 512), a persistent grid over the scan's work list.  A list entry is one PASS: up
// to 512 tile-class parts whose windows fit the image budget, with its job records (sorted: radius class, then longer side)
// and its copy orders (a run's range of 16-byte pieces, per arena) waiting in memory -- k_plan wrote them -- so a pass here
// is: stage the image by LDS-DMA, one barrier, the lanes' DP (a wave pulls 64 records at a time; the radius-3 records at
// the head of the order sixteen at a time, four lanes a job: run_dp), the costs straight to out[anchor], one barrier.  Nothing a pass needs from memory is waited for at its
// start: while pass i is computed, pass i + 1's records and copy orders come in by LDS-DMA into the other buffer, pass
// i + 2's list entry is on its way and pass i + 3's ticket is being drawn.
// DIAG: the instance with the timing experiments ("stream_debug" masks) and the phase stamps; the production instance
// carries none of their branches.
template <int TT, bool DIAG>
__global__ __launch_bounds__(TT, 4) void k_runs(const StreamArgs a, const uint32_t lds_floats)
{
    const uint32_t dbg = DIAG ? a.debug : 0u;
    constexpr uint32_t kWaves = TT / 64, kRT = 2u * kStreamMaxSeg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *win = smem;                                                    // the pass's LDS image
    uint2 *rec = reinterpret_cast<uint2 *>(smem + lds_floats);            // kStreamTile job records: the pass's (they come in with its image)
    uint4 *rtab = reinterpret_cast<uint4 *>(rec + kStreamTile);           // 2 x kRT copy orders: the pass's and the next one's
    __shared__ uint4 s_ent[2];                 // the passes' list entries, by parity (x = 0xffffffff: none)
    __shared__ uint32_t s_seq;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wv = (uint32_t)tid >> 6;

    // a batch the scan declined (invalid anchors, a chain without anchors, a band nobody takes) is redone through the job
    // list: nothing to do here, and nothing may be derived from its records
    // (every wave reads the counters itself -- uniform loads, asked for together with the list's sizes below: one round trip
    // and no barrier in front of a workgroup's first pass)
    if (a.cnt[kCntBad] != ~0ull || a.cnt[kCntOverflow] != ~0ull || a.cnt[kCntUnsupported] != 0ull || a.cnt[kCntOthers] > a.others_cap) return;

    // LDS-DMA: one wave instruction moves 16 bytes a lane straight into LDS at `dst` + 16 * lane; inactive lanes move
    // nothing.  `dst` must be the same in every lane.
    auto dma16 = [](const void *src, void *dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    // a pass's records (all waves: two records a piece): asked for with the pass's image and landed with it -- the lanes read
    // them behind B1 only, so they need no buffer of their own a pass ahead (4 KB of LDS a workgroup: a fifth workgroup a CU)
    auto fetch_recs = [&](const uint4 e) {
        const uint32_t n_jobs = e.z & 0xffffu, pieces = (n_jobs + 1u) >> 1;
        const uint2 *src = a.recs + (uint64_t)e.x * kStreamRecStride + (e.w >> 16);
        for (uint32_t q0 = wv * 64u; q0 < pieces; q0 += (uint32_t)TT)
            if (q0 + (uint32_t)lane < pieces) dma16(src + 2u * (q0 + (uint32_t)lane), rec + 2u * q0);
    };
    // ... and its copy orders, which the staging itself reads: a pass ahead, into buffer `buf` (one wave)
    auto fetch_orders = [&](const uint4 e, const uint32_t buf) {
        const uint32_t n_ord = 2u * (e.z >> 16);
        if ((uint32_t)lane < n_ord) dma16(a.runtab + (uint64_t)e.y * kRT + lane, rtab + buf * kRT);
    };

    // diagnostic build of the launch ("stream_debug" 256): where a wave's cycles go, phase by phase (s_memtime around the
    // phases, summed per wave in LDS and added to the counter block's words kCntStamp0.. at the end; it perturbs the run)
    __shared__ unsigned long long s_stamp[kWaves][kStamps];
    unsigned long long t_prev = 0;
    const bool stamps = DIAG && (dbg & 256u) != 0u;
    if (stamps) {
        if (lane < (int)kStamps) s_stamp[wv][lane] = 0;
        t_prev = __builtin_amdgcn_s_memtime();
    }
    auto stamp = [&](const int ph) {
        if (!stamps) return;
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) s_stamp[wv][ph] += t - t_prev;
        t_prev = t;
    };
    // ---- passes: the entries of the scan's work list, pulled from the queue two ahead ----
    // list index i: the first pass of the scan's i-th tile (slot i), then the pool's passes (slots n_tiles ..)
    const uint32_t n_first = (uint32_t)min<unsigned long long>(a.cnt[kCntTodo], (unsigned long long)a.n_tiles);
    const uint32_t n_pass = n_first + (uint32_t)min<unsigned long long>(a.cnt[kCntPool], (unsigned long long)(a.n_slots - a.n_tiles));
    auto entry_of = [&](const uint32_t i) { return a.todo[i < n_first ? i : a.n_tiles + (i - n_first)]; };
    // The queue, three passes deep, so that no pass waits for a round trip at its start.  At the top of pass t thread 0 turns
    // the ticket it drew a pass ago into the list index of pass t + 2 and draws the ticket of pass t + 3; wave 0 asks for
    // the entry of pass t + 2 (it has the whole pass to arrive) and publishes the entry of pass t + 1 (asked for a pass ago).
    // A workgroup's first two passes are dealt by block index (list entries b and b + grid): their entries can be asked for at
    // once -- two queue tickets in front of the first pass were two dependent round trips.  The queue deals the entries from
    // 2 * grid on.
    uint32_t head = (dbg & 8u) ? blockIdx.x : (blockIdx.x & 7u);
    const uint32_t q_base = (dbg & 8u) ? 0u : 2u * gridDim.x;
    unsigned long long ticket = 0;
    bool more = false; // (thread 0) a ticket is out
    const uint4 none = make_uint4(0xffffffffu, 0u, 0u, 0u);
    uint4 e_next = none, e_load = none;
    if (wv == 0) {
        uint32_t i0 = 0xffffffffu, i1 = 0xffffffffu;
        if (dbg & 8u) {
            if (tid == 0) {
                i0 = next_tile(a, dbg, head, n_pass);
                i1 = i0 != 0xffffffffu ? next_tile(a, dbg, head, n_pass) : 0xffffffffu;
                more = i1 != 0xffffffffu;
            }
        } else {
            i0 = blockIdx.x < n_pass ? blockIdx.x : 0xffffffffu;
            i1 = blockIdx.x + gridDim.x < n_pass ? blockIdx.x + gridDim.x : 0xffffffffu;
            if (tid == 0) {
                more = i1 != 0xffffffffu;
                if (more) ticket = atomicAdd(&a.cnt[kCntHeads + 16 * head], 1ull); // pass 2's
            }
        }
        i0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i0);
        i1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i1);
        const uint4 e0 = i0 != 0xffffffffu ? entry_of(i0) : none;
        if (i1 != 0xffffffffu) e_next = entry_of(i1);
        if (tid == 0) s_ent[0] = e0;
    }
    __syncthreads();
    if (s_ent[0].x == 0xffffffffu) return;
    if (wv == kWaves - 1u) fetch_orders(s_ent[0], 0u); // the first pass's copy orders: nobody to fetch them ahead
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    // the DP of one pass: waves pull chunks of the sorted records (the heavy class first), so the waves of the workgroup
    // finish together whatever the mix
    auto run_dp = [&](const uint2 *rc_base, const uint32_t n_jobs, const uint32_t end_nom) {
        // The radius-3 records come first in the order: those among the pass's first 64 go sixteen to a wave, four lanes a
        // job (quad_dp_r3) -- chunks 0 .. q3 - 1; the records behind them 64 to a wave as ever (a tile with more than 64
        // radius-3 parts leaves the others to the lanes' generic body).
        uint32_t n3;
        {
            const uint2 r0 = rc_base[min((uint32_t)lane, n_jobs - 1u)];
            n3 = (uint32_t)__popcll(__ballot((uint32_t)lane < n_jobs && ((r0.y >> 14) & 3u) == 3u));
        }
        const uint32_t q3 = (n3 + 15u) >> 4, n_chunks = q3 + ((n_jobs - n3 + 63u) >> 6);
        for (uint32_t c = wv; !(dbg & 1u);) { // (a wave's first chunk is its own number: no round trip through the counter)
            if (c >= n_chunks) break;
            if ((dbg & 512u) && c == 0u) { uint32_t cn0 = 0; if (lane == 0) cn0 = atomicAdd(&s_seq, 1u); c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cn0); continue; } // (timing: a pass without its first chunk)
            if ((dbg & 1024u) && c != 0u) break; // (timing: a pass's first chunk only)
            if (c < q3) {
                const uint32_t r = c * 16u + ((uint32_t)lane >> 2);
                const bool act = r < n3;
                const uint2 rc = rc_base[act ? r : n3 - 1u];
                const uint32_t N = rc.y & 127u, M = (rc.y >> 7) & 127u, u = (rc.y >> 17) & (kStreamTile - 1u);
                const float *LA = win + (rc.x & 0xffffu), *LB = win + (rc.x >> 16);
                float res = quad_dp_r3(LA, LB, N, M, lane, wave_max_u32(N));
                if (act && ((lane & 3) == 2)) {
                    if ((rc.y >> 16) & 1u) res = res - dist(LA[N - 1], LB[M - 1]);
                    a.out[end_nom - 1u - u] = res;
                }
            } else {
                const uint32_t r = n3 + (c - q3) * 64u + lane;
                const bool act = r < n_jobs;
                const uint2 rc = rc_base[act ? r : n_jobs - 1u];
                const uint32_t N = rc.y & 127u, M = (rc.y >> 7) & 127u, R = (rc.y >> 14) & 3u, u = (rc.y >> 17) & (kStreamTile - 1u);
                const float res = stream_lane_job(win + (rc.x & 0xffffu), win + (rc.x >> 16), N, M, R, (rc.y >> 16) & 1u, act);
                if (act) a.out[end_nom - 1u - u] = res; // the part that ends at anchor (tile end - 1 - u)
            }
            uint32_t cn = 0;
            if (lane == 0) cn = atomicAdd(&s_seq, 1u);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cn);
        }
    };
    uint32_t cur = 0;
    for (;;) {
        const uint4 e = s_ent[cur];
        if (e.x == 0xffffffffu) break;
        const uint32_t n_jobs = e.z & 0xffffu, n_ord = 2u * (e.z >> 16);
        // thread 0: the next pass's entry is published before this pass's first barrier; the pass after that gets its list
        // index (the ticket drawn a pass ago) and wave 0 asks for its entry; the ticket of the pass after THAT is drawn
        uint32_t i2 = 0xffffffffu;
        if (tid == 0) {
            s_ent[cur ^ 1u] = e_next;
            s_seq = kWaves; // (the chunks behind the waves' first ones)
            if (more) {
                if (dbg & 8u) i2 = next_tile(a, dbg, head, n_pass);
                else {
                    const unsigned long long t = (unsigned long long)q_base + ticket * 8ull + head;
                    i2 = t < n_pass ? (uint32_t)t : next_tile(a, dbg, head, n_pass, q_base); // (this head is dry: try the others)
                }
                more = i2 != 0xffffffffu;
                if (more && !(dbg & 8u)) ticket = atomicAdd(&a.cnt[kCntHeads + 16 * head], 1ull);
            }
        }
        if (wv == 0) {
            i2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i2);
            e_load = i2 != 0xffffffffu ? entry_of(i2) : none;
            // the next pass's copy orders into the other buffer (read last in the pass before this one's staging): wave 0 holds
            // the entry -- one DMA instruction, in flight beside this pass's staging and landed with it (the wait in front of B1)
            if (e_next.x != 0xffffffffu) fetch_orders(e_next, cur ^ 1u);
        }
        fetch_recs(e); // (every wave is past B2: the records of the pass before are read)
        stamp(0);
        // ---- staging: a wave's share of the copy orders, 16 bytes a lane, consecutive lanes consecutive pieces; nothing
        // waits between a wave's pieces: all of them are in flight at once ----
        __builtin_amdgcn_s_setprio(2); // a fresh pass's loads must not queue behind the DP of the older workgroups
        if (!(dbg & 2u)) {
            // (one LDS read for all of the pass's orders, a lane each; the wave's share comes out of it by v_readlane: no LDS
            // round trip per order in front of its copies)
            static_assert(kRT <= 64, "an order a lane");
            const uint4 o_all = (uint32_t)lane < n_ord ? rtab[cur * kRT + lane] : make_uint4(0u, 0u, 0u, 0u);
            for (uint32_t it = wv; it < n_ord; it += kWaves) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)o_all.x, (int)it), hi = (uint32_t)__builtin_amdgcn_readlane((int)o_all.y, (int)it);
                const long long off = (long long)((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)o_all.z, (int)it) |
                                                  ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)o_all.w, (int)it) << 32));
                const float4 *src = reinterpret_cast<const float4 *>(((it & 1u) ? a.ref : a.ev) + off);
                for (uint32_t q0 = lo; q0 < hi; q0 += 64u)
                    if (q0 + (uint32_t)lane < hi) dma16(src + q0 + lane, win + 4u * q0);
            }
        }
        stamp(1);
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): this wave's pieces have landed (the barrier below covers the others')
        __builtin_amdgcn_s_setprio(0);
        stamp(2);
        __syncthreads(); // B1: the image is staged; the next pass's entry is published
        stamp(3);
        run_dp(rec, n_jobs, (e.x + 1u) * kStreamTile);
        stamp(4);
        if (wv == 0) e_next = e_load; // (asked for at this pass's start)
        stamp(5);
        __syncthreads(); // B2: every wave is done with the image and with this pass's records
        stamp(6);
        cur ^= 1u;
    }
    if (stamps && lane < (int)kStamps && s_stamp[wv][lane]) atomicAdd(&a.cnt[kCntStamp0 + lane], s_stamp[wv][lane]);
}

// cells of a batch (reporting only; the walk costs as much as scoring the jobs): every part from its two anchors, plus the
// statistics of the tiles
__global__ __launch_bounds__(kT) void k_stream_cells(const StreamArgs a, unsigned long long *__restrict__ total)
{
    // thread = anchor index i; the part ending there exists iff anchor i + 1 belongs to the same chain: one 64-way search per
    // wave for its first anchor's chain, then a walk along the offsets
    const uint64_t i = (uint64_t)blockIdx.x * kT + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint64_t i_w = i - lane;
    unsigned long long cells = 0;
    if (i_w < a.n_anchors) {
        uint64_t c = find_chain(a.anchor_off, a.n_chains, i_w, lane);
        if (i < a.n_anchors) {
            while (a.anchor_off[c + 1] <= i) c++;
            if (i + 1 < a.anchor_off[c + 1]) {
                const rawdtw_anchor_t s = a.anchors[i + 1], e = a.anchors[i];
                const Part pt = classify(a, s, e);
                if (pt.asc) {
                    int r0 = (int)((float)pt.n * a.frac);
                    r0 = r0 > 1 ? r0 : 1;
                    cells = d_banded_cells(pt.n, pt.m, d_slanted_radius(pt.n, pt.m, r0));
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) cells += __shfl_down(cells, off);
    __shared__ unsigned long long s_c[kT / 64];
    if (lane == 0) s_c[threadIdx.x >> 6] = cells;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kT / 64; w++) t += s_c[w];
        if (t) atomicAdd(total, t);
    }
}

// the tiles' statistics summed into the counter block (reporting only)
__global__ __launch_bounds__(1024) void k_stream_stats(const StreamArgs a)
{
    unsigned long long t[3] = {0, 0, 0};
    const uint64_t n_units = (a.n_anchors + kScanUnit - 1) / kScanUnit;
    for (uint64_t u = threadIdx.x; u < n_units; u += 1024)
        for (int q = 0; q < 3; q++) t[q] += a.tile_stats[3 * u + q];
    __shared__ unsigned long long s_t[3];
    if (threadIdx.x < 3) s_t[threadIdx.x] = 0;
    __syncthreads();
    for (int q = 0; q < 3; q++) {
        for (int off = 32; off > 0; off >>= 1) t[q] += __shfl_down(t[q], off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_t[q], t[q]);
    }
    __syncthreads();
    if (threadIdx.x == 0) { a.cnt[kCntTileJobs] = s_t[0]; a.cnt[kCntTileBytes] = s_t[1]; a.cnt[kCntOtherBytes] = s_t[2]; }
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
// k_fold_select: the fold of align_chain (rmap.cpp:238-306) and the accept/cut loop of gen_chains (rmap.cpp:515-524) for
// kFoldReads consecutive reads a wave.  A sync-free batch keeps one cost per ANCHOR (part p of a chain at out[a1 - 2 - p]),
// and a read's chains are consecutive stretches of the anchor list.  The wave sweeps its reads' stretch from the top down
// in windows of kFoldWin floats: a window comes into LDS in whole 16-byte pieces (coalesced), then every lane folds the
// parts of its chain that lie in the window -- in the reference's order, one packed fp32 add a part (the fold is
// inherently sequential) -- and carries its sums into the next window.  Going down the addresses meets the chains last to
// first and every chain's parts first to last, so a lane with several chains (more than 64 chains a wave) finishes one
// before it meets the next, and a chain of any length is folded out of LDS.  The first lanes then run their reads'
// accept/cut loops over scores that never left the workgroup.
// (The lane-per-chain fold over the arena read 4 bytes a lane from 64 different lines per instruction and needed the chains
// sorted by length -- a one-workgroup sort on the scan's critical path; this one needs no order.)
// ---------------------------------------------------------------------------------------------------------------------
// (windows of 2 048 floats: 10 KB of LDS a workgroup -- a fold workgroup of the batch in front fits into the 13 KB five workgroups of
// k_runs leave on a CU; at 4 096 the launch alone is a quarter faster and the pipeline half a per cent slower, at 1 024 both are slower)
constexpr uint32_t kFoldReads = 8, kFoldWin = 2048, kFoldCap = 256;
__global__ __launch_bounds__(64) void k_fold_select(const StreamArgs a, const ChainDesc *__restrict__ chains, const uint64_t *__restrict__ chain_off,
                                                    const uint64_t n_reads, const float bonus, const int fused, const float min_score,
                                                    float *__restrict__ full_score, float *__restrict__ att_last, float *__restrict__ score,
                                                    uint8_t *__restrict__ keep)
{
    __shared__ __attribute__((aligned(16))) float s_win[kFoldWin + 4];
    __shared__ float s_full[kFoldCap], s_gate[kFoldCap];
    const uint32_t lane = threadIdx.x;
    // (a batch the scan or the DTW launch declined is redone through the job list: nothing here may be derived from it)
    const bool declined = a.cnt[kCntBad] != ~0ull || a.cnt[kCntOverflow] != ~0ull || a.cnt[kCntUnsupported] != 0ull || a.cnt[kCntOthers] > a.others_cap;
    const uint64_t r0 = (uint64_t)blockIdx.x * kFoldReads, r1 = min(r0 + (uint64_t)kFoldReads, n_reads);
    const uint32_t nr = (uint32_t)(r1 - r0);
    const uint64_t co = chain_off[r0 + min(lane, nr)], co1 = chain_off[r0 + min(lane + 1u, nr)]; // lane < nr: its read's chains
    if (declined) return;
    const uint64_t cA = __shfl((unsigned long long)co, 0), cB = __shfl((unsigned long long)co, (int)nr);
    if (cB <= cA) return;
    const uint64_t aA = a.anchor_off[cA], aB = a.anchor_off[cB];
    // the lane's chains: cA + lane + 64 k, the highest first
    const uint64_t n_ch = cB - cA;
    long long c = lane < n_ch ? (long long)(cA + lane + ((n_ch - 1u - lane) & ~63ull)) : -1; // (-1: none)
    ChainDesc d = chains[c >= 0 ? (uint64_t)c : cA];
    uint32_t p = 0; // parts of chain c folded so far
    v2f acc = {0.0f, (float)d.span * bonus}; // {cost, attainable}  (rmap.cpp:205,246)
    auto finish = [&](const float last) {
        const float gate = d.n_jobs ? acc.y : __builtin_inff(); // tested before the last (or only) DTW call; none: no check
        const float cost = d.n_jobs ? acc.x + last : acc.x;    // the last part only adds to the cost (rmap.cpp:279-280)
        float sc;
        if (fused) sc = __builtin_fmaf((float)d.num_aligned, bonus, -cost);
        else { const float prod = (float)d.num_aligned * bonus; sc = prod - cost; }
        full_score[c] = sc; att_last[c] = gate;
        if ((uint64_t)c - cA < kFoldCap) { s_full[(uint64_t)c - cA] = sc; s_gate[(uint64_t)c - cA] = gate; }
        c = (uint64_t)c >= cA + 64u ? c - 64 : -1;
        if (c >= 0) { d = chains[c]; p = 0; acc = v2f{0.0f, (float)d.span * bonus}; }
    };
    for (uint64_t hi = aB; hi > aA;) { // (uniform)
        const uint64_t lo4 = (hi > aA + kFoldWin ? hi - kFoldWin : aA) & ~3ull;
        // LDS-DMA: 16 bytes a lane straight into LDS, every piece of the window in flight at once (the last piece of the
        // list may reach past its end: element by element)
        for (uint64_t i0 = lo4; i0 < hi; i0 += 256ull) { // (uniform)
            const uint64_t i = i0 + 4ull * lane;
            if (i < hi) {
                if (i + 4 <= a.n_anchors)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.out + i),
                                                     (__attribute__((address_space(3))) void *)(s_win + (i0 - lo4)), 16, 0, 0);
                else
                    for (uint64_t j = i; j < a.n_anchors; j++) s_win[j - lo4] = a.out[j];
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
        __syncthreads();
        while (c >= 0) {
            if (d.n_jobs == 0) { finish(0.0f); continue; }
            const uint64_t idx = d.job_first - p; // the next part: out[job_first - p]
            if (idx < lo4) break;                 // the rest lies in the windows below
            const float *w = s_win + (idx - lo4);
            const uint32_t body = d.n_jobs - 1u;
            // parts of the body in this window: cost += sub (rmap.cpp:279), attainable -= sub (rmap.cpp:280), one packed add a part
            const uint32_t nb = p < body ? (uint32_t)min<uint64_t>(body - p, idx - lo4 + 1ull) : 0u;
            // The adds are two dependent chains (cost up, attainable down: 4 cycles an add, interleaved they never wait for
            // each other); the parts come out of LDS sixteen at a time in 16-byte pieces, the next sixteen on their way
            // while these are added.  Part q of this stretch is w[-q]: single parts until a piece ends on one.
            uint32_t q = 0;
            float cost = acc.x, att = acc.y;
            const uint32_t j0 = (uint32_t)(idx - lo4); // w = s_win + j0
            for (; q < nb && ((j0 - q) & 3u) != 3u; q++) { const float x = w[-(int)q]; cost += x; att -= x; }
            if (q + 16u <= nb) {
                float4 x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) x[u] = *reinterpret_cast<const float4 *>(w - (int)(q + 4u * u) - 3);
#pragma unroll 2
                for (; q + 32u <= nb; q += 16u) {
                    float4 y[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) y[u] = *reinterpret_cast<const float4 *>(w - (int)(q + 16u + 4u * u) - 3);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        cost += x[u].w; att -= x[u].w; cost += x[u].z; att -= x[u].z;
                        cost += x[u].y; att -= x[u].y; cost += x[u].x; att -= x[u].x;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) x[u] = y[u];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    cost += x[u].w; att -= x[u].w; cost += x[u].z; att -= x[u].z;
                    cost += x[u].y; att -= x[u].y; cost += x[u].x; att -= x[u].x;
                }
                q += 16u;
            }
            for (; q < nb; q++) { const float x = w[-(int)q]; cost += x; att -= x; }
            acc = v2f{cost, att};
            p += nb;
            if (p == body && idx - nb >= lo4 && idx >= nb) finish(w[-(int)nb]); // the last part is in this window too
            else break;
        }
        __syncthreads();
        hi = lo4;
    }
    if (lane < nr) {
        float best = 0.0f; // rmap.cpp:515
        for (uint64_t cc = co; cc < co1; cc++) {
            float f, g;
            if (cc - cA < kFoldCap) { f = s_full[cc - cA]; g = s_gate[cc - cA]; }
            else { // (written by another lane of this wave: read past the vector cache)
                f = __hip_atomic_load(&full_score[cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g = __hip_atomic_load(&att_last[cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const float sv = (g < best) ? -1e10f : f; // rmap.cpp:206-209, 265-268
            const bool k = sv >= min_score;           // rmap.cpp:518
            if (k && sv > best) best = sv;            // rmap.cpp:519-521
            score[cc] = sv;
            keep[cc] = k ? 1 : 0;
        }
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
    (void)threads;
    return lds_floats * 4u + kStreamTile * 8u + 2u * 2u * kStreamMaxSeg * 16u; // image, the pass's records, two passes' copy orders
}

// everything rawdtw_batch_create enqueues for a sparse + banded batch: the scan of the anchor list (side list, checks,
// the tiles' first chains), the fold's chain records and order, the side list's class order
hipError_t stream_plan(const StreamArgs &a, ChainDesc *d_chains, uint32_t *d_fold_order, hipStream_t s)
{
    (void)hipGetLastError();
    if (a.n_anchors == 0 && a.n_chains == 0) return hipSuccess;
    const uint32_t n_units = (uint32_t)((a.n_anchors + kScanUnit - 1) / kScanUnit), n_desc = (uint32_t)((a.n_chains + kScanT - 1) / kScanT);
    if (a.steps) {
        hipLaunchKernelGGL(k_scan_compact, dim3(n_units + 1u), dim3(kScanTC), 0, s, a, d_fold_order);
        if (n_desc) hipLaunchKernelGGL(k_scan_desc, dim3(n_desc), dim3(kScanT), 0, s, a, d_chains);
    } else hipLaunchKernelGGL(k_scan, dim3(n_units + 1u + n_desc), dim3(kScanT), 0, s, a, d_chains, d_fold_order);
    if (a.n_tiles) hipLaunchKernelGGL(k_side, dim3(kSideGroups), dim3(1024), 0, s, a);
    return hipGetLastError();
}

// a chunk round's last launch in front of the fold: every chain's costs in full (the stretch taken over + the new parts')
hipError_t stream_gather(const StreamArgs &a, ChainDesc *, hipStream_t s)
{
    if (!a.carry || a.n_chains == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_gather, dim3((uint32_t)((a.n_chains + kGatherT / 64 - 1) / (kGatherT / 64))), dim3(kGatherT), 0, s, a);
    return hipGetLastError();
}

// ... and the passes of the DTW launch (behind the scan: the side list's launch can start beside this one)
hipError_t stream_plan_passes(const StreamArgs &a, hipStream_t s)
{
    if (a.n_tiles == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_plan, dim3((a.n_tiles + kPlanT / 64 - 1) / (kPlanT / 64)), dim3(kPlanT), 0, s, a);
    return hipGetLastError();
}

// the side list's launch (beside the tiles': its own stream)
hipError_t stream_wide(const StreamArgs &a, uint32_t blocks, hipStream_t s)
{
    if (a.n_tiles == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_wide, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// fold + select of a sync-free batch in one launch (`chains`: the scan's records)
hipError_t stream_fold_select(const StreamArgs &a, const ChainDesc *chains, const uint64_t *chain_off, uint64_t n_reads, float bonus, int fused,
                              float min_score, float *full_score, float *att_last, float *score, uint8_t *keep, hipStream_t s)
{
    if (n_reads == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_fold_select, dim3((uint32_t)((n_reads + kFoldReads - 1) / kFoldReads)), dim3(64), 0, s, a, chains, chain_off, n_reads, bonus,
                       fused, min_score, full_score, att_last, score, keep);
    return hipGetLastError();
}

// reset_queue: the batch has run before (the tile queue's heads start at zero with the counters' initial values)
hipError_t stream_run(const StreamArgs &a, uint32_t blocks, uint32_t lds_floats, int threads, bool reset_queue, hipStream_t s)
{
    if (a.n_tiles == 0) return hipSuccess;
    (void)hipGetLastError();
    hipError_t e = reset_queue ? hipMemsetAsync(&a.cnt[kCntHeads], 0, 8 * 16 * sizeof(unsigned long long), s) : hipSuccess;
    if (e != hipSuccess) return e;
    const uint32_t lds_bytes = stream_lds_bytes_t(lds_floats, threads);
    const bool diag = a.debug != 0;
    auto launch = [&](auto kern, int tt) -> hipError_t {
        if (lds_bytes > 64 * 1024) {
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e2 != hipSuccess) return e2;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(tt), lds_bytes, s, a, lds_floats);
        return hipGetLastError();
    };
    if (threads == 512) return diag ? launch(k_runs<512, true>, 512) : launch(k_runs<512, false>, 512);
    return diag ? launch(k_runs<256, true>, 256) : launch(k_runs<256, false>, 256);
}

// workgroups of k_runs one compute unit holds at this LDS size (for the persistent grid)
int stream_blocks_per_cu(uint32_t lds_floats, int threads)
{
    int n = 0;
    const uint32_t lds_bytes = stream_lds_bytes_t(lds_floats, threads);
    const void *fn = threads == 512 ? reinterpret_cast<const void *>(k_runs<512, false>) : reinterpret_cast<const void *>(k_runs<256, false>);
    if (lds_bytes > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds_bytes) != hipSuccess) return 0;
    return n;
}

hipError_t stream_count_cells(const StreamArgs &a, unsigned long long *d_total, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || a.n_anchors == 0) return e;
    hipLaunchKernelGGL(k_stream_cells, dim3(blocks_for(a.n_anchors)), dim3(kT), 0, s, a, d_total);
    return hipGetLastError();
}

hipError_t stream_sum_stats(const StreamArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_stream_stats, dim3(1), dim3(1024), 0, s, a);
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
