// rawdtw_chain.hip -- the anchor sort and the chaining DP of gen_chains on the device: what bounds the chunk-round
// mapper once the DTW block is on the GPU (rawdtw_mapper.cpp: 8 us a read and host thread, 60 % of a round on 16 threads).
//
//   src/rmap.cpp:396-401   every (sequence, strand) list of anchors sorted by (target, query)
//   src/rmap.cpp:430-507   the chaining DP: per anchor the best predecessor inside the band, the skip counter, the running maximum, the end
//                          candidates (score filter), the num_best_chains best ends
//   src/rmap.cpp:130-173   traceback_chains: predecessor walk with the `used` marks, the score of a chain that runs into a used anchor
//   src/rmap.cpp:512       the evaluation order: chaining score, descending
// as restated on the host by rawdtw_chain_anchors / rawdtw_sort_by_chaining_score (rawdtw_host.cpp), which the tests compare this with,
// chain by chain and bit for bit.
//
// A WAVE A READ.  The read's seeds (the previous chains' anchors and the chunk's hits, unsorted, as the caller has them) go into LDS and are
// sorted there (bitonic, key = (sequence * 2 + strand, target, query): equal seeds are indistinguishable, so any sort gives the array the
// reference's std::sort gives).  The DP runs anchor by anchor -- that order is the algorithm's -- with the predecessor loop of an anchor, the
// part that is long, taken 64 candidates at a time: whether a candidate is passed over, ends the loop or competes, and with which value,
// depends on the anchors and on scores that are final; what is sequential in the source -- `best` only ever grows, the skip counter moves
// by one a competing candidate, the loop ends at the first candidate past max_num_skips -- is a prefix maximum and a prefix sum over the
// lanes, and the loop's exit is the first lane whose prefix says so.  Ends, traceback and order are short and run on one lane.
// Results: per read its chains in evaluation order (records + anchors, end-first); a scan and a compaction launch lay all reads' chains out
// as ONE candidate batch in device memory -- chain_off / anchor_off / anchors / ref_base / read_base, exactly what rawdtw_batch_submit
// takes -- so the anchors never cross PCIe on their way into the DTW.
//
// What the device declines (the caller chains that round on the host, same results): a read with more seeds than the wave's piece of LDS
// holds (2 048), more than 32 chains, or more than 16 chains with two equal scores among them (std::sort's order of equal elements is its
// own beyond 16; up to 16 it is an insertion sort and stable).
#include "rawdtw_capi.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace rawdtw {
namespace {

constexpr uint32_t kChainMaxSeeds = 2048; // a read's seeds in LDS: 21 bytes each
constexpr uint32_t kChainCap = 32;        // chains a read
constexpr uint32_t kChainStable = 16;     // std::sort is an insertion sort up to here (libstdc++'s _S_threshold)

struct ChainRecDev { float score; uint32_t key, start, end, n, a_off; };
struct ChainCnt { uint32_t nc, na, flags, pad; };

struct ChainArgs {
    const uint64_t *seed_off;
    const rawdtw_seed_t *seeds;
    uint32_t n_reads, n2;
    rawdtw_chain_opt_t opt;
    rawdtw_anchor_t *tmp_anchors; // read r's chains, as generated: [seed_off[r], seed_off[r + 1])
    ChainRecDev *tmp_recs;        // [r * kChainCap ..): in evaluation order
    ChainCnt *cnt;
};

__device__ __forceinline__ void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// inclusive scans over the wave in lane order (rows of 16 by row_shr, then the rows' last lanes into the rows behind them)
__device__ __forceinline__ int scan_add(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}
// (values are >= -1: as floats; a lane without a source takes -1)
__device__ __forceinline__ float scan_max(float v)
{
    const int neg = __builtin_bit_cast(int, -1.0f);
#define RAWDTW_SM(ctrl, rm) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(neg, __builtin_bit_cast(int, v), ctrl, rm, 0xf, false)))
    RAWDTW_SM(0x111, 0xf); RAWDTW_SM(0x112, 0xf); RAWDTW_SM(0x114, 0xf); RAWDTW_SM(0x118, 0xf); RAWDTW_SM(0x142, 0xa); RAWDTW_SM(0x143, 0xc);
#undef RAWDTW_SM
    return v;
}
__device__ __forceinline__ uint32_t uni(const uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t lane_of(const uint32_t x, const uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)uni(l)); }

__global__ __launch_bounds__(64) void k_chain(const ChainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t n2 = a.n2, lane = threadIdx.x, r = blockIdx.x;
    unsigned long long *K1 = reinterpret_cast<unsigned long long *>(smem); // key << 32 | target
    uint32_t *Q = reinterpret_cast<uint32_t *>(K1 + n2);
    float *SC = reinterpret_cast<float *>(Q + n2);
    uint32_t *PR = reinterpret_cast<uint32_t *>(SC + n2);
    unsigned char *FL = reinterpret_cast<unsigned char *>(PR + n2); // 1 used, 2 end candidate, 4 taken
    __shared__ ChainRecDev s_rec[kChainCap];
    __shared__ uint32_t s_perm[kChainCap];
    const uint64_t s0 = a.seed_off[r];
    const uint32_t n = (uint32_t)(a.seed_off[r + 1] - s0);
    if (n == 0 || n > n2) { // (no seeds: no chains.  Too many: declined -- the host sizes n2 by the round's largest read, so only past the cap)
        if (lane == 0) a.cnt[r] = ChainCnt{0u, 0u, n > n2 ? 1u : 0u, 0u};
        return;
    }
    for (uint32_t i = lane; i < n2; i += 64) {
        if (i < n) { const rawdtw_seed_t s = a.seeds[s0 + i]; K1[i] = ((unsigned long long)s.key << 32) | s.target_position; Q[i] = s.query_position; }
        else { K1[i] = ~0ull; Q[i] = ~0u; }
        FL[i] = 0;
    }
    lds_sync();
    // ---- rmap.cpp:396-401: ascending by (key, target, query) ----
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < n2 / 2; t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j; // the pair's lower and upper element
                const unsigned long long ka = K1[i], kb = K1[p];
                const uint32_t qa = Q[i], qb = Q[p];
                const bool gt = ka > kb || (ka == kb && qa > qb);
                if (gt == ((i & k) == 0)) { K1[i] = kb; K1[p] = ka; Q[i] = qb; Q[p] = qa; }
            }
            lds_sync();
        }
    const rawdtw_chain_opt_t &o = a.opt;
    const float e_f = (float)o.e;
    float maxs = 0.0f;
    uint32_t nc = 0, na = 0, flags = 0;
    for (uint32_t g0 = 0; g0 < n;) {
        // the (sequence, strand) list [g0, g1)
        const uint32_t key = (uint32_t)(K1[g0] >> 32);
        uint32_t g1 = n;
        for (uint32_t i = g0; i < n; i += 64) {
            const uint32_t x = i + lane;
            const unsigned long long diff = __ballot(x < n && (uint32_t)(K1[x < n ? x : g0] >> 32) != key);
            if (diff) { g1 = i + (uint32_t)__builtin_ctzll(diff); break; }
        }
        // ---- rmap.cpp:436-494 ----
        for (uint32_t ai = g0; ai < g1; ai++) {
            const int32_t ct = (int32_t)(uint32_t)K1[ai], cq = (int32_t)Q[ai];
            float best = e_f;
            uint32_t pred = ai;
            int32_t skips = 0;
            const int32_t lo = (ai - g0 > (uint32_t)o.chaining_band_length) ? (int32_t)ai - o.chaining_band_length : (int32_t)g0;
            for (int32_t base = (int32_t)ai - 1; base >= lo; base -= 64) {
                const int32_t pi = base - (int32_t)lane;
                const bool valid = pi >= lo;
                const int32_t pt = (int32_t)(uint32_t)K1[valid ? pi : lo], pq = (int32_t)Q[valid ? pi : lo];
                const float sp = SC[valid ? pi : lo];
                const bool pass12 = pq == cq || pt == ct;                                   // rmap.cpp:458-459
                const bool stop_gap = valid && !pass12 && pt + o.max_target_gap_length < ct; // rmap.cpp:460
                const int32_t td = ct - pt, qd = cq - pq;
                const bool active = valid && !pass12 && !stop_gap && qd >= 0;               // rmap.cpp:467
                float cur = 0.0f;
                {
                    const float matching = (float)min(min(td, qd), o.e);                    // rmap.cpp:469
                    const int gap = abs(td - qd);
                    const float scale = td > 0 ? __fdiv_rn((float)qd, (float)td) : 1.0f;
                    if (gap < o.max_gap_length && scale < 5.0f && scale > 0.75f) cur = sp + matching; // rmap.cpp:474-476
                }
                const float cv = active ? cur : -1.0f;
                // the candidates before this one, in loop order = lane order: their largest value, and `best` as it stood at the loop's entry
                const float incl = scan_max(cv);
                const float before = fmaxf(best, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1.0f), __builtin_bit_cast(int, incl), 0x138, 0xf, 0xf, false)));
                const bool improver = active && cur > before;                               // rmap.cpp:478
                const int32_t moves = scan_add(improver ? -1 : (active ? 1 : 0));
                const bool stop_skip = active && !improver && skips + moves > o.max_num_skips; // rmap.cpp:482-484
                const unsigned long long stop = __ballot(!valid || stop_gap || stop_skip);
                const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                const unsigned long long live = first >= 64u ? ~0ull : ((1ull << first) - 1ull);
                const unsigned long long imp = __ballot(improver) & live;
                if (imp) { // (the improvers' values ascend: the last one stands)
                    const uint32_t last = 63u - (uint32_t)__builtin_clzll(imp);
                    best = __builtin_bit_cast(float, lane_of(__builtin_bit_cast(uint32_t, cur), last));
                    pred = (uint32_t)(base - (int32_t)last);
                }
                if (first > 0) skips += (int32_t)lane_of((uint32_t)moves, first - 1u);
                if (first < 64u) break;
            }
            if (best > maxs) maxs = best;                                                   // rmap.cpp:486-488
            const bool is_end = o.disable_score_filtering || (best >= o.min_chaining_score && best > maxs / 2); // rmap.cpp:489-493
            if (lane == 0) { SC[ai] = best; PR[ai] = pred; FL[ai] = is_end ? 2 : 0; }
            lds_sync();
        }
        // ---- the num_best_chains best ends (rmap.cpp:175-179: score descending, then index descending), traceback_chains ----
        for (int k = 0; k < o.num_best_chains; k++) {
            unsigned long long top = 0;
            for (uint32_t i = g0 + lane; i < g1; i += 64)
                if ((FL[i] & 6) == 2) top = max(top, (1ull << 63) | ((unsigned long long)__builtin_bit_cast(uint32_t, SC[i]) << 32) | i); // (scores are positive: their bits ascend with them)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(top >> 32), d), lw = (uint32_t)__shfl_xor((int)(uint32_t)top, d);
                top = max(top, ((unsigned long long)hi << 32) | lw);
            }
            if (!(top >> 63)) break;
            const uint32_t end = (uint32_t)top;
            bool below = false;
            if (lane == 0) {
                FL[end] |= 4;
                if (!(FL[end] & 1)) {
                    const uint64_t out0 = s0 + na;
                    uint32_t cur = end, len = 1;
                    bool stop_at_used = false;
                    a.tmp_anchors[out0] = rawdtw_anchor_t{(uint32_t)K1[cur], Q[cur]};
                    if (PR[cur] != cur && (FL[PR[cur]] & 1)) stop_at_used = true;
                    FL[cur] |= 1;
                    while (PR[cur] != cur && !(FL[PR[cur]] & 1)) {
                        cur = PR[cur];
                        a.tmp_anchors[out0 + len] = rawdtw_anchor_t{(uint32_t)K1[cur], Q[cur]};
                        len++;
                        if (PR[cur] != cur && (FL[PR[cur]] & 1)) stop_at_used = true;
                        FL[cur] |= 1;
                    }
                    if (len >= (uint32_t)o.min_num_anchors) {
                        float adj = SC[end];
                        if (stop_at_used) adj -= SC[PR[cur]];
                        if (nc < kChainCap) s_rec[nc] = ChainRecDev{adj, key, (uint32_t)K1[cur], (uint32_t)K1[end], len, na};
                        else flags |= 2u;
                        nc++; na += len;
                    }
                }
                below = !o.disable_score_filtering && SC[end] < maxs / 2;                  // rmap.cpp:502-504
            }
            nc = uni(nc); na = uni(na); flags = uni(flags);
            lds_sync();
            if (uni(below ? 1u : 0u)) break;
        }
        g0 = g1;
    }
    // ---- rmap.cpp:512: by chaining score, descending; equal scores keep their order (std::sort up to 16 elements) ----
    if (lane == 0) {
        const uint32_t m = min(nc, kChainCap);
        bool ties = false;
        for (uint32_t i = 0; i < m; i++) {
            const float v = s_rec[i].score;
            uint32_t j = i;
            while (j > 0 && v > s_rec[s_perm[j - 1]].score) { s_perm[j] = s_perm[j - 1]; j--; }
            if (j > 0 && v == s_rec[s_perm[j - 1]].score) ties = true;
            s_perm[j] = i;
        }
        if (nc > kChainStable && ties) flags |= 4u;
        for (uint32_t i = 0; i < m; i++) a.tmp_recs[(uint64_t)r * kChainCap + i] = s_rec[s_perm[i]];
        a.cnt[r] = ChainCnt{nc, na, flags, 0u};
    }
}

// the reads' chains and anchors before each read; the round's totals and its flags
__global__ __launch_bounds__(1024) void k_chain_scan(const ChainCnt *__restrict__ cnt, const uint32_t n_reads, uint64_t *__restrict__ chain_off,
                                                     uint64_t *__restrict__ read_anchor0, uint64_t *__restrict__ totals /* chains, anchors, flags */)
{
    __shared__ uint64_t s_c[1024], s_a[1024];
    __shared__ uint32_t s_f[1024];
    const uint32_t t = threadIdx.x, per = (n_reads + 1023u) / 1024u, lo = min(n_reads, t * per), hi = min(n_reads, lo + per);
    uint64_t c = 0, x = 0;
    uint32_t f = 0;
    for (uint32_t r = lo; r < hi; r++) { c += min(cnt[r].nc, kChainCap); x += cnt[r].na; f |= cnt[r].flags; }
    s_c[t] = c; s_a[t] = x; s_f[t] = f;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint64_t pc = t >= d ? s_c[t - d] : 0, pa = t >= d ? s_a[t - d] : 0;
        const uint32_t pf = t >= d ? s_f[t - d] : 0;
        __syncthreads();
        s_c[t] += pc; s_a[t] += pa; s_f[t] |= pf;
        __syncthreads();
    }
    uint64_t bc = s_c[t] - c, ba = s_a[t] - x;
    for (uint32_t r = lo; r < hi; r++) { chain_off[r] = bc; read_anchor0[r] = ba; bc += min(cnt[r].nc, kChainCap); ba += cnt[r].na; }
    if (t == 1023) { chain_off[n_reads] = s_c[t]; totals[0] = s_c[t]; totals[1] = s_a[t]; totals[2] = s_f[t]; totals[3] = 0; }
}

// a wave a read: its chains, in evaluation order, into the batch's arrays
__global__ __launch_bounds__(64) void k_chain_compact(const ChainArgs a, const uint64_t *__restrict__ chain_off, const uint64_t *__restrict__ read_anchor0,
                                                      const uint32_t *__restrict__ read_base, const uint64_t *__restrict__ key_base, const uint32_t n_keys,
                                                      uint64_t *__restrict__ anchor_off, rawdtw_anchor_t *__restrict__ anchors, uint64_t *__restrict__ ref_base,
                                                      uint32_t *__restrict__ read_base_c, rawdtw_chain_rec_t *__restrict__ recs, uint64_t *__restrict__ totals,
                                                      // the caller's page-locked host arrays, written from here (null: they are copied afterwards)
                                                      uint64_t *__restrict__ h_anchor_off, rawdtw_chain_rec_t *__restrict__ h_recs, rawdtw_anchor_t *__restrict__ h_anchors,
                                                      const uint64_t h_chains_cap)
{
    const uint32_t r = blockIdx.x, lane = threadIdx.x;
    const uint32_t nc = min(a.cnt[r].nc, kChainCap);
    const uint64_t c0 = chain_off[r], s0 = a.seed_off[r];
    const bool host = h_anchor_off && totals[2] == 0 && totals[0] <= h_chains_cap; // (a declined round leaves the host arrays alone)
    uint64_t dst = read_anchor0[r];
    for (uint32_t i = 0; i < nc; i++) {
        const ChainRecDev rec = a.tmp_recs[(uint64_t)r * kChainCap + i];
        for (uint32_t k = lane; k < rec.n; k += 64) {
            const rawdtw_anchor_t v = a.tmp_anchors[s0 + rec.a_off + k];
            anchors[dst + k] = v;
            if (host && h_anchors) h_anchors[dst + k] = v;
        }
        if (lane == 0) {
            const rawdtw_chain_rec_t out{rec.score, rec.key, rec.start, rec.end, rec.n};
            anchor_off[c0 + i] = dst;
            ref_base[c0 + i] = rec.key < n_keys ? key_base[rec.key] : 0ull;
            if (rec.key >= n_keys) atomicOr(reinterpret_cast<unsigned long long *>(totals + 3), 1ull); // (a seed on a key the caller gave no base for)
            read_base_c[c0 + i] = read_base[r];
            recs[c0 + i] = out;
            if (host) { h_anchor_off[c0 + i] = dst; h_recs[c0 + i] = out; }
        }
        dst += rec.n;
    }
    if (r == 0 && lane == 0) { anchor_off[totals[0]] = totals[1]; if (host) h_anchor_off[totals[0]] = totals[1]; }
}

struct ChainWs {
    void *dev = nullptr;
    size_t dev_bytes = 0;
    void *pin = nullptr; // totals
    hipEvent_t done = nullptr; // behind a round's last copy: what rawdtw_chain_round_end waits for (not for what the caller enqueued behind the round)
    // a round begun and not ended
    bool pending = false, direct = false;
    uint64_t n_reads = 0, chains_cap = 0;
    uint64_t *h_anchor_off = nullptr;
    rawdtw_chain_rec_t *h_recs = nullptr;
    rawdtw_anchor_t *h_anchors = nullptr;
    const uint64_t *d_aoff = nullptr;
    const rawdtw_chain_rec_t *d_recs = nullptr;
    const rawdtw_anchor_t *d_anch = nullptr;
    const uint64_t *d_refb = nullptr;
    const uint32_t *d_rbc = nullptr;
};

} // namespace
} // namespace rawdtw

using namespace rawdtw;
using namespace rawdtw::capi;

struct rawdtw_chain_ws { ChainWs w; };

extern "C" {

int rawdtw_chain_round_begin(rawdtw_ctx *ctx, const rawdtw_chain_opt_t *opt, uint64_t n_reads, const uint64_t *seed_off, const rawdtw_seed_t *seeds,
                             const uint32_t *read_base, uint32_t n_keys, const uint64_t *key_base, uint64_t *chain_off, uint64_t *anchor_off,
                             rawdtw_chain_rec_t *recs, uint64_t chains_cap, rawdtw_anchor_t *anchors)
{
    if (!ctx) return RAWDTW_ERR_INVALID;
    if (!opt || !seed_off || (!seeds && n_reads && seed_off[n_reads]) || !read_base || (!key_base && n_keys) || !chain_off || !anchor_off || !recs)
        return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    if (ctx->chain_ws && ctx->chain_ws->w.pending) return fail(ctx, RAWDTW_ERR_INVALID, "a chaining round is begun on this context and not ended");
    if (n_reads == 0 || n_reads > 0xffffffffull) return fail(ctx, RAWDTW_ERR_INVALID, "no reads, or more than 2^32");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n_seeds = seed_off[n_reads];
    uint32_t most = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        if (seed_off[r + 1] < seed_off[r]) return fail(ctx, RAWDTW_ERR_INVALID, "offsets do not ascend");
        most = (uint32_t)std::max<uint64_t>(most, std::min<uint64_t>(seed_off[r + 1] - seed_off[r], 0xffffffffull));
    }
    uint32_t seed_cap = kChainMaxSeeds; // (tests: RAWDTW_CHAIN_MAX_SEEDS lowers the cap, so that small rounds take the declined path)
    if (const char *e = getenv("RAWDTW_CHAIN_MAX_SEEDS")) seed_cap = std::min<uint32_t>(kChainMaxSeeds, (uint32_t)std::max(1l, strtol(e, nullptr, 10)));
    if (most > seed_cap) return fail(ctx, RAWDTW_ERR_UNSUPPORTED, "a read has more seeds than the device chains (2048): chain this round on the host");
    uint32_t n2 = 64;
    while (n2 < most) n2 <<= 1;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // one block of device memory, grow-only: inputs, per-read scratch, the batch's arrays
    const size_t b_soff = al((n_reads + 1) * 8), b_seeds = al((size_t)n_seeds * sizeof(rawdtw_seed_t) + 16), b_rb = al(n_reads * 4), b_kb = al((size_t)n_keys * 8 + 8),
                 b_tmpa = al((size_t)n_seeds * 8 + 16), b_trec = al(n_reads * kChainCap * sizeof(ChainRecDev)), b_cnt = al(n_reads * sizeof(ChainCnt)),
                 b_coff = al((n_reads + 1) * 8), b_ra0 = al(n_reads * 8), b_tot = al(32), b_aoff = al((n_reads * kChainCap + 1) * 8), b_anch = al((size_t)n_seeds * 8 + 16),
                 b_refb = al(n_reads * kChainCap * 8 + 8), b_rbc = al(n_reads * kChainCap * 4 + 8), b_recs = al(n_reads * kChainCap * sizeof(rawdtw_chain_rec_t) + 8);
    const size_t need = b_soff + b_seeds + b_rb + b_kb + b_tmpa + b_trec + b_cnt + b_coff + b_ra0 + b_tot + b_aoff + b_anch + b_refb + b_rbc + b_recs;
    if (!ctx->chain_ws) ctx->chain_ws = new (std::nothrow) rawdtw_chain_ws;
    if (!ctx->chain_ws) return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed");
    ChainWs &w = ctx->chain_ws->w;
    if (w.dev_bytes < need) {
        if (w.dev) (void)hipFree(w.dev);
        w.dev = nullptr; w.dev_bytes = 0;
        const size_t want = need + need / 4;
        if (hipMalloc(&w.dev, want) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, RAWDTW_ERR_OOM, "chaining workspace allocation failed"); }
        w.dev_bytes = want;
    }
    if (!w.pin && hipHostMalloc(&w.pin, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); w.pin = nullptr; return fail(ctx, RAWDTW_ERR_OOM, "pinned allocation failed"); }
    char *p = static_cast<char *>(w.dev);
    uint64_t *d_soff = reinterpret_cast<uint64_t *>(p); p += b_soff;
    rawdtw_seed_t *d_seeds = reinterpret_cast<rawdtw_seed_t *>(p); p += b_seeds;
    uint32_t *d_rb = reinterpret_cast<uint32_t *>(p); p += b_rb;
    uint64_t *d_kb = reinterpret_cast<uint64_t *>(p); p += b_kb;
    rawdtw_anchor_t *d_tmpa = reinterpret_cast<rawdtw_anchor_t *>(p); p += b_tmpa;
    ChainRecDev *d_trec = reinterpret_cast<ChainRecDev *>(p); p += b_trec;
    ChainCnt *d_cnt = reinterpret_cast<ChainCnt *>(p); p += b_cnt;
    uint64_t *d_coff = reinterpret_cast<uint64_t *>(p); p += b_coff;
    uint64_t *d_ra0 = reinterpret_cast<uint64_t *>(p); p += b_ra0;
    uint64_t *d_tot = reinterpret_cast<uint64_t *>(p); p += b_tot;
    uint64_t *d_aoff = reinterpret_cast<uint64_t *>(p); p += b_aoff;
    rawdtw_anchor_t *d_anch = reinterpret_cast<rawdtw_anchor_t *>(p); p += b_anch;
    uint64_t *d_refb = reinterpret_cast<uint64_t *>(p); p += b_refb;
    uint32_t *d_rbc = reinterpret_cast<uint32_t *>(p); p += b_rbc;
    rawdtw_chain_rec_t *d_recs = reinterpret_cast<rawdtw_chain_rec_t *>(p);
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(d_soff, seed_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, s));
    if (n_seeds) HIP_TRY(ctx, hipMemcpyAsync(d_seeds, seeds, (size_t)n_seeds * sizeof(rawdtw_seed_t), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d_rb, read_base, n_reads * 4, hipMemcpyHostToDevice, s));
    if (n_keys) HIP_TRY(ctx, hipMemcpyAsync(d_kb, key_base, (size_t)n_keys * 8, hipMemcpyHostToDevice, s));
    ChainArgs a{d_soff, d_seeds, (uint32_t)n_reads, n2, *opt, d_tmpa, d_trec, d_cnt};
    const size_t lds = (size_t)n2 * 21 + 16;
    hipLaunchKernelGGL(k_chain, dim3((uint32_t)n_reads), dim3(64), lds, s, a);
    hipLaunchKernelGGL(k_chain_scan, dim3(1), dim3(1024), 0, s, d_cnt, (uint32_t)n_reads, d_coff, d_ra0, d_tot);
    // the caller's arrays: written by the compaction launch itself when they are page-locked (rawdtw_host_alloc) -- no copy command, no second
    // wait for sizes only the device knows; else copied at the round's end
    auto page_locked = [](const void *q) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    };
    const bool direct = page_locked(anchor_off) && page_locked(recs) && (!anchors || page_locked(anchors)) && page_locked(chain_off);
    hipLaunchKernelGGL(k_chain_compact, dim3((uint32_t)n_reads), dim3(64), 0, s, a, d_coff, d_ra0, d_rb, d_kb, n_keys, d_aoff, d_anch, d_refb, d_rbc, d_recs, d_tot,
                       direct ? anchor_off : nullptr, direct ? recs : nullptr, direct ? anchors : nullptr, chains_cap);
    HIP_TRY(ctx, hipGetLastError());
    uint64_t *h_tot = static_cast<uint64_t *>(w.pin);
    HIP_TRY(ctx, hipMemcpyAsync(h_tot, d_tot, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(chain_off, d_coff, (n_reads + 1) * 8, hipMemcpyDeviceToHost, s));
    if (!w.done) HIP_TRY(ctx, hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(w.done, s));
    w.pending = true; w.direct = direct; w.n_reads = n_reads; w.chains_cap = chains_cap;
    w.h_anchor_off = anchor_off; w.h_recs = recs; w.h_anchors = anchors;
    w.d_aoff = d_aoff; w.d_recs = d_recs; w.d_anch = d_anch; w.d_refb = d_refb; w.d_rbc = d_rbc;
    return RAWDTW_OK;
}

int rawdtw_chain_round_end(rawdtw_ctx *ctx, const rawdtw_anchor_t **d_anchors, const uint64_t **d_ref_base, const uint32_t **d_read_base)
{
    if (!ctx) return RAWDTW_ERR_INVALID;
    if (!d_anchors || !d_ref_base || !d_read_base) return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    if (!ctx->chain_ws || !ctx->chain_ws->w.pending) return fail(ctx, RAWDTW_ERR_INVALID, "no chaining round begun on this context");
    ChainWs &w = ctx->chain_ws->w;
    w.pending = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipEventSynchronize(w.done)); // (the round's own work: what was enqueued behind it -- the caller's next uploads -- goes on)
    const uint64_t *h_tot = static_cast<const uint64_t *>(w.pin);
    const uint64_t nc = h_tot[0], na = h_tot[1], flags = h_tot[2];
    if (h_tot[3]) return fail(ctx, RAWDTW_ERR_INVALID, "a seed's key is not below n_keys");
    if (flags) return fail(ctx, RAWDTW_ERR_UNSUPPORTED, flags & 4 ? "a read with more than 16 chains, two of them with equal scores: chain this round on the host"
                                                               : "a read with more than 32 chains (or more seeds than the device chains): chain this round on the host");
    if (nc > w.chains_cap) return fail(ctx, RAWDTW_ERR_RANGE, "chain output arrays too small");
    if (!w.direct) {
        HIP_TRY(ctx, hipMemcpyAsync(w.h_anchor_off, w.d_aoff, (nc + 1) * 8, hipMemcpyDeviceToHost, s));
        if (nc) HIP_TRY(ctx, hipMemcpyAsync(w.h_recs, w.d_recs, nc * sizeof(rawdtw_chain_rec_t), hipMemcpyDeviceToHost, s));
        if (w.h_anchors && na) HIP_TRY(ctx, hipMemcpyAsync(w.h_anchors, w.d_anch, na * sizeof(rawdtw_anchor_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
    }
    *d_anchors = w.d_anch; *d_ref_base = w.d_refb; *d_read_base = w.d_rbc;
    return RAWDTW_OK;
}

int rawdtw_chain_round(rawdtw_ctx *ctx, const rawdtw_chain_opt_t *opt, uint64_t n_reads, const uint64_t *seed_off, const rawdtw_seed_t *seeds,
                       const uint32_t *read_base, uint32_t n_keys, const uint64_t *key_base, uint64_t *chain_off, uint64_t *anchor_off,
                       rawdtw_chain_rec_t *recs, uint64_t chains_cap, rawdtw_anchor_t *anchors, const rawdtw_anchor_t **d_anchors,
                       const uint64_t **d_ref_base, const uint32_t **d_read_base)
{
    if (ctx && (!d_anchors || !d_ref_base || !d_read_base)) return fail(ctx, RAWDTW_ERR_INVALID, "null argument");
    const int st = rawdtw_chain_round_begin(ctx, opt, n_reads, seed_off, seeds, read_base, n_keys, key_base, chain_off, anchor_off, recs, chains_cap, anchors);
    return st == RAWDTW_OK ? rawdtw_chain_round_end(ctx, d_anchors, d_ref_base, d_read_base) : st;
}

} // extern "C"

namespace rawdtw { namespace capi {
void chain_ws_free(rawdtw_ctx *ctx)
{
    if (!ctx || !ctx->chain_ws) return;
    if (ctx->chain_ws->w.dev) (void)hipFree(ctx->chain_ws->w.dev);
    if (ctx->chain_ws->w.pin) (void)hipHostFree(ctx->chain_ws->w.pin);
    if (ctx->chain_ws->w.done) (void)hipEventDestroy(ctx->chain_ws->w.done);
    delete ctx->chain_ws;
    ctx->chain_ws = nullptr;
}
} }
