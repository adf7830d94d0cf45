// rawdtw_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for RawAlign's DTW hot path.
//
// What they replace (reference file:line):
//   k_band_tile / k_band_wreg / k_band_wave : DTW_global_slantedbanded_antidiagonalwise  src/dtw.cpp:273-520
//   k_full_wave<.,false>      : DTW_global                                 src/dtw.cpp:37-66
//   k_full_wave<.,true> + k_tb_walk_wave + k_tb_finish : DTW_global_tb                        src/dtw.cpp:595-667
//
// All of them are scalar fp32 min/add recurrences (no MFMA: nothing here is a contraction).
// Cell values do not depend on evaluation order (min is exact, each cell is one rounded
// subtract and one rounded add), so any wavefront order reproduces the reference bit for
// bit as long as the same cell set and the same neighbour rules are used.  Built with
// -ffp-contract=off.
#include "rawdtw_internal.h"
#include "rawdtw_dp.h"

namespace rawdtw {

// Tile kernel.  A tile is a run of CONSECUTIVE jobs of the batch (consecutive parts of the same
// chains), so the windows it needs form a few contiguous spans of the event and reference arenas:
// the workgroup copies each span HBM -> LDS once with coalesced 16-byte loads (adjacent parts share
// their anchor element and their cache lines; an earlier version that sorted jobs globally by shape
// fetched a whole 128-byte line per ~24-byte window: L2 miss rate 94 %, 3x the algorithmic bytes).
// The planner stores the tile's job records already ordered by (radius, longer side, shorter side),
// so that the 64 lanes of a wave get jobs of (nearly) one shape; records carry LDS offsets, the
// swap (dtw.cpp:284-292) and the slant-corrected radius (dtw.cpp:298-300) are resolved on the host.
template <int R>
__device__ __forceinline__ float tile_job(const float *win, const TileJob &tj)
{
    const float *LA = win + tj.offA;
    const float *LB = win + tj.offB;
    const uint32_t N = tj.N, M = tj.M;
    // wave-uniform shape?
    const uint32_t N0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)N);
    const uint32_t M0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)M);
    float res;
    // records are sorted by (longer side, shorter side): a wave usually shares the longer side (then the
    // column loop and half of the clipping are scalar) and often the whole shape (then everything is)
    if (__all(N == N0)) {
        if (__all(M == M0)) res = lane_dp<R>(LA, LB, N0, M0);
        else res = lane_dp_sel<R>(LA, LB, N0, M);
    } else res = lane_dp_sel<R>(LA, LB, N, M);
    if (tj.flags & kFlagExcludeLast) res = res - dist(LA[N - 1], LB[M - 1]);
    return res;
}

template <int W>
__device__ __forceinline__ float micro_job(const float *win, const TileJob &tj, const unsigned long long *masks)
{
    const float *LA = win + tj.offA;
    const float *LB = win + tj.offB;
    const uint32_t N = tj.N, M = tj.M;
    const unsigned long long mask = masks[tj.pad];
    // a tile's records are sorted by longer side, descending, inside a kind: the wave's first active lane has the
    // longest job, and the grid need not be wider than that
    const uint32_t Nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)N);
    float res;
    if constexpr (W == 4) {
        if (Nw <= 2) res = micro_job_cols<4, 2>(LA, LB, N, M, mask);
        else if (Nw == 3) res = micro_job_cols<4, 3>(LA, LB, N, M, mask);
        else res = micro_job_cols<4, 4>(LA, LB, N, M, mask);
    } else {
        if (Nw <= 5) res = micro_job_cols<8, 5>(LA, LB, N, M, mask);
        else if (Nw == 6) res = micro_job_cols<8, 6>(LA, LB, N, M, mask);
        else if (Nw == 7) res = micro_job_cols<8, 7>(LA, LB, N, M, mask);
        else res = micro_job_cols<8, 8>(LA, LB, N, M, mask);
    }
    if (tj.flags & kFlagExcludeLast) res = res - dist(LA[N - 1], LB[M - 1]);
    return res;
}

enum : int { kTileKindMicro4 = 0, kTileKindMicro8 = 1, kTileKindLane0 = 2 }; // lane kinds: 2 + radius

template <int R>
__device__ __forceinline__ void tile_dispatch(const float *win, const TileJob &tj, int kind, float &res)
{
    if (__any(kind == kTileKindLane0 + R)) {
        if (kind == kTileKindLane0 + R) res = tile_job<R>(win, tj);
    }
}

template <int R, int RHI>
__device__ __forceinline__ void tile_dispatch_range(const float *win, const TileJob &tj, int kind, float &res)
{
    tile_dispatch<R>(win, tj, kind, res);
    if constexpr (R < RHI) tile_dispatch_range<R + 1, RHI>(win, tj, kind, res);
}

// Two instances: <0, 3, true, 256> for the bulk (radius <= 3, micro paths for the tiniest shapes) and
// <0, 8, false, 64> (optional, "lane_hi") for the rare wider bands -- kept apart so that their registers and their long
// columns do not tax the waves of the bulk (0.4 % of the jobs cost half the time when mixed in).
template <int RLO, int RHI, bool MICRO, int THREADS>
__device__ __forceinline__ void tile_block(float *win, const TileDesc td, const TileSpan *__restrict__ spans,
                                           const TileJob *__restrict__ tjobs,
                                           const unsigned long long *__restrict__ masks,
                                           const float *__restrict__ ev, const float *__restrict__ ref,
                                           float *__restrict__ out)
{
    const int tid = threadIdx.x;
    // staging waves go first: a fresh workgroup's loads must not queue behind the VALU work of the older
    // workgroups on its SIMDs (the scheduler favours older waves)
    __builtin_amdgcn_s_setprio(3);
    auto load_job = [&](uint32_t r) { return tjobs[td.job_first + (r < td.n_jobs ? r : td.n_jobs - 1)]; };
    // the first round's records travel while the spans are staged
    TileJob tj_next = load_job((uint32_t)tid);
    // stage the spans: a few long ones (consecutive parts of a chain) with the whole workgroup on each, or -- tiles of
    // shape-sorted jobs: two short spans per job -- eight lanes per span, THREADS/8 spans at a time
    if (td.n_spans >> 31) {
        // four spans per lane group and pass: all descriptors first, then every chunk of the four spans (up to 24
        // chunks = 96 floats each, more than the longest tile-eligible window), then the LDS writes -- two memory
        // round trips per pass.  The staging registers are dead before the DP's become live.
        constexpr uint32_t G = THREADS / 8, U = 4, CH = 3;
        const uint32_t g = (uint32_t)tid >> 3, l = (uint32_t)tid & 7u;
        const uint32_t n_spans = td.n_spans & 0x7fffffffu;
        for (uint32_t s0 = g; s0 < n_spans; s0 += G * U) {
            TileSpan sp[U];
            float4 v[U][CH];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t sidx = s0 + u * G;
                sp[u] = spans[td.span_first + (sidx < n_spans ? sidx : s0)];
                if (sidx >= n_spans) sp[u].chunks_arena = 0;
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t chunks = sp[u].chunks_arena & 0x7fffffffu;
                const float4 *src = reinterpret_cast<const float4 *>(((sp[u].chunks_arena >> 31) ? ref : ev) + sp[u].src);
#pragma unroll
                for (uint32_t c = 0; c < CH; c++)
                    if (l + 8 * c < chunks) v[u][c] = src[l + 8 * c];
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t chunks = sp[u].chunks_arena & 0x7fffffffu;
                const float4 *src = reinterpret_cast<const float4 *>(((sp[u].chunks_arena >> 31) ? ref : ev) + sp[u].src);
                float4 *dst = reinterpret_cast<float4 *>(win + sp[u].lds_off);
#pragma unroll
                for (uint32_t c = 0; c < CH; c++)
                    if (l + 8 * c < chunks) dst[l + 8 * c] = v[u][c];
                for (uint32_t k = l + 8 * CH; k < chunks; k += 8) dst[k] = src[k]; // (merged spans can be longer)
            }
        }
    } else {
        for (uint32_t sidx = 0; sidx < td.n_spans; sidx++) {
            const TileSpan sp = spans[td.span_first + sidx];
            const uint32_t chunks = sp.chunks_arena & 0x7fffffffu;
            const float4 *src = reinterpret_cast<const float4 *>(((sp.chunks_arena >> 31) ? ref : ev) + sp.src);
            float4 *dst = reinterpret_cast<float4 *>(win + sp.lds_off);
            for (uint32_t k = tid; k < chunks; k += THREADS) dst[k] = src[k];
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    const uint32_t rounds = (td.n_jobs + THREADS - 1) / THREADS;
    for (uint32_t rd = 0; rd < rounds; rd++) {
        const uint32_t r = rd * THREADS + tid;
        const bool act = r < td.n_jobs;
        const TileJob tj = tj_next;
        tj_next = load_job(r + THREADS); // next round's records in flight behind this round's DP
        const int kind = act ? (int)tj.R : -1; // the planner stores the dispatch kind in the R byte
        float res = 0.0f;
        if constexpr (MICRO) {
            if (__any(kind == kTileKindMicro4)) { if (kind == kTileKindMicro4) res = micro_job<4>(win, tj, masks); }
            if (__any(kind == kTileKindMicro8)) { if (kind == kTileKindMicro8) res = micro_job<8>(win, tj, masks); }
        }
        tile_dispatch_range<RLO, RHI>(win, tj, kind, res);
        if (act) out[tj.aux] = res; // job order (aux = the job's index in the caller's batch)
    }
}

template <int RLO, int RHI, bool MICRO, int THREADS>
__global__ __launch_bounds__(THREADS) void k_band_tile(const TileDesc *__restrict__ tiles,
                                                       const TileSpan *__restrict__ spans,
                                                       const TileJob *__restrict__ tjobs,
                                                       const unsigned long long *__restrict__ masks,
                                                       const float *__restrict__ ev,
                                                       const float *__restrict__ ref,
                                                       float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float win[];
    tile_block<RLO, RHI, MICRO, THREADS>(win, tiles[blockIdx.x], spans, tjobs, masks, ev, ref, out);
}

// ---------------------------------------------------------------------------------------------
// Wave-per-job banded kernel (any radius whose three buffers fit LDS).  The 64 lanes sweep the
// offsets of one antidiagonal; the three rotating buffers are LDS arrays with the reference's
// physical indexing, so stale-slot behaviour is the reference's too.  All control values are
// wave-uniform.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_band_wave(const DevJob *__restrict__ jobs,
                                                  const float *__restrict__ ev,
                                                  const float *__restrict__ ref,
                                                  float *__restrict__ out)
{
    extern __shared__ float lds[];
    const DevJob jb = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const float *A = ev + jb.read_off;
    const float *B = ref + jb.ref_off;
    uint32_t N = jb.n, M = jb.m;
    if (N < M) {
        const float *tp = A; A = B; B = tp;
        uint32_t tn = N; N = M; M = tn;
    }
    const int R = jb.R;
    const int P = R + ((R % 2 == 0) ? 1 : 0);
    const int S = R + ((R % 2 == 1) ? 1 : 0);
    const int K = P > S ? P : S;
    const int SH = P > S ? 0 : 1;
    float *d0 = lds, *d1 = lds + K, *d2 = lds + 2 * K;
    for (int x = lane; x < 3 * K; x += 64) lds[x] = kInf;
    __syncthreads();
    if (lane == 0) d1[P / 2 + SH] = dist(A[0], B[0]);
    __syncthreads();

    int row = 0;
    uint32_t rem = 0;
    int prev_adv = 0;
    for (uint32_t col = 1; col < N; col++) {
        rem += M;
        const int adv = rem >= N;
        if (adv) { rem -= N; row++; }
        for (int pass = adv ? 0 : 1; pass < 2; pass++) {
            const int len = pass == 0 ? S : P;
            const int si = pass == 0 ? (int)col + S / 2 - 1 : (int)col + P / 2;
            const int sj = pass == 0 ? row - S / 2 : row - P / 2;
            int lo = 0, hi = len;
            if (si - (int)N + 1 > lo) lo = si - (int)N + 1;
            if (-sj > lo) lo = -sj;
            if (si + 1 < hi) hi = si + 1;
            if ((int)M - sj < hi) hi = (int)M - sj;
            // neighbour rule of this antidiagonal kind (see oracle RULES / dtw.cpp:368-485)
            int dt, dtl, dl, ds, g_top_first, g_tl_first, g_left_last;
            if (SH == 0) {
                if (pass == 0) { dt = 0; dtl = 0; dl = 1; ds = 0; g_top_first = 0; g_tl_first = 0; g_left_last = 0; }
                else if (adv)  { dt = -1; dtl = 0; dl = 0; ds = 0; g_top_first = 1; g_tl_first = 0; g_left_last = 1; }
                else           { dt = -1; dtl = -1; dl = 0; ds = 0; g_top_first = 1; g_tl_first = 1; g_left_last = 0; }
            } else {
                if (pass == 0) { dt = 0; dtl = 0; dl = 1; ds = 0; g_top_first = 1; g_tl_first = !prev_adv; g_left_last = 1; }
                else if (adv)  { dt = 0; dtl = 1; dl = 1; ds = 1; g_top_first = 0; g_tl_first = 0; g_left_last = 0; }
                else           { dt = 0; dtl = 0; dl = 1; ds = 1; g_top_first = 1; g_tl_first = !prev_adv; g_left_last = 0; }
            }
            for (int o = lo + lane; o < hi; o += 64) {
                const bool first = (o == 0), last = (o == len - 1);
                const float top = (g_top_first && first) ? kInf : d1[o + dt];
                const float tl = (g_tl_first && first) ? kInf : d0[o + dtl];
                const float left = (g_left_last && last) ? kInf : d1[o + dl];
                d2[o + ds] = min3f(top, left, tl) + dist(A[si - o], B[sj + o]);
            }
            __syncthreads();
            float *t = d0; d0 = d1; d1 = d2; d2 = t;
        }
        prev_adv = adv;
    }
    if (lane == 0) {
        float res = d1[P / 2 + SH];
        if (jb.flags & kFlagExcludeLast) res = res - dist(A[N - 1], B[M - 1]);
        out[jb.aux] = res;
    }
}

// ---------------------------------------------------------------------------------------------
// Full-matrix wavefront (DTW_global / DTW_global_tb fill).  One wave per job.  The shorter
// sequence Y is laid over lanes, RPL consecutive rows per lane, 64*RPL rows per strip; the
// longer sequence X is swept column by column with a one-column skew per lane (lane l works on
// column t-l at step t), so every step advances a 64-cell-wide anti-diagonal wavefront whose
// cells exchange values through DPP wave shifts only.  The DTW recurrence is symmetric under
// transposition, so which operand is "a" does not change any cell value.  Strips hand their
// last row to the next strip through a boundary row in HBM, read and written in 64-float
// chunks.  With TB the 2-bit move of dtw.cpp:633-646 is decided at fill time from the same
// three values and packed RPL codes per lane per step.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_band_grp16(const DevJob *__restrict__ jobs, uint32_t count,
                                                   const float *__restrict__ ev,
                                                   const float *__restrict__ ref,
                                                   float *__restrict__ out)
{
    grp_wave<16>(jobs, count, blockIdx.x, (int)threadIdx.x, ev, ref, out);
}

// ... and eight jobs per wave for bands of at most 8 offsets (radius <= 7)
__global__ __launch_bounds__(64) void k_band_grp8(const DevJob *__restrict__ jobs, uint32_t count,
                                                  const float *__restrict__ ev,
                                                  const float *__restrict__ ref,
                                                  float *__restrict__ out)
{
    grp_wave<8>(jobs, count, blockIdx.x, (int)threadIdx.x, ev, ref, out);
}

template <int C>
__global__ __launch_bounds__(64) void k_band_wreg(const DevJob *__restrict__ jobs,
                                                  const float *__restrict__ ev,
                                                  const float *__restrict__ ref,
                                                  float *__restrict__ out)
{
    wreg_body<C>(jobs[blockIdx.x], (int)threadIdx.x, ev, ref, out);
}

// bands of 257 .. 512 slots (global-mode chains of a few thousand events at banded=0.10): the LDS-window body with an ODD number of registers a
// lane -- five, seven or nine, the fewest that hold the band.  A lane's window is C consecutive floats of LDS: with eight the lanes of a
// read sit 8 dwords apart, on four of the 32 banks (16 passes a dword; the launch was LDS-bound at 1.2-1.5 TCUPS, 4 x 16-byte reads
// or 8 x 2 dwords alike); an odd stride visits every bank.
constexpr uint32_t kWband8LdsFloats = 128u * 9u + 2u * 1300u;
__global__ __launch_bounds__(64) void k_band_wband8(const DevJob *__restrict__ jobs, const float *__restrict__ ev, const float *__restrict__ ref,
                                                    float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float s_w[kWband8LdsFloats];
    const DevJob &jb = jobs[blockIdx.x];
    const int K = __builtin_amdgcn_readfirstlane(jb.R) + 1;
    if (K <= 320) wband_gen<5>(jb, (int)threadIdx.x, ev, ref, out, s_w, kWband8LdsFloats);
    else if (K <= 448) wband_gen<7>(jb, (int)threadIdx.x, ev, ref, out, s_w, kWband8LdsFloats);
    else wband_gen<9>(jb, (int)threadIdx.x, ev, ref, out, s_w, kWband8LdsFloats);
}

__global__ __launch_bounds__(64) void k_band_wreg_small(const DevJob *__restrict__ jobs,
                                                        const float *__restrict__ ev,
                                                        const float *__restrict__ ref,
                                                        float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float s_w[kWbandLdsFloats];
    wreg_small_job(jobs[blockIdx.x], (int)threadIdx.x, ev, ref, out, s_w);
}

// One launch for the three kernels every sparse batch needs -- the tile kernel (bulk), the 16-lane-row kernel
// (radius 4..15) and the register-resident wave kernel (radius + 1 <= 256, a few hundred long jobs).  A stream
// runs its kernels one after another and gfx950 ignores hipExtAnyOrderLaunch (scripts/experiments/anyorder.hip), so
// as separate launches the ~0.3 ms of the longest wide-band job sat in the batch's critical path while using a few
// percent of the chip.  Here it is the first workgroups of the launch and runs next to the tiles.  Workgroups are
// 256 threads: [0, nbw) take four wave jobs each, [nbw, nbw+nbg) sixteen 16-lane-row jobs each, the rest one tile each.
__global__ __launch_bounds__(256) void k_band_merged(const TileDesc *__restrict__ tiles,
                                                     const TileSpan *__restrict__ spans,
                                                     const TileJob *__restrict__ tjobs,
                                                     const unsigned long long *__restrict__ masks,
                                                     const DevJob *__restrict__ wjobs, uint32_t n_w,
                                                     const DevJob *__restrict__ gjobs, uint32_t n_g,
                                                     const DevJob *__restrict__ hjobs, uint32_t n_h,
                                                     const float *__restrict__ ev, const float *__restrict__ ref,
                                                     float *__restrict__ out, const uint32_t wave_floats)
{
    extern __shared__ __attribute__((aligned(16))) float win[];
    const uint32_t nbw = (n_w + 3u) / 4u, nbg = (n_g + 15u) / 16u, nbh = (n_h + 31u) / 32u;
    uint32_t b = blockIdx.x;
    const uint32_t wv = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    if (b < nbw) {
        const uint32_t j = b * 4u + wv;
        // (the tiles' image is these workgroups' to use: a quarter a wave, launch_band_merged sees to its size)
        if (j < n_w) wreg_small_job(wjobs[j], lane, ev, ref, out, win + wv * wave_floats, wave_floats);
        return;
    }
    b -= nbw;
    if (b < nbg) {
        grp_wave<16>(gjobs, n_g, b * 4u + wv, lane, ev, ref, out);
        return;
    }
    b -= nbg;
    if (b < nbh) {
        grp_wave<8>(hjobs, n_h, b * 4u + wv, lane, ev, ref, out);
        return;
    }
    b -= nbh;
    tile_block<0, kMaxLaneRadius, true, 256>(win, tiles[b], spans, tjobs, masks, ev, ref, out);
}

template <int RPL> struct DirWord { using type = uint8_t; };
template <> struct DirWord<8> { using type = uint16_t; };

// WAVES = 1: one wave per job, strips in sequence.  WAVES = 4 (jobs with several strips): the four waves
// of a workgroup take strips w, w+4, ... and run them as a pipeline -- strip s+1 trails strip s by at
// least one 64-column chunk -- so a long job finishes up to four times sooner (a launch of long jobs is
// bounded by its longest job, not by throughput).  Boundary rows then form a ring of WAVES rows in HBM;
// a strip publishes how many of its columns are flushed through an LDS progress word (strip index in the
// high bits, so the word only ever grows) and its successor spins on that word before each chunk load.
template <int RPL, bool TB, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_full_wave(const DevJob *__restrict__ jobs,
                                                  const FullAux *__restrict__ aux,
                                                  const float *__restrict__ ev,
                                                  const float *__restrict__ ref,
                                                  float *__restrict__ out, float *__restrict__ bnd_ws,
                                                  uint8_t *__restrict__ dir_ws)
{
    using word_t = typename DirWord<RPL>::type;
    const DevJob jb = jobs[blockIdx.x];
    const FullAux ax = aux[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const uint32_t wv = threadIdx.x >> 6;
    const float *a = ev + jb.read_off;
    const float *b = ref + jb.ref_off;
    const bool swapped = jb.n > jb.m; // the longer sequence is swept
    const float *X = swapped ? a : b;
    const float *Y = swapped ? b : a;
    const uint32_t NX = swapped ? jb.n : jb.m;
    const uint32_t NY = swapped ? jb.m : jb.n;
    constexpr uint32_t STRIP = 64u * RPL;
    const uint32_t nstrips = (NY + STRIP - 1) / STRIP;
    // Direction words are written in blocks of SPB steps: a lane's SPB consecutive words are 16 contiguous
    // bytes, so a block leaves the wave as one 1-KiB dwordx4 store ([strip][block][lane][step in block]).
    // Per step the word goes to LDS (2 KiB transposition buffer), per block each lane reads back its 16 bytes.
    constexpr uint32_t SPB = 16u / sizeof(word_t);
    const uint32_t TXB = (NX + 63 + SPB - 1) / SPB; // blocks per strip
    const uint32_t row_stride = (NX + 63u) & ~63u; // floats per boundary row (matches the planner)
    float *bnd_base = bnd_ws + ax.bnd_off;
    uint4 *dirs = reinterpret_cast<uint4 *>(dir_ws + ax.dir_off);
    __shared__ __attribute__((aligned(16))) word_t tbuf_all[TB ? WAVES * 64 * SPB : 1];
    word_t *tbuf = tbuf_all + (TB ? wv * 64 * SPB : 0);
    __shared__ uint32_t progress[WAVES]; // progress[s % WAVES] = (s << 21) | columns of strip s flushed
    if (WAVES > 1) {
        if (threadIdx.x < WAVES) progress[threadIdx.x] = 0;
        __syncthreads();
    }

    float result = 0.0f;
    for (uint32_t s = wv; s < nstrips; s += WAVES) {
        const float *bnd_in = bnd_base + (uint64_t)((s + WAVES - 1) % WAVES) * row_stride; // written by strip s-1
        float *bnd = bnd_base + (uint64_t)(s % WAVES) * row_stride;                          // read by strip s+1
        const uint32_t y0 = (s * 64u + lane) * RPL;
        float yv[RPL], v[RPL];
#pragma unroll
        for (int k = 0; k < RPL; k++) {
            uint32_t y = y0 + k;
            yv[k] = Y[y < NY ? y : NY - 1];
            v[k] = kInf; // column -1
        }
        const uint32_t rows_here = (NY - s * STRIP) < STRIP ? (NY - s * STRIP) : STRIP;
        const uint32_t lanes_here = (rows_here + RPL - 1) / RPL;
        const uint32_t steps = NX + lanes_here - 1;
        const bool has_rows = y0 < NY;
        const bool hands_down = (s + 1 < nstrips); // then the strip is full and lane 63 owns its last row
        // virtual row above row 0 is +inf, its corner D[-1][-1] is 0 so that D[0][0] = dist
        float diag_in = (s == 0 && lane == 0) ? 0.0f : kInf;
        float last_out = kInf, xval = 0.0f, xchunk = 0.0f, bchunk = kInf, wchunk = 0.0f;
        // (bit of the 2-bit code each plane sets: "up" bit 0 and "lf" bit 1, the other way round for a swapped job)

        for (uint32_t t = 0; t < steps; t++) {
            if ((t & 63u) == 0) {
                const uint32_t idx = t + lane;
                xchunk = X[idx < NX ? idx : NX - 1];
                if (s > 0) {
                    if (WAVES > 1) {
                        // wait until strip s-1 has flushed the columns of this chunk
                        const uint32_t need = ((s - 1) << 21) | ((t + 64u < NX) ? t + 64u : NX);
                        while (__hip_atomic_load(&progress[(s - 1) % WAVES], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                            __builtin_amdgcn_s_sleep(1);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    bchunk = idx < NX ? bnd_in[idx] : kInf;
                }
            }
            const float x0 = read_lane(xchunk, (int)(t & 63u));
            const float b0 = read_lane(bchunk, (int)(t & 63u));
            const float up_in = wave_shr1(last_out, b0);
            xval = wave_shr1(xval, x0);
            const uint32_t x = t - (uint32_t)lane;
            if (x < NX) {
                float d = diag_in, ab = up_in;
                uint32_t code = 0;
#pragma unroll
                for (int k = 0; k < RPL; k++) {
                    const float left = v[k];
                    const float nv = min3f(ab, left, d) + dist(xval, yv[k]);
                    if (TB) {
                        // dtw.cpp:633-646 with left = D[i-1][j], top = D[i][j-1] (i over a, j over b):
                        //   left < min(top, tl) -> 1 (i-1) ; else top < min(left, tl) -> 2 (j-1) ; else 0.
                        // In kernel coordinates: up = ab (row-1), lf = left (column-1).  "up strictly best"
                        // and "lf strictly best" exclude each other, so the code is two bit planes; Y rows
                        // are a-indices unless swapped, which only decides which plane is bit 0.
                        // Six instructions a cell: per plane v_min_f32 (as the instruction: through fminf the compiler quiets
                        // signalling NaNs first, two more instructions an operand), v_cmp_lt_f32 into VCC and v_addc_co_u32
                        // code = 2 code + VCC -- the compare's bit shifts in from below, no select, no shift-or.  The word grows
                        // most significant cell first, "up" before "lf": turned round behind the loop (below).
                        float t_up, t_lf;
                        asm("v_min_f32 %0, %1, %2" : "=v"(t_up) : "v"(left), "v"(d));
                        asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(code) : "v"(ab), "v"(t_up) : "vcc");
                        asm("v_min_f32 %0, %1, %2" : "=v"(t_lf) : "v"(ab), "v"(d));
                        asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(code) : "v"(left), "v"(t_lf) : "vcc");
                    }
                    d = left; ab = nv; v[k] = nv;
                }
                diag_in = up_in;
                last_out = v[RPL - 1];
                if (TB) {
                    // bit reversal puts cell k at bits 2k ("up", shifted in first) and 2k + 1 ("lf"): the layout of sh_up = 0,
                    // sh_lf = 1; a swapped job has the planes the other way round -- its adjacent bits change places
                    uint32_t w = __builtin_bitreverse32(code) >> (32 - 2 * RPL);
                    if (swapped) w = ((w & 0x55555555u) << 1) | ((w >> 1) & 0x55555555u);
                    tbuf[lane * SPB + (t % SPB)] = (word_t)w;
                }
            }
            if (TB && ((t % SPB) == SPB - 1 || t + 1 == steps)) {
                // same wave wrote and reads the buffer; LDS operations of a wave complete in order, the
                // fence keeps the compiler from reordering the differently typed accesses
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const uint4 blk = *reinterpret_cast<const uint4 *>(&tbuf[lane * SPB]);
                if (has_rows) dirs[((uint64_t)s * TXB + t / SPB) * 64u + lane] = blk;
            }
            if (hands_down && t >= 63u) {
                // lane 63 finished column c = t-63 in this step; gather 64 of them, store coalesced
                const uint32_t c = t - 63u;
                const float v63 = read_lane(last_out, 63);
                if ((uint32_t)lane == (c & 63u)) wchunk = v63;
                if ((c & 63u) == 63u || c == NX - 1) {
                    const uint32_t base = c & ~63u;
                    if (base + lane <= c) bnd[base + lane] = wchunk;
                    if (WAVES > 1) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0)
                            __hip_atomic_store(&progress[s % WAVES], (s << 21) | (c + 1u), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        if (s + 1 == nstrips) {
            const uint32_t yl = NY - 1;
            const uint32_t owner = (yl / RPL) & 63u;
            float mine = 0.0f;
#pragma unroll
            for (int k = 0; k < RPL; k++)
                if ((yl % RPL) == (uint32_t)k) mine = v[k];
            result = read_lane(mine, (int)owner);
        }
        __threadfence_block();
    }
    // the wave that ran the last strip owns the result
    if (lane == 0 && nstrips > 0 && ((nstrips - 1) % WAVES) == wv) {
        if (jb.flags & kFlagExcludeLast) result = result - dist(a[jb.n - 1], b[jb.m - 1]);
        out[jb.aux] = result;
    }
}

// Traceback walk, one wave per job.  (A lane-per-job walk chases one direction word per step through HBM:
// ~3.5 us per step, 5 ms for a thousand 1500-step paths.)  The direction buffer is laid out in 1-KiB block rows
// ([strip][block][lane][step in block]) and a path moving up the diagonal stays ~7 steps inside one block row: the
// wave fetches the block row with one coalesced load into LDS and all lanes then walk it in lockstep on uniform
// values.  Steps are buffered one per lane and leave as coalesced 64-step stores.  Output: (i, j) end-first; the
// distances and the start-first order are produced afterwards by k_tb_finish, one thread per path element.
template <int RPL>
__global__ __launch_bounds__(64) void k_tb_walk_wave(const DevJob *__restrict__ jobs, const FullAux *__restrict__ aux,
                                                     const uint8_t *__restrict__ dir_ws,
                                                     const uint64_t *__restrict__ path_off,
                                                     uint32_t *__restrict__ path_len, uint32_t *__restrict__ tmp_i,
                                                     uint32_t *__restrict__ tmp_j)
{
    using word_t = typename DirWord<RPL>::type;
    constexpr uint32_t SPB = 16u / sizeof(word_t);
    __shared__ __attribute__((aligned(16))) word_t row[64 * SPB]; // one block row: [lane][step in block]
    const int lane = threadIdx.x;
    const DevJob jb = jobs[blockIdx.x];
    const FullAux ax = aux[blockIdx.x];
    const bool swapped = jb.n > jb.m;
    const uint32_t NX = swapped ? jb.n : jb.m;
    const uint64_t TXB = ((uint64_t)NX + 63u + SPB - 1) / SPB; // blocks per strip
    const uint4 *dirs = reinterpret_cast<const uint4 *>(dir_ws + ax.dir_off);
    const uint64_t po = path_off[blockIdx.x];
    uint32_t i = jb.n - 1, j = jb.m - 1, k = 0;
    uint32_t bi = i, bj = j; // this lane's slot of the 64-step output buffer (lane = step & 63)
    uint64_t cur = ~0ull;    // block row held in LDS (prefetching the next one was measured: no gain, the walk's own
                             // dependent chain -- LDS read, decode, branch -- is what a step costs)
    auto flush = [&](uint32_t upto) { // steps [upto - ((upto - 1) & 63) - 1 .. upto) sit in lanes 0..(upto-1)&63
        const uint32_t base = (upto - 1) & ~63u;
        if ((uint32_t)lane <= ((upto - 1) & 63u)) { tmp_i[po + base + lane] = bi; tmp_j[po + base + lane] = bj; }
    };
    // step 0: the end cell
    if (lane == 0) { bi = i; bj = j; }
    k = 1;
    while (i > 0 || j > 0) {
        if (i == 0) j--;
        else if (j == 0) i--;
        else {
            const uint32_t y = swapped ? j : i, x = swapped ? i : j;
            const uint32_t sidx = y / (64u * RPL), l = (y / RPL) & 63u, kk = y % RPL;
            const uint32_t t = x + l;
            const uint64_t key = (uint64_t)sidx * TXB + t / SPB;
            if (key != cur) {
                reinterpret_cast<uint4 *>(row)[lane] = dirs[key * 64u + lane];
                cur = key;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            uint32_t word = row[l * SPB + (t % SPB)];
            word = (uint32_t)__builtin_amdgcn_readfirstlane((int)word);
            const uint32_t code = (word >> (2 * kk)) & 3u;
            if (code == 1u) i--;
            else if (code == 2u) j--;
            else { i--; j--; }
            // (the next block row may overwrite `row` only after every lane has read this word: the barrier above
            // is the only writer and all lanes reach it together, the walk being uniform)
        }
        if ((uint32_t)lane == (k & 63u)) { bi = i; bj = j; }
        k++;
        if ((k & 63u) == 0) flush(k);
    }
    if ((k & 63u) != 0) flush(k);
    if (lane == 0) path_len[blockIdx.x] = k;
}

// Start-first order and the per-step distance: one thread per path element.  What leaves the device per element is its
// distance and ONE byte: the step from the element before it (bit 0: i advanced, bit 1: j advanced; a global path starts at
// (0, 0): dtw.cpp:640-655) -- the host rebuilds (i, j) while it writes the caller's arrays; 5 bytes an element over the bus
// instead of 12.
__global__ __launch_bounds__(256) void k_tb_finish(const DevJob *__restrict__ jobs, uint32_t count,
                                                   const float *__restrict__ ev, const float *__restrict__ ref,
                                                   const uint64_t *__restrict__ path_off,
                                                   const uint32_t *__restrict__ path_len,
                                                   const uint32_t *__restrict__ tmp_i, const uint32_t *__restrict__ tmp_j,
                                                   uint8_t *__restrict__ path_mv, float *__restrict__ path_d)
{
    const uint32_t g = blockIdx.x;
    if (g >= count) return;
    const DevJob jb = jobs[g];
    const float *a = ev + jb.read_off;
    const float *b = ref + jb.ref_off;
    const uint64_t po = path_off[g];
    const uint32_t len = path_len[g];
    for (uint32_t q = threadIdx.x; q < len; q += 256) {
        const uint32_t i = tmp_i[po + len - 1 - q], j = tmp_j[po + len - 1 - q];
        uint32_t mv = 0;
        if (q) mv = (i - tmp_i[po + len - q]) | ((j - tmp_j[po + len - q]) << 1);
        path_mv[po + q] = (uint8_t)mv; path_d[po + q] = dist(a[i], b[j]);
    }
}

// ---------------------------------------------------------------------------------------------
// Per-candidate selection: the fold of align_chain (rmap.cpp:238-306) and the best-so-far loop of
// gen_chains (rmap.cpp:515-524), on the device.  A chain's running `current_max_attainable_score`
// (rmap.cpp:246,280) never increases (every sub-cost is >= 0), so "some check before part p
// fails" <=> "the check before the LAST part fails"; the fold therefore does not depend on the
// running best and runs one lane per chain, and only the tiny accept/cut loop is sequential per
// read.  All arithmetic is fp32 in the reference's order; the final score is a single fma when the
// reference build contracts it.
// ---------------------------------------------------------------------------------------------
// One wave per chain.  The part costs are fetched 64 at a time (coalesced, next chunk in flight while the current one is
// folded); the fp32 fold itself is inherently sequential and runs on wave-uniform values taken out of the chunk with
// v_readlane: two instructions a part, but no memory round trip per 32 parts as in the lane-per-chain form -- the form for
// the few chains long enough to bound the launch.
__device__ __forceinline__ void fold_wave_body(const uint32_t c, const ChainDesc d, const int lane, const float *__restrict__ job_cost,
                                               const float bonus, const int fused, float *__restrict__ full_score,
                                               float *__restrict__ att_last)
{
    float attainable = (float)d.span * bonus; // rmap.cpp:205,246
    float cost = 0.0f;
    const float *jc = job_cost + d.job_first;
    const long long dir = d.descending ? -1 : 1; // part p is jc[dir * p]
    // all parts but the last: cost += sub; attainable -= sub (two independent fp32 chains)
    const uint32_t body = d.n_jobs ? d.n_jobs - 1 : 0;
    uint32_t base = 0;
    float nxt = ((uint32_t)lane < d.n_jobs) ? jc[dir * lane] : 0.0f;
    for (; base + 64 <= body; base += 64) {
        const float chunk = nxt;
        nxt = (base + 64 + (uint32_t)lane < d.n_jobs) ? jc[dir * (long long)(base + 64 + lane)] : 0.0f;
        v2f acc = {cost, attainable};
#pragma unroll
        for (int k = 0; k < 64; k++) {
            const float sub = read_lane(chunk, k);
            acc += v2f{sub, -sub}; // one packed add: cost += sub (rmap.cpp:279), attainable -= sub (rmap.cpp:280)
        }
        cost = acc.x; attainable = acc.y;
    }
    {
        const uint32_t cnt = d.n_jobs - base; // 0..64 parts left, the last one included
        const float chunk = nxt;
        for (uint32_t k = 0; k + 1 < cnt; k++) {
            const float sub = read_lane(chunk, (int)k);
            cost += sub;
            attainable -= sub;
        }
        if (cnt) cost += read_lane(chunk, (int)(cnt - 1));
    }
    float gate = attainable; // the value tested before the last (or only) DTW call (rmap.cpp:205,265)
    if (d.n_jobs == 0) gate = __builtin_inff(); // no DTW call, no check
    float score;
    if (fused) score = __builtin_fmaf((float)d.num_aligned, bonus, -cost);
    else { const float prod = (float)d.num_aligned * bonus; score = prod - cost; }
    if (lane == 0) {
        full_score[c] = score;
        att_last[c] = gate;
    }
}


__global__ __launch_bounds__(256) void k_chain_fold(const ChainDesc *__restrict__ chains,
                                                    const uint32_t *__restrict__ order, uint64_t n_chains,
                                                    const float *__restrict__ job_cost, float bonus,
                                                    int fused, float *__restrict__ full_score,
                                                    float *__restrict__ att_last)
{
    // (`order`: longest chains first, so the long folds start first)
    const uint64_t t = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6;
    if (t >= n_chains) return;
    const uint32_t c = order[t];
    fold_wave_body(c, chains[c], threadIdx.x & 63, job_cost, bonus, fused, full_score, att_last);
}

// Lane-per-chain fold: 64 chains per wave (`order`: longest first, so a wave's chains have similar lengths), each
// lane adds its own chain's part costs in order.  One packed add per 64 parts instead of a v_readlane + add per
// part: ~30x less VALU work than k_chain_fold, which matters once several batches share the chip (the DTW kernels
// are VALU-bound).  Each lane streams 4 bytes at a time through its own cache lines; U parts are fetched one
// round ahead so that a round costs one memory round trip.
template <int U>
__device__ __forceinline__ void fold_lane_body(const bool act, const uint32_t c, ChainDesc d, const float *__restrict__ job_cost,
                                               const float bonus, const int fused, float *__restrict__ full_score,
                                               float *__restrict__ att_last)
{
    if (!act) d.n_jobs = 0;    const uint32_t body = d.n_jobs ? d.n_jobs - 1 : 0; // all parts but the last: cost += sub; attainable -= sub
    const float *jc = job_cost + d.job_first;
    const long long dir = d.descending ? -1 : 1; // part p is jc[dir * p]
    v2f acc = {0.0f, (float)d.span * bonus}; // {cost, attainable}  (rmap.cpp:205,246)
    float nx[U];
#pragma unroll
    for (int u = 0; u < U; u++) nx[u] = ((uint32_t)u < body) ? jc[dir * u] : 0.0f;
    for (uint32_t k = 0; __any(k < body); k += U) {
        float cur[U];
#pragma unroll
        for (int u = 0; u < U; u++) cur[u] = nx[u];
#pragma unroll
        for (int u = 0; u < U; u++) nx[u] = (k + U + (uint32_t)u < body) ? jc[dir * (long long)(k + U + u)] : 0.0f;
#pragma unroll
        for (int u = 0; u < U; u++) acc += v2f{cur[u], -cur[u]}; // x + 0 and x - 0 are exact: finished chains idle
    }
    float cost = acc.x;
    const float gate = d.n_jobs ? acc.y : __builtin_inff(); // tested before the last (or only) DTW call; none: no check
    if (d.n_jobs) cost += jc[dir * (long long)(d.n_jobs - 1)]; // the last part only adds to the cost (rmap.cpp:279-280)
    float score;
    if (fused) score = __builtin_fmaf((float)d.num_aligned, bonus, -cost);
    else { const float prod = (float)d.num_aligned * bonus; score = prod - cost; }
    if (act) {
        full_score[c] = score;
        att_last[c] = gate;
    }
}

template <int U>
__global__ __launch_bounds__(64) void k_chain_fold_lane(const ChainDesc *__restrict__ chains,
                                                        const uint32_t *__restrict__ order, uint64_t n_chains,
                                                        const float *__restrict__ job_cost, float bonus, int fused,
                                                        float *__restrict__ full_score, float *__restrict__ att_last)
{
    const uint64_t t = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    const bool act = t < n_chains;
    const uint32_t c = act ? order[t] : 0u;
    fold_lane_body<U>(act, c, chains[act ? c : order[0]], job_cost, bonus, fused, full_score, att_last);
}

// Both forms in one launch of one-wave workgroups (`order`: longest chains first): the first `long_waves` take the entries
// order[0 .. long_waves) a wave each -- those of at least `long_parts` parts; a shorter one is left alone -- and the others
// take every entry a lane each, idling on the chains the waves took.  The lane form alone ends with its longest chain
// (one memory round trip per 32 parts: 43 rounds for the bench batch's 1373 parts); a thousand chains of 768+ parts cost
// the wave form 2 M instructions and leave the lanes 24 rounds.  (One wave per workgroup: the lane form's scattered loads
// are bound by a CU's texture path -- four such waves on one CU took twice as long.)
__global__ __launch_bounds__(64) void k_chain_fold_hybrid(const ChainDesc *__restrict__ chains, const uint32_t *__restrict__ order,
                                                          uint64_t n_chains, const float *__restrict__ job_cost, float bonus, int fused,
                                                          float *__restrict__ full_score, float *__restrict__ att_last,
                                                          uint32_t long_parts, uint32_t long_waves)
{
    if (blockIdx.x < long_waves) {
        const uint32_t c = order[blockIdx.x]; // (long_waves <= n_chains)
        const ChainDesc d = chains[c];
        if (d.n_jobs < long_parts) return;
        fold_wave_body(c, d, threadIdx.x, job_cost, bonus, fused, full_score, att_last);
        return;
    }
    const uint64_t t = (uint64_t)(blockIdx.x - long_waves) * 64u + threadIdx.x;
    bool act = t < n_chains;
    const uint32_t c = act ? order[t] : 0u;
    const ChainDesc d = chains[act ? c : order[0]];
    if (t < long_waves && d.n_jobs >= long_parts) act = false; // (a wave has it)
    fold_lane_body<32>(act, c, d, job_cost, bonus, fused, full_score, att_last);
}

__global__ __launch_bounds__(256) void k_read_select(const uint64_t *__restrict__ chain_off, uint64_t n_reads,
                                                     const float *__restrict__ full_score,
                                                     const float *__restrict__ att_last, float min_score,
                                                     float *__restrict__ score, uint8_t *__restrict__ keep)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= n_reads) return;
    float best = 0.0f; // rmap.cpp:515
    for (uint64_t c = chain_off[r]; c < chain_off[r + 1]; c++) {
        const float s = (att_last[c] < best) ? -1e10f : full_score[c]; // rmap.cpp:206-209, 265-268
        const bool k = s >= min_score;                                  // rmap.cpp:518
        if (k && s > best) best = s;                                    // rmap.cpp:519-521
        score[c] = s;
        keep[c] = k ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_chain_fold(int mode, const ChainDesc *chains, const uint32_t *order, uint64_t n_chains,
                             const float *job_cost, float bonus, int fused, float *full_score, float *att_last, uint32_t long_parts,
                             hipStream_t s)
{
    if (n_chains == 0) return hipSuccess;
    if (mode == 3) {
        const uint32_t long_waves = (uint32_t)std::min<uint64_t>(n_chains, 4096); // at most this many chains a wave each
        hipLaunchKernelGGL(k_chain_fold_hybrid, dim3(long_waves + (uint32_t)((n_chains + 63) / 64)), dim3(64), 0, s, chains, order,
                           n_chains, job_cost, bonus, fused, full_score, att_last, long_parts, long_waves);
    } else if (mode == 1)
        hipLaunchKernelGGL(k_chain_fold_lane<16>, dim3((uint32_t)((n_chains + 63) / 64)), dim3(64), 0, s, chains, order,
                           n_chains, job_cost, bonus, fused, full_score, att_last);
    else if (mode == 2)
        hipLaunchKernelGGL(k_chain_fold_lane<32>, dim3((uint32_t)((n_chains + 63) / 64)), dim3(64), 0, s, chains, order,
                           n_chains, job_cost, bonus, fused, full_score, att_last);
    else
        hipLaunchKernelGGL(k_chain_fold, dim3((uint32_t)((n_chains + 3) / 4)), dim3(256), 0, s, chains, order, n_chains,
                           job_cost, bonus, fused, full_score, att_last);
    return hipGetLastError();
}

hipError_t launch_read_select(const uint64_t *chain_off, uint64_t n_reads, const float *full_score,
                              const float *att_last, float min_score, float *score, uint8_t *keep, hipStream_t s)
{
    if (n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(k_read_select, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, s, chain_off, n_reads,
                       full_score, att_last, min_score, score, keep);
    return hipGetLastError();
}

template <int RLO, int RHI, bool MICRO, int THREADS>
static hipError_t launch_tile_t(const TileDesc *tiles, uint64_t n_tiles, const TileSpan *spans, const TileJob *tjobs,
                                const unsigned long long *masks, uint32_t lds_floats, const float *ev, const float *ref,
                                float *out, hipStream_t s)
{
    const size_t lds_bytes = (size_t)lds_floats * sizeof(float);
    auto kern = k_band_tile<RLO, RHI, MICRO, THREADS>;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((uint32_t)n_tiles), dim3(THREADS), lds_bytes, s, tiles, spans, tjobs, masks, ev, ref, out);
    return hipGetLastError();
}

hipError_t launch_band_tile(bool hi, int threads, const TileDesc *tiles, uint64_t n_tiles, const TileSpan *spans,
                            const TileJob *tjobs, const unsigned long long *masks, uint32_t lds_floats, const float *ev,
                            const float *ref, float *out, hipStream_t s)
{
    if (n_tiles == 0) return hipSuccess;
    if (hi) return launch_tile_t<0, kMaxLaneRadiusHi, false, 64>(tiles, n_tiles, spans, tjobs, masks, lds_floats, ev, ref, out, s);
    if (threads >= 1024) return launch_tile_t<0, kMaxLaneRadius, true, 1024>(tiles, n_tiles, spans, tjobs, masks, lds_floats, ev, ref, out, s);
    if (threads >= 512) return launch_tile_t<0, kMaxLaneRadius, true, 512>(tiles, n_tiles, spans, tjobs, masks, lds_floats, ev, ref, out, s);
    return launch_tile_t<0, kMaxLaneRadius, true, 256>(tiles, n_tiles, spans, tjobs, masks, lds_floats, ev, ref, out, s);
}

hipError_t launch_band_merged(const TileDesc *tiles, uint64_t n_tiles, const TileSpan *spans, const TileJob *tjobs,
                              const unsigned long long *masks, uint32_t lds_floats, const DevJob *wjobs, uint64_t n_w,
                              const DevJob *gjobs, uint64_t n_g, const DevJob *hjobs, uint64_t n_h, const float *ev,
                              const float *ref, float *out, hipStream_t s)
{
    const uint64_t blocks = (n_w + 3) / 4 + (n_g + 15) / 16 + (n_h + 31) / 32 + n_tiles;
    if (blocks == 0) return hipSuccess;
    if (n_w) lds_floats = std::max(lds_floats, 4u * kWbandLdsFloats); // (the wave-per-job bands keep their operand windows there)
    const uint32_t wave_floats = (lds_floats / 4u) & ~3u;
    const size_t lds_bytes = (size_t)lds_floats * sizeof(float);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_band_merged),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_band_merged, dim3((uint32_t)blocks), dim3(256), lds_bytes, s, tiles, spans, tjobs, masks, wjobs,
                       (uint32_t)n_w, gjobs, (uint32_t)n_g, hjobs, (uint32_t)n_h, ev, ref, out, wave_floats);
    return hipGetLastError();
}

hipError_t launch_band_wave(const DevJob *jobs, uint64_t count, uint32_t lds_floats, const float *ev,
                            const float *ref, float *out, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    const size_t lds_bytes = (size_t)lds_floats * sizeof(float);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_band_wave),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_band_wave, dim3((uint32_t)count), dim3(64), lds_bytes, s, jobs, ev, ref, out);
    return hipGetLastError();
}

template <int C>
static hipError_t launch_wreg_c(const DevJob *jobs, uint64_t count, const float *ev, const float *ref, float *out,
                                hipStream_t s)
{
    hipLaunchKernelGGL(k_band_wreg<C>, dim3((uint32_t)count), dim3(64), 0, s, jobs, ev, ref, out);
    return hipGetLastError();
}

hipError_t launch_band_wreg(int chunks, const DevJob *jobs, uint64_t count, const float *ev, const float *ref,
                            float *out, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    switch (chunks) {
    case -8: // eight jobs per wave, radius + 1 <= 8
        hipLaunchKernelGGL(k_band_grp8, dim3((uint32_t)((count + 7) / 8)), dim3(64), 0, s, jobs, (uint32_t)count, ev, ref, out);
        return hipGetLastError();
    case -16: // four jobs per wave, radius + 1 <= 16
        hipLaunchKernelGGL(k_band_grp16, dim3((uint32_t)((count + 3) / 4)), dim3(64), 0, s, jobs, (uint32_t)count, ev, ref, out);
        return hipGetLastError();
    case 0: // merged launch: radius + 1 <= 256, variant chosen per job
        hipLaunchKernelGGL(k_band_wreg_small, dim3((uint32_t)count), dim3(64), 0, s, jobs, ev, ref, out);
        return hipGetLastError();
    case 1: return launch_wreg_c<1>(jobs, count, ev, ref, out, s);
    case 2: return launch_wreg_c<2>(jobs, count, ev, ref, out, s);
    case 4: return launch_wreg_c<4>(jobs, count, ev, ref, out, s);
    case 8:
        hipLaunchKernelGGL(k_band_wband8, dim3((uint32_t)count), dim3(64), 0, s, jobs, ev, ref, out);
        return hipGetLastError();
    case 16: return launch_wreg_c<16>(jobs, count, ev, ref, out, s);
    case 32: return launch_wreg_c<32>(jobs, count, ev, ref, out, s);
    default: return hipErrorInvalidValue;
    }
}

template <int RPL, bool TB>
static hipError_t launch_full_t(const DevJob *jobs, uint64_t count, const FullAux *aux, const float *ev,
                                const float *ref, float *out, float *bnd_ws, uint8_t *dir_ws,
                                hipStream_t s)
{
    hipLaunchKernelGGL((k_full_wave<RPL, TB, 1>), dim3((uint32_t)count), dim3(64), 0, s, jobs, aux, ev, ref,
                       out, bnd_ws, dir_ws);
    return hipGetLastError();
}

hipError_t launch_full_wave(int rpl, bool tb, const DevJob *jobs, uint64_t count, const FullAux *aux,
                            const float *ev, const float *ref, float *out, float *bnd_ws,
                            uint8_t *dir_ws, hipStream_t s)
{
    if (count == 0) return hipSuccess;
#define RAWDTW_FULL_CASE(r)                                                                           \
    case r:                                                                                           \
        return tb ? launch_full_t<r, true>(jobs, count, aux, ev, ref, out, bnd_ws, dir_ws, s)         \
                  : launch_full_t<r, false>(jobs, count, aux, ev, ref, out, bnd_ws, dir_ws, s);
    switch (rpl) {
        RAWDTW_FULL_CASE(1)
        RAWDTW_FULL_CASE(2)
        RAWDTW_FULL_CASE(4)
        RAWDTW_FULL_CASE(8)
    case 8 + 256: // multi-strip jobs: four waves per job (kFullWgWaves)
        if (tb) hipLaunchKernelGGL((k_full_wave<8, true, kFullWgWaves>), dim3((uint32_t)count), dim3(64 * kFullWgWaves), 0, s,
                                   jobs, aux, ev, ref, out, bnd_ws, dir_ws);
        else hipLaunchKernelGGL((k_full_wave<8, false, kFullWgWaves>), dim3((uint32_t)count), dim3(64 * kFullWgWaves), 0, s,
                                jobs, aux, ev, ref, out, bnd_ws, dir_ws);
        return hipGetLastError();
    default: return hipErrorInvalidValue;
    }
#undef RAWDTW_FULL_CASE
}

// wave-per-job walk (end-first (i, j) into tmp_i / tmp_j) followed by k_tb_finish (start-first i, j, d)
hipError_t launch_tb_walk_wave(const DevJob *jobs, uint64_t count, const FullAux *aux, int rpl, const float *ev,
                               const float *ref, const uint8_t *dir_ws, const uint64_t *path_off, uint32_t *path_len,
                               uint32_t *tmp_i, uint32_t *tmp_j, uint8_t *path_mv, float *path_d, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    const dim3 grid((uint32_t)count), block(64);
    switch (rpl) {
    case 1: hipLaunchKernelGGL(k_tb_walk_wave<1>, grid, block, 0, s, jobs, aux, dir_ws, path_off, path_len, tmp_i, tmp_j); break;
    case 2: hipLaunchKernelGGL(k_tb_walk_wave<2>, grid, block, 0, s, jobs, aux, dir_ws, path_off, path_len, tmp_i, tmp_j); break;
    case 4: hipLaunchKernelGGL(k_tb_walk_wave<4>, grid, block, 0, s, jobs, aux, dir_ws, path_off, path_len, tmp_i, tmp_j); break;
    default: hipLaunchKernelGGL(k_tb_walk_wave<8>, grid, block, 0, s, jobs, aux, dir_ws, path_off, path_len, tmp_i, tmp_j); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_tb_finish, grid, dim3(256), 0, s, jobs, (uint32_t)count, ev, ref, path_off, path_len, tmp_i, tmp_j,
                       path_mv, path_d);
    return hipGetLastError();
}

} // namespace rawdtw
