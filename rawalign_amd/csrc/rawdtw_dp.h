// rawdtw_dp.h -- the DP bodies shared by the kernels of rawdtw_kernels.hip (job-list plans) and rawdtw_stream.hip
// (sync-free candidate batches): lane-per-job, micro, group-per-job and wave-per-job forms of
// DTW_global_slantedbanded_antidiagonalwise (src/dtw.cpp:273-520).  Device code only; include from .hip files.
#pragma once
#include "rawdtw_internal.h"

namespace rawdtw {

__device__ __forceinline__ float min3f(float top, float left, float tl)
{
    // std::min(std::min(top,left),topleft); identical for non-NaN operands.  Written as the instruction: through
    // fminf the compiler must quiet signalling NaNs first (a v_max_f32 x,x per operand it cannot prove canonical,
    // and two v_min_f32 instead of one v_min3_f32) -- a quarter of the lane DP's VALU work.
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(top), "v"(left), "v"(tl));
    return r;
}

__device__ __forceinline__ float dist(float x, float y) { return __builtin_fabsf(x - y); }

// Per-lane selects on a 64-bit lane mask, written as the instruction's VOP3 form.  Left to itself the compiler keeps a
// compare's result in VCC and shrinks the selects on it to the 32-bit encoding (v_cndmask_b32_e32 ..., vcc), which gfx950
// issues at 16-23 clocks a wave instruction whatever else the SIMD has to do; the 64-bit encoding with the mask in an SGPR
// pair (or VCC) takes 4.5, as v_min3_f32 does (scripts/experiments/valu_rate.hip; v_add_f32 / v_sub_f32: 2.5-3).  The
// lane bodies' row advance is per lane -- a third of their instructions are such selects.
typedef unsigned long long lmask;
__device__ __forceinline__ float selm(const lmask m, const float t, const float f) // m ? t : f
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t selm(const lmask m, const uint32_t t, const uint32_t f)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}
// rem >= N ? rem - N : rem without a compare (unsigned: the difference wraps far above rem when rem < N)
__device__ __forceinline__ uint32_t wrap_sub(const uint32_t rem, const uint32_t N) { return min(rem, rem - N); }

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_shr1(float v, float fill)
{
    // lane l receives lane l-1's value; lane 0 keeps `fill`  (DPP wave_shr:1 = 0x138)
    int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                        0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}

__device__ __forceinline__ float read_lane(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ __forceinline__ float wave_shl1(float v, float fill)
{
    // lane l receives lane l+1's value; lane 63 keeps `fill`  (DPP wave_shl:1 = 0x130)
    int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                        0x130, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}

// ---------------------------------------------------------------------------------------------
// Lane-per-job banded kernel.  Sparse-mode segments are tiny (2..~70 events, radius 1..6):
// intra-job parallelism is a handful of cells per antidiagonal, so each lane owns one job and
// the antidiagonal buffers live in registers (radius is a template parameter, every buffer index
// is a compile-time constant).
//
// Buffers.  The reference rotates three buffers (dtw.cpp:305-314, 410-413, 487-490); slots an
// antidiagonal does not overwrite keep stale values, but no in-matrix cell ever reads such a slot
// (every neighbour read is either a cell of the band's cell set or a guarded/never-written slot
// holding 1e10 -- the oracle's orc_dtw_banded_cellset states exactly this and is bit-identical to
// the reference).  So clipped cells are written as 1e10 and two buffers suffice: p1 = latest
// antidiagonal, p2 = the one before.  A column whose centre row advances computes its secondary
// antidiagonal in place over p2 and its primary in place over p1 (each cell reads only its own
// slot of the buffer it overwrites); a column that stays on the row computes its primary over p2
// and swaps.  No rotation copies.
//
// Control.  Jobs are sorted by (longer side, shorter side), so the 64 jobs of a wave almost always
// share one shape; then the whole band geometry (row advance, clipping) is wave-uniform and the
// loop runs with scalar control flow (`lane_dp` instantiated on readfirstlane'd lengths).
//
// Operands.  The workgroup first stages the windows of its JOBS jobs from HBM into LDS with
// coalesced 16-byte loads (a lane reading its own window 4 bytes at a time would pull a whole
// 128-byte line per element through L2).  Each job owns STRIDE floats of LDS: the 16-byte
// aligned chunks covering its longer window at [0, CAP) and its shorter one at [CAP, 2*CAP);
// STRIDE/4 is odd so that lanes spread over the banks.  The DP slides both windows through
// registers -- one new a-value per column, one new b-value when the centre row moves -- fetched
// from LDS one step ahead of their use.
// ---------------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ float lane_dp(const float *LA, const float *LB, const uint32_t N, const uint32_t M)
{
    constexpr int P = R + ((R % 2 == 0) ? 1 : 0); // dtw.cpp:301
    constexpr int S = R + ((R % 2 == 1) ? 1 : 0); // dtw.cpp:302
    constexpr int K = (P > S) ? P : S;
    constexpr int SH = (P > S) ? 0 : 1; // primaries live at index+1 when the secondary is longer
    constexpr int HP = P / 2, HS = S / 2;
    constexpr bool EVEN = (R % 2 == 0);
    const int iN = (int)N, iM = (int)M;

    float p1[K], p2[K];
    float aw[K];     // aw[x] = A[col + HP - x]
    float bw[K + 1]; // bw[x] = B[row - HP - 1 + x]
#pragma unroll
    for (int x = 0; x < K; x++) {
        p1[x] = kInf; p2[x] = kInf;
        const int ia = HP - x;
        aw[x] = LA[ia < 0 ? 0 : (ia >= iN ? iN - 1 : ia)];
    }
#pragma unroll
    for (int x = 0; x <= K; x++) {
        const int ib = x - HP - 1;
        bw[x] = LB[ib < 0 ? 0 : (ib >= iM ? iM - 1 : ib)];
    }
    // column 0: only the corner (dtw.cpp:317-347)
    p1[HP + SH] = dist(LA[0], LB[0]);

    int row = 0;
    uint32_t rem = 0; // M*col - row*N, so "row advances" <=> rem + M >= N  (dtw.cpp:352-359)
    bool prev_adv = false;
    // one-ahead operand fetch: a_next = A[col + HP] for the coming column,
    // b_next = B[row + 1 - HP - 1 + K] for the coming row advance
    float a_next, b_next;
    {
        const int ia = 1 + HP, ib = 1 - HP - 1 + K;
        a_next = LA[ia >= iN ? iN - 1 : ia];
        b_next = LB[ib < 0 ? 0 : (ib >= iM ? iM - 1 : ib)];
    }
    for (uint32_t col = 1; col < N; col++) {
        rem += M;
        const bool adv = rem >= N;
#pragma unroll
        for (int x = K - 1; x > 0; x--) aw[x] = aw[x - 1];
        aw[0] = a_next;
        {
            const int ia = (int)col + 1 + HP;
            a_next = LA[ia >= iN ? iN - 1 : ia];
        }
        if (adv) {
            rem -= N;
            row++;
#pragma unroll
            for (int x = 0; x < K; x++) bw[x] = bw[x + 1];
            bw[K] = b_next;
            {
                const int ib = row + 1 - HP - 1 + K;
                b_next = LB[ib < 0 ? 0 : (ib >= iM ? iM - 1 : ib)];
            }
            // secondary antidiagonal (dtw.cpp:361-414), in place over p2
#pragma unroll
            for (int o = 0; o < S; o++) {
                const int i = (int)col + HS - 1 - o;
                const int j = row - HS + o;
                const bool valid = (uint32_t)i < N && (uint32_t)j < M;
                const float av = EVEN ? aw[o + 1 < K ? o + 1 : K - 1] : aw[o];
                const float bv = EVEN ? bw[o + 1] : bw[o];
                float top, tl, left;
                if (SH == 0) {
                    top = p1[o]; tl = p2[o]; left = p1[o + 1 < K ? o + 1 : K - 1];
                } else {
                    top = (o == 0) ? kInf : p1[o];
                    tl = (o == 0 && !prev_adv) ? kInf : p2[o];
                    left = (o == S - 1) ? kInf : p1[o + 1 < K ? o + 1 : K - 1];
                }
                const float v = min3f(top, left, tl) + dist(av, bv);
                p2[o] = valid ? v : kInf;
            }
            // primary, centre row advanced (dtw.cpp:430-436, 459-463): dp1 = the secondary (now p2),
            // dp0 = the previous primary (p1), in place over p1
#pragma unroll
            for (int o = 0; o < P; o++) {
                const int i = (int)col + HP - o;
                const int j = row - HP + o;
                const bool valid = (uint32_t)i < N && (uint32_t)j < M;
                const float av = aw[o];
                const float bv = bw[o + 1];
                if (SH == 0) {
                    const float top = (o == 0) ? kInf : p2[o > 0 ? o - 1 : 0];
                    const float tl = p1[o];
                    const float left = (o == P - 1) ? kInf : p2[o];
                    const float v = min3f(top, left, tl) + dist(av, bv);
                    p1[o] = valid ? v : kInf;
                } else {
                    constexpr int kmax = K - 1;
                    const float top = p2[o];
                    const float tl = p1[o + 1 < K ? o + 1 : kmax];
                    const float left = p2[o + 1 < K ? o + 1 : kmax];
                    const float v = min3f(top, left, tl) + dist(av, bv);
                    p1[o + 1 < K ? o + 1 : kmax] = valid ? v : kInf;
                }
            }
        } else {
            // primary, same centre row (dtw.cpp:437-444, 464-473): dp1 = p1, dp0 = p2; computed in place
            // over p2 from the highest offset down, then the two buffers trade places
#pragma unroll
            for (int o = P - 1; o >= 0; o--) {
                const int i = (int)col + HP - o;
                const int j = row - HP + o;
                const bool valid = (uint32_t)i < N && (uint32_t)j < M;
                const float av = aw[o];
                const float bv = bw[o + 1];
                if (SH == 0) {
                    const float top = (o == 0) ? kInf : p1[o > 0 ? o - 1 : 0];
                    const float tl = (o == 0) ? kInf : p2[o > 0 ? o - 1 : 0];
                    const float left = p1[o];
                    const float v = min3f(top, left, tl) + dist(av, bv);
                    p2[o] = valid ? v : kInf;
                } else {
                    constexpr int kmax = K - 1;
                    const float top = (o == 0) ? kInf : p1[o];
                    const float tl = (o == 0 && !prev_adv) ? kInf : p2[o];
                    const float left = p1[o + 1 < K ? o + 1 : kmax];
                    const float v = min3f(top, left, tl) + dist(av, bv);
                    p2[o + 1 < K ? o + 1 : kmax] = valid ? v : kInf;
                }
            }
#pragma unroll
            for (int x = 0; x < K; x++) { const float t = p1[x]; p1[x] = p2[x]; p2[x] = t; }
        }
        prev_adv = adv;
    }
    return p1[HP + SH]; // dtw.cpp:506-512
}

// The same DP for waves whose lanes do NOT share one shape.  There the lanes disagree on `adv` (does the centre row
// advance at this column?) at almost every column, and a branch on it makes the wave run both bodies of lane_dp --
// secondary + primary, and the lone primary -- every time.  This variant has one body: the secondary is always
// computed (its result only matters when the row advances), and the primary takes its neighbours through selects on
// `adv`; the two cases of lane_dp then differ in operands only:
//   X  = adv ? (p2 after the secondary) : p1     -- the antidiagonal just before this primary
//   tl = adv ? p1 : p2                           -- the one before that (shifted by one slot when the row stays)
// and both end with p1 = the new primary, p2 = X.  About 55 vector instructions per column against ~85 for the two
// bodies; a wave of one shape is still better off with lane_dp (scalar branch, ~47).
template <int R>
__device__ __forceinline__ float lane_dp_sel(const float *LA, const float *LB, const uint32_t N, const uint32_t M)
{
    constexpr int P = R + ((R % 2 == 0) ? 1 : 0); // dtw.cpp:301
    constexpr int S = R + ((R % 2 == 1) ? 1 : 0); // dtw.cpp:302
    constexpr int K = (P > S) ? P : S;
    constexpr int SH = (P > S) ? 0 : 1; // primaries live at index+1 when the secondary is longer
    constexpr int HP = P / 2, HS = S / 2;
    constexpr bool EVEN = (R % 2 == 0);
    const int iN = (int)N, iM = (int)M;

    float p1[K], p2[K];
    float aw[K];     // aw[x] = A[col + HP - x]
    float bw[K + 1]; // bw[x] = B[row - HP - 1 + x]
#pragma unroll
    for (int x = 0; x < K; x++) {
        p1[x] = kInf; p2[x] = kInf;
        const int ia = HP - x;
        aw[x] = LA[ia < 0 ? 0 : (ia >= iN ? iN - 1 : ia)];
    }
#pragma unroll
    for (int x = 0; x <= K; x++) {
        const int ib = x - HP - 1;
        bw[x] = LB[ib < 0 ? 0 : (ib >= iM ? iM - 1 : ib)];
    }
    p1[HP + SH] = dist(LA[0], LB[0]); // column 0: only the corner (dtw.cpp:317-347)

    int row = 0;
    uint32_t rem = 0;
    bool prev_adv = false;
    float a_next, b_next;
    {
        const int ia = 1 + HP, ib = 1 - HP - 1 + K;
        a_next = LA[ia >= iN ? iN - 1 : ia];
        b_next = LB[ib < 0 ? 0 : (ib >= iM ? iM - 1 : ib)];
    }
    for (uint32_t col = 1; col < N; col++) {
        rem += M;
        const bool adv = rem >= N;
        rem -= adv ? N : 0u;
        row += adv ? 1 : 0;
        // a-window: one step per column
#pragma unroll
        for (int x = K - 1; x > 0; x--) aw[x] = aw[x - 1];
        aw[0] = a_next;
        {
            const int ia = (int)col + 1 + HP;
            a_next = LA[ia >= iN ? iN - 1 : ia];
        }
        // b-window: one step when the row advances.  (b_next is a function of the row alone: reloading it every column
        // returns the same value while the row stays.)
#pragma unroll
        for (int x = 0; x < K; x++) bw[x] = adv ? bw[x + 1] : bw[x];
        bw[K] = adv ? b_next : bw[K];
        {
            const int ib = row - HP + K;
            b_next = LB[ib < 0 ? 0 : (ib >= iM ? iM - 1 : ib)];
        }
        // secondary antidiagonal (dtw.cpp:361-414), wanted only when the row advanced
        float q[K];
#pragma unroll
        for (int o = 0; o < K; o++) q[o] = p2[o];
#pragma unroll
        for (int o = 0; o < S; o++) {
            const int i = (int)col + HS - 1 - o;
            const int j = row - HS + o;
            const bool valid = (uint32_t)i < N && (uint32_t)j < M;
            const float av = EVEN ? aw[o + 1 < K ? o + 1 : K - 1] : aw[o];
            const float bv = EVEN ? bw[o + 1] : bw[o];
            float top, tl, left;
            if (SH == 0) {
                top = p1[o]; tl = p2[o]; left = p1[o + 1 < K ? o + 1 : K - 1];
            } else {
                top = (o == 0) ? kInf : p1[o];
                tl = (o == 0 && !prev_adv) ? kInf : p2[o];
                left = (o == S - 1) ? kInf : p1[o + 1 < K ? o + 1 : K - 1];
            }
            const float v = min3f(top, left, tl) + dist(av, bv);
            q[o] = valid ? v : kInf;
        }
        float X[K];
#pragma unroll
        for (int x = 0; x < K; x++) X[x] = adv ? q[x] : p1[x];
        // primary antidiagonal (dtw.cpp:416-485)
        float nw[P];
#pragma unroll
        for (int o = 0; o < P; o++) {
            const int i = (int)col + HP - o;
            const int j = row - HP + o;
            const bool valid = (uint32_t)i < N && (uint32_t)j < M;
            float top, left, tl;
            if (SH == 0) {
                top = (o == 0) ? kInf : X[o > 0 ? o - 1 : 0];
                left = (o == P - 1) ? (adv ? kInf : X[o]) : X[o];
                tl = adv ? p1[o] : ((o == 0) ? kInf : p2[o > 0 ? o - 1 : 0]);
            } else {
                top = (o == 0) ? (adv ? X[0] : kInf) : X[o];
                left = X[o + 1 < K ? o + 1 : K - 1];
                const float tl_stay = (o == 0 && !prev_adv) ? kInf : p2[o];
                tl = adv ? p1[o + 1 < K ? o + 1 : K - 1] : tl_stay;
            }
            const float v = min3f(top, left, tl) + dist(aw[o], bw[o + 1]);
            nw[o] = valid ? v : kInf;
        }
        if (SH == 0) {
#pragma unroll
            for (int o = 0; o < P; o++) p1[o] = nw[o];
        } else {
            p1[0] = adv ? p1[0] : p2[0];
#pragma unroll
            for (int o = 0; o < P; o++) p1[o + 1 < K ? o + 1 : K - 1] = nw[o];
        }
#pragma unroll
        for (int x = 0; x < K; x++) p2[x] = X[x];
        prev_adv = adv;
    }
    return p1[HP + SH]; // dtw.cpp:506-512
}

// One body for every radius 1..3 and every shape, for waves whose lanes hold DIFFERENT radii (the tile kernel of the
// sync-free path sorts its lane-class jobs by longer side only: radii 1 and 3 are too rare inside a tile to fill waves of
// their own -- a quarter-full wave costs as much as a full one).  Physical-slot form of dtw.cpp:305-491 with K = 4 slots
// in registers: slot p of a secondary antidiagonal is cell (col - 1 + off - p, row - off + p), of a primary
// (col + off - p, row - off + p), off = P/2 + SH; primaries live at slot o + SH.  Two buffers: d1 = latest antidiagonal,
// d0 = the one before; every column computes the secondary (its result only counts when the centre row advances) and
// feeds the primary through selects on `adv` (see lane_dp_sel).
//
// What makes it cheap: NO per-cell validity.  (a) Slots outside an antidiagonal's extent (p >= S, o outside [0, P)) and
// cells beyond the matrix on the HIGH side (i >= N or j >= M) may hold anything: a cell reads (i-1, j), (i, j-1),
// (i-1, j-1) only, so a cell inside the matrix never reads one beyond it, and every read of a slot outside the extent
// is one of the reference's guarded reads (is_first / is_last / previous_increment_center_row, dtw.cpp:373-375, 392-397,
// 428-441, 461-473), reproduced below.  (b) Cells beyond the matrix on the LOW side (i < 0 or j < 0) must read as 1e10:
// they exist only while row < off, i.e. in the first columns of a job, which run in a masked copy of the step; the wave
// leaves it as soon as no lane needs it.  Operands beyond a window's end are read unclamped: they feed only cells beyond
// the matrix (the tile's LDS image has slack behind the last window).
struct GenLane {
    float d0[4], d1[4], ap[4], bp[4];
    float a_next, b_next, res;
    uint32_t rem;
    int row;
    lmask prev_adv;
};

template <bool MASKED, bool CLAMP>
__device__ __forceinline__ void lane_gen_step(GenLane &g, const float *LA, const float *LB, const uint32_t N, const uint32_t M,
                                              const int off, const lmask sh, const lmask r1, const lmask r2, const uint32_t col)
{
    // (every per-lane select on a lane mask, by selm: see there)
    g.rem += M;
    const lmask adv = __ballot(g.rem >= N);
    g.rem = wrap_sub(g.rem, N);
    g.row = (int)selm(adv, (uint32_t)g.row + 1u, (uint32_t)g.row);
    // b-window: one step when the row advances (b_next is a function of the row alone)
    float bn[4];
#pragma unroll
    for (int p = 0; p < 3; p++) bn[p] = selm(adv, g.bp[p + 1], g.bp[p]);
    bn[3] = selm(adv, g.b_next, g.bp[3]);
    g.b_next = LB[CLAMP ? min(g.row + 4 - off, (int)M - 1) : g.row + 4 - off];
    // secondary antidiagonal (dtw.cpp:361-414): a-window of the previous column, b-window of the new row
    float X[4];
    {
        const float top0 = selm(sh, kInf, g.d1[0]);                       // is_first
        const float tl0 = selm(sh & ~g.prev_adv, kInf, g.d0[0]);          // previous_increment_center_row
        const float left1 = selm(r1, kInf, g.d1[2]);                      // is_last (radius 1: S - 1 = 1; radius 3: slot 3 below)
        float sec[4];
        sec[0] = min3f(top0, g.d1[1], tl0) + dist(g.ap[0], bn[0]);
        sec[1] = min3f(g.d1[1], left1, g.d0[1]) + dist(g.ap[1], bn[1]);
        sec[2] = min3f(g.d1[2], g.d1[3], g.d0[2]) + dist(g.ap[2], bn[2]);
        sec[3] = min3f(g.d1[3], kInf, g.d0[3]) + dist(g.ap[3], bn[3]);
        if (MASKED) { // low side: j = row - off + p >= 0 and i = col - 1 + off - p >= 0
#pragma unroll
            for (int p = 0; p < 4; p++) sec[p] = selm(__ballot(p < off - g.row || p > (int)col - 1 + off), kInf, sec[p]);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) X[p] = selm(adv, sec[p], g.d1[p]); // the antidiagonal just before this column's primary
    }
    // a-window: one step per column
    g.ap[3] = g.ap[2]; g.ap[2] = g.ap[1]; g.ap[1] = g.ap[0]; g.ap[0] = g.a_next;
    g.a_next = LA[CLAMP ? min(col + 1u + (uint32_t)off, N - 1u) : col + 1u + (uint32_t)off];
    // primary antidiagonal (dtw.cpp:416-485)
    {
        const float top1 = selm(sh & ~adv, kInf, X[0]);                   // o == 0 of an odd radius when the row stays
        const float left2 = selm(r2 & adv, kInf, X[2]);                   // last offset of an even radius after a secondary
        const float t1 = selm(g.prev_adv, g.d0[0], kInf);                 // (o == 0, odd radius: only after an advance)
        const float tl0 = selm(adv, g.d1[0], kInf);
        const float tl1 = selm(adv, g.d1[1], selm(sh, t1, g.d0[0]));
        const float tl2 = selm(adv, g.d1[2], g.d0[1]);
        const float tl3 = selm(adv, g.d1[3], g.d0[2]);
        float pr[4];
        pr[0] = min3f(kInf, X[0], tl0) + dist(g.ap[0], bn[0]);
        pr[1] = min3f(top1, X[1], tl1) + dist(g.ap[1], bn[1]);
        pr[2] = min3f(X[1], left2, tl2) + dist(g.ap[2], bn[2]);
        pr[3] = min3f(X[2], X[3], tl3) + dist(g.ap[3], bn[3]);
        if (MASKED) { // low side: j = row - off + p >= 0 (i = col + off - p >= 0 for every slot of the antidiagonal)
#pragma unroll
            for (int p = 0; p < 4; p++) pr[p] = selm(__ballot(p < off - g.row), kInf, pr[p]);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) { g.d0[p] = X[p]; g.d1[p] = pr[p]; g.bp[p] = bn[p]; }
    }
    g.prev_adv = adv;
    // dtw.cpp:506-512: the centre of the last primary
    const float centre = selm(__ballot(off == 1), g.d1[1], g.d1[2]);
    g.res = selm(__ballot(col == N - 1u), centre, g.res);
}

// N, M, R per lane (R in 1..3, N >= M, N >= 2); n_max = the largest N of the wave.  Lanes past their last column keep
// stepping on values nobody reads.  CLAMP: the windows lie in the arenas, not in a tile's image with slack behind it --
// reads past a window's end go to its last element instead.
template <bool CLAMP = false>
__device__ __forceinline__ float lane_dp_gen(const float *LA, const float *LB, const uint32_t N, const uint32_t M, const uint32_t R,
                                             const uint32_t n_max)
{
    const lmask sh = __ballot((R & 1u) != 0u); // odd radius: P = R, S = R + 1, primaries at slot o + 1 (dtw.cpp:301-303, 459, 479)
    const int off = (int)((R + 1u) >> 1);   // P/2 + SH, the centre slot: radius 1 -> 1, 2 -> 1, 3 -> 2
    const lmask r1 = __ballot(R == 1u), r2 = __ballot(R == 2u);
    const int iN = (int)N, iM = (int)M;
    GenLane g;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        g.d0[p] = kInf; g.d1[p] = kInf;
        const int ia = off - p, ib = p - off;
        g.ap[p] = LA[ia < 0 ? 0 : (ia >= iN ? iN - 1 : ia)];
        g.bp[p] = LB[ib < 0 ? 0 : (ib >= iM ? iM - 1 : ib)];
    }
    {   // column 0: only the corner (dtw.cpp:317-347), at the centre slot
        const float c = dist(LA[0], LB[0]);
        g.d1[1] = off == 1 ? c : kInf;
        g.d1[2] = off == 2 ? c : kInf;
    }
    g.res = g.d1[off == 1 ? 1 : 2]; // (N == 1 cannot occur here; kept for completeness)
    g.rem = 0; g.row = 0; g.prev_adv = 0;
    g.a_next = LA[CLAMP ? min(1 + off, iN - 1) : 1 + off];
    g.b_next = LB[CLAMP ? min(4 - off, iM - 1) : 4 - off];
    uint32_t col = 1;
    // first columns: some lane still has cells above row 0 (or left of column 0) inside its band
    for (; col < n_max && __any((g.row < off || col < 3u) && col < N); col++) lane_gen_step<true, CLAMP>(g, LA, LB, N, M, off, sh, r1, r2, col);
    for (; col + 1u < n_max; col += 2u) { // (two columns a round: the window and antidiagonal moves of one fold into the other's operands)
        lane_gen_step<false, CLAMP>(g, LA, LB, N, M, off, sh, r1, r2, col);
        lane_gen_step<false, CLAMP>(g, LA, LB, N, M, off, sh, r1, r2, col + 1u);
    }
    if (col < n_max) lane_gen_step<false, CLAMP>(g, LA, LB, N, M, off, sh, r1, r2, col);
    return g.res;
}

// Radius 3 with the band's four slots over the four lanes of a QUAD: sixteen jobs a wave.  A tile holds some ten radius-3
// parts (read side of 20..39 events): one a lane they fill a sixth of a wave, which then runs the four-slot body above for
// the longest of them -- the tile's slowest wave.  Here lane p of a quad keeps slot p of the job's two antidiagonals (d0,
// d1), of its a-window and of its b-window; a column costs a lane one secondary and one primary cell, the neighbours'
// values come by DPP quad permutes.  Same cells, same operands and the same guarded reads as lane_gen_step with
// sh = 1, off = 2, r1 = r2 = false (R = 3: P = 3 primaries at slots 1..3, S = 4 secondaries):
//     sec[p] = min3(p == 0 ? 1e10 : d1[p],  p == 3 ? 1e10 : d1[p + 1],  p == 0 && !prev_adv ? 1e10 : d0[p]) + |ap[p] - bn[p]|
//     pr[p]  = min3(p == 0 || (p == 1 && !adv) ? 1e10 : X[p - 1],  X[p],
//                   adv ? d1[p] : (p == 0 || (p == 1 && !prev_adv) ? 1e10 : d0[p - 1]))             + |ap'[p] - bn[p]|
// The job's bookkeeping (rem, row, the operand loads) runs in all four lanes alike.  The result is the centre slot's:
// lane 2 of the quad.
__device__ __forceinline__ float quad_up(float v) // lane p of a quad: lane p + 1's value (lane 3: its own)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xF9, 0xf, 0xf, false)); // quad_perm:[1,2,3,3]
}
__device__ __forceinline__ float quad_dn(float v) // lane p of a quad: lane p - 1's value (lane 0: its own)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x90, 0xf, 0xf, false)); // quad_perm:[0,0,1,2]
}

struct QuadLane {
    float d0, d1, ap, bp;
    float a_next, b_next, res;
    uint32_t rem;
    int row;
    lmask prev_adv;
};

template <bool MASKED>
__device__ __forceinline__ void quad_r3_step(QuadLane &g, const float *LA, const float *LB, const uint32_t N, const uint32_t M, const int p,
                                             const lmask is0, const lmask is1, const lmask is3, const uint32_t col)
{
    g.rem += M;
    const lmask adv = __ballot(g.rem >= N);
    g.rem = wrap_sub(g.rem, N);
    g.row = (int)selm(adv, (uint32_t)g.row + 1u, (uint32_t)g.row);
    // b-window: one step when the row advances
    const float b_up = quad_up(g.bp);
    const float bn = selm(adv, selm(is3, g.b_next, b_up), g.bp);
    g.b_next = LB[g.row + 2];
    // secondary antidiagonal (dtw.cpp:361-414): a-window of the previous column
    const float d1_up = quad_up(g.d1);
    float sec = min3f(selm(is0, kInf, g.d1), selm(is3, kInf, d1_up), selm(is0 & ~g.prev_adv, kInf, g.d0)) + dist(g.ap, bn);
    if (MASKED) sec = selm(__ballot(p < 2 - g.row || p > (int)col + 1), kInf, sec); // low side: j = row - 2 + p >= 0 and i = col + 1 - p >= 0
    const float X = selm(adv, sec, g.d1); // the antidiagonal just before this column's primary
    // a-window: one step per column
    const float a_dn = quad_dn(g.ap);
    g.ap = selm(is0, g.a_next, a_dn);
    g.a_next = LA[col + 3u];
    // primary antidiagonal (dtw.cpp:416-485)
    const float X_dn = quad_dn(X), d0_dn = quad_dn(g.d0);
    const float top = selm(is0 | (is1 & ~adv), kInf, X_dn);
    const float stay = selm(is0 | (is1 & ~g.prev_adv), kInf, d0_dn);
    float pr = min3f(top, X, selm(adv, g.d1, stay)) + dist(g.ap, bn);
    if (MASKED) pr = selm(__ballot(p < 2 - g.row), kInf, pr); // low side: j = row - 2 + p >= 0
    g.d0 = X; g.d1 = pr; g.bp = bn;
    g.prev_adv = adv;
    g.res = selm(__ballot(col == N - 1u), pr, g.res); // (dtw.cpp:506-512: the centre of the last primary, in the quad's lane 2)
}

// N, M of the quad's job in all four of its lanes (R = 3: N >= M, N >= 20); n_max = the largest N of the wave.  The result is
// valid in lane 2 of the quad.  Operands beyond a window's end are read unclamped (a tile's image has slack behind it).
__device__ __forceinline__ float quad_dp_r3(const float *LA, const float *LB, const uint32_t N, const uint32_t M, const int lane,
                                            const uint32_t n_max)
{
    const int p = lane & 3;
    const lmask is0 = __ballot(p == 0), is1 = __ballot(p == 1), is3 = __ballot(p == 3);
    const int iN = (int)N, iM = (int)M;
    QuadLane g;
    g.d0 = kInf;
    {
        const int ia = 2 - p, ib = p - 2;
        g.ap = LA[ia < 0 ? 0 : (ia >= iN ? iN - 1 : ia)];
        g.bp = LB[ib < 0 ? 0 : (ib >= iM ? iM - 1 : ib)];
    }
    g.d1 = p == 2 ? dist(LA[0], LB[0]) : kInf; // column 0: only the corner (dtw.cpp:317-347), at the centre slot
    g.res = g.d1;
    g.rem = 0; g.row = 0; g.prev_adv = 0;
    g.a_next = LA[3];
    g.b_next = LB[2];
    uint32_t col = 1;
    // first columns: some job still has cells above row 0 (or left of column 0) inside its band
    for (; col < n_max && __any((g.row < 2 || col < 3u) && col < N); col++) quad_r3_step<true>(g, LA, LB, N, M, p, is0, is1, is3, col);
    for (; col + 1u < n_max; col += 2u) {
        quad_r3_step<false>(g, LA, LB, N, M, p, is0, is1, is3, col);
        quad_r3_step<false>(g, LA, LB, N, M, p, is0, is1, is3, col + 1u);
    }
    if (col < n_max) quad_r3_step<false>(g, LA, LB, N, M, p, is0, is1, is3, col);
    return g.res;
}

// One lane per job for bands of up to 8 slots (radius <= 7), any radius mix in a wave: the scheme of wreg_gen_step with
// the band's slots in the lane's own registers instead of across lanes.  Physical slot p of a secondary antidiagonal is
// cell (col - 1 + off - p, row - off + p), of a primary (col + off - p, row - off + p); primaries live at slots SH ..
// SH + P - 1, secondaries at 0 .. S - 1 (dtw.cpp:301-303, 459, 479).  No guards: a slot outside its antidiagonal's extent
// holds 1e10 (one select per slot on a lane-constant flag), which is what every guarded read of the reference yields
// (argument at wreg_gen_step); no per-cell validity on the high side (lane_dp_gen); the low side in a masked copy of
// the step for the first columns.  The row advance is per lane: selects, not branches (X = adv ? secondary : previous
// primary).  Operand reads are clamped into the windows: they come from the arenas, not from a tile's image.
// ~130 VALU per column for 64 jobs, against ~75 per column for the 8 jobs of grp_wave<8>.
struct K8Lane {
    float d0[8], d1[8], ap[8], bp[8];
    uint32_t rem;
    int row;
};

template <bool MASKED>
__device__ __forceinline__ void lane_k8_step(K8Lane &g, const float *A, const float *B, const uint32_t N, const uint32_t M,
                                             const int off, const bool (&in_sec)[8], const bool (&in_prim)[8], const uint32_t col)
{
    g.rem += M;
    const bool adv = g.rem >= N;
    g.rem -= adv ? N : 0u;
    g.row += adv ? 1 : 0;
    // b-window: every slot takes its right neighbour's value when the row advances.  (The fresh operands are loaded where
    // they are used: one column ahead costs two more registers, which this body -- the kernel's widest -- does not have
    // without spilling; its waves are few and run beside the tiles.)
    const float fresh_b = B[min(max(g.row - off + 7, 0), (int)M - 1)];
    float bn[8];
#pragma unroll
    for (int p = 0; p < 7; p++) bn[p] = adv ? g.bp[p + 1] : g.bp[p];
    bn[7] = adv ? fresh_b : g.bp[7];
    // secondary antidiagonal (dtw.cpp:361-414): a-window of the previous column; counts only after an advance
    float X[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const float left = p < 7 ? g.d1[p < 7 ? p + 1 : p] : kInf;
        float v = min3f(g.d1[p], left, g.d0[p]) + dist(g.ap[p], bn[p]);
        bool keep = in_sec[p];
        if (MASKED) keep = keep && !(p < off - g.row || p > (int)col - 1 + off);
        v = keep ? v : kInf;
        X[p] = adv ? v : g.d1[p];
    }
    // a-window: every slot takes its left neighbour's value
    const float fresh_a = A[min(col + (uint32_t)off, N - 1u)];
#pragma unroll
    for (int p = 7; p > 0; p--) g.ap[p] = g.ap[p - 1];
    g.ap[0] = fresh_a;
    // primary antidiagonal (dtw.cpp:416-485): after a secondary (top, left, diagonal) = (X[p-1], X[p], d1[p]); otherwise X is
    // the primary before this one and they are (X[p-1], X[p], d0[p-1])
    float pr[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const float top = p > 0 ? X[p > 0 ? p - 1 : 0] : kInf;
        const float tl_stay = p > 0 ? g.d0[p > 0 ? p - 1 : 0] : kInf;
        const float tl = adv ? g.d1[p] : tl_stay;
        float v = min3f(top, X[p], tl) + dist(g.ap[p], bn[p]);
        bool keep = in_prim[p];
        if (MASKED) keep = keep && !(p < off - g.row);
        pr[p] = keep ? v : kInf;
    }
#pragma unroll
    for (int p = 0; p < 8; p++) { g.d0[p] = X[p]; g.d1[p] = pr[p]; g.bp[p] = bn[p]; }
}

// N >= M (the caller swaps), R in 1..7; n_max = the largest N of the wave.
__device__ __forceinline__ float lane_dp_k8(const float *A, const float *B, const uint32_t N, const uint32_t M, const int R,
                                            const uint32_t n_max)
{
    const int P = R + ((R % 2 == 0) ? 1 : 0);
    const int S = R + ((R % 2 == 1) ? 1 : 0);
    const int SH = P > S ? 0 : 1;
    const int off = P / 2 + SH, K = P > S ? P : S;
    const int iN = (int)N, iM = (int)M;
    K8Lane g;
    bool in_sec[8], in_prim[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        g.d0[p] = kInf; g.d1[p] = kInf;
        g.ap[p] = A[min(max(off - p, 0), iN - 1)];
        g.bp[p] = B[min(max(p - off, 0), iM - 1)];
        in_sec[p] = p < S;
        in_prim[p] = p >= SH && p < SH + P;
    }
    const float corner = dist(A[0], B[0]); // dtw.cpp:317-347, at the centre slot
#pragma unroll
    for (int p = 1; p <= 4; p++) g.d1[p] = (p == off) ? corner : kInf;
    g.rem = 0; g.row = 0;
    float res = corner;
    auto centre = [&]() { return off == 1 ? g.d1[1] : off == 2 ? g.d1[2] : off == 3 ? g.d1[3] : g.d1[4]; }; // dtw.cpp:506-512
    uint32_t col = 1;
    // first columns: cells above row 0 or left of column 0 exist while row < off or col - 1 + off < K - 1
    for (; col < n_max && __any((g.row < off || (int)col < K - off + 1) && col < N); col++) {
        lane_k8_step<true>(g, A, B, N, M, off, in_sec, in_prim, col);
        res = (col == N - 1u) ? centre() : res;
    }
    for (; col < n_max; col++) {
        lane_k8_step<false>(g, A, B, N, M, off, in_sec, in_prim, col);
        res = (col == N - 1u) ? centre() : res;
    }
    return res;
}

// Radii 1 and 2 mixed in a wave -- 98 % of a sparse batch's jobs (radius 1 = square parts, radius 2 = every other part
// whose read side is under 20 events): the generic body with K = 3 slots and off = 1 for both radii.  Radius 2 has
// S = 2 secondaries (slots 0, 1) and P = 3 primaries (slots 0..2, no shift); radius 1 has S = 2 and P = 1 at slot 1.
// Dropped against the 4-slot body: slot 3, d0[2] (read by slot 3 only), the secondaries' low-side mask (a secondary
// counts only after a row advance, when row >= 1 = off), and the masked copy of the loop -- the one low-side cell left,
// primary slot 0 = (col + 1, row - 1) while row == 0, is a select in the loop.  X[2] = d1[2] whether or not the row
// advances: after an advance it is read behind the is_last guard only (left2).
__device__ __forceinline__ float lane_dp_r12(const float *LA, const float *LB, const uint32_t N, const uint32_t M, const uint32_t R,
                                             const uint32_t n_max)
{
    const lmask r1 = __ballot(R == 1u);
    const int iM = (int)M;
    float d00 = kInf, d01 = kInf;                       // the antidiagonal before the latest: slots 0, 1
    float d10 = kInf, d11 = dist(LA[0], LB[0]), d12 = kInf; // the latest: the corner at the centre slot (dtw.cpp:317-347)
    float ap0 = LA[1], ap1 = LA[0], ap2 = LA[0];
    float bp0 = LB[0], bp1 = LB[0], bp2 = LB[iM > 1 ? 1 : 0];
    float a_next = LA[2], b_next = LB[2];
    float res = d11;
    uint32_t rem = 0, row = 0;
    lmask prev_adv = 0;
    auto step = [&](const uint32_t col) {
        rem += M;
        const lmask adv = __ballot(rem >= N);
        rem = wrap_sub(rem, N);
        row = selm(adv, row + 1u, row);
        const float bn0 = selm(adv, bp1, bp0), bn1 = selm(adv, bp2, bp1), bn2 = selm(adv, b_next, bp2);
        b_next = LB[row + 2];
        // secondary antidiagonal (dtw.cpp:361-414): a-window of the previous column
        const float top0 = selm(r1, kInf, d10);                   // is_first (odd radius)
        const float tl0s = selm(r1 & ~prev_adv, kInf, d00);       // previous_increment_center_row
        const float left1 = selm(r1, kInf, d12);                  // is_last (odd radius)
        const float sec0 = min3f(top0, d11, tl0s) + dist(ap0, bn0);
        const float sec1 = min3f(d11, left1, d01) + dist(ap1, bn1);
        const float X0 = selm(adv, sec0, d10), X1 = selm(adv, sec1, d11);
        ap2 = ap1; ap1 = ap0; ap0 = a_next;
        a_next = LA[col + 2];
        // primary antidiagonal (dtw.cpp:416-485)
        const float top1 = selm(r1 & ~adv, kInf, X0);
        const float left2 = selm(~r1 & adv, kInf, d12);
        const float tl0 = selm(adv, d10, kInf);
        const float tl1 = selm(adv, d11, tl0s);                   // (odd radius, row stays: d0[0] only after an advance)
        const float tl2 = selm(adv, d12, d01);
        float pr0 = min3f(kInf, X0, tl0) + dist(ap0, bn0);
        const float pr1 = min3f(top1, X1, tl1) + dist(ap1, bn1);
        const float pr2 = min3f(X1, left2, tl2) + dist(ap2, bn2);
        pr0 = selm(__ballot(row == 0u), kInf, pr0);               // (col + 1, -1): above the matrix
        d00 = X0; d01 = X1; d10 = pr0; d11 = pr1; d12 = pr2;
        bp0 = bn0; bp1 = bn1; bp2 = bn2;
        prev_adv = adv;
        res = selm(__ballot(col == N - 1u), d11, res);
    };
    uint32_t col = 1;
    for (; col + 1u < n_max; col += 2u) { step(col); step(col + 1u); } // (two columns a round: the moves of one fold into the other's operands)
    if (col < n_max) step(col);
    return res;
}

// Radius 2 in every lane of the wave -- two thirds of a sparse batch's jobs and three quarters of its cells (every
// non-square part whose read side is under 20 events): the body above with R = 2 folded in (S = 2 secondaries at slots 0
// and 1, P = 3 primaries at slots 0..2, no shift): the selects on the radius drop out (is_first / is_last belong to odd
// radii, previous_increment_center_row to radius 1), the ones on the per-lane row advance stay.
__device__ __forceinline__ float lane_dp_r2(const float *LA, const float *LB, const uint32_t N, const uint32_t M, const uint32_t n_max)
{
    const int iM = (int)M;
    float d00 = kInf, d01 = kInf;                           // the antidiagonal before the latest: slots 0, 1
    float d10 = kInf, d11 = dist(LA[0], LB[0]), d12 = kInf; // the latest: the corner at the centre slot (dtw.cpp:317-347)
    float ap0 = LA[1], ap1 = LA[0], ap2 = LA[0];
    float bp0 = LB[0], bp1 = LB[0], bp2 = LB[iM > 1 ? 1 : 0];
    float a_next = LA[2], b_next = LB[2];
    float res = d11;
    uint32_t rem = 0, row = 0;
    auto step = [&](const uint32_t col) {
        rem += M;
        const lmask adv = __ballot(rem >= N);
        rem = wrap_sub(rem, N);
        row = selm(adv, row + 1u, row);
        const float bn0 = selm(adv, bp1, bp0), bn1 = selm(adv, bp2, bp1), bn2 = selm(adv, b_next, bp2);
        b_next = LB[row + 2];
        // secondary antidiagonal (dtw.cpp:361-414): a-window of the previous column
        const float sec0 = min3f(d10, d11, d00) + dist(ap0, bn0);
        const float sec1 = min3f(d11, d12, d01) + dist(ap1, bn1);
        const float X0 = selm(adv, sec0, d10), X1 = selm(adv, sec1, d11);
        ap2 = ap1; ap1 = ap0; ap0 = a_next;
        a_next = LA[col + 2];
        // primary antidiagonal (dtw.cpp:416-485)
        const float left2 = selm(adv, kInf, d12);
        const float tl0 = selm(adv, d10, kInf);
        const float tl1 = selm(adv, d11, d00);
        const float tl2 = selm(adv, d12, d01);
        float pr0 = min3f(kInf, X0, tl0) + dist(ap0, bn0);
        const float pr1 = min3f(X0, X1, tl1) + dist(ap1, bn1);
        const float pr2 = min3f(X1, left2, tl2) + dist(ap2, bn2);
        pr0 = selm(__ballot(row == 0u), kInf, pr0);               // (col + 1, -1): above the matrix
        d00 = X0; d01 = X1; d10 = pr0; d11 = pr1; d12 = pr2;
        bp0 = bn0; bp1 = bn1; bp2 = bn2;
        res = selm(__ballot(col == N - 1u), d11, res);
    };
    uint32_t col = 1;
    for (; col + 1u < n_max; col += 2u) { step(col); step(col + 1u); } // (two columns a round: the moves of one fold into the other's operands)
    if (col < n_max) step(col);
    return res;
}

// Radius 1 in every lane of the wave -- 97 % of a sparse batch's jobs.  Radius 1 means a SQUARE part: the reference's
// radius is r0 + ((N - M) * r0 + N - 1) / N with r0 >= 1 (dtw.cpp:298-300), which is 1 only for r0 = 1 and N = M.  On a
// square the centre row advances with every column (rem += M reaches N each time), so the body above with R = 1 folded in
// (sh = 1, off = 1, S = 2 secondaries at slots 0 and 1, one primary at slot 1) loses its row bookkeeping and every select
// on `adv`: per column the two secondaries (col, col - 1) and (col - 1, col), then the primary (col, col).
// State: prim = d1[1]; x0, x1 = d0[0], d0[1] (the secondaries of the column before).  x0 is read behind
// previous_increment_center_row (dtw.cpp:373-375, 392-397), false only in column 1 -- where x0 still holds 1e10, the value
// the guard yields.  The 1e10 operands of the reference's guarded reads stay in the min3s (is_first: no top; is_last: no
// left), so every value is the one the generic body computes.  About 15 instructions per column against 30.
__device__ __forceinline__ float lane_dp_r1(const float *LA, const float *LB, const uint32_t N, const uint32_t n_max)
{
    float prim = dist(LA[0], LB[0]); // the corner (dtw.cpp:317-347)
    float x0 = kInf, x1 = kInf;
    float ap0 = LA[1], ap1 = LA[0];
    float bp1 = LB[0];
    float a_next = LA[2], b_next = LB[1];
    float res = prim;
    auto step = [&](const uint32_t col) {
        const float bn0 = bp1, bn1 = b_next;                         // the b-window moves with the row: every column
        b_next = LB[col + 1];
        const float sec0 = min3f(kInf, prim, x0) + dist(ap0, bn0);   // (col, col - 1): no top (is_first)
        const float sec1 = min3f(prim, kInf, x1) + dist(ap1, bn1);   // (col - 1, col): no left (is_last)
        ap1 = ap0; ap0 = a_next;
        a_next = LA[col + 2];
        const float pr1 = min3f(sec0, sec1, prim) + dist(ap1, bn1);  // (col, col)
        x0 = sec0; x1 = sec1; prim = pr1; bp1 = bn1;
        res = selm(__ballot(col == N - 1u), prim, res);
    };
    uint32_t col = 1;
    for (; col + 1u < n_max; col += 2u) { step(col); step(col + 1u); } // (two columns a round: the moves of one fold into the other's operands)
    if (col < n_max) step(col);
    return res;
}

// Micro path for the shapes that dominate sparse mode (longer side <= W, W = 4 or 8): the whole
// band fits a W x W grid, so the DP runs row by row over W statically indexed registers and band
// membership comes from a per-shape bitmask the planner computed by walking the reference's
// antidiagonals once (bit 8*j + i <=> cell (i over the longer sequence, j over the shorter) is in the
// band's cell set).  Same values as the antidiagonal order: a cell is min3 of its in-band neighbours
// (absent = 1e10) plus its distance, whatever the evaluation order.
template <int W, int NC>
__device__ __forceinline__ float micro_job_cols(const float *LA, const float *LB, const uint32_t N, const uint32_t M,
                                                const unsigned long long mask)
{
    // NC = columns any lane of the wave needs (its longest job): the grid is W rows x NC columns
    float a[NC], v[NC];
#pragma unroll
    for (int i = 0; i < NC; i++) { a[i] = LA[i]; v[i] = kInf; } // columns >= N are never in the mask
#pragma unroll
    for (int j = 0; j < NC; j++) {              // (M <= N <= NC: rows beyond NC do not exist either)
        if (__any((uint32_t)j < M)) {           // wave-uniform: skip rows no lane needs
            if ((uint32_t)j < M) {              // per lane
                const float bj = LB[j];
                const uint32_t rowmask = (uint32_t)(mask >> (8 * j)) & 0xffu;
                float diag = (j == 0) ? 0.0f : kInf; // virtual corner: D[0][0] = 0 + dist
                float left = kInf;
#pragma unroll
                for (int i = 0; i < NC; i++) {
                    const float up = v[i];
                    const float val = min3f(up, left, diag) + dist(a[i], bj);
                    const float keep = (rowmask & (1u << i)) ? val : kInf;
                    diag = up; left = keep; v[i] = keep;
                }
            }
        }
    }
    float res = v[0];
#pragma unroll
    for (int i = 1; i < NC; i++) res = (N - 1 == (uint32_t)i) ? v[i] : res;
    return res;
}

// ---------------------------------------------------------------------------------------------
// Register-resident wave-per-job banded kernel, band offsets laid across lanes (physical buffer
// index p = c*64 + lane, C registers per lane per buffer, radius + 1 <= 64*C).  The three
// rotating antidiagonal buffers, and the operand windows, never leave registers: a neighbour at
// p-1 / p+1 is a DPP wave shift (plus one v_readlane to carry across a 64-lane boundary), the
// a-window slides one lane per column, the b-window one lane per centre-row advance, and the
// single fresh value each of them needs comes out of a 64-wide chunk prefetched one chunk ahead
// -- so no memory access sits on the antidiagonal-to-antidiagonal dependency chain.
// Same physical indexing, guards and stale-slot behaviour as dtw.cpp:305-491.
// ---------------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void wreg_body(const DevJob &jb, const int lane, const float *__restrict__ ev,
                                          const float *__restrict__ ref, float *__restrict__ out)
{
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
    const int SH = P > S ? 0 : 1;
    const int HP = P / 2;
    const int iN = (int)N, iM = (int)M;
    auto ldA = [&](int i) { return A[i < 0 ? 0 : (i >= iN ? iN - 1 : i)]; };
    auto ldB = [&](int i) { return B[i < 0 ? 0 : (i >= iM ? iM - 1 : i)]; };

    float d0[C], d1[C], d2[C], ap[C], bp[C];
    // windows at column 0 / row 0:  ap(p) = A[col + HP + SH - p],  bp(p) = B[row - HP - SH + p]
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int p = c * 64 + lane;
        d0[c] = kInf; d1[c] = kInf; d2[c] = kInf;
        ap[c] = ldA(HP + SH - p);
        bp[c] = ldB(p - HP - SH);
        if (p == HP + SH) d1[c] = dist(A[0], B[0]); // the corner, already rotated into place
    }
    // fresh-value chunks: column col needs A[col + HP + SH]; the r-th row advance needs B[b0 + r - 1]
    const int a0 = HP + SH + 1, b0 = 64 * C - HP - SH;
    float acur = ldA(a0 + lane), anxt = ldA(a0 + 64 + lane);
    float bcur = ldB(b0 + lane), bnxt = ldB(b0 + 64 + lane);

    int row = 0;
    uint32_t rem = 0;
    bool prev_adv = false;
    for (uint32_t col = 1; col < N; col++) {
        rem += M;
        const bool adv = rem >= N;
        const uint32_t ca = (col - 1) & 63u;
        if (ca == 0 && col > 1) { acur = anxt; anxt = ldA(a0 + (int)(col - 1) + 64 + lane); }
        const float fresh_a = read_lane(acur, (int)ca);
        if (adv) {
            rem -= N;
            row++;
            const uint32_t cb = (uint32_t)(row - 1) & 63u;
            if (cb == 0 && row > 1) { bcur = bnxt; bnxt = ldB(b0 + (row - 1) + 64 + lane); }
            const float fresh_b = read_lane(bcur, (int)cb);
            // b-window: every lane takes its right neighbour's value
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float fill = (c + 1 < C) ? read_lane(bp[c + 1 < C ? c + 1 : c], 0) : fresh_b;
                bp[c] = wave_shl1(bp[c], fill);
            }
            // secondary antidiagonal (dtw.cpp:361-414): cell of lane p is (si - p, sj + p)
            const int si = (int)col - 1 + HP + SH, sj = row - HP - SH;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int p = c * 64 + lane;
                const float fill = (c + 1 < C) ? read_lane(d1[c + 1 < C ? c + 1 : c], 0) : kInf;
                float left = wave_shl1(d1[c], fill); // dp1[p+1]
                float top = d1[c], tl = d0[c];
                if (SH) {
                    if (p == 0) { top = kInf; if (!prev_adv) tl = kInf; }
                    if (p == S - 1) left = kInf;
                }
                const float v = min3f(top, left, tl) + dist(ap[c], bp[c]);
                const bool valid = p < S && (uint32_t)(si - p) < N && (uint32_t)(sj + p) < M;
                if (valid) d2[c] = v;
            }
#pragma unroll
            for (int c = 0; c < C; c++) { const float t = d0[c]; d0[c] = d1[c]; d1[c] = d2[c]; d2[c] = t; }
        }
        // a-window: every lane takes its left neighbour's value
#pragma unroll
        for (int c = C - 1; c >= 0; c--) {
            const float fill = (c > 0) ? read_lane(ap[c > 0 ? c - 1 : 0], 63) : fresh_a;
            ap[c] = wave_shr1(ap[c], fill);
        }
        // primary antidiagonal (dtw.cpp:416-485): offset o = p - SH, cell (si - p, sj + p)
        {
            const int si = (int)col + HP + SH, sj = row - HP - SH;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int p = c * 64 + lane;
                const int o = p - SH;
                const float f1 = (c > 0) ? read_lane(d1[c > 0 ? c - 1 : 0], 63) : kInf;
                float top = wave_shr1(d1[c], f1); // dp1[p-1]
                float tl, left = d1[c];
                if (adv) {
                    tl = d0[c];
                    if (!SH && p == P - 1) left = kInf;
                } else {
                    const float f0 = (c > 0) ? read_lane(d0[c > 0 ? c - 1 : 0], 63) : kInf;
                    tl = wave_shr1(d0[c], f0); // dp0[p-1]
                    if (o == 0) { top = kInf; if (!SH || !prev_adv) tl = kInf; }
                }
                const float v = min3f(top, left, tl) + dist(ap[c], bp[c]);
                const bool valid = o >= 0 && o < P && (uint32_t)(si - p) < N && (uint32_t)(sj + p) < M;
                if (valid) d2[c] = v;
            }
        }
#pragma unroll
        for (int c = 0; c < C; c++) { const float t = d0[c]; d0[c] = d1[c]; d1[c] = d2[c]; d2[c] = t; }
        prev_adv = adv;
    }
    const int pstar = P / 2 + SH; // dtw.cpp:506-512
    float res = 0.0f;
#pragma unroll
    for (int c = 0; c < C; c++)
        if ((pstar >> 6) == c) res = read_lane(d1[c], pstar & 63);
    if (lane == 0) {
        if (jb.flags & kFlagExcludeLast) res = res - dist(A[N - 1], B[M - 1]);
        out[jb.aux] = res;
    }
}

// The wave-per-job scheme without per-cell validity (see lane_dp_gen for the argument: slots outside an antidiagonal's
// extent and cells beyond the matrix on the high side are never read by a cell inside it, except through the reference's
// guarded reads; cells beyond the matrix on the LOW side exist only in a job's first columns, which take the masked copy
// of the step).  Two buffers (d1 = latest antidiagonal, d0 = the one before), control flow scalar: the job is the wave's,
// so `adv` is uniform.
// Layout: BLOCKED -- lane l holds the physical slots l*C .. l*C + C-1 in its C registers.  A neighbour at p-1 / p+1 is a
// register of the same lane except at the block's edge: ONE DPP wave shift per direction and buffer, whatever C is, and
// no v_readlane carry between registers (the strided layout p = c*64 + lane of wreg_body needs C shifts and C-1 carries
// for each, every carry a VALU -> SALU -> VALU round trip in the column's dependent chain).  It is the longest
// wide-band job of a batch that decides when the batch's DTW launch ends, and that job has a band of 65..128 slots.
// Guards: none in the step.  Invariant instead: a slot outside its antidiagonal's extent (secondaries: p < S; primaries:
// SH <= p < SH + P) holds 1e10 -- one select per antidiagonal on a loop-invariant lane mask.  Every guarded read of the
// reference (is_first, is_last, previous_increment_center_row: dtw.cpp:373-375, 392-397, 428-441, 461-473) is a read of
// such a slot, or of the slot before slot 0 (the shift's fill), and yields 1e10 there:
//   secondary p reads d1[p], d1[p+1] (the primary before it) and d0[p]: odd radius -- primaries live at 1..R, so d1[0]
//   (is_first) and d1[R+1] (is_last, p = S-1) are outside, and d0[0] is outside exactly when d0 is a primary, i.e. when
//   the row did not advance the column before; even radius -- all inside;
//   primary p after a secondary reads X[p-1], X[p], d1[p]: even radius, p = P-1 = S reads X[S], outside (is_last);
//   primary p without one reads d1[p-1], d1[p], d0[p-1]: at p = SH these are slot SH-1 of primaries (outside: slot 0 of
//   an odd radius, the fill of an even one) -- d0[0] of an odd radius again inside exactly after an advance.
template <int C, bool MASKED>
__device__ __forceinline__ void wreg_gen_step(float (&d0)[C], float (&d1)[C], float (&ap)[C], float (&bp)[C], const bool adv,
                                              const float fresh_a, const float fresh_b, const bool (&in_sec)[C],
                                              const bool (&in_prim)[C], const int lane, const int off, const int row, const int col)
{
    const int p0 = lane * C;
    float X[C];
    if (adv) {
        // b-window: every slot takes its right neighbour's value (the last physical slot the fresh one)
        {
            const float in = wave_shl1(bp[0], fresh_b);
#pragma unroll
            for (int c = 0; c + 1 < C; c++) bp[c] = bp[c + 1];
            bp[C - 1] = in;
        }
        // secondary antidiagonal (dtw.cpp:361-414): a-window of the previous column
        const float d1_up = wave_shl1(d1[0], kInf); // the next lane's first slot
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float left = (c + 1 < C) ? d1[c + 1 < C ? c + 1 : c] : d1_up;
            float v = min3f(d1[c], left, d0[c]) + dist(ap[c], bp[c]);
            bool keep = in_sec[c];
            if (MASKED) keep = keep && !(p0 + c < off - row || p0 + c > col - 1 + off);
            X[c] = keep ? v : kInf;
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) X[c] = d1[c];
    }
    // a-window: every slot takes its left neighbour's value (slot 0 the fresh one)
    {
        const float in = wave_shr1(ap[C - 1], fresh_a);
#pragma unroll
        for (int c = C - 1; c > 0; c--) ap[c] = ap[c - 1];
        ap[0] = in;
    }
    // primary antidiagonal (dtw.cpp:416-485): after a secondary (top, left, diagonal) = (X[p-1], X[p], d1[p]); otherwise
    // X is the primary before this one and they are (X[p-1], X[p], d0[p-1])
    const float X_dn = wave_shr1(X[C - 1], kInf); // the previous lane's last slot
    float d0_dn = kInf;
    if (!adv) d0_dn = wave_shr1(d0[C - 1], kInf);
    float pr[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float top = c > 0 ? X[c > 0 ? c - 1 : 0] : X_dn;
        const float tl = adv ? d1[c] : (c > 0 ? d0[c > 0 ? c - 1 : 0] : d0_dn);
        float v = min3f(top, X[c], tl) + dist(ap[c], bp[c]);
        bool keep = in_prim[c];
        if (MASKED) keep = keep && !(p0 + c < off - row);
        pr[c] = keep ? v : kInf;
    }
#pragma unroll
    for (int c = 0; c < C; c++) { d0[c] = X[c]; d1[c] = pr[c]; }
}

template <int C>
__device__ __forceinline__ void wreg_gen(const DevJob &jb, const int lane, const float *__restrict__ ev,
                                         const float *__restrict__ ref, float *__restrict__ out)
{
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
    const int SH = P > S ? 0 : 1;
    const int off = P / 2 + SH, K = P > S ? P : S;
    const int iN = (int)N, iM = (int)M;
    auto ldA = [&](int i) { return A[i < 0 ? 0 : (i >= iN ? iN - 1 : i)]; };
    auto ldB = [&](int i) { return B[i < 0 ? 0 : (i >= iM ? iM - 1 : i)]; };
    float d0[C], d1[C], ap[C], bp[C];
    bool in_sec[C], in_prim[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int p = lane * C + c;
        d0[c] = kInf; d1[c] = kInf;
        ap[c] = ldA(off - p);
        bp[c] = ldB(p - off);
        if (p == off) d1[c] = dist(A[0], B[0]); // the corner (dtw.cpp:317-347)
        in_sec[c] = p < S;
        in_prim[c] = p >= SH && p < SH + P;
    }
    // fresh-value chunks, 64 values a lane each, read by cursor: column col needs A[col + off] = A[a0 + col - 1]; the r-th
    // row advance needs B[b0 + r - 1].  A chunk is replaced right after its last value went out (the one after it is
    // already in registers; the load issued here is for the chunk after that).
    const int a0 = off + 1, b0 = 64 * C - off;
    float acur = ldA(a0 + lane), anxt = ldA(a0 + 64 + lane);
    float bcur = ldB(b0 + lane), bnxt = ldB(b0 + 64 + lane);
    uint32_t ca = 0, cb = 0;
    int row = 0;
    uint32_t rem = 0, col = 1;
    auto next_a = [&]() {
        const float v = read_lane(acur, (int)ca);
        if (++ca == 64u) { ca = 0; acur = anxt; anxt = ldA(a0 + (int)col + 64 + lane); }
        return v;
    };
    auto next_b = [&]() { // (row already counts this advance)
        const float v = read_lane(bcur, (int)cb);
        if (++cb == 64u) { cb = 0; bcur = bnxt; bnxt = ldB(b0 + row + 64 + lane); }
        return v;
    };
    // first columns: cells above row 0 or left of column 0 exist while row < off or col - 1 + off < K - 1
    for (; col < N && (row < off || (int)col < K - off + 1); col++) {
        rem += M;
        const bool adv = rem >= N;
        const float fresh_a = next_a();
        float fresh_b = 0.0f;
        if (adv) { rem -= N; row++; fresh_b = next_b(); }
        wreg_gen_step<C, true>(d0, d1, ap, bp, adv, fresh_a, fresh_b, in_sec, in_prim, lane, off, row, (int)col);
    }
    // the rest: one uniform branch per column (did the centre row advance?), nothing else
    for (; col < N; col++) {
        rem += M;
        const float fresh_a = next_a();
        if (rem >= N) {
            rem -= N; row++;
            const float fresh_b = next_b();
            wreg_gen_step<C, false>(d0, d1, ap, bp, true, fresh_a, fresh_b, in_sec, in_prim, lane, off, row, (int)col);
        } else {
            wreg_gen_step<C, false>(d0, d1, ap, bp, false, fresh_a, 0.0f, in_sec, in_prim, lane, off, row, (int)col);
        }
    }
    float res = 0.0f; // dtw.cpp:506-512: the centre of the last primary
#pragma unroll
    for (int c = 0; c < C; c++)
        if (off % C == c) res = read_lane(d1[c], off / C);
    if (lane == 0) {
        if (jb.flags & kFlagExcludeLast) res = res - dist(A[N - 1], B[M - 1]);
        out[jb.aux] = res;
    }
}

// ---------------------------------------------------------------------------------------------
// wband_gen<C>: the wave-per-job band of wreg_gen with its two operand windows read from LDS.
//
// What a column of wreg_gen costs is not its cells: the band's arithmetic is ~10 VALU instructions, the column took 350-390
// clocks (profiles/r03_wreg_probe.txt: 144 ns with one register a lane, 162 with two) -- a wave that is alone on its SIMD issues
// an instruction every ~7 clocks WHATEVER it is, and two thirds of the column's ~45 instructions kept the windows moving:
// two v_readlane -> SGPR -> v_mov -> DPP round trips for the fresh operands, the cursors' increments, refill tests and
// branches, a fill move in front of every shift, the register copies of the blocked layout.  Here the job's two operand
// stretches are staged ONCE into a wave-private piece of LDS (A reversed, so that both windows ascend with the slot), and a
// column's windows are one LDS read each -- 4 C bytes a lane, whatever C is -- asked for a column ahead; the row-advance
// test is the only branch; the two DPP shifts of the DP values write into registers whose fill lane (1e10) is never
// overwritten, so nothing has to be moved in front of them.
//   per column with a row advance: 2 LDS reads + their two address adds, 2 DPP shifts, 8 C VALU for the 2 C cells, ~5 SALU.
// Same cells, same neighbours, same masks as wreg_gen_step (the argument there): bit-identical costs.
// Staging: lds_a[k] = A[clamp(i_hi - k)], lds_b[k] = B[clamp(j_lo + k)] for the columns [col, ce) of a segment; a job longer
// than the wave's piece of LDS takes several segments.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_shl_keep(float &keep, const float v)
{
    // lane l receives lane l+1's value; lane 63 keeps what `keep` holds there (1e10, written once: no lane ever writes it)
    keep = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
    return keep;
}
__device__ __forceinline__ float dpp_shr_keep(float &keep, const float v)
{
    keep = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
    return keep;
}

template <int C> struct WinVec { float v[C]; };
template <int C> __device__ __forceinline__ WinVec<C> lds_window(const float *p) // C consecutive floats at p (any 4-byte boundary: ds_read2_b32 pairs)
{
    WinVec<C> w;
    __builtin_memcpy(w.v, p, sizeof(float) * C);
    return w;
}

// one column.  X = the secondary antidiagonal (row advance) or the primary before this one; see wreg_gen_step.
template <int C, bool MASKED>
__device__ __forceinline__ void wband_step(float (&d0)[C], float (&d1)[C], const float (&ap_prev)[C], const float (&ap)[C], const float (&bp)[C],
                                           const bool adv, const lmask (&m_sec)[C], const lmask (&m_prim)[C], float &keep_hi, float &keep_lo, float &keep_lo2,
                                           const int lane, const int off, const int row, const int col)
{
    const int p0 = lane * C;
    float X[C];
    if (adv) {
        const float d1_up = dpp_shl_keep(keep_hi, d1[0]); // the next lane's first slot
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float left = (c + 1 < C) ? d1[c + 1 < C ? c + 1 : c] : d1_up;
            const float v = min3f(d1[c], left, d0[c]) + dist(ap_prev[c], bp[c]);
            if (MASKED) X[c] = (((m_sec[c] >> lane) & 1ull) && !(p0 + c < off - row || p0 + c > col - 1 + off)) ? v : kInf;
            else X[c] = selm(m_sec[c], v, kInf);
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) X[c] = d1[c];
    }
    const float X_dn = dpp_shr_keep(keep_lo, X[C - 1]); // the previous lane's last slot
    float d0_dn = kInf;
    if (!adv) d0_dn = dpp_shr_keep(keep_lo2, d0[C - 1]);
    float pr[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float top = c > 0 ? X[c > 0 ? c - 1 : 0] : X_dn;
        const float tl = adv ? d1[c] : (c > 0 ? d0[c > 0 ? c - 1 : 0] : d0_dn);
        const float v = min3f(top, X[c], tl) + dist(ap[c], bp[c]);
        if (MASKED) pr[c] = (((m_prim[c] >> lane) & 1ull) && !(p0 + c < off - row)) ? v : kInf;
        else pr[c] = selm(m_prim[c], v, kInf);
    }
#pragma unroll
    for (int c = 0; c < C; c++) { d0[c] = X[c]; d1[c] = pr[c]; }
}

} // namespace rawdtw
#include "rawdtw_wband_asm.h" // wband_loop_asm: the main loop below, hand-scheduled (scripts/gen_wband_asm.py); opens the namespace itself
namespace rawdtw {

template <int C>
__device__ __forceinline__ void wband_gen(const DevJob &jb, const int lane, const float *__restrict__ ev, const float *__restrict__ ref,
                                          float *__restrict__ out, float *lds, const uint32_t lds_floats)
{
    // (the job is the wave's: its shape in scalar registers, whatever the caller derived its index from)
    const float *A = ev + (uint32_t)__builtin_amdgcn_readfirstlane((int)jb.read_off);
    const float *B = ref + (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(jb.ref_off >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)jb.ref_off));
    uint32_t N = (uint32_t)__builtin_amdgcn_readfirstlane((int)jb.n), M = (uint32_t)__builtin_amdgcn_readfirstlane((int)jb.m);
    if (N < M) {
        const float *tp = A; A = B; B = tp;
        uint32_t tn = N; N = M; M = tn;
    }
    const int R = __builtin_amdgcn_readfirstlane(jb.R);
    const int P = R + ((R % 2 == 0) ? 1 : 0);
    const int S = R + ((R % 2 == 1) ? 1 : 0);
    const int SH = P > S ? 0 : 1;
    const int off = P / 2 + SH, K = P > S ? P : S;
    const int iN = (int)N, iM = (int)M;
    constexpr int W = 64 * C; // slots the wave holds
    float d0[C], d1[C];
    lmask m_sec[C], m_prim[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int p = lane * C + c;
        d0[c] = kInf;
        d1[c] = p == off ? dist(A[0], B[0]) : kInf; // the corner (dtw.cpp:317-347)
        m_sec[c] = __ballot(p < S);
        m_prim[c] = __ballot(p >= SH && p < SH + P);
    }
    float keep_hi = kInf, keep_lo = kInf, keep_lo2 = kInf;
    // the wave's piece of LDS: the DP state's way into and out of the hand-scheduled loop (2 C floats a lane), then a segment's
    // columns: both stretches (+ a window each, + a few floats of slack: the loop asks for the windows of the column behind its
    // last one) in half of the rest each
    float *lds_st = lds + lane * 2 * C;
    const int half = (int)((lds_floats - 128u * C) / 2u) & ~3;
    const int seg = half - W - 12;
    float *lds_a = lds + 128 * C + 4, *lds_b = lds + 128 * C + half;
    WbandMasks wm;
    {   // register c's lanes inside [0, H): l C + c < H  <=>  l <= H / C for c < H % C, l < H / C otherwise
        auto lanes_below = [](const int n) { // (in scalar registers: the loop takes them as such)
            const unsigned long long m = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
            return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
        };
        const int Hp = SH + P;
        if constexpr (C >= 3) {
            // the hand-scheduled loop forces ONE slot an antidiagonal -- the one right behind its extent (slot S, slot SH + P): register
            // (slot % C) on every lane but (slot / C) -- and, for primaries that start at slot 1, slot 0 (rawdtw_wband_asm.h: force)
            auto all_but = [&](const int l) { return ~(lanes_below(l + 1) & ~lanes_below(l)); }; // (l >= 64: every lane)
            wm.sec_lo = 0; wm.sec_hi = all_but(S / C); wm.sec_c0 = (uint32_t)(S % C);
            wm.prim_lo = 0; wm.prim_hi = all_but(Hp / C); wm.prim_c0 = (uint32_t)(Hp % C);
            wm.prim_0 = lanes_below(64) & (SH ? ~1ull : ~0ull);
        } else {
            wm.sec_lo = lanes_below(S / C); wm.sec_hi = lanes_below(S / C + 1); wm.sec_c0 = (uint32_t)(S % C);
            wm.prim_lo = lanes_below(Hp / C); wm.prim_hi = lanes_below(Hp / C + 1); wm.prim_c0 = (uint32_t)(Hp % C);
            wm.prim_0 = lanes_below(0 < Hp % C ? Hp / C + 1 : Hp / C) & (SH ? ~1ull : ~0ull); // (the primaries start at slot SH)
        }
        wm.prim_0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wm.prim_0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wm.prim_0);
    }
    int row = 0;
    uint32_t rem = 0;
    int col = 1;
    while (col < iN) {
        const int ce = min(iN, col + seg);
        // A: the windows of columns col - 1 .. ce - 1 (the secondary of a column reads the window of the column before):
        // slot p of column c holds A[c + off - p]  =>  lds_a[k] = A[i_hi - k], window of column c from k = ce - 1 - c on
        const int i_hi = ce - 1 + off, na = ce - col + W;
        for (int k = lane; k < na; k += 64) { const int i = i_hi - k; lds_a[k] = A[i < 0 ? 0 : (i >= iN ? iN - 1 : i)]; }
        // B: slot p at centre row r holds B[r - off + p]; the rows of this segment: row .. row + (ce - col) at most
        const int row0 = row, j_lo = row - off, nb = min(ce - col, iM - 1 - row) + W;
        for (int k = lane; k < nb; k += 64) { const int j = j_lo + k; lds_b[k] = B[j < 0 ? 0 : (j >= iM ? iM - 1 : j)]; }
        // (the wave's own LDS writes are ordered with its reads; the compiler must not move the reads up)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float *wa = lds_a + (ce - 1) + lane * C; // window of column c: wa - c
        const float *wb = lds_b + lane * C - row0;     // window at centre row r: wb + r
        auto columns = [&](const int upto) { // columns [col, upto), windows straight from LDS
            for (; col < upto; col++) {
                rem += M;
                const bool adv = rem >= N;
                if (adv) { rem -= N; row++; }
                const WinVec<C> ap_prev = lds_window<C>(wa - (col - 1)), ap = lds_window<C>(wa - col), bp = lds_window<C>(wb + row);
                // first columns: cells above row 0 or left of column 0 exist while row < off or col - 1 + off < K - 1
                if (row < off || col < K - off + 1) wband_step<C, true>(d0, d1, ap_prev.v, ap.v, bp.v, adv, m_sec, m_prim, keep_hi, keep_lo, keep_lo2, lane, off, row, col);
                else wband_step<C, false>(d0, d1, ap_prev.v, ap.v, bp.v, adv, m_sec, m_prim, keep_hi, keep_lo, keep_lo2, lane, off, row, col);
            }
        };
        while (col < ce && (row < off || col < K - off + 1)) columns(col + 1); // (the masks' conditions only ever go from true to false)
        const uint32_t iters = (uint32_t)(ce - col) / 6u;
        if (iters) { // the rest, six columns a turn: the hand-scheduled loop
            uint32_t va = (uint32_t)(size_t)(__attribute__((address_space(3))) const float *)(wa - (col + 6));
            uint32_t vb = (uint32_t)(size_t)(__attribute__((address_space(3))) const float *)(wb + row);
            const uint32_t vb0 = vb, vst = (uint32_t)(size_t)(__attribute__((address_space(3))) const float *)lds_st;
#pragma unroll
            for (int c = 0; c < C; c++) { lds_st[c] = d0[c]; lds_st[C + c] = d1[c]; }
            wband_loop_asm<C>(va, vb, vst, rem, iters, M, N, wm);
#pragma unroll
            for (int c = 0; c < C; c++) { d0[c] = lds_st[c]; d1[c] = lds_st[C + c]; }
            col += 6 * (int)iters;
            row += (int)(__builtin_amdgcn_readfirstlane((int)(vb - vb0)) / 4);
        }
        columns(ce);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the next segment's staging overwrites what this one read)
        __builtin_amdgcn_wave_barrier();
    }
    float res = 0.0f; // dtw.cpp:506-512: the centre of the last primary
#pragma unroll
    for (int c = 0; c < C; c++)
        if (off % C == c) res = read_lane(d1[c], off / C);
    if (lane == 0) {
        if (jb.flags & kFlagExcludeLast) res = res - dist(A[N - 1], B[M - 1]);
        out[jb.aux] = res;
    }
}

// ---------------------------------------------------------------------------------------------
// Four jobs per wave: the same register-resident scheme as wreg_body<1>, but for bands that fit 16
// lanes (radius + 1 <= 16) each job takes one 16-lane DPP row, so a wave advances four jobs per
// instruction instead of one.  Control values (lengths, centre row, remainder) are per-row vector
// registers; neighbours move with row_shr/row_shl, the fresh operand of each row comes out of a
// 16-wide per-row chunk through ds_bpermute.  Two buffers, clipped cells written as 1e10 (lane_dp).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float row_shr1(float v, float fill)
{
    int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float row_shl1(float v, float fill)
{
    int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float lane_gather(float v, int src_lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

template <int W>
__device__ __forceinline__ void grp_wave(const DevJob *__restrict__ jobs, uint32_t count, uint32_t wave, int lane,
                                           const float *__restrict__ ev, const float *__restrict__ ref,
                                           float *__restrict__ out)
{
    static_assert(W == 16 || W == 8, "group width");
    const int p = lane & (W - 1), rowbase = lane & (64 - W);
    const uint32_t idx = wave * (64u / W) + (uint32_t)(lane / W);
    // neighbour shifts inside a group: a DPP row shift; 8-lane groups patch the lane at the group's edge
    auto shl1 = [&](float v, float fill) { const float t = row_shl1(v, fill); return (W == 8 && p == 7) ? fill : t; };
    auto shr1 = [&](float v, float fill) { const float t = row_shr1(v, fill); return (W == 8 && p == 0) ? fill : t; };
    const bool have = idx < count;
    const DevJob jb = jobs[have ? idx : count - 1];
    const float *A = ev + jb.read_off;
    const float *B = ref + jb.ref_off;
    uint32_t N = jb.n, M = jb.m;
    if (N < M) {
        const float *tp = A; A = B; B = tp;
        const uint32_t tn = N; N = M; M = tn;
    }
    if (!have) N = 1; // an empty row never enters the column loop
    const int R = jb.R;
    const int P = R + ((R % 2 == 0) ? 1 : 0);
    const int S = R + ((R % 2 == 1) ? 1 : 0);
    const int SH = P > S ? 0 : 1;
    const int HP = P / 2;
    const int iN = (int)N, iM = (int)M;
    auto ldA = [&](int i) { return A[i < 0 ? 0 : (i >= iN ? iN - 1 : i)]; };
    auto ldB = [&](int i) { return B[i < 0 ? 0 : (i >= iM ? iM - 1 : i)]; };

    float p1 = (p == HP + SH) ? dist(A[0], B[0]) : kInf, p2 = kInf;
    float ap = ldA(HP + SH - p), bp = ldB(p - HP - SH);
    const bool in_sec = p < S, in_prim = (p - SH) >= 0 && (p - SH) < P;
    const int a0 = HP + SH + 1, b0 = W - HP - SH;
    float acur = ldA(a0 + p), anxt = ldA(a0 + W + p);
    float bcur = ldB(b0 + p), bnxt = ldB(b0 + W + p);

    int row = 0;
    uint32_t rem = 0;
    bool prev_adv = false;
    // the wave runs as long as its longest row
    uint32_t Nmax = 1;
#pragma unroll
    for (int g = 0; g < 64; g += W) Nmax = max(Nmax, (uint32_t)__builtin_amdgcn_readlane((int)N, g));
    for (uint32_t col = 1; col < Nmax; col++) {
        const bool live = col < N;
        const uint32_t ca = (col - 1) & (uint32_t)(W - 1);
        if (ca == 0 && col > 1) { acur = anxt; anxt = ldA(a0 + (int)(col - 1) + W + p); }
        const float fresh_a = lane_gather(acur, rowbase + (int)ca);
        bool adv = false;
        if (live) {
            rem += M;
            adv = rem >= N;
            if (adv) { rem -= N; row++; }
        }
        const int cb = (row - 1) & (W - 1);
        if (adv && cb == 0 && row > 1) { bcur = bnxt; bnxt = ldB(b0 + (row - 1) + W + p); }
        const float fresh_b = lane_gather(bcur, rowbase + cb);
        // shifted views of the two buffers (taken before anything is overwritten)
        const float p1_up = shl1(p1, kInf);   // dp1[p+1]
        if (adv) {
            // b-window: every lane takes its right neighbour's value
            bp = shl1(bp, fresh_b);
            // secondary antidiagonal (dtw.cpp:361-414), in place over p2: cell of lane p is (si - p, sj + p)
            const int si = (int)col - 1 + HP + SH, sj = row - HP - SH;
            float left = p1_up, top = p1, tl = p2;
            if (SH) {
                if (p == 0) { top = kInf; if (!prev_adv) tl = kInf; }
                if (p == S - 1) left = kInf;
            }
            const float v = min3f(top, left, tl) + dist(ap, bp);
            const bool valid = in_sec && (uint32_t)(si - p) < N && (uint32_t)(sj + p) < M;
            p2 = valid ? v : kInf;
        }
        if (live) {
            // a-window: every lane takes its left neighbour's value
            ap = shr1(ap, fresh_a);
            // primary antidiagonal (dtw.cpp:416-485): offset o = p - SH, cell (si - p, sj + p)
            const int si = (int)col + HP + SH, sj = row - HP - SH;
            const bool valid = in_prim && (uint32_t)(si - p) < N && (uint32_t)(sj + p) < M;
            const float p2_dn = shr1(p2, kInf); // after a secondary: dp1[p-1]; otherwise dp0[p-1]
            if (adv) {
                float left = p2;
                if (!SH && p == P - 1) left = kInf;
                const float v = min3f(p2_dn, left, p1) + dist(ap, bp);
                p1 = valid ? v : kInf;
            } else {
                float top = shr1(p1, kInf), tl = p2_dn;
                if (p - SH == 0) { top = kInf; if (!SH || !prev_adv) tl = kInf; }
                const float v = min3f(top, p1, tl) + dist(ap, bp);
                p2 = p1;
                p1 = valid ? v : kInf;
            }
            prev_adv = adv;
        }
    }
    if (have && p == P / 2 + SH) { // dtw.cpp:506-512
        float res = p1;
        if (jb.flags & kFlagExcludeLast) res = res - dist(A[N - 1], B[M - 1]);
        out[jb.aux] = res;
    }
}

// One launch for every job whose band fits 256 lanes-slots (radius + 1 <= 256): the registers-per-lane
// variant is picked per job (wave-uniform).  Jobs are sorted longest first, so the few long jobs
// that bound the launch's duration start first and the many short ones fill in around them --
// as separate launches they were separate long poles on separate streams.
constexpr uint32_t kWbandLdsFloats = 2048; // a wave's piece of LDS (8 KB): the DP state's 128 C floats + segments of 940 columns for one-register bands, 500 for four
__device__ __forceinline__ void wreg_small_job(const DevJob &jb, int lane, const float *__restrict__ ev,
                                               const float *__restrict__ ref, float *__restrict__ out, float *lds, uint32_t lds_floats = kWbandLdsFloats)
{
    const int K = jb.R + 1;
    if (K <= 64) wband_gen<1>(jb, lane, ev, ref, out, lds, lds_floats);
    else if (K <= 128) wband_gen<2>(jb, lane, ev, ref, out, lds, lds_floats);
    else if (K <= 192) wband_gen<3>(jb, lane, ev, ref, out, lds, lds_floats); // (odd: a lane's window is C consecutive floats of LDS, and with four the lanes of a
    else wband_gen<5>(jb, lane, ev, ref, out, lds, lds_floats);               //  read sit on eight of the 32 banks -- see k_band_wband8)
}

} // namespace rawdtw
