/*
 * rawdtw_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see rawdtw_oracle.h).
 *
 * A plain-C restatement of the three DTW functions RawAlign's mapper calls
 * (src/dtw.cpp) and of align_chain (src/rmap.cpp:181-313).  Written from the
 * behaviour of those functions, not from their text: the band function is
 * table-driven (one rule per antidiagonal kind) instead of four hand-unrolled
 * loops, and a second, geometric formulation (orc_dtw_banded_cellset) checks
 * it from a different angle.
 *
 * Arithmetic contract: fp32 throughout, local distance
 * |x-y| with one rounding, cell = min3 + dist with one rounding, sentinel =
 * the float nearest 1e10, no multiply anywhere in the DP => no FMA question.
 * Build with -ffp-contract=off so the only fused operation is the explicit
 * fmaf in the final score when opt->fused_score is set.
 */
#include "rawdtw_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_INF 1e10f /* dtw.cpp:38, 310-313 : "1e10" stored into a float */

static inline float orc_dist(float x, float y) { return fabsf(x - y); } /* dtw.cpp:12 */

/* std::min(std::min(top,left),topleft) with std::min(a,b) = (b<a)?b:a */
static inline float orc_min3(float top, float left, float topleft)
{
    float inner = (left < top) ? left : top;
    return (topleft < inner) ? topleft : inner;
}

/* ------------------------------------------------------------------ */
/* DTW_global  (dtw.cpp:37-66): one rolling row over a, swept over b.  */
/* ------------------------------------------------------------------ */
float orc_dtw_global(const float *a, uint32_t n, const float *b, uint32_t m, int exclude_last)
{
    float *row = (float *)malloc((size_t)n * sizeof(float));
    uint32_t i, j;
    row[0] = orc_dist(a[0], b[0]);
    for (j = 1; j < n; j++) row[j] = row[j - 1] + orc_dist(a[j], b[0]); /* dtw.cpp:41-43 */
    for (i = 1; i < m; i++) {                                           /* dtw.cpp:45-59 */
        float diag = row[0];
        row[0] = row[0] + orc_dist(a[0], b[i]);
        for (j = 1; j < n; j++) {
            float up = row[j];
            float v = orc_min3(row[j - 1], up, diag) + orc_dist(a[j], b[i]);
            row[j] = v;
            diag = up;
        }
    }
    float res = row[n - 1];
    free(row);
    if (exclude_last) res = res - orc_dist(a[n - 1], b[m - 1]); /* dtw.cpp:60-62 */
    return res;
}

/* ------------------------------------------------------------------ */
/* Band geometry shared by the two banded formulations.                */
/* ------------------------------------------------------------------ */
typedef struct {
    const float *A, *B; /* A = longer sequence after the swap (dtw.cpp:284-292) */
    uint32_t N, M;
    int R, P, S, K, shift;
} band_t;

static void band_setup(band_t *g, const float *a, uint32_t n, const float *b, uint32_t m,
                       int band_radius)
{
    if (n < m) {
        g->A = b; g->N = m; g->B = a; g->M = n;
    } else {
        g->A = a; g->N = n; g->B = b; g->M = m;
    }
    /* dtw.cpp:298 -- evaluated in unsigned 32-bit arithmetic (uint32 * int) */
    uint32_t extra = ((g->N - g->M) * (uint32_t)band_radius + g->N - 1u) / g->N;
    g->R = band_radius + (int)extra;                    /* dtw.cpp:300 */
    g->P = g->R + ((g->R % 2 == 0) ? 1 : 0);            /* dtw.cpp:301 */
    g->S = g->R + ((g->R % 2 == 1) ? 1 : 0);            /* dtw.cpp:302 */
    g->K = g->P > g->S ? g->P : g->S;                   /* dtw.cpp:305 */
    g->shift = (g->P > g->S) ? 0 : 1;                   /* dtw.cpp:459,479: primaries at index+1 */
}

/* centre row advances iff (row+1)*N <= M*column  (dtw.cpp:352-359) */
static inline int band_advances(const band_t *g, int centre_row, int column)
{
    return ((int64_t)(centre_row + 1) * (int64_t)g->N) <= ((int64_t)g->M * (int64_t)column);
}

static inline void clip_range(const band_t *g, int start_i, int start_j, int len, int *lo, int *hi)
{
    int s = 0, e = len; /* dtw.cpp:365-366, 419-420 */
    if (start_i - (int)g->N + 1 > s) s = start_i - (int)g->N + 1;
    if (-start_j > s) s = -start_j;
    if (start_i + 1 < e) e = start_i + 1;
    if ((int)g->M - start_j < e) e = (int)g->M - start_j;
    *lo = s; *hi = e;
}

/* One rule per antidiagonal kind: physical-index deltas of the three
 * neighbours relative to the cell's offset o, which of them are guarded at the
 * first/last offset, and where the result is stored. */
typedef struct {
    int d_top, d_tl, d_left, d_store;
    int guard_top_first, guard_tl_first, guard_left_last;
} rule_t;

enum { KIND_SEC = 0, KIND_PRIM_DOWN = 1, KIND_PRIM_FLAT = 2 };

/* even R (primary longer): dtw.cpp:368-386, 426-452 ; odd R: dtw.cpp:388-408, 455-485 */
static const rule_t RULES[2][3] = {
    /* shift 0 */
    {
        {0, 0, +1, 0, 0, 0, 0},  /* secondary             */
        {-1, 0, 0, 0, 1, 0, 1},  /* primary, row advanced */
        {-1, -1, 0, 0, 1, 1, 0}, /* primary, same row     */
    },
    /* shift 1; guard_tl_first == 2 means "guarded unless the previous column advanced" */
    {
        {0, 0, +1, 0, 1, 2, 1},
        {0, +1, +1, +1, 0, 0, 0},
        {0, 0, +1, +1, 1, 2, 0},
    },
};

float orc_dtw_banded(const float *a, uint32_t n, const float *b, uint32_t m, int band_radius,
                     int exclude_last)
{
    band_t g;
    band_setup(&g, a, n, b, m, band_radius);
    const int K = g.K;
    float *store = (float *)malloc((size_t)3 * K * sizeof(float));
    float *d0 = store, *d1 = store + K, *d2 = store + 2 * K, *t;
    for (int x = 0; x < 3 * K; x++) store[x] = ORC_INF; /* dtw.cpp:309-314 */

    /* column 0: only the corner cell exists (dtw.cpp:317-347) */
    d2[g.P / 2 + g.shift] = orc_dist(g.A[0], g.B[0]);
    t = d0; d0 = d1; d1 = d2; d2 = t;

    int row = 0, prev_adv = 0;
    for (int col = 1; (uint32_t)col < g.N; col++) {
        int adv = band_advances(&g, row, col);
        if (adv) row++;
        for (int pass = adv ? 0 : 1; pass < 2; pass++) {
            int kind = pass == 0 ? KIND_SEC : (adv ? KIND_PRIM_DOWN : KIND_PRIM_FLAT);
            int len = pass == 0 ? g.S : g.P;
            int start_i = pass == 0 ? col + g.S / 2 - 1 : col + g.P / 2; /* dtw.cpp:362,416 */
            int start_j = pass == 0 ? row - g.S / 2 : row - g.P / 2;     /* dtw.cpp:363,417 */
            const rule_t *r = &RULES[g.shift][kind];
            int lo, hi;
            clip_range(&g, start_i, start_j, len, &lo, &hi);
            for (int o = lo; o < hi; o++) {
                int first = (o == 0), last = (o == len - 1);
                int tl_guard = r->guard_tl_first == 2 ? (first && !prev_adv)
                                                      : (r->guard_tl_first && first);
                float top = (r->guard_top_first && first) ? ORC_INF : d1[o + r->d_top];
                float tl = tl_guard ? ORC_INF : d0[o + r->d_tl];
                float left = (r->guard_left_last && last) ? ORC_INF : d1[o + r->d_left];
                d2[o + r->d_store] = orc_min3(top, left, tl) + orc_dist(g.A[start_i - o], g.B[start_j + o]);
            }
            t = d0; d0 = d1; d1 = d2; d2 = t; /* dtw.cpp:410-413, 487-490 */
        }
        prev_adv = adv;
    }
    float res = d1[g.P / 2 + g.shift]; /* dtw.cpp:506-512 */
    free(store);
    if (exclude_last) res = res - orc_dist(g.A[g.N - 1], g.B[g.M - 1]); /* dtw.cpp:514-516 */
    return res;
}

/* Walk the band's antidiagonals, calling visit(i,j) for every evaluated cell
 * in evaluation order. */
typedef void (*cell_fn)(void *ctx, uint32_t i, uint32_t j);

static uint64_t band_walk(const band_t *g, cell_fn visit, void *ctx)
{
    uint64_t cells = 1;
    if (visit) visit(ctx, 0, 0);
    int row = 0;
    for (int col = 1; (uint32_t)col < g->N; col++) {
        int adv = band_advances(g, row, col);
        if (adv) row++;
        for (int pass = adv ? 0 : 1; pass < 2; pass++) {
            int len = pass == 0 ? g->S : g->P;
            int start_i = pass == 0 ? col + g->S / 2 - 1 : col + g->P / 2;
            int start_j = pass == 0 ? row - g->S / 2 : row - g->P / 2;
            int lo, hi;
            clip_range(g, start_i, start_j, len, &lo, &hi);
            if (hi > lo) cells += (uint64_t)(hi - lo);
            if (visit)
                for (int o = lo; o < hi; o++) visit(ctx, (uint32_t)(start_i - o), (uint32_t)(start_j + o));
        }
    }
    return cells;
}

uint64_t orc_banded_cells(uint32_t n, uint32_t m, int band_radius)
{
    band_t g;
    band_setup(&g, NULL, n, NULL, m, band_radius);
    return band_walk(&g, NULL, NULL);
}

typedef struct {
    const band_t *g;
    float *val;
    uint8_t *present;
} cellset_ctx;

static void cellset_visit(void *p, uint32_t i, uint32_t j)
{
    cellset_ctx *c = (cellset_ctx *)p;
    const band_t *g = c->g;
    size_t M = g->M, at = (size_t)i * M + j;
    float v;
    if (i == 0 && j == 0) {
        v = orc_dist(g->A[0], g->B[0]);
    } else {
        /* "left" = (i-1,j), "top" = (i,j-1), as the reference names them */
        float left = (i > 0 && c->present[at - M]) ? c->val[at - M] : ORC_INF;
        float top = (j > 0 && c->present[at - 1]) ? c->val[at - 1] : ORC_INF;
        float tl = (i > 0 && j > 0 && c->present[at - M - 1]) ? c->val[at - M - 1] : ORC_INF;
        v = orc_min3(top, left, tl) + orc_dist(g->A[i], g->B[j]);
    }
    c->val[at] = v;
    c->present[at] = 1;
}

float orc_dtw_banded_cellset(const float *a, uint32_t n, const float *b, uint32_t m,
                             int band_radius, int exclude_last, uint64_t *cells, uint8_t *mask)
{
    band_t g;
    band_setup(&g, a, n, b, m, band_radius);
    size_t sz = (size_t)g.N * g.M;
    cellset_ctx c;
    c.g = &g;
    c.val = (float *)malloc(sz * sizeof(float));
    c.present = (uint8_t *)calloc(sz, 1);
    uint64_t cnt = band_walk(&g, cellset_visit, &c);
    float res = c.present[sz - 1] ? c.val[sz - 1] : ORC_INF;
    if (cells) *cells = cnt;
    if (mask) memcpy(mask, c.present, sz);
    free(c.val);
    free(c.present);
    if (exclude_last) res = res - orc_dist(g.A[g.N - 1], g.B[g.M - 1]);
    return res;
}

/* ------------------------------------------------------------------ */
/* DTW_global_tb (dtw.cpp:595-667)                                     */
/* ------------------------------------------------------------------ */
static float *full_matrix(const float *a, uint32_t n, const float *b, uint32_t m)
{
    float *D = (float *)malloc((size_t)n * m * sizeof(float));
    size_t M = m;
    D[0] = orc_dist(a[0], b[0]);
    for (uint32_t i = 1; i < n; i++) D[i * M] = D[(i - 1) * M] + orc_dist(a[i], b[0]); /* dtw.cpp:602-604 */
    for (uint32_t j = 1; j < m; j++) D[j] = D[j - 1] + orc_dist(a[0], b[j]);           /* dtw.cpp:605-607 */
    for (uint32_t i = 1; i < n; i++)
        for (uint32_t j = 1; j < m; j++) /* dtw.cpp:609-614 */
            D[i * M + j] = orc_min3(D[(i - 1) * M + j], D[i * M + j - 1], D[(i - 1) * M + j - 1]) +
                           orc_dist(a[i], b[j]);
    return D;
}

/* the three-way decision of dtw.cpp:633-646: 1 = i--, 2 = j--, 0 = both */
static inline int tb_direction(float left, float top, float tl)
{
    float m_top_tl = (tl < top) ? tl : top;
    if (left < m_top_tl) return 1;
    float m_left_tl = (tl < left) ? tl : left;
    if (top < m_left_tl) return 2;
    return 0;
}

float orc_dtw_global_tb(const float *a, uint32_t n, const float *b, uint32_t m, int exclude_last,
                        uint32_t *path_i, uint32_t *path_j, float *path_d, uint32_t *path_len)
{
    float *D = full_matrix(a, n, b, m);
    size_t M = m;
    uint32_t cap = n + m - 1, len = 0;
    uint32_t *ri = (uint32_t *)malloc(cap * sizeof(uint32_t));
    uint32_t *rj = (uint32_t *)malloc(cap * sizeof(uint32_t));
    uint32_t i = n - 1, j = m - 1;
    ri[len] = i; rj[len] = j; len++;
    while (i > 0 || j > 0) {
        if (i == 0) j--;
        else if (j == 0) i--;
        else {
            int d = tb_direction(D[(i - 1) * M + j], D[i * M + j - 1], D[(i - 1) * M + j - 1]);
            if (d == 1) i--;
            else if (d == 2) j--;
            else { i--; j--; }
        }
        ri[len] = i; rj[len] = j; len++;
    }
    float cost = D[(size_t)(n - 1) * M + (m - 1)];
    uint32_t out = len;
    if (exclude_last) { /* dtw.cpp:659-663 */
        out = len - 1;
        cost = cost - orc_dist(a[n - 1], b[m - 1]);
    }
    for (uint32_t k = 0; k < out; k++) {
        uint32_t src = len - 1 - k;
        path_i[k] = ri[src];
        path_j[k] = rj[src];
        path_d[k] = orc_dist(a[ri[src]], b[rj[src]]);
    }
    *path_len = out;
    free(ri); free(rj); free(D);
    return cost;
}

void orc_dtw_directions(const float *a, uint32_t n, const float *b, uint32_t m, uint8_t *dirs)
{
    float *D = full_matrix(a, n, b, m);
    size_t M = m;
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t j = 0; j < m; j++) {
            int d;
            if (i == 0 && j == 0) d = 0;
            else if (i == 0) d = 2;
            else if (j == 0) d = 1;
            else d = tb_direction(D[(i - 1) * M + j], D[i * M + j - 1], D[(i - 1) * M + j - 1]);
            dirs[i * M + j] = (uint8_t)d;
        }
    free(D);
}

/* ------------------------------------------------------------------ */
/* align_chain (rmap.cpp:181-313)                                      */
/* ------------------------------------------------------------------ */
static inline int band_radius_for(uint32_t read_region_size, float frac)
{
    int r = (int)((float)read_region_size * frac); /* rmap.cpp:214,276 : uint32*float in fp32 */
    return r > 1 ? r : 1;
}

static float final_score(uint32_t num_aligned, const orc_opt_t *opt, float cost)
{
    if (opt->fused_score) return fmaf((float)num_aligned, opt->match_bonus, -cost); /* SURVEY 8a-4 */
    float prod = (float)num_aligned * opt->match_bonus;                              /* rmap.cpp:306 */
    return prod - cost;
}

static float one_cost(const float *rd, uint32_t rn, const float *rf, uint32_t fm,
                      const orc_opt_t *opt, int excl, orc_stats_t *st)
{
    if (opt->fill_method == ORC_FILL_FULL) {
        if (st) { st->dtw_calls++; st->cells += (uint64_t)rn * fm; }
        return orc_dtw_global(rd, rn, rf, fm, excl);
    }
    int R0 = band_radius_for(rn, opt->band_radius_frac);
    if (st) { st->dtw_calls++; st->cells += orc_banded_cells(rn, fm, R0); }
    return orc_dtw_banded(rd, rn, rf, fm, R0, excl);
}

float orc_align_chain(const orc_anchor_t *anchors, uint32_t n_anchors, const float *ref_events,
                      const float *read_events, const orc_opt_t *opt, float min_score,
                      orc_stats_t *stats)
{
    float cost = 0.0f;
    uint32_t num_aligned = 0;
    const orc_anchor_t *first = &anchors[n_anchors - 1]; /* chain start (rmap.cpp:195) */
    const orc_anchor_t *lastA = &anchors[0];             /* chain end   (rmap.cpp:196) */
    if (opt->border_constraint == ORC_BORDER_GLOBAL) {
        uint32_t fm = lastA->target_position - first->target_position + 1;
        uint32_t rn = lastA->query_position - first->query_position + 1;
        float attainable = (float)rn * opt->match_bonus; /* rmap.cpp:205 */
        if (attainable < min_score) return -1e10f;       /* rmap.cpp:206-209 */
        cost = one_cost(read_events + first->query_position, rn,
                        ref_events + first->target_position, fm, opt, 0, stats);
        num_aligned = rn;
    } else {
        uint32_t parts = n_anchors - 1;
        uint32_t span = lastA->query_position - first->query_position + 1;
        float attainable = (float)span * opt->match_bonus; /* rmap.cpp:246 */
        for (uint32_t p = 0; p < parts; p++) {
            const orc_anchor_t *s = &anchors[parts - p];
            const orc_anchor_t *e = &anchors[parts - p - 1];
            uint32_t fm = e->target_position - s->target_position + 1;
            uint32_t rn = e->query_position - s->query_position + 1;
            if (attainable < min_score) return -1e10f; /* rmap.cpp:265-268 */
            float sub = one_cost(read_events + s->query_position, rn,
                                 ref_events + s->target_position, fm, opt, p != parts - 1, stats);
            cost += sub;       /* rmap.cpp:279 */
            attainable -= sub; /* rmap.cpp:280 */
            num_aligned += rn; /* rmap.cpp:292 */
        }
    }
    return final_score(num_aligned, opt, cost);
}

float orc_align_chain_cigar(const orc_anchor_t *anchors, uint32_t n_anchors,
                            const float *ref_events, const float *read_events,
                            const orc_opt_t *opt, uint64_t *path_i, uint64_t *path_j,
                            float *path_d, uint64_t *path_len, float *dtw_cost_out)
{
    float cost = 0.0f;
    uint32_t num_aligned = 0;
    uint64_t total = 0;
    const orc_anchor_t *first = &anchors[n_anchors - 1];
    const orc_anchor_t *lastA = &anchors[0];
    if (opt->border_constraint == ORC_BORDER_GLOBAL) {
        if (opt->fill_method != ORC_FILL_FULL) { /* rmap.cpp:223-225: assert(false) */
            *path_len = (uint64_t)-1;
            return NAN;
        }
        uint32_t fm = lastA->target_position - first->target_position + 1;
        uint32_t rn = lastA->query_position - first->query_position + 1;
        uint32_t cap = rn + fm - 1, len = 0;
        uint32_t *pi = (uint32_t *)malloc(cap * sizeof(uint32_t));
        uint32_t *pj = (uint32_t *)malloc(cap * sizeof(uint32_t));
        cost = orc_dtw_global_tb(read_events + first->query_position, rn,
                                 ref_events + first->target_position, fm, 0, pi, pj, path_d, &len);
        for (uint32_t k = 0; k < len; k++) { path_i[k] = pi[k]; path_j[k] = pj[k]; }
        /* rmap.cpp:230-233: the loop adds the offsets to the LAST element, once per element */
        if (len) {
            path_i[len - 1] += (uint64_t)len * first->query_position;
            path_j[len - 1] += (uint64_t)len * first->target_position;
        }
        total = len;
        free(pi); free(pj);
        num_aligned = rn;
    } else {
        uint32_t parts = n_anchors - 1;
        for (uint32_t p = 0; p < parts; p++) {
            const orc_anchor_t *s = &anchors[parts - p];
            const orc_anchor_t *e = &anchors[parts - p - 1];
            uint32_t fm = e->target_position - s->target_position + 1;
            uint32_t rn = e->query_position - s->query_position + 1;
            uint32_t cap = rn + fm - 1, len = 0;
            uint32_t *pi = (uint32_t *)malloc(cap * sizeof(uint32_t));
            uint32_t *pj = (uint32_t *)malloc(cap * sizeof(uint32_t));
            /* rmap.cpp:283-284: traceback parts never exclude their last element */
            float sub = orc_dtw_global_tb(read_events + s->query_position, rn,
                                          ref_events + s->target_position, fm, 0, pi, pj,
                                          path_d + total, &len);
            for (uint32_t k = 0; k < len; k++) {
                path_i[total + k] = (uint64_t)pi[k] + s->query_position;  /* rmap.cpp:287 */
                path_j[total + k] = (uint64_t)pj[k] + s->target_position; /* rmap.cpp:288 */
            }
            total += len;
            cost += sub; /* rmap.cpp:290 */
            num_aligned += rn;
            free(pi); free(pj);
        }
    }
    *path_len = total;
    if (dtw_cost_out) *dtw_cost_out = cost;
    return final_score(num_aligned, opt, cost);
}
