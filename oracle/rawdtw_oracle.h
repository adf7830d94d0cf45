/*
 * rawdtw_oracle.h -- CPU oracle for the RawAlign DTW hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (rawalign_amd/, the
 * C-ABI library) may include, link or call this.  Only tests/, the smoke check
 * in __graft_entry__.py and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every function restates, in plain C, what one reference function computes;
 * the reference file:line each follows is cited at its definition in
 * rawdtw_oracle.c.  Parity pin: oracle/_ref (the reference's own dtw.cpp
 * compiled where it lies) and tests/golden/ (vectors captured from it).
 */
#ifndef RAWDTW_ORACLE_H
#define RAWDTW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: DTW_global, src/dtw.cpp:37-66 */
float orc_dtw_global(const float *a, uint32_t n, const float *b, uint32_t m, int exclude_last);

/* reference: DTW_global_slantedbanded_antidiagonalwise, src/dtw.cpp:273-520
 * (three rotating antidiagonal buffers, same physical indexing and guards). */
float orc_dtw_banded(const float *a, uint32_t n, const float *b, uint32_t m, int band_radius,
                     int exclude_last);

/* Independent second formulation of the same function: builds the band's cell
 * set explicitly and runs the plain recurrence over it (absent neighbour =
 * 1e10).  O(n*m) memory -- small cases only.  Returns the cost, stores the
 * number of cells evaluated in *cells (may be NULL) and, when mask != NULL,
 * writes mask[i*M + j] = 1 for each evaluated cell (N >= M after the swap,
 * i over the longer sequence). */
float orc_dtw_banded_cellset(const float *a, uint32_t n, const float *b, uint32_t m,
                             int band_radius, int exclude_last, uint64_t *cells, uint8_t *mask);

/* Number of DP cells the banded function evaluates (no operands needed). */
uint64_t orc_banded_cells(uint32_t n, uint32_t m, int band_radius);

/* reference: DTW_global_tb, src/dtw.cpp:595-667.  path_* must hold n+m-1
 * entries; *path_len receives the number written (after the optional pop).
 * path_i indexes a, path_j indexes b, path_d = |a[i]-b[j]|. */
float orc_dtw_global_tb(const float *a, uint32_t n, const float *b, uint32_t m, int exclude_last,
                        uint32_t *path_i, uint32_t *path_j, float *path_d, uint32_t *path_len);

/* Same traceback, but also emits the 2-bit direction matrix the GPU keeps
 * (0 = diagonal, 1 = i-1 "left", 2 = j-1 "top"), row-major dirs[i*m + j],
 * one byte per cell here.  Used to check the packed buffer. */
void orc_dtw_directions(const float *a, uint32_t n, const float *b, uint32_t m, uint8_t *dirs);

/* ---- align_chain (src/rmap.cpp:181-313) and the DTW block of gen_chains
 * (src/rmap.cpp:509-530), restated.  Anchors are stored end-first exactly as
 * the reference keeps them (anchors[n-1] is the chain start). ---- */
typedef struct {
    uint32_t target_position;
    uint32_t query_position;
} orc_anchor_t;

enum { ORC_BORDER_GLOBAL = 0, ORC_BORDER_SPARSE = 1 };
enum { ORC_FILL_FULL = 0, ORC_FILL_BANDED = 1 };

typedef struct {
    int border_constraint;  /* roptions.h:21-23 */
    int fill_method;        /* roptions.h:25-26 */
    float band_radius_frac; /* roptions.c:51 */
    float match_bonus;      /* roptions.c:52 */
    float min_score;        /* roptions.c:53 */
    int fused_score;        /* 1: final score as fmaf(n,bonus,-cost) (FMA-contracting reference build) */
} orc_opt_t;

typedef struct {
    uint64_t dtw_calls;
    uint64_t cells;
} orc_stats_t;

/* Score-only align_chain (cigar=false).  Returns the alignment score
 * (-1e10 when cut by min_score).  ref_events is the strand's signal array. */
float orc_align_chain(const orc_anchor_t *anchors, uint32_t n_anchors, const float *ref_events,
                      const float *read_events, const orc_opt_t *opt, float min_score,
                      orc_stats_t *stats);

/* cigar=true variant: path arrays must hold sum over parts of (n+m-1). */
float orc_align_chain_cigar(const orc_anchor_t *anchors, uint32_t n_anchors,
                            const float *ref_events, const float *read_events,
                            const orc_opt_t *opt, uint64_t *path_i, uint64_t *path_j,
                            float *path_d, uint64_t *path_len, float *dtw_cost_out);

#ifdef __cplusplus
}
#endif
#endif
