/*
 * rawdtw_oracle_batch.c -- oracle-side batch drivers (TEST INFRASTRUCTURE).
 *
 *  orc_batch_costs     : the flat score-only job list of one GPU batch, run on
 *                        nthreads host threads (bench.py cpu_baseline "port").
 *  orc_evaluate_chains : the DTW block of gen_chains (src/rmap.cpp:509-530)
 *                        for one read, calling orc_align_chain sequentially with
 *                        the running best score, exactly as the reference does.
 *
 * The chain order handed in must already be the reference's evaluation order
 * (std::sort by chaining_score descending, rmap.cpp:512); sorting is done by
 * the caller so that the C oracle does not have to imitate libstdc++'s
 * unstable sort.
 */
#define _GNU_SOURCE
#include "rawdtw_oracle.h"

#include <pthread.h>
#include <stdlib.h>

typedef struct {
    uint64_t ref_off;
    uint32_t read_off;
    uint32_t n;
    uint32_t m;
    int32_t band_radius; /* < 0: full DTW_global */
    uint32_t exclude_last;
    uint32_t pad;
} orc_job_t;

typedef struct {
    const orc_job_t *jobs;
    uint64_t n_jobs;
    const float *events, *ref;
    float *out;
    uint64_t *next;
    pthread_mutex_t *mu;
    uint64_t total; /* n_jobs * passes over the list */
} batch_ctx;

static void *batch_worker(void *p)
{
    batch_ctx *c = (batch_ctx *)p;
    const uint64_t grain = 64;
    for (;;) {
        pthread_mutex_lock(c->mu);
        uint64_t s = *c->next;
        *c->next = s + grain;
        pthread_mutex_unlock(c->mu);
        if (s >= c->total) break;
        uint64_t e = s + grain < c->total ? s + grain : c->total;
        for (uint64_t q = s; q < e; q++) {
            const uint64_t k = q % c->n_jobs;
            const orc_job_t *j = &c->jobs[k];
            const float *a = c->events + j->read_off;
            const float *b = c->ref + j->ref_off;
            c->out[k] = j->band_radius < 0
                            ? orc_dtw_global(a, j->n, b, j->m, (int)j->exclude_last)
                            : orc_dtw_banded(a, j->n, b, j->m, j->band_radius, (int)j->exclude_last);
        }
    }
    return NULL;
}

/* `reps` passes over the job list inside one pool of threads (bench.py: one pass is too short to time). */
void orc_batch_costs_reps(const void *jobs, uint64_t n_jobs, const float *events, const float *ref,
                          float *out, int nthreads, int reps)
{
    uint64_t next = 0;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    batch_ctx c = {(const orc_job_t *)jobs, n_jobs, events, ref, out, &next, &mu, n_jobs * (uint64_t)(reps > 1 ? reps : 1)};
    if (nthreads <= 1) {
        batch_worker(&c);
        return;
    }
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, batch_worker, &c);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
}

void orc_batch_costs(const void *jobs, uint64_t n_jobs, const float *events, const float *ref,
                     float *out, int nthreads)
{
    orc_batch_costs_reps(jobs, n_jobs, events, ref, out, nthreads, 1);
}

/* One read's candidate chains, already in evaluation order.
 * anchors: concatenated, chain c owns anchors[anchor_off[c] .. anchor_off[c+1]).
 * ref_base[c]: pointer to the strand/sequence signal array the chain maps to.
 * Outputs: score[c] (alignment score, -1e10 when cut) and keep[c] (1 when the
 * chain survives the dtw_min_score filter, rmap.cpp:518). Returns #kept. */
uint32_t orc_evaluate_chains(uint32_t n_chains, const uint32_t *anchor_off,
                             const orc_anchor_t *anchors, const float *const *ref_base,
                             const float *read_events, const orc_opt_t *opt, float *score,
                             uint8_t *keep, orc_stats_t *stats)
{
    float best = 0.0f; /* rmap.cpp:515 */
    uint32_t kept = 0;
    for (uint32_t c = 0; c < n_chains; c++) {
        uint32_t na = anchor_off[c + 1] - anchor_off[c];
        float s = orc_align_chain(anchors + anchor_off[c], na, ref_base[c], read_events, opt, best,
                                  stats);
        score[c] = s;
        keep[c] = 0;
        if (s >= opt->min_score) { /* rmap.cpp:518 */
            if (s > best) best = s; /* rmap.cpp:519-521 */
            keep[c] = 1;
            kept++;
        }
    }
    return kept;
}
