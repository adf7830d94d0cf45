// ref_wrap.cpp -- C entry points around the REFERENCE's own DTW functions.
//
// TEST INFRASTRUCTURE.  This file is ours; it is compiled together with
// /root/reference/src/dtw.cpp (read where it lies, never copied) into
// oracle/_ref/libref_dtw.so by oracle/Makefile.  It only flattens the C++
// signatures of src/dtw.hpp:21-29 into plain C so that ctypes can call the
// real reference, and adds a threaded batch runner used as the
// cpu_baseline("reference") leg of bench.py.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include <algorithm>

#include "dtw.hpp" // from -I/root/reference/src

extern "C" {

float ref_dtw_global(const float *a, uint32_t n, const float *b, uint32_t m, int excl)
{
    return DTW_global(a, n, b, m, excl != 0);
}

float ref_dtw_global_slow(const float *a, uint32_t n, const float *b, uint32_t m, int excl)
{
    return DTW_global_slow(a, n, b, m, excl != 0);
}

float ref_dtw_slantedbanded(const float *a, uint32_t n, const float *b, uint32_t m, int radius,
                            int excl)
{
    return DTW_global_slantedbanded(a, n, b, m, radius, excl != 0);
}

float ref_dtw_banded(const float *a, uint32_t n, const float *b, uint32_t m, int radius, int excl)
{
    return DTW_global_slantedbanded_antidiagonalwise(a, n, b, m, radius, excl != 0);
}

// path_* hold n+m-1 entries; returns cost, *path_len = elements written
float ref_dtw_global_tb(const float *a, uint32_t n, const float *b, uint32_t m, int excl,
                        uint32_t *path_i, uint32_t *path_j, float *path_d, uint32_t *path_len)
{
    dtw_result r = DTW_global_tb(a, n, b, m, excl != 0);
    uint32_t k = 0;
    for (const alignment_element &e : r.alignment) {
        path_i[k] = (uint32_t)e.position.i;
        path_j[k] = (uint32_t)e.position.j;
        path_d[k] = e.difference;
        k++;
    }
    *path_len = k;
    return r.cost;
}

// Batch of score-only jobs over shared operand arrays, nthreads workers pulling
// jobs from an atomic counter (the reference's own parallelism is one task per
// read on a pthread pool, src/kthread.c:54-72; a job pool is the closest
// equivalent for a flat job list).  band_radius < 0 selects DTW_global.
struct ref_job {
    uint64_t ref_off;
    uint32_t read_off;
    uint32_t n;
    uint32_t m;
    int32_t band_radius;
    uint32_t exclude_last;
    uint32_t pad;
};

// `reps` passes over the job list inside ONE pool of threads (bench.py: a 5 M-job batch is 50 ms of work for 16
// threads, too short to time next to the threads' start-up).
void ref_batch_costs_reps(const ref_job *jobs, uint64_t n_jobs, const float *events, const float *ref,
                          float *out, int nthreads, int reps)
{
    std::atomic<uint64_t> next(0);
    const uint64_t total = n_jobs * (uint64_t)(reps > 1 ? reps : 1);
    auto work = [&]() {
        const uint64_t grain = 64; // (1 024 and 16 384 were measured: no difference)
        for (;;) {
            uint64_t s = next.fetch_add(grain);
            if (s >= total) break;
            uint64_t e = s + grain < total ? s + grain : total;
            for (uint64_t q = s; q < e; q++) {
                const uint64_t k = q % n_jobs;
                const ref_job &j = jobs[k];
                const float *a = events + j.read_off;
                const float *b = ref + j.ref_off;
                out[k] = j.band_radius < 0
                             ? DTW_global(a, j.n, b, j.m, j.exclude_last != 0)
                             : DTW_global_slantedbanded_antidiagonalwise(a, j.n, b, j.m, j.band_radius,
                                                                         j.exclude_last != 0);
            }
        }
    };
    if (nthreads <= 1) {
        work();
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) pool.emplace_back(work);
    for (auto &t : pool) t.join();
}

void ref_batch_costs(const ref_job *jobs, uint64_t n_jobs, const float *events, const float *ref,
                     float *out, int nthreads)
{
    ref_batch_costs_reps(jobs, n_jobs, events, ref, out, nthreads, 1);
}


// ---- the mapper's DTW block (src/rmap.cpp:509-530) on the host cores, with the REFERENCE's own DTW functions: the scorer
// bench.py plugs into the library's mapper for its cpu_baseline leg (rawdtw_mapper_set_scorer; same control flow, the DTW on
// the CPU).  align_chain's score-only form (rmap.cpp:181-306) around DTW_global / DTW_global_slantedbanded_antidiagonalwise,
// early exits included; one task per read on `threads` threads, as kt_for deals reads (rmap.cpp:916, kthread.c:54-72).
struct ref_anchor { uint32_t target_position, query_position; };
struct ref_scorer_ctx {
    const float *const *fwd; // forward_signals[seq] (chain.strand == 1 selects them: rmap.cpp:183-188)
    const float *const *rev;
    int border_constraint, fill_method;
    float band_radius_frac, match_bonus, min_score;
    int fused_score, threads;
    uint64_t dtw_calls; // (out) DTW calls made
};

static float ref_align_chain(const ref_anchor *anchors, uint32_t n_anchors, const float *ref_events, const float *read_events,
                             const ref_scorer_ctx &o, float min_score, uint64_t &calls)
{
    float cost = 0.0f;
    uint32_t num_aligned = 0;
    const ref_anchor &first = anchors[n_anchors - 1], &last = anchors[0];
    auto one = [&](const float *rd, uint32_t rn, const float *rf, uint32_t fm, bool excl) {
        calls++;
        if (o.fill_method == 0) return DTW_global(rd, rn, rf, fm, excl);
        const int R0 = std::max(1, (int)(rn * o.band_radius_frac)); // rmap.cpp:214,276
        return DTW_global_slantedbanded_antidiagonalwise(rd, rn, rf, fm, R0, excl);
    };
    if (o.border_constraint == 0) {
        const uint32_t fm = last.target_position - first.target_position + 1, rn = last.query_position - first.query_position + 1;
        if ((float)rn * o.match_bonus < min_score) return -1e10f; // rmap.cpp:205-209
        cost = one(read_events + first.query_position, rn, ref_events + first.target_position, fm, false);
        num_aligned = rn;
    } else {
        const uint32_t parts = n_anchors - 1;
        float attainable = (float)(last.query_position - first.query_position + 1) * o.match_bonus; // rmap.cpp:245-246
        for (uint32_t p = 0; p < parts; p++) {
            const ref_anchor &s = anchors[parts - p], &e = anchors[parts - p - 1];
            const uint32_t fm = e.target_position - s.target_position + 1, rn = e.query_position - s.query_position + 1;
            if (attainable < min_score) return -1e10f; // rmap.cpp:265-268
            const float sub = one(read_events + s.query_position, rn, ref_events + s.target_position, fm, p != parts - 1);
            cost += sub; attainable -= sub; num_aligned += rn; // rmap.cpp:279-280,292
        }
    }
    if (o.fused_score) return __builtin_fmaf((float)num_aligned, o.match_bonus, -cost); // (the -O3 -march=native build contracts rmap.cpp:306)
    const float prod = (float)num_aligned * o.match_bonus;
    return prod - cost;
}

// signature of rawdtw_scorer_fn (include/rawdtw.h); user = ref_scorer_ctx*
int ref_scorer(void *user, uint64_t n_reads, const uint64_t *chain_off, const uint64_t *anchor_off, const ref_anchor *anchors,
               const uint32_t *chain_seq, const int32_t *chain_strand, const float *const *read_events, const uint32_t *read_n_events,
               float *score, uint8_t *keep)
{
    (void)read_n_events;
    ref_scorer_ctx &o = *static_cast<ref_scorer_ctx *>(user);
    std::atomic<uint64_t> next(0), calls_all(0);
    auto work = [&]() {
        uint64_t calls = 0;
        for (;;) {
            const uint64_t r = next.fetch_add(1);
            if (r >= n_reads) break;
            float best = 0.0f; // rmap.cpp:515
            for (uint64_t c = chain_off[r]; c < chain_off[r + 1]; c++) {
                const float *ref_events = chain_strand[c] == 1 ? o.fwd[chain_seq[c]] : o.rev[chain_seq[c]];
                const float s = ref_align_chain(anchors + anchor_off[c], (uint32_t)(anchor_off[c + 1] - anchor_off[c]), ref_events, read_events[r], o, best, calls);
                score[c] = s;
                keep[c] = 0;
                if (s >= o.min_score) { if (s > best) best = s; keep[c] = 1; } // rmap.cpp:518-523
            }
        }
        calls_all += calls;
    };
    if (o.threads <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < o.threads; t++) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    o.dtw_calls += calls_all.load();
    return 0;
}

} // extern "C"
