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

} // extern "C"
