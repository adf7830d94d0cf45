/*
 * rawdtw.h -- C ABI of the MI355X-native DTW alignment engine (librawdtw.so).
 *
 * This is the drop-in boundary for RawAlign's DTW hot path.  The reference has
 * no plugin API; its boundary is the C++ free-function interface of
 * src/dtw.hpp:21-29, called from exactly one place, align_chain
 * (src/rmap.cpp:211,215,221,273,277,284).  Each entry point below names the
 * reference interface it replaces.  Plain pointers and sizes only; every
 * function returns a status code (no exceptions cross the ABI, no abort()).
 *
 * Threading: a rawdtw_ctx owns one HIP stream and is NOT re-entrant; use one
 * ctx per host thread (or serialise).  Different ctxs are independent.
 *
 * Arithmetic contract: fp32, local distance |x-y|, cell = min3 + dist, sentinel
 * = float(1e10); device code is built with -ffp-contract=off.  Costs are
 * bit-identical to the reference for finite inputs (NaN inputs: unspecified).
 */
#ifndef RAWDTW_H
#define RAWDTW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAWDTW_ABI_VERSION 2

typedef enum {
    RAWDTW_OK = 0,
    RAWDTW_ERR_INVALID = 1,     /* zero length / negative radius: the reference assert()s (dtw.cpp:79,274-277) */
    RAWDTW_ERR_DEVICE = 2,      /* HIP runtime error; see rawdtw_last_error */
    RAWDTW_ERR_OOM = 3,
    RAWDTW_ERR_RANGE = 4,       /* a job window lies outside the uploaded arenas */
    RAWDTW_ERR_UNSUPPORTED = 5, /* e.g. traceback of a banded global job (rmap.cpp:223-225 assert(false)) */
    RAWDTW_ERR_NO_DEVICE = 6
} rawdtw_status;

#define RAWDTW_FULL (-1) /* band_radius value selecting DTW_global (dtw.hpp:21) */

/* One DTW sub-problem = one call of DTW_global / DTW_global_slantedbanded_antidiagonalwise
 * (dtw.hpp:21,25) as align_chain issues it: a = read events window, b = reference
 * signal window.  32 bytes. */
typedef struct {
    uint64_t ref_off;      /* element offset of b[0] in the reference arena (rawdtw_reference_offset) */
    uint32_t read_off;     /* element offset of a[0] in the batch's event arena */
    uint32_t n;            /* a_length: read events in the window (>=1) */
    uint32_t m;            /* b_length: reference signals in the window (>=1) */
    int32_t band_radius;   /* the band_radius argument (>=0), or RAWDTW_FULL */
    uint32_t exclude_last; /* exclude_last_element */
    uint32_t reserved;     /* 0 */
} rawdtw_job_t;

typedef struct rawdtw_ctx rawdtw_ctx;
typedef struct rawdtw_plan rawdtw_plan;

/* ---- lifetime ---------------------------------------------------------- */
int rawdtw_abi_version(void);
int rawdtw_device_count(int *count);
int rawdtw_create(int device_ordinal, rawdtw_ctx **out);
/* Teardown order is free.  rawdtw_destroy waits for the context's stream, then detaches every plan and batch still alive
 * on it: their device memory and pooled workspaces are released there and then, and rawdtw_plan_destroy /
 * rawdtw_batch_destroy called afterwards only delete the host records (every other entry point refuses a detached
 * batch with RAWDTW_ERR_INVALID).  tests/abi/host_shim.cpp --teardown exercises exactly this. */
int rawdtw_destroy(rawdtw_ctx *ctx);
const char *rawdtw_last_error(const rawdtw_ctx *ctx);
const char *rawdtw_status_string(int status);
int rawdtw_sync(rawdtw_ctx *ctx);
/* tuning knobs (results never depend on them; tests/test_gpu_parity.py checks that):
 *   "plan_threads"    host threads of the batch planner (0 = from the job count and the machine, <= 16)
 *   "serial_launches" 0/1: with side streams, run a batch's launches in sequence anyway
 *   "lane_max_radius" 0..3: largest post-slant radius on the tile kernel; "lane_max_n": longest side there
 *   "lane_hi", "lane_hi_max_n": optional second tile instance for radii up to 8
 *   "micro_max_n" 0/4/8, "grp16" 0/1, "grp8" 0/1, "full_wg" 0/1, "tile_lds_floats", "tile_max_jobs", "tile_max_spans",
 *   "tile_threads" 256/512/1024: kernel selection
 *   "sort_n", "sort_r1_n", "sort_r3", "sorted_tile_jobs": optional by-shape tiles for long / rare tile jobs (default off)
 *   "device_plan" 0/1, "device_plan_min_jobs": rawdtw_batch_create takes the sync-free path (planning on the device, in LDS)
 *   "stream_tile_radius" 1..3 (default 3), "stream_threads" 256/512, "stream_blocks_per_cu" (default 0 = as many as fit:
 *   4): the device-planned batch's DTW launch over the tiles' passes (k_runs) -- tiles (512 consecutive anchors) take radii up
 *   to stream_tile_radius, the radii between that and lane_max_radius are scored a lane per job from the side list, bucketed
 *   by length over the whole batch; a pass's LDS image is 5800 floats unless "tile_lds_floats" is given
 *   "wide_blocks" (default 256): workgroups of the side list's launch (k_wide); "wide_beside" 0/1 (default 0): that launch
 *   on the context's second stream beside the tiles' launch instead of in line (measured slower)
 *   "resident_arrays" 0/1: rawdtw_batch_create's anchors / ref_base / read_base are device pointers, used in place
 *   "time_plan" 0/1: event pair around a batch's planning kernels
 *   "merge_small" 0/1: a sparse batch's tile, 16-lane-row and register-wave kernels as ONE launch (default 1)
 *   "fold_mode" 0..4: chain fold as a wave per chain, a lane per chain with 16 / 32 parts per round, a lane per chain plus a
 *   wave for each chain of at least "fold_long_parts" (768) parts (3), or (default 4) device-planned batches fold and select
 *   in one launch out of LDS and job-list batches as 3; a batch keeps the form it was created under
 *   "debug_skip_kinds": timing experiments only -- launches of the masked kinds are not issued (results wrong)
 * The environment variable RAWDTW_OPTS="name=value,..." applies options at rawdtw_create. */
int rawdtw_set_option(rawdtw_ctx *ctx, const char *name, int64_t value);
/* the ctx's hipStream_t, as void* (for event timing on the stream kernels run on) */
int rawdtw_stream(rawdtw_ctx *ctx, void **stream);
/* the device ordinal the context was created on */
int rawdtw_context_device(const rawdtw_ctx *ctx, int *device_ordinal);

/* ---- reference signals: replaces ri_idx_t.forward_signals / reverse_signals
 * (src/rawindex.h:32-34) as the `b` operand.  Uploaded once, resident in HBM.
 * strand uses the reference's convention: chain.strand==1 selects fwd
 * (rmap.cpp:182-188). ---- */
int rawdtw_upload_reference(rawdtw_ctx *ctx, uint32_t n_seq, const float *const *fwd,
                            const float *const *rev, const uint32_t *len);
int rawdtw_reference_offset(const rawdtw_ctx *ctx, uint32_t seq, int strand, uint64_t *off);
/* Adopt a device-resident arena instead (caller keeps ownership; 16-byte aligned, and readable up to the next multiple
 * of four floats: the kernels copy whole 16-byte pieces of it). */
int rawdtw_set_reference_device(rawdtw_ctx *ctx, const float *d_ref, uint64_t n_floats);
/* Several contexts on one device (one per pipeline worker, rmap.cpp:1033) share ONE resident copy: `ctx` adopts the
 * arena and the sequence table of `owner`.  An arena the library allocated (rawdtw_upload_reference, rawdtw_index_upload)
 * is reference-counted: it lives until the last context using it uploads another reference or is destroyed, so the
 * owner may go first.  An arena adopted with rawdtw_set_reference_device stays the caller's to keep alive. */
int rawdtw_share_reference(rawdtw_ctx *ctx, const rawdtw_ctx *owner);

/* ---- index file reader: the part of ri_idx_load (src/rawindex.cpp:317-377) the DTW path needs --
 * header, sequence table and the per-sequence forward/reverse signal arrays of a RawAlign `.ind`
 * file (format written by ri_idx_dump, src/rawindex.cpp:275-315: magic "RI" (2 bytes, rawindex.h:7-8), 8 x u32 parameters
 * {w,e,n,q,lq,k,n_seq,flag}, then per sequence {u8 name_len, name, u32 len, len fwd floats, len rev
 * floats}; the hash buckets that follow are not read).  rawdtw_index_upload streams the signal
 * arrays straight into the ctx's reference arena, so a human-size index never sits in host memory. */
typedef struct rawdtw_index rawdtw_index;
int rawdtw_index_open(const char *path, rawdtw_index **out);
int rawdtw_index_info(const rawdtw_index *idx, uint32_t *n_seq, uint32_t pars[8]);
int rawdtw_index_seq(const rawdtw_index *idx, uint32_t i, const char **name, uint32_t *len);
int rawdtw_index_upload(rawdtw_ctx *ctx, const rawdtw_index *idx);
/* host copy of one strand's array (tests / CPU baseline); out must hold len floats */
int rawdtw_index_read_signal(const rawdtw_index *idx, uint32_t i, int strand, float *out);
int rawdtw_index_close(rawdtw_index *idx);

/* ---- read events: the `a` operand (p->events[read].values, rmap.cpp:517) of all
 * reads of a batch, concatenated by the caller. ---- */
int rawdtw_upload_events(rawdtw_ctx *ctx, const float *h_events, uint64_t n_floats);
/* (a caller's device array: 16-byte aligned, readable up to the next multiple of four floats) */
int rawdtw_set_events_device(rawdtw_ctx *ctx, const float *d_events, uint64_t n_floats);
/* Incremental form.  A read's event array only grows (ri_map_frag appends each chunk's events, rmap.cpp:554-567), so a
 * mapper that keeps one slot per read in the arena uploads only the round's NEW events: reserve grows the arena to
 * n_floats (contents kept; the arena's logical size, against which job windows are checked, becomes at least
 * n_floats); append copies h_new[seg_src_off[s] .. seg_src_off[s+1]) to arena offset seg_dst_off[s] for every
 * segment s (one H2D copy of the packed new events + a scatter kernel), asynchronously on the ctx stream.  h_new and
 * the segment tables must stay valid until the stream has passed the call (rawdtw_sync / a fetch); memory from
 * rawdtw_host_alloc makes the copies asynchronous. */
int rawdtw_events_reserve(rawdtw_ctx *ctx, uint64_t n_floats);
int rawdtw_events_append(rawdtw_ctx *ctx, const float *h_new, uint64_t n_new, uint32_t n_segments,
                         const uint64_t *seg_src_off, const uint32_t *seg_dst_off);
/* Page-locked host memory for the arrays handed to rawdtw_batch_create / rawdtw_events_append and for the result
 * arrays of rawdtw_batch_fetch.  Any host memory works; with pinned memory the copies do not block the caller. */
int rawdtw_host_alloc(uint64_t bytes, void **out);
int rawdtw_host_free(void *p);
/* 1 when p points into page-locked host memory (rawdtw_host_alloc's, or any the HIP runtime knows), else 0 */
int rawdtw_host_is_page_locked(const void *p);

/* ---- score-only batches: replaces the calls at rmap.cpp:211,215,273,277 ---- */
/* One shot: upload events, bin + launch, copy costs back (out_cost[k] for jobs[k]). */
int rawdtw_score_batch(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs,
                       const float *h_events, uint64_t n_events, float *out_cost);

/* Split form, for callers that keep inputs resident (and for benchmarking):
 * plan_create validates and size-bins the jobs and uploads device descriptors;
 * plan_run only launches kernels on the ctx stream (asynchronous); costs stay in
 * HBM, out[k] for jobs[k]; plan_fetch waits and copies them to the host. */
typedef struct {
    uint64_t n_jobs;
    uint64_t cells;            /* DP cells the batch evaluates (exact, band cell sets counted) */
    uint64_t algorithmic_bytes; /* sum over jobs of 4(n+m)+4+32 */
    uint64_t n_lane_jobs;      /* jobs on the lane-per-job banded kernel */
    uint64_t n_wave_band_jobs; /* jobs on the wave-per-job banded kernel */
    uint64_t n_full_jobs;      /* jobs on the full-matrix wavefront kernel */
    uint64_t workspace_bytes;
    uint32_t n_launches;
} rawdtw_plan_info_t;

int rawdtw_plan_create(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs,
                       rawdtw_plan **out);
int rawdtw_plan_info(const rawdtw_plan *plan, rawdtw_plan_info_t *info);
/* Host half of rawdtw_plan_create only -- no device is touched and nothing is scored.  Bins the
 * jobs against arenas of n_events / n_reference floats with `threads` planner threads and the
 * given planner options (names as for rawdtw_set_option), then checks every invariant the
 * kernels rely on (each job planned exactly once, launches partition the plan, every tile's
 * windows staged inside its LDS budget at the right place).  Returns the status plan_create
 * would return; `message` (optional) receives the reason of a failure.  The extra option
 * "verify" = 0 skips the self-check and the cell count (to time the planner alone).  For
 * host-side tests and for sizing a batch before a device is attached. */
int rawdtw_plan_dry_run(uint64_t n_events, uint64_t n_reference, const rawdtw_job_t *jobs,
                        uint64_t n_jobs, int threads, const char *const *option_names,
                        const int64_t *option_values, uint32_t n_options,
                        rawdtw_plan_info_t *info, uint64_t *n_tiles, char *message,
                        uint32_t message_cap);
int rawdtw_plan_run(rawdtw_ctx *ctx, rawdtw_plan *plan);
int rawdtw_plan_fetch(rawdtw_ctx *ctx, rawdtw_plan *plan, float *out_cost);
/* device pointer to the costs (job order) and the host array mapping launch order -> job index */
int rawdtw_plan_device_costs(const rawdtw_plan *plan, const float **d_cost,
                             const uint32_t **h_order);
/* per-launch kernel names and HIP-event durations (ms) of the most recent
 * rawdtw_plan_run_timed; arrays hold up to n_launches entries */
int rawdtw_plan_run_timed(rawdtw_ctx *ctx, rawdtw_plan *plan, float *launch_ms,
                          uint32_t *launch_kind, uint32_t cap);
int rawdtw_plan_destroy(rawdtw_plan *plan);

/* ---- traceback batches: replaces DTW_global_tb (dtw.hpp:28) at rmap.cpp:221,284.
 * Jobs must have band_radius == RAWDTW_FULL.  The direction matrix is kept as a
 * packed 2-bit buffer in HBM and walked on the GPU.  path_off[k] (caller
 * supplied, elements) locates job k's path inside path_i/path_j/path_d, which
 * must have room for n+m-1 entries per job; entries come out in forward order,
 * i indexing a (read window), j indexing b (reference window). ---- */
int rawdtw_traceback_batch(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs,
                           const float *h_events, uint64_t n_events, float *out_cost,
                           const uint64_t *path_off, uint32_t *path_len, uint32_t *path_i,
                           uint32_t *path_j, float *path_d);

/* The same with the paths in the form they leave the device in: per element ONE byte, the step from the element before it
 * (bit 0: i advanced, bit 1: j advanced; element 0 of a global path is (0, 0): dtw.cpp:640-655, its step byte is 0), and its
 * distance -- 5 bytes an element over the bus and into the caller's arrays instead of 12.  A consumer that walks the path in
 * order (the aln:s: writer, rmap.cpp:580-592) needs nothing else. */
int rawdtw_traceback_batch_steps(rawdtw_ctx *ctx, const rawdtw_job_t *jobs, uint64_t n_jobs,
                                 const float *h_events, uint64_t n_events, float *out_cost,
                                 const uint64_t *path_off, uint32_t *path_len, uint8_t *path_step, float *path_d);

/* Device time of the most recent rawdtw_traceback_batch on this context (HIP events on its stream, summed over its
 * sub-batches): fill = the full-matrix kernels that write the packed 2-bit directions, walk = the traceback walk and the
 * path finish; direction_bytes = the packed direction buffers of its jobs; path_elements = the elements of all paths. */
int rawdtw_traceback_timing(const rawdtw_ctx *ctx, float *fill_ms, float *walk_ms, uint64_t *direction_bytes,
                            uint64_t *path_elements);

/* ---- single-call drop-ins with the reference's own signatures flattened
 * (dtw.hpp:21,25,28).  Host pointers; convenient, not fast. ---- */
int rawdtw_dtw_global(rawdtw_ctx *ctx, const float *a, uint32_t n, const float *b, uint32_t m,
                      int exclude_last, float *cost);
int rawdtw_dtw_global_slantedbanded_antidiagonalwise(rawdtw_ctx *ctx, const float *a, uint32_t n,
                                                     const float *b, uint32_t m, int band_radius,
                                                     int exclude_last, float *cost);
int rawdtw_dtw_global_tb(rawdtw_ctx *ctx, const float *a, uint32_t n, const float *b, uint32_t m,
                         int exclude_last, float *cost, uint32_t *path_len, uint32_t *path_i,
                         uint32_t *path_j, float *path_d);

/* ---- host-side mirror of align_chain / the DTW block of gen_chains
 * (src/rmap.cpp:181-313, 509-530): job decomposition and exact replay.  Pure
 * host code, no device work. ---- */
typedef struct {
    uint32_t target_position;
    uint32_t query_position;
} rawdtw_anchor_t; /* ri_anchor_t, rmap.h:21-27; stored end-first like the reference */

typedef struct {
    int border_constraint;  /* 0 global, 1 sparse (roptions.h:21-23) */
    int fill_method;        /* 0 full, 1 banded (roptions.h:25-26) */
    float band_radius_frac; /* roptions.c:51 */
    float match_bonus;      /* roptions.c:52 */
    float min_score;        /* roptions.c:53 */
    int fused_score;        /* 1: score = fmaf(n, bonus, -cost) as the reference's -O3 -march=native build contracts it */
} rawdtw_align_opt_t;

/* Number of DTW jobs align_chain issues for a chain (before any early exit). */
uint32_t rawdtw_chain_job_count(const rawdtw_align_opt_t *opt, uint32_t n_anchors);
/* Emit those jobs. ref_base = arena offset of the chain's strand array
 * (rawdtw_reference_offset), read_base = arena offset of the read's events. */
int rawdtw_chain_build_jobs(const rawdtw_align_opt_t *opt, const rawdtw_anchor_t *anchors,
                            uint32_t n_anchors, uint64_t ref_base, uint32_t read_base, int cigar,
                            rawdtw_job_t *jobs_out);
/* Fold per-job costs back into align_chain's result, replaying its early exits
 * (rmap.cpp:205-209, 265-268) with min_score. Returns the alignment score. */
float rawdtw_chain_replay(const rawdtw_align_opt_t *opt, const rawdtw_anchor_t *anchors,
                          uint32_t n_anchors, const float *job_cost, float min_score);
/* The sequential loop of rmap.cpp:515-524 over one read's chains, given in
 * evaluation order; chain c owns job_cost[job_off[c]..job_off[c+1]).  Writes
 * score[c], keep[c]; returns the number kept. */
uint32_t rawdtw_read_replay(const rawdtw_align_opt_t *opt, uint32_t n_chains,
                            const uint32_t *anchor_off, const rawdtw_anchor_t *anchors,
                            const uint64_t *job_off, const float *job_cost, float *score,
                            uint8_t *keep);


/* ---- consumers of alignment_score downstream of the DTW block (pure host code) ---- */
typedef struct {
    float chaining_score;
    float alignment_score;
    uint32_t reference_sequence_index;
    uint32_t start_position;
    uint32_t end_position;
    uint32_t n_anchors;
    int32_t strand;
    uint32_t mapq; /* out: comp_mapq's value for the first kept chain */
    uint32_t tag;  /* caller's handle; travels with the record through the sort */
} rawdtw_chain_t; /* the scalar fields of ri_chain_t (src/rmap.h:29-46) */

typedef struct {
    int evaluate_chains;      /* opt->flag & RI_M_DTW_EVALUATE_CHAINS */
    float min_bestmap_ratio;  /* roptions.c:28 (1.2) */
    float min_meanmap_ratio;  /* roptions.c:31 (5)   */
    uint32_t min_chain_anchor; /* roptions.c:25 (2)  */
} rawdtw_select_opt_t;

/* gen_primary_chains + comp_mapq (src/rmap.cpp:90-128, 65-88): sorts `chains` in place exactly as
 * std::sort(..., std::greater<ri_chain_t>()) does, writes the indices (into the SORTED array) of
 * the primary chains to kept[] and sets chains[kept[0]].mapq.  Returns the number kept. */
uint32_t rawdtw_gen_primary_chains(rawdtw_chain_t *chains, uint32_t n_chains,
                                   const rawdtw_select_opt_t *opt, uint32_t *kept);
/* is_mapped_with_high_confidence (src/rmap.cpp:594-665) over the primary chains, best first. */
int rawdtw_is_mapped_with_high_confidence(const rawdtw_chain_t *primary, uint32_t n_chains,
                                          const rawdtw_select_opt_t *opt);
/* find_outlier (src/sequence_until.c:4-18): x[point][dim], n dims, m points.  The first form is the source's
 * arithmetic (one rounded product and one rounded add per element, in order; equal bit for bit to the reference
 * file compiled with -ffp-contract=off).  The second is what the reference's default build computes when the
 * compiler has FMA (GCC -O3 -march=native on AVX2 hosts): same order, but elements past the last full group
 * of four are accumulated with one fused multiply-add each; equal bit for bit to that build. */
float rawdtw_find_outlier(const float *const *x, uint32_t n, uint32_t m);
float rawdtw_find_outlier_contracted(const float *const *x, uint32_t n, uint32_t m);

/* ---- chaining DP of gen_chains (src/rmap.cpp:430-507) and traceback_chains (src/rmap.cpp:130-173) for
 * one (reference sequence, strand): anchors must be sorted by (target_position, query_position) as
 * rmap.cpp:396-401 does.  Emits up to num_best_chains chains; chain k owns
 * out_anchors[out_off[k]..out_off[k+1]) (end-first).  max_chaining_score is the running maximum over
 * all (sequence, strand) pairs of the read and is updated in place (rmap.cpp:431,485-487).
 * Returns the number of chains written, or a negative value when an output array is too small. */
typedef struct {
    int max_gap_length;        /* roptions.c:13 (2000) */
    int max_target_gap_length; /* roptions.c:14 (5000) */
    int chaining_band_length;  /* roptions.c:15 (5000) */
    int max_num_skips;         /* roptions.c:16 (25)   */
    int min_num_anchors;       /* roptions.c:17 (2)    */
    int num_best_chains;       /* roptions.c:18 (3)    */
    float min_chaining_score;  /* roptions.c:19 (10)   */
    int e;                     /* ri->e: events per seed */
    int disable_score_filtering; /* RI_M_DISABLE_CHAININGSCORE_FILTERING */
} rawdtw_chain_opt_t;

typedef struct {
    float chaining_score;
    uint32_t start_position, end_position, n_anchors;
} rawdtw_chain_out_t;

int rawdtw_chain_anchors(const rawdtw_chain_opt_t *opt, const rawdtw_anchor_t *anchors, uint32_t n_anchors,
                         float *max_chaining_score, rawdtw_chain_out_t *out_chains, uint64_t *out_off,
                         rawdtw_anchor_t *out_anchors, uint32_t chains_cap, uint64_t anchors_cap);

/* The evaluation order of one read's chains: the permutation std::sort (libstdc++, unstable)
 * produces for the comparator a.chaining_score > b.chaining_score (rmap.cpp:512). */
int rawdtw_sort_by_chaining_score(const float *chaining_score, uint32_t n_chains, uint32_t *perm_out);

/* ---- the same on the device for a whole chunk round (rawdtw_chain.hip): per read the anchor sort
 * (rmap.cpp:396-401), the chaining DP and traceback_chains per (sequence, strand) list (rmap.cpp:430-507, 130-173) with the
 * read's running maximum carried from list to list, and the evaluation order (rmap.cpp:512) -- a wave a read -- and all
 * reads' chains laid out in device memory as ONE candidate batch: what rawdtw_batch_submit_device takes, so the anchors
 * never cross PCIe on their way into the DTW.
 *   in  (host): read r's seeds seeds[seed_off[r] .. seed_off[r+1]), UNSORTED (key = sequence * 2 + strand as the caller numbers
 *        its reference arrays); read_base[r] = the read's first event in the event arena; key_base[key] = the arena offset of
 *        that key's reference array (rawdtw_reference_offset)
 *   out (host): chain_off[n_reads + 1]; anchor_off[n_chains + 1] and recs[n_chains] (at most chains_cap chains; 32 a read is the
 *        device's own limit), chains of a read in evaluation order; anchors[...] (end-first per chain; may be NULL: room for
 *        seed_off[n_reads] entries)
 *   out (device, the context's, valid until its next rawdtw_chain_round): *d_anchors, *d_ref_base, *d_read_base
 * The call returns when the host arrays are filled.  Results equal rawdtw_chain_anchors list by list and
 * rawdtw_sort_by_chaining_score read by read, bit for bit.  RAWDTW_ERR_UNSUPPORTED (the out arrays hold nothing of use): a read with more than
 * 2 048 seeds, with more than 32 chains, or with more than 16 chains two of which have equal scores (std::sort's order of
 * equal elements is an insertion sort's only up to 16) -- chain that round on the host.  RAWDTW_ERR_INVALID: a chain on a key
 * that is not below n_keys. */
typedef struct { uint32_t key, target_position, query_position; } rawdtw_seed_t;                       /* 12 bytes */
typedef struct { float chaining_score; uint32_t key, start_position, end_position, n_anchors; } rawdtw_chain_rec_t; /* 20 bytes */
int rawdtw_chain_round(rawdtw_ctx *ctx, const rawdtw_chain_opt_t *opt, uint64_t n_reads, const uint64_t *seed_off,
                       const rawdtw_seed_t *seeds, const uint32_t *read_base, uint32_t n_keys, const uint64_t *key_base,
                       uint64_t *chain_off, uint64_t *anchor_off, rawdtw_chain_rec_t *recs, uint64_t chains_cap,
                       rawdtw_anchor_t *anchors, const rawdtw_anchor_t **d_anchors, const uint64_t **d_ref_base,
                       const uint32_t **d_read_base);
/* The same in two halves, for a host that has other work while the device chains: _begin enqueues everything (the input
 * arrays must stay as they are until _end) and returns; _end waits and reports.  When chain_off, anchor_off, recs and
 * anchors are page-locked (rawdtw_host_alloc) the device writes them itself and _end is one wait; else they are copied in
 * _end.  One round at a time a context. */
int rawdtw_chain_round_begin(rawdtw_ctx *ctx, const rawdtw_chain_opt_t *opt, uint64_t n_reads, const uint64_t *seed_off,
                             const rawdtw_seed_t *seeds, const uint32_t *read_base, uint32_t n_keys, const uint64_t *key_base,
                             uint64_t *chain_off, uint64_t *anchor_off, rawdtw_chain_rec_t *recs, uint64_t chains_cap,
                             rawdtw_anchor_t *anchors);
int rawdtw_chain_round_end(rawdtw_ctx *ctx, const rawdtw_anchor_t **d_anchors, const uint64_t **d_ref_base,
                           const uint32_t **d_read_base);

/* Batched forms over many reads (what rmap.cpp's per-read worker does for every read of a
 * mini-batch, hoisted around one GPU submission).  Chains are listed read by read, each read's
 * chains already in the reference's evaluation order (std::sort by chaining_score descending,
 * rmap.cpp:512).  chain c owns anchors[anchor_off[c]..anchor_off[c+1]); read r owns chains
 * [chain_off[r], chain_off[r+1]).  ref_base[c]/read_base[c]: arena offsets for chain c.
 * job_off (n_chains+1 entries) is written by build and read by replay. */
int rawdtw_batch_build_jobs(const rawdtw_align_opt_t *opt, uint64_t n_chains, const uint64_t *anchor_off,
                            const rawdtw_anchor_t *anchors, const uint64_t *ref_base,
                            const uint32_t *read_base, uint64_t *job_off, rawdtw_job_t *jobs_out,
                            uint64_t jobs_cap, uint64_t *n_jobs_out);
int rawdtw_batch_replay(const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                        const uint64_t *anchor_off, const rawdtw_anchor_t *anchors,
                        const uint64_t *job_off, const float *job_cost, float *score, uint8_t *keep);


/* ---- whole-batch form: the DTW block of gen_chains (rmap.cpp:509-530) for every read of a
 * mini-batch in one submission -- job decomposition on the host, DTW scoring, the align_chain
 * fold and the per-read accept/cut loop all on the device.  Inputs as rawdtw_batch_build_jobs;
 * events and reference arenas must already be set on the ctx.  Outputs per chain: score[c]
 * (chain.alignment_score, -1e10 when cut) and keep[c] (survives dtw_min_score). ---- */
/* rawdtw_batch_create is asynchronous for sparse + banded batches (the default options): it enqueues the copies of the
 * five arrays and the planning kernels on the ctx stream and returns -- no host synchronisation, and no allocation
 * once the context's workspace pool is warm.  Consequences for the caller: the five arrays must stay valid and
 * unchanged until rawdtw_batch_fetch has returned (or the batch is destroyed); an invalid batch (anchors not ascending,
 * a window outside the arenas) is reported by rawdtw_batch_fetch with the status and message rawdtw_plan_create would
 * give.  Other modes (global border constraint, full fill) and "device_plan"=0 plan on the host inside create.
 * The arenas may be re-uploaded or grown between create and run (every run reads the context's current arrays); a run
 * after an arena SHRANK below the size the batch was planned against returns RAWDTW_ERR_INVALID. */
typedef struct rawdtw_batch rawdtw_batch;
int rawdtw_batch_create(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads,
                        const uint64_t *chain_off, const uint64_t *anchor_off,
                        const rawdtw_anchor_t *anchors, const uint64_t *ref_base,
                        const uint32_t *read_base, rawdtw_batch **out);
/* Self-check of a batch's plan (tests, bring-up): downloads the tile records the kernels will read and
 * checks them against `jobs` (the batch's job list as rawdtw_batch_build_jobs gives it): every job in
 * exactly one launch, every window staged inside its tile's LDS image at the right place.
 * *device_planned tells whether the tile class was planned on the device. */
int rawdtw_batch_verify_plan(rawdtw_ctx *ctx, const rawdtw_batch *batch, const rawdtw_job_t *jobs,
                             uint64_t n_jobs, int *device_planned, char *message, uint32_t message_cap);
int rawdtw_batch_info(const rawdtw_batch *batch, rawdtw_plan_info_t *info, uint64_t *n_chains);
int rawdtw_batch_run(rawdtw_ctx *ctx, rawdtw_batch *batch);
/* as rawdtw_plan_run_timed, with two more launches (kinds 6, 7) for fold and select */
int rawdtw_batch_run_timed(rawdtw_ctx *ctx, rawdtw_batch *batch, float *launch_ms,
                           uint32_t *launch_kind, uint32_t cap, uint32_t *n_launches);
/* `reps` back-to-back runs with no host synchronisation in between, one at the end.  When
 * launch_ms != NULL a HIP event pair brackets every launch on the ctx stream and launch_ms[i]
 * receives launch i's MEAN duration over the reps. */
int rawdtw_batch_run_reps(rawdtw_ctx *ctx, rawdtw_batch *batch, uint32_t reps, float *launch_ms,
                          uint32_t *launch_kind, uint32_t cap, uint32_t *n_launches);
/* Pipelined form of the above: enqueue one run without any host synchronisation (with a HIP
 * event pair per launch when timed != 0), call it as often as wanted, also alternating between
 * batches of different contexts so that several mini-batches are in flight (the reference keeps two,
 * rmap.cpp:1033); then, after rawdtw_sync(), collect the mean per-launch durations of all runs
 * enqueued since the last collect. */
int rawdtw_batch_enqueue(rawdtw_ctx *ctx, rawdtw_batch *batch, int timed);
int rawdtw_batch_collect(rawdtw_ctx *ctx, rawdtw_batch *batch, float *launch_ms, uint32_t *launch_kind,
                         uint32_t cap, uint32_t *n_launches, uint32_t *n_runs);
/* static facts about launch i of the batch (same indexing as launch_ms): kind, parameter
 * (band radius / rows per lane / LDS floats), jobs, their algorithmic bytes and DP cells */
int rawdtw_batch_launch_stats(const rawdtw_batch *batch, uint32_t i, uint32_t *kind, int32_t *param,
                              uint64_t *n_jobs, uint64_t *algorithmic_bytes, uint64_t *cells);
/* job_cost may be NULL; otherwise receives the per-job costs too */
int rawdtw_batch_fetch(rawdtw_ctx *ctx, rawdtw_batch *batch, float *score, uint8_t *keep,
                       float *job_cost);
/* GPU time of the batch's planning kernels (HIP events; 0 unless the option "time_plan" was set before create): the scan of
 * the anchor list, the side list's class order, the passes' records and copy orders.  The wide bands' launch that goes out
 * between them for the batch's first run is DTW work: rawdtw_batch_wide_ms. */
int rawdtw_batch_plan_ms(rawdtw_ctx *ctx, rawdtw_batch *batch, float *ms);
int rawdtw_batch_wide_ms(rawdtw_ctx *ctx, rawdtw_batch *batch, float *ms);
/* Diagnostics of a device-planned batch (waits for its planning kernels): the planner's counter block (rawdtw_internal.h,
 * StreamCounter).  The layout is the library's own and moves between versions: look a word up by NAME with
 * rawdtw_batch_stream_counter_index -- "bad" first invalid part (~0 none), "overflow" first tile over a capacity (~0 none),
 * "unsupported" parts with a band nobody takes, "side_jobs" side-list entries, "class0" the first of the per-class totals,
 * "cells", "tile_jobs" / "tile_bytes" / "side_bytes" (reporting, filled on request), "todo" listed tiles, "reused" parts
 * taken over from the round before, "pool" passes beyond one a tile, "stamp0" the first of ten phase-stamp words -- which
 * returns -1 for a name it does not know.  *n_out = 0 for a batch planned on the host. */
int rawdtw_batch_stream_counter_index(const char *name);
int rawdtw_batch_stream_counters(rawdtw_ctx *ctx, rawdtw_batch *batch, uint64_t *out, uint32_t cap, uint32_t *n_out);
int rawdtw_batch_destroy(rawdtw_batch *batch);
/* ---- compact hand-over of the anchor lists.  A mini-batch's anchors are its largest array (8 bytes an anchor: as much
 * as the round's new events), and consecutive anchors of a chain differ by little: the chaining DP bounds the gaps
 * (rmap.cpp:456-472, roptions.c:13-15).  The compact form sends, per chain, its first entry whole (= the chain's END:
 * chains are stored end-first, rmap.cpp:193-196) and for every further entry the step back from the entry before it as
 * two bytes (query step, target step), 2 bytes an anchor instead of 8; every RAWDTW_COMPACT_STRIDE-th entry of the flat
 * list is sent whole as well (the decoder works in units of that many); a step of 255 or more in either component is an
 * escape: the entry's two steps are listed in `wide`, ascending by index.  rawdtw_anchors_pack builds the form (pure host
 * code; RAWDTW_ERR_INVALID when a chain's positions do not descend along its list -- such a chain no mapper produces, send
 * the batch with rawdtw_batch_submit -- and RAWDTW_ERR_RANGE when wide_cap is too small, *n_wide then says how many).
 * rawdtw_batch_submit_compact = rawdtw_batch_submit with the lists in that form: they are decoded on the device, inside the
 * scan of the anchor list. ---- */
#define RAWDTW_COMPACT_STRIDE 8192u
typedef struct {
    uint32_t index;                  /* position in the flat anchor list */
    uint32_t query_step, target_step; /* anchors[index - 1] - anchors[index], per component */
} rawdtw_wide_step_t;
int rawdtw_anchors_pack(uint64_t n_chains, const uint64_t *anchor_off, const rawdtw_anchor_t *anchors,
                        rawdtw_anchor_t *heads /* n_chains */, rawdtw_anchor_t *unit_abs /* ceil(n_anchors / STRIDE) */,
                        uint16_t *steps /* n_anchors: query step | target step << 8 */, rawdtw_wide_step_t *wide,
                        uint64_t wide_cap, uint64_t *n_wide);
/* the inverse, on the host (tests; the library's own fallback paths) */
int rawdtw_anchors_unpack(uint64_t n_chains, const uint64_t *anchor_off, const rawdtw_anchor_t *heads,
                          const rawdtw_anchor_t *unit_abs, const uint16_t *steps, const rawdtw_wide_step_t *wide,
                          uint64_t n_wide, rawdtw_anchor_t *anchors_out);
int rawdtw_batch_submit_compact(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                                const uint64_t *anchor_off, const rawdtw_anchor_t *heads, const rawdtw_anchor_t *unit_abs,
                                const uint16_t *steps, const rawdtw_wide_step_t *wide, uint64_t n_wide,
                                const uint64_t *ref_base, const uint32_t *read_base, rawdtw_batch **out);

/* ---- chunk rounds.  A read is consumed chunk by chunk (rmap.cpp:685-693) and every round re-aligns every surviving chain
 * from scratch over all its anchors (rmap.cpp:516-517), although a part between two anchors that were there the round before
 * has the same operands (events are append-only, rmap.cpp:554-567; the read keeps its place in the event arena) and
 * therefore the same cost.  A chain grows at its END from round to round (rmap.cpp:344-357 re-seeds the chaining with the
 * previous chains' anchors), i.e. at the FRONT of its end-first list: what a chain shares with the chain it continues is a
 * run of leading parts counted from its START = the tail of its list.
 *
 * rawdtw_round_match_chains (pure host code) finds, per chain of the new round, the chain of the round before it continues
 * (same read, same bases, same start anchor, the longest common tail) and VALIDATES the common tail anchor by anchor:
 * carry[c] says how many leading parts are taken over and where their costs lie in the previous batch.  It also packs what
 * the device still needs of the chain's list -- its first n_anchors - (parts + 1) entries (the NEW anchors) followed, when
 * parts > 0, by ONE more entry, the junction (the end anchor of the carried stretch: the new part behind it starts there) --
 * into new_anchors (chain c owns new_anchors[new_off[c] .. new_off[c + 1])).  A part that was not its chain's last then
 * and is now cannot be taken over (exclude_last_element, rmap.cpp:270: there is no exact way back) and is counted out; a part
 * that was the last then and is not now loses its last cell's distance on the device exactly as the DTW functions take it off
 * (dtw.cpp:514-519).
 *
 * rawdtw_batch_submit_carry = rawdtw_batch_submit for such a round.  ONLY the new anchors and the junctions cross the bus, and
 * the device's whole planning and scoring pipeline (scan, side list, passes, DTW launches) runs on that SHORT list -- a carried
 * round costs what its new parts cost; one gather launch then lays every chain's costs out in full (the carried stretch out
 * of `prev`'s cost array, one contiguous copy a chain, then the new parts') for the unchanged fold and accept/cut loop.
 * Results are bit-identical to scoring everything again.  Preconditions (else RAWDTW_ERR_UNSUPPORTED, nothing enqueued:
 * submit the round whole with rawdtw_batch_submit): `prev` is a batch of this context that was scored on the device-planned
 * path with the same options (rawdtw_batch_can_carry tells), has been run and is not destroyed.  carry[] is trusted as
 * rawdtw_round_match_chains wrote it -- the device only checks that the counts add up and stay inside `prev`'s cost array; a
 * caller that invents it gets wrong costs.  `anchor_off` are the offsets of the FULL lists (n_chains + 1 entries); `anchors` the
 * full lists themselves on the host, read only if the batch has to be redone through the job-list path (a band nobody takes,
 * a list over a capacity): they may be NULL, the fetch then fails with RAWDTW_ERR_UNSUPPORTED instead.  All host arrays must
 * stay valid until the batch is fetched; `prev` until this batch is fetched or destroyed (its cost array is read by this
 * batch's gather launch).  rawdtw_batch_round_stats: the parts scored and the parts taken over. ---- */
#define RAWDTW_NO_CHAIN (~(uint64_t)0)
typedef struct {
    uint64_t prev_src;     /* index in the previous batch's (full) anchor list of the carried stretch's first entry; RAWDTW_NO_CHAIN: none */
    uint32_t parts;        /* leading parts (from the chain's start) whose costs are taken over; 0: none */
    uint32_t flags;        /* bit 0: the stretch's first part was its chain's last then and is not now (loses its last cell's distance) */
    rawdtw_anchor_t start; /* the chain's start anchor (the last entry of its full list): the fold's span needs it (rmap.cpp:245) */
} rawdtw_carry_t;
/* read r of the new round is read prev_read[r] of the previous one (or RAWDTW_NO_CHAIN: a new read); new_anchors must have
 * room for anchor_off[n_chains] entries, new_off for n_chains + 1 */
int rawdtw_round_match_chains(uint64_t n_reads, const uint64_t *chain_off, const uint64_t *anchor_off, const rawdtw_anchor_t *anchors,
                              const uint64_t *ref_base, const uint32_t *read_base, const uint64_t *prev_read,
                              const uint64_t *prev_chain_off, const uint64_t *prev_anchor_off, const rawdtw_anchor_t *prev_anchors,
                              const uint64_t *prev_ref_base, const uint32_t *prev_read_base, rawdtw_carry_t *carry, uint64_t *new_off,
                              rawdtw_anchor_t *new_anchors);
/* 1 when `prev` can serve as the previous batch of a rawdtw_batch_submit_carry with options `opt` (waits for nothing; a batch
 * not fetched yet may still turn out declined: the carried batch then falls back by itself) */
int rawdtw_batch_can_carry(const rawdtw_ctx *ctx, const rawdtw_batch *prev, const rawdtw_align_opt_t *opt);
int rawdtw_batch_submit_carry(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                              const uint64_t *anchor_off, const rawdtw_anchor_t *anchors /* full lists: fallback only, may be NULL */,
                              const uint64_t *new_off, const rawdtw_anchor_t *new_anchors, const uint64_t *ref_base,
                              const uint32_t *read_base, const rawdtw_batch *prev, const rawdtw_carry_t *carry, rawdtw_batch **out);
int rawdtw_batch_round_stats(rawdtw_ctx *ctx, rawdtw_batch *batch, uint64_t *parts_scored, uint64_t *parts_reused);

/* The two calls a pipelined host makes per mini-batch: submit = rawdtw_batch_create +
 * rawdtw_batch_run (everything enqueued, nothing waited for; O(1) host work for sparse + banded batches), and, when the
 * worker's slot comes round again, fetch_destroy = rawdtw_batch_fetch of score / keep + rawdtw_batch_destroy. */
int rawdtw_batch_submit(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                        const uint64_t *anchor_off, const rawdtw_anchor_t *anchors, const uint64_t *ref_base,
                        const uint32_t *read_base, rawdtw_batch **out);
int rawdtw_batch_fetch_destroy(rawdtw_ctx *ctx, rawdtw_batch *batch, float *score, uint8_t *keep);
/* rawdtw_batch_submit with anchors / ref_base / read_base in DEVICE memory (rawdtw_chain_round's arrays, or the caller's
 * own): used in place, they must stay as they are until the batch is fetched.  chain_off / anchor_off are host arrays. */
int rawdtw_batch_submit_device(rawdtw_ctx *ctx, const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                               const uint64_t *anchor_off, const rawdtw_anchor_t *d_anchors, const uint64_t *d_ref_base,
                               const uint32_t *d_read_base, rawdtw_batch **out);

/* ---- the chunk-round mapping loop on the host side of the library: the control flow of map_worker_for / ri_map_frag /
 * gen_chains (src/rmap.cpp:667-822, 545-578, 315-541) turned inside out so that every chunk round makes ONE device
 * submission for all active reads, and the PAF line of a read (src/rmap.cpp:696-801, 950-965).  The caller keeps event
 * detection and seeding (revent.c, rsketch.c, rawindex.cpp) and hands in, per round and active read, the chunk's events and
 * seed hits; the mapper appends the events to the read's slot in the event arena (rmap.cpp:554-567), re-seeds with the
 * previous chains' anchors (344-357), chains (396-507), orders the chains (512), scores them all in one batch on the
 * device (509-530; with `carry` the unchanged parts' costs are taken over from the round before), and finishes the round:
 * gen_primary_chains, comp_mapq, the stop rule (532-541, 692).  rawdtw_mapper_finish runs the --dtw-output-cigar
 * traceback of every mapped read's best chain (715-717); rawdtw_mapper_paf writes a read's line (mt:f:, wall-clock in
 * the reference, as 0).  The reference arrays must be on the context (rawdtw_upload_reference / rawdtw_index_upload);
 * seq_len[s] is the length of sequence s's signal arrays.  rawalign_amd/mapper.py is the Python mirror. ---- */
typedef struct rawdtw_mapper rawdtw_mapper;
typedef struct {
    int flag;                  /* RI_M_DTW_EVALUATE_CHAINS 0x2 | RI_M_DTW_OUTPUT_CIGAR 0x4 | RI_M_DTW_LOG_SCORES 0x8 (roptions.h:13-15) */
    rawdtw_align_opt_t align;
    rawdtw_chain_opt_t chain;
    float min_bestmap_ratio, min_meanmap_ratio; /* roptions.c:28,31 */
    uint32_t min_chain_anchor;                  /* roptions.c:25 */
    uint32_t bp_per_sec, sample_rate, chunk_size, max_num_chunk; /* roptions.c:9-11, 24 */
    uint32_t slot_events;      /* events a read may reach: its slot in the event arena */
    uint32_t max_reads;        /* reads the mapper holds at a time (rawdtw_mapper_release_read gives a finished read's slot back) */
    int carry;                 /* 1: a round takes the unchanged parts' costs over from the round before (rawdtw_batch_submit_carry) */
    uint32_t min_events;       /* roptions.c:23 (50): a chunk with fewer events is appended to the read's events, but the round leaves
                                  the read's chains and its offset alone (rmap.cpp:569-575) */
    int threads;               /* host threads of a round's per-read work (re-seeding, sort, chaining DP, carry matching, primary
                                  chains): the reference runs kt_for over n_threads reads (rmap.cpp:916).  <= 1: the calling thread */
    int groups;                /* 1 or 2 read groups, each on a context of its own (the second is created by the mapper and shares
                                  the reference arena): one group's host phase runs while the other's batch is on the device, as the
                                  reference's two pipeline workers overlap (rmap.cpp:1015,1033) */
    int device_chain;          /* 1: the anchor sort and the chaining DP of a round run on the device too (rawdtw_chain_round) and hand
                                  their chains to the DTW in device memory; the host phase is then the events and the seed lists.  Costs
                                  are not carried in this mode (nothing of the anchor lists crosses PCIe either way).  A round the
                                  device declines is chained on the host: same lines. */
} rawdtw_mapper_opt_t;
typedef struct {
    uint32_t ref_seq;
    int32_t strand;
    uint32_t target_position;
    uint32_t query_position;   /* inside the chunk */
} rawdtw_seed_hit_t;
/* `ctx` may be NULL for a mapper that scores through rawdtw_mapper_set_scorer only (no device is touched then) */
int rawdtw_mapper_create(rawdtw_ctx *ctx, const rawdtw_mapper_opt_t *opt, uint32_t n_seq, const char *const *seq_names,
                         const uint32_t *seq_len, rawdtw_mapper **out);
int rawdtw_mapper_add_read(rawdtw_mapper *m, const char *name, uint32_t qlen /* samples */, uint32_t n_chunks_available,
                           uint32_t *read_id);
/* a finished read whose line has been written: its slot in the event arena goes to the next rawdtw_mapper_add_read */
int rawdtw_mapper_release_read(rawdtw_mapper *m, uint32_t read_id);
/* one chunk round: read read_ids[k] gets events[event_off[k] .. event_off[k+1]) and hits[hit_off[k] .. hit_off[k+1]).
 * All reads are checked before anything changes; when the device round fails afterwards, the reads are put back as they
 * were (a failed round can be repeated). */
int rawdtw_mapper_round(rawdtw_mapper *m, uint32_t n_reads, const uint32_t *read_ids, const uint64_t *event_off,
                        const float *events, const uint64_t *hit_off, const rawdtw_seed_hit_t *hits);
/* (With device_chain and one read group a page-locked `events` array -- rawdtw_host_alloc -- goes to the device as it is, without a
 * copy into the mapper's own staging.) */
int rawdtw_mapper_read_state(const rawdtw_mapper *m, uint32_t read_id, int *finished, uint32_t *chunks_done);
int rawdtw_mapper_finish(rawdtw_mapper *m);
/* *len = the line's length; RAWDTW_ERR_RANGE when buf (cap bytes) is too small for it and its terminator */
int rawdtw_mapper_paf(const rawdtw_mapper *m, uint32_t read_id, char *buf, uint32_t cap, uint32_t *len);
/* the lines --dtw-log-scores writes to stderr (rmap.cpp:308-312), in order */
int rawdtw_mapper_log(const rawdtw_mapper *m, const char **text);
int rawdtw_mapper_stats(const rawdtw_mapper *m, uint64_t *rounds, uint64_t *parts_scored, uint64_t *parts_reused);
/* where a round's time went, accumulated over the mapper's rounds, in milliseconds of host wall time: 0 the per-read host
 * phase (events, re-seeding, sort, chaining, evaluation order, carry matching), 1 laying the round's arrays out, 2 the
 * submissions (enqueue only), 3 waiting for the device in fetch, 4 the round's end per read (primary chains, MAPQ, stop
 * rule); 5 bytes handed to the device for anchor lists, 6 for events, 7 for everything else (offsets, bases, carry records).
 * With device_chain: 0 = the events and the seed lists (the checks and the round's set-up count here too), 1 = the round's
 * chains per read out of the arrays the device wrote, 2 = the submissions and the wait for the device's sort + DP, 5 = 0 (no
 * anchor list goes up), 7 includes the seed lists. */
int rawdtw_mapper_timing(const rawdtw_mapper *m, double out[8]);
/* Harness hook: score a round's chains with `fn` instead of on the device -- for timing or checking the SAME control flow
 * with another DTW implementation (bench.py's cpu_baseline, the CPU-only tests).  fn receives the round's chains read by
 * read, each read's chains in evaluation order (rmap.cpp:512), and must write score[c] / keep[c] for every chain exactly as
 * the DTW block of gen_chains would (rmap.cpp:515-524); a non-zero return fails the round.  The product's own path never
 * sets one. */
typedef int (*rawdtw_scorer_fn)(void *user, uint64_t n_reads, const uint64_t *chain_off, const uint64_t *anchor_off,
                                const rawdtw_anchor_t *anchors, const uint32_t *chain_seq, const int32_t *chain_strand,
                                const float *const *read_events, const uint32_t *read_n_events, float *score, uint8_t *keep);
int rawdtw_mapper_set_scorer(rawdtw_mapper *m, rawdtw_scorer_fn fn, void *user);
const char *rawdtw_mapper_last_error(const rawdtw_mapper *m);
int rawdtw_mapper_destroy(rawdtw_mapper *m);

#ifdef __cplusplus
}
#endif
#endif /* RAWDTW_H */
