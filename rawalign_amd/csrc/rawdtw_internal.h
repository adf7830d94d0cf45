// rawdtw_internal.h -- types shared by the HIP kernels and the host-side planner.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

#include "../../include/rawdtw.h"

namespace rawdtw {

constexpr float kInf = 1e10f; // dtw.cpp:38,310-313: the float nearest 1e10

// Device job record, 32 bytes, in PLAN order.  Written by the planner.
struct DevJob {
    uint64_t ref_off;  // element offset of b[0] in the reference arena
    uint32_t read_off; // element offset of a[0] in the event arena
    uint32_t n;        // a_length
    uint32_t m;        // b_length
    int32_t R;         // banded: radius AFTER the slant correction (dtw.cpp:298-300); full: -1
    uint32_t flags;    // bit0: exclude_last_element
    uint32_t aux;      // the job's index in the caller's batch: kernels store their cost at out[aux]
};
static_assert(sizeof(DevJob) == 32, "DevJob must stay 32 bytes");

enum : uint32_t { kFlagExcludeLast = 1u };

// launch kinds (also reported by rawdtw_plan_run_timed)
enum LaunchKind : uint32_t {
    kKindBandLane = 1,  // lane-per-job banded DP inside the tile kernel (one launch for all eligible jobs)
    kKindBandWave = 2,  // wave-per-job banded kernel
    kKindFullWave = 3,  // wave-per-job full-matrix wavefront (score only)
    kKindFullTb = 4,    // same, writing packed directions
    kKindTbWalk = 5,    // traceback walk
    kKindChainFold = 6, // per-chain fold of the part costs (align_chain)
    kKindReadSelect = 7, // per-read accept/cut loop (gen_chains DTW block)
    kKindBandWreg = 8,   // wave-per-job banded kernel, band in registers (param = registers per lane per buffer)
    kKindBandLaneHi = 9, // tile kernel instance for the wider bands (radius 4..8)
    kKindBandMerged = 10 // reporting only: the tile launch when it also carries the two small banded classes (k_band_merged)
};

// Tile kernel records (planner output)
struct TileDesc { uint32_t job_first, n_jobs, span_first, n_spans; }; // n_spans bit 31: many short spans (a by-shape tile)
struct TileSpan {
    uint64_t src;          // float index of the span's first (16-byte aligned) element in its arena
    uint32_t lds_off;      // float offset in the tile's LDS image (multiple of 4)
    uint32_t chunks_arena; // 16-byte chunks to copy | (1u<<31 when the span lies in the reference arena)
};
struct TileJob {
    uint16_t offA, offB;   // LDS float offsets of the longer / shorter window
    uint8_t N, M, R, flags; // longer side, shorter side, dispatch kind (0 micro4, 1 micro8, 2+radius lane DP), kFlagExcludeLast
    uint32_t aux;          // job index in the caller's batch
    uint32_t pad;          // micro kinds: index of the shape's band mask
};
static_assert(sizeof(TileDesc) == 16 && sizeof(TileSpan) == 16 && sizeof(TileJob) == 16, "tile records are 16 bytes");

// One candidate chain as the replay kernels see it (24 bytes).
struct ChainDesc {
    uint64_t job_first;   // first job of the chain in the batch's job order
    uint32_t n_jobs;      // 1 (global) or n_anchors-1 (sparse)
    uint32_t span;        // read events between the chain's first and last anchor, inclusive (rmap.cpp:245)
    uint32_t num_aligned; // sum of the parts' read_region_size (rmap.cpp:236,292)
    uint32_t descending;  // 1: the chain's part p is job_cost[job_first - p] (sync-free batches: costs are stored per anchor)
};
static_assert(sizeof(ChainDesc) == 24, "ChainDesc must stay 24 bytes");

constexpr int kMaxLaneRadius = 3;      // lane-per-job DP is instantiated for R in [0, 3]: a sweep showed that the rare
                                       // jobs with larger radii (0.4 % of a sparse batch) cost half of the tile kernel's
                                       // time through divergence and register pressure; they go to k_band_wreg<1>
constexpr int kMaxLaneRadiusHi = 8;    // second tile-kernel instance: radii kMaxLaneRadius+1 .. 8
constexpr int kLaneMaxN = 73;          // ... and for jobs whose longer side is at most this
// tile kernel: a tile = consecutive lane-eligible jobs whose windows fit this much LDS
constexpr uint32_t kTileLdsFloats = 4800;  // job-list tile kernel: 8 workgroups per CU
constexpr uint32_t kStreamTileFloats = 5800; // k_runs: image + the pass's records + two passes' copy orders = 29 KB a workgroup: five workgroups per
                                             // CU, and 13 KB of the CU's 160 KB stay free for a planner workgroup of the batches behind (at 7000 floats
                                             // and two record buffers it was four workgroups and 5 KB: nothing fitted beside them; the bench batch's
                                             // tiles average 5 200 floats: below 5 400 every other tile takes two passes and the planning doubles)
constexpr uint32_t kTileMaxJobs = 1024;
constexpr uint32_t kTileHiLdsFloats = 14336, kTileHiMaxJobs = 64; // wide-band instance: one wave per tile
constexpr uint32_t kTileMaxSpans = 96;
constexpr uint32_t kSpanGapFloats = 48; // a window starting this close behind a span extends it instead of opening a new one
constexpr int kFullWgWaves = 4;        // waves per job in the pipelined-strip variant of the full-matrix kernel
constexpr int kMaxWregChunks = 32;     // register-resident wave kernel: radius + 1 <= 64 * 32
constexpr int kMaxWaveBandK = 13000;   // 3*K floats of LDS must fit 160 KiB

struct Launch {
    uint32_t kind;
    int32_t param;      // band tile: LDS floats ; wreg: registers per lane ; full: rows per lane ; band wave: LDS floats
    uint64_t first;     // first plan-order job
    uint64_t count;     // jobs in this launch
};

// per-job workspace for the full-matrix kernels
struct FullAux {
    uint64_t bnd_off; // float offset of the strip-boundary row in the workspace (multi-strip jobs)
    uint64_t dir_off; // byte offset of the packed direction buffer (traceback jobs)
};

// ---- sync-free candidate batches (rawdtw_runs.hip) ----
// A candidate batch is scored straight from the caller's anchor lists: nothing is written per job but its cost.
//
//   tile      a fixed range of kTileAnchors consecutive entries of anchors[]; its DTW jobs are the parts that END at one
//             of its anchors (part i runs from anchors[i + 1] to anchors[i]: rmap.cpp:251-254 stores chains end-first).
//             Everything about a tile -- its chains, the parts' windows, radii and classes, the layout of its LDS
//             image -- is derived inside the DTW launch from those anchors and the chains' offsets.
//   out[i]    the cost of part i (one float per anchor; the slot of a chain's first entry stays unused).  A chain's
//             parts p = 0 .. n - 2 in the reference's order are out[a1 - 2 - p]: the fold walks it downwards.
//   side list parts the lane-per-job bodies do not take (radius > lane_max_radius or longer side > lane_max_n), found
//             by k_scan ahead of the DTW launch and scored there first, wave-cooperatively, longest first.
constexpr uint32_t kStreamTile = 512;         // anchors (= candidate parts) of a tile: what one wave of the scan plans (eight a lane)
constexpr uint32_t kStreamRecStride = 576;    // job records of a tile: its passes' records one behind the other, each pass on a 16-byte boundary
constexpr uint32_t kStreamMaxSeg = 32;        // runs of one pass over a tile's image (more: the tile takes another pass)
// side-list classes, in launch order: wave-per-job by longer side (>= 1024, >= 256, >= 64, shorter), 16-lane groups, 8-lane groups
constexpr uint32_t kStreamClasses = 21, kClsW0 = 0, kClsG16 = 4, kClsL0 = 5, kClsLCount = 8, kClsM0 = 13, kClsMCount = 8;
// Side-list classes 13..20: bands of up to 8 slots (radius <= 7) that are not in the classes below: one lane per job too
// (lane_dp_k8), same length buckets -- as eight-lane groups (grp_wave<8>) they were a third of the DTW launch's arithmetic
// for 0.4 % of its jobs.
// Side-list classes 5..12: jobs the lane-per-job body takes (radius <= side_lane_radius, longer side <= lane_max_n) but the
// tiles do not (radius > lane_max_radius): scored 64 to a wave straight from the arenas, bucketed by longer side so that
// a wave's jobs have similar lengths -- over the whole batch there are enough of them to fill waves, inside one tile not.
__host__ __device__ inline uint32_t side_lane_bucket(uint32_t N)
{
    return N >= 65u ? 0u : N >= 49u ? 1u : N >= 41u ? 2u : N >= 33u ? 3u : N >= 29u ? 4u : N >= 25u ? 5u : N >= 21u ? 6u : 7u;
}

enum StreamCounter : int {
    kCntBad = 0,        // min: first anchor index whose part is invalid (anchors not ascending, window outside an arena); ~0 = none
    kCntOverflow,       // min: a tile over one of the kernel's capacities (~0 = none): never with a correct kernel
    kCntUnsupported,    // parts whose band is wider than the side list's kernels take (radius + 1 > 256), chains without anchors
    kCntOthers,         // side-list entries appended
    kCntCls0,           // kStreamClasses totals
    kCntCur0 = kCntCls0 + 21, // kStreamClasses scatter cursors
    kCntCells = kCntCur0 + 21,
    kCntTileJobs, kCntTileBytes, kCntOtherBytes, // (reporting: summed from the tiles' statistics on request)
    kCntTodo,           // entries of the DTW launch's work list
    kCntReused,         // (unused: the parts taken over are counted from the carry records on the host -- a counter every chain's wave adds to
                        // is forty thousand same-address atomics, 0.3 ms)
    kCntPool,           // record slots handed out beyond one a tile (tiles whose image takes several passes)
    kCntStamp0,         // 10 words: cycles per phase of k_runs, summed over waves ("stream_debug" 256: diagnostic runs only)
    kCntHeads = 64,     // tile queue: 8 heads, one per 128-byte line (head h deals the tiles t with t % 8 == h)
    kStreamCounters = kCntHeads + 8 * 16
};
static_assert(kCntStamp0 + 10 <= kCntHeads && kCntCur0 == kCntCls0 + kStreamClasses, "counter layout");
struct StreamArgs {
    uint64_t n_anchors, n_chains, n_reads, n_ev, n_ref, others_cap;
    float frac;                  // dtw_band_radius_frac
    int32_t lane_max_radius;     // tile class: radius <= this and longer side <= lane_max_n
    int32_t side_lane_radius;    // side-list lane classes: radius in (lane_max_radius, this], longer side <= lane_max_n
    uint32_t lane_max_n;
    uint32_t n_tiles;            // ceil(n_anchors / tile_anchors)
    uint32_t tile_anchors;       // kStreamTile: the anchors one wave of the scan plans together
    uint32_t n_slots;            // copy-order slots (= the most passes a batch can have): one a tile, then a pool for the further passes
                                 // of tiles that take several
    uint32_t lds_floats;         // the DTW launch's image budget (floats): a pass's windows fit it
    uint32_t debug;              // timing experiments only (results wrong below 128 except 8; != 0 selects k_runs' diagnostic
                                 // instance): k_runs 1 no DP, 2 no staging, 8 passes dealt by block index, 256 phase stamps, 512 a pass without its
                                 // first chunk of 64 jobs, 1024 a pass's first chunk only;
                                 // k_wide 32 no wave-per-job items, 64 no group / lane items, 2048 items over all waves, 4096
                                 // items dealt straight (not alternating); 4 no k_wide launch at all
    // inputs (device)
    const uint64_t *anchor_off;
    const rawdtw_anchor_t *anchors;
    const uint64_t *ref_base;
    const uint32_t *read_base;
    const float *ev, *ref;
    // compact hand-over (rawdtw_batch_submit_compact; `steps` null otherwise): k_scan decodes the lists into `anchors_w`
    // (= `anchors`, writable) unit by unit before it looks at them
    const rawdtw_anchor_t *heads, *unit_abs;
    const uint16_t *steps;
    const rawdtw_wide_step_t *wide;
    uint64_t n_wide;
    rawdtw_anchor_t *anchors_w;
    // chunk rounds (rawdtw_batch_submit_carry; `carry` null otherwise).  The lists above (anchor_off, anchors, n_anchors) are then the
    // round's SHORT lists: per chain its new entries and the junction (the end anchor of the stretch taken over) -- everything from
    // the scan to the DTW launches runs on them and leaves the new parts' costs in `out`, one per short-list anchor.  The fold
    // wants every chain's costs in full: k_gather lays them out in `out_full` by the FULL lists' offsets `full_off` -- a chain's
    // new parts from `out`, the stretch taken over from the previous batch's full cost array, one contiguous copy a chain (the
    // host validated it anchor by anchor: rawdtw_round_match_chains).  A batch without a predecessor has out_full = out and
    // full_off = anchor_off.
    const rawdtw_carry_t *carry;      // per chain: {first entry of the stretch in the previous full list, parts taken over, flags, start anchor}
    const uint64_t *full_off;         // n_chains + 1 offsets of the full lists
    uint64_t n_full;                  // full_off[n_chains]
    float *out_full;                  // n_full costs: part p of chain c (rmap.cpp:248-293) at full_off[c + 1] - 2 - p
    const float *prev_out_full;       // the previous batch's
    uint64_t prev_n_full;
    const unsigned long long *prev_cnt; // its counter block (a batch the scan declined has no costs to take over)
    uint64_t prev_others_cap;
    // workspace and outputs (device)
    uint2 *tlist;                // the scan's tile list: (tile, the chain its first anchor belongs to) of every tile that has a part for the
                                 // lane bodies; cnt[kCntTodo] entries, at most n_tiles
    uint4 *todo;                 // the DTW launch's work list (k_plan): one entry a PASS -- a tile's tile-class parts, or as many of them
                                 // as fit the image budget and the run table: (tile, copy-order slot, jobs | runs << 16, floats of
                                 // the image's event region | first record << 16), at the index of its slot: [0, cnt[kCntTodo]) the listed
                                 // tiles' first passes, [n_tiles, n_tiles + cnt[kCntPool]) the others
    uint2 *recs;                 // n_tiles x kStreamRecStride job records, a pass's in the order the lanes take them (radius class, then longer
                                 // side, descending): x = event window | reference window << 16 (float offsets into the pass's
                                 // image, longer sequence first), y = N | M << 7 | R << 14 | exclude_last << 16 | item << 17
                                 // (item u = the part that ends at anchor (tile end - 1 - u): its cost goes to out[that anchor])
    uint4 *runtab;               // n_slots x 2 kStreamMaxSeg copy orders of 16-byte pieces, entry 2 g + w = run g of arena w (0 events,
                                 // 1 reference): pieces [x, y) of the image come from arena float index (4 piece + (int64)(z | w << 32))
    unsigned long long *tile_stats; // per scan unit (8192 anchors): tile-class parts, their algorithmic bytes, the side list's bytes
    DevJob *omix, *ojobs;
    uint8_t *ocls;
    unsigned long long *cnt;
    float *out;                  // n_anchors entries
};
int stream_blocks_per_cu(uint32_t lds_floats, int threads);
hipError_t stream_plan(const StreamArgs &a, ChainDesc *d_chains, uint32_t *d_fold_order, hipStream_t s);
hipError_t stream_gather(const StreamArgs &a, ChainDesc *d_chains, hipStream_t s);
hipError_t stream_run(const StreamArgs &a, uint32_t blocks, uint32_t lds_floats, int threads, bool reset_queue, hipStream_t s);
hipError_t stream_plan_passes(const StreamArgs &a, hipStream_t s);
hipError_t stream_wide(const StreamArgs &a, uint32_t blocks, hipStream_t s);
hipError_t stream_fold_select(const StreamArgs &a, const ChainDesc *chains, const uint64_t *chain_off, uint64_t n_reads, float bonus, int fused,
                              float min_score, float *full_score, float *att_last, float *score, uint8_t *keep, hipStream_t s);
hipError_t stream_count_cells(const StreamArgs &a, unsigned long long *d_total, hipStream_t s);
hipError_t stream_sum_stats(const StreamArgs &a, hipStream_t s);
hipError_t launch_events_scatter(const float *d_src, float *d_dst, const uint64_t *d_seg_src, const uint32_t *d_seg_dst,
                                 uint32_t n_seg, hipStream_t s);

hipError_t launch_band_merged(const TileDesc *tiles, uint64_t n_tiles, const TileSpan *spans, const TileJob *tjobs,
                              const unsigned long long *masks, uint32_t lds_floats, const DevJob *wjobs, uint64_t n_w,
                              const DevJob *gjobs, uint64_t n_g, const DevJob *hjobs, uint64_t n_h, const float *ev,
                              const float *ref, float *out, hipStream_t s);
hipError_t launch_band_tile(bool hi, int threads, const TileDesc *tiles, uint64_t n_tiles, const TileSpan *spans, const TileJob *tjobs,
                            const unsigned long long *masks, uint32_t lds_floats, const float *ev, const float *ref,
                            float *out, hipStream_t s);
hipError_t launch_band_wreg(int chunks, const DevJob *jobs, uint64_t count, const float *ev,
                            const float *ref, float *out, hipStream_t s);
hipError_t launch_band_wave(const DevJob *jobs, uint64_t count, uint32_t lds_floats,
                            const float *ev, const float *ref, float *out, hipStream_t s);
hipError_t launch_full_wave(int rows_per_lane, bool traceback, const DevJob *jobs, uint64_t count,
                            const FullAux *aux, const float *ev, const float *ref, float *out,
                            float *bnd_ws, uint8_t *dir_ws, hipStream_t s);
hipError_t launch_tb_walk_wave(const DevJob *jobs, uint64_t count, const FullAux *aux, int rpl, const float *ev,
                               const float *ref, const uint8_t *dir_ws, const uint64_t *path_off, uint32_t *path_len,
                               uint32_t *tmp_i, uint32_t *tmp_j, uint8_t *path_mv, float *path_d, hipStream_t s);

hipError_t launch_chain_fold(int mode, const ChainDesc *chains, const uint32_t *order, uint64_t n_chains,
                             const float *job_cost, float bonus, int fused, float *full_score, float *att_last, uint32_t long_parts,
                             hipStream_t s);
hipError_t launch_read_select(const uint64_t *chain_off, uint64_t n_reads, const float *full_score,
                              const float *att_last, float min_score, float *score, uint8_t *keep,
                              hipStream_t s);

// geometry helpers shared with the planner
inline uint32_t full_strip_rows(int rpl) { return 64u * (uint32_t)rpl; }

} // namespace rawdtw
