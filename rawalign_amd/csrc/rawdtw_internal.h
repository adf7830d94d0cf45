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
    uint32_t reserved;
};
static_assert(sizeof(ChainDesc) == 24, "ChainDesc must stay 24 bytes");

constexpr int kMaxLaneRadius = 3;      // lane-per-job DP is instantiated for R in [0, 3]: a sweep showed that the rare
                                       // jobs with larger radii (0.4 % of a sparse batch) cost half of the tile kernel's
                                       // time through divergence and register pressure; they go to k_band_wreg<1>
constexpr int kMaxLaneRadiusHi = 8;    // second tile-kernel instance: radii kMaxLaneRadius+1 .. 8
constexpr int kLaneMaxN = 73;          // ... and for jobs whose longer side is at most this
// tile kernel: a tile = consecutive lane-eligible jobs whose windows fit this much LDS
constexpr uint32_t kTileLdsFloats = 4800;  // job-list tile kernel: 8 workgroups per CU
constexpr uint32_t kStreamTileFloats = 7000; // k_stream: 4 workgroups per CU (image + records + sort table = 39 KB: ~7 KB of the
                                             // CU's 160 KB stay free for the planning kernels of the batches behind it)
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

// ---- sync-free candidate batches (rawdtw_stream.hip) ----
// One record per job of the batch, job order: the windows' arena offsets and, for the jobs the lane-per-job DP takes,
// the packed shape: bits 0-6 longer side, 7-13 shorter side, 14-15 radius after the slant correction (dtw.cpp:298-300),
// 16 exclude_last_element, 17 swap (the reference window is the longer one, dtw.cpp:284-292), 18 run start, 19 tile class.
struct JobRec { uint64_t ref_off; uint32_t read_off; uint32_t meta; };
static_assert(sizeof(JobRec) == 16, "JobRec is 16 bytes");
constexpr uint32_t kMetaStarts = 1u << 18, kMetaTile = 1u << 19;
constexpr uint32_t kStreamItems = 4;          // jobs per thread in a tile's prologue: a tile's range holds kStreamItems * threads jobs
constexpr uint32_t kStreamMaxTileJobs = kStreamItems * 512; // ... of 512-thread workgroups (256-thread: half)
constexpr uint32_t kStreamSlack = 32;         // LDS floats of a tile's image kept free for region alignment
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

// Layout of a tile's LDS image.  The image has an event region and a reference region; consecutive parts of a chain
// (a "run") share their anchor elements, so a run is ONE contiguous piece of each arena and of each region.  Every job
// adds `read` / `ref` floats to the regions: a run start its whole window plus 3 floats of slack, a continuing part its
// window minus the shared first element, and a run's last part the padding that rounds the run's END up to a 16-byte
// boundary of the arena.  With c = the running sum BEFORE the job (a global exclusive scan) the job's window starts at
//     start:       c + ((off - c) & 3)            continuing:  c - 4 + ((off - c) & 3)
// (off = the window's arena offset): a closed form of the job's own scan value that is congruent to the arena offset
// modulo 4 -- 16-byte chunks of the image are 16-byte chunks of the arena -- consistent along a run, and such that two
// runs never share a chunk.  So a tile is staged by a flat, fully coalesced copy of chunks, with no per-tile run
// table and no per-tile scan.  `cost` (eighths of a float, with a floor that bounds the jobs of a tile) cuts the batch
// into tiles: tile k = the jobs whose exclusive cost lies in [k w, (k + 1) w).
constexpr uint32_t kMetaEnds = 1u << 20; // the run's last part (the next part of the chain is not a tile job)
struct Cum { uint32_t read, ref; uint64_t cost; };
static_assert(sizeof(Cum) == 16, "Cum is 16 bytes");
__host__ __device__ inline Cum job_cum(const JobRec &r, uint32_t min_cost8)
{
    Cum c{0u, 0u, (uint64_t)min_cost8};
    const uint32_t meta = r.meta;
    if (!(meta & kMetaTile)) return c;
    const uint32_t N = meta & 127u, M = (meta >> 7) & 127u;
    const bool swap = (meta >> 17) & 1u, starts = (meta >> 18) & 1u, ends = (meta >> 20) & 1u;
    const uint32_t n_read = swap ? M : N, n_ref = swap ? N : M;
    c.read = (starts ? n_read + 3u : n_read - 1u) + (ends ? (0u - (r.read_off + n_read)) & 3u : 0u);
    c.ref = (starts ? n_ref + 3u : n_ref - 1u) + (ends ? (0u - ((uint32_t)r.ref_off + n_ref)) & 3u : 0u);
    const uint32_t c8 = 8u * (c.read + c.ref);
    c.cost = c8 > min_cost8 ? c8 : min_cost8;
    return c;
}
__host__ __device__ inline Cum cum_of(uint64_t packed) { return Cum{(uint32_t)packed, (uint32_t)(packed >> 32), 0ull}; }
__host__ __device__ inline uint32_t image_pos(uint32_t c_excl, uint64_t arena_off, bool starts)
{
    return (starts ? c_excl : c_excl - 4u) + (((uint32_t)arena_off - c_excl) & 3u);
}
// One tile: its job range and the geometry of its image (k_tile_first)
struct alignas(16) TileInfo { uint32_t first, n, base_read, base_ref, ref_region, image, first_tile /* index in the range of its first tile-class job */, pad1; };
static_assert(sizeof(TileInfo) == 32, "TileInfo is 32 bytes");

enum StreamCounter : int {
    kCntBad = 0,        // min: first job with invalid anchors or a window outside the arenas (~0 = none)
    kCntOverflow,       // min: a tile over one of the kernel's capacities (~0 = none): never with a correct planner
    kCntUnsupported,    // jobs whose band is wider than the side list's kernels take (radius + 1 > 256)
    kCntTileJobs, kCntTileBytes, kCntOtherBytes, kCntOthers, kCntTiles, kCntLdsMax,
    kCntCls0,           // kStreamClasses totals
    kCntCur0 = kCntCls0 + 21, // kStreamClasses scatter cursors
    kCntCells = kCntCur0 + 21,
    kCntHeads = 64,     // tile queue: 8 heads, one per 128-byte line (head h deals the tiles t with t % 8 == h)
    kStreamCounters = kCntHeads + 8 * 16
};
static_assert(kCntCells < kCntHeads && kCntCur0 == kCntCls0 + kStreamClasses, "counter layout");
struct StreamArgs {
    uint64_t n_jobs, n_chains, n_reads, n_ev, n_ref, others_cap;
    float frac;                  // dtw_band_radius_frac
    int32_t lane_max_radius;     // tile class: radius <= this and longer side <= lane_max_n
    int32_t side_lane_radius;    // side-list lane classes: radius in (lane_max_radius, this], longer side <= lane_max_n
    uint32_t lane_max_n;
    uint32_t min_cost8;          // cost floor in eighths of a float: bounds the jobs of a tile's range
    uint64_t width8;             // bracket width of the tile rule: 8 * (image floats - slack) - the largest cost a job can have
    uint32_t tiles_cap;
    uint32_t debug;              // timing experiments only (results wrong below 128 except 8): 1 no DP, 2 no staging, 4 no side
                                 // list, 16 no marks, 32 no wave-per-job items, 64 no group / lane items; 8 tiles without the
                                 // ticket queue, 2048 side items over all waves, 4096 side items dealt straight (not alternating)
    // inputs (device)
    const uint32_t *unit_chain;  // per k_pre unit of 1024 jobs: the chain its first job belongs to (n_units + 1 entries)
    const uint64_t *job_off, *anchor_off;
    const rawdtw_anchor_t *anchors;
    const uint64_t *ref_base;
    const uint32_t *read_base;
    const float *ev, *ref;
    // planning arrays and outputs (device)
    JobRec *jrec;
    uint64_t *cpos;              // inclusive sums of job_cum INSIDE the job's unit of 1024: event floats | reference floats << 32
    uint32_t *ccost;             // ... and cost
    uint64_t *unit_pos, *unit_cost; // the sums before each unit (n_units + 1 entries)
    TileInfo *tiles;             // tiles_cap + 1 entries; tiles[n_tiles].n = 0
    unsigned long long *unit_stats; // per k_pre unit: tile jobs, tile bytes, side-list bytes
    DevJob *omix, *ojobs;
    uint8_t *ocls;
    unsigned long long *cnt;
    float *out;
};
int stream_blocks_per_cu(uint32_t lds_floats, int threads);
hipError_t stream_plan(const StreamArgs &a, ChainDesc *d_chains, uint32_t *d_fold_order, hipStream_t s);
hipError_t stream_run(const StreamArgs &a, uint32_t blocks, uint32_t lds_floats, int threads, bool reset_queue, hipStream_t s);
hipError_t stream_count_cells(const StreamArgs &a, unsigned long long *d_total, hipStream_t s);
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
                               uint32_t *tmp_i, uint32_t *tmp_j, uint32_t *path_i, uint32_t *path_j, float *path_d,
                               hipStream_t s);

hipError_t launch_chain_fold(int mode, const ChainDesc *chains, const uint32_t *order, uint64_t n_chains,
                             const float *job_cost, float bonus, int fused, float *full_score, float *att_last, uint32_t long_parts,
                             hipStream_t s);
hipError_t launch_read_select(const uint64_t *chain_off, uint64_t n_reads, const float *full_score,
                              const float *att_last, float min_score, float *score, uint8_t *keep,
                              hipStream_t s);

// geometry helpers shared with the planner
inline uint32_t full_strip_rows(int rpl) { return 64u * (uint32_t)rpl; }

} // namespace rawdtw
