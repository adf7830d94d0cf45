// rawdtw_host.cpp -- host-side mirror of align_chain (src/rmap.cpp:181-313) and of the DTW block
// of gen_chains (src/rmap.cpp:509-530), split in two so that the DTW calls in the middle can be
// executed as one GPU batch:
//
//   rawdtw_chain_build_jobs : the decomposition of a chain into DTW sub-problems
//   rawdtw_chain_replay     : the fold of the per-job costs, including the early exits that
//                             depend on the running best score (rmap.cpp:205-209, 265-268)
//   rawdtw_read_replay      : the sequential best-so-far loop over one read's chains
//
// The GPU evaluates every job of every chain speculatively; the replay then reproduces exactly
// which chains the reference would have cut, so the filtered chain set (and everything
// downstream: primary chains, MAPQ, the stop rule) is unchanged.  No device code here.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/rawdtw.h"

namespace {

inline int band_radius_for(uint32_t read_region_size, float frac)
{
    // rmap.cpp:214,276: std::max(1, (int)(read_region_size * opt->dtw_band_radius_frac)), fp32 product
    int r = (int)((float)read_region_size * frac);
    return r > 1 ? r : 1;
}

inline float final_score(uint32_t num_aligned, const rawdtw_align_opt_t *opt, float cost)
{
    // rmap.cpp:306.  With the reference's flags (-O3 -march=native, GCC's default
    // -ffp-contract=fast) this statement is a single fused multiply-subtract on FMA hosts.
    if (opt->fused_score) return std::fmaf((float)num_aligned, opt->match_bonus, -cost);
    volatile float prod = (float)num_aligned * opt->match_bonus; // keep the product rounded
    return prod - cost;
}

} // namespace

extern "C" {

uint32_t rawdtw_chain_job_count(const rawdtw_align_opt_t *opt, uint32_t n_anchors)
{
    if (!opt || n_anchors == 0) return 0;
    return opt->border_constraint == 0 ? 1u : n_anchors - 1;
}

int rawdtw_chain_build_jobs(const rawdtw_align_opt_t *opt, const rawdtw_anchor_t *anchors, uint32_t n_anchors,
                            uint64_t ref_base, uint32_t read_base, int cigar, rawdtw_job_t *jobs_out)
{
    if (!opt || !anchors || n_anchors == 0 || !jobs_out) return RAWDTW_ERR_INVALID;
    if (opt->border_constraint != 0 && opt->border_constraint != 1)
        return RAWDTW_ERR_INVALID; // rmap.cpp:301-304: "invalid border constraint" -> exit
    const bool banded = opt->fill_method != 0;
    if (opt->border_constraint == 0) {
        const rawdtw_anchor_t &s = anchors[n_anchors - 1]; // chain start (rmap.cpp:195)
        const rawdtw_anchor_t &e = anchors[0];             // chain end   (rmap.cpp:196)
        rawdtw_job_t &j = jobs_out[0];
        j.ref_off = ref_base + s.target_position;
        j.read_off = read_base + s.query_position;
        j.m = e.target_position - s.target_position + 1;
        j.n = e.query_position - s.query_position + 1;
        j.exclude_last = 0;
        j.reserved = 0;
        if (cigar) {
            if (banded) return RAWDTW_ERR_UNSUPPORTED; // rmap.cpp:223-225 assert(false)
            j.band_radius = RAWDTW_FULL;
        } else {
            j.band_radius = banded ? band_radius_for(j.n, opt->band_radius_frac) : RAWDTW_FULL;
        }
        return RAWDTW_OK;
    }
    const uint32_t parts = n_anchors - 1;
    for (uint32_t p = 0; p < parts; p++) {
        const rawdtw_anchor_t &s = anchors[parts - p];     // rmap.cpp:253
        const rawdtw_anchor_t &e = anchors[parts - p - 1]; // rmap.cpp:254
        rawdtw_job_t &j = jobs_out[p];
        j.ref_off = ref_base + s.target_position;
        j.read_off = read_base + s.query_position;
        j.m = e.target_position - s.target_position + 1;
        j.n = e.query_position - s.query_position + 1;
        j.reserved = 0;
        if (cigar) {
            // rmap.cpp:283-284: always DTW_global_tb, exclude_last_element never passed
            j.band_radius = RAWDTW_FULL;
            j.exclude_last = 0;
        } else {
            j.band_radius = banded ? band_radius_for(j.n, opt->band_radius_frac) : RAWDTW_FULL;
            j.exclude_last = (p != parts - 1) ? 1u : 0u; // rmap.cpp:270
        }
    }
    return RAWDTW_OK;
}

float rawdtw_chain_replay(const rawdtw_align_opt_t *opt, const rawdtw_anchor_t *anchors, uint32_t n_anchors,
                          const float *job_cost, float min_score)
{
    float cost = 0.0f;
    uint32_t num_aligned = 0;
    const rawdtw_anchor_t &first = anchors[n_anchors - 1];
    const rawdtw_anchor_t &last = anchors[0];
    if (opt->border_constraint == 0) {
        const uint32_t rn = last.query_position - first.query_position + 1;
        const float attainable = (float)rn * opt->match_bonus; // rmap.cpp:205
        if (attainable < min_score) return -1e10f;             // rmap.cpp:206-209
        cost = job_cost[0];
        num_aligned = rn;
    } else {
        const uint32_t parts = n_anchors - 1;
        const uint32_t span = last.query_position - first.query_position + 1;
        float attainable = (float)span * opt->match_bonus; // rmap.cpp:246
        for (uint32_t p = 0; p < parts; p++) {
            const rawdtw_anchor_t &s = anchors[parts - p];
            const rawdtw_anchor_t &e = anchors[parts - p - 1];
            if (attainable < min_score) return -1e10f; // rmap.cpp:265-268
            const float sub = job_cost[p];
            cost += sub;       // rmap.cpp:279 (fp32, in part order)
            attainable -= sub; // rmap.cpp:280
            num_aligned += e.query_position - s.query_position + 1; // rmap.cpp:292
        }
    }
    return final_score(num_aligned, opt, cost);
}

uint32_t rawdtw_read_replay(const rawdtw_align_opt_t *opt, uint32_t n_chains, const uint32_t *anchor_off,
                            const rawdtw_anchor_t *anchors, const uint64_t *job_off, const float *job_cost,
                            float *score, uint8_t *keep)
{
    float best = 0.0f; // rmap.cpp:515
    uint32_t kept = 0;
    for (uint32_t c = 0; c < n_chains; c++) {
        const uint32_t na = anchor_off[c + 1] - anchor_off[c];
        const float s = rawdtw_chain_replay(opt, anchors + anchor_off[c], na, job_cost + job_off[c], best);
        score[c] = s;
        keep[c] = 0;
        if (s >= opt->min_score) {  // rmap.cpp:518
            if (s > best) best = s; // rmap.cpp:519-521
            keep[c] = 1;
            kept++;
        }
    }
    return kept;
}

uint32_t rawdtw_gen_primary_chains(rawdtw_chain_t *chains, uint32_t n_chains, const rawdtw_select_opt_t *opt,
                                   uint32_t *kept)
{
    if (!chains || !opt || !kept || n_chains == 0) return 0;
    // rmap.h:41-45: operator> compares the tuple (alignment_score, chaining_score, n_anchors, strand,
    // reference_sequence_index, start_position, end_position); rmap.cpp:91 sorts descending with it
    std::sort(chains, chains + n_chains, [](const rawdtw_chain_t &a, const rawdtw_chain_t &b) {
        return std::tie(a.alignment_score, a.chaining_score, a.n_anchors, a.strand, a.reference_sequence_index,
                        a.start_position, a.end_position) >
               std::tie(b.alignment_score, b.chaining_score, b.n_anchors, b.strand, b.reference_sequence_index,
                        b.start_position, b.end_position);
    });
    uint32_t nk = 0;
    kept[nk++] = 0; // rmap.cpp:94
    for (uint32_t ci = 1; ci < n_chains; ++ci) {
        const rawdtw_chain_t &back = chains[kept[nk - 1]];
        if (opt->evaluate_chains) { // rmap.cpp:100-104: below a third of the last primary's score: stop
            if (chains[ci].alignment_score < back.alignment_score / 3) break;
        } else {
            if (chains[ci].chaining_score < back.chaining_score / 3) break;
        }
        bool is_primary = true; // rmap.cpp:113-120: overlap with an earlier primary on the same sequence
        for (uint32_t pi = 0; pi < nk; ++pi) {
            const rawdtw_chain_t &p = chains[kept[pi]];
            if (chains[ci].reference_sequence_index == p.reference_sequence_index &&
                std::max(chains[ci].start_position, p.start_position) <= std::min(chains[ci].end_position, p.end_position)) {
                is_primary = false;
                break;
            }
        }
        if (is_primary) kept[nk++] = ci;
    }
    // comp_mapq, rmap.cpp:65-88
    if (nk == 1) chains[kept[0]].mapq = 60;
    else {
        const rawdtw_chain_t &c0 = chains[kept[0]], &c1 = chains[kept[1]];
        int mapq;
        if (opt->evaluate_chains) mapq = (int)(40 * (1 - c1.alignment_score / c0.alignment_score));
        else mapq = (int)(40 * (1 - c1.chaining_score / c0.chaining_score));
        if (mapq > 60) mapq = 60;
        if (mapq < 0) mapq = 0;
        chains[kept[0]].mapq = (uint32_t)(uint8_t)mapq;
    }
    return nk;
}

int rawdtw_is_mapped_with_high_confidence(const rawdtw_chain_t *c, uint32_t n_chains, const rawdtw_select_opt_t *opt)
{
    const uint32_t n_anchors0 = n_chains ? c[0].n_anchors : 0; // rmap.cpp:597-598
    if (n_anchors0 == 0) return 0;
    auto score = [&](uint32_t k) { return opt->evaluate_chains ? c[k].alignment_score : c[k].chaining_score; };
    if (n_chains >= 2) {
        if (score(0) / score(1) >= opt->min_bestmap_ratio) return 1; // rmap.cpp:604, 651
        float mean = 0;
        for (uint32_t k = 0; k < n_chains; ++k) mean += score(k);
        mean /= n_chains;
        if (score(0) >= opt->min_meanmap_ratio * mean) return 1; // rmap.cpp:615, 658
        return 0;
    }
    return (n_chains == 1 && c[0].n_anchors >= opt->min_chain_anchor) ? 1 : 0; // rmap.cpp:620, 659
}

// sequence_until.c:5-19.  The source is one rounded product and one rounded add per element, summed in order;
// `contracted_tail` selects what the reference's own default build (GCC -O3 with FMA available) computes instead:
// the same in-order sum, but the elements after the last full group of four go through one fused multiply-add each
// (the vectorised groups keep mul + add; only the scalar remainder loop is contracted).
static float find_outlier_impl(const float *const *x, uint32_t n, uint32_t m, bool contracted_tail)
{
    uint32_t outlier = 0;
    float max_dist = 0.0f;
    const uint32_t n_plain = contracted_tail ? (n & ~3u) : n;
    for (uint32_t i = 0; i < m; i++) {
        float dist = 0.0f;
        for (uint32_t j = 0; j < n_plain; j++) {
            const float d = x[i][j] - x[outlier][j];
            volatile float sq = d * d; // product rounded, then added
            dist += sq;
        }
        for (uint32_t j = n_plain; j < n; j++) {
            const float d = x[i][j] - x[outlier][j];
            dist = std::fmaf(d, d, dist);
        }
        if (dist > max_dist) { max_dist = dist; outlier = i; }
    }
    return max_dist;
}

float rawdtw_find_outlier(const float *const *x, uint32_t n, uint32_t m) { return find_outlier_impl(x, n, m, false); }

float rawdtw_find_outlier_contracted(const float *const *x, uint32_t n, uint32_t m)
{
    return find_outlier_impl(x, n, m, true);
}

int rawdtw_chain_anchors(const rawdtw_chain_opt_t *opt, const rawdtw_anchor_t *anchors, uint32_t n_anchors,
                         float *max_chaining_score, rawdtw_chain_out_t *out_chains, uint64_t *out_off,
                         rawdtw_anchor_t *out_anchors, uint32_t chains_cap, uint64_t anchors_cap)
{
    if (!opt || (!anchors && n_anchors) || !max_chaining_score || !out_chains || !out_off || !out_anchors) return -1;
    std::vector<float> score(n_anchors);
    std::vector<size_t> pred(n_anchors);
    std::vector<char> used(n_anchors, 0);
    std::vector<std::pair<float, size_t>> ends;
    float maxs = *max_chaining_score;
    for (size_t ai = 0; ai < n_anchors; ++ai) {
        score[ai] = (float)opt->e; // rmap.cpp:444-445 (distance coefficient 1)
        pred[ai] = ai;
        const int32_t ct = (int32_t)anchors[ai].target_position, cq = (int32_t)anchors[ai].query_position;
        int32_t start = 0;
        if (ai > (size_t)opt->chaining_band_length) start = (int32_t)ai - opt->chaining_band_length;
        int32_t skips = 0;
        for (int32_t pi = (int32_t)ai - 1; pi >= start; --pi) {
            const int32_t pt = (int32_t)anchors[pi].target_position, pq = (int32_t)anchors[pi].query_position;
            if (pq == cq) continue;                                   // rmap.cpp:458
            if (pt == ct) continue;                                   // rmap.cpp:459
            if (pt + opt->max_target_gap_length < ct) break;          // rmap.cpp:460
            const int32_t td = ct - pt, qd = cq - pq;
            float cur = 0;
            if (qd < 0) continue;                                     // rmap.cpp:467
            const float matching = (float)std::min(std::min(td, qd), opt->e); // rmap.cpp:469
            const int gap = std::abs(td - qd);
            const float scale = td > 0 ? (float)qd / td : 1;
            if (gap < opt->max_gap_length && scale < 5 && scale > 0.75) cur = score[pi] + matching; // rmap.cpp:474-476
            if (cur > score[ai]) { score[ai] = cur; pred[ai] = (size_t)pi; --skips; }
            else { ++skips; if (skips > opt->max_num_skips) break; }  // rmap.cpp:478-484
        }
        if (score[ai] > maxs) maxs = score[ai];                      // rmap.cpp:486-488
        if (opt->disable_score_filtering || (score[ai] >= opt->min_chaining_score && score[ai] > maxs / 2))
            ends.emplace_back(score[ai], ai);                        // rmap.cpp:489-493
    }
    *max_chaining_score = maxs;
    // rmap.cpp:175-179 `compare`: score descending, then anchor index descending
    std::sort(ends.begin(), ends.end(), [](const std::pair<float, size_t> &l, const std::pair<float, size_t> &r) {
        if (l.first > r.first) return true;
        if (l.first == r.first) return l.second > r.second;
        return false;
    });
    uint32_t nc = 0;
    uint64_t na = 0;
    out_off[0] = 0;
    for (size_t k = 0; k < ends.size() && k < (size_t)opt->num_best_chains; ++k) {
        const size_t end_idx = ends[k].second;
        if (!used[end_idx]) { // traceback_chains, rmap.cpp:130-173
            std::vector<rawdtw_anchor_t> chain;
            bool stop_at_used = false;
            size_t cur = end_idx;
            chain.push_back(anchors[cur]);
            if (pred[cur] != cur && used[pred[cur]]) stop_at_used = true;
            used[cur] = 1;
            while (pred[cur] != cur && !used[pred[cur]]) {
                cur = pred[cur];
                chain.push_back(anchors[cur]);
                if (pred[cur] != cur && used[pred[cur]]) stop_at_used = true;
                used[cur] = 1;
            }
            if (chain.size() >= (size_t)opt->min_num_anchors) {
                float adj = score[end_idx];
                if (stop_at_used) adj -= score[pred[cur]];
                if (nc >= chains_cap || na + chain.size() > anchors_cap) return -2;
                out_chains[nc] = rawdtw_chain_out_t{adj, anchors[cur].target_position, anchors[end_idx].target_position,
                                                    (uint32_t)chain.size()};
                for (size_t q = 0; q < chain.size(); ++q) out_anchors[na + q] = chain[q]; // end-first, as pushed
                na += chain.size();
                out_off[++nc] = na;
            }
        }
        if (!opt->disable_score_filtering && score[end_idx] < maxs / 2) break; // rmap.cpp:502-504
    }
    return (int)nc;
}

int rawdtw_sort_by_chaining_score(const float *chaining_score, uint32_t n_chains, uint32_t *perm_out)
{
    if ((!chaining_score || !perm_out) && n_chains) return RAWDTW_ERR_INVALID;
    struct Item { float score; uint32_t idx; };
    std::vector<Item> v(n_chains);
    for (uint32_t c = 0; c < n_chains; c++) v[c] = Item{chaining_score[c], c};
    // same algorithm, same comparator outcomes as rmap.cpp:512 => same permutation, ties included
    std::sort(v.begin(), v.end(), [](const Item &a, const Item &b) { return a.score > b.score; });
    for (uint32_t c = 0; c < n_chains; c++) perm_out[c] = v[c].idx;
    return RAWDTW_OK;
}

int rawdtw_batch_build_jobs(const rawdtw_align_opt_t *opt, uint64_t n_chains, const uint64_t *anchor_off,
                            const rawdtw_anchor_t *anchors, const uint64_t *ref_base, const uint32_t *read_base,
                            uint64_t *job_off, rawdtw_job_t *jobs_out, uint64_t jobs_cap, uint64_t *n_jobs_out)
{
    if (!opt || !anchor_off || !anchors || !ref_base || !read_base || !job_off || !n_jobs_out)
        return RAWDTW_ERR_INVALID;
    uint64_t total = 0;
    for (uint64_t c = 0; c < n_chains; c++) {
        const uint32_t na = (uint32_t)(anchor_off[c + 1] - anchor_off[c]);
        const uint32_t nj = rawdtw_chain_job_count(opt, na);
        job_off[c] = total;
        if (jobs_out) {
            if (total + nj > jobs_cap) return RAWDTW_ERR_RANGE;
            if (nj) {
                int st = rawdtw_chain_build_jobs(opt, anchors + anchor_off[c], na, ref_base[c], read_base[c], 0,
                                                 jobs_out + total);
                if (st != RAWDTW_OK) return st;
            }
        }
        total += nj;
    }
    job_off[n_chains] = total;
    *n_jobs_out = total;
    return RAWDTW_OK;
}

int rawdtw_batch_replay(const rawdtw_align_opt_t *opt, uint64_t n_reads, const uint64_t *chain_off,
                        const uint64_t *anchor_off, const rawdtw_anchor_t *anchors, const uint64_t *job_off,
                        const float *job_cost, float *score, uint8_t *keep)
{
    if (!opt || !chain_off || !anchor_off || !anchors || !job_off || !job_cost || !score || !keep)
        return RAWDTW_ERR_INVALID;
    for (uint64_t r = 0; r < n_reads; r++) {
        float best = 0.0f; // rmap.cpp:515
        for (uint64_t c = chain_off[r]; c < chain_off[r + 1]; c++) {
            const uint32_t na = (uint32_t)(anchor_off[c + 1] - anchor_off[c]);
            float s;
            if (na < 1) s = 0.0f;
            else s = rawdtw_chain_replay(opt, anchors + anchor_off[c], na, job_cost + job_off[c], best);
            score[c] = s;
            keep[c] = 0;
            if (s >= opt->min_score) {  // rmap.cpp:518
                if (s > best) best = s; // rmap.cpp:519-521
                keep[c] = 1;
            }
        }
    }
    return RAWDTW_OK;
}

} // extern "C"

// ---- compact anchor lists (include/rawdtw.h): steps back from the entry before, two bytes an anchor ----
extern "C" int rawdtw_anchors_pack(uint64_t n_chains, const uint64_t *anchor_off, const rawdtw_anchor_t *anchors, rawdtw_anchor_t *heads,
                                   rawdtw_anchor_t *unit_abs, uint16_t *steps, rawdtw_wide_step_t *wide, uint64_t wide_cap, uint64_t *n_wide)
{
    if (!anchor_off || (!anchors && n_chains) || !heads || !unit_abs || !steps || !n_wide || (!wide && wide_cap)) return RAWDTW_ERR_INVALID;
    uint64_t nw = 0;
    for (uint64_t c = 0; c < n_chains; c++) {
        const uint64_t a0 = anchor_off[c], a1 = anchor_off[c + 1];
        if (a1 > a0) heads[c] = anchors[a0]; else heads[c] = rawdtw_anchor_t{0, 0};
        for (uint64_t i = a0; i < a1; i++) {
            if (i % RAWDTW_COMPACT_STRIDE == 0) unit_abs[i / RAWDTW_COMPACT_STRIDE] = anchors[i];
            if (i == a0) { steps[i] = 0; continue; }
            const rawdtw_anchor_t p = anchors[i - 1], x = anchors[i];
            if (x.query_position > p.query_position || x.target_position > p.target_position) return RAWDTW_ERR_INVALID; // (not descending)
            const uint32_t dq = p.query_position - x.query_position, dt = p.target_position - x.target_position;
            if (dq >= 255u || dt >= 255u) {
                if (nw < wide_cap) wide[nw] = rawdtw_wide_step_t{(uint32_t)i, dq, dt};
                nw++;
                steps[i] = 0xffffu;
            } else steps[i] = (uint16_t)(dq | (dt << 8));
        }
    }
    *n_wide = nw;
    return nw > wide_cap ? RAWDTW_ERR_RANGE : RAWDTW_OK;
}

extern "C" int rawdtw_anchors_unpack(uint64_t n_chains, const uint64_t *anchor_off, const rawdtw_anchor_t *heads, const rawdtw_anchor_t *unit_abs,
                                     const uint16_t *steps, const rawdtw_wide_step_t *wide, uint64_t n_wide, rawdtw_anchor_t *anchors_out)
{
    if (!anchor_off || !heads || !unit_abs || !steps || (!wide && n_wide) || (!anchors_out && n_chains)) return RAWDTW_ERR_INVALID;
    uint64_t w = 0;
    for (uint64_t c = 0; c < n_chains; c++) {
        const uint64_t a0 = anchor_off[c], a1 = anchor_off[c + 1];
        for (uint64_t i = a0; i < a1; i++) {
            if (i == a0) { anchors_out[i] = heads[c]; continue; }
            uint32_t dq = steps[i] & 0xffu, dt = steps[i] >> 8;
            if (steps[i] == 0xffffu) {
                while (w < n_wide && wide[w].index < i) w++;
                if (w >= n_wide || wide[w].index != i) return RAWDTW_ERR_INVALID;
                dq = wide[w].query_step; dt = wide[w].target_step;
            }
            anchors_out[i] = rawdtw_anchor_t{anchors_out[i - 1].target_position - dt, anchors_out[i - 1].query_position - dq};
            // (every STRIDE-th entry travels whole as well: the device decodes unit by unit; it must agree)
            if (i % RAWDTW_COMPACT_STRIDE == 0 && (unit_abs[i / RAWDTW_COMPACT_STRIDE].target_position != anchors_out[i].target_position ||
                                                   unit_abs[i / RAWDTW_COMPACT_STRIDE].query_position != anchors_out[i].query_position))
                return RAWDTW_ERR_INVALID;
        }
    }
    return RAWDTW_OK;
}

// ---- chunk rounds: which chain of the round before does a chain continue (include/rawdtw.h) ----
extern "C" int rawdtw_round_match_chains(uint64_t n_reads, const uint64_t *chain_off, const uint64_t *anchor_off, const rawdtw_anchor_t *anchors,
                                         const uint64_t *ref_base, const uint32_t *read_base, const uint64_t *prev_read,
                                         const uint64_t *prev_chain_off, const uint64_t *prev_anchor_off, const rawdtw_anchor_t *prev_anchors,
                                         const uint64_t *prev_ref_base, const uint32_t *prev_read_base, rawdtw_carry_t *carry, uint64_t *new_off,
                                         rawdtw_anchor_t *new_anchors)
{
    if (!chain_off || !anchor_off || !ref_base || !read_base || !prev_read || !prev_chain_off || !prev_anchor_off || !prev_ref_base ||
        !prev_read_base || !carry || !new_off || (!new_anchors && anchor_off[chain_off[n_reads]]))
        return RAWDTW_ERR_INVALID;
    uint64_t at = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t pr = prev_read[r];
        for (uint64_t c = chain_off[r]; c < chain_off[r + 1]; c++) {
            carry[c] = rawdtw_carry_t{RAWDTW_NO_CHAIN, 0u, 0u, rawdtw_anchor_t{0u, 0u}};
            const uint64_t a0 = anchor_off[c], a1 = anchor_off[c + 1];
            uint64_t best = 0; // anchors of the longest common tail
            if (pr != RAWDTW_NO_CHAIN && a1 >= a0 + 2) {
                // the previous chain on the same strand array that starts on the same anchor and shares the longest run of parts
                for (uint64_t pc = prev_chain_off[pr]; pc < prev_chain_off[pr + 1]; pc++) {
                    const uint64_t b0 = prev_anchor_off[pc], b1 = prev_anchor_off[pc + 1];
                    if (b1 < b0 + 2 || prev_ref_base[pc] != ref_base[c] || prev_read_base[pc] != read_base[c]) continue;
                    uint64_t same = 0; // anchors equal, counted from the chains' starts (= the lists' ends)
                    while (same < a1 - a0 && same < b1 - b0 && anchors[a1 - 1 - same].target_position == prev_anchors[b1 - 1 - same].target_position &&
                           anchors[a1 - 1 - same].query_position == prev_anchors[b1 - 1 - same].query_position)
                        same++;
                    // exclude_last_element (rmap.cpp:270): a part that is this chain's last and was not the other's cannot be taken
                    // over -- there is no exact way back from a cost without its last cell's distance (dtw.cpp:514-519)
                    if (same == a1 - a0 && same < b1 - b0) same--;
                    if (same >= 2 && same > best) { best = same; carry[c].prev_src = pc; }
                }
            }
            const uint64_t na = a1 - a0;
            if (best >= 2) {
                const uint64_t pc = carry[c].prev_src; // (the chain found above)
                const uint64_t b0 = prev_anchor_off[pc], b1 = prev_anchor_off[pc + 1];
                carry[c].parts = (uint32_t)(best - 1);
                carry[c].prev_src = b1 - best;                                 // the stretch's first entry in the previous full list
                carry[c].flags = (best == b1 - b0 && na > best) ? 1u : 0u;      // its first part was the last one then and is not now
            } else carry[c].prev_src = RAWDTW_NO_CHAIN;
            if (na) carry[c].start = anchors[a1 - 1];
            // what the device still needs: the new entries, then the junction (the carried stretch's end anchor)
            const uint64_t n_keep = carry[c].parts ? na - carry[c].parts : na;
            new_off[c] = at;
            if (n_keep) memcpy(new_anchors + at, anchors + a0, n_keep * sizeof(rawdtw_anchor_t));
            at += n_keep;
        }
    }
    new_off[chain_off[n_reads]] = at;
    return RAWDTW_OK;
}
