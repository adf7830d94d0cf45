// rawdtw_mapper.cpp -- the chunk-round mapping loop on the library's host side (include/rawdtw.h, rawdtw_mapper_*).
//
// What it restates: the control flow of map_worker_for / ri_map_frag / gen_chains (src/rmap.cpp:667-822, 545-578, 315-541)
// turned inside out so that every chunk round makes ONE device submission for all active reads (option A),
// and the PAF line of a read (src/rmap.cpp:696-801, 950-965).  Per round and active read: append the chunk's events
// (rmap.cpp:554-567), re-seed with the previous chains' anchors plus the chunk's seed hits (344-391), sort (396-401), the
// chaining DP per (sequence, strand) (430-507: rawdtw_chain_anchors), evaluation order (512).  Then one batch scores
// every chain of every read on the device (rawdtw_batch_submit_round: DTW + fold + accept/cut, unchanged parts taken over
// from the round before), and the host finishes the round: gen_primary_chains, comp_mapq, the stop rule (532-541, 692).
// Event detection and seeding stay in RawAlign (revent.c, rsketch.c, rawindex.cpp): the caller hands in each chunk's
// events and seed hits.  rawalign_amd/mapper.py is the Python mirror of this file; tests/test_abi_shim.py compares the two
// and the oracle-scored flow line by line.  Pure host code above the C ABI: no kernel is launched from here directly.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rawdtw.h"

namespace {

struct MChain {
    float chaining_score = 0.f, alignment_score = 0.f;
    uint32_t ref = 0;
    int32_t strand = 0;
    uint32_t start_position = 0, end_position = 0, mapq = 0;
    std::vector<rawdtw_anchor_t> anchors; // end-first (rmap.cpp:193-196)
    std::string aln;                      // aln:s: of the best chain (--dtw-output-cigar)
    bool has_aln = false;
};

struct MRead {
    std::string name;
    uint32_t qlen = 0, n_chunks = 0, chunks_done = 0;
    bool finished = false, broke_early = false;
    std::vector<float> events;   // p->events[read].values
    std::vector<MChain> chains;  // reg0->chains: the primary chains, best first
    uint32_t slot = 0;
};

struct PrevRound {
    rawdtw_batch *batch = nullptr;
    std::vector<uint64_t> chain_off, anchor_off, ref_base;
    std::vector<rawdtw_anchor_t> anchors;
    std::vector<uint32_t> read_base;
    std::map<uint32_t, uint64_t> index_of; // read id -> its index in that round
};

rawdtw_chain_t record_of(const MChain &c, uint32_t tag)
{
    return rawdtw_chain_t{c.chaining_score, c.alignment_score, c.ref, c.start_position, c.end_position, (uint32_t)c.anchors.size(), c.strand, 0u, tag};
}

std::string fmt_f(double x) // std::to_string(float/double) == printf("%f")
{
    char b[64];
    snprintf(b, sizeof b, "%f", x);
    return b;
}

} // namespace

struct rawdtw_mapper {
    rawdtw_ctx *ctx = nullptr;
    rawdtw_mapper_opt_t opt{};
    std::vector<std::string> seq_names;
    std::vector<uint32_t> seq_len;
    std::vector<MRead> reads;
    PrevRound prev;
    std::string log;
    uint64_t rounds = 0, parts_scored = 0, parts_reused = 0;
    std::string err;
};

namespace {

rawdtw_select_opt_t select_opt(const rawdtw_mapper *m)
{
    return rawdtw_select_opt_t{(m->opt.flag & 0x2) ? 1 : 0, m->opt.min_bestmap_ratio, m->opt.min_meanmap_ratio, m->opt.min_chain_anchor};
}

// gen_primary_chains + comp_mapq over `post` (rmap.cpp:532-536): the primary chains, best first
std::vector<MChain> primary_chains(const rawdtw_mapper *m, std::vector<MChain> &post)
{
    std::vector<MChain> out;
    if (post.empty()) return out;
    std::vector<rawdtw_chain_t> rec(post.size());
    for (size_t k = 0; k < post.size(); k++) rec[k] = record_of(post[k], (uint32_t)k);
    std::vector<uint32_t> kept(post.size());
    const rawdtw_select_opt_t so = select_opt(m);
    const uint32_t nk = rawdtw_gen_primary_chains(rec.data(), (uint32_t)rec.size(), &so, kept.data());
    for (uint32_t k = 0; k < nk; k++) out.push_back(std::move(post[rec[kept[k]].tag]));
    if (nk) out[0].mapq = rec[kept[0]].mapq;
    return out;
}

bool high_confidence(const rawdtw_mapper *m, const std::vector<MChain> &primary)
{
    if (primary.empty()) return false;
    std::vector<rawdtw_chain_t> rec(primary.size());
    for (size_t k = 0; k < primary.size(); k++) rec[k] = record_of(primary[k], (uint32_t)k);
    const rawdtw_select_opt_t so = select_opt(m);
    return rawdtw_is_mapped_with_high_confidence(rec.data(), (uint32_t)rec.size(), &so) != 0;
}

int fail(rawdtw_mapper *m, int st, const std::string &msg)
{
    if (m) m->err = msg;
    return st;
}

void drop_prev(rawdtw_mapper *m)
{
    if (m->prev.batch) rawdtw_batch_destroy(m->prev.batch);
    m->prev = PrevRound{};
}

} // namespace

extern "C" {

int rawdtw_mapper_create(rawdtw_ctx *ctx, const rawdtw_mapper_opt_t *opt, uint32_t n_seq, const char *const *seq_names,
                         const uint32_t *seq_len, rawdtw_mapper **out)
{
    if (!out) return RAWDTW_ERR_INVALID;
    *out = nullptr;
    if (!ctx || !opt || (n_seq && (!seq_names || !seq_len)) || opt->slot_events == 0 || opt->max_reads == 0) return RAWDTW_ERR_INVALID;
    if (opt->align.border_constraint != 0 && opt->align.border_constraint != 1) return RAWDTW_ERR_INVALID; // rmap.cpp:301-304
    rawdtw_mapper *m = new (std::nothrow) rawdtw_mapper;
    if (!m) return RAWDTW_ERR_OOM;
    m->ctx = ctx; m->opt = *opt;
    for (uint32_t s = 0; s < n_seq; s++) { m->seq_names.emplace_back(seq_names[s]); m->seq_len.push_back(seq_len[s]); }
    const int st = rawdtw_events_reserve(ctx, (uint64_t)opt->slot_events * opt->max_reads);
    if (st != RAWDTW_OK) { delete m; return st; }
    *out = m;
    return RAWDTW_OK;
}

int rawdtw_mapper_destroy(rawdtw_mapper *m)
{
    if (!m) return RAWDTW_OK;
    drop_prev(m);
    delete m;
    return RAWDTW_OK;
}

const char *rawdtw_mapper_last_error(const rawdtw_mapper *m) { return m ? m->err.c_str() : "null mapper"; }

int rawdtw_mapper_add_read(rawdtw_mapper *m, const char *name, uint32_t qlen, uint32_t n_chunks_available, uint32_t *read_id)
{
    if (!m || !name || !read_id) return RAWDTW_ERR_INVALID;
    if (m->reads.size() >= m->opt.max_reads) return fail(m, RAWDTW_ERR_RANGE, "more reads than the mapper has slots for");
    MRead r;
    r.name = name; r.qlen = qlen; r.n_chunks = n_chunks_available; r.slot = (uint32_t)m->reads.size();
    *read_id = r.slot;
    m->reads.push_back(std::move(r));
    return RAWDTW_OK;
}

int rawdtw_mapper_read_state(const rawdtw_mapper *m, uint32_t read_id, int *finished, uint32_t *chunks_done)
{
    if (!m || read_id >= m->reads.size()) return RAWDTW_ERR_INVALID;
    if (finished) *finished = m->reads[read_id].finished ? 1 : 0;
    if (chunks_done) *chunks_done = m->reads[read_id].chunks_done;
    return RAWDTW_OK;
}

int rawdtw_mapper_stats(const rawdtw_mapper *m, uint64_t *rounds, uint64_t *parts_scored, uint64_t *parts_reused)
{
    if (!m) return RAWDTW_ERR_INVALID;
    if (rounds) *rounds = m->rounds;
    if (parts_scored) *parts_scored = m->parts_scored;
    if (parts_reused) *parts_reused = m->parts_reused;
    return RAWDTW_OK;
}

int rawdtw_mapper_log(const rawdtw_mapper *m, const char **text)
{
    if (!m || !text) return RAWDTW_ERR_INVALID;
    *text = m->log.c_str();
    return RAWDTW_OK;
}

int rawdtw_mapper_round(rawdtw_mapper *m, uint32_t n_reads, const uint32_t *read_ids, const uint64_t *event_off, const float *events,
                        const uint64_t *hit_off, const rawdtw_seed_hit_t *hits)
{
    if (!m || (n_reads && (!read_ids || !event_off || !hit_off)) || (n_reads && event_off[n_reads] && !events) || (n_reads && hit_off[n_reads] && !hits))
        return RAWDTW_ERR_INVALID;
    if (n_reads == 0) return RAWDTW_OK;
    const uint32_t n_seq = (uint32_t)m->seq_len.size();
    m->rounds++;
    // ---- per read: the chunk's events, the round's anchors, chaining, evaluation order ----
    std::vector<std::vector<MChain>> round_chains(n_reads);
    std::vector<uint64_t> seg_src{0};
    std::vector<uint32_t> seg_dst;
    std::vector<float> new_events;
    for (uint32_t k = 0; k < n_reads; k++) {
        if (read_ids[k] >= m->reads.size()) return fail(m, RAWDTW_ERR_INVALID, "unknown read id");
        MRead &rd = m->reads[read_ids[k]];
        if (rd.finished) return fail(m, RAWDTW_ERR_INVALID, "a finished read in a round");
        const uint64_t ne = event_off[k + 1] - event_off[k];
        const uint32_t chunk_start = (uint32_t)rd.events.size(); // reg->offset (rmap.cpp:574)
        if ((uint64_t)chunk_start + ne > m->opt.slot_events) return fail(m, RAWDTW_ERR_RANGE, "a read outgrew its slot in the event arena");
        rd.events.insert(rd.events.end(), events + event_off[k], events + event_off[k + 1]); // rmap.cpp:554-567
        if (ne) {
            new_events.insert(new_events.end(), events + event_off[k], events + event_off[k + 1]);
            seg_src.push_back(seg_src.back() + ne);
            seg_dst.push_back(rd.slot * m->opt.slot_events + chunk_start);
        }
        // rmap.cpp:344-357: re-seed with the previous chains' anchors; rmap.cpp:371-391: the chunk's seed hits
        std::map<std::pair<uint32_t, int32_t>, std::vector<std::pair<uint32_t, uint32_t>>> per;
        for (const MChain &ch : rd.chains)
            for (const rawdtw_anchor_t &a : ch.anchors) per[{ch.ref, ch.strand}].push_back({a.target_position, a.query_position});
        for (uint64_t h = hit_off[k]; h < hit_off[k + 1]; h++) {
            if (hits[h].ref_seq >= n_seq) return fail(m, RAWDTW_ERR_INVALID, "seed hit on an unknown sequence");
            per[{hits[h].ref_seq, hits[h].strand}].push_back({hits[h].target_position, hits[h].query_position + chunk_start});
        }
        std::vector<MChain> chains;
        float maxs = 0.0f;
        for (uint32_t s = 0; s < n_seq; s++)       // rmap.cpp:432-433: sequence-major, strand 0 then 1
            for (int32_t st = 0; st < 2; st++) {
                auto it = per.find({s, st});
                if (it == per.end() || it->second.empty()) continue;
                std::vector<std::pair<uint32_t, uint32_t>> &lst = it->second;
                std::sort(lst.begin(), lst.end()); // by (target, query): rmap.cpp:396-401
                std::vector<rawdtw_anchor_t> a(lst.size());
                for (size_t q = 0; q < lst.size(); q++) a[q] = rawdtw_anchor_t{lst[q].first, lst[q].second};
                const uint32_t cap = (uint32_t)std::max(1, m->opt.chain.num_best_chains);
                std::vector<rawdtw_chain_out_t> outc(cap);
                std::vector<uint64_t> off(cap + 1);
                std::vector<rawdtw_anchor_t> outa(std::max<size_t>(a.size(), 1));
                const int nc = rawdtw_chain_anchors(&m->opt.chain, a.data(), (uint32_t)a.size(), &maxs, outc.data(), off.data(), outa.data(), cap, outa.size());
                if (nc < 0) return fail(m, RAWDTW_ERR_RANGE, "chain output buffers too small");
                for (int c = 0; c < nc; c++) {
                    MChain ch;
                    ch.chaining_score = outc[c].chaining_score; ch.ref = s; ch.strand = st;
                    ch.start_position = outc[c].start_position; ch.end_position = outc[c].end_position;
                    ch.anchors.assign(outa.begin() + off[c], outa.begin() + off[c + 1]);
                    chains.push_back(std::move(ch));
                }
            }
        if (!chains.empty()) { // rmap.cpp:512: std::sort by chaining score, descending (its permutation)
            std::vector<float> cs(chains.size());
            for (size_t c = 0; c < chains.size(); c++) cs[c] = chains[c].chaining_score;
            std::vector<uint32_t> perm(chains.size());
            if (rawdtw_sort_by_chaining_score(cs.data(), (uint32_t)cs.size(), perm.data()) != RAWDTW_OK) return fail(m, RAWDTW_ERR_INVALID, "sort failed");
            std::vector<MChain> sorted;
            sorted.reserve(chains.size());
            for (uint32_t p : perm) sorted.push_back(std::move(chains[p]));
            chains.swap(sorted);
        }
        round_chains[k] = std::move(chains);
    }
    // ---- the DTW block of gen_chains for every read of the round (rmap.cpp:509-530), one device submission ----
    const bool runs_dtw = (m->opt.flag & (0x2 | 0x8)) != 0; // rmap.cpp:509
    std::vector<std::vector<uint8_t>> keep_of(n_reads);
    if (runs_dtw) {
        if (seg_dst.size()) {
            const int st = rawdtw_events_append(m->ctx, new_events.data(), new_events.size(), (uint32_t)seg_dst.size(), seg_src.data(), seg_dst.data());
            if (st != RAWDTW_OK) return fail(m, st, rawdtw_last_error(m->ctx));
        }
        PrevRound cur;
        cur.chain_off.push_back(0); cur.anchor_off.push_back(0);
        for (uint32_t k = 0; k < n_reads; k++) {
            const MRead &rd = m->reads[read_ids[k]];
            cur.index_of[read_ids[k]] = k;
            for (const MChain &ch : round_chains[k]) {
                cur.anchors.insert(cur.anchors.end(), ch.anchors.begin(), ch.anchors.end());
                cur.anchor_off.push_back(cur.anchors.size());
                uint64_t rb = 0;
                if (rawdtw_reference_offset(m->ctx, ch.ref, ch.strand, &rb) != RAWDTW_OK) return fail(m, RAWDTW_ERR_INVALID, "no reference array for a chain");
                cur.ref_base.push_back(rb);
                cur.read_base.push_back(rd.slot * m->opt.slot_events);
            }
            cur.chain_off.push_back(cur.ref_base.size());
        }
        const uint64_t nc = cur.ref_base.size();
        std::vector<float> score(std::max<uint64_t>(nc, 1));
        std::vector<uint8_t> keep(std::max<uint64_t>(nc, 1));
        if (cur.ref_base.empty()) { cur.ref_base.push_back(0); cur.read_base.push_back(0); } // (non-null pointers for a round without chains)
        if (cur.anchors.empty()) cur.anchors.push_back(rawdtw_anchor_t{0, 0});
        std::vector<uint64_t> carry(std::max<uint64_t>(nc, 1), RAWDTW_NO_CHAIN);
        const bool carry_on = m->opt.carry && m->prev.batch && m->opt.align.border_constraint == 1;
        if (carry_on) {
            std::vector<uint64_t> prev_read(n_reads, RAWDTW_NO_CHAIN);
            for (uint32_t k = 0; k < n_reads; k++) {
                auto it = m->prev.index_of.find(read_ids[k]);
                if (it != m->prev.index_of.end()) prev_read[k] = it->second;
            }
            rawdtw_round_match_chains(n_reads, cur.chain_off.data(), cur.anchor_off.data(), cur.anchors.data(), cur.ref_base.data(), cur.read_base.data(),
                                      prev_read.data(), m->prev.chain_off.data(), m->prev.anchor_off.data(), m->prev.anchors.data(),
                                      m->prev.ref_base.data(), m->prev.read_base.data(), carry.data());
        }
        rawdtw_batch *b = nullptr;
        int st = rawdtw_batch_submit_round(m->ctx, &m->opt.align, n_reads, cur.chain_off.data(), cur.anchor_off.data(), cur.anchors.data(),
                                           cur.ref_base.data(), cur.read_base.data(), carry_on ? m->prev.batch : nullptr, carry.data(), &b);
        if (st == RAWDTW_OK) st = rawdtw_batch_fetch(m->ctx, b, score.data(), keep.data(), nullptr);
        if (st != RAWDTW_OK) { if (b) rawdtw_batch_destroy(b); return fail(m, st, rawdtw_last_error(m->ctx)); }
        uint64_t sc = 0, ru = 0;
        if (rawdtw_batch_round_stats(m->ctx, b, &sc, &ru) == RAWDTW_OK) { m->parts_scored += sc; m->parts_reused += ru; }
        drop_prev(m);
        if (m->opt.carry) { cur.batch = b; m->prev = std::move(cur); }
        else rawdtw_batch_destroy(b);
        const std::vector<uint64_t> &coff = m->opt.carry ? m->prev.chain_off : cur.chain_off;
        for (uint32_t k = 0; k < n_reads; k++) {
            keep_of[k].assign(round_chains[k].size(), 1);
            for (size_t c = 0; c < round_chains[k].size(); c++) {
                MChain &ch = round_chains[k][c];
                ch.alignment_score = score[coff[k] + c];
                keep_of[k][c] = keep[coff[k] + c];
                // --dtw-log-scores (rmap.cpp:308-312): in evaluation order; a cut chain returns before the fprintf
                if ((m->opt.flag & 0x8) && ch.alignment_score != -1e10f) {
                    char line[128];
                    snprintf(line, sizeof line, "chaining_score=%f alignment_score=%f\n", (double)ch.chaining_score, (double)ch.alignment_score);
                    m->log += line;
                }
            }
        }
    }
    // ---- the round's end per read: post-alignment chains, primary chains, MAPQ, stop rule ----
    for (uint32_t k = 0; k < n_reads; k++) {
        MRead &rd = m->reads[read_ids[k]];
        std::vector<MChain> post;
        for (size_t c = 0; c < round_chains[k].size(); c++)
            if (!(m->opt.flag & 0x2) || !runs_dtw || keep_of[k][c]) post.push_back(std::move(round_chains[k][c])); // rmap.cpp:525: replaced only under EVALUATE_CHAINS
        rd.chains = primary_chains(m, post);
        rd.chunks_done++;
        if (high_confidence(m, rd.chains)) { rd.finished = true; rd.broke_early = true; } // rmap.cpp:692
        else if (rd.chunks_done >= std::min(rd.n_chunks, m->opt.max_num_chunk)) rd.finished = true;
    }
    return RAWDTW_OK;
}

// --dtw-output-cigar (rmap.cpp:715-717): the best chain of every mapped read through DTW_global_tb once more, its path as the
// aln:s: string with the reference's two quirks (rmap.cpp:230-233, 283-289)
int rawdtw_mapper_finish(rawdtw_mapper *m)
{
    if (!m) return RAWDTW_ERR_INVALID;
    if (!(m->opt.flag & 0x4)) return RAWDTW_OK;
    for (MRead &rd : m->reads) {
        if (!high_confidence(m, rd.chains)) continue;
        MChain &ch = rd.chains[0];
        const uint32_t na = (uint32_t)ch.anchors.size();
        const uint32_t nj = rawdtw_chain_job_count(&m->opt.align, na);
        std::vector<rawdtw_job_t> jobs(std::max<uint32_t>(nj, 1));
        uint64_t rb = 0;
        if (rawdtw_reference_offset(m->ctx, ch.ref, ch.strand, &rb) != RAWDTW_OK) return fail(m, RAWDTW_ERR_INVALID, "no reference array for a chain");
        int st = rawdtw_chain_build_jobs(&m->opt.align, ch.anchors.data(), na, rb, 0, 1, jobs.data());
        if (st != RAWDTW_OK) return fail(m, st, st == RAWDTW_ERR_UNSUPPORTED ? "banded global alignment with --dtw-output-cigar is not implemented (rmap.cpp:223-225)" : "job building failed");
        std::vector<uint64_t> poff(nj + 1, 0);
        for (uint32_t k = 0; k < nj; k++) poff[k + 1] = poff[k] + jobs[k].n + jobs[k].m - 1;
        std::vector<uint32_t> plen(nj), pi(poff[nj]), pj(poff[nj]);
        std::vector<float> pd(poff[nj]), cost(nj);
        st = rawdtw_traceback_batch(m->ctx, jobs.data(), nj, rd.events.data(), rd.events.size(), cost.data(), poff.data(), plen.data(), pi.data(),
                                    pj.data(), pd.data());
        if (st != RAWDTW_OK) return fail(m, st, rawdtw_last_error(m->ctx));
        ch.alignment_score = rawdtw_chain_replay(&m->opt.align, ch.anchors.data(), na, cost.data(), -1e10f); // rmap.cpp:306 on the summed costs
        std::string s;
        const uint32_t parts = na - 1;
        char el[96];
        for (uint32_t k = 0; k < nj; k++) {
            // sparse: every element offset by its part's start anchor (rmap.cpp:286-289); global: the offsets are added to
            // alignment.back() once per element (rmap.cpp:230-233), i.e. only the last tuple moves
            const rawdtw_anchor_t &s0 = m->opt.align.border_constraint == 0 ? ch.anchors[na - 1] : ch.anchors[parts - k];
            for (uint32_t q = 0; q < plen[k]; q++) {
                unsigned long long i = pi[poff[k] + q], j = pj[poff[k] + q];
                if (m->opt.align.border_constraint != 0) { i += s0.query_position; j += s0.target_position; }
                else if (q + 1 == plen[k]) { i += (unsigned long long)plen[k] * s0.query_position; j += (unsigned long long)plen[k] * s0.target_position; }
                snprintf(el, sizeof el, "(%llu,%llu,%g)", i, j, (double)pd[poff[k] + q]); // ostream << float == %g (rmap.cpp:580-592)
                s += el;
            }
        }
        ch.aln = std::move(s);
        ch.has_aln = true;
        if (m->opt.flag & 0x8) {
            char line[128];
            snprintf(line, sizeof line, "chaining_score=%f alignment_score=%f\n", (double)ch.chaining_score, (double)ch.alignment_score);
            m->log += line;
        }
    }
    return RAWDTW_OK; // (the traceback calls replaced the event arena's contents: the mapper's rounds are over)
}

// The PAF line of one read (rmap.cpp:696-801 for the fields and tags, 956-965 for the format).  `mt:f:` is wall-clock in the
// reference and therefore written as 0 here.
int rawdtw_mapper_paf(const rawdtw_mapper *m, uint32_t read_id, char *buf, uint32_t cap, uint32_t *len)
{
    if (!m || read_id >= m->reads.size() || !len) return RAWDTW_ERR_INVALID;
    const MRead &rd = m->reads[read_id];
    const uint32_t l_chunk = m->opt.chunk_size, max_chunk = m->opt.max_num_chunk;
    uint32_t current_chunk = rd.broke_early ? rd.chunks_done - 1 : rd.chunks_done; // the loop's current_chunk when it exits
    const uint64_t chunk_start = (uint64_t)current_chunk * l_chunk;
    // rmap.cpp:696: step back one chunk when the loop ran out of signal or chunks rather than breaking
    if (!rd.broke_early && current_chunk > 0 && (chunk_start >= rd.qlen || current_chunk == max_chunk)) current_chunk -= 1;
    const uint32_t offset = (uint32_t)rd.events.size(); // reg0->offset: events consumed so far
    // rmap.cpp:698, float arithmetic throughout
    const float scale = offset ? ((float)(current_chunk + 1) * (float)l_chunk / (float)offset) / ((float)m->opt.sample_rate / (float)m->opt.bp_per_sec)
                               : INFINITY;
    const std::vector<MChain> &chains = rd.chains;
    const uint32_t n_chains = (uint32_t)chains.size(), n_anchors0 = n_chains ? (uint32_t)chains[0].anchors.size() : 0;
    float mean_chain_score = 0.f;
    for (const MChain &c : chains) mean_chain_score += c.chaining_score;
    if (n_chains) mean_chain_score /= (float)n_chains;
    const bool mapped = high_confidence(m, chains);
    float at = 0.f, aq = 0.f;
    if (n_chains) { // rmap.cpp:719-724: uint32 differences accumulated in float
        const std::vector<rawdtw_anchor_t> &a = chains[0].anchors;
        for (uint32_t ai = 0; ai + 1 < n_anchors0; ai++) {
            at += (float)(uint32_t)(a[ai].target_position - a[ai + 1].target_position);
            aq += (float)(uint32_t)(a[ai].query_position - a[ai + 1].query_position);
        }
        if (n_anchors0) { at /= (float)n_anchors0; aq /= (float)n_anchors0; }
    }
    std::string tags = "mt:f:" + fmt_f(0.0) + "\tci:i:" + std::to_string(current_chunk + 1) + "\tsl:i:" + std::to_string(rd.qlen);
    std::string line;
    char head[512];
    if (n_chains) {
        tags += "\tcm:i:" + std::to_string(n_anchors0) + "\tnc:i:" + std::to_string(n_chains) + "\ts1:f:" + fmt_f(chains[0].chaining_score) + "\ts2:f:" +
                fmt_f(n_chains > 1 ? (double)chains[1].chaining_score : 0.0) + "\tsm:f:" + fmt_f(mean_chain_score) + "\tat:f:" + fmt_f(at) + "\taq:f:" + fmt_f(aq);
    } else tags += "\tcm:i:0\tnc:i:0\ts1:f:0\ts2:f:0\tsm:f:0\tat:f:0\taq:f:0";
    if (mapped) {
        const MChain &c0 = chains[0];
        if ((m->opt.flag & 0x4) && c0.has_aln) tags += "\talns:f:" + fmt_f(c0.alignment_score) + "\taln:s:" + c0.aln;
        const std::vector<rawdtw_anchor_t> &a = c0.anchors;
        const uint32_t read_end = (uint32_t)(scale * (float)a[0].query_position);
        const uint32_t read_start = (uint32_t)(scale * (float)a[n_anchors0 - 1].query_position);
        const uint32_t ref_len = m->seq_len[c0.ref];
        const uint32_t frag_start = c0.strand ? ref_len + 1u - c0.end_position : c0.start_position; // rmap.cpp:751
        const uint32_t frag_len = c0.end_position - c0.start_position + 1u;
        snprintf(head, sizeof head, "%s\t%u\t%u\t%u\t%s\t%s\t%u\t%u\t%u\t%u\t%u\t%d\t", rd.name.c_str(), read_end, read_start, read_end,
                 c0.strand ? "-" : "+", m->seq_names[c0.ref].c_str(), ref_len, frag_start, frag_start + frag_len, read_end - read_start - 1u, frag_len,
                 (int)c0.mapq); // rmap.cpp:961-963
        line = head + tags;
    } else {
        const uint32_t read_length = offset ? (uint32_t)(scale * (float)offset) : 0u;
        snprintf(head, sizeof head, "%s\t%u\t*\t*\t*\t*\t*\t*\t*\t*\t*\t%d\t", rd.name.c_str(), read_length, 0); // rmap.cpp:965
        line = head + tags;
    }
    *len = (uint32_t)line.size();
    if (buf && cap) {
        const uint32_t n = std::min<uint32_t>(cap - 1, (uint32_t)line.size());
        memcpy(buf, line.data(), n);
        buf[n] = 0;
    }
    return (buf && cap > line.size()) || !buf ? RAWDTW_OK : RAWDTW_ERR_RANGE;
}

} // extern "C"
