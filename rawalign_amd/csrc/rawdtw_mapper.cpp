// rawdtw_mapper.cpp -- the chunk-round mapping loop on the library's host side (include/rawdtw.h, rawdtw_mapper_*).
//
// What it restates: the control flow of map_worker_for / ri_map_frag / gen_chains (src/rmap.cpp:667-822, 545-578, 315-541)
// turned inside out so that every chunk round makes ONE device submission per read group (option A), and the
// PAF line of a read (src/rmap.cpp:696-801, 950-965).
//
// A round, per read group (the reads are dealt over one or two groups, each with a context of its own):
//   host phase   a pool of threads over the group's reads, as the reference runs kt_for over n_threads reads
//                (rmap.cpp:916): append the chunk's events (rmap.cpp:554-567), re-seed with the previous chains' anchors plus the
//                chunk's seed hits (344-391), sort (396-401), the chaining DP per (sequence, strand) (430-507:
//                rawdtw_chain_anchors), evaluation order (512), and -- chunk rounds with carry -- per chain the chain of the
//                round before it continues and the number of leading parts that did not change, compared anchor by anchor
//   lay-out      the round's arrays in pinned memory: chain and anchor offsets, bases, the NEW anchors and the carry records
//                (or the whole lists for a round without a predecessor), the new events' segments
//   submit       rawdtw_events_append + rawdtw_batch_submit_carry / rawdtw_batch_submit: enqueued, not waited for
// then, group by group: fetch (the only wait), and the round's end per read on the pool: gen_primary_chains, comp_mapq, the
// stop rule (532-541, 692).  With two groups one group's host phase runs while the other's batch is on the device, and one
// group's round end while the other's batch finishes -- the overlap the reference gets from its two pipeline workers
// (rmap.cpp:1015,1033).
//
// Event detection and seeding stay in RawAlign (revent.c, rsketch.c, rawindex.cpp): the caller hands in each chunk's events
// and seed hits.  rawalign_amd/mapper.py is the Python mirror of the control flow; tests/test_mapper.py and
// tests/test_abi_shim.py compare the two and the oracle-scored flow line by line.  Pure host code above the C ABI: no kernel
// is launched from here directly.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
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
    bool finished = false, broke_early = false, released = false;
    std::vector<float> events;   // p->events[read].values -- the host's copy: kept for what reads it (an external scorer, --dtw-output-cigar's
                                 // traceback at the end); a mapper that only scores on the device keeps the count (the events are in the arena)
    uint32_t n_events = 0;
    uint32_t offset = 0;         // reg->offset: events of the chunks that were chained (rmap.cpp:574; a chunk below min_events does not count)
    std::vector<MChain> chains;  // reg0->chains: the primary chains, best first
    uint32_t slot = 0;           // its place in the mapper's event arenas: group = slot % groups, place there = slot / groups
    uint64_t last_round = 0;     // the round it was last scored in, and its position among its group's reads then
    uint64_t last_pos = 0;
    uint64_t seen_round = 0;     // (duplicate check)
};

// growable array in page-locked memory (plain memory for a mapper without a device); contents are NOT kept over a growth
template <typename T> struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    ~PinBuf() { release(); }
    void release()
    {
        if (p) { if (pinned) rawdtw_host_free(p); else free(p); }
        p = nullptr; cap = 0;
    }
    // (`keep`: the first `keep` elements survive a move)
    bool ensure(size_t n, bool want_pinned, size_t keep = 0)
    {
        if (n <= cap) return true;
        const size_t c = n + n / 4 + 64;
        void *q = nullptr;
        bool pin = false;
        if (want_pinned && rawdtw_host_alloc(c * sizeof(T), &q) == RAWDTW_OK && q) pin = true;
        else q = malloc(c * sizeof(T));
        if (!q) return false;
        if (p && keep) memcpy(q, p, std::min(keep, cap) * sizeof(T));
        release();
        p = static_cast<T *>(q); cap = c; pinned = pin;
        return true;
    }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

// one round's arrays of one read group, as handed to the device (and kept for the next round's matching)
struct RoundArrays {
    std::vector<uint32_t> ks;  // the group's reads: indices into the round's read list, in order
    PinBuf<uint64_t> chain_off, anchor_off, ref_base, new_off, seg_src;
    PinBuf<uint32_t> read_base, seg_dst;
    PinBuf<rawdtw_anchor_t> anchors, new_anchors;
    PinBuf<rawdtw_carry_t> carry;
    PinBuf<float> new_events, score;
    PinBuf<uint8_t> keep;
    PinBuf<uint64_t> seed_off;           // device chaining (opt.device_chain): the reads' seed lists in, the chains' records out
    PinBuf<rawdtw_seed_t> seeds;
    PinBuf<rawdtw_chain_rec_t> recs;
    bool device_chained = false;
    std::vector<uint32_t> chain_seq; // (the external scorer's view)
    std::vector<int32_t> chain_strand;
    uint64_t n_reads = 0, n_chains = 0, n_anchors = 0, n_new = 0, n_new_events = 0, n_seg = 0;
    rawdtw_batch *batch = nullptr;
    uint64_t round_id = 0;
    bool carried = false;
};

struct Group {
    rawdtw_ctx *ctx = nullptr;
    bool own_ctx = false;
    RoundArrays buf[2];
    int cur = 0;          // buf[cur]: the round at hand; buf[cur ^ 1]: the round before (when has_prev)
    bool has_prev = false;
    // the largest round so far: BOTH buffers are sized to it when it grows (page-locked memory is slow to get -- ~0.2 ms a megabyte --, and a
    // round that is the first of its size in ITS buffer would pay that again one round after its neighbour did)
    uint64_t hw_reads = 0, hw_chains = 0, hw_anchors = 0, hw_new = 0, hw_events = 0, hw_seg = 0, hw_seeds = 0;
};

// what the host phase leaves per read of the round
struct RoundRead {
    std::vector<MChain> chains;          // the round's candidate chains in evaluation order
    std::vector<rawdtw_carry_t> carry;   // per chain
    std::vector<uint64_t> ref_base;      // per chain
    std::vector<uint8_t> keep;
    std::string log;
    uint64_t ne = 0;
    uint32_t ev_before = 0, off_before = 0;
    bool skipped = false;                // a chunk below min_events: no chaining, chains and offset stay (rmap.cpp:569-575)
    bool high = false;                   // the round's end: mapped with high confidence (rmap.cpp:692)
    uint64_t chain0 = 0, anchor0 = 0, new0 = 0, ev0 = 0; // its first chain / anchor / new anchor / new event in the group's arrays
    uint64_t seed0 = 0, n_seeds = 0;     // device chaining: its seeds in the group's list
    uint32_t chunk_start = 0;
    int err = RAWDTW_OK;
};

// a pool of threads running one loop at a time; the calling thread works too
class Pool {
public:
    explicit Pool(int threads)
    {
        for (int t = 1; t < threads; t++) th_.emplace_back([this] { worker(); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // fn(i) for i in [0, n), dealt in pieces of `grain` from a shared counter (as kt_for deals reads: kthread.c:54-72).
    // A round is half a dozen such loops a few hundred microseconds apart: a worker that has run out of work keeps looking for the next loop
    // for a short while before it goes to sleep, and the caller for the last worker -- waking sixteen threads through a condition variable
    // was 50-100 us a loop, most of a small round.
    void run(size_t n, size_t grain, const std::function<void(size_t)> &fn)
    {
        if (n == 0) return;
        if (th_.empty() || n <= grain) { for (size_t i = 0; i < n; i++) fn(i); return; }
        fn_ = &fn; n_ = n; grain_ = grain; next_.store(0, std::memory_order_relaxed);
        busy_.store((int)th_.size(), std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (sleeping_ > 0) cv_.notify_all();
        }
        work();
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(2000);
        while (busy_.load(std::memory_order_acquire) != 0) {
            if (std::chrono::steady_clock::now() > until) {
                std::unique_lock<std::mutex> lk(mu_);
                waiting_ = true;
                done_.wait(lk, [this] { return busy_.load(std::memory_order_acquire) == 0; });
                waiting_ = false;
                break;
            }
            relax();
        }
        fn_ = nullptr;
    }
    int threads() const { return (int)th_.size() + 1; }

private:
    static void relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void work()
    {
        for (;;) {
            const size_t s = next_.fetch_add(grain_);
            if (s >= n_) break;
            const size_t e = std::min(n_, s + grain_);
            for (size_t i = s; i < e; i++) (*fn_)(i);
        }
    }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            uint64_t g = gen_.load(std::memory_order_acquire);
            if (g == seen) { // nothing yet: look for a while, then sleep
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
                while ((g = gen_.load(std::memory_order_acquire)) == seen && std::chrono::steady_clock::now() < until) relax();
                if (g == seen) {
                    std::unique_lock<std::mutex> lk(mu_);
                    sleeping_++;
                    cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                    sleeping_--;
                    g = gen_.load(std::memory_order_acquire);
                }
            }
            seen = g;
            if (stop_) return;
            work();
            if (busy_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                if (waiting_) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0, grain_ = 1;
    std::atomic<size_t> next_{0};
    std::atomic<int> busy_{0};
    std::atomic<uint64_t> gen_{0};
    int sleeping_ = 0;      // (under mu_)
    bool waiting_ = false;  // (under mu_) the caller sleeps on done_
    std::atomic<bool> stop_{false};
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

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

struct rawdtw_mapper {
    rawdtw_ctx *ctx = nullptr;
    rawdtw_mapper_opt_t opt{};
    std::vector<std::string> seq_names;
    std::vector<uint32_t> seq_len;
    std::vector<uint64_t> ref_off; // [seq * 2 + strand]: the strand array's offset in the reference arena (a scorer-only mapper: its index)
    std::vector<MRead> reads;
    std::vector<uint32_t> free_slots;
    uint32_t slots_used = 0;
    Group groups_store[2];
    struct GroupSpan { Group *b; size_t n; Group *begin() const { return b; } Group *end() const { return b + n; } size_t size() const { return n; } Group &operator[](size_t i) const { return b[i]; } } groups{groups_store, 1};
    Pool *pool = nullptr;
    std::string log;
    uint64_t rounds = 0, parts_scored = 0, parts_reused = 0;
    double timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    rawdtw_scorer_fn scorer = nullptr;
    void *scorer_user = nullptr;
    bool keep_host_events = true; // (false: scored on the device only and no CIGAR asked for -- nothing reads the host's copy of a read's events)
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
    static thread_local std::vector<rawdtw_chain_t> rec; // (scratch of the pool's threads: a round calls this once a read)
    static thread_local std::vector<uint32_t> kept;
    rec.resize(post.size()); kept.resize(post.size());
    for (size_t k = 0; k < post.size(); k++) rec[k] = record_of(post[k], (uint32_t)k);
    const rawdtw_select_opt_t so = select_opt(m);
    const uint32_t nk = rawdtw_gen_primary_chains(rec.data(), (uint32_t)rec.size(), &so, kept.data());
    out.reserve(nk);
    for (uint32_t k = 0; k < nk; k++) out.push_back(std::move(post[rec[kept[k]].tag]));
    if (nk) out[0].mapq = rec[kept[0]].mapq;
    return out;
}

bool high_confidence(const rawdtw_mapper *m, const std::vector<MChain> &primary)
{
    if (primary.empty()) return false;
    static thread_local std::vector<rawdtw_chain_t> rec;
    rec.resize(primary.size());
    for (size_t k = 0; k < primary.size(); k++) rec[k] = record_of(primary[k], (uint32_t)k);
    const rawdtw_select_opt_t so = select_opt(m);
    return rawdtw_is_mapped_with_high_confidence(rec.data(), (uint32_t)rec.size(), &so) != 0;
}

int fail(rawdtw_mapper *m, int st, const std::string &msg)
{
    if (m) m->err = msg;
    return st;
}

void drop_batches(rawdtw_mapper *m)
{
    for (Group &g : m->groups)
        for (RoundArrays &ra : g.buf) {
            if (ra.batch) rawdtw_batch_destroy(ra.batch);
            ra.batch = nullptr;
        }
}

struct Seed { uint32_t key, t, q; };

// The host phase of one read: the chunk's events, the round's anchors, chaining, evaluation order, carry records.
// `pv` = the arrays of the round before of the read's group (null: none, or the read was not in it).
// the chunk's events (rmap.cpp:554-575); false: the chunk is below min_events -- no gen_chains this round
bool host_phase_events(rawdtw_mapper *m, MRead &rd, RoundRead &rr, const float *ev, uint64_t ne)
{
    rr.ne = ne;
    rr.ev_before = rd.n_events;
    rr.off_before = rd.offset;
    if (m->keep_host_events) rd.events.insert(rd.events.end(), ev, ev + ne); // rmap.cpp:554-567
    rd.n_events += (uint32_t)ne;
    if (ne < m->opt.min_events) { rr.skipped = true; return false; } // rmap.cpp:569-572: no gen_chains, reg->offset stays
    rr.chunk_start = rd.offset;  // reg->offset (rmap.cpp:574)
    rd.offset += (uint32_t)ne;   // rmap.cpp:575
    return true;
}

void host_phase_chain(rawdtw_mapper *m, MRead &rd, RoundRead &rr, const rawdtw_seed_hit_t *hits, uint64_t n_hits, const RoundArrays *pv, bool runs_dtw);

void host_phase_read(rawdtw_mapper *m, MRead &rd, RoundRead &rr, const float *ev, uint64_t ne, const rawdtw_seed_hit_t *hits, uint64_t n_hits,
                     const RoundArrays *pv, bool runs_dtw)
{
    if (host_phase_events(m, rd, rr, ev, ne)) host_phase_chain(m, rd, rr, hits, n_hits, pv, runs_dtw);
}

void host_phase_chain(rawdtw_mapper *m, MRead &rd, RoundRead &rr, const rawdtw_seed_hit_t *hits, uint64_t n_hits, const RoundArrays *pv, bool runs_dtw)
{
    const uint32_t chunk_start = rr.chunk_start;
    // rmap.cpp:344-357: re-seed with the previous chains' anchors; rmap.cpp:371-391: the chunk's seed hits
    std::vector<Seed> seeds;
    size_t n_prev = 0;
    for (const MChain &ch : rd.chains) n_prev += ch.anchors.size();
    seeds.reserve(n_prev + n_hits);
    for (const MChain &ch : rd.chains)
        for (const rawdtw_anchor_t &a : ch.anchors) seeds.push_back(Seed{ch.ref * 2u + (uint32_t)ch.strand, a.target_position, a.query_position});
    for (uint64_t h = 0; h < n_hits; h++)
        seeds.push_back(Seed{hits[h].ref_seq * 2u + (uint32_t)(hits[h].strand ? 1 : 0), hits[h].target_position, hits[h].query_position + chunk_start});
    // by (sequence, strand), then (target, query): rmap.cpp:396-401 sorts every list; rmap.cpp:432-433 walks them sequence-major,
    // strand 0 then 1
    std::sort(seeds.begin(), seeds.end(), [](const Seed &a, const Seed &b) {
        if (a.key != b.key) return a.key < b.key;
        if (a.t != b.t) return a.t < b.t;
        return a.q < b.q;
    });
    std::vector<MChain> chains;
    float maxs = 0.0f;
    const uint32_t cap = (uint32_t)std::max(1, m->opt.chain.num_best_chains);
    std::vector<rawdtw_chain_out_t> outc(cap);
    std::vector<uint64_t> off(cap + 1);
    std::vector<rawdtw_anchor_t> a, outa;
    for (size_t s0 = 0; s0 < seeds.size();) {
        size_t s1 = s0;
        while (s1 < seeds.size() && seeds[s1].key == seeds[s0].key) s1++;
        a.resize(s1 - s0);
        for (size_t q = s0; q < s1; q++) a[q - s0] = rawdtw_anchor_t{seeds[q].t, seeds[q].q};
        outa.resize(std::max<size_t>(a.size(), 1));
        const int nc = rawdtw_chain_anchors(&m->opt.chain, a.data(), (uint32_t)a.size(), &maxs, outc.data(), off.data(), outa.data(), cap, outa.size());
        if (nc < 0) { rr.err = RAWDTW_ERR_RANGE; return; }
        for (int c = 0; c < nc; c++) {
            MChain ch;
            ch.chaining_score = outc[c].chaining_score; ch.ref = seeds[s0].key >> 1; ch.strand = (int32_t)(seeds[s0].key & 1u);
            ch.start_position = outc[c].start_position; ch.end_position = outc[c].end_position;
            ch.anchors.assign(outa.begin() + off[c], outa.begin() + off[c + 1]);
            chains.push_back(std::move(ch));
        }
        s0 = s1;
    }
    if (!chains.empty() && runs_dtw) { // rmap.cpp:512: std::sort by chaining score, descending (its permutation)
        std::vector<float> cs(chains.size());
        for (size_t c = 0; c < chains.size(); c++) cs[c] = chains[c].chaining_score;
        std::vector<uint32_t> perm(chains.size());
        if (rawdtw_sort_by_chaining_score(cs.data(), (uint32_t)cs.size(), perm.data()) != RAWDTW_OK) { rr.err = RAWDTW_ERR_INVALID; return; }
        std::vector<MChain> sorted;
        sorted.reserve(chains.size());
        for (uint32_t p : perm) sorted.push_back(std::move(chains[p]));
        chains.swap(sorted);
    }
    rr.chains = std::move(chains);
    if (!runs_dtw) return;
    rr.ref_base.resize(rr.chains.size());
    rr.carry.assign(rr.chains.size(), rawdtw_carry_t{RAWDTW_NO_CHAIN, 0u, 0u, rawdtw_anchor_t{0u, 0u}});
    for (size_t c = 0; c < rr.chains.size(); c++) rr.ref_base[c] = m->ref_off[rr.chains[c].ref * 2u + (uint32_t)rr.chains[c].strand];
    if (!pv) return;
    // chunk rounds: the chain of the round before this chain continues -- same strand array, same start anchor, the longest
    // common tail, compared anchor by anchor -- and the leading parts taken over (as rawdtw_round_match_chains)
    const uint64_t pr = rd.last_pos;
    for (size_t c = 0; c < rr.chains.size(); c++) {
        const std::vector<rawdtw_anchor_t> &an = rr.chains[c].anchors;
        const uint64_t na = an.size();
        if (na) rr.carry[c].start = an[na - 1];
        if (na < 2) continue;
        uint64_t best = 0, best_b0 = 0, best_b1 = 0;
        for (uint64_t pc = pv->chain_off[pr]; pc < pv->chain_off[pr + 1]; pc++) {
            const uint64_t b0 = pv->anchor_off[pc], b1 = pv->anchor_off[pc + 1];
            if (b1 < b0 + 2 || pv->ref_base[pc] != rr.ref_base[c]) continue;
            const rawdtw_anchor_t *pa = pv->anchors.p;
            uint64_t same = 0;
            while (same < na && same < b1 - b0 && an[na - 1 - same].target_position == pa[b1 - 1 - same].target_position &&
                   an[na - 1 - same].query_position == pa[b1 - 1 - same].query_position)
                same++;
            if (same == na && same < b1 - b0) same--; // (its last part was not the last then: rmap.cpp:270, no exact way back)
            if (same >= 2 && same > best) { best = same; best_b0 = b0; best_b1 = b1; }
        }
        if (best >= 2) {
            rr.carry[c].parts = (uint32_t)(best - 1);
            rr.carry[c].prev_src = best_b1 - best;                                    // the stretch's first entry in the previous full list
            rr.carry[c].flags = (best == best_b1 - best_b0 && na > best) ? 1u : 0u;    // its first part was the last one then and is not now
        }
    }
}

} // namespace

extern "C" {

int rawdtw_mapper_create(rawdtw_ctx *ctx, const rawdtw_mapper_opt_t *opt, uint32_t n_seq, const char *const *seq_names,
                         const uint32_t *seq_len, rawdtw_mapper **out)
{
    if (!out) return RAWDTW_ERR_INVALID;
    *out = nullptr;
    if (!opt || (n_seq && (!seq_names || !seq_len)) || opt->slot_events == 0 || opt->max_reads == 0) return RAWDTW_ERR_INVALID;
    if (opt->align.border_constraint != 0 && opt->align.border_constraint != 1) return RAWDTW_ERR_INVALID; // rmap.cpp:301-304
    rawdtw_mapper *m = new (std::nothrow) rawdtw_mapper;
    if (!m) return RAWDTW_ERR_OOM;
    m->ctx = ctx; m->opt = *opt;
    m->opt.groups = (ctx && opt->groups >= 2) ? 2 : 1;
    m->keep_host_events = !ctx || (opt->flag & 0x4) || !(opt->flag & (0x2 | 0x8));
    m->opt.threads = std::max(1, std::min(opt->threads, 256));
    for (uint32_t s = 0; s < n_seq; s++) { m->seq_names.emplace_back(seq_names[s]); m->seq_len.push_back(seq_len[s]); }
    m->ref_off.resize(2ull * n_seq);
    for (uint32_t s = 0; s < n_seq; s++)
        for (int st = 0; st < 2; st++) {
            uint64_t off = 2ull * s + (uint64_t)st;
            if (ctx && rawdtw_reference_offset(ctx, s, st, &off) != RAWDTW_OK) { delete m; return RAWDTW_ERR_INVALID; } // (no reference array for a sequence)
            m->ref_off[2ull * s + (uint64_t)st] = off;
        }
    m->groups.n = (size_t)m->opt.groups;
    m->groups[0].ctx = ctx;
    int st = RAWDTW_OK;
    if (m->opt.groups == 2) { // the second group's context: same device, the same resident reference
        int dev = 0;
        st = rawdtw_context_device(ctx, &dev);
        if (st == RAWDTW_OK) st = rawdtw_create(dev, &m->groups[1].ctx);
        if (st == RAWDTW_OK) { m->groups[1].own_ctx = true; st = rawdtw_share_reference(m->groups[1].ctx, ctx); }
    }
    const uint64_t per_group = ((uint64_t)opt->max_reads + (uint64_t)m->opt.groups - 1) / (uint64_t)m->opt.groups;
    for (Group &g : m->groups)
        if (st == RAWDTW_OK && g.ctx) st = rawdtw_events_reserve(g.ctx, (uint64_t)opt->slot_events * per_group);
    if (st != RAWDTW_OK) { rawdtw_mapper_destroy(m); return st; }
    m->pool = new (std::nothrow) Pool(m->opt.threads);
    if (!m->pool) { rawdtw_mapper_destroy(m); return RAWDTW_ERR_OOM; }
    *out = m;
    return RAWDTW_OK;
}

int rawdtw_mapper_destroy(rawdtw_mapper *m)
{
    if (!m) return RAWDTW_OK;
    drop_batches(m);
    for (Group &g : m->groups) {
        for (RoundArrays &ra : g.buf) { // (pinned memory goes before the context that may own the device)
            ra.chain_off.release(); ra.anchor_off.release(); ra.ref_base.release(); ra.new_off.release(); ra.seg_src.release();
            ra.read_base.release(); ra.seg_dst.release(); ra.anchors.release(); ra.new_anchors.release(); ra.carry.release();
            ra.new_events.release(); ra.score.release(); ra.keep.release();
        }
        if (g.own_ctx && g.ctx) rawdtw_destroy(g.ctx);
    }
    delete m->pool;
    delete m;
    return RAWDTW_OK;
}

const char *rawdtw_mapper_last_error(const rawdtw_mapper *m) { return m ? m->err.c_str() : "null mapper"; }

int rawdtw_mapper_set_scorer(rawdtw_mapper *m, rawdtw_scorer_fn fn, void *user)
{
    if (!m) return RAWDTW_ERR_INVALID;
    if (fn && !m->keep_host_events) { // (an external scorer reads the host's copy of the reads' events: from the first round on)
        for (const MRead &rd : m->reads)
            if (rd.n_events && !rd.released) return fail(m, RAWDTW_ERR_INVALID, "set the scorer before the first round");
        m->keep_host_events = true;
    }
    m->scorer = fn; m->scorer_user = user;
    drop_batches(m); // (a round scored elsewhere leaves nothing to carry from)
    for (Group &g : m->groups) g.has_prev = false;
    return RAWDTW_OK;
}

int rawdtw_mapper_add_read(rawdtw_mapper *m, const char *name, uint32_t qlen, uint32_t n_chunks_available, uint32_t *read_id)
{
    if (!m || !name || !read_id) return RAWDTW_ERR_INVALID;
    uint32_t slot;
    if (!m->free_slots.empty()) { slot = m->free_slots.back(); m->free_slots.pop_back(); }
    else if (m->slots_used < m->opt.max_reads) slot = m->slots_used++;
    else return fail(m, RAWDTW_ERR_RANGE, "more reads than the mapper has slots for (rawdtw_mapper_release_read gives a finished read's slot back)");
    MRead r;
    r.name = name; r.qlen = qlen; r.n_chunks = n_chunks_available; r.slot = slot;
    *read_id = (uint32_t)m->reads.size();
    m->reads.push_back(std::move(r));
    return RAWDTW_OK;
}

int rawdtw_mapper_release_read(rawdtw_mapper *m, uint32_t read_id)
{
    if (!m || read_id >= m->reads.size()) return RAWDTW_ERR_INVALID;
    MRead &rd = m->reads[read_id];
    if (rd.released) return RAWDTW_OK;
    if (!rd.finished) return fail(m, RAWDTW_ERR_INVALID, "only a finished read can be released");
    rd.released = true;
    std::vector<float>().swap(rd.events);
    std::vector<MChain>().swap(rd.chains);
    m->free_slots.push_back(rd.slot);
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

int rawdtw_mapper_timing(const rawdtw_mapper *m, double out[8])
{
    if (!m || !out) return RAWDTW_ERR_INVALID;
    for (int i = 0; i < 8; i++) out[i] = m->timing[i];
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
    double t0 = now_ms(); // (the checks and the round's set-up count as host phase, its commit as round end: the five times add up to the call)
    const uint32_t n_seq = (uint32_t)m->seq_len.size();
    const bool runs_dtw = (m->opt.flag & (0x2 | 0x8)) != 0; // rmap.cpp:509
    if (runs_dtw && !m->scorer && !m->ctx) return fail(m, RAWDTW_ERR_NO_DEVICE, "a mapper without a context needs a scorer (rawdtw_mapper_set_scorer)");
    // ---- every read is checked before anything changes ----
    const uint64_t stamp = m->rounds + 1;
    for (uint32_t k = 0; k < n_reads; k++) {
        if (read_ids[k] >= m->reads.size()) return fail(m, RAWDTW_ERR_INVALID, "unknown read id");
        MRead &rd = m->reads[read_ids[k]];
        bool dup = rd.seen_round == stamp;
        rd.seen_round = stamp;
        if (dup || rd.finished || rd.released || event_off[k + 1] < event_off[k] || hit_off[k + 1] < hit_off[k]) {
            for (uint32_t q = 0; q <= k; q++) m->reads[read_ids[q]].seen_round = 0;
            return fail(m, RAWDTW_ERR_INVALID, dup ? "a read twice in one round" : rd.finished || rd.released ? "a finished read in a round" : "offsets do not ascend");
        }
    }
    for (uint32_t k = 0; k < n_reads; k++) m->reads[read_ids[k]].seen_round = 0; // (a failed round does not count)
    for (uint32_t k = 0; k < n_reads; k++) {
        const MRead &rd = m->reads[read_ids[k]];
        if ((uint64_t)rd.n_events + (event_off[k + 1] - event_off[k]) > m->opt.slot_events) return fail(m, RAWDTW_ERR_RANGE, "a read outgrew its slot in the event arena");
        for (uint64_t h = hit_off[k]; h < hit_off[k + 1]; h++)
            if (hits[h].ref_seq >= n_seq) return fail(m, RAWDTW_ERR_INVALID, "seed hit on an unknown sequence");
    }
    const uint64_t round_id = m->rounds + 1;
    const uint32_t G = (uint32_t)m->groups.size();
    const bool on_device = runs_dtw && !m->scorer;
    std::vector<RoundRead> rr(n_reads);
    for (Group &g : m->groups) { g.cur ^= 1; g.buf[g.cur].ks.clear(); }
    for (uint32_t k = 0; k < n_reads; k++) { Group &g = m->groups[m->reads[read_ids[k]].slot % G]; g.buf[g.cur].ks.push_back(k); }
    int status = RAWDTW_OK;
    std::string status_msg;
    auto set_fail = [&](int st, const std::string &msg) { if (status == RAWDTW_OK) { status = st; status_msg = msg; } };
    // ---- device chaining (opt.device_chain), first half: the host phase is the events and the seed lists; sort, chaining DP, traceback and order
    // are enqueued (rawdtw_chain_round_begin) and run while the next group's host phase does.  false: the device declined (its cap on seeds a read)
    // -- the events are in place on both sides, the round is chained on the host.
    struct DevRound { bool pending = false; uint64_t ns = 0, nev = 0, nseg = 0; };
    std::vector<DevRound> dev(G);
    // the caller's event array goes to the device as it is when it is page-locked (rawdtw_host_alloc) and one read group takes the whole round:
    // its reads' chunks ARE the segments, in order -- no copy into the mapper's own staging (a third of the host phase)
    const bool events_in_place = on_device && m->opt.device_chain && G == 1 && event_off[n_reads] > 0 && rawdtw_host_is_page_locked(events) == 1;
    auto device_begin = [&](const uint32_t gi) -> bool {
        Group &g = m->groups[gi];
        RoundArrays &ra = g.buf[g.cur];
        const size_t nr = ra.ks.size();
        if (nr == 0) return true; // (none of the round's reads is this group's)
        {
            m->pool->run(nr, 64, [&](size_t i) {
                const uint32_t k = ra.ks[i];
                MRead &rd = m->reads[read_ids[k]];
                RoundRead &r = rr[k];
                if (!host_phase_events(m, rd, r, events + event_off[k], event_off[k + 1] - event_off[k])) return;
                uint64_t n = hit_off[k + 1] - hit_off[k];
                for (const MChain &ch : rd.chains) n += ch.anchors.size(); // rmap.cpp:344-357: re-seeding with the previous chains' anchors
                r.n_seeds = n;
            });
            uint64_t ns = 0, nev = 0, nseg = 0;
            for (size_t i = 0; i < nr; i++) { RoundRead &r = rr[ra.ks[i]]; r.seed0 = ns; ns += r.n_seeds; r.ev0 = nev; nev += r.ne; nseg += r.ne ? 1 : 0; }
            g.hw_reads = std::max<uint64_t>(g.hw_reads, nr); g.hw_seeds = std::max(g.hw_seeds, ns); g.hw_events = std::max(g.hw_events, nev); g.hw_seg = std::max(g.hw_seg, nseg);
            g.hw_chains = std::max<uint64_t>(g.hw_chains, g.hw_reads * 32);
            bool ok = true;
            for (RoundArrays *x : {&ra, &g.buf[g.cur ^ 1]})
                ok = ok && x->seed_off.ensure(g.hw_reads + 1, true) && x->seeds.ensure(g.hw_seeds + 1, true) && x->read_base.ensure(std::max(g.hw_reads, g.hw_chains) + 1, true) &&
                     x->chain_off.ensure(g.hw_reads + 1, true) && x->anchor_off.ensure(g.hw_chains + 1, true) && x->recs.ensure(g.hw_chains + 1, true) &&
                     x->anchors.ensure(g.hw_seeds + 1, true) && x->score.ensure(g.hw_chains + 1, true) && x->keep.ensure(g.hw_chains + 1, true) &&
                     x->new_events.ensure((events_in_place ? 0 : g.hw_events) + 1, true) && x->seg_src.ensure(g.hw_seg + 2, true) &&
                     x->seg_dst.ensure(std::max(g.hw_seg, g.hw_reads) + 1, true);
            if (!ok) { set_fail(RAWDTW_ERR_OOM, "host allocation failed"); return true; }
            if (events_in_place) { // (every read a segment, empty ones too: event_off itself is the table of sources)
                for (size_t i = 0; i < nr; i++) ra.seg_dst[i] = m->reads[read_ids[ra.ks[i]]].slot * m->opt.slot_events + rr[ra.ks[i]].ev_before;
            } else {
                uint64_t sg = 0, at = 0;
                for (size_t i = 0; i < nr; i++) {
                    const RoundRead &r = rr[ra.ks[i]];
                    if (!r.ne) continue;
                    const MRead &rd = m->reads[read_ids[ra.ks[i]]];
                    ra.seg_src[sg] = at;
                    ra.seg_dst[sg] = (rd.slot / G) * m->opt.slot_events + r.ev_before;
                    at += r.ne; sg++;
                }
                ra.seg_src[sg] = at;
            }
            ra.seed_off[nr] = ns;
            m->pool->run(nr, 64, [&](size_t i) {
                const uint32_t k = ra.ks[i];
                const RoundRead &r = rr[k];
                const MRead &rd = m->reads[read_ids[k]];
                ra.seed_off[i] = r.seed0;
                ra.read_base[i] = (rd.slot / G) * m->opt.slot_events;
                rawdtw_seed_t *out = ra.seeds.p + r.seed0;
                if (r.n_seeds) {
                    for (const MChain &ch : rd.chains) {
                        const uint32_t key = ch.ref * 2u + (uint32_t)ch.strand;
                        for (const rawdtw_anchor_t &an : ch.anchors) *out++ = rawdtw_seed_t{key, an.target_position, an.query_position};
                    }
                    for (uint64_t h = hit_off[k]; h < hit_off[k + 1]; h++) // rmap.cpp:371-391
                        *out++ = rawdtw_seed_t{hits[h].ref_seq * 2u + (uint32_t)(hits[h].strand ? 1 : 0), hits[h].target_position, hits[h].query_position + r.chunk_start};
                }
                if (r.ne && !events_in_place) memcpy(ra.new_events.p + r.ev0, events + event_off[k], r.ne * sizeof(float));
            });
            double td = now_ms();
            m->timing[0] += td - t0; t0 = td;
            // the chaining first, the events behind it: the sort + DP does not read them, and rawdtw_chain_round_end waits for the round's own work only
            // -- the events' upload (the round's largest) runs on while the host goes on
            int st = rawdtw_chain_round_begin(g.ctx, &m->opt.chain, nr, ra.seed_off.p, ra.seeds.p, ra.read_base.p, (uint32_t)m->ref_off.size(), m->ref_off.data(),
                                              ra.chain_off.p, ra.anchor_off.p, ra.recs.p, g.hw_chains, ra.anchors.p);
            const bool declined = st == RAWDTW_ERR_UNSUPPORTED;
            if (st == RAWDTW_OK || declined) {
                int se = RAWDTW_OK;
                if (nseg && events_in_place) se = rawdtw_events_append(g.ctx, events, event_off[n_reads], (uint32_t)nr, event_off, ra.seg_dst.p);
                else if (nseg) se = rawdtw_events_append(g.ctx, ra.new_events.p, nev, (uint32_t)nseg, ra.seg_src.p, ra.seg_dst.p);
                if (se != RAWDTW_OK) {
                    if (st == RAWDTW_OK) { const rawdtw_anchor_t *x = nullptr; const uint64_t *y = nullptr; const uint32_t *z = nullptr; (void)rawdtw_chain_round_end(g.ctx, &x, &y, &z); }
                    st = se;
                }
            }
            td = now_ms();
            m->timing[2] += td - t0; t0 = td;
            if (declined && st == RAWDTW_ERR_UNSUPPORTED) return false;
            if (st != RAWDTW_OK) { set_fail(st, rawdtw_last_error(g.ctx)); return true; }
            dev[gi].pending = true; dev[gi].ns = ns; dev[gi].nev = nev; dev[gi].nseg = nseg;
            return true;
        }
    };
    // second half: the wait, the DTW submission straight from the device's arrays, and -- while that batch runs -- the round's chains per read, as
    // the host phase would have left them.  false: declined (a read with too many chains, or an order only std::sort knows)
    auto device_end = [&](const uint32_t gi) -> bool {
        Group &g = m->groups[gi];
        RoundArrays &ra = g.buf[g.cur];
        const size_t nr = ra.ks.size();
        dev[gi].pending = false;
        const rawdtw_anchor_t *d_anchors = nullptr;
        const uint64_t *d_ref_base = nullptr;
        const uint32_t *d_read_base = nullptr;
        int st = rawdtw_chain_round_end(g.ctx, &d_anchors, &d_ref_base, &d_read_base);
        if (st == RAWDTW_ERR_UNSUPPORTED) { double td = now_ms(); m->timing[2] += td - t0; t0 = td; return false; }
        if (st != RAWDTW_OK) { set_fail(st, rawdtw_last_error(g.ctx)); return true; }
        const uint64_t nc = ra.chain_off[nr], na = ra.anchor_off[nc];
        ra.n_chains = nc; ra.n_anchors = na; ra.n_new = 0; ra.n_new_events = dev[gi].nev; ra.n_seg = dev[gi].nseg;
        st = rawdtw_batch_submit_device(g.ctx, &m->opt.align, nr, ra.chain_off.p, ra.anchor_off.p, d_anchors, d_ref_base, d_read_base, &ra.batch);
        if (st != RAWDTW_OK) { set_fail(st, rawdtw_last_error(g.ctx)); return true; }
        ra.device_chained = true;
        m->timing[6] += (double)(dev[gi].nev * sizeof(float));
        m->timing[7] += (double)(dev[gi].ns * sizeof(rawdtw_seed_t) + (nr + 1) * 16 + nr * 4 + (nc + 1) * 8 + dev[gi].nseg * 12);
        double td = now_ms();
        m->timing[2] += td - t0; t0 = td;
        m->pool->run(nr, 32, [&](size_t i) {
            RoundRead &r = rr[ra.ks[i]];
            r.chain0 = ra.chain_off[i];
            const uint64_t n = ra.chain_off[i + 1] - ra.chain_off[i];
            r.chains.resize(n);
            for (uint64_t c = 0; c < n; c++) {
                const rawdtw_chain_rec_t &rec = ra.recs[r.chain0 + c];
                MChain &ch = r.chains[c];
                ch.chaining_score = rec.chaining_score; ch.ref = rec.key >> 1; ch.strand = (int32_t)(rec.key & 1u);
                ch.start_position = rec.start_position; ch.end_position = rec.end_position;
                const rawdtw_anchor_t *an = ra.anchors.p + ra.anchor_off[r.chain0 + c];
                ch.anchors.assign(an, an + rec.n_anchors);
            }
        });
        td = now_ms();
        m->timing[1] += td - t0; t0 = td;
        return true;
    };
    // ---- a group's round with the chains made on the host: host phase, lay-out, submit (`events_done`: a round the device declined to chain --
    // its events are appended already, on both sides) ----
    auto host_round = [&](const uint32_t gi, const RoundArrays *pv, const bool events_done) {
        Group &g = m->groups[gi];
        RoundArrays &ra = g.buf[g.cur];
        const RoundArrays &pb = g.buf[g.cur ^ 1];
        const size_t nr = ra.ks.size();
        if (events_done)
            m->pool->run(nr, 16, [&](size_t i) {
                const uint32_t k = ra.ks[i];
                RoundRead &r = rr[k];
                if (!r.skipped) host_phase_chain(m, m->reads[read_ids[k]], r, hits + hit_off[k], hit_off[k + 1] - hit_off[k], nullptr, runs_dtw);
            });
        else
            m->pool->run(nr, 16, [&](size_t i) {
                const uint32_t k = ra.ks[i];
                MRead &rd = m->reads[read_ids[k]];
                const bool in_prev = pv && rd.last_round == pv->round_id;
                host_phase_read(m, rd, rr[k], events + event_off[k], event_off[k + 1] - event_off[k], hits + hit_off[k], hit_off[k + 1] - hit_off[k],
                                in_prev ? pv : nullptr, runs_dtw);
            });
        double t1 = now_ms();
        m->timing[0] += t1 - t0; t0 = t1;
        for (size_t i = 0; i < nr; i++) if (rr[ra.ks[i]].err != RAWDTW_OK) set_fail(rr[ra.ks[i]].err, "chaining failed (chain output buffers too small)");
        if (status != RAWDTW_OK || !runs_dtw) return;
        // ---- lay-out: offsets by a running sum, then every read copies its own stretch ----
        uint64_t nc = 0, na = 0, nn = 0, nev = 0, nseg = 0;
        for (size_t i = 0; i < nr; i++) {
            RoundRead &r = rr[ra.ks[i]];
            r.chain0 = nc; r.anchor0 = na; r.new0 = nn; r.ev0 = nev;
            nc += r.chains.size();
            for (size_t c = 0; c < r.chains.size(); c++) {
                const uint64_t n = r.chains[c].anchors.size();
                na += n;
                nn += n - r.carry[c].parts; // (the new entries and, when a stretch is taken over, the junction)
            }
            nev += events_done ? 0 : r.ne;
            nseg += !events_done && r.ne ? 1 : 0;
        }
        ra.n_chains = nc; ra.n_anchors = na; ra.n_new = nn; ra.n_new_events = nev; ra.n_seg = nseg;
        const bool pin = on_device;
        g.hw_reads = std::max<uint64_t>(g.hw_reads, nr); g.hw_chains = std::max(g.hw_chains, nc); g.hw_anchors = std::max(g.hw_anchors, na);
        g.hw_new = std::max(g.hw_new, nn); g.hw_events = std::max(g.hw_events, nev); g.hw_seg = std::max(g.hw_seg, nseg);
        auto size_arrays = [&](RoundArrays &x, const bool kept) { // (`kept`: the round before's arrays, read again by the next round's matching)
            const size_t k_r = kept ? x.n_reads + 1 : 0, k_c = kept ? x.n_chains + 1 : 0, k_a = kept ? x.n_anchors + 1 : 0;
            return x.chain_off.ensure(g.hw_reads + 1, pin, k_r) && x.anchor_off.ensure(g.hw_chains + 1, pin, k_c) && x.ref_base.ensure(g.hw_chains + 1, pin, k_c) &&
                   x.read_base.ensure(g.hw_chains + 1, pin, k_c) && x.anchors.ensure(g.hw_anchors + 1, pin && !(m->opt.carry && g.has_prev), k_a) &&
                   x.score.ensure(g.hw_chains + 1, pin) && x.keep.ensure(g.hw_chains + 1, pin) &&
                   (!(on_device && m->opt.carry) || (x.new_off.ensure(g.hw_chains + 1, pin) && x.new_anchors.ensure(g.hw_new + 1, pin) && x.carry.ensure(g.hw_chains + 1, pin))) &&
                   (!on_device || (x.new_events.ensure(g.hw_events + 1, pin) && x.seg_src.ensure(g.hw_seg + 2, pin) && x.seg_dst.ensure(g.hw_seg + 1, pin)));
        };
        const bool ok = size_arrays(ra, false) && size_arrays(g.buf[g.cur ^ 1], true);
        if (!ok) { set_fail(RAWDTW_ERR_OOM, "host allocation failed"); return; }
        if (m->scorer) { ra.chain_seq.resize(nc); ra.chain_strand.resize(nc); }
        ra.chain_off[nr] = nc; ra.anchor_off[nc] = na;
        if (ra.carried) ra.new_off[nc] = nn;
        ra.ref_base[nc] = 0; ra.read_base[nc] = 0; // (non-null, initialised arrays for a round without chains)
        ra.anchors[na] = rawdtw_anchor_t{0, 0};
        {   // the new events' segments (reads with a chunk this round, in order)
            uint64_t s = 0, at = 0;
            for (size_t i = 0; i < nr && on_device && !events_done; i++) {
                const RoundRead &r = rr[ra.ks[i]];
                if (!r.ne) continue;
                const MRead &rd = m->reads[read_ids[ra.ks[i]]];
                ra.seg_src[s] = at;
                ra.seg_dst[s] = (rd.slot / G) * m->opt.slot_events + r.ev_before;
                at += r.ne; s++;
            }
            if (on_device) ra.seg_src[s] = at;
        }
        m->pool->run(nr, 32, [&](size_t i) {
            const uint32_t k = ra.ks[i];
            const RoundRead &r = rr[k];
            const MRead &rd = m->reads[read_ids[k]];
            ra.chain_off[i] = r.chain0;
            uint64_t at = r.anchor0, nat = r.new0;
            for (size_t c = 0; c < r.chains.size(); c++) {
                const std::vector<rawdtw_anchor_t> &an = r.chains[c].anchors;
                const uint64_t cc = r.chain0 + c;
                ra.anchor_off[cc] = at;
                ra.ref_base[cc] = r.ref_base[c];
                ra.read_base[cc] = (rd.slot / G) * m->opt.slot_events;
                if (m->scorer) { ra.chain_seq[cc] = r.chains[c].ref; ra.chain_strand[cc] = r.chains[c].strand; }
                memcpy(ra.anchors.p + at, an.data(), an.size() * sizeof(rawdtw_anchor_t));
                if (ra.carried) {
                    const uint64_t n_new = an.size() - r.carry[c].parts; // (with the junction)
                    ra.carry[cc] = r.carry[c];
                    ra.new_off[cc] = nat;
                    memcpy(ra.new_anchors.p + nat, an.data(), n_new * sizeof(rawdtw_anchor_t));
                    nat += n_new;
                }
                at += an.size();
            }
            if (on_device && r.ne && !events_done) memcpy(ra.new_events.p + r.ev0, events + event_off[k], r.ne * sizeof(float));
        });
        t1 = now_ms();
        m->timing[1] += t1 - t0; t0 = t1;
        // ---- submit: the DTW block of gen_chains for every read of the group (rmap.cpp:509-530), one device submission ----
        if (on_device) {
            int st = RAWDTW_OK;
            if (nseg) st = rawdtw_events_append(g.ctx, ra.new_events.p, nev, (uint32_t)nseg, ra.seg_src.p, ra.seg_dst.p);
            if (st == RAWDTW_OK) {
                if (ra.carried) {
                    st = rawdtw_batch_submit_carry(g.ctx, &m->opt.align, nr, ra.chain_off.p, ra.anchor_off.p, ra.anchors.p, ra.new_off.p, ra.new_anchors.p,
                                                   ra.ref_base.p, ra.read_base.p, pb.batch, ra.carry.p, &ra.batch);
                    if (st == RAWDTW_ERR_UNSUPPORTED) { ra.carried = false; st = RAWDTW_OK; } // (e.g. a round without a chain: nothing to plan on the device)
                }
                if (st == RAWDTW_OK && !ra.carried)
                    st = rawdtw_batch_submit(g.ctx, &m->opt.align, nr, ra.chain_off.p, ra.anchor_off.p, ra.anchors.p, ra.ref_base.p, ra.read_base.p, &ra.batch);
            }
            if (st != RAWDTW_OK) set_fail(st, rawdtw_last_error(g.ctx));
            m->timing[5] += (double)((ra.carried ? nn : na) * sizeof(rawdtw_anchor_t));
            m->timing[6] += (double)(nev * sizeof(float));
            m->timing[7] += (double)((nr + 1) * 8 + (nc + 1) * 8 + nc * 12 + (ra.carried ? nc * 32 + 8 : 0) + nseg * 12);
        } else {
            std::vector<const float *> evp(nr);
            std::vector<uint32_t> evn(nr);
            for (size_t i = 0; i < nr; i++) { const MRead &rd = m->reads[read_ids[ra.ks[i]]]; evp[i] = rd.events.data(); evn[i] = (uint32_t)rd.events.size(); }
            if (m->scorer(m->scorer_user, nr, ra.chain_off.p, ra.anchor_off.p, ra.anchors.p, ra.chain_seq.data(), ra.chain_strand.data(), evp.data(), evn.data(),
                          ra.score.p, ra.keep.p) != 0)
                set_fail(RAWDTW_ERR_DEVICE, "the external scorer failed");
        }
        t1 = now_ms();
        m->timing[2] += t1 - t0; t0 = t1;
    };
    for (uint32_t gi = 0; gi < G && status == RAWDTW_OK; gi++) {
        Group &g = m->groups[gi];
        RoundArrays &ra = g.buf[g.cur];
        const RoundArrays &pb = g.buf[g.cur ^ 1];
        const RoundArrays *pv = nullptr;
        const size_t nr = ra.ks.size();
        ra.carried = false;
        ra.round_id = round_id;
        ra.n_reads = nr;
        ra.device_chained = false;
        if (on_device && m->opt.device_chain) {
            if (!device_begin(gi) && status == RAWDTW_OK) host_round(gi, nullptr, true);
            continue;
        }
        if (on_device && m->opt.carry && g.has_prev && pb.batch && rawdtw_batch_can_carry(g.ctx, pb.batch, &m->opt.align)) {
            size_t known = 0; // (a round none of whose reads was in the round before has nothing to take over: submitted whole)
            for (size_t i = 0; i < nr && !known; i++) known += m->reads[read_ids[ra.ks[i]]].last_round == pb.round_id;
            if (known) { pv = &pb; ra.carried = true; }
        }
        host_round(gi, pv, false);
    }
    for (uint32_t gi = 0; gi < G; gi++) {
        if (!dev[gi].pending) continue;
        if (status != RAWDTW_OK) { // (a failure elsewhere: the round begun is ended, nothing of it is used)
            const rawdtw_anchor_t *x = nullptr; const uint64_t *y = nullptr; const uint32_t *z = nullptr;
            (void)rawdtw_chain_round_end(m->groups[gi].ctx, &x, &y, &z);
            dev[gi].pending = false;
            continue;
        }
        if (!device_end(gi) && status == RAWDTW_OK) host_round(gi, nullptr, true);
    }
    // ---- per group: fetch, then the round's end per read ----
    for (uint32_t gi = 0; gi < G; gi++) {
        Group &g = m->groups[gi];
        RoundArrays &ra = g.buf[g.cur];
        const size_t nr = ra.ks.size();
        if (on_device && ra.batch) {
            int st = rawdtw_batch_fetch(g.ctx, ra.batch, ra.score.p, ra.keep.p, nullptr); // (also after a failure elsewhere: the arrays it reads go out of use here)
            if (st != RAWDTW_OK) set_fail(st, rawdtw_last_error(g.ctx));
            uint64_t sc = 0, ru = 0;
            if (st == RAWDTW_OK && status == RAWDTW_OK && rawdtw_batch_round_stats(g.ctx, ra.batch, &sc, &ru) == RAWDTW_OK) { m->parts_scored += sc; m->parts_reused += ru; }
        }
        double t1 = now_ms();
        m->timing[3] += t1 - t0; t0 = t1;
        if (status != RAWDTW_OK) continue;
        const bool evaluate = (m->opt.flag & 0x2) != 0, log_scores = (m->opt.flag & 0x8) != 0;
        m->pool->run(nr, 16, [&](size_t i) {
            const uint32_t k = ra.ks[i];
            RoundRead &r = rr[k];
            MRead &rd = m->reads[read_ids[k]];
            if (r.skipped) { r.high = high_confidence(m, rd.chains); return; } // rmap.cpp:569-572: the chains stay as they were
            std::vector<MChain> post;
            post.reserve(r.chains.size());
            for (size_t c = 0; c < r.chains.size(); c++) {
                MChain &ch = r.chains[c];
                bool keep = true;
                if (runs_dtw) {
                    ch.alignment_score = ra.score[r.chain0 + c];
                    keep = ra.keep[r.chain0 + c] != 0;
                    // --dtw-log-scores (rmap.cpp:308-312): in evaluation order; a cut chain returns before the fprintf
                    if (log_scores && ch.alignment_score != -1e10f) {
                        char line[128];
                        snprintf(line, sizeof line, "chaining_score=%f alignment_score=%f\n", (double)ch.chaining_score, (double)ch.alignment_score);
                        r.log += line;
                    }
                }
                if (!evaluate || !runs_dtw || keep) post.push_back(std::move(ch)); // rmap.cpp:525: replaced only under EVALUATE_CHAINS
            }
            rd.chains = primary_chains(m, post);
            r.high = high_confidence(m, rd.chains);
        });
        t1 = now_ms();
        m->timing[4] += t1 - t0; t0 = t1;
    }
    if (status != RAWDTW_OK) { // put the reads back as they were; nothing of the round stays
        for (uint32_t k = 0; k < n_reads; k++) {
            MRead &rd = m->reads[read_ids[k]];
            if (rd.n_events >= rr[k].ev_before && rr[k].ne + rr[k].ev_before == rd.n_events) {
                rd.n_events = rr[k].ev_before; rd.offset = rr[k].off_before;
                if (rd.events.size() > rd.n_events) rd.events.resize(rd.n_events);
            }
        }
        for (Group &g : m->groups) {
            RoundArrays &ra = g.buf[g.cur];
            if (ra.batch) { rawdtw_batch_destroy(ra.batch); ra.batch = nullptr; }
            g.cur ^= 1; // (the round before stays the round before)
        }
        return fail(m, status, status_msg);
    }
    // ---- commit ----
    m->rounds = round_id;
    for (uint32_t gi = 0; gi < G; gi++) {
        Group &g = m->groups[gi];
        RoundArrays &ra = g.buf[g.cur], &pb = g.buf[g.cur ^ 1];
        if (pb.batch) { rawdtw_batch_destroy(pb.batch); pb.batch = nullptr; }
        g.has_prev = on_device && m->opt.carry && !m->opt.device_chain && ra.batch != nullptr;
        if (!g.has_prev && ra.batch) { rawdtw_batch_destroy(ra.batch); ra.batch = nullptr; }
        for (size_t i = 0; i < ra.ks.size(); i++) {
            MRead &rd = m->reads[read_ids[ra.ks[i]]];
            rd.last_round = round_id; rd.last_pos = i;
        }
    }
    for (uint32_t k = 0; k < n_reads; k++) {
        MRead &rd = m->reads[read_ids[k]];
        if (!rr[k].log.empty()) m->log += rr[k].log;
        rd.chunks_done++;
        if (rr[k].high) { rd.finished = true; rd.broke_early = true; } // rmap.cpp:692 (evaluated with the round's end, per read on the pool)
        else if (rd.chunks_done >= std::min(rd.n_chunks, m->opt.max_num_chunk)) rd.finished = true;
    }
    // (the round's per-read state goes on the pool: ten vectors a read, freed one read after the other they were milliseconds of a large round)
    m->pool->run(n_reads, 64, [&](size_t k) { RoundRead gone; std::swap(gone, rr[k]); });
    m->timing[4] += now_ms() - t0;
    return RAWDTW_OK;
}

// --dtw-output-cigar (rmap.cpp:715-717): the best chain of every mapped read through DTW_global_tb once more, its path as the
// aln:s: string with the reference's two quirks (rmap.cpp:230-233, 283-289).  All reads' jobs go to the device in ONE call;
// the paths come back as steps and distances (rawdtw_traceback_batch_steps: 5 bytes an element) and every read's string is
// written on the pool while the steps are walked.
int rawdtw_mapper_finish(rawdtw_mapper *m)
{
    if (!m) return RAWDTW_ERR_INVALID;
    if (!(m->opt.flag & 0x4)) return RAWDTW_OK;
    if (!m->ctx) return fail(m, RAWDTW_ERR_NO_DEVICE, "--dtw-output-cigar needs a device context");
    drop_batches(m); // (the traceback call replaces the event arena's contents: the mapper's rounds are over)
    for (Group &g : m->groups) g.has_prev = false;
    struct Item { uint32_t read; uint64_t job0; uint32_t nj; uint64_t ev0; };
    std::vector<Item> items;
    std::vector<rawdtw_job_t> jobs;
    std::vector<float> events;
    for (uint32_t r = 0; r < m->reads.size(); r++) {
        MRead &rd = m->reads[r];
        if (rd.released || !high_confidence(m, rd.chains)) continue;
        MChain &ch = rd.chains[0];
        const uint32_t na = (uint32_t)ch.anchors.size();
        const uint32_t nj = rawdtw_chain_job_count(&m->opt.align, na);
        if ((uint64_t)events.size() + rd.events.size() >= (1ull << 32)) return fail(m, RAWDTW_ERR_RANGE, "the mapped reads' events exceed one event arena");
        const uint64_t j0 = jobs.size();
        jobs.resize(j0 + std::max<uint32_t>(nj, 1));
        const uint64_t rb = m->ref_off[ch.ref * 2u + (uint32_t)ch.strand];
        const int st = rawdtw_chain_build_jobs(&m->opt.align, ch.anchors.data(), na, rb, (uint32_t)events.size(), 1, jobs.data() + j0);
        if (st != RAWDTW_OK) return fail(m, st, st == RAWDTW_ERR_UNSUPPORTED ? "banded global alignment with --dtw-output-cigar is not implemented (rmap.cpp:223-225)" : "job building failed");
        jobs.resize(j0 + nj);
        items.push_back(Item{r, j0, nj, events.size()});
        events.insert(events.end(), rd.events.begin(), rd.events.end());
    }
    if (items.empty()) return RAWDTW_OK;
    const uint64_t nj_all = jobs.size();
    std::vector<uint64_t> poff(nj_all + 1, 0);
    for (uint64_t k = 0; k < nj_all; k++) poff[k + 1] = poff[k] + jobs[k].n + jobs[k].m - 1;
    std::vector<uint32_t> plen(nj_all);
    std::vector<uint8_t> step(poff[nj_all] + 1);
    std::vector<float> pd(poff[nj_all] + 1), cost(nj_all);
    if (nj_all) {
        const int st = rawdtw_traceback_batch_steps(m->ctx, jobs.data(), nj_all, events.data(), events.size(), cost.data(), poff.data(), plen.data(), step.data(), pd.data());
        if (st != RAWDTW_OK) return fail(m, st, rawdtw_last_error(m->ctx));
    }
    std::vector<std::string> logs(items.size());
    m->pool->run(items.size(), 4, [&](size_t x) {
        const Item &it = items[x];
        MRead &rd = m->reads[it.read];
        MChain &ch = rd.chains[0];
        const uint32_t na = (uint32_t)ch.anchors.size(), nj = it.nj;
        ch.alignment_score = rawdtw_chain_replay(&m->opt.align, ch.anchors.data(), na, cost.data() + it.job0, -1e10f); // rmap.cpp:306 on the summed costs
        std::string s;
        const uint32_t parts = na - 1;
        char el[96];
        for (uint32_t k = 0; k < nj; k++) {
            // sparse: every element offset by its part's start anchor (rmap.cpp:286-289); global: the offsets are added to
            // alignment.back() once per element (rmap.cpp:230-233), i.e. only the last tuple moves
            const rawdtw_anchor_t &s0 = m->opt.align.border_constraint == 0 ? ch.anchors[na - 1] : ch.anchors[parts - k];
            const uint64_t p0 = poff[it.job0 + k];
            const uint32_t len = plen[it.job0 + k];
            unsigned long long pi = 0, pj = 0; // (i, j) from the steps: a global path starts at (0, 0)
            for (uint32_t q = 0; q < len; q++) {
                pi += step[p0 + q] & 1u; pj += step[p0 + q] >> 1;
                unsigned long long i = pi, j = pj;
                if (m->opt.align.border_constraint != 0) { i += s0.query_position; j += s0.target_position; }
                else if (q + 1 == len) { i += (unsigned long long)len * s0.query_position; j += (unsigned long long)len * s0.target_position; }
                snprintf(el, sizeof el, "(%llu,%llu,%g)", i, j, (double)pd[p0 + q]); // ostream << float == %g (rmap.cpp:580-592)
                s += el;
            }
        }
        ch.aln = std::move(s);
        ch.has_aln = true;
        if (m->opt.flag & 0x8) {
            char line[128];
            snprintf(line, sizeof line, "chaining_score=%f alignment_score=%f\n", (double)ch.chaining_score, (double)ch.alignment_score);
            logs[x] = line;
        }
    });
    for (const std::string &l : logs) m->log += l; // (read order)
    return RAWDTW_OK;
}

// The PAF line of one read (rmap.cpp:696-801 for the fields and tags, 956-965 for the format).  `mt:f:` is wall-clock in the
// reference and therefore written as 0 here.
int rawdtw_mapper_paf(const rawdtw_mapper *m, uint32_t read_id, char *buf, uint32_t cap, uint32_t *len)
{
    if (!m || read_id >= m->reads.size() || !len || m->reads[read_id].released) return RAWDTW_ERR_INVALID;
    const MRead &rd = m->reads[read_id];
    const uint32_t l_chunk = m->opt.chunk_size, max_chunk = m->opt.max_num_chunk;
    uint32_t current_chunk = rd.broke_early ? rd.chunks_done - 1 : rd.chunks_done; // the loop's current_chunk when it exits
    const uint64_t chunk_start = (uint64_t)current_chunk * l_chunk;
    // rmap.cpp:696: step back one chunk when the loop ran out of signal or chunks rather than breaking
    if (!rd.broke_early && current_chunk > 0 && (chunk_start >= rd.qlen || current_chunk == max_chunk)) current_chunk -= 1;
    const uint32_t offset = rd.offset; // reg0->offset: the events of the chunks that were chained (rmap.cpp:574-575)
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
