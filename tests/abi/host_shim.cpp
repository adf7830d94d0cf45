// host_shim.cpp -- a C++11 consumer of include/rawdtw.h, written the way a RawAlign maintainer would bind
// librawdtw.so inside src/rmap.cpp (INTEGRATION.md sections 2 and 4).  Test infrastructure: built and run by
// tests/test_abi_shim.py, which compares what it prints with the oracle's results.
//
// It mirrors, for one mini-batch read from a file the test wrote:
//   * index hand-over                                   main.cpp:354 / rawindex.h:32-34  -> rawdtw_upload_reference
//   * the DTW block of gen_chains for every read        rmap.cpp:509-530                 -> sort, rawdtw_batch_create / run / fetch, filter
//   * gen_primary_chains, comp_mapq, the stop rule      rmap.cpp:532-536, 594-665        -> rawdtw_gen_primary_chains, rawdtw_is_mapped_...
//   * --dtw-output-cigar for the best chain             rmap.cpp:715-717, 741-744        -> rawdtw_chain_build_jobs(cigar) + rawdtw_traceback_batch
//   * the tag text                                      rmap.cpp:580-592 (ostream << float), std::to_string
//
// Build: g++ -std=c++11 -I<repo>/include host_shim.cpp -L<repo>/rawalign_amd -lrawdtw
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <sstream>
#include <string>
#include <vector>

#include "rawdtw.h"

namespace {

struct Blob {
    std::vector<std::vector<float>> fwd, rev;
    std::vector<float> events;
    uint64_t n_reads = 0;
    std::vector<uint64_t> chain_off, anchor_off; // generation order, per read contiguous
    std::vector<rawdtw_anchor_t> anchors;
    std::vector<uint32_t> chain_seq;
    std::vector<int32_t> chain_strand;
    std::vector<float> chaining_score;
    std::vector<uint32_t> read_base; // per read: offset of its events
    rawdtw_align_opt_t opt;
    int32_t flag = 0;
};

template <typename T> bool rd(FILE *f, T *p, size_t n) { return n == 0 || fread(p, sizeof(T), n, f) == n; }

bool load(const char *path, Blob &b)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint32_t magic = 0, n_seq = 0;
    bool ok = rd(f, &magic, 1) && magic == 0x52445457u && rd(f, &n_seq, 1);
    b.fwd.resize(n_seq); b.rev.resize(n_seq);
    for (uint32_t s = 0; ok && s < n_seq; s++) {
        uint32_t len = 0;
        ok = rd(f, &len, 1);
        b.fwd[s].resize(len); b.rev[s].resize(len);
        ok = ok && rd(f, b.fwd[s].data(), len) && rd(f, b.rev[s].data(), len);
    }
    uint64_t n_ev = 0, n_chains = 0;
    ok = ok && rd(f, &n_ev, 1);
    b.events.resize(n_ev);
    ok = ok && rd(f, b.events.data(), n_ev) && rd(f, &b.n_reads, 1);
    b.chain_off.resize(b.n_reads + 1);
    ok = ok && rd(f, b.chain_off.data(), b.n_reads + 1);
    n_chains = ok ? b.chain_off[b.n_reads] : 0;
    b.anchor_off.resize(n_chains + 1);
    ok = ok && rd(f, b.anchor_off.data(), n_chains + 1);
    b.anchors.resize(ok ? b.anchor_off[n_chains] : 0);
    b.chain_seq.resize(n_chains); b.chain_strand.resize(n_chains); b.chaining_score.resize(n_chains);
    b.read_base.resize(b.n_reads);
    ok = ok && rd(f, b.anchors.data(), b.anchors.size()) && rd(f, b.chain_seq.data(), n_chains) &&
         rd(f, b.chain_strand.data(), n_chains) && rd(f, b.chaining_score.data(), n_chains) &&
         rd(f, b.read_base.data(), b.n_reads);
    int32_t o[2]; float fo[3];
    ok = ok && rd(f, o, 2) && rd(f, fo, 3) && rd(f, &b.flag, 1);
    b.opt.border_constraint = o[0]; b.opt.fill_method = o[1];
    b.opt.band_radius_frac = fo[0]; b.opt.match_bonus = fo[1]; b.opt.min_score = fo[2];
    b.opt.fused_score = 1;
    fclose(f);
    return ok;
}

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int st_ = (call);                                                                             \
        if (st_ != RAWDTW_OK) {                                                                       \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, st_, rawdtw_status_string(st_), rawdtw_last_error(dtw)); \
            return 2;                                                                                 \
        }                                                                                             \
    } while (0)

unsigned bits(float x) { unsigned u; memcpy(&u, &x, 4); return u; }

} // namespace

// ---- `host_shim map.bin --map`: the chunk-round mapping loop through rawdtw_mapper_* (INTEGRATION.md section 5): what
// map_worker_for / ri_map_frag / gen_chains do for a mini-batch of reads (rmap.cpp:667-822), one device submission per chunk
// round, PAF lines out.  The blob holds what stays in RawAlign: every chunk's events (revent.c) and seed hits (rsketch.c,
// rawindex.cpp). ----
static int run_map(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", path); return 1; }
    uint32_t magic = 0, n_seq = 0;
    bool ok = rd(f, &magic, 1) && magic == 0x504D4452u && rd(f, &n_seq, 1);
    std::vector<std::vector<float>> fwd(n_seq), rev(n_seq);
    std::vector<uint32_t> len(n_seq);
    for (uint32_t s = 0; ok && s < n_seq; s++) {
        ok = rd(f, &len[s], 1);
        fwd[s].resize(len[s]); rev[s].resize(len[s]);
        ok = ok && rd(f, fwd[s].data(), len[s]) && rd(f, rev[s].data(), len[s]);
    }
    int32_t io[2], flag = 0, carry = 0, min_chain_anchor = 2; float fo[3], ratio[2];
    uint32_t n_reads = 0;
    ok = ok && rd(f, io, 2) && rd(f, fo, 3) && rd(f, &flag, 1) && rd(f, &carry, 1) && rd(f, ratio, 2) && rd(f, &min_chain_anchor, 1) && rd(f, &n_reads, 1);
    struct Chunk { std::vector<float> ev; std::vector<rawdtw_seed_hit_t> hits; };
    struct Rd { uint32_t qlen, n_chunks; std::vector<Chunk> chunks; };
    std::vector<Rd> reads(n_reads);
    uint32_t slot = 8;
    for (uint32_t r = 0; ok && r < n_reads; r++) {
        ok = rd(f, &reads[r].qlen, 1) && rd(f, &reads[r].n_chunks, 1);
        reads[r].chunks.resize(reads[r].n_chunks);
        uint32_t total = 0;
        for (uint32_t c = 0; ok && c < reads[r].n_chunks; c++) {
            uint32_t ne = 0, nh = 0;
            ok = rd(f, &ne, 1);
            reads[r].chunks[c].ev.resize(ne);
            ok = ok && rd(f, reads[r].chunks[c].ev.data(), ne) && rd(f, &nh, 1);
            reads[r].chunks[c].hits.resize(nh);
            ok = ok && rd(f, reads[r].chunks[c].hits.data(), nh);
            total += ne;
        }
        slot = total + 8 > slot ? total + 8 : slot;
    }
    fclose(f);
    if (!ok) { fprintf(stderr, "malformed %s\n", path); return 1; }
    rawdtw_ctx *dtw = nullptr;
    if (rawdtw_create(0, &dtw) != RAWDTW_OK) { fprintf(stderr, "no device\n"); return 3; }
    std::vector<const float *> pf(n_seq), pr(n_seq);
    std::vector<std::string> names(n_seq);
    std::vector<const char *> pn(n_seq);
    for (uint32_t s = 0; s < n_seq; s++) { pf[s] = fwd[s].data(); pr[s] = rev[s].data(); names[s] = "seq" + std::to_string(s); pn[s] = names[s].c_str(); }
    CHECK(rawdtw_upload_reference(dtw, n_seq, pf.data(), pr.data(), len.data()));
    rawdtw_mapper_opt_t mo;
    memset(&mo, 0, sizeof mo);
    mo.flag = flag;
    mo.align.border_constraint = io[0]; mo.align.fill_method = io[1]; mo.align.band_radius_frac = fo[0]; mo.align.match_bonus = fo[1];
    mo.align.min_score = fo[2]; mo.align.fused_score = 1;
    mo.chain.max_gap_length = 2000; mo.chain.max_target_gap_length = 5000; mo.chain.chaining_band_length = 5000; mo.chain.max_num_skips = 25;
    mo.chain.min_num_anchors = 2; mo.chain.num_best_chains = 3; mo.chain.min_chaining_score = 10.0f; mo.chain.e = 6; // roptions.c:13-19, main.cpp:138
    mo.min_bestmap_ratio = ratio[0]; mo.min_meanmap_ratio = ratio[1]; mo.min_chain_anchor = min_chain_anchor;        // roptions.c:25-31 (1.2, 5, 2)
    mo.bp_per_sec = 450; mo.sample_rate = 4000; mo.chunk_size = 4000; mo.max_num_chunk = 30;                          // roptions.c:9-11, 24
    mo.slot_events = slot; mo.max_reads = n_reads ? n_reads : 1; mo.carry = carry;
    // (host threads, read groups and --min-events of the run: roptions.c:23, rmap.cpp:916,1033)
    if (const char *e = getenv("RAWDTW_SHIM_THREADS")) mo.threads = atoi(e);
    if (const char *e = getenv("RAWDTW_SHIM_GROUPS")) mo.groups = atoi(e);
    mo.min_events = 50;
    if (const char *e = getenv("RAWDTW_SHIM_MIN_EVENTS")) mo.min_events = (uint32_t)atoi(e);
    if (const char *e = getenv("RAWDTW_SHIM_DEVICE_CHAIN")) mo.device_chain = atoi(e); // (the anchor sort and the chaining DP on the device: rawdtw_chain_round)
    rawdtw_mapper *mp = nullptr;
    CHECK(rawdtw_mapper_create(dtw, &mo, n_seq, pn.data(), len.data(), &mp));
    std::vector<uint32_t> ids(n_reads);
    for (uint32_t r = 0; r < n_reads; r++) {
        const std::string nm = "read_" + std::to_string(r);
        CHECK(rawdtw_mapper_add_read(mp, nm.c_str(), reads[r].qlen, reads[r].n_chunks, &ids[r]));
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) { // chunk rounds until every read has stopped
        std::vector<uint32_t> act;
        std::vector<uint64_t> eoff{0}, hoff{0};
        std::vector<float> ev;
        std::vector<rawdtw_seed_hit_t> hits;
        for (uint32_t r = 0; r < n_reads; r++) {
            int fin = 0; uint32_t done = 0;
            CHECK(rawdtw_mapper_read_state(mp, ids[r], &fin, &done));
            if (fin || done >= reads[r].n_chunks) continue;
            const Chunk &c = reads[r].chunks[done];
            act.push_back(ids[r]);
            ev.insert(ev.end(), c.ev.begin(), c.ev.end()); eoff.push_back(ev.size());
            hits.insert(hits.end(), c.hits.begin(), c.hits.end()); hoff.push_back(hits.size());
        }
        if (act.empty()) break;
        const int st = rawdtw_mapper_round(mp, (uint32_t)act.size(), act.data(), eoff.data(), ev.data(), hoff.data(), hits.data());
        if (st != RAWDTW_OK) { fprintf(stderr, "round -> %d: %s\n", st, rawdtw_mapper_last_error(mp)); return 2; }
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    {
        const int st = rawdtw_mapper_finish(mp);
        if (st == RAWDTW_ERR_UNSUPPORTED) { printf("cigar=unsupported\n"); return 0; } // rmap.cpp:223-225 assert(false)
        if (st != RAWDTW_OK) { fprintf(stderr, "finish -> %d: %s\n", st, rawdtw_mapper_last_error(mp)); return 2; }
    }
    std::vector<char> buf(1 << 16);
    for (uint32_t r = 0; r < n_reads; r++) {
        uint32_t n = 0;
        int st = rawdtw_mapper_paf(mp, ids[r], buf.data(), (uint32_t)buf.size(), &n);
        if (st == RAWDTW_ERR_RANGE) { buf.resize((size_t)n + 1); st = rawdtw_mapper_paf(mp, ids[r], buf.data(), (uint32_t)buf.size(), &n); }
        if (st != RAWDTW_OK) return 2;
        printf("%s\n", buf.data());
    }
    const char *log = nullptr;
    rawdtw_mapper_log(mp, &log);
    for (const char *p = log; p && *p;) { const char *e = strchr(p, '\n'); printf("log %.*s\n", (int)(e ? e - p : (long)strlen(p)), p); p = e ? e + 1 : p + strlen(p); }
    uint64_t rounds = 0, scored = 0, reused = 0;
    rawdtw_mapper_stats(mp, &rounds, &scored, &reused);
    printf("map rounds=%llu parts_scored=%llu parts_reused=%llu rounds_per_s=%.1f\n", (unsigned long long)rounds, (unsigned long long)scored,
           (unsigned long long)reused, rounds / secs);
    rawdtw_mapper_destroy(mp);
    rawdtw_destroy(dtw);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[2], "--map")) return run_map(argv[1]);
    if (argc < 2) { fprintf(stderr, "usage: host_shim batch.bin [--pipeline steps | --teardown]\n"); return 1; }
    const int pipeline_steps = (argc >= 4 && !strcmp(argv[2], "--pipeline")) ? atoi(argv[3]) : 0;
    const bool teardown = argc >= 3 && !strcmp(argv[2], "--teardown");
    Blob b;
    if (!load(argv[1], b)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    const bool evaluate = b.flag & 0x2, cigar = b.flag & 0x4, log_scores = b.flag & 0x8; // roptions.h:13-15

    // ---- once per process ----
    rawdtw_ctx *dtw = nullptr;
    if (rawdtw_create(0, &dtw) != RAWDTW_OK) { fprintf(stderr, "no device\n"); return 3; }
    const uint32_t n_seq = (uint32_t)b.fwd.size();
    std::vector<const float *> fwd(n_seq), rev(n_seq);
    std::vector<uint32_t> len(n_seq);
    for (uint32_t s = 0; s < n_seq; s++) { fwd[s] = b.fwd[s].data(); rev[s] = b.rev[s].data(); len[s] = (uint32_t)b.fwd[s].size(); }
    CHECK(rawdtw_upload_reference(dtw, n_seq, fwd.data(), rev.data(), len.data()));

    // ---- one chunk round: chains of every read in evaluation order (rmap.cpp:512) ----
    const uint64_t n_chains = b.chain_off[b.n_reads];
    std::vector<uint64_t> order(n_chains); // evaluation position -> chain in generation order
    std::vector<uint32_t> perm;
    for (uint64_t r = 0; r < b.n_reads; r++) {
        const uint64_t c0 = b.chain_off[r], nc = b.chain_off[r + 1] - c0;
        perm.resize(nc);
        CHECK(rawdtw_sort_by_chaining_score(b.chaining_score.data() + c0, (uint32_t)nc, perm.data()));
        for (uint64_t k = 0; k < nc; k++) order[c0 + k] = c0 + perm[k];
    }
    std::vector<uint64_t> anchor_off(n_chains + 1, 0), ref_base(n_chains);
    std::vector<uint32_t> read_base(n_chains);
    std::vector<rawdtw_anchor_t> anchors;
    anchors.reserve(b.anchors.size());
    for (uint64_t r = 0; r < b.n_reads; r++)
        for (uint64_t e = b.chain_off[r]; e < b.chain_off[r + 1]; e++) {
            const uint64_t c = order[e];
            anchors.insert(anchors.end(), b.anchors.begin() + b.anchor_off[c], b.anchors.begin() + b.anchor_off[c + 1]);
            anchor_off[e + 1] = anchors.size();
            CHECK(rawdtw_reference_offset(dtw, b.chain_seq[c], b.chain_strand[c], &ref_base[e]));
            read_base[e] = b.read_base[r];
        }
    std::vector<float> score(n_chains, 0.0f);
    std::vector<uint8_t> keep(n_chains, 0);
    if (evaluate || log_scores) { // rmap.cpp:509
        CHECK(rawdtw_upload_events(dtw, b.events.data(), b.events.size()));
        rawdtw_batch *batch = nullptr;
        CHECK(rawdtw_batch_create(dtw, &b.opt, b.n_reads, b.chain_off.data(), anchor_off.data(), anchors.data(), ref_base.data(),
                                  read_base.data(), &batch));
        CHECK(rawdtw_batch_run(dtw, batch));
        CHECK(rawdtw_batch_fetch(dtw, batch, score.data(), keep.data(), nullptr));
        CHECK(rawdtw_batch_destroy(batch));
    }

    // ---- the same round as a pipeline ("The calls are asynchronous"): one context per pipeline
    // worker sharing the resident reference, the batch's arrays in page-locked memory, create + run enqueued for step k
    // while the steps before it are still on the device, fetch only when a worker's slot comes round again ----
    if (pipeline_steps > 0 && (evaluate || log_scores)) {
        const int W = 4;
        rawdtw_ctx *wctx[W] = {nullptr, nullptr, nullptr, nullptr};
        rawdtw_batch *wb[W] = {nullptr, nullptr, nullptr, nullptr};
        void *pin[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        const size_t bytes[5] = {(b.n_reads + 1) * 8, (n_chains + 1) * 8, anchors.size() * sizeof(rawdtw_anchor_t), n_chains * 8, n_chains * 4};
        const void *src[5] = {b.chain_off.data(), anchor_off.data(), anchors.data(), ref_base.data(), read_base.data()};
        for (int k = 0; k < 5; k++) {
            CHECK(rawdtw_host_alloc(bytes[k] ? bytes[k] : 8, &pin[k]));
            memcpy(pin[k], src[k], bytes[k]);
        }
        for (int w = 0; w < W; w++) {
            if (rawdtw_create(0, &wctx[w]) != RAWDTW_OK) { fprintf(stderr, "no device\n"); return 3; }
            CHECK(rawdtw_share_reference(wctx[w], dtw));
            CHECK(rawdtw_upload_events(wctx[w], b.events.data(), b.events.size()));
        }
        std::vector<float> s2(n_chains);
        std::vector<uint8_t> k2(n_chains);
        int bad = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int step = 0; step < pipeline_steps + W; step++) {
            const int w = step % W;
            if (wb[w]) {
                if (rawdtw_batch_fetch(wctx[w], wb[w], s2.data(), k2.data(), nullptr) != RAWDTW_OK) { fprintf(stderr, "%s\n", rawdtw_last_error(wctx[w])); return 2; }
                rawdtw_batch_destroy(wb[w]);
                wb[w] = nullptr;
                if (memcmp(s2.data(), score.data(), n_chains * 4) || memcmp(k2.data(), keep.data(), n_chains)) bad++;
            }
            if (step < pipeline_steps) {
                if (rawdtw_batch_create(wctx[w], &b.opt, b.n_reads, (const uint64_t *)pin[0], (const uint64_t *)pin[1], (const rawdtw_anchor_t *)pin[2],
                                        (const uint64_t *)pin[3], (const uint32_t *)pin[4], &wb[w]) != RAWDTW_OK ||
                    rawdtw_batch_run(wctx[w], wb[w]) != RAWDTW_OK) { fprintf(stderr, "%s\n", rawdtw_last_error(wctx[w])); return 2; }
            }
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("pipeline steps=%d mismatches=%d ms_per_step=%.4f\n", pipeline_steps, bad, ms / pipeline_steps);
        for (int w = 0; w < W; w++) rawdtw_destroy(wctx[w]);
        for (int k = 0; k < 5; k++) rawdtw_host_free(pin[k]);
    }

    // ---- teardown in the "wrong" order (include/rawdtw.h, lifetime): a context destroyed before its batch and its plan,
    // the owner of a shared reference before the sharer, an event arena re-uploaded between create and run.  A C++ host
    // tears down by scope exit, not by the library's preferred order: none of it may touch freed memory. ----
    if (teardown && (evaluate || log_scores)) {
        rawdtw_ctx *own = nullptr, *shr = nullptr;
        if (rawdtw_create(0, &own) != RAWDTW_OK || rawdtw_create(0, &shr) != RAWDTW_OK) { fprintf(stderr, "no device\n"); return 3; }
        int bad = 0;
        std::vector<float> s2(n_chains);
        std::vector<uint8_t> k2(n_chains);
        auto same = [&]() { return !memcmp(s2.data(), score.data(), n_chains * 4) && !memcmp(k2.data(), keep.data(), n_chains); };
#define TD(call) do { int st_ = (call); if (st_ != RAWDTW_OK) { fprintf(stderr, "%s -> %d\n", #call, st_); return 2; } } while (0)
        TD(rawdtw_upload_reference(own, n_seq, fwd.data(), rev.data(), len.data()));
        TD(rawdtw_share_reference(shr, own));
        TD(rawdtw_upload_events(own, b.events.data(), b.events.size()));
        TD(rawdtw_upload_events(shr, b.events.data(), b.events.size()));
        rawdtw_batch *b1 = nullptr, *b2 = nullptr, *b3 = nullptr;
        TD(rawdtw_batch_create(own, &b.opt, b.n_reads, b.chain_off.data(), anchor_off.data(), anchors.data(), ref_base.data(), read_base.data(), &b1));
        TD(rawdtw_batch_run(own, b1));
        TD(rawdtw_batch_fetch(own, b1, s2.data(), k2.data(), nullptr));
        bad += !same();
        TD(rawdtw_batch_create(shr, &b.opt, b.n_reads, b.chain_off.data(), anchor_off.data(), anchors.data(), ref_base.data(), read_base.data(), &b2));
        std::vector<uint64_t> job_off(n_chains + 1);
        uint64_t nj = 0;
        TD(rawdtw_batch_build_jobs(&b.opt, n_chains, anchor_off.data(), anchors.data(), ref_base.data(), read_base.data(), job_off.data(), nullptr, 0, &nj));
        std::vector<rawdtw_job_t> jobs(nj);
        TD(rawdtw_batch_build_jobs(&b.opt, n_chains, anchor_off.data(), anchors.data(), ref_base.data(), read_base.data(), job_off.data(), jobs.data(), nj, &nj));
        rawdtw_plan *pl = nullptr;
        TD(rawdtw_plan_create(own, jobs.data(), nj, &pl));
        TD(rawdtw_destroy(own)); // the owner goes first: b1 and pl are detached, the arena lives on for `shr`
        rawdtw_plan_info_t info;
        if (rawdtw_batch_info(b1, &info, nullptr) == RAWDTW_OK) bad++; // a detached batch is refused, not dereferenced
        TD(rawdtw_plan_info(pl, &info));                                // (a plan's host record still answers)
        bad += info.n_jobs != nj;
        TD(rawdtw_batch_run(shr, b2));
        TD(rawdtw_batch_fetch(shr, b2, s2.data(), k2.data(), nullptr));
        bad += !same();
        // a batch created before its context's event arena is re-uploaded into a new, larger allocation
        TD(rawdtw_batch_create(shr, &b.opt, b.n_reads, b.chain_off.data(), anchor_off.data(), anchors.data(), ref_base.data(), read_base.data(), &b3));
        std::vector<float> grown(b.events);
        grown.resize(2 * b.events.size() + 4096, 0.0f);
        TD(rawdtw_upload_events(shr, grown.data(), grown.size()));
        TD(rawdtw_batch_run(shr, b3));
        TD(rawdtw_batch_fetch(shr, b3, s2.data(), k2.data(), nullptr));
        bad += !same();
        // ... and one whose arena shrank under it is refused
        rawdtw_batch *b4 = nullptr;
        TD(rawdtw_batch_create(shr, &b.opt, b.n_reads, b.chain_off.data(), anchor_off.data(), anchors.data(), ref_base.data(), read_base.data(), &b4));
        TD(rawdtw_upload_events(shr, b.events.data(), b.events.size() / 2));
        if (rawdtw_batch_run(shr, b4) != RAWDTW_ERR_INVALID) bad++;
        TD(rawdtw_destroy(shr)); // again the context first ...
        TD(rawdtw_batch_destroy(b4));
        TD(rawdtw_batch_destroy(b3));
        TD(rawdtw_batch_destroy(b2));
        TD(rawdtw_batch_destroy(b1));
        TD(rawdtw_plan_destroy(pl));
#undef TD
        printf("teardown mismatches=%d\n", bad);
    }

    // ---- per read: post_alignment_chains, primary chains, MAPQ, stop rule, tags ----
    rawdtw_select_opt_t so = {evaluate ? 1 : 0, 1.2f, 5.0f, 2u}; // roptions.c:25,28,31
    for (uint64_t r = 0; r < b.n_reads; r++) {
        std::vector<rawdtw_chain_t> cand;
        for (uint64_t e = b.chain_off[r]; e < b.chain_off[r + 1]; e++) {
            if (log_scores && score[e] != -1e10f) // a cut chain returns before the fprintf (rmap.cpp:206-209, 265-268, 308-312)
                printf("log chaining_score=%f alignment_score=%f\n", b.chaining_score[order[e]], score[e]);
            if (evaluate && !keep[e]) continue; // rmap.cpp:518-529: the list is replaced only under EVALUATE_CHAINS
            const uint64_t c = order[e];
            const rawdtw_anchor_t *a = anchors.data() + anchor_off[e];
            const uint32_t na = (uint32_t)(anchor_off[e + 1] - anchor_off[e]);
            rawdtw_chain_t ch = {b.chaining_score[c], score[e], b.chain_seq[c], a[na - 1].target_position, a[0].target_position,
                                 na, b.chain_strand[c], 0u, (uint32_t)e};
            cand.push_back(ch);
        }
        printf("read %llu", (unsigned long long)r);
        if (cand.empty()) { printf(" nc=0\n"); continue; }
        std::vector<uint32_t> kept(cand.size());
        const uint32_t nk = rawdtw_gen_primary_chains(cand.data(), (uint32_t)cand.size(), &so, kept.data());
        std::vector<rawdtw_chain_t> prim(nk);
        for (uint32_t k = 0; k < nk; k++) prim[k] = cand[kept[k]];
        const int mapped = rawdtw_is_mapped_with_high_confidence(prim.data(), nk, &so);
        printf(" nc=%u mapq=%u mapped=%d primary=", nk, prim[0].mapq, mapped);
        for (uint32_t k = 0; k < nk; k++) printf("%s%u:%08x", k ? "," : "", prim[k].tag, bits(prim[k].alignment_score));
        if (mapped && cigar) {
            // rmap.cpp:715-717: align_chain(chains[0], ..., cigar=true): every part through DTW_global_tb
            const uint64_t e = prim[0].tag;
            const rawdtw_anchor_t *a = anchors.data() + anchor_off[e];
            const uint32_t na = (uint32_t)(anchor_off[e + 1] - anchor_off[e]);
            const uint32_t nj = rawdtw_chain_job_count(&b.opt, na);
            std::vector<rawdtw_job_t> jobs(nj);
            int st = rawdtw_chain_build_jobs(&b.opt, a, na, ref_base[e], read_base[e], 1, jobs.data());
            if (st == RAWDTW_ERR_UNSUPPORTED) { printf(" cigar=unsupported\n"); continue; } // rmap.cpp:223-225 assert(false)
            CHECK(st);
            std::vector<uint64_t> poff(nj + 1, 0);
            for (uint32_t k = 0; k < nj; k++) poff[k + 1] = poff[k] + jobs[k].n + jobs[k].m - 1;
            std::vector<uint32_t> plen(nj), pi(poff[nj]), pj(poff[nj]);
            std::vector<float> pd(poff[nj]), cost(nj);
            CHECK(rawdtw_traceback_batch(dtw, jobs.data(), nj, b.events.data(), b.events.size(), cost.data(), poff.data(),
                                         plen.data(), pi.data(), pj.data(), pd.data()));
            const float alns = rawdtw_chain_replay(&b.opt, a, na, cost.data(), -1e10f); // rmap.cpp:306 on the summed costs
            std::stringstream ss; // dtwresult_to_string, rmap.cpp:580-592
            const uint32_t parts = na - 1;
            for (uint32_t k = 0; k < nj; k++) {
                // sparse: every element offset by its part's start anchor (rmap.cpp:286-289); global: the offsets are
                // added to alignment.back() once per element (rmap.cpp:230-233), i.e. only the last tuple moves
                const rawdtw_anchor_t &s0 = b.opt.border_constraint == 0 ? a[na - 1] : a[parts - k];
                for (uint32_t q = 0; q < plen[k]; q++) {
                    size_t i = pi[poff[k] + q], j = pj[poff[k] + q];
                    if (b.opt.border_constraint != 0) { i += s0.query_position; j += s0.target_position; }
                    else if (q + 1 == plen[k]) { i += (size_t)plen[k] * s0.query_position; j += (size_t)plen[k] * s0.target_position; }
                    ss << "(" << i << "," << j << "," << pd[poff[k] + q] << ")";
                }
            }
            printf("\talns:f:%s\taln:s:%s", std::to_string(alns).c_str(), ss.str().c_str()); // rmap.cpp:741-744
        }
        printf("\n");
    }
    rawdtw_destroy(dtw);
    return 0;
}
