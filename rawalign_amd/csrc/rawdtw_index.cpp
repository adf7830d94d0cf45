// rawdtw_index.cpp -- the part of ri_idx_load (src/rawindex.cpp:317-377) the DTW path needs: header, sequence table and the
// per-sequence signal arrays of a RawAlign .ind file, streamed into the context's reference arena (include/rawdtw.h).
#include "rawdtw_capi.h"

using namespace rawdtw;
using namespace rawdtw::capi;

extern "C" {

// ---- index reader --------------------------------------------------------------------------------
int rawdtw_index_open(const char *path, rawdtw_index **out)
{
    if (!path || !out) return RAWDTW_ERR_INVALID;
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return RAWDTW_ERR_INVALID;
    rawdtw_index *ix = new (std::nothrow) rawdtw_index;
    if (!ix) { fclose(f); return RAWDTW_ERR_OOM; }
    ix->path = path;
    char magic[2];
    bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'R' && magic[1] == 'I'; // rawindex.h:7-8 RI_IDX_MAGIC, 2 bytes
    ok = ok && fread(ix->pars, 4, 8, f) == 8;
    const uint32_t n_seq = ok ? ix->pars[6] : 0;
    for (uint32_t i = 0; ok && i < n_seq; i++) {
        uint8_t l = 0;
        ok = fread(&l, 1, 1, f) == 1;
        std::string name(l, '\0');
        if (ok && l) ok = fread(&name[0], 1, l, f) == l;
        uint32_t len = 0;
        ok = ok && fread(&len, 4, 1, f) == 1;
        if (!ok) break;
        ix->names.push_back(name);
        ix->lens.push_back(len);
        ix->fwd_pos.push_back((uint64_t)ftello(f));
        ok = fseeko(f, (off_t)len * 8, SEEK_CUR) == 0; // skip forward + reverse arrays
    }
    fclose(f);
    if (!ok) { delete ix; return RAWDTW_ERR_INVALID; }
    *out = ix;
    return RAWDTW_OK;
}

int rawdtw_index_info(const rawdtw_index *idx, uint32_t *n_seq, uint32_t pars[8])
{
    if (!idx) return RAWDTW_ERR_INVALID;
    if (n_seq) *n_seq = (uint32_t)idx->lens.size();
    if (pars) memcpy(pars, idx->pars, sizeof(idx->pars));
    return RAWDTW_OK;
}

int rawdtw_index_seq(const rawdtw_index *idx, uint32_t i, const char **name, uint32_t *len)
{
    if (!idx || i >= idx->lens.size()) return RAWDTW_ERR_INVALID;
    if (name) *name = idx->names[i].c_str();
    if (len) *len = idx->lens[i];
    return RAWDTW_OK;
}

int rawdtw_index_read_signal(const rawdtw_index *idx, uint32_t i, int strand, float *out)
{
    if (!idx || i >= idx->lens.size() || !out) return RAWDTW_ERR_INVALID;
    FILE *f = fopen(idx->path.c_str(), "rb");
    if (!f) return RAWDTW_ERR_INVALID;
    // file order: forward_signals[i] then reverse_signals[i]; strand==1 selects forward (rmap.cpp:182-188)
    const uint64_t pos = idx->fwd_pos[i] + (strand == 1 ? 0 : (uint64_t)idx->lens[i] * 4);
    bool ok = fseeko(f, (off_t)pos, SEEK_SET) == 0 && fread(out, 4, idx->lens[i], f) == idx->lens[i];
    fclose(f);
    return ok ? RAWDTW_OK : RAWDTW_ERR_INVALID;
}

int rawdtw_index_upload(rawdtw_ctx *ctx, const rawdtw_index *idx)
{
    if (!ctx || !idx) return RAWDTW_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n_seq = (uint32_t)idx->lens.size();
    drop_reference(ctx);
    ctx->ref_off.assign(2ull * n_seq, 0);
    ctx->ref_len = idx->lens;
    uint64_t total = 0;
    for (uint32_t s = 0; s < n_seq; s++) {
        ctx->ref_off[2 * s] = total; total += ((uint64_t)idx->lens[s] + 3) & ~3ull;
        ctx->ref_off[2 * s + 1] = total; total += ((uint64_t)idx->lens[s] + 3) & ~3ull;
    }
    int st = dev_alloc(ctx, &ctx->d_ref, std::max<uint64_t>(total, 4));
    if (st != RAWDTW_OK) return st;
    ctx->ref_hold = new (std::nothrow) RefHold;
    if (!ctx->ref_hold) { (void)hipFree(ctx->d_ref); ctx->d_ref = nullptr; return fail(ctx, RAWDTW_ERR_OOM, "host allocation failed"); }
    ctx->ref_hold->d = ctx->d_ref;
    ctx->n_ref = total;
    FILE *f = fopen(idx->path.c_str(), "rb");
    if (!f) return fail(ctx, RAWDTW_ERR_INVALID, "cannot reopen index file");
    // stream through two pinned staging buffers so that the file read overlaps the H2D copy
    const size_t CH = 16u << 20; // floats per staging buffer (64 MiB)
    float *stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipError_t e = hipHostMalloc((void **)&stage[0], CH * 4, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&stage[1], CH * 4, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreate(&done[0]);
    if (e == hipSuccess) e = hipEventCreate(&done[1]);
    bool ok = e == hipSuccess;
    int which = 0;
    bool used[2] = {false, false};
    for (uint32_t s = 0; ok && s < n_seq; s++) {
        ok = fseeko(f, (off_t)idx->fwd_pos[s], SEEK_SET) == 0;
        for (int strand_slot = 0; ok && strand_slot < 2; strand_slot++) {
            uint64_t left = idx->lens[s], at = ctx->ref_off[2 * s + strand_slot];
            while (ok && left) {
                const size_t take = (size_t)std::min<uint64_t>(left, CH);
                if (used[which]) ok = hipEventSynchronize(done[which]) == hipSuccess;
                ok = ok && fread(stage[which], 4, take, f) == take;
                ok = ok && hipMemcpyAsync(ctx->d_ref + at, stage[which], take * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
                ok = ok && hipEventRecord(done[which], ctx->stream) == hipSuccess;
                used[which] = true;
                which ^= 1; left -= take; at += take;
            }
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) ok = false;
    fclose(f);
    for (int k = 0; k < 2; k++) { if (stage[k]) (void)hipHostFree(stage[k]); if (done[k]) (void)hipEventDestroy(done[k]); }
    if (!ok) return fail(ctx, RAWDTW_ERR_DEVICE, "index upload failed (short file or HIP error)");
    return RAWDTW_OK;
}

int rawdtw_index_close(rawdtw_index *idx)
{
    delete idx;
    return RAWDTW_OK;
}

} // extern "C"
