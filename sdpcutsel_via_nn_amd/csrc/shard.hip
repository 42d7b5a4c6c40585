// Sharded selection round (SURVEY.md section 8 e): the two device-side halves around the one
// RCCL all-gather of a multi-GPU round, each enqueued without host synchronisation.
//
//   sdpcut_shard_head_device   local head of the ranking -> one packed int64 record
//                              [class size, nb_violated, nb_positive, k_eff, void, 0,0,0 | scores | ids]
//                              (fp64 bit-cast; unused slots = (-inf, INT64_MAX)); header and padding are
//                              written by the selection's last kernel from device memory, the host
//                              never sees the counters here
//   (caller)                   all_gather_into_tensor of the records (torch.distributed / RCCL)
//   sdpcut_shard_finish_round  unpack -> replicated merge (score desc, id asc) -> eigen-cut rows of
//                              the merged head that belong to THIS shard -> one D2H, one sync
#include <cstring>

#include "common.h"
#include "keys.h"

#define SHARD_HDR 8

__global__ void shard_fill_kernel(int64_t count, int64_t *rec, int fields)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < SHARD_HDR) rec[i] = 0;
    if (i < count) {
        rec[SHARD_HDR + i] = __double_as_longlong(-__builtin_huge_val());
        rec[SHARD_HDR + count + i] = 0x7fffffffffffffffLL;
        if (fields == 3) rec[SHARD_HDR + 2 * count + i] = __double_as_longlong(-__builtin_huge_val());
    }
}

// Secondary key of a COMBALL record (every entry visited by the combined scan): obj_improve of each head entry;
// (-inf) behind the entries the selection wrote.
__global__ void shard_secondary_kernel(int64_t count, const int64_t *d_c4, const int64_t *ids, int64_t base, int64_t n,
                                       const double *obj, int64_t *sec_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t written = d_c4[4] ? 0 : d_c4[3];
    double v = -__builtin_huge_val();
    if (i < written) {
        const int64_t c = ids[i] - base;
        if (c >= 0 && c < n) v = obj[c];
    }
    sec_out[i] = __double_as_longlong(v);
}

// Replicated merge of the gathered heads.  Every head is already ordered (score descending, [secondary
// descending,] id ascending) -- the order a stable sort gives on one list (cut_select_qp.py:601, :625, :653) -- so
// the position of an entry in the merged list is its own position plus, per other rank, the number
// of entries that precede it there (one binary search each).  Pads (-inf, INT64_MAX) compare equal
// across ranks and are ordered by rank, which keeps the positions a permutation.
// fields = 2: [scores | ids]; 3: [scores | ids | secondary] (the combined strategy with every entry visited:
// equal new scores keep obj_improve order, the second stable sort of :625 under the first of :601).
// rl = distance between the records of consecutive ranks in words (>= 8 + fields * count: several lists' records may share one
// gathered buffer, the QCQP round's two covers travel in ONE all-gather)
__global__ void shard_mergerank_kernel(int world, int64_t count, int fields, int64_t rl, const int64_t *allrec, int64_t sel, double *scores,
                                       int64_t *ids, int64_t *headers)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)world * SHARD_HDR) headers[i] = allrec[(i / SHARD_HDR) * rl + (i % SHARD_HDR)];
    if (i >= (int64_t)world * count) return;
    const int r = (int)(i / count);
    const int64_t e = i - (int64_t)r * count;
    const int64_t sbits = allrec[r * rl + SHARD_HDR + e];
    const int64_t id = allrec[r * rl + SHARD_HDR + count + e];
    const uint64_t key = key_of(__longlong_as_double(sbits));
    const uint64_t sec = fields == 3 ? key_of(__longlong_as_double(allrec[r * rl + SHARD_HDR + 2 * count + e])) : 0ull;
    int64_t pos = e;
    for (int q = 0; q < world; ++q) {
        if (q == r) continue;
        const int64_t *qs = allrec + q * rl + SHARD_HDR, *qi = qs + count, *qx = qi + count;
        int64_t lo = 0, hi = count;   // first entry of rank q that does NOT precede (key, sec, id, r)
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            const uint64_t km = key_of(__longlong_as_double(qs[mid]));
            const int64_t im = qi[mid];
            bool before;
            if (km != key) before = km > key;
            else {
                const uint64_t sm = fields == 3 ? key_of(__longlong_as_double(qx[mid])) : 0ull;
                if (sm != sec) before = sm > sec;
                else before = im < id || (im == id && q < r);
            }
            if (before) lo = mid + 1; else hi = mid;
        }
        pos += lo;
    }
    if (pos < sel) {
        scores[pos] = __longlong_as_double(sbits);
        ids[pos] = id;
    }
}

extern "C" int sdpcut_shard_head_device(sdpcut_handle h, int strat, int64_t count, void *d_record)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (count < 1 || count > 16384 || !d_record) return sdpcut_fail(h, SDPCUT_EINVAL, "shard head: count must be 1..16384");
    const bool comball = strat == SDPCUT_PART_COMBALL;
    if (strat != SDPCUT_STRAT_FEAS && strat != SDPCUT_STRAT_OPT && strat != SDPCUT_PART_STRONG && !comball)
        return sdpcut_fail(h, SDPCUT_EINVAL, "shard head: strategy must be 1, 2, SDPCUT_PART_STRONG or SDPCUT_PART_COMBALL");
    const uint32_t need = strat == SDPCUT_STRAT_FEAS ? SDPCUT_EIG
                          : strat == SDPCUT_STRAT_OPT ? SDPCUT_NN : (SDPCUT_EIG | SDPCUT_NN);
    if (h->N > 0 && (!h->have_point || !h->d_eig)) return sdpcut_fail(h, SDPCUT_ESTATE, "set_candidates and set_point first");
    HIP_TRY(h, hipSetDevice(h->device));
    // Not scored at this point yet: the call scores the shard itself, and the score kernels count the leading
    // digit of the selection keys on the way (score_for_selection) -- a sharded round needs no sdpcut_score.
    int stage = 0;
    bool auto_regime = false;
    if (h->N > 0) {
        int rc0;
        if (comball) rc0 = ((h->scored & need) != need) ? sdpcut_score(h, need & ~h->scored) : 0;     // (normally scored by the round's first attempt)
        else rc0 = score_for_selection(h, strat, 0, count, need, false, &stage, &auto_regime);
        if (rc0) return rc0;
    }
    int64_t *rec = (int64_t *)d_record;
    const int grid = (int)((count + 255) / 256);
    if (h->N > 0) {
        // the selection's last kernel writes the header (counters straight from device memory) and the
        // padding behind the entries it emits: no extra launch
        const int64_t *d_c4 = nullptr;
        h->shard_rec = rec;
        h->shard_rec_count = count;
        h->shard_rec_len = (strat == SDPCUT_STRAT_OPT || comball) ? h->N : -1;
        int rc;
        if (comball) {
            // every entry of the shard is visited by the scan: its combined ranking is a sub-list of the global one; the
            // record carries obj_improve as secondary key for the merge
            rc = topk_select_enqueue(h, 4 /* TK_MODE_COMBALL */, count < h->N ? count : h->N, 0.0, rec + SHARD_HDR + count,
                                     (double *)(rec + SHARD_HDR), &d_c4, 0, 0);
            rc = rc ? rc : 1;
        } else {
            rc = rank_fast_enqueue(h, strat, 0, count, rec + SHARD_HDR + count, (double *)(rec + SHARD_HDR), &d_c4, stage, false);
        }
        h->shard_rec = nullptr;
        if (rc < 0) return rc;
        if (rc != 1) return sdpcut_fail(h, SDPCUT_EINVAL, "shard head: request not eligible for the select path");
        if (comball)
            hipLaunchKernelGGL(shard_secondary_kernel, dim3(grid), dim3(256), 0, h->stream, count, d_c4, rec + SHARD_HDR + count, h->base,
                               h->N, h->d_obj, rec + SHARD_HDR + 2 * count);
    } else {
        hipLaunchKernelGGL(shard_fill_kernel, dim3(grid), dim3(256), 0, h->stream, count, rec, comball ? 3 : 2);   // empty shard
    }
    HIP_TRY(h, hipGetLastError());
    return SDPCUT_OK;
}

// Second half of a sharded round, enqueued WITHOUT host synchronisation: unpack the gathered records, merge,
// rows of the merged head that live on this shard, everything stored straight into the handle's pinned block.
// sdpcut_shard_finish_wait collects it.  (Two lists per round -- the QCQP composition -- enqueue both halves
// before the first wait: one host wait per round.)
extern "C" int sdpcut_shard_finish_enqueue(sdpcut_handle h, int32_t world, int64_t count, int32_t fields, const void *d_allrec,
                                           int64_t pitch_words, int64_t sel_size, int32_t coef_ld)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    const int64_t rl = SHARD_HDR + (int64_t)fields * count;
    if (pitch_words == 0) pitch_words = rl;
    if (world < 1 || count < 1 || !d_allrec || sel_size < 1 || sel_size > (int64_t)world * count || (fields != 2 && fields != 3) ||
        pitch_words < rl)
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad shard_finish arguments");
    if (sel_size > 16384) return sdpcut_fail(h, SDPCUT_EINVAL, "shard_finish: at most 16384 entries");
    if (coef_ld < h->row_len_max || coef_ld > SDPCUT_ROW_LD) return sdpcut_fail(h, SDPCUT_EINVAL, "bad coef_ld");
    if (!h->have_point) return sdpcut_fail(h, SDPCUT_ESTATE, "set_point first");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t tot = (size_t)world * count, s = (size_t)sel_size;
    // pinned host block, written by the device: headers | ids | scores | lam | rhs | coef | ks
    const size_t hdr_b = (size_t)world * SHARD_HDR * 8;
    const size_t ret_bytes = hdr_b + s * 8 * (4 + (size_t)coef_ld) + s * 4;
    int rc = ensure_stage(h, s * 16 + 64);
    if (rc) return rc;
    rc = ensure_pinned(h, ret_bytes + s * 4);      // + int32 pos[sel] of the compacted form
    if (rc) return rc;
    int64_t *d_mi = (int64_t *)h->d_stage;
    double *d_ms = (double *)(d_mi + s);
    const size_t nthr = tot > (size_t)world * SHARD_HDR ? tot : (size_t)world * SHARD_HDR;
    hipLaunchKernelGGL(shard_mergerank_kernel, dim3((int)((nthr + 255) / 256)), dim3(256), 0, h->stream, (int)world,
                       count, (int)fields, pitch_words, (const int64_t *)d_allrec, sel_size, d_ms, d_mi, (int64_t *)h->pinned_dev);
    // rows of the merged head that live on this shard (the others are marked ks = 0, lam = NaN),
    // stored -- with the merged ids and scores -- straight into the host block
    // (completion: the rows kernel's last workgroup stores the round's serial number into word 7 of the
    // first header -- a pad word of the record -- and the host polls it, see wait_round_done)
    const int64_t serial = ++h->round_serial;
    rc = launch_round_rows(h, sel_size, nullptr, d_mi, d_ms, coef_ld, h->pinned_dev, (int64_t)hdr_b, serial);
    if (rc) return rc;
    h->shard_pending_serial = serial;
    h->shard_pending_world = world;
    h->shard_pending_sel = sel_size;
    h->shard_pending_ld = coef_ld;
    return SDPCUT_OK;
}

// compact_own != 0: the rows of THIS shard are moved to the front of lam / rhs / coef / ks (head order) and their
// positions in the head appended as int32 pos[sel] behind ks; *n_own = their number.
extern "C" int sdpcut_shard_finish_wait(sdpcut_handle h, int32_t compact_own, const void **block, int64_t *n_own)
{
    if (!h) return SDPCUT_EINVAL;
    if (!block || (compact_own && !n_own)) return sdpcut_fail(h, SDPCUT_EINVAL, "bad shard_finish_wait arguments");
    if (!h->shard_pending_serial) return sdpcut_fail(h, SDPCUT_ESTATE, "no sharded round pending: sdpcut_shard_finish_enqueue first");
    int rc = wait_round_done(h, (const int64_t *)h->pinned + 7, h->shard_pending_serial);
    h->shard_pending_serial = 0;
    if (rc) return rc;
    ((int64_t *)h->pinned)[7] = 0;      // the caller sees the record's pad word, not the completion mark
    *block = h->pinned;
    if (!compact_own) return SDPCUT_OK;
    const size_t s = (size_t)h->shard_pending_sel, ld = (size_t)h->shard_pending_ld;
    char *p = (char *)h->pinned + (size_t)h->shard_pending_world * SHARD_HDR * 8 + s * 16;      // behind headers, ids, scores
    double *lam = (double *)p;
    double *rhs = lam + s;
    double *coef = rhs + s;
    int32_t *ks = (int32_t *)(coef + s * ld);
    int32_t *pos = ks + s;
    size_t w = 0;
    for (size_t i = 0; i < s; ++i) {
        if (ks[i] <= 0) continue;
        if (w != i) {
            lam[w] = lam[i];
            rhs[w] = rhs[i];
            std::memmove(coef + w * ld, coef + i * ld, ld * sizeof(double));
            ks[w] = ks[i];
        }
        pos[w++] = (int32_t)i;
    }
    *n_own = (int64_t)w;
    return SDPCUT_OK;
}

extern "C" int sdpcut_shard_finish_round_view(sdpcut_handle h, int32_t world, int64_t count, const void *d_allrec,
                                              int64_t sel_size, int32_t coef_ld, const void **block)
{
    int rc = sdpcut_shard_finish_enqueue(h, world, count, 2, d_allrec, 0, sel_size, coef_ld);
    if (rc) return rc;
    return sdpcut_shard_finish_wait(h, 0, block, nullptr);
}

extern "C" int sdpcut_shard_finish_round_own(sdpcut_handle h, int32_t world, int64_t count, const void *d_allrec,
                                             int64_t sel_size, int32_t coef_ld, const void **block, int64_t *n_own)
{
    if (!h) return SDPCUT_EINVAL;
    if (!n_own) return sdpcut_fail(h, SDPCUT_EINVAL, "bad shard_finish_round arguments");
    int rc = sdpcut_shard_finish_enqueue(h, world, count, 2, d_allrec, 0, sel_size, coef_ld);
    if (rc) return rc;
    return sdpcut_shard_finish_wait(h, 1, block, n_own);
}

extern "C" int sdpcut_shard_finish_round(sdpcut_handle h, int32_t world, int64_t count, const void *d_allrec,
                                         int64_t sel_size, int32_t coef_ld, int64_t *headers_out, int64_t *idx_out,
                                         double *score_out, double *lam_min, double *coef, double *rhs, int32_t *ks)
{
    if (!h) return SDPCUT_EINVAL;
    if (!headers_out || !idx_out || !score_out || !lam_min || !coef || !rhs || !ks)
        return sdpcut_fail(h, SDPCUT_EINVAL, "bad shard_finish_round arguments");
    const void *block = nullptr;
    int rc = sdpcut_shard_finish_round_view(h, world, count, d_allrec, sel_size, coef_ld, &block);
    if (rc) return rc;
    const size_t s = (size_t)sel_size, hdr_b = (size_t)world * SHARD_HDR * 8;
    const char *q = (const char *)block;
    std::memcpy(headers_out, q, hdr_b); q += hdr_b;
    std::memcpy(idx_out, q, s * 8); q += s * 8;
    std::memcpy(score_out, q, s * 8); q += s * 8;
    std::memcpy(lam_min, q, s * 8); q += s * 8;
    std::memcpy(rhs, q, s * 8); q += s * 8;
    std::memcpy(coef, q, s * 8 * (size_t)coef_ld); q += s * 8 * (size_t)coef_ld;
    std::memcpy(ks, q, s * 4);
    return SDPCUT_OK;
}
