// Ranking on the device (reference: the sorts and the combined scan of
// _sel_eigcut_by_ordering_on_measure, cut_select_qp.py:601-632 and :649-654).
//
// Python's list.sort(reverse=True) is stable, so ties keep ascending candidate index.  The
// same order is produced here by a stable LSD radix sort (rocPRIM, AMD's native primitive
// library) over an order-preserving u64 image of the fp64 score with the candidate index as
// payload.  The combined strategy's sequential early-exit scan is evaluated in closed form:
// a prefix count of "positive and violated" entries in first-sort order tells every entry
// whether the reference's loop would have reached it.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "common.h"
#include "keys.h"

// counters: [0] nb_violated  [1] strong  [2] violated_in_scan  [3] nb_positive
__global__ void keys_first_kernel(int strat, int64_t n, const double *eig, const double *obj, uint64_t *key,
                                  uint32_t *val, int64_t *counters)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int viol = 0, pos = 0;
    if (i < n) {
        double s;
        if (strat == SDPCUT_STRAT_FEAS) {
            const double lam = eig[i];
            viol = lam < SDPCUT_NEG_EIGVAL;            // cut_select_qp.py:648
            s = viol ? -lam : 0.0;                     // (0, 0) entries sort behind every violated one
        } else if (strat == SDPCUT_PART_STRONG) {
            // class-restricted ranking; `viol` doubles as the class-membership count
            const double o = obj[i];
            viol = (o > 0.0) && (eig[i] < SDPCUT_NEG_EIGVAL);
            s = viol ? o : -__builtin_huge_val();
        } else {
            s = obj[i];
            pos = s > 0.0;
            if (eig) viol = eig[i] < SDPCUT_NEG_EIGVAL;
        }
        key[i] = key_of(s);
        val[i] = (uint32_t)i;
    }
    // wave-level reduction before the atomics
    const unsigned long long mv = __ballot(viol), mp = __ballot(pos);
    if ((threadIdx.x & 63) == 0) {
        if (mv) atomicAdd((unsigned long long *)&counters[0], (unsigned long long)__popcll(mv));
        if (mp) atomicAdd((unsigned long long *)&counters[3], (unsigned long long)__popcll(mp));
    }
}

// flag[i] = 1 iff entry i of the first-sort order is positive and violated
__global__ void comb_flag_kernel(int64_t n, const uint64_t *key_sorted, const uint32_t *val_sorted,
                                 const double *eig, int32_t *flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = score_of(key_sorted[i]);
    flag[i] = (s > 0.0) && (eig[val_sorted[i]] < SDPCUT_NEG_EIGVAL);
}

// second-sort keys (cut_select_qp.py:606-623); `before` = exclusive prefix count of flag
__global__ void comb_keys_kernel(int64_t n, int64_t sel_size, const uint64_t *key_sorted,
                                 const uint32_t *val_sorted, const int32_t *flag, const int32_t *before,
                                 const double *eig, uint64_t *key2, int64_t *counters)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int up = 0, low = 0;
    if (i < n) {
        double s = score_of(key_sorted[i]);
        const bool visited = (int64_t)before[i] < sel_size;
        if (visited) {
            const double lam = eig[val_sorted[i]];
            const bool viol = lam < SDPCUT_NEG_EIGVAL;
            if (s > 0.0) {
                if (viol) { s = s + SDPCUT_BIG_M; up = 1; }
                else      { s = s - SDPCUT_BIG_M; }
            } else if (viol) {
                s = -lam; low = 1;
            }
        }
        key2[i] = key_of(s);
    }
    const unsigned long long mu = __ballot(up), ml = __ballot(low);
    if ((threadIdx.x & 63) == 0) {
        if (mu) atomicAdd((unsigned long long *)&counters[1], (unsigned long long)__popcll(mu));
        if (mu | ml)
            atomicAdd((unsigned long long *)&counters[2], (unsigned long long)(__popcll(mu) + __popcll(ml)));
    }
}

__global__ void emit_kernel(int64_t count, int64_t base, const uint64_t *key, const uint32_t *val,
                            int64_t *idx_out, double *score_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    idx_out[i] = base + (int64_t)val[i];
    score_out[i] = score_of(key[i]);
}

// merge helpers: three stable passes over a permutation (id asc, secondary desc, score desc)
__global__ void merge_idkey_kernel(int64_t n, const int64_t *ids, uint64_t *key, uint32_t *perm)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = (uint64_t)ids[i];
    perm[i] = (uint32_t)i;
}
__global__ void merge_gatherkey_kernel(int64_t n, const double *vals, const uint32_t *perm, uint64_t *key)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = key_of(vals[perm[i]]);
}
__global__ void merge_emit_kernel(int64_t count, const uint32_t *perm, const double *scores, const int64_t *ids,
                                  double *score_out, int64_t *id_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    score_out[i] = scores[perm[i]] + 0.0;
    id_out[i] = ids[perm[i]];
}
__global__ void gather_scores_kernel(int64_t count, int64_t base, int64_t n, const int64_t *ids, const double *eig,
                                     const double *obj, double *eig_out, double *obj_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t c = ids[i] - base;
    const bool ok = c >= 0 && c < n;
    if (eig_out) eig_out[i] = ok ? eig[c] : __builtin_nan("");
    if (obj_out) obj_out[i] = ok ? obj[c] : __builtin_nan("");
}

void free_rank_ws(sdpcut_ctx *h)
{
    hipFree(h->d_key_a); hipFree(h->d_key_b); hipFree(h->d_val_a); hipFree(h->d_val_b);
    hipFree(h->d_flag); hipFree(h->d_scan); hipFree(h->d_tmp);
    h->d_key_a = h->d_key_b = nullptr; h->d_val_a = h->d_val_b = nullptr;
    h->d_flag = h->d_scan = nullptr; h->d_tmp = nullptr; h->tmp_bytes = 0; h->ws_n = 0; h->key_n = 0;
}

// the key array alone (8 B per candidate): all that the head selection (topk.hip) may need of this workspace -- a round whose score
// kernels counted the leading digit does not even read it
int ensure_key_ws(sdpcut_ctx *h, int64_t n)
{
    if (n <= h->key_n) return 0;
    (void)hipFree(h->d_key_a);
    h->d_key_a = nullptr;
    h->key_n = 0;
    HIP_TRY(h, hipMalloc((void **)&h->d_key_a, (size_t)(n < 1 ? 1 : n) * 8));
    h->key_n = n;
    return 0;
}

// the whole workspace of the FULL-LIST ranking (~50 B per candidate: two key and two payload arrays, flags, scan, rocPRIM's
// temporary storage).  Allocated when a full list is first asked for -- sdpcut_rank / rank_device with max_out > 16 384, what the
// reference's method returns (cut_select_qp.py:601, :653: the whole sorted list) and a caller reads beyond the head
// (sdpcut_rank_fetch) -- never by a round of the cutting-plane loop, whose heads of <= 5000 come from the radix select (r5;
// until r4 every candidate list allocated it up front).
int ensure_rank_ws(sdpcut_ctx *h, int64_t n)
{
    int rc = ensure_key_ws(h, n);
    if (rc) return rc;
    if (n <= h->ws_n) return 0;
    hipFree(h->d_key_b); hipFree(h->d_val_a); hipFree(h->d_val_b); hipFree(h->d_flag); hipFree(h->d_scan); hipFree(h->d_tmp);
    h->d_key_b = nullptr; h->d_val_a = h->d_val_b = nullptr; h->d_flag = h->d_scan = nullptr; h->d_tmp = nullptr;
    h->tmp_bytes = 0; h->ws_n = 0;
    const size_t nn = (size_t)(n < 1 ? 1 : n);
    HIP_TRY(h, hipMalloc((void **)&h->d_key_b, nn * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_val_a, nn * 8));   // 8 B/entry: also used as u64 by the merge
    HIP_TRY(h, hipMalloc((void **)&h->d_val_b, nn * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_flag, nn * 4));
    HIP_TRY(h, hipMalloc((void **)&h->d_scan, nn * 4));
    size_t t1 = 0, t2 = 0, t3 = 0;
    HIP_TRY(h, rocprim::radix_sort_pairs_desc(nullptr, t1, h->d_key_a, h->d_key_b, h->d_val_a, h->d_val_b, nn,
                                              0, 64, h->stream));
    HIP_TRY(h, rocprim::radix_sort_pairs(nullptr, t2, h->d_key_a, h->d_key_b, (uint64_t *)h->d_val_a,
                                         (uint64_t *)h->d_val_b, nn, 0, 64, h->stream));
    HIP_TRY(h, rocprim::exclusive_scan(nullptr, t3, h->d_flag, h->d_scan, 0, nn, rocprim::plus<int32_t>(),
                                       h->stream));
    size_t t = t1 > t2 ? t1 : t2;
    t = t > t3 ? t : t3;
    HIP_TRY(h, hipMalloc(&h->d_tmp, t < 256 ? 256 : t));
    h->tmp_bytes = t;
    h->ws_n = n;
    return 0;
}

static inline int nblk(int64_t n) { return (int)((n + 255) / 256); }

// Fast path of the ranking: the caller wants only a short head (the loop consumes <= 5000
// entries, _SDP_CUTS_PER_ROUND_MAX): radix select + small sort instead of sorting all N.
// Returns 1 if the selection has been enqueued (no host synchronisation), 0 if the request is
// not eligible, < 0 on error.  *d_c4 = device address of {class size, nb_violated, nb_positive, k_eff}.
int rank_fast_mode(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t max_out, double *score_add)
{
    const int64_t n = h->N;
    if (sel_size > n) sel_size = n;
    if (score_add) *score_add = 0.0;
    if (!(n > 0 && max_out >= 1 && max_out <= 16384)) return 0;      // TK_MAXK (topk_dev.h)
    if (max_out > 8192 && strat == SDPCUT_STRAT_COMB) return 0;     // big heads: plain rankings only (the device-resolved regime's sort keeps keys AND indices in LDS)
    if (strat == SDPCUT_STRAT_FEAS) return 1;
    if (strat == SDPCUT_STRAT_OPT) return 2;
    if (strat == SDPCUT_PART_STRONG) return 3;
    if (strat == SDPCUT_STRAT_COMB && sel_size >= 1 && max_out <= sel_size) {
        // combined scan, common regime: at least sel_size candidates are positive AND violated.
        // The scan stops after sel_size of them; the re-sorted list starts with exactly those,
        // +BIG_M, in obj_improve order (cut_select_qp.py:606-625).  rank_fast_finish verifies
        // the regime through the class size.
        if (score_add) *score_add = SDPCUT_BIG_M;
        return 3;
    }
    return 0;
}

int rank_fast_enqueue(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t max_out, int64_t *d_idx_out,
                      double *d_score_out, const int64_t **d_c4, int stage, bool auto_regime)
{
    double add = 0.0;
    int mode = rank_fast_mode(h, strat, sel_size, max_out, &add);
    if (!mode) return 0;
    int64_t sel = sel_size < h->N ? sel_size : h->N;
    if (auto_regime && strat == SDPCUT_STRAT_COMB) mode = 5 /* TK_MODE_COMBAUTO */;
    int rc = topk_select_enqueue(h, mode, max_out, add, d_idx_out, d_score_out, d_c4, stage, sel);
    return rc ? rc : 1;
}

// Host side of the fast path once the counters are on the host ({class size, nb_violated,
// nb_positive, k_eff, void flag, strong count, mode}).  Returns 1 if the enqueued selection is the
// answer, 0 if the general path has to run (a selection that gave up; the combined scan visiting
// everything when the regime was not resolved on the device).
int rank_fast_finish(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t max_out, const int64_t c4[7],
                     int64_t *n_written, int64_t *n_total, int32_t *new_strat, int64_t *counters_out)
{
    const int64_t n = h->N;
    if (sel_size > n) sel_size = n;
    const bool comb = strat == SDPCUT_STRAT_COMB;
    if (c4[4]) return 0;
    const bool all_visited = comb && c4[6] == 4 /* TK_MODE_COMBALL: resolved on the device, strong < sel_size */;
    if (comb && !all_visited && c4[0] < sel_size) return 0;
    const int64_t total = (strat == SDPCUT_STRAT_OPT || comb) ? n : c4[0];
    const int64_t w = total < max_out ? total : max_out;
    h->last_total = -1;                     // only the head exists: nothing to fetch later
    if (n_written) *n_written = w;
    if (n_total) *n_total = total;
    int64_t cnt[4] = {c4[1], 0, 0, c4[2]};
    if (strat == SDPCUT_STRAT_FEAS || strat == SDPCUT_PART_STRONG) cnt[0] = c4[0];
    if (comb) { cnt[1] = sel_size; cnt[2] = sel_size; }
    if (all_visited) { cnt[1] = c4[5]; cnt[2] = c4[1]; }     // every violated entry is seen by the scan
    if (new_strat) {
        *new_strat = strat;
        if (comb && (double)cnt[1] / (double)sel_size < (double)cnt[2] / (double)n) *new_strat = SDPCUT_STRAT_FEAS;
    }
    if (counters_out)
        for (int i = 0; i < 4; ++i) counters_out[i] = cnt[i];
    return 1;
}

int rank_on_device(sdpcut_ctx *h, int strat, int64_t sel_size, int64_t max_out, int64_t *d_idx_out,
                   double *d_score_out, int64_t *n_written, int64_t *n_total, int32_t *new_strat,
                   int64_t *counters_out, int64_t strong_hint)
{
    const int64_t n = h->N;
    int rc = ensure_rank_ws(h, n);
    if (rc) return rc;
    if (sel_size > n) sel_size = n;                 // cut_select_qp.py:551
    if (sel_size < 0) sel_size = 0;
    int64_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // ---- fast path: the caller wants only a short head (topk.hip).  strong_hint >= 0: the caller
    // has just tried it and learnt that only strong_hint < sel_size candidates are strong.
    int64_t strong = strong_hint;
    if (strong < 0) {
        const int64_t *d_c4 = nullptr;
        rc = rank_fast_enqueue(h, strat, sel_size, max_out, d_idx_out, d_score_out, &d_c4);
        if (rc < 0) return rc;
        if (rc == 1) {
            int64_t c4[7] = {0, 0, 0, 0, 0, 0, 0};
            HIP_TRY(h, hipMemcpyAsync(c4, d_c4, 5 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, sdpcut_sync(h));
            if (rank_fast_finish(h, strat, sel_size, max_out, c4, n_written, n_total, new_strat, counters_out)) return 0;
            // fewer strong candidates than sel_size: every entry is visited
            if (!c4[4] && strat == SDPCUT_STRAT_COMB) strong = c4[0];
        }
    }
    // ---- combined scan that visits EVERY entry (strong < sel_size): the new score of an entry is
    // then a function of its own (obj_improve, eigmin) alone (cut_select_qp.py:606-623), and the head
    // of the re-sorted list is a top-k by (new score, obj_improve, index) -- radix select again, with
    // obj_improve as secondary key where new scores tie (second stable sort of :625).
    if (strat == SDPCUT_STRAT_COMB && strong >= 0 && strong < sel_size && n > 0 && max_out >= 1 && max_out <= 16384) {
        const int64_t *d_c4 = nullptr;
        rc = topk_select_enqueue(h, 4 /* TK_MODE_COMBALL */, max_out, 0.0, d_idx_out, d_score_out, &d_c4);
        if (rc) return rc;
        int64_t c4[5] = {0, 0, 0, 0, 0};
        HIP_TRY(h, hipMemcpyAsync(c4, d_c4, 5 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, sdpcut_sync(h));
        bool answered = !c4[4];
        if (c4[4] == 2) {   // more equal new scores at the threshold than the sort buffers hold: cut the group by its secondary key
            rc = topk_tie_split(h, max_out, d_idx_out, d_score_out, nullptr);
            if (rc < 0) return rc;
            answered = rc == 0;
        }
        if (answered) {
            const int64_t w = n < max_out ? n : max_out;
            h->last_total = -1;
            if (n_written) *n_written = w;
            if (n_total) *n_total = n;
            // every violated entry is seen by the scan: violated_in_scan = nb_violated
            const int64_t c[4] = {c4[1], strong, c4[1], c4[2]};
            if (new_strat)
                *new_strat = ((double)c[1] / (double)sel_size < (double)c[2] / (double)n) ? SDPCUT_STRAT_FEAS : SDPCUT_STRAT_COMB;
            if (counters_out)
                for (int i = 0; i < 4; ++i) counters_out[i] = c[i];
            return 0;
        }
    }
    HIP_TRY(h, hipMemsetAsync(h->d_counters, 0, 8 * sizeof(int64_t), h->stream));
    const uint64_t *fkey = h->d_key_b;
    const uint32_t *fval = h->d_val_b;
    if (n > 0) {
        const double *eig = (h->scored & SDPCUT_EIG) ? h->d_eig : nullptr;
        hipLaunchKernelGGL(keys_first_kernel, dim3(nblk(n)), dim3(256), 0, h->stream, strat, n, eig, h->d_obj,
                           h->d_key_a, h->d_val_a, h->d_counters);
        size_t tb = h->tmp_bytes;
        HIP_TRY(h, rocprim::radix_sort_pairs_desc(h->d_tmp, tb, h->d_key_a, h->d_key_b, h->d_val_a, h->d_val_b,
                                                  (size_t)n, 0, 64, h->stream));
        if (strat == SDPCUT_STRAT_COMB) {
            hipLaunchKernelGGL(comb_flag_kernel, dim3(nblk(n)), dim3(256), 0, h->stream, n, h->d_key_b, h->d_val_b,
                               h->d_eig, h->d_flag);
            tb = h->tmp_bytes;
            HIP_TRY(h, rocprim::exclusive_scan(h->d_tmp, tb, h->d_flag, h->d_scan, 0, (size_t)n,
                                               rocprim::plus<int32_t>(), h->stream));
            hipLaunchKernelGGL(comb_keys_kernel, dim3(nblk(n)), dim3(256), 0, h->stream, n, sel_size, h->d_key_b,
                               h->d_val_b, h->d_flag, h->d_scan, h->d_eig, h->d_key_a, h->d_counters);
            // second stable sort: payload keeps first-sort order among equal new scores (:625)
            tb = h->tmp_bytes;
            HIP_TRY(h, hipMemcpyAsync(h->d_val_a, h->d_val_b, (size_t)n * 4, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, rocprim::radix_sort_pairs_desc(h->d_tmp, tb, h->d_key_a, h->d_key_b, h->d_val_a, h->d_val_b,
                                                      (size_t)n, 0, 64, h->stream));
        }
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipMemcpyAsync(cnt, h->d_counters, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    // :654 rank_list[0:nb_violated]; class-restricted rankings list their class only
    const int64_t total = (strat == SDPCUT_STRAT_FEAS || strat == SDPCUT_PART_STRONG) ? cnt[0] : n;
    int64_t w = total < max_out ? total : max_out;
    if (w < 0) w = 0;
    if (w > 0) {
        hipLaunchKernelGGL(emit_kernel, dim3(nblk(w)), dim3(256), 0, h->stream, w, h->base, fkey, fval, d_idx_out,
                           d_score_out);
        HIP_TRY(h, hipGetLastError());
    }
    h->last_total = total;
    if (n_written) *n_written = w;
    if (n_total) *n_total = total;
    if (new_strat) {
        *new_strat = strat;
        if (strat == SDPCUT_STRAT_COMB && sel_size > 0 && n > 0) {
            // (1, rank_list) if strong/sel_size < violated/len(rank_list) else (strat, rank_list), :630-631
            if ((double)cnt[1] / (double)sel_size < (double)cnt[2] / (double)n) *new_strat = SDPCUT_STRAT_FEAS;
        }
    }
    if (counters_out)
        for (int i = 0; i < 4; ++i) counters_out[i] = cnt[i];
    return 0;
}

int rank_fetch_on_device(sdpcut_ctx *h, int64_t offset, int64_t count, int64_t *d_idx_out, double *d_score_out)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(emit_kernel, dim3(nblk(count)), dim3(256), 0, h->stream, count, h->base, h->d_key_b + offset,
                       h->d_val_b + offset, d_idx_out, d_score_out);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int merge_topk_on_device(sdpcut_ctx *h, int64_t count, const double *d_scores, const double *d_secondary,
                         const int64_t *d_ids, int64_t max_out, double *d_score_out, int64_t *d_id_out)
{
    if (count <= 0 || max_out <= 0) return 0;
    h->last_total = -1;   // the merge reuses the ranking workspace
    int rc = ensure_rank_ws(h, count);
    if (rc) return rc;
    uint64_t *ka = h->d_key_a, *kb = h->d_key_b;
    uint32_t *pa = h->d_val_a, *pb = h->d_val_b;
    const int g = nblk(count);
    size_t tb = h->tmp_bytes;
    // 1) ascending by id
    hipLaunchKernelGGL(merge_idkey_kernel, dim3(g), dim3(256), 0, h->stream, count, d_ids, ka, pa);
    HIP_TRY(h, rocprim::radix_sort_pairs(h->d_tmp, tb, ka, kb, pa, pb, (size_t)count, 0, 64, h->stream));
    // 2) stable descending by the secondary key (first-sort order of the combined strategy)
    if (d_secondary) {
        hipLaunchKernelGGL(merge_gatherkey_kernel, dim3(g), dim3(256), 0, h->stream, count, d_secondary, pb, ka);
        tb = h->tmp_bytes;
        HIP_TRY(h, rocprim::radix_sort_pairs_desc(h->d_tmp, tb, ka, kb, pb, pa, (size_t)count, 0, 64, h->stream));
        uint32_t *t = pa; pa = pb; pb = t;
    }
    // 3) stable descending by score
    hipLaunchKernelGGL(merge_gatherkey_kernel, dim3(g), dim3(256), 0, h->stream, count, d_scores, pb, ka);
    tb = h->tmp_bytes;
    HIP_TRY(h, rocprim::radix_sort_pairs_desc(h->d_tmp, tb, ka, kb, pb, pa, (size_t)count, 0, 64, h->stream));
    const int64_t w = count < max_out ? count : max_out;
    hipLaunchKernelGGL(merge_emit_kernel, dim3(nblk(w)), dim3(256), 0, h->stream, w, pa, d_scores, d_ids,
                       d_score_out, d_id_out);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int gather_scores_on_device(sdpcut_ctx *h, int64_t count, const int64_t *d_ids, double *d_eig_out, double *d_obj_out)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(gather_scores_kernel, dim3(nblk(count)), dim3(256), 0, h->stream, count, h->base, h->N, d_ids,
                       h->d_eig, h->d_obj, d_eig_out, d_obj_out);
    HIP_TRY(h, hipGetLastError());
    return 0;
}
