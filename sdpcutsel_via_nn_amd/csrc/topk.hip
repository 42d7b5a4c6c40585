// Top-k selection on the device (k <= 8192) -- the fast path of the ranking.
//
// The reference sorts all N candidates (cut_select_qp.py:601, :653) but its caller consumes
// only the first sel_size <= 5000 (_SDP_CUTS_PER_ROUND_MAX, :37).  For such heads a full sort
// is wasted HBM traffic: this file finds the k-th largest key with an MSD radix select (eight
// 8-bit histogram passes over the u64 score images, candidates outside the class masked to
// key 0), compacts the k selected (key, index) pairs -- all keys above the threshold plus the
// lowest-index keys equal to it, exactly what a stable descending sort would keep -- and sorts
// them by (key desc, index asc) with one workgroup in LDS.  No host round trip between passes:
// every block re-derives the running prefix from the global histograms of the earlier passes.
#include "common.h"
#include "keys.h"

#define TK_THREADS 256
#define TK_MAXBLK 512
#define TK_MAXK 8192

enum { TK_MODE_FEAS = 1, TK_MODE_OPT = 2, TK_MODE_STRONG = 3 };

struct TopkWs {
    uint32_t hist[8][256];   // [digit 7..0 -> row 0..7][bin]
    int64_t counters[4];     // [0] class size  [1] nb_violated  [2] nb_positive  [3] unused
    uint32_t gt_counter;
    uint32_t pad;
    uint32_t blk_eq[TK_MAXBLK];
};

struct Resolved {
    uint64_t prefix;   // digits resolved so far, in place
    int64_t need;      // how many of the elements matching the prefix are still wanted
};

// Re-derive (prefix, need) after `done` passes (digits 7, 6, ...).  All 256 threads call it.
__device__ Resolved resolve_prefix(const TopkWs *ws, int done, int64_t k)
{
    __shared__ uint32_t suf[256];
    __shared__ uint64_t s_prefix;
    __shared__ int64_t s_need;
    const int t = threadIdx.x;
    int64_t need = k < ws->counters[0] ? k : ws->counters[0];
    uint64_t prefix = 0;
    for (int p = 0; p < done; ++p) {
        // suffix sums S[t] = sum_{b >= t} hist[p][b]
        __syncthreads();
        suf[t] = ws->hist[p][t];
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const uint32_t v = (t + off < 256) ? suf[t + off] : 0u;
            __syncthreads();
            suf[t] += v;
            __syncthreads();
        }
        const int64_t here = suf[t], above = (t < 255) ? suf[t + 1] : 0;
        if (here >= need && above < need) {     // exactly one bin satisfies this when need >= 1
            s_prefix = prefix | ((uint64_t)t << (8 * (7 - p)));
            s_need = need - above;
        }
        __syncthreads();
        if (need >= 1) {
            prefix = s_prefix;
            need = s_need;
        }
    }
    Resolved r;
    r.prefix = prefix;
    r.need = need;
    return r;
}

__device__ __forceinline__ uint64_t masked_key(int mode, double eig, double obj)
{
    if (mode == TK_MODE_OPT) return key_of(obj);
    const bool viol = eig < SDPCUT_NEG_EIGVAL;
    if (mode == TK_MODE_FEAS) return viol ? key_of(-eig) : 0ull;
    return (obj > 0.0 && viol) ? key_of(obj) : 0ull;
}

// pass 0: build the keys, histogram of digit 7, class / violated / positive counts
__global__ __launch_bounds__(TK_THREADS) void tk_keys_kernel(int mode, int64_t n, const double *eig, const double *obj,
                                                             uint64_t *keys, TopkWs *ws)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cnt[3];
    hist[threadIdx.x] = 0;
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t c_class = 0, c_viol = 0, c_pos = 0;
    for (int64_t i = (int64_t)blockIdx.x * TK_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * TK_THREADS) {
        const double e = eig ? eig[i] : 0.0, o = obj ? obj[i] : 0.0;
        const uint64_t key = masked_key(mode, e, o);
        keys[i] = key;
        atomicAdd(&hist[key >> 56], 1u);
        c_class += (mode == TK_MODE_OPT) ? 1u : (key != 0ull);
        c_viol += (eig != nullptr) && (e < SDPCUT_NEG_EIGVAL);
        c_pos += (obj != nullptr) && (o > 0.0);
    }
    atomicAdd(&cnt[0], c_class);
    atomicAdd(&cnt[1], c_viol);
    atomicAdd(&cnt[2], c_pos);
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&ws->hist[0][threadIdx.x], hist[threadIdx.x]);
    if (threadIdx.x < 3 && cnt[threadIdx.x])
        atomicAdd((unsigned long long *)&ws->counters[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

// pass p = 1..7: histogram of digit 7-p among the keys that match the prefix resolved so far
__global__ __launch_bounds__(TK_THREADS) void tk_hist_kernel(int p, int64_t n, int64_t k, const uint64_t *keys, TopkWs *ws)
{
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    const Resolved r = resolve_prefix(ws, p, k);     // contains __syncthreads
    if (r.need < 1) return;                          // uniform: empty class
    const int shift = 8 * (7 - p);
    for (int64_t i = (int64_t)blockIdx.x * TK_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * TK_THREADS) {
        const uint64_t key = keys[i];
        if (((key ^ r.prefix) >> (shift + 8)) == 0) atomicAdd(&hist[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&ws->hist[p][threadIdx.x], hist[threadIdx.x]);
}

// threshold known: collect keys above it (any order) and count the keys equal to it per block
__global__ __launch_bounds__(TK_THREADS) void tk_collect_kernel(int64_t n, int64_t k, int64_t chunk, const uint64_t *keys,
                                                                TopkWs *ws, uint64_t *sel_key, uint32_t *sel_idx)
{
    __shared__ uint32_t eq;
    if (threadIdx.x == 0) eq = 0;
    const Resolved r = resolve_prefix(ws, 8, k);
    if (r.need < 1) return;
    const uint64_t T = r.prefix;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    uint32_t my_eq = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += TK_THREADS) {
        const uint64_t key = keys[i];
        if (key > T) {
            const uint32_t slot = atomicAdd(&ws->gt_counter, 1u);
            sel_key[slot] = key;
            sel_idx[slot] = (uint32_t)i;
        }
        my_eq += (key == T);
    }
    if (my_eq) atomicAdd(&eq, my_eq);
    __syncthreads();
    if (threadIdx.x == 0) ws->blk_eq[blockIdx.x] = eq;
}

// keys equal to the threshold: the `need` lowest indices, placed behind the greater ones
__global__ __launch_bounds__(TK_THREADS) void tk_equal_kernel(int64_t n, int64_t k, int64_t chunk, const uint64_t *keys,
                                                              TopkWs *ws, uint64_t *sel_key, uint32_t *sel_idx)
{
    __shared__ uint32_t red[TK_THREADS];
    __shared__ uint32_t wave_cnt[TK_THREADS / 64];
    const Resolved r = resolve_prefix(ws, 8, k);
    if (r.need < 1) return;
    const uint64_t T = r.prefix;
    const int64_t k_eff = k < ws->counters[0] ? k : ws->counters[0];
    const int64_t greater = k_eff - r.need;
    // equal keys in the blocks before this one
    uint32_t part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += TK_THREADS) part += ws->blk_eq[b];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int off = TK_THREADS / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    int64_t base = red[0];
    if (base >= r.need || ws->blk_eq[blockIdx.x] == 0) return;   // uniform
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row = lo; row < hi; row += TK_THREADS) {
        const int64_t i = row + threadIdx.x;
        const bool is_eq = (i < hi) && (keys[i] == T);
        const unsigned long long m = __ballot(is_eq);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        uint32_t row_total = 0;
        for (int w = 0; w < TK_THREADS / 64; ++w) {
            if (w < wave) before += wave_cnt[w];
            row_total += wave_cnt[w];
        }
        const int64_t rank = base + before;
        if (is_eq && rank < r.need) {
            sel_key[greater + rank] = T;
            sel_idx[greater + rank] = (uint32_t)i;
        }
        base += row_total;
        __syncthreads();
        if (base >= r.need) break;   // uniform
    }
}

// one workgroup: bitonic sort of the selected pairs by (key desc, idx asc), then emit
__global__ __launch_bounds__(1024) void tk_sort_emit_kernel(int64_t k, int64_t base, double score_add, const TopkWs *ws,
                                                            const uint64_t *sel_key, const uint32_t *sel_idx,
                                                            int64_t *idx_out, double *score_out)
{
    __shared__ uint64_t sk[TK_MAXK];
    __shared__ uint32_t si[TK_MAXK];
    const int64_t k_eff = k < ws->counters[0] ? k : ws->counters[0];
    if (k_eff < 1) return;
    int P = 2;
    while (P < k_eff) P <<= 1;
    for (int t = threadIdx.x; t < P; t += 1024) {
        // ascending order on (~key, idx)  ==  descending key, ascending index; padding sorts last
        sk[t] = (t < k_eff) ? ~sel_key[t] : ~0ull;
        si[t] = (t < k_eff) ? sel_idx[t] : 0xffffffffu;
    }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += 1024) {
                const int pos = 2 * t - (t & (stride - 1));
                const int par = pos + stride;
                const bool up = (pos & size) == 0;
                const uint64_t ka = sk[pos], kb = sk[par];
                const uint32_t ia = si[pos], ib = si[par];
                const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                if (a_gt_b == up) {
                    sk[pos] = kb; sk[par] = ka;
                    si[pos] = ib; si[par] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < k_eff; t += 1024) {
        idx_out[t] = base + (int64_t)si[t];
        score_out[t] = score_of(~sk[t]) + score_add;
    }
}

// ------------------------------------------------------------------------------------------
int ensure_topk_ws(sdpcut_ctx *h)
{
    if (h->d_topk_ws) return 0;
    HIP_TRY(h, hipMalloc(&h->d_topk_ws, sizeof(TopkWs)));
    HIP_TRY(h, hipMalloc((void **)&h->d_sel_key, TK_MAXK * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc((void **)&h->d_sel_idx, TK_MAXK * sizeof(uint32_t)));
    return 0;
}

void free_topk_ws(sdpcut_ctx *h)
{
    (void)hipFree(h->d_topk_ws); (void)hipFree(h->d_sel_key); (void)hipFree(h->d_sel_idx);
    h->d_topk_ws = nullptr; h->d_sel_key = nullptr; h->d_sel_idx = nullptr;
}

// Head of a ranking by selection.  mode: 1 feasibility, 2 optimality, 3 strong class.  Writes
// min(k, class size) entries; class size and the violated / positive counts come back in cnt[0..2].
int topk_select_on_device(sdpcut_ctx *h, int mode, int64_t k, double score_add, int64_t *d_idx_out,
                          double *d_score_out, int64_t cnt[4])
{
    const int64_t n = h->N;
    if (k < 1 || k > TK_MAXK || n < 1) return sdpcut_fail(h, SDPCUT_EINVAL, "top-k select: k out of range");
    int rc = ensure_topk_ws(h);
    if (rc) return rc;
    rc = ensure_rank_ws(h, n);
    if (rc) return rc;
    TopkWs *ws = (TopkWs *)h->d_topk_ws;
    HIP_TRY(h, hipMemsetAsync(ws, 0, sizeof(TopkWs), h->stream));
    const double *eig = (h->scored & SDPCUT_EIG) ? h->d_eig : nullptr;
    const double *obj = (h->scored & SDPCUT_NN) ? h->d_obj : nullptr;
    int64_t nb = (n + TK_THREADS - 1) / TK_THREADS;
    const int grid = (int)(nb < TK_MAXBLK ? nb : TK_MAXBLK);
    hipLaunchKernelGGL(tk_keys_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, mode, n, eig, obj, h->d_key_a, ws);
    for (int p = 1; p < 8; ++p)
        hipLaunchKernelGGL(tk_hist_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, p, n, k, h->d_key_a, ws);
    int64_t chunk = (n + grid - 1) / grid;
    chunk = (chunk + TK_THREADS - 1) / TK_THREADS * TK_THREADS;
    hipLaunchKernelGGL(tk_collect_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, k, chunk, h->d_key_a, ws,
                       h->d_sel_key, h->d_sel_idx);
    hipLaunchKernelGGL(tk_equal_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, k, chunk, h->d_key_a, ws,
                       h->d_sel_key, h->d_sel_idx);
    hipLaunchKernelGGL(tk_sort_emit_kernel, dim3(1), dim3(1024), 0, h->stream, k, h->base, score_add, ws, h->d_sel_key,
                       h->d_sel_idx, d_idx_out, d_score_out);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(cnt, ws->counters, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return 0;
}
