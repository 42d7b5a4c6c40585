// Top-k selection on the device (k <= 16384) -- the fast path of the ranking.
//
// The reference sorts all N candidates (cut_select_qp.py:601, :653) but its caller consumes
// only the first sel_size <= 5000 (_SDP_CUTS_PER_ROUND_MAX, :37).  For such heads a full sort
// is wasted HBM traffic: this file finds the k-th largest key with an MSD radix select (eight
// 8-bit histogram passes over the u64 score images; candidates outside the class are masked
// to key 0), compacts the k selected (key, index) pairs -- every key above the threshold plus
// the lowest-index keys equal to it, exactly what a stable descending sort would keep -- and
// orders them by (key desc, index asc) with a counting sort spread over the chip.
//
// The passes stop as soon as the keys above the threshold bin plus the whole bin fit the sort
// buffers (<= 8192): that superset is compacted and sorted, and only its first k entries are
// emitted.  Spread-out scores need 2-3 of the 8 digits (the remaining launches return at once);
// masses of equal keys run to the last digit, where ties are cut by index.
//
// No host round trip anywhere: the last workgroup to finish a histogram pass (ticket counter)
// resolves that digit and publishes (prefix, need) for the next launch; the kernel boundary is
// the release/acquire.  Histogram cells and counters are written with device-scope atomics
// and read back by the resolving block with device-scope atomic loads (per-XCD L2s are not
// coherent for plain loads inside a launch).
//
// A round that scores for itself (sdpcut_select_round on a fresh point) does not run the first pass
// here: the score kernels count the leading digit of the class members' keys (score.hip, ScoreArgs::tk)
// and tk_refine_kernel<true> starts at the second digit, building the keys from the scores as it
// reads them -- tk_keys_kernel and the key array are for selections over scores that exist already.

#include "topk_dev.h"

// pass 0: build the keys, histogram of digit 7, class / violated / positive counts
__global__ __launch_bounds__(TK_THREADS) void tk_keys_kernel(int mode, int64_t sel, int64_t n, int64_t k, const double *eig,
                                                             const double *obj, uint64_t *keys, TopkWs *ws)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cnt[3];
    mode = resolve_mode(mode, ws, sel);      // uniform over the grid: counters[5] is final before this launch
    hist[threadIdx.x] = 0;
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t c_class = 0, c_viol = 0, c_pos = 0;
    const int64_t stride = (int64_t)gridDim.x * TK_THREADS;
    const int64_t rounds = (n + stride - 1) / stride;
    // TK_UNROLL rounds at a time with all loads issued first: a thread only has ~8 rounds, and one
    // dependent HBM round trip per round (~2 us) was the whole cost of the pass
    for (int64_t r0 = 0; r0 < rounds; r0 += TK_UNROLL) {
        double e[TK_UNROLL], o[TK_UNROLL];
        int64_t idx[TK_UNROLL];
#pragma unroll
        for (int u = 0; u < TK_UNROLL; ++u) {
            idx[u] = (r0 + u) * stride + (int64_t)blockIdx.x * TK_THREADS + threadIdx.x;
            const bool in = idx[u] < n;
            e[u] = (in && eig) ? eig[idx[u]] : 0.0;
            o[u] = (in && obj) ? obj[idx[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < TK_UNROLL; ++u) {
            const bool in = idx[u] < n;
            uint64_t key = 0;
            if (in) {
                key = masked_key(mode, e[u], o[u]);
                keys[idx[u]] = key;
                c_class += (mode == TK_MODE_OPT || mode == TK_MODE_COMBALL) ? 1u : (key != 0ull);
                c_viol += (eig != nullptr) && (e[u] < SDPCUT_NEG_EIGVAL);
                c_pos += (obj != nullptr) && (o[u] > 0.0);
            }
            hist_add(hist, (uint32_t)(key >> 56), in);
        }
    }
    if (c_class) atomicAdd(&cnt[0], c_class);
    if (c_viol) atomicAdd(&cnt[1], c_viol);
    if (c_pos) atomicAdd(&cnt[2], c_pos);
    __syncthreads();
    if (threadIdx.x < 3 && cnt[threadIdx.x])
        atomicAdd((unsigned long long *)&ws->counters[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st_i64(&ws->mode, mode);
        st_i64(&ws->counters[6], mode);
        st_i64(&ws->counters[5], strong_total(ws));      // for the host (round header)
    }
    finish_pass(ws, 0, k, hist, gridDim.x);
}

// pass p = 1..7: histogram of digit 7-p among the keys that match the prefix resolved so far
__global__ __launch_bounds__(TK_THREADS) void tk_hist_kernel(int p, int64_t n, int64_t k, const uint64_t *keys, TopkWs *ws)
{
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const TkState st = ws->state[p];
    if (st.stop) return;                              // uniform: selection already closed
    if (st.need >= 1) {                               // uniform
        const int shift = 8 * (7 - p);
        const int64_t stride = (int64_t)gridDim.x * TK_THREADS;
        const int64_t rounds = (n + stride - 1) / stride;
        for (int64_t r0 = 0; r0 < rounds; r0 += TK_UNROLL) {
            uint64_t key[TK_UNROLL];
            bool in[TK_UNROLL];
#pragma unroll
            for (int u = 0; u < TK_UNROLL; ++u) {
                const int64_t i = (r0 + u) * stride + (int64_t)blockIdx.x * TK_THREADS + threadIdx.x;
                in[u] = i < n;
                key[u] = in[u] ? keys[i] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < TK_UNROLL; ++u) {
                const bool match = in[u] && (((key[u] ^ st.prefix) >> (shift + 8)) == 0);
                hist_add(hist, (uint32_t)((key[u] >> shift) & 255), match);
            }
        }
        __syncthreads();
    }
    finish_pass(ws, p, k, hist, gridDim.x);
}

// bounded waits of the fused selection kernel (tk_refine_kernel): x s_sleep, a few milliseconds; a legitimate wait is
// tens of microseconds.  When a flag does not come (the GPU shared with a kernel that keeps workgroups of the grid from
// starting) counters[4] is raised, every workgroup leaves, and the host answers through a path without waits.
#define TK_SPIN_LIMIT (1 << 16)

// One-shot grid barrier `b` of a selection (its arrival counter starts at zero with the workspace).
// Every thread's device-scope atomics are drained before the workgroup arrives.  Bounded like the
// wait for a published state: if the other workgroups do not show up (the GPU shared with a kernel
// that keeps them from starting) counters[4] marks the selection void and everybody leaves.
static __device__ bool grid_barrier(TopkWs *ws, int b, uint32_t nblocks)
{
    __shared__ int bar_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ws->bar[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        uint32_t it = 0;
        while (ld_u32(&ws->bar[b]) < nblocks) {
            __builtin_amdgcn_s_sleep(4);
            const int64_t gone = ld_i64(&ws->counters[4]);      // 2: the selection has declared itself void (tie group): leave, keep the 2
            if (++it > TK_SPIN_LIMIT || gone) {
                if (!gone) st_i64(&ws->counters[4], 1);
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        bar_ok = ok;
    }
    __syncthreads();
    return bar_ok != 0;
}

// Passes 1..7, the count and the compaction in ONE launch (the fast path's replacement of
// tk_hist_kernel (x7) + tk_count_kernel + tk_write_kernel: launch hand-offs
// of ~5 us each become one or two grid barriers).  After pass 1 the selection is normally closed
// (early stop: every key >= T, at most TK_MAXK of them, in any order -- the sort that follows orders
// them): each workgroup counts its keys >= T, reserves its slice of the output with ONE fetch-add and
// writes.  Masses of equal keys run the remaining digits behind grid barriers and cut the last group by
// index, which needs the per-workgroup counts of all workgroups: one more barrier.
//
// ONFLY: the score kernels have already counted the leading digit (ScoreArgs::tk -> TopkWs::hist_score) and
// there is no key array: every workgroup resolves pass 0 for itself (same inputs, same result), and the
// keys are built from the scores as they are read -- once, into the LDS cache, when the chunk fits.
// Mode COMBAUTO resolved to COMBALL (fewer strong candidates than asked for -- the score kernels
// counted the STRONG keys): this launch runs its own pass 0 first, histogram in TopkWs::hist_alt.
#define TK_CACHE 4096      // keys of a workgroup's chunk kept in LDS between the passes (32 KB)
// (r5) DIRECT: the score / eigenvalue kernels also left the FINE histogram of the class (TopkWs::pf_tab, topk_dev.h).  Every
// workgroup resolves from it the window bin e* that holds the k-th largest key; if e* lies at or above the floor the producers
// published and the members at or above e* fit the sort buffers -- the usual case: 5000 .. 5100 of 10^6 -- they are compacted in
// ONE pass over the scores and handed to the sort exactly like an early stop of the digit passes: no histogram pass, no grid
// barrier, no wait.  pf_k = 0, a workgroup rich in head members, a fat bin, or the every-entry-visited regime: the passes below
// run as before.
template <bool ONFLY>
__global__ __launch_bounds__(TK_THREADS) void tk_refine_kernel(int64_t n, int64_t k, int64_t chunk, const uint64_t *keys,
                                                               TopkWs *ws, uint64_t *sel_key, uint32_t *sel_idx, int mode,
                                                               int64_t sel, const double *eig, const double *obj, int64_t pf_k,
                                                               unsigned long long *d_stats)
{
    __shared__ uint32_t hist[256];
    __shared__ int go;
    __shared__ uint32_t red_gt[TK_THREADS], red_eq[TK_THREADS], all_gt[TK_THREADS], all_eq[TK_THREADS];
    __shared__ uint32_t wave_cnt[TK_THREADS / 64];
    __shared__ uint32_t c_gt, c_eq, gt_local, c_above;
    __shared__ unsigned long long slice;
    __shared__ uint64_t cache[TK_CACHE];
    __shared__ TkState st1;                         // ONFLY: state after pass 0, resolved by this workgroup
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    const bool use_cache = chunk <= TK_CACHE;      // uniform: the chunk is read from memory once
    bool cached = false;
    int last_pass = -1;                             // last digit pass this launch ran (its histogram is still in LDS)
    int p_first = 1;
    bool direct = false;                            // (r5) resolved from the fine table: uniform over the grid
    __shared__ int pf_e, pf_floor_f;
    __shared__ int64_t pf_count;
    if (threadIdx.x == 0) c_above = 0;
    auto key_at = [&](int64_t i) -> uint64_t {
        if constexpr (ONFLY) return masked_key(mode, eig[i], obj[i]);     // (both valid: see the launch)
        else return keys[i];
    };
    // ONFLY: the scores of the first batch are requested before digit 0 is resolved (they do not depend on it)
    double pre_e[TK_UNROLL], pre_o[TK_UNROLL], pre_e2[TK_UNROLL], pre_o2[TK_UNROLL];
    uint32_t pf_q[2 * (PF_BINS / 1024)][4];      // (r5) this thread's words of the fine table, both replicas, and of the floor
    uint32_t pf_fl = 0;
    if constexpr (ONFLY) {
        if (pf_k > 0) {      // uniform; coalesced 16-byte loads, independent of everything else the kernel reads
            static_assert(PF_REP == 2 && PF_BINS % 1024 == 0, "two replicas of 1024-word blocks");
#pragma unroll
            for (int r = 0; r < PF_REP; ++r)
#pragma unroll
                for (int i = 0; i < PF_BINS / 1024; ++i) {
                    const uint4 q = *(const uint4 *)&ws->pf_tab[r][1024 * i + 4 * threadIdx.x];
                    pf_q[r * (PF_BINS / 1024) + i][0] = q.x; pf_q[r * (PF_BINS / 1024) + i][1] = q.y;
                    pf_q[r * (PF_BINS / 1024) + i][2] = q.z; pf_q[r * (PF_BINS / 1024) + i][3] = q.w;
                }
            if ((threadIdx.x & 63) < PF_FLOOR_REP) pf_fl = ws->pf_floor[threadIdx.x & 63][0];
        }
        if (lo < hi) {
            // (r5) only the measure the mode ranks by: a feasibility / optimality selection reads 8 bytes per candidate, not 16 -- the
            // scan of the scores is what this kernel waits for longest (phase stamps: table 3.2 us, scores 3.5-4.3 more)
            const double *m0 = mode == TK_MODE_OPT ? obj : eig;
            const bool two = mode != TK_MODE_OPT && mode != TK_MODE_FEAS;      // uniform
#pragma unroll
            for (int u = 0; u < TK_UNROLL; ++u) {
                const int64_t i = lo + (int64_t)u * TK_THREADS + threadIdx.x;
                const int64_t ic = i < hi ? i : hi - 1;
                pre_e[u] = m0[ic];
            }
            if (two) {
#pragma unroll
                for (int u = 0; u < TK_UNROLL; ++u) {
                    const int64_t i = lo + (int64_t)u * TK_THREADS + threadIdx.x;
                    const int64_t ic = i < hi ? i : hi - 1;
                    pre_o[u] = obj[ic];
                }
            } else {
#pragma unroll
                for (int u = 0; u < TK_UNROLL; ++u) pre_o[u] = pre_e[u];      // (masked_key looks at one of the two)
            }
            if (pf_k > 0 && use_cache) {      // uniform: the direct path reads its whole chunk (<= 4096 scores) without a second round trip
#pragma unroll
                for (int u = 0; u < TK_UNROLL; ++u) {
                    const int64_t i = lo + (int64_t)(TK_UNROLL + u) * TK_THREADS + threadIdx.x;
                    const int64_t ic = i < hi ? i : hi - 1;
                    pre_e2[u] = m0[ic];
                }
                if (two) {
#pragma unroll
                    for (int u = 0; u < TK_UNROLL; ++u) {
                        const int64_t i = lo + (int64_t)(TK_UNROLL + u) * TK_THREADS + threadIdx.x;
                        const int64_t ic = i < hi ? i : hi - 1;
                        pre_o2[u] = obj[ic];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < TK_UNROLL; ++u) pre_o2[u] = pre_e2[u];
                }
            }
        }
        const bool both = mode == TK_MODE_COMBAUTO;
        // (everything read here was written by earlier launches: plain loads)
        int64_t strong = 0;
#pragma unroll
        for (int r = 0; r < TK_SREP; ++r) strong += ws->strong_rep[r];
        if (both) mode = strong >= sel ? TK_MODE_STRONG : TK_MODE_COMBALL;      // uniform over the grid
        int64_t nviol = 0, npos = 0;      // counted by the score / eigenvalue kernels, replicated by workgroup
#pragma unroll 4
        for (int r = 0; r < TK_SHREP; ++r) { nviol += ws->viol_rep[r]; npos += ws->pos_rep[r]; }
        const int64_t cls = (mode == TK_MODE_OPT || mode == TK_MODE_COMBALL) ? n
                            : (mode == TK_MODE_FEAS) ? nviol : strong;
        if (threadIdx.x == 0 && blockIdx.x == 0) {
            st_i64(&ws->counters[1], nviol);
            st_i64(&ws->counters[2], npos);
            st_i64(&ws->mode, mode);
            st_i64(&ws->counters[6], mode);
            st_i64(&ws->counters[5], strong);      // for the host (round header)
            st_i64(&ws->counters[0], cls);
        }
        if (both && mode == TK_MODE_COMBALL) {      // uniform over the grid
            p_first = 0;
            if (threadIdx.x == 0) { st1.prefix = 0; st1.need = k < cls ? k : cls; st1.stop = 0; }
        } else {
            if (pf_k > 0 && mode != TK_MODE_COMBALL) {
                // ---- the fine table (requested at the top of the kernel, in memory order: two replicas x 2048 words, consecutive
                // bins in consecutive lines) goes through LDS into bin order -- the key cache is not in use yet --; thread t then owns
                // bins 8 t .. 8 t + 7, suffix sums from the top.
                __shared__ uint32_t pf_wtot[TK_THREADS / 64];
                uint32_t *nat = (uint32_t *)cache;
                const int t = threadIdx.x, ln = t & 63, wv = t >> 6;
#pragma unroll
                for (int i = 0; i < PF_BINS / 1024; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int w = 1024 * i + 4 * t + j;      // word of a replica: bin (w % 32) * 64 + w / 32
                        nat[(w & 31) * 64 + (w >> 5)] = pf_q[i][j] + pf_q[PF_BINS / 1024 + i][j];
                    }
                uint32_t fl = pf_fl;
                for (int off = 8; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)fl, off); fl = o > fl ? o : fl; }
                if (t == 0) { pf_e = -1; pf_count = 0; pf_floor_f = (int)fl; }
                __syncthreads();
                constexpr int PER = PF_BINS / TK_THREADS;      // 8
                uint32_t hf[PER], mine8 = 0;
#pragma unroll
                for (int j = 0; j < PER; ++j) { hf[j] = nat[PER * t + j]; mine8 += hf[j]; }
                uint32_t v = mine8;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_down((int)v, off);
                    if (ln + off < 64) v += o;
                }
                if (ln == 0) pf_wtot[wv] = v;
                __syncthreads();
                for (int w = wv + 1; w < TK_THREADS / 64; ++w) v += pf_wtot[w];
                const int64_t need = k < cls ? k : cls;
                int64_t above = (int64_t)(v - mine8);
#pragma unroll
                for (int j = PER - 1; j >= 0; --j) {
                    const int64_t here = above + (int64_t)hf[j];
                    if (need >= 1 && here >= need && above < need) { pf_e = PER * t + j; pf_count = here; }      // one bin of one thread
                    above = here;
                }
                __syncthreads();
                const int64_t maxk = k <= TK_LDSK ? TK_LDSK : TK_MAXK;
                // e* at or above the floor: every workgroup reported every bin from e* up, the counts there are exact and the members
                // there are all of the class's members with such keys.  (Below the floor some workgroup kept members to itself: the
                // table undercounts, e* would lie too low -- never trusted.)
                direct = pf_e >= 0 && pf_e >= pf_floor_f && pf_count <= maxk;
                if (d_stats && blockIdx.x == 0 && threadIdx.x == 0) {      // what the last selection saw (sdpcut_get_stat, diagnostics)
                    d_stats[1] = (unsigned long long)(long long)pf_e;
                    d_stats[2] = (unsigned long long)(long long)pf_floor_f;
                    d_stats[3] = (unsigned long long)pf_count;
                }
                if (direct) {
                    const bool all_members = mode == TK_MODE_OPT;
                    const uint64_t edge = pf_edge(pf_e, mode == TK_MODE_FEAS);
                    if (threadIdx.x == 0) {
                        st1.prefix = edge > 0ull || all_members ? edge : 1ull;      // (key 0 = not in the class)
                        st1.need = 1;
                        st1.stop = 1;
                        if (blockIdx.x == 0) {
                            st_i64(&ws->counters[3], need);      // k_eff for the sort
                            if (d_stats) atomicAdd(&d_stats[0], 1ull);
                        }
                    }
                }
            }
            if (!direct) resolve_digit(ws, 0, k, ws->hist_score, cls, &st1, blockIdx.x == 0, mode, true, TK_SHREP);
        }
        __syncthreads();
    }
    TkState st;
    if constexpr (ONFLY) {
        if (direct && use_cache) {      // uniform over the grid
            // ---- (r5) the whole chunk is the two batches requested at the top of the kernel: keys and membership stay in registers,
            // ONE scan over the workgroup gives every thread its offset and the workgroup its count, one returning atomic reserves the
            // slice, the members are written.  (The general path below counts, reserves, then re-reads its keys row by row with an
            // LDS atomic per row: 16 dependent LDS round trips, 2.5 us of this kernel's 15 by its phase stamps.)
            const uint64_t T0 = st1.prefix;
            const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
            uint64_t dk[2 * TK_UNROLL];
            uint32_t mask = 0;
#pragma unroll
            for (int u = 0; u < 2 * TK_UNROLL; ++u) {
                const int64_t i = lo + (int64_t)u * TK_THREADS + threadIdx.x;
                const bool in = i < hi;
                dk[u] = in ? masked_key(mode, u < TK_UNROLL ? pre_e[u % TK_UNROLL] : pre_e2[u % TK_UNROLL],
                                        u < TK_UNROLL ? pre_o[u % TK_UNROLL] : pre_o2[u % TK_UNROLL]) : 0ull;
                mask |= (uint32_t)(in && dk[u] >= T0) << u;
            }
            const uint32_t cnt = (uint32_t)__popc(mask);
            uint32_t incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
                if (ln >= off) incl += o;
            }
            if (ln == 63) wave_cnt[wv] = incl;
            __syncthreads();
            uint32_t wbase = 0, total = 0;
#pragma unroll
            for (int w = 0; w < TK_THREADS / 64; ++w) {
                if (w < wv) wbase += wave_cnt[w];
                total += wave_cnt[w];
            }
            if (total == 0) return;      // uniform per workgroup
            if (threadIdx.x == 0)
                slice = __hip_atomic_fetch_add((unsigned long long *)&ws->n_sel, (unsigned long long)total, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            int64_t slot = (int64_t)slice + wbase + incl - cnt;
#pragma unroll
            for (int u = 0; u < 2 * TK_UNROLL; ++u) {
                if ((mask >> u) & 1u) {
                    if (slot < (int64_t)TK_MAXK) {
                        sel_key[slot] = dk[u];
                        sel_idx[slot] = (uint32_t)(lo + (int64_t)u * TK_THREADS + threadIdx.x);
                    } else {
                        st_i64(&ws->counters[4], 1);      // cannot happen (the table is exact from e* up): the selection is void, the host's general path answers
                    }
                    ++slot;
                }
            }
            return;
        }
    }
    if (direct) {
        // ---- one pass: the keys of this chunk (into the LDS cache when they fit), how many of them lie at or above the edge
        st = st1;
        const uint64_t T0 = st.prefix;
        uint32_t my = 0;
        if (threadIdx.x == 0) c_gt = 0;
        __syncthreads();
        if constexpr (ONFLY) {
            for (int64_t r0 = lo; r0 < hi; r0 += (int64_t)TK_UNROLL * TK_THREADS) {
                double e[TK_UNROLL], o[TK_UNROLL];
                bool in[TK_UNROLL];
                if (r0 == lo) {      // uniform: the batch requested at the top of the kernel
#pragma unroll
                    for (int u = 0; u < TK_UNROLL; ++u) {
                        in[u] = lo + (int64_t)u * TK_THREADS + threadIdx.x < hi;
                        e[u] = pre_e[u];
                        o[u] = pre_o[u];
                    }
                } else if (use_cache && r0 == lo + (int64_t)TK_UNROLL * TK_THREADS) {      // uniform: the second batch, requested there as well
#pragma unroll
                    for (int u = 0; u < TK_UNROLL; ++u) {
                        in[u] = r0 + (int64_t)u * TK_THREADS + threadIdx.x < hi;
                        e[u] = pre_e2[u];
                        o[u] = pre_o2[u];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < TK_UNROLL; ++u) {
                        const int64_t i = r0 + (int64_t)u * TK_THREADS + threadIdx.x;
                        in[u] = i < hi;
                        const int64_t ic = in[u] ? i : hi - 1;
                        e[u] = eig[ic];
                        o[u] = obj[ic];
                    }
                }
#pragma unroll
                for (int u = 0; u < TK_UNROLL; ++u) {
                    const int64_t i = r0 + (int64_t)u * TK_THREADS + threadIdx.x;
                    const uint64_t key = in[u] ? masked_key(mode, e[u], o[u]) : 0ull;
                    if (use_cache && in[u]) cache[i - lo] = key;
                    my += in[u] && key >= T0;
                }
            }
        }
        cached = use_cache;
        for (int off = 32; off > 0; off >>= 1) my += __shfl_xor((int)my, off);
        if ((threadIdx.x & 63) == 0 && my) atomicAdd(&c_gt, my);
        __syncthreads();
    }
    for (int p = p_first; !direct; ++p) {
        if (p > p_first) {        // state[p] is published inside this launch
            if (threadIdx.x == 0) {
                int ok = 1;
                uint32_t it = 0;
                while (ld_u32(&ws->ready[p]) == 0u) {
                    __builtin_amdgcn_s_sleep(4);
                    const int64_t gone = ld_i64(&ws->counters[4]);
                    if (++it > TK_SPIN_LIMIT || gone) {
                        if (!gone) st_i64(&ws->counters[4], 1);
                        ok = 0;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // pairs with the release in front of the flag (finish_pass)
                go = ok;
            }
            __syncthreads();
            if (!go) return;
        }
        if (ONFLY && p == p_first) {
            st = st1;
        } else {
            st.prefix = (uint64_t)ld_i64((const int64_t *)&ws->state[p].prefix);
            st.need = ld_i64(&ws->state[p].need);
            st.stop = ld_i64(&ws->state[p].stop);
        }
        if (st.stop || st.need < 1 || p == 8) break;      // uniform over the grid
        hist[threadIdx.x] = 0;
        if (threadIdx.x == 0) c_above = 0;
        __syncthreads();
        const int shift = 8 * (7 - p);
        uint32_t above = 0;                               // keys of this chunk beyond the prefix' range
        for (int64_t r0 = lo; r0 < hi; r0 += (int64_t)TK_UNROLL * TK_THREADS) {
            uint64_t key[TK_UNROLL];
            bool in[TK_UNROLL];
            if (ONFLY && !cached) {      // uniform
                // all 2 x TK_UNROLL score loads are issued before the first key is built (unconditional,
                // from a clamped position: a load inside a branch is waited for on the spot)
                double e[TK_UNROLL], o[TK_UNROLL];
                if (r0 == lo) {      // uniform: the batch requested at the top of the kernel
#pragma unroll
                    for (int u = 0; u < TK_UNROLL; ++u) {
                        in[u] = lo + (int64_t)u * TK_THREADS + threadIdx.x < hi;
                        e[u] = pre_e[u];
                        o[u] = pre_o[u];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < TK_UNROLL; ++u) {
                        const int64_t i = r0 + (int64_t)u * TK_THREADS + threadIdx.x;
                        in[u] = i < hi;
                        const int64_t ic = in[u] ? i : hi - 1;
                        e[u] = eig[ic];
                        o[u] = obj[ic];
                    }
                }
#pragma unroll
                for (int u = 0; u < TK_UNROLL; ++u) key[u] = in[u] ? masked_key(mode, e[u], o[u]) : 0ull;
            } else {
#pragma unroll
                for (int u = 0; u < TK_UNROLL; ++u) {
                    const int64_t i = r0 + (int64_t)u * TK_THREADS + threadIdx.x;
                    in[u] = i < hi;
                    key[u] = !in[u] ? 0ull : (cached ? cache[i - lo] : keys[i]);
                }
            }
#pragma unroll
            for (int u = 0; u < TK_UNROLL; ++u) {
                const int64_t i = r0 + (int64_t)u * TK_THREADS + threadIdx.x;
                if (use_cache && !cached && in[u]) cache[i - lo] = key[u];
                const uint64_t hi_part = p ? key[u] >> (shift + 8) : 0ull, pre_part = p ? st.prefix >> (shift + 8) : 0ull;
                const bool match = in[u] && hi_part == pre_part;
                above += in[u] && hi_part > pre_part;
                hist_add(hist, (uint32_t)((key[u] >> shift) & 255), match);
            }
        }
        cached = use_cache;
        for (int off = 32; off > 0; off >>= 1) above += __shfl_xor((int)above, off);
        if ((threadIdx.x & 63) == 0 && above) atomicAdd(&c_above, above);
        __syncthreads();
        last_pass = p;
        finish_pass(ws, p, k, hist, gridDim.x, true, p ? nullptr : ws->hist_alt);
        __syncthreads();
    }
    if (st.need < 1) return;                               // empty class: n_sel stays 0
    const uint64_t T = st.prefix;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (st.stop) {
        // early stop: every key >= T is wanted, in any order (the sort that follows orders them)
        uint32_t mine;
        if (last_pass >= 0) {
            // closed by the pass just run: this chunk's count is in its own histogram -- the keys beyond
            // the prefix' range plus the bins from the threshold bin up -- no counting pass over the keys
            const int shift = 8 * (7 - last_pass);
            const uint32_t tbin = (uint32_t)((T >> shift) & 255);
            uint32_t part = (threadIdx.x >= tbin) ? hist[threadIdx.x] : 0u;
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor((int)part, off);
            if (threadIdx.x == 0) c_gt = 0;
            __syncthreads();
            if (lane == 0 && part) atomicAdd(&c_gt, part);
            __syncthreads();
            mine = c_gt + c_above;
        } else if (direct) {
            mine = c_gt;                                   // counted by the direct pass above
        } else {
            if (threadIdx.x == 0) c_gt = 0;
            __syncthreads();
            uint32_t my = 0;
            for (int64_t i = lo + threadIdx.x; i < hi; i += TK_THREADS) my += key_at(i) >= T;
            for (int off = 32; off > 0; off >>= 1) my += __shfl_xor((int)my, off);
            if (lane == 0 && my) atomicAdd(&c_gt, my);
            __syncthreads();
            mine = c_gt;
        }
        if (mine == 0) return;                             // uniform per workgroup
        if (threadIdx.x == 0) {
            gt_local = 0;
            slice = __hip_atomic_fetch_add((unsigned long long *)&ws->n_sel, (unsigned long long)mine, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int64_t base = (int64_t)slice;
        // (rows in batches of TK_UNROLL with all their loads issued first: a chunk that does not fit the LDS cache
        // -- 4.9e4 keys per workgroup on a 1.25e7-candidate shard -- would otherwise pay a trip to HBM per row)
        for (int64_t r0 = lo; r0 < hi; r0 += (int64_t)TK_UNROLL * TK_THREADS) {
            uint64_t kk[TK_UNROLL];
            if (ONFLY && !cached) {      // uniform
                double e[TK_UNROLL], o[TK_UNROLL];
#pragma unroll
                for (int u = 0; u < TK_UNROLL; ++u) {
                    const int64_t i = r0 + (int64_t)u * TK_THREADS + threadIdx.x;
                    const int64_t ic = i < hi ? i : hi - 1;
                    e[u] = eig[ic];
                    o[u] = obj[ic];
                }
#pragma unroll
                for (int u = 0; u < TK_UNROLL; ++u) kk[u] = masked_key(mode, e[u], o[u]);
            } else {
#pragma unroll
                for (int u = 0; u < TK_UNROLL; ++u) {
                    const int64_t i = r0 + (int64_t)u * TK_THREADS + threadIdx.x;
                    kk[u] = (i < hi) ? (cached ? cache[i - lo] : keys[i]) : 0ull;
                }
            }
#pragma unroll
            for (int u = 0; u < TK_UNROLL; ++u) {
                const int64_t i = r0 + (int64_t)u * TK_THREADS + threadIdx.x;
                if (r0 + (int64_t)u * TK_THREADS >= hi) break;      // uniform
                const uint64_t key = kk[u];
                const bool take = (i < hi) && key >= T;
                const unsigned long long m = __ballot(take);
                uint32_t wbase = 0;
                if (lane == 0 && m) wbase = atomicAdd(&gt_local, (uint32_t)__popcll(m));
                wbase = (uint32_t)__shfl((int)wbase, 0);
                if (take) {
                    const int64_t slot = base + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    sel_key[slot] = key;
                    sel_idx[slot] = (uint32_t)i;
                }
            }
        }
        return;
    }
    // ---- counts of this workgroup's chunk
    if (threadIdx.x == 0) { c_gt = 0; c_eq = 0; gt_local = 0; }
    __syncthreads();
    {
        uint32_t my_gt = 0, my_eq = 0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += TK_THREADS) {
            const uint64_t key = cached ? cache[i - lo] : key_at(i);
            my_gt += (key > T);
            my_eq += (key == T);
        }
        for (int off = 32; off > 0; off >>= 1) {
            my_gt += __shfl_xor((int)my_gt, off);
            my_eq += __shfl_xor((int)my_eq, off);
        }
        if ((threadIdx.x & 63) == 0) {
            if (my_gt) atomicAdd(&c_gt, my_gt);
            if (my_eq) atomicAdd(&c_eq, my_eq);
        }
    }
    __syncthreads();
    // ---- exact cut at the last digit: offsets from the counts of ALL workgroups (tk_write_kernel's scheme)
    if (threadIdx.x == 0) {
        __hip_atomic_store(&ws->blk_gt[blockIdx.x], c_gt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws->blk_eq[blockIdx.x], c_eq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_barrier(ws, 0, gridDim.x)) return;
    uint32_t pg = 0, pe = 0, tg = 0, te = 0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += TK_THREADS) {
        const uint32_t g = ld_u32(&ws->blk_gt[b]), e = ld_u32(&ws->blk_eq[b]);
        tg += g; te += e;
        if (b < (int)blockIdx.x) { pg += g; pe += e; }
    }
    red_gt[threadIdx.x] = pg;
    red_eq[threadIdx.x] = pe;
    all_gt[threadIdx.x] = tg;
    all_eq[threadIdx.x] = te;
    __syncthreads();
    for (int off = TK_THREADS / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            red_gt[threadIdx.x] += red_gt[threadIdx.x + off]; red_eq[threadIdx.x] += red_eq[threadIdx.x + off];
            all_gt[threadIdx.x] += all_gt[threadIdx.x + off]; all_eq[threadIdx.x] += all_eq[threadIdx.x + off];
        }
        __syncthreads();
    }
    const int64_t base_gt = red_gt[0];
    int64_t base_eq = red_eq[0];
    const int64_t greater = all_gt[0];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        st_i64(&ws->n_sel, greater + (st.need < (int64_t)all_eq[0] ? st.need : (int64_t)all_eq[0]));
    const bool want_gt = c_gt != 0;
    const bool want_eq = c_eq != 0 && base_eq < st.need;
    if (!want_gt && !want_eq) return;     // uniform
    for (int64_t row = lo; row < hi; row += TK_THREADS) {
        const int64_t i = row + threadIdx.x;
        const uint64_t key = (i < hi) ? (cached ? cache[i - lo] : key_at(i)) : 0ull;
        if (i < hi && key > T) {
            const int64_t slot = base_gt + atomicAdd(&gt_local, 1u);
            sel_key[slot] = key;
            sel_idx[slot] = (uint32_t)i;
        }
        if (want_eq) {     // uniform
            const bool is_eq = (i < hi) && (key == T);
            const unsigned long long m = __ballot(is_eq);
            if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            uint32_t row_total = 0;
            for (int w = 0; w < TK_THREADS / 64; ++w) {
                if (w < wave) before += wave_cnt[w];
                row_total += wave_cnt[w];
            }
            const int64_t rank = base_eq + before;
            if (is_eq && rank < st.need) {
                sel_key[greater + rank] = T;
                sel_idx[greater + rank] = (uint32_t)i;
            }
            base_eq += row_total;
            __syncthreads();
        }
    }
}

// threshold known: per block (contiguous chunk of the index space) count the keys above it and
// the keys equal to it -- no global atomics, the write pass derives its offsets from these
__global__ __launch_bounds__(TK_THREADS) void tk_count_kernel(int64_t n, int64_t chunk, const uint64_t *keys, TopkWs *ws)
{
    __shared__ uint32_t c_gt, c_eq;
    if (threadIdx.x == 0) { c_gt = 0; c_eq = 0; }
    __syncthreads();
    const TkState st = ws->state[8];
    if (st.need < 1) return;
    const uint64_t T = st.prefix;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    uint32_t my_gt = 0, my_eq = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += TK_THREADS) {
        const uint64_t key = keys[i];
        my_gt += (key > T);
        my_eq += (key == T);
    }
    if (my_gt) atomicAdd(&c_gt, my_gt);
    if (my_eq) atomicAdd(&c_eq, my_eq);
    __syncthreads();
    if (threadIdx.x == 0) { ws->blk_gt[blockIdx.x] = c_gt; ws->blk_eq[blockIdx.x] = c_eq; }
}

// write pass: keys above the threshold go to slots [0, greater) (order inside a block is free,
// the final sort fixes it); of the keys equal to it the `need` lowest indices follow
__global__ __launch_bounds__(TK_THREADS) void tk_write_kernel(int64_t n, int64_t chunk, const uint64_t *keys, TopkWs *ws,
                                                              uint64_t *sel_key, uint32_t *sel_idx)
{
    __shared__ uint32_t red_gt[TK_THREADS], red_eq[TK_THREADS], all_gt[TK_THREADS], all_eq[TK_THREADS];
    __shared__ uint32_t wave_cnt[TK_THREADS / 64];
    __shared__ uint32_t gt_local;
    const TkState st = ws->state[8];
    if (st.need < 1) return;
    const uint64_t T = st.prefix;
    uint32_t pg = 0, pe = 0, tg = 0, te = 0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += TK_THREADS) {
        const uint32_t g = ws->blk_gt[b], e = ws->blk_eq[b];
        tg += g; te += e;
        if (b < (int)blockIdx.x) { pg += g; pe += e; }
    }
    red_gt[threadIdx.x] = pg;
    red_eq[threadIdx.x] = pe;
    all_gt[threadIdx.x] = tg;
    all_eq[threadIdx.x] = te;
    if (threadIdx.x == 0) gt_local = 0;
    __syncthreads();
    for (int off = TK_THREADS / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            red_gt[threadIdx.x] += red_gt[threadIdx.x + off]; red_eq[threadIdx.x] += red_eq[threadIdx.x + off];
            all_gt[threadIdx.x] += all_gt[threadIdx.x + off]; all_eq[threadIdx.x] += all_eq[threadIdx.x + off];
        }
        __syncthreads();
    }
    const int64_t base_gt = red_gt[0];
    int64_t base_eq = red_eq[0];
    const int64_t greater = all_gt[0];                 // keys above the threshold, over all blocks
    if (blockIdx.x == 0 && threadIdx.x == 0) ws->n_sel = greater + (st.need < (int64_t)all_eq[0] ? st.need : (int64_t)all_eq[0]);
    const bool want_gt = ws->blk_gt[blockIdx.x] != 0;
    const bool want_eq = ws->blk_eq[blockIdx.x] != 0 && base_eq < st.need;
    if (!want_gt && !want_eq) return;     // uniform
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row = lo; row < hi; row += TK_THREADS) {
        const int64_t i = row + threadIdx.x;
        const uint64_t key = (i < hi) ? keys[i] : 0ull;
        if (i < hi && key > T) {
            const int64_t slot = base_gt + atomicAdd(&gt_local, 1u);
            sel_key[slot] = key;
            sel_idx[slot] = (uint32_t)i;
        }
        if (want_eq) {     // uniform
            const bool is_eq = (i < hi) && (key == T);
            const unsigned long long m = __ballot(is_eq);
            if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            uint32_t row_total = 0;
            for (int w = 0; w < TK_THREADS / 64; ++w) {
                if (w < wave) before += wave_cnt[w];
                row_total += wave_cnt[w];
            }
            const int64_t rank = base_eq + before;
            if (is_eq && rank < st.need) {
                sel_key[greater + rank] = T;
                sel_idx[greater + rank] = (uint32_t)i;
            }
            base_eq += row_total;
            __syncthreads();
        }
    }
}

// Final order of the selected pairs, (key desc, idx asc) = ascending composite (~key, idx):
//   1. tk_tilesort_kernel: bitonic sort of 512-entry tiles in LDS (one workgroup per tile);
//   2. tk_mergerank_kernel: every entry's final rank = its position in its own tile + the number
//      of entries preceding it in every other tile (binary searches over tiles staged in LDS;
//      composites are unique, so ranks are a permutation).
// ~k log k work instead of the k^2 of a counting sort, two short launches.
#define TK_TILE 512

// obj != NULL (mode COMBALL): equal keys are ordered by obj_improve descending before the index --
// the first stable sort of the reference (:601) under its second one (:625).  The scores are only
// fetched for equal keys (rare unless the point is degenerate); padding never reaches the fetch.
// The sort kernels are instantiated twice: TIE = false is the plain composite compare (a memory
// fetch and a branch inside the comparator cost the common modes 30 % of both kernels).
template <bool TIE>
__device__ __forceinline__ bool comp_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib, const double *obj)
{
    if constexpr (!TIE) {
        return ka < kb || (ka == kb && ia < ib);
    } else {
        if (ka != kb) return ka < kb;
        if (ia != 0xffffffffu && ib != 0xffffffffu) {
            const uint64_t oa = key_of(obj[ia]), ob = key_of(obj[ib]);
            if (oa != ob) return oa > ob;
        }
        return ia < ib;
    }
}

template <bool TIE>
__device__ __forceinline__ void tilesort_body(const TopkWs *ws, const uint64_t *sel_key, const uint32_t *sel_idx,
                                              uint64_t *tile_key, uint32_t *tile_idx, const double *obj, uint64_t *sk,
                                              uint32_t *si)
{
    const int k_eff = (int)ws->n_sel;              // compacted entries (a superset of the head after an early stop)
    const int lo = blockIdx.x * TK_TILE;
    if (lo >= k_eff) return;                       // uniform
    for (int t = threadIdx.x; t < TK_TILE; t += TK_THREADS) {
        const int j = lo + t;
        sk[t] = (j < k_eff) ? ~sel_key[j] : ~0ull;       // padding sorts last
        si[t] = (j < k_eff) ? sel_idx[j] : 0xffffffffu;
    }
    __syncthreads();
    for (int size = 2; size <= TK_TILE; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int t = threadIdx.x;
            const int pos = 2 * t - (t & (stride - 1));
            const int par = pos + stride;
            const bool up = (pos & size) == 0;
            const uint64_t ka = sk[pos], kb = sk[par];
            const uint32_t ia = si[pos], ib = si[par];
            if (comp_less<TIE>(kb, ib, ka, ia, obj) == up) {
                sk[pos] = kb; sk[par] = ka;
                si[pos] = ib; si[par] = ia;
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < TK_TILE; t += TK_THREADS) {
        tile_key[lo + t] = sk[t];
        tile_idx[lo + t] = si[t];
    }
}

// TIE: 0 plain composite compare, 1 obj_improve as tie key (mode COMBALL), 2 decided by the mode the
// selection resolved on the device (TK_MODE_COMBAUTO): one uniform branch at entry picks the
// specialised body, the comparators stay branch-free
template <int TIE>
__global__ __launch_bounds__(TK_THREADS) void tk_tilesort_kernel(const TopkWs *ws, const uint64_t *sel_key,
                                                                 const uint32_t *sel_idx, uint64_t *tile_key,
                                                                 uint32_t *tile_idx, const double *obj)
{
    __shared__ uint64_t sk[TK_TILE];
    __shared__ uint32_t si[TK_TILE];
    if (TIE == 1 || (TIE == 2 && ws->mode == TK_MODE_COMBALL))
        tilesort_body<true>(ws, sel_key, sel_idx, tile_key, tile_idx, obj, sk, si);
    else
        tilesort_body<false>(ws, sel_key, sel_idx, tile_key, tile_idx, obj, sk, si);
}

template <bool TIE>
__device__ __forceinline__ void mergerank_body(int64_t base, double score_add, const TopkWs *ws, const uint64_t *tile_key,
                                               const uint32_t *tile_idx, int64_t *idx_out, double *score_out,
                                               const double *obj, uint64_t *sk, uint32_t *si)
{
    const int n_sel = (int)ws->n_sel, k_eff = (int)ws->counters[3];
    if (blockIdx.x * TK_THREADS >= n_sel) return;   // uniform
    const int ntiles = (n_sel + TK_TILE - 1) / TK_TILE;
    // (a head of 5000: 11 tiles, 22 entries per thread; each batch is a round trip to L2)
    for (int j0 = 0; j0 < ntiles * TK_TILE; j0 += 24 * TK_THREADS) {      // 24 pairs in flight per thread: 12 tiles in ONE trip to L2
        uint64_t kk[24];
        uint32_t ii[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int j = j0 + u * TK_THREADS + threadIdx.x;
            const bool in = j < ntiles * TK_TILE;
            kk[u] = in ? tile_key[j] : 0ull;
            ii[u] = in ? tile_idx[j] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int j = j0 + u * TK_THREADS + threadIdx.x;
            if (j < ntiles * TK_TILE) { sk[j] = kk[u]; si[j] = ii[u]; }
        }
    }
    __syncthreads();
    const int e = blockIdx.x * TK_THREADS + threadIdx.x;      // position in the tiled array
    if (e >= ntiles * TK_TILE) return;
    const uint64_t ke = sk[e];
    const uint32_t ie = si[e];
    if (ie == 0xffffffffu && ke == ~0ull) return;             // padding
    const int te = e / TK_TILE;
    int rank = e - te * TK_TILE;
    // lower bound of e's composite inside every other tile (all composites are distinct).  (r5) The searches of FOUR tiles run
    // side by side, branch-free: a search is a chain of ten dependent LDS reads, and one after the other the up-to-15 chains were
    // a third of this kernel's time (11.6 -> 8.8 us).  Steps 256, 128 .. 1 count the entries below e among the first 511 of a
    // tile, one more comparison settles the 512-th.
    constexpr int MR_T = 4;
    // (the searches are bound by the LDS's throughput for scattered reads -- 8 or 16 at a time are no faster than 4, and they take
    // as long as the number of tiles searched says -- so e's own tile and tiles that do not exist are not searched: the other
    // ntiles - 1 tiles in batches of four, the last batch with its own trip count; branch-free inside a batch)
    auto search = [&](auto live_tag, const int t0) __attribute__((always_inline)) {
        constexpr int LIVE = decltype(live_tag)::value;
        int tb[LIVE], pos[LIVE];
#pragma unroll
        for (int u = 0; u < LIVE; ++u) {
            const int o = t0 + u;                    // the o-th OTHER tile
            tb[u] = (o < te ? o : o + 1) * TK_TILE;
            pos[u] = 0;
        }
#pragma unroll
        for (int step = TK_TILE / 2; step >= 1; step >>= 1) {
            uint64_t km[LIVE];
            uint32_t im[LIVE];
#pragma unroll
            for (int u = 0; u < LIVE; ++u) { km[u] = sk[tb[u] + pos[u] + step - 1]; im[u] = si[tb[u] + pos[u] + step - 1]; }
#pragma unroll
            for (int u = 0; u < LIVE; ++u) pos[u] += comp_less<TIE>(km[u], im[u], ke, ie, obj) ? step : 0;
        }
#pragma unroll
        for (int u = 0; u < LIVE; ++u) {
            const bool last = pos[u] == TK_TILE - 1 && comp_less<TIE>(sk[tb[u] + TK_TILE - 1], si[tb[u] + TK_TILE - 1], ke, ie, obj);
            rank += pos[u] + (last ? 1 : 0);
        }
    };
    const int others = ntiles - 1;      // uniform per workgroup
    int t0 = 0;
    for (; t0 + MR_T <= others; t0 += MR_T) search(std::integral_constant<int, MR_T>{}, t0);
    switch (others - t0) {
    case 3: search(std::integral_constant<int, 3>{}, t0); break;
    case 2: search(std::integral_constant<int, 2>{}, t0); break;
    case 1: search(std::integral_constant<int, 1>{}, t0); break;
    default: break;
    }
    if (rank >= k_eff) return;                                // superset entries beyond the head
    idx_out[rank] = base + (int64_t)ie;
    score_out[rank] = score_of(~ke) + score_add;
}

template <int TIE>
__global__ __launch_bounds__(TK_THREADS) void tk_mergerank_kernel(int64_t base, double score_add, const TopkWs *ws,
                                                                  const uint64_t *tile_key, const uint32_t *tile_idx,
                                                                  int64_t *idx_out, double *score_out, const double *obj,
                                                                  int64_t *rec_hdr, int64_t rec_count, int64_t rec_len)
{
    __shared__ uint64_t sk[TK_LDSK];
    __shared__ uint32_t si[TK_LDSK];
    if (rec_hdr) {
        // shard record (shard.hip): this launch also writes the 8-word header in front of the head
        // and pads the slots behind the entries it emits with (-inf, INT64_MAX); rec_len >= 0 is the
        // length of the shard's list when that is not the class size (optimality ranking)
        const int64_t g = (int64_t)blockIdx.x * TK_THREADS + threadIdx.x;
        const int64_t written = ws->counters[4] ? 0 : ws->counters[3];
        if (g < rec_count && g >= written) {
            score_out[g] = -__builtin_huge_val();
            idx_out[g] = 0x7fffffffffffffffLL;
        }
        if (g < 8)
            rec_hdr[g] = g == 0 ? (rec_len >= 0 ? rec_len : ws->counters[0]) : g <= 4 ? ws->counters[g] : 0;
    }
    if (TIE == 1 || (TIE == 2 && ws->mode == TK_MODE_COMBALL)) {
        // (device-resolved regime: BIG_M belongs to the strong class only, not to COMBALL's own scores)
        mergerank_body<true>(base, TIE == 2 ? 0.0 : score_add, ws, tile_key, tile_idx, idx_out, score_out, obj, sk, si);
    } else {
        mergerank_body<false>(base, score_add, ws, tile_key, tile_idx, idx_out, score_out, obj, sk, si);
    }
}

// Heads of 8193 .. 16384 entries: the composite (key, index) pairs of all tiles no longer fit LDS, the
// keys alone do (128 KB); an index is fetched from the tile array only where two keys are equal.
// raw: score_out receives the key's low 63 bits as a double (keys that are not score images: the
// triangle inequalities' (density, violation) composite).
template <bool TIE>
__global__ __launch_bounds__(TK_THREADS) void tk_mergerank_big_kernel(int64_t base, double score_add, const TopkWs *ws,
                                                                      const uint64_t *tile_key, const uint32_t *tile_idx,
                                                                      int64_t *idx_out, double *score_out, const double *obj,
                                                                      int raw, int64_t emit_limit, int64_t *rec_hdr,
                                                                      int64_t rec_count, int64_t rec_len)
{
    __shared__ uint64_t sk[TK_MAXK];
    if (rec_hdr) {      // shard record: header and padding, as in tk_mergerank_kernel
        const int64_t g = (int64_t)blockIdx.x * TK_THREADS + threadIdx.x;
        const int64_t written = ws->counters[4] ? 0 : ws->counters[3];
        if (g < rec_count && g >= written) {
            score_out[g] = -__builtin_huge_val();
            idx_out[g] = 0x7fffffffffffffffLL;
        }
        if (g < 8)
            rec_hdr[g] = g == 0 ? (rec_len >= 0 ? rec_len : ws->counters[0]) : g <= 4 ? ws->counters[g] : 0;
    }
    const int n_sel = (int)ws->n_sel, k_eff = (int)ws->counters[3];
    if (blockIdx.x * TK_THREADS >= n_sel) return;   // uniform
    const int ntiles = (n_sel + TK_TILE - 1) / TK_TILE;
    for (int j = threadIdx.x; j < ntiles * TK_TILE; j += TK_THREADS) sk[j] = tile_key[j];
    __syncthreads();
    const int e = blockIdx.x * TK_THREADS + threadIdx.x;
    if (e >= ntiles * TK_TILE) return;
    const uint64_t ke = sk[e];
    const uint32_t ie = tile_idx[e];
    if (ie == 0xffffffffu && ke == ~0ull) return;             // padding
    const int te = e / TK_TILE;
    int rank = e - te * TK_TILE;
    for (int t = 0; t < ntiles; ++t) {
        if (t == te) continue;
        int lo = 0, hi = TK_TILE;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const uint64_t km = sk[t * TK_TILE + mid];
            bool less = km < ke;
            if (km == ke) less = comp_less<TIE>(km, tile_idx[t * TK_TILE + mid], ke, ie, obj);
            lo = less ? mid + 1 : lo;
            hi = less ? hi : mid;
        }
        rank += lo;
    }
    if (rank >= k_eff || rank >= emit_limit) return;
    idx_out[rank] = base + (int64_t)ie;
    score_out[rank] = raw ? __longlong_as_double((long long)(~ke & 0x7fffffffffffffffull)) : score_of(~ke) + score_add;
}

// pass 0 over keys that already exist (key 0 = not in the class): leading-digit histogram and class size
__global__ __launch_bounds__(TK_THREADS) void tk_prekeys_kernel(int64_t n, int64_t k, const uint64_t *keys, TopkWs *ws)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cnt;
    hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    uint32_t c_class = 0;
    const int64_t stride = (int64_t)gridDim.x * TK_THREADS;
    const int64_t rounds = (n + stride - 1) / stride;
    for (int64_t r = 0; r < rounds; ++r) {          // every lane runs every round: hist_add is wave-cooperative
        const int64_t i = r * stride + (int64_t)blockIdx.x * TK_THREADS + threadIdx.x;
        const bool in = i < n;
        const uint64_t key = in ? keys[i] : 0ull;
        c_class += in && key != 0ull;
        hist_add(hist, (uint32_t)(key >> 56), in);
    }
    if (c_class) atomicAdd(&cnt, c_class);
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd((unsigned long long *)&ws->counters[0], (unsigned long long)cnt);
    if (blockIdx.x == 0 && threadIdx.x == 0) { st_i64(&ws->mode, TK_MODE_FEAS); st_i64(&ws->counters[6], TK_MODE_FEAS); }
    finish_pass(ws, 0, k, hist, gridDim.x);
}

// ------------------------------------------------------------------------------------------
int ensure_topk_ws(sdpcut_ctx *h)
{
    if (h->d_topk_ws) return 0;
    HIP_TRY(h, hipMalloc(&h->d_topk_ws, sizeof(TopkWs)));
    // first half: compacted selection, second half: the sorted tiles
    HIP_TRY(h, hipMalloc((void **)&h->d_sel_key, 2 * TK_MAXK * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc((void **)&h->d_sel_idx, 2 * TK_MAXK * sizeof(uint32_t)));
    // workgroups of the fused selection kernel the device holds at once (its grid barriers rely on it)
    int b_on = 0, b_off = 0;
    HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b_on, tk_refine_kernel<true>, TK_THREADS, 0));
    HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b_off, tk_refine_kernel<false>, TK_THREADS, 0));
    h->tk_coresident = (int64_t)(b_on < b_off ? b_on : b_off) * h->n_cu;
    if (h->tk_coresident < 1) h->fused_tail = false;      // (never on gfx950: 4 per CU by LDS) the launch-per-digit path has no waits
    return 0;
}

void free_topk_ws(sdpcut_ctx *h)
{
    (void)hipFree(h->d_topk_ws); (void)hipFree(h->d_topk_ws_alt); (void)hipFree(h->d_sel_key); (void)hipFree(h->d_sel_idx);
    h->d_topk_ws = nullptr; h->d_topk_ws_alt = nullptr; h->d_sel_key = nullptr; h->d_sel_idx = nullptr;
    h->topk_alt_clean = false;
}

int topk_alt_ws(sdpcut_ctx *h, uint64_t **ptr, int *words)
{
    static_assert(sizeof(TopkWs) % 8 == 0, "TopkWs is zeroed in 8-byte words");
    if (!h->d_topk_ws_alt) HIP_TRY(h, hipMalloc(&h->d_topk_ws_alt, sizeof(TopkWs)));
    *ptr = (uint64_t *)h->d_topk_ws_alt;
    // (r5) lists too short for the fine histogram (it sits at the end of the struct) zero only what lies in front of it: the epilogue
    // of a round with a handful of cuts is a handful of workgroups, and every word is a store on its critical path
    static_assert(offsetof(TopkWs, pf_floor) % 8 == 0 && offsetof(TopkWs, pf_tab) > offsetof(TopkWs, pf_floor), "fine histogram last");
    *words = (int)((h->N >= SDPCUT_PF_MIN_N ? sizeof(TopkWs) : offsetof(TopkWs, pf_floor)) / 8);
    return 0;
}

// Enqueue the selection of the head of a ranking (no host synchronisation).
// mode: 1 feasibility, 2 optimality, 3 strong class.  min(k, class size) entries are written.
// *d_counters_out receives the device address of {class size, nb_violated, nb_positive, k_eff}.
int topk_begin(sdpcut_ctx *h, void **ws_out, uint64_t **keys_out)
{
    int rc = ensure_topk_ws(h);
    if (rc) return rc;
    rc = ensure_key_ws(h, h->N);
    if (rc) return rc;
    if (h->topk_alt_clean && h->d_topk_ws_alt) {
        // the epilogue of the previous round zeroed the other workspace (stream-ordered): swap
        void *t = h->d_topk_ws; h->d_topk_ws = h->d_topk_ws_alt; h->d_topk_ws_alt = t;
        h->topk_alt_clean = false;
    } else {
        HIP_TRY(h, hipMemsetAsync(h->d_topk_ws, 0, sizeof(TopkWs), h->stream));
    }
    if (ws_out) *ws_out = h->d_topk_ws;
    if (keys_out) *keys_out = h->d_key_a;
    return 0;
}

int64_t *topk_strong_counter(void *ws) { return ((TopkWs *)ws)->strong_rep; }

// may the score kernels count the leading digit for a head of k entries (ScoreFuse, stage 3)?  Lists
// that fit the sort buffers whole skip the radix passes altogether (tk_small_kernel).
// Which lists the one-workgroup selection takes (measured, profiles/r04_short_list_selection.txt: fused rounds on lists of 64 ..
// 65 536 candidates, both paths alternating on one box).  Below ~3000 candidates sorting every class member costs less than
// selecting first (two tiles at most); above 8192 the multi-workgroup passes win again -- except in the combined strategy, whose
// tie-aware sort of a whole class is dearer: there 2048 .. 12 288.
static bool smallsel_range(int64_t n, int64_t k, bool comb)
{
    if (!TK_SMALLSEL || k > TK_LDSK || n > TK_SMALLSEL_N) return false;
    // (a head of more than a quarter of the list -- the 5000 of a 7899-candidate QCQP cover -- is not worth selecting: the
    // threshold lies deep in the list, five or six passes and the cut by index, 32 us where sorting the whole class takes 39 - 14)
    if (4 * k > n) return false;
    return comb ? (n > 2048 && n <= 12288) : (n > 3072 && n <= 8192);
}

// ... and with the sort inside (heads of one tile at most, no shard record to write behind the head): lists of at most 4096
// candidates -- ONE launch instead of three, 55 -> 47 us per combined round on 64 candidates, 68 -> 59 on 2048, 70 -> 66 on 4096;
// beyond that the four waves that sort lose to the tile-sort launch what the saved hand-offs gain (same evidence file)
static bool smallsort_range(const sdpcut_ctx *h, int64_t n, int64_t k)
{
    return TK_SMALLSEL && TK_SMALLSORT && k <= TK_TILE && n <= 4096 && !h->shard_rec;
}

bool topk_fuse_ok(const sdpcut_ctx *h, int64_t k, bool comb)
{
    const int64_t maxk = k <= TK_LDSK ? TK_LDSK : TK_MAXK;
    if (smallsel_range(h->N, k, comb) || smallsort_range(h, h->N, k)) return false;      // tk_smallsel_kernel builds its own keys: nothing to count
    return h->fused_tail && !h->coop_launch && k >= 1 && k <= TK_MAXK && h->N > maxk;
}

// Lists that fit the sort buffers whole (n <= 8192 -- most of the reference's BoxQP / QCQP instances)
// need no radix passes: ONE workgroup builds the keys, counts the class and compacts its members;
// the sort that follows orders all of them and emits the first k_eff.  Three launches instead of
// seven on the latency-bound end of the problem sizes.
#ifndef TK_SMALL_THREADS
#define TK_SMALL_THREADS 1024     // (one workgroup: a row of the list per 1024 candidates instead of 256)
#endif
__global__ __launch_bounds__(TK_SMALL_THREADS) void tk_small_kernel(int mode, int64_t sel, int64_t n, int64_t k, const double *eig,
                                                              const double *obj, TopkWs *ws, uint64_t *sel_key,
                                                              uint32_t *sel_idx)
{
    __shared__ uint32_t cnt[4];      // class members, violated, positive, next slot
    mode = resolve_mode(mode, ws, sel);
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // (r4) eight rows of loads in flight: ONE workgroup has nobody to hide a memory round trip behind, and a 7899-candidate cover
    // is 31 rows -- 23 us of dependent trips on the critical path of a QCQP round before, ~5 now
    const double *pe = eig ? eig : obj, *po = obj ? obj : eig;      // (a measure the mode does not use is never looked at)
    double pre_e[8], pre_o[8];
    for (int64_t i0 = 0; i0 < n; i0 += TK_SMALL_THREADS) {
        const int u = (int)((i0 / TK_SMALL_THREADS) & 7);
        if (u == 0) {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int64_t j = i0 + (int64_t)v * TK_SMALL_THREADS + threadIdx.x;
                const int64_t jc = j < n ? j : n - 1;
                pre_e[v] = pe[jc];
                pre_o[v] = po[jc];
            }
        }
        double e_u = pre_e[0], o_u = pre_o[0];
#pragma unroll
        for (int v = 1; v < 8; ++v) { e_u = (u == v) ? pre_e[v] : e_u; o_u = (u == v) ? pre_o[v] : o_u; }
        const int64_t i = i0 + threadIdx.x;
        const bool in = i < n;
        const double e = (in && eig) ? e_u : 0.0, o = (in && obj) ? o_u : 0.0;
        const uint64_t key = in ? masked_key(mode, e, o) : 0ull;
        const bool member = in && ((mode == TK_MODE_OPT || mode == TK_MODE_COMBALL) ? true : key != 0ull);
        const unsigned long long mm = __ballot(member);
        const unsigned long long mv = __ballot(in && eig != nullptr && e < SDPCUT_NEG_EIGVAL);
        const unsigned long long mp = __ballot(in && obj != nullptr && o > 0.0);
        uint32_t base = 0;
        if (lane == 0) {
            if (mm) base = atomicAdd(&cnt[3], (uint32_t)__popcll(mm));
            if (mv) atomicAdd(&cnt[1], (uint32_t)__popcll(mv));
            if (mp) atomicAdd(&cnt[2], (uint32_t)__popcll(mp));
        }
        base = (uint32_t)__shfl((int)base, 0);
        if (member) {
            const uint32_t slot = base + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
            sel_key[slot] = key;
            sel_idx[slot] = (uint32_t)i;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t cls = cnt[3];
        ws->counters[0] = cls;
        ws->counters[1] = cnt[1];
        ws->counters[2] = cnt[2];
        ws->counters[3] = k < cls ? k : cls;
        ws->n_sel = cls;
        ws->mode = mode;
        ws->counters[6] = mode;
        ws->counters[5] = strong_total(ws);
    }
}

// ------------------------------------------------------------------------------------------
// Short lists WITH a selection (r4, late).  tk_small_kernel hands EVERY class member to the sort: fine for a few hundred
// candidates, but a cover of 8192 with a head of 409 was sorted whole -- 16 tiles, and a merge in which every entry searches
// fifteen other tiles: tile sort 14.5 + merge 31 us of a 100 us round (a 1024-candidate list: 13 + 5.4) -- and lists of
// 8193 .. 16384 went through tk_refine_kernel with four workgroups (27 us of flag waits).  Here ONE workgroup of 1024 threads
// keeps the keys of n <= TK_SMALLSEL_N candidates in LDS, runs the MSD radix select over them -- the leading bytes all class
// members share are skipped, a pass is sixteen LDS rows at most -- and stops as soon as the keys above the threshold bin plus the
// bin fit the tiles the head needs anyway (a multiple of TK_TILE): that superset goes to the sort.  Same keys, same tie rules,
// same counters as the other paths: a tie group cut at the last digit is cut by index (lowest first); in the every-entry-visited
// regime, whose ties go by obj_improve, the whole group is taken, or -- if it does not fit the merge's LDS -- the selection is
// declared void with flag 2 and T / need left in state[8] for topk_tie_split, exactly like resolve_digit.
// one digit resolved by ONE wave (lanes own four bins each): suffix sums over the 256 bins, the bin that holds the need-th largest
// key, the early-stop decision.  hist is cleared for the next pass on the way.  (Called by wave 0 between two workgroup barriers.)
struct SmallSelState {
    uint64_t prefix;
    int need, stop, is_void, in_bin;
};
__device__ __forceinline__ void smallsel_resolve(uint32_t *hist, SmallSelState *st, int p, int k_eff, int cap, int group_max, bool comball,
                                                 TopkWs *ws)
{
    const int lane = threadIdx.x & 63;
    uint32_t h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = hist[4 * lane + j]; hist[4 * lane + j] = 0; }
    const uint32_t mine = h[0] + h[1] + h[2] + h[3];
    uint32_t v = mine;                       // inclusive suffix sum over the lanes
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_down((int)v, off);
        if (lane + off < 64) v += o;
    }
    const int need = st->need;
    const uint64_t prefix = st->prefix;
    const int shift = 8 * (7 - p);
    int above = (int)(v - mine);             // keys in the bins above this lane's four
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        const int here = above + (int)h[j];
        if (here >= need && above < need) {      // exactly one bin of one lane
            const int bin = 4 * lane + j;
            const uint64_t pre = prefix | ((uint64_t)bin << shift);
            const int in_bin = (int)h[j];
            const int superset = k_eff - (need - above) + in_bin;      // every key >= the bin's lowest value
            const bool whole_group = p == 7 && comball && in_bin > need - above;      // a tie group that is ordered by obj_improve
            if (whole_group && superset > group_max) {
                // more equal new scores at the threshold than the sort holds: void, flag 2; T and the number still wanted from
                // the group for topk_tie_split
                st->is_void = 1;
                ws->counters[4] = 2;
                ws->state[8].prefix = pre;
                ws->state[8].need = need - above;
                ws->state[8].stop = 0;
            }
            st->prefix = pre;
            st->in_bin = in_bin;
            if ((p < 7 && superset <= cap) || (p == 7 && comball && superset <= group_max)) {
                st->need = in_bin;       // the whole bin goes into the sort, which puts the wanted k_eff first
                st->stop = 1;
            } else {
                st->need = need - above;
            }
        }
        above = here;
    }
}

// SORT (heads of at most TK_TILE entries, the usual 5-10 % of a short list): the superset is at most one tile -- it stays in LDS,
// the workgroup sorts it (all sixteen waves reach every barrier, r5; bitonic, (key desc, [obj_improve desc,] index asc): tk_tilesort_kernel's network) and emit the head.
// The round's selection is ONE launch instead of three (tile sort and merge ranks have nothing left to do); an every-entry-visited
// tie group of more than TK_SORTMAX entries is declared void like a group beyond the merge's LDS in the other variant.
#define TK_SORTMAX 2048
template <bool TIE>
__device__ __forceinline__ void smallsel_sort(uint64_t *sk, uint32_t *si, int P, const double *obj)
{
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int q = threadIdx.x; q < (P >> 1); q += TK_SMALLSEL_THREADS) {
                const int pos = 2 * q - (q & (stride - 1));
                const int par = pos + stride;
                const bool up = (pos & size) == 0;
                const uint64_t ka = sk[pos], kb = sk[par];
                const uint32_t ia = si[pos], ib = si[par];
                if (comp_less<TIE>(kb, ib, ka, ia, obj) == up) {
                    sk[pos] = kb; sk[par] = ka;
                    si[pos] = ib; si[par] = ia;
                }
            }
            __syncthreads();      // (every wave of the workgroup reaches every barrier: P / 2 <= 1024 compare-exchanges, one per thread)
        }
    }
}

template <bool SORT>
__global__ __launch_bounds__(TK_SMALLSEL_THREADS) void tk_smallsel_kernel(int mode, int64_t sel, int n, int k, const double *eig,
                                                                          const double *obj, TopkWs *ws, uint64_t *sel_key,
                                                                          uint32_t *sel_idx, int64_t base, double score_add,
                                                                          int64_t *idx_out, double *score_out)
{
    constexpr int NT = TK_SMALLSEL_THREADS, NW = NT / 64, R = TK_SMALLSEL_N / NT;      // R rows of NT candidates at most
    __shared__ uint64_t surv[TK_SMALLSEL_N];     // keys that still match the prefix after the first pass
    __shared__ uint64_t sk[SORT ? TK_SORTMAX : 1];      // SORT: the superset, inverted keys (ascending composite order)
    __shared__ uint32_t si[SORT ? TK_SORTMAX : 1];
    const bool auto_mode = mode == TK_MODE_COMBAUTO;
    const int group_max = SORT ? TK_SORTMAX : TK_LDSK;
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cnt[5];                  // class members, violated, positive, next free slot of the compaction, survivors
    __shared__ uint32_t wave_cnt[NW][2];
    __shared__ uint64_t wave_and[NW], wave_or[NW];
    __shared__ SmallSelState st;
    __shared__ int s_p0;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef TK_SMALLSEL_TIMING      // debug build: phase stamps in 10 ns ticks, printed by thread 0
    unsigned long long ph[8]; int nph = 0;
#define SEL_STAMP() do { if (nph < 8) ph[nph++] = wall_clock64(); } while (0)
#else
#define SEL_STAMP()
#endif
    SEL_STAMP();
    // the scores of the first eight rows are requested before the mode is resolved (they do not depend on it, the strong count
    // is another trip to memory: ONE workgroup has nobody to hide either behind)
    const int rows = (n + TK_SMALLSEL_THREADS - 1) / TK_SMALLSEL_THREADS;          // uniform
    const double *pe = eig ? eig : obj, *po = obj ? obj : eig;      // a measure the mode does not use is never looked at
    double e[8], o[8];
    int64_t strong = 0;                          // (left by the score kernels of this round; requested first, returned first)
#pragma unroll
    for (int r = 0; r < TK_SREP; ++r) strong += ld_i64(&ws->strong_rep[r]);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = u * TK_SMALLSEL_THREADS + t;
        const int ic = i < n ? i : n - 1;
        if (u < rows) {      // uniform
            e[u] = pe[ic];
            o[u] = po[ic];
        }
    }
    if (mode == TK_MODE_COMBAUTO) mode = strong >= sel ? TK_MODE_STRONG : TK_MODE_COMBALL;
    const bool all_members = mode == TK_MODE_OPT || mode == TK_MODE_COMBALL;
    const bool comball = mode == TK_MODE_COMBALL;
    if (t < 5) cnt[t] = 0;
    if (t < 256) hist[t] = 0;
    if (t == 0) { st.prefix = 0; st.stop = 0; st.is_void = 0; st.in_bin = 0; s_p0 = 0; }
    __syncthreads();
    // ---- keys of this thread's candidates (row r: candidate r * NT + t) in registers; class size, violated, positive; the bits
    // all class members share
    uint64_t key[R];
    {
        uint32_t c_class = 0, c_viol = 0, c_pos = 0;
        uint64_t k_and = ~0ull, k_or = 0ull;
        // (eight rows of loads in flight per thread)
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 8) {
            if (r0 > 0 && r0 < rows) {      // uniform
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (r0 + u >= R) continue;      // (compile time)
                    const int i = (r0 + u) * NT + t;
                    const int ic = i < n ? i : n - 1;
                    e[u] = pe[ic];
                    o[u] = po[ic];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (r0 + u >= R) continue;          // (compile time)
                const int i = (r0 + u) * NT + t;
                uint64_t kk = 0ull;
                if (r0 + u < rows && i < n) {
                    const double ev = eig ? e[u] : 0.0, ov = obj ? o[u] : 0.0;
                    kk = masked_key(mode, ev, ov);
                    const bool member = all_members || kk != 0ull;
                    c_class += member;
                    k_and &= member ? kk : ~0ull;
                    k_or |= member ? kk : 0ull;
                    c_viol += (eig != nullptr) && (ev < SDPCUT_NEG_EIGVAL);
                    c_pos += (obj != nullptr) && (ov > 0.0);
                }
                key[r0 + u] = kk;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            c_class += __shfl_xor((int)c_class, off);
            c_viol += __shfl_xor((int)c_viol, off);
            c_pos += __shfl_xor((int)c_pos, off);
            k_and &= (uint64_t)__shfl_xor((long long)k_and, off);
            k_or |= (uint64_t)__shfl_xor((long long)k_or, off);
        }
        if (lane == 0) {
            if (c_class) atomicAdd(&cnt[0], c_class);
            if (c_viol) atomicAdd(&cnt[1], c_viol);
            if (c_pos) atomicAdd(&cnt[2], c_pos);
            wave_and[wave] = k_and;
            wave_or[wave] = k_or;
        }
    }
    __syncthreads();
    SEL_STAMP();      // [1] keys built
    const int cls = (int)cnt[0];
    const int k_eff = k < cls ? k : cls;
    // what the early stop may hand to the sort: the tiles the head needs anyway, never more than the merge's LDS holds
    int cap = (k_eff + TK_TILE - 1) / TK_TILE * TK_TILE;
    cap = cap < TK_LDSK ? cap : TK_LDSK;
    if (SORT) cap = TK_TILE;      // (k <= TK_TILE: the launch's condition)
    if (t == 0) {
        ws->counters[0] = cls;
        ws->counters[1] = cnt[1];
        ws->counters[2] = cnt[2];
        ws->counters[3] = k_eff;
        ws->counters[5] = strong;
        ws->counters[6] = mode;
        ws->mode = mode;
        st.need = k_eff;
        uint64_t a = ~0ull, o = 0ull;
        for (int w = 0; w < NW; ++w) { a &= wave_and[w]; o |= wave_or[w]; }
        // leading bytes every class member agrees on: nothing to select there (key images of scores of one sign and similar size
        // share two of their eight bytes or more).  A list with non-members (key 0) must not skip a prefix of zero bytes: the
        // passes tell members from non-members by it.
        int p0 = 0;
        const uint64_t diff = a ^ o;
        while (p0 < 7 && ((diff >> (8 * (7 - p0))) & 255ull) == 0ull) ++p0;
        if (cls == 0 || (!all_members && p0 > 0 && (a >> (8 * (8 - p0))) == 0ull)) p0 = 0;
        s_p0 = p0;
        st.prefix = p0 ? (a >> (8 * (8 - p0))) << (8 * (8 - p0)) : 0ull;
        if (cls <= cap) st.stop = 1;      // the whole class fits: no pass at all, every member goes to the sort (T = 0)
    }
    __syncthreads();
    if (k_eff == 0) return;      // (n_sel stays 0 with the zeroed workspace)
    // ---- MSD radix select: threshold key T (st.prefix) and how many of the keys equal to it are wanted.  The first pass runs over
    // all keys (registers) and leaves the keys of the threshold bin in LDS; the later ones run over those survivors only.
    int p = s_p0;
    SEL_STAMP();      // [2] header
    if (!st.stop) {
        const int shift = 8 * (7 - p);
        const uint64_t prefix = st.prefix;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r < rows) {      // uniform (every lane runs every row: hist_add is wave-cooperative)
                const int i = r * NT + t;
                const bool match = i < n && (p == 0 || ((key[r] ^ prefix) >> (shift + 8)) == 0);
                hist_add(hist, (uint32_t)((key[r] >> shift) & 255), match);
            }
        }
        __syncthreads();
        if (wave == 0) smallsel_resolve(hist, &st, p, k_eff, cap, group_max, comball, ws);
        __syncthreads();
        if (!st.stop && !st.is_void && p < 7) {      // uniform: survivors = the keys of the threshold bin
            const uint64_t pre = st.prefix;
            // (ONE reservation per wave for all its rows: an LDS atomic with a return value per row is a round trip per row)
            unsigned long long mrow[R];
            uint32_t wtot = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                mrow[r] = 0ull;
                if (r < rows) {      // uniform
                    const int i = r * NT + t;
                    mrow[r] = __ballot(i < n && ((key[r] ^ pre) >> shift) == 0);
                    wtot += (uint32_t)__popcll(mrow[r]);
                }
            }
            uint32_t wbase = 0;
            if (lane == 0 && wtot) wbase = atomicAdd(&cnt[4], wtot);
            wbase = (uint32_t)__shfl((int)wbase, 0);
            uint64_t s_and = ~0ull, s_or = 0ull;      // the bits the survivors share
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r < rows) {
                    if ((mrow[r] >> lane) & 1ull) {
                        surv[wbase + (uint32_t)__popcll(mrow[r] & ((1ull << lane) - 1ull))] = key[r];
                        s_and &= key[r];
                        s_or |= key[r];
                    }
                    wbase += (uint32_t)__popcll(mrow[r]);
                }
            }
            for (int off = 32; off > 0; off >>= 1) {
                s_and &= (uint64_t)__shfl_xor((long long)s_and, off);
                s_or |= (uint64_t)__shfl_xor((long long)s_or, off);
            }
            if (lane == 0) { wave_and[wave] = s_and; wave_or[wave] = s_or; }
            __syncthreads();
            // Structured LP points: the threshold bin is ONE value shared by hundreds or thousands of candidates (a QCQP cover
            // of 7899: eight passes to the last digit, 30 us).  The bytes all survivors share are skipped; if they share all of
            // them the selection is finished here -- the group is cut by index below, or, in the every-entry-visited regime, taken
            // whole / declared void exactly as the last digit would have done.
            if (t == 0) {
                uint64_t a = ~0ull, o = 0ull;
                for (int w = 0; w < NW; ++w) { a &= wave_and[w]; o |= wave_or[w]; }
                const uint64_t diff = a ^ o;
                const int ns = (int)cnt[4], need = st.need;
                int pn = p + 1;
                while (pn < 8 && ((diff >> (8 * (7 - pn))) & 255ull) == 0ull) ++pn;
                if (pn == 8) {      // one value
                    const int superset = k_eff - need + ns;
                    st.prefix = a;
                    st.in_bin = ns;
                    if (comball && ns > need) {
                        if (superset <= group_max) {
                            st.need = ns;
                            st.stop = 1;
                        } else {
                            st.is_void = 1;
                            ws->counters[4] = 2;
                            ws->state[8].prefix = a;
                            ws->state[8].need = need;
                            ws->state[8].stop = 0;
                        }
                    }
                } else {
                    st.prefix = (a >> (8 * (8 - pn))) << (8 * (8 - pn));
                }
                s_p0 = pn;
            }
            __syncthreads();
            p = s_p0;
        } else {
            ++p;
        }
    }
    SEL_STAMP();      // [3] first pass + survivors
    for (; p < 8 && !st.stop && !st.is_void; ++p) {
        const int shift = 8 * (7 - p);
        const int ns = (int)cnt[4];              // every survivor matches the prefix down to the digit of this pass
        for (int i0 = 0; i0 < ns; i0 += NT) {
            const int i = i0 + t;
            const uint64_t kk = i < ns ? surv[i] : 0ull;
            const bool match = i < ns && ((kk ^ st.prefix) >> (shift + 8)) == 0;
            hist_add(hist, (uint32_t)((kk >> shift) & 255), match);
        }
        __syncthreads();
        if (wave == 0) smallsel_resolve(hist, &st, p, k_eff, cap, group_max, comball, ws);
        __syncthreads();
    }
    if (st.is_void) return;
    SEL_STAMP();      // [4] later passes
    const uint64_t T = st.prefix;
    if (st.stop) {
        // ---- early stop: every class member >= T (the lowest value of the threshold bin), in any order -- the sort orders them
        unsigned long long mrow[R];
        uint32_t wtot = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            mrow[r] = 0ull;
            if (r < rows) {      // uniform
                const int i = r * NT + t;
                mrow[r] = __ballot(i < n && key[r] >= T && (all_members || key[r] != 0ull));
                wtot += (uint32_t)__popcll(mrow[r]);
            }
        }
        uint32_t wbase = 0;
        if (lane == 0 && wtot) wbase = atomicAdd(&cnt[3], wtot);      // (one reservation per wave)
        wbase = (uint32_t)__shfl((int)wbase, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r < rows) {
                if ((mrow[r] >> lane) & 1ull) {
                    const uint32_t slot = wbase + (uint32_t)__popcll(mrow[r] & ((1ull << lane) - 1ull));
                    if constexpr (SORT) {
                        sk[slot] = ~key[r];
                        si[slot] = (uint32_t)(r * NT + t);
                    } else {
                        sel_key[slot] = key[r];
                        sel_idx[slot] = (uint32_t)(r * NT + t);
                    }
                }
                wbase += (uint32_t)__popcll(mrow[r]);
            }
        }
        __syncthreads();
        if (t == 0) ws->n_sel = SORT ? k_eff : (int)cnt[3];
        if constexpr (SORT) {
            const int M = (int)cnt[3];
            int P = 64;
            while (P < M) P <<= 1;
            for (int j = M + t; j < P; j += NT) { sk[j] = ~0ull; si[j] = 0xffffffffu; }      // padding sorts last
            __syncthreads();
            if (comball) smallsel_sort<true>(sk, si, P, obj);
            else smallsel_sort<false>(sk, si, P, obj);
            const double add = (auto_mode && comball) ? 0.0 : score_add;      // (device-resolved regime: BIG_M belongs to the strong class only)
            for (int r = t; r < k_eff; r += NT) {
                idx_out[r] = base + (int64_t)si[r];
                score_out[r] = score_of(~sk[r]) + add;
            }
        }
#ifdef TK_SMALLSEL_TIMING
        SEL_STAMP();
        if (t == 0) printf("smallsel n %d k %d cls %d p0 %d last p %d n_sel %u: keys %llu header %llu pass1 %llu passes %llu compaction %llu (x 10 ns)\n", n, k, cls, s_p0, p, cnt[3],
                           ph[1] - ph[0], ph[2] - ph[1], ph[3] - ph[2], ph[4] - ph[3], ph[5] - ph[4]);
#endif
        return;
    }
    // ---- all eight digits used: every key above T and, of the keys equal to T, the `need` lowest indices (index order)
    const int need = st.need;
    int greater = 0;
    {
        uint32_t g = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = r * NT + t;
            g += r < rows && i < n && key[r] > T && (all_members || key[r] != 0ull);
        }
        for (int off = 32; off > 0; off >>= 1) g += __shfl_xor((int)g, off);
        if (lane == 0) wave_cnt[wave][0] = g;
        __syncthreads();
        for (int w = 0; w < NW; ++w) greater += (int)wave_cnt[w][0];
        __syncthreads();
    }
    // counts per (row, wave) in one table: the offsets of the index-ordered compaction need ONE barrier, not two per row
    __shared__ uint16_t tab_gt[R][NW], tab_eq[R][NW];
    unsigned long long mg[R], me[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mg[r] = me[r] = 0ull;
        if (r < rows) {      // uniform
            const int i = r * NT + t;
            const bool member = i < n && (all_members || key[r] != 0ull);
            mg[r] = __ballot(member && key[r] > T);
            me[r] = __ballot(member && key[r] == T);
            if (lane == 0) { tab_gt[r][wave] = (uint16_t)__popcll(mg[r]); tab_eq[r][wave] = (uint16_t)__popcll(me[r]); }
        }
    }
    __syncthreads();
    int base_gt = 0, base_eq = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r < rows) {
            int bg = base_gt, be = base_eq, tg = 0, te = 0;
            for (int w = 0; w < NW; ++w) {
                const int cg = tab_gt[r][w], ce = tab_eq[r][w];
                if (w < wave) { bg += cg; be += ce; }
                tg += cg;
                te += ce;
            }
            const int i = r * NT + t;
            if ((mg[r] >> lane) & 1ull) {
                const int slot = bg + __popcll(mg[r] & ((1ull << lane) - 1ull));
                if constexpr (SORT) { sk[slot] = ~key[r]; si[slot] = (uint32_t)i; }
                else { sel_key[slot] = key[r]; sel_idx[slot] = (uint32_t)i; }
            }
            if ((me[r] >> lane) & 1ull) {
                const int q = be + __popcll(me[r] & ((1ull << lane) - 1ull));
                if (q < need) {
                    if constexpr (SORT) { sk[greater + q] = ~key[r]; si[greater + q] = (uint32_t)i; }
                    else { sel_key[greater + q] = key[r]; sel_idx[greater + q] = (uint32_t)i; }
                }
            }
            base_gt += tg;
            base_eq += te;
        }
    }
    __syncthreads();
    const int M = greater + (base_eq < need ? base_eq : need);      // = k_eff
    if (t == 0) ws->n_sel = M;
#ifdef TK_SMALLSEL_TIMING
    SEL_STAMP();
    if (t == 0) printf("smallsel n %d k %d cls %d p0 %d survivors %u, exact cut, greater %d need %d: keys %llu header %llu pass1 %llu passes %llu compaction %llu (x 10 ns)\n", n, k, cls, s_p0,
                       cnt[4], greater, need, ph[1] - ph[0], ph[2] - ph[1], ph[3] - ph[2], ph[4] - ph[3], ph[5] - ph[4]);
#endif
    if constexpr (SORT) {
        int P = 64;
        while (P < M) P <<= 1;
        for (int j = M + t; j < P; j += NT) { sk[j] = ~0ull; si[j] = 0xffffffffu; }
        __syncthreads();
        // (the every-entry-visited regime gets here only with a tie group that is wanted whole: its members still go by obj_improve)
        if (comball) smallsel_sort<true>(sk, si, P, obj);
        else smallsel_sort<false>(sk, si, P, obj);
        const double add = (auto_mode && comball) ? 0.0 : score_add;
        for (int r = t; r < k_eff; r += NT) {
            idx_out[r] = base + (int64_t)si[r];
            score_out[r] = score_of(~sk[r]) + add;
        }
    }
}

#undef SEL_STAMP

// Everything behind pass 0: the remaining digit passes, the compaction, the sort and the ranks.
// n keys in h->d_key_a; k <= TK_MAXK; raw: see tk_mergerank_big_kernel (heads > TK_LDSK only).
// onfly: pass 0 was counted by the score kernels and there are no keys (tk_refine_kernel<true>; eig / obj /
// sel as for tk_keys_kernel).
static int topk_enqueue_after_pass0(sdpcut_ctx *h, TopkWs *ws, int mode, int64_t n, int64_t k, double score_add,
                                    int64_t *d_idx_out, double *d_score_out, bool compacted, int raw, int64_t base,
                                    int64_t emit_limit = TK_MAXK, bool onfly = false, int64_t sel = 0,
                                    const double *eig = nullptr, const double *obj = nullptr)
{
    if (onfly && (compacted || !h->fused_tail)) return sdpcut_fail(h, SDPCUT_ESTATE, "top-k select: no key pass to continue from");
    // Workgroups of the passes: one LDS-cached chunk of TK_CACHE keys each, as many as can be RESIDENT together -- the
    // fused kernel's grid barriers (masses of equal keys only) need that, and its waits are bounded anyway.  A fixed
    // grid of 256 read the 200 MB of a 1.25e7-candidate shard at 18 % of the HBM rate: the loads in
    // flight, not the bandwidth, were the limit.
    const int64_t want = (n + TK_CACHE - 1) / TK_CACHE;
    const int64_t cap = h->fused_tail ? (h->tk_coresident < TK_MAXBLK ? h->tk_coresident : TK_MAXBLK) : TK_MAXBLK;
    const int grid = (int)(want < 1 ? 1 : (want < cap ? want : cap));
    if (!compacted) {
        int64_t chunk = (n + grid - 1) / grid;
        chunk = (chunk + TK_THREADS - 1) / TK_THREADS * TK_THREADS;
        if (h->fused_tail) {
            const uint64_t *keys_arg = h->d_key_a;
            uint64_t *sk_arg = h->d_sel_key;
            uint32_t *si_arg = h->d_sel_idx;
            int64_t n_arg = n, k_arg = k, chunk_arg = chunk, sel_arg = sel;
            int mode_arg = mode;
            int64_t pf_arg = 0;
            unsigned long long *stats_arg = nullptr;
            void *args[] = {&n_arg, &k_arg, &chunk_arg, &keys_arg, &ws, &sk_arg, &si_arg, &mode_arg, &sel_arg, &eig, &obj, &pf_arg, &stats_arg};
            if (onfly)      // (a measure the mode does not use is never looked at: any readable array of n doubles will do)
                hipLaunchKernelGGL(tk_refine_kernel<true>, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, k, chunk, nullptr, ws,
                                   h->d_sel_key, h->d_sel_idx, mode, sel, eig ? eig : obj, obj ? obj : eig,
                                   (int64_t)((h->prefilter && h->pf_counted && h->N >= SDPCUT_PF_MIN_N) ? k : 0), h->d_stats);
            else if (h->coop_launch)      // the runtime guarantees the co-residency (+20 us per launch)
                HIP_TRY(h, hipLaunchCooperativeKernel((const void *)tk_refine_kernel<false>, dim3(grid), dim3(TK_THREADS), args, 0, h->stream));
            else
                hipLaunchKernelGGL(tk_refine_kernel<false>, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, k, chunk, h->d_key_a, ws,
                                   h->d_sel_key, h->d_sel_idx, mode, sel, eig, obj, (int64_t)0, (unsigned long long *)nullptr);
        } else {
            // one launch per digit, no wait anywhere inside a kernel: the path that always answers (each launch returns at
            // once when the selection has been closed by an earlier digit)
            for (int p = 1; p < 8; ++p)
                hipLaunchKernelGGL(tk_hist_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, p, n, k, h->d_key_a, ws);
            hipLaunchKernelGGL(tk_count_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, chunk, h->d_key_a, ws);
            hipLaunchKernelGGL(tk_write_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, chunk, h->d_key_a, ws,
                               h->d_sel_key, h->d_sel_idx);
        }
    }
    const int64_t maxk = k <= TK_LDSK ? TK_LDSK : TK_MAXK;    // an early stop compacts up to maxk entries; idle tiles exit at once
    const int ntiles = (int)(maxk / TK_TILE);
    uint64_t *tile_key = h->d_sel_key + TK_MAXK;
    uint32_t *tile_idx = h->d_sel_idx + TK_MAXK;
    const double *tie_obj = (mode == TK_MODE_COMBALL || mode == TK_MODE_COMBAUTO) ? h->d_obj : nullptr;
    const dim3 g_sort(ntiles), g_merge(ntiles * TK_TILE / TK_THREADS), blk(TK_THREADS);
    if (maxk > TK_LDSK) {
        // big heads (8193 .. 16384): keys-only merge; the device-resolved regime never asks for them
        if (mode == TK_MODE_COMBAUTO) return sdpcut_fail(h, SDPCUT_EINVAL, "top-k select: head too long for this mode");
        if (tie_obj) {
            hipLaunchKernelGGL(tk_tilesort_kernel<1>, g_sort, blk, 0, h->stream, ws, h->d_sel_key, h->d_sel_idx, tile_key, tile_idx, tie_obj);
            hipLaunchKernelGGL(tk_mergerank_big_kernel<true>, g_merge, blk, 0, h->stream, base, score_add, ws, tile_key, tile_idx,
                               d_idx_out, d_score_out, tie_obj, raw, emit_limit, h->shard_rec, h->shard_rec_count, h->shard_rec_len);
        } else {
            hipLaunchKernelGGL(tk_tilesort_kernel<0>, g_sort, blk, 0, h->stream, ws, h->d_sel_key, h->d_sel_idx, tile_key, tile_idx, tie_obj);
            hipLaunchKernelGGL(tk_mergerank_big_kernel<false>, g_merge, blk, 0, h->stream, base, score_add, ws, tile_key, tile_idx,
                               d_idx_out, d_score_out, tie_obj, raw, emit_limit, h->shard_rec, h->shard_rec_count, h->shard_rec_len);
        }
    } else {
        if (raw) return sdpcut_fail(h, SDPCUT_EINVAL, "top-k select: raw output needs the big-head merge");
#define TK_SORT_LAUNCH(T)                                                                                                  \
    do {                                                                                                                   \
        hipLaunchKernelGGL(tk_tilesort_kernel<T>, g_sort, blk, 0, h->stream, ws, h->d_sel_key, h->d_sel_idx, tile_key,     \
                           tile_idx, tie_obj);                                                                             \
        hipLaunchKernelGGL(tk_mergerank_kernel<T>, g_merge, blk, 0, h->stream, base, score_add, ws, tile_key, tile_idx,    \
                           d_idx_out, d_score_out, tie_obj, h->shard_rec, h->shard_rec_count, h->shard_rec_len);           \
    } while (0)
        if (mode == TK_MODE_COMBAUTO) TK_SORT_LAUNCH(2);
        else if (mode == TK_MODE_COMBALL) TK_SORT_LAUNCH(1);
        else TK_SORT_LAUNCH(0);
#undef TK_SORT_LAUNCH
    }
    HIP_TRY(h, hipGetLastError());
    return 0;
}

// stage 0: fresh selection; 1: the workspace has been handed out by topk_begin already (the score
// kernels left their strong count in it); 3: the score kernels also counted the leading digit of the keys
// and the violated / positive candidates (ScoreFuse; topk_fuse_ok).  mode TK_MODE_COMBAUTO: resolved by
// the first pass against `sel` (stages 1 and 3).
int topk_select_enqueue(sdpcut_ctx *h, int mode, int64_t k, double score_add, int64_t *d_idx_out,
                        double *d_score_out, const int64_t **d_counters_out, int stage, int64_t sel)
{
    const int64_t n = h->N;
    if (k < 1 || k > TK_MAXK || n < 1) return sdpcut_fail(h, SDPCUT_EINVAL, "top-k select: k out of range");
    if (mode == TK_MODE_COMBAUTO && stage != 1 && stage != 3) return sdpcut_fail(h, SDPCUT_ESTATE, "top-k select: no strong count");
    int rc = 0;
    if (stage == 0) {
        rc = topk_begin(h, nullptr, nullptr);
        if (rc) return rc;
    }
    const bool digit_done = stage == 3;      // the score kernels counted digit 0; no key array (tk_refine_kernel<true>)
    TopkWs *ws = (TopkWs *)h->d_topk_ws;
    const double *eig = (h->scored & SDPCUT_EIG) ? h->d_eig : nullptr;
    const double *obj = (h->scored & SDPCUT_NN) ? h->d_obj : nullptr;
    int64_t nb = (n + TK_THREADS - 1) / TK_THREADS;
    const int grid = (int)(nb < TK_MAXBLK ? nb : TK_MAXBLK);
    const int64_t maxk = k <= TK_LDSK ? TK_LDSK : TK_MAXK;
    const bool comb = mode == TK_MODE_COMBAUTO || mode == TK_MODE_COMBALL;
    if (!digit_done && smallsort_range(h, n, k)) {
        // a short list with a head of one tile at most: selection, sort and emission in ONE launch
        hipLaunchKernelGGL(tk_smallsel_kernel<true>, dim3(1), dim3(TK_SMALLSEL_THREADS), 0, h->stream, mode, sel, (int)n, (int)k, eig, obj,
                           ws, h->d_sel_key, h->d_sel_idx, h->base, score_add, d_idx_out, d_score_out);
        HIP_TRY(h, hipGetLastError());
        if (d_counters_out) *d_counters_out = ws->counters;
        return 0;
    }
    const bool smallsel = !digit_done && smallsel_range(n, k, comb);
    const bool small = smallsel || (!digit_done && n <= maxk);
    if (smallsel) {
        hipLaunchKernelGGL(tk_smallsel_kernel<false>, dim3(1), dim3(TK_SMALLSEL_THREADS), 0, h->stream, mode, sel, (int)n, (int)k, eig, obj,
                           ws, h->d_sel_key, h->d_sel_idx, h->base, score_add, d_idx_out, d_score_out);
    } else if (small) {
        hipLaunchKernelGGL(tk_small_kernel, dim3(1), dim3(TK_SMALL_THREADS), 0, h->stream, mode, sel, n, k, eig, obj, ws, h->d_sel_key,
                           h->d_sel_idx);
    } else if (!digit_done) {
        hipLaunchKernelGGL(tk_keys_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, mode, sel, n, k, eig, obj, h->d_key_a, ws);
    }
    rc = topk_enqueue_after_pass0(h, ws, mode, n, k, score_add, d_idx_out, d_score_out, small, 0, h->base, TK_MAXK, digit_done, sel,
                                  eig, obj);
    if (rc) return rc;
    if (d_counters_out) *d_counters_out = ws->counters;
    return 0;
}

// Head of a ranking over n PRECOMPUTED keys in h->d_key_a (0 = not in the class), ties by index:
// idx_out = entry index, val_out = the key's low 63 bits as a double; cnt as in topk_select_on_device.
int topk_select_keys_on_device(sdpcut_ctx *h, int64_t n, int64_t k, int64_t *d_idx_out, double *d_val_out, int64_t cnt[5])
{
    if (k < 1 || k > TK_MAXK || n < 1 || n > h->key_n) return sdpcut_fail(h, SDPCUT_EINVAL, "top-k select: k / n out of range");
    int rc = topk_begin(h, nullptr, nullptr);
    if (rc) return rc;
    TopkWs *ws = (TopkWs *)h->d_topk_ws;
    int64_t nb = (n + TK_THREADS - 1) / TK_THREADS;
    const int grid = (int)(nb < TK_MAXBLK ? nb : TK_MAXBLK);
    // (always through the big-head merge: it is the one with the raw key output)
    const int64_t kk = k <= TK_LDSK ? TK_LDSK + 1 : k;
    hipLaunchKernelGGL(tk_prekeys_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, kk, h->d_key_a, ws);
    rc = topk_enqueue_after_pass0(h, ws, TK_MODE_FEAS, n, kk, 0.0, d_idx_out, d_val_out, false, 1, 0, k);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(cnt, ws->counters, 5 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    return 0;
}

// ------------------------------------------------------------------------------------------
// The threshold tie group of the every-entry-visited combined ranking, cut by its SECONDARY key (r4).
//
// Mode COMBALL orders equal new scores by obj_improve, then index (the first stable sort of the reference,
// cut_select_qp.py:601, under its second, :625), so a group of equal keys that straddles position k cannot be cut by
// index.  As long as the whole group fits the sort buffers it is taken whole (resolve_digit); at a structured LP vertex --
// round 1 of every BoxQP run under the combined strategy with fewer than sel strong candidates: x = 0.5, X in {0, 0.5},
// thousands of candidates with the SAME -lambda_min -- it does not, the selection declares itself void (counters[4] = 2)
// and, until round 3, a rocPRIM sort of the full list answered.  Now: the void selection has left the threshold key T, the
// number of keys above it and the number `need` still wanted from the group in its workspace, and
//     head = [ every key > T, ordered (key, obj_improve, index) ]  ++  [ top `need` of the group by (obj_improve, index) ]
// -- two ordinary radix selections over precomputed keys: A keeps key > T (all of them are wanted: nothing to cut), B ranks
// the group by the image of obj_improve and cuts ITS ties by index, which is exactly the reference's order.  Hand-written
// path, no library sort, a few launches more than a normal round.
__global__ __launch_bounds__(TK_THREADS) void tk_tiekeys_kernel(int64_t n, uint64_t T, int part, const double *eig, const double *obj,
                                                                uint64_t *keys)
{
    for (int64_t i = (int64_t)blockIdx.x * TK_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * TK_THREADS) {
        const double o = obj[i];
        const uint64_t key = masked_key(TK_MODE_COMBALL, eig[i], o);
        keys[i] = part == 0 ? (key > T ? key : 0ull) : (key == T ? key_of(o) : 0ull);
    }
}

__global__ __launch_bounds__(TK_THREADS) void tk_fillscore_kernel(double *out, int64_t count, uint64_t T)
{
    const int64_t i = (int64_t)blockIdx.x * TK_THREADS + threadIdx.x;
    if (i < count) out[i] = score_of(T);
}

// head of a ranking over the n precomputed keys in h->d_key_a (0 = not in the class); tie: equal keys by obj_improve, then
// index (else by index).  Enqueued; *d_void receives the device address of the selection's void flag.
static int select_prekeys_enqueue(sdpcut_ctx *h, int64_t n, int64_t k, bool tie, int64_t *d_idx_out, double *d_score_out,
                                  const int64_t **d_void)
{
    int rc = topk_begin(h, nullptr, nullptr);
    if (rc) return rc;
    TopkWs *ws = (TopkWs *)h->d_topk_ws;
    const int64_t nb = (n + TK_THREADS - 1) / TK_THREADS;
    const int grid = (int)(nb < TK_MAXBLK ? nb : TK_MAXBLK);
    hipLaunchKernelGGL(tk_prekeys_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, k, h->d_key_a, ws);
    *d_void = &ws->counters[4];
    return topk_enqueue_after_pass0(h, ws, tie ? TK_MODE_COMBALL : TK_MODE_FEAS, n, k, 0.0, d_idx_out, d_score_out, false, 0,
                                    h->base, TK_MAXK);
}

// h->d_topk_ws holds a COMBALL selection for a head of k entries that declared itself void because of its threshold tie
// group (counters[4] == 2).  Writes the head (k_eff entries) to d_idx_out / d_score_out.
// -> 0 done; 1 not applicable (another kind of void: the caller takes its general path); < 0 error.
int topk_tie_split(sdpcut_ctx *h, int64_t k, int64_t *d_idx_out, double *d_score_out, int64_t *k_eff_out)
{
    const int64_t n = h->N;
    if (!h->d_topk_ws || (h->scored & (SDPCUT_EIG | SDPCUT_NN)) != (SDPCUT_EIG | SDPCUT_NN) || k < 1 || k > TK_MAXK) return 1;
    const TopkWs *ws = (const TopkWs *)h->d_topk_ws;
    TkState st8;
    int64_t c[8];
    HIP_TRY(h, hipMemcpyAsync(&st8, &ws->state[8], sizeof(st8), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(c, ws->counters, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    const int64_t k_eff = c[3], need = st8.need, above = k_eff - need;
    if (c[4] != 2 || c[6] != TK_MODE_COMBALL || need < 1 || above < 0 || k_eff > k) return 1;
    const uint64_t T = st8.prefix;
    int rc = ensure_key_ws(h, n);
    if (rc) return rc;
    const int64_t nb = (n + TK_THREADS - 1) / TK_THREADS;
    const int grid = (int)(nb < 4096 ? nb : 4096);
    int64_t void_a = 0, void_b = 0;
    const int64_t *d_void = nullptr;
    if (above > 0) {
        hipLaunchKernelGGL(tk_tiekeys_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, T, 0, h->d_eig, h->d_obj, h->d_key_a);
        rc = select_prekeys_enqueue(h, n, above, true, d_idx_out, d_score_out, &d_void);
        if (rc) return rc;
        HIP_TRY(h, hipMemcpyAsync(&void_a, d_void, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    }
    hipLaunchKernelGGL(tk_tiekeys_kernel, dim3(grid), dim3(TK_THREADS), 0, h->stream, n, T, 1, h->d_eig, h->d_obj, h->d_key_a);
    rc = select_prekeys_enqueue(h, n, need, false, d_idx_out + above, d_score_out + above, &d_void);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(&void_b, d_void, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    hipLaunchKernelGGL(tk_fillscore_kernel, dim3((unsigned)((need + TK_THREADS - 1) / TK_THREADS)), dim3(TK_THREADS), 0, h->stream,
                       d_score_out + above, need, T);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, sdpcut_sync(h));
    if (void_a || void_b) return 1;      // a bounded wait expired (the GPU shared with a blocking kernel): the general path answers
    ++h->stat_tie_splits;
    if (k_eff_out) *k_eff_out = k_eff;
    return 0;
}

int topk_select_on_device(sdpcut_ctx *h, int mode, int64_t k, double score_add, int64_t *d_idx_out,
                          double *d_score_out, int64_t cnt[5])
{
    const int64_t *d_cnt = nullptr;
    int rc = topk_select_enqueue(h, mode, k, score_add, d_idx_out, d_score_out, &d_cnt, 0, 0);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(cnt, d_cnt, 5 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, sdpcut_sync(h));
    return 0;
}
