// Device-side pieces of the top-k selection shared by topk.hip and score.hip (the score kernel
// can emit the radix keys and the first histogram itself, see ScoreArgs::tk).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "keys.h"

#define TK_THREADS 256
#define TK_MAXBLK 1024   // most workgroups a selection pass is launched with (blk_eq / blk_gt slots)
#define TK_LDSK 8192     // heads whose compacted superset (keys AND indices) fits the merge kernel's LDS
#define TK_MAXK 16384    // largest head: the merge keeps the keys in LDS and reads indices only on ties
#define TK_UNROLL 8      // grid-stride rounds whose loads are issued together
// lists of at most TK_SMALLSEL_N candidates with a head of at most TK_LDSK: selection by ONE workgroup, keys in LDS (tk_smallsel_kernel)
#ifndef TK_SMALLSEL
#define TK_SMALLSEL 1
#endif
#ifndef TK_SMALLSORT
#define TK_SMALLSORT 1     // ... and, for heads of one tile at most, sorted and emitted by it too (ONE launch for the round's selection)
#endif
#define TK_SMALLSEL_N 12288
#define TK_SMALLSEL_THREADS 1024
// Replicas of the global histogram: same-address device-scope atomics are serialised at ~15-20 ns
// each, so 256 workgroups flushing into ONE row cost ~5 us per pass; block b adds into replica
// b % TK_HREP (32 adds per cell) and the resolving block sums the replicas.
#ifndef TK_HREP
#define TK_HREP 8
#endif
#define TK_SREP 8
// Replicas of the leading-digit histogram the SCORE kernels flush into (TopkWs::hist_score): thousands of workgroups add
// their handful of non-empty bins when they retire; with 8 replicas the eigenvalue kernel's 3907 workgroups queued ~500
// deep on every cell (+11 us on a 48 us kernel), with 32 the chains are ~120 deep and hidden behind the kernel's tail.
#define TK_SHREP 32

// ---- the fine histogram of a selection's class (r5) ------------------------------------------------------
// The LEADING byte of a key image (sign + seven exponent bits) is shared by nearly every member of a class, so the
// leading-digit histogram the score kernels count (hist_score) only saves the selection its first pass: tk_refine_kernel still
// ran one or two histogram passes behind grid barriers over all 10^6 keys to find ~5000 of them (25 us of a 387 us round,
// 75 % of what a feasibility round spends outside its 28 us eigenvalue kernel).  Now the kernels that produce the scores also
// count the members by the top SEVENTEEN bits of the key (sign, exponent, five mantissa bits: 3 % resolution) inside a window
// of 2048 codes and report only the TOP of what they saw: a score workgroup keeps a table in LDS (one LDS atomic per
// candidate) and reports, when it retires, the bins down to the one that holds its m-th largest member, m a few times its
// expected share of the head (pf_mloc); so does a workgroup of the eigenvalue kernel (two or three tiles each when it counts).
// Each also publishes the lowest bin it reported (pf_floor, a maximum over all of them).  Every bin at or above the floor is
// then EXACT.  The selection finds the bin e* of the k-th largest key in the global table; if e*
// lies at or above the floor -- no reporting unit held more members of the head than it reported: always, unless the list is
// sorted by score -- and the members at or above e* fit the sort buffers, it compacts them in ONE pass over the scores: no
// histogram pass, no grid barrier, and everything it reads is requested by its very first instructions (tk_refine_kernel,
// `direct`).  Otherwise (a unit rich in head members; a fat bin: masses of equal keys at a structured LP vertex; the
// every-entry-visited regime, whose keys are not the ones counted) the radix passes run as before: same result either way.
// What it costs the producers: one LDS atomic per candidate, ~170 instructions of ONE wave when a workgroup retires, and ~15
// no-return device atomics per workgroup.  Device atomics serialise per ADDRESS at 11 ns and per 128-B line at 1.6-2 ns
// (profiles/r05_ubench_atomics.txt; the first version -- every member reported at 16-bit resolution into 4 KB -- queued 3e5
// atomics on ~25 hot words of two lines and cost the eigenvalue kernel 200 us): consecutive bins therefore sit in consecutive
// LINES (a hot run of bins = as many lines), and there are two replicas.
// Window of 2048 bins over the 17-bit codes, clamped on both sides (a bin at an end of a range also holds everything beyond it):
//   feasibility keys (-lambda_min in (1e-15, 8]):  one range, 2^-56 .. 2^8 (64 binades of 32 codes);
//   obj_improve keys:  1024 bins for -2^16 .. -2^-16 and 1024 for 2^-16 .. 2^16 -- on dense instances the optimality ranking's
//   5000-th score is NEGATIVE from the fourth round on (spar090-075-1, spar125-075-1 dim 3: fewer than 300 positive scores).
#define PF_BINS 2048                    // 17-bit codes
#define PF_REP 2                        // replicas of the global table (by workgroup)
#define PF_FLOOR_REP 16                 // replicas of the floor word, one 128-B line each
#define PF_FEAS_BASE (0x10000 + ((1023 + 8) << 5) - PF_BINS)      // code of 2^8 minus the window
#define PF_POS_BASE (0x10000 + ((1023 - 16) << 5))                 // code of 2^-16
#define PF_NEG_BASE (0xFFFF - ((1023 + 16) << 5) + 1)              // code of the smallest key above -2^16
__device__ __forceinline__ int pf_clamp(int c, int n) { return c < 0 ? 0 : (c > n - 1 ? n - 1 : c); }
__device__ __forceinline__ int pf_code(uint64_t key, bool feas)
{
    const int c = (int)(uint32_t)(key >> 47);
    if (feas) return pf_clamp(c - PF_FEAS_BASE, PF_BINS);
    return c >= 0x10000 ? PF_BINS / 2 + pf_clamp(c - PF_POS_BASE, PF_BINS / 2) : pf_clamp(c - PF_NEG_BASE, PF_BINS / 2);
}
// lowest key of bin f (bin 0: everything; the lowest bin of the positive half: every key from +0 up)
__device__ __forceinline__ uint64_t pf_edge(int f, bool feas)
{
    if (f <= 0) return 0ull;
    if (feas) return (uint64_t)(uint32_t)(PF_FEAS_BASE + f) << 47;
    if (f < PF_BINS / 2) return (uint64_t)(uint32_t)(PF_NEG_BASE + f) << 47;
    return f == PF_BINS / 2 ? 0x8000000000000000ull : (uint64_t)(uint32_t)(PF_POS_BASE + f - PF_BINS / 2) << 47;
}
// word of bin f inside a replica of the global table: consecutive bins in consecutive 128-B lines
__device__ __forceinline__ int pf_slot(int f) { return (f & 63) * 32 + (f >> 6); }

// 4: combined strategy when the scan visits every entry -- the key is the new score of
// cut_select_qp.py:606-623, ties between equal new scores go by obj_improve, then index
// 5: combined strategy, regime resolved ON THE DEVICE by the first pass: STRONG if the score kernels
// counted at least `sel` strong candidates (counters[5]), else COMBALL -- one selection, no host
// round trip in either regime
enum { TK_MODE_FEAS = 1, TK_MODE_OPT = 2, TK_MODE_STRONG = 3, TK_MODE_COMBALL = 4, TK_MODE_COMBAUTO = 5 };

__device__ __forceinline__ int64_t ld_i64(const int64_t *p);

struct TkState {
    uint64_t prefix;   // digits resolved so far, in place
    int64_t need;      // how many of the keys matching the prefix are still wanted (0: nothing)
    int64_t stop;      // 1: the selection was closed early (see resolve_digit), later passes are no-ops
};

struct TopkWs {
    uint32_t hist[8][TK_HREP][256];   // [pass 0..7 = digit 7..0][replica][bin]
    int64_t counters[8];     // [0] class size  [1] nb_violated  [2] nb_positive  [3] k_eff
                             // [4] != 0: a bounded wait of tk_refine_kernel expired, the selection is void
                             // [5] strong candidates counted by the score kernels (TK_MODE_COMBAUTO)
                             // [6] mode the selection ran in (copy of `mode` for the host)
    TkState state[9];        // state[p]: after p digits
    uint32_t done[8];        // ticket counters of the passes
    uint32_t blk_eq[TK_MAXBLK];
    uint32_t blk_gt[TK_MAXBLK];
    int64_t n_sel;           // entries compacted by tk_write_kernel (>= k_eff after an early stop)
    int64_t mode;            // TK_MODE_* of the running selection (written by its pass 0)
    uint32_t ready[9];       // tk_refine_kernel: state[p] has been published inside the launch
    uint32_t pad_[1];
    int64_t strong_rep[TK_SREP];   // strong candidates counted by the score kernels, replicated by workgroup
                                   // (same-address device atomics are serialised, see TK_HREP)
    uint32_t bar[8];               // one-shot grid barriers of the fused kernels (arrival counters)
    uint32_t hist_alt[TK_HREP][256];   // leading-digit histogram of the COMBALL keys when the score kernels counted the
                                       // STRONG keys into hist_score and the device resolved the other regime
    uint32_t hist_score[TK_SHREP][256];   // leading digit of the selection's keys, counted by the score / eigenvalue kernels
    int64_t viol_rep[TK_SHREP];           // violated / positive candidates counted by the same kernels, replicated by workgroup:
    int64_t pos_rep[TK_SHREP];            // ONE counter each would queue thousands of retiring workgroups on one address (~15 ns
                                          // apiece: 58 us for the eigenvalue kernel's 3907 workgroups -- longer than the kernel runs)
    // (r5) the FINE histogram the score / eigenvalue kernels leave for the selection (see above); LAST in the struct: lists too
    // short for it zero only what lies in front (offsetof(TopkWs, pf_floor))
    uint32_t pf_floor[PF_FLOOR_REP][32];  // max over the reporting units of the lowest bin they reported (replicas in separate lines)
    uint32_t pf_tab[PF_REP][PF_BINS];     // class members by window code, word pf_slot(f) of replica blockIdx % PF_REP
};

__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int64_t ld_i64(const int64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_i64(int64_t *p, int64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS histogram update with one round of wave aggregation: in the first passes nearly every
// key of a wave falls into the same bin, and 64 same-address LDS atomics would serialise.
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t bin, bool active)
{
    const unsigned long long act = __ballot(active);
    if (!act) return;
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t lead_bin = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);   // (leader is wave-uniform: v_readlane, not ds_bpermute)
    const unsigned long long same = __ballot(active && bin == lead_bin);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lead_bin], (uint32_t)__popcll(same));
    if (active && bin != lead_bin) atomicAdd(&hist[bin], 1u);
}

// The same for digits that take only a handful of values across a wave (the LEADING digit of the keys:
// sign and high exponent bits -- a few bins hold everything): rounds of wave aggregation, so that a wave
// issues one LDS atomic per distinct bin instead of one per lane into the same few cells.
// (Device atomics straight into the global histogram, one per distinct bin and strip, were measured: +15 us
// on the score kernel -- its next loads queue behind them in the memory pipeline.)
__device__ __forceinline__ void hist_add_few(uint32_t *hist, uint32_t bin, bool active)
{
    unsigned long long rem = __ballot(active);
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll 1
    for (int it = 0; it < 4 && rem; ++it) {
        const int leader = __ffsll((long long)rem) - 1;
        const uint32_t lead_bin = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
        const unsigned long long same = __ballot(active && bin == lead_bin);
        if (lane == leader) atomicAdd(&hist[lead_bin], (uint32_t)__popcll(same));
        rem &= ~same;
    }
    if ((rem >> lane) & 1ull) atomicAdd(&hist[bin], 1u);
}

// ---- fine histogram: device side of the producers (score.hip, eig.hip) -------------------------------------------------
__device__ __forceinline__ void pf_publish_floor(TopkWs *ws, int floor_f)
{
    if (floor_f > 0)
        __hip_atomic_fetch_max(&ws->pf_floor[blockIdx.x % PF_FLOOR_REP][0], (uint32_t)floor_f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pf_report(TopkWs *ws, int f, uint32_t count)
{
    __hip_atomic_fetch_add(&ws->pf_tab[blockIdx.x % PF_REP][pf_slot(f)], count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The producers' side: a workgroup's table `tab` (LDS, PF_BINS counters of 16 bits, two per word; complete: call behind a
// workgroup barrier; 256 threads).  Thread t owns bins 8 t .. 8 t + 7.  Reports every bin down to the one that holds the
// workgroup's mloc-th largest member (fewer members than that: everything) and publishes that bin as its floor.  All four waves
// work (a retiring workgroup holds its slot until its last wave is done: 40 instructions each beat 170 of one wave); two barriers.
#ifndef SDPCUT_PF_NOINLINE
#define SDPCUT_PF_NOINLINE 0
#endif
#if SDPCUT_PF_NOINLINE
__attribute__((noinline))
#endif
static __device__ void pf_retire_table(TopkWs *ws, const uint32_t *tab, int mloc)
{
    __shared__ uint32_t wtot[4];
    __shared__ int s_floor;
    const int t = threadIdx.x, ln = t & 63, wv = t >> 6;
    uint32_t h[8], mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = tab[4 * t + j];
        h[2 * j] = w & 0xffffu;
        h[2 * j + 1] = w >> 16;
        mine += h[2 * j] + h[2 * j + 1];
    }
    uint32_t v = mine;      // suffix sums: thread t's bins lie above thread t - 1's
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_down((int)v, off);
        if (ln + off < 64) v += o;
    }
    if (ln == 0) wtot[wv] = v;
    if (t == 0) s_floor = 0;
    __syncthreads();
    for (int w = wv + 1; w < 4; ++w) v += wtot[w];
    uint32_t above = v - mine;
#pragma unroll
    for (int c = 7; c >= 0; --c) {
        const uint32_t here = above + h[c];
        if (here >= (uint32_t)mloc && above < (uint32_t)mloc) s_floor = 8 * t + c;      // one bin of one thread
        above = here;
    }
    __syncthreads();
    const int floor_f = s_floor;
    if (t == 0) pf_publish_floor(ws, floor_f);
    if (8 * t + 7 < floor_f || mine == 0) return;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (8 * t + c >= floor_f && h[c]) pf_report(ws, 8 * t + c, h[c]);
}

// Executed by the LAST block of pass p: resolve digit 7-p and publish state[p+1].
// row / cls_known / mine / publish: a pass 0 that the score kernels counted is resolved by EVERY block of
// the selection for itself -- state[1] goes to *mine (LDS); only the block with publish = true also
// stores the global copy (the resolving block of pass 1 reads it; 256 blocks storing the same words
// would queue up on one cache line).
static __device__ void resolve_digit(TopkWs *ws, int p, int64_t k, const uint32_t (*row)[256] = nullptr,
                                     int64_t cls_known = -1, TkState *mine = nullptr, bool publish = true,
                                     int mode_known = 0, bool plain = false, int nrep = TK_HREP)
{
    __shared__ uint32_t suf[TK_THREADS];
    const int t = threadIdx.x;
    int64_t need;
    uint64_t prefix;
    if (!row) row = ws->hist[p];
    if (p == 0) {
        const int64_t cls = cls_known >= 0 ? cls_known : ld_i64(&ws->counters[0]);
        need = k < cls ? k : cls;
        prefix = 0;
        if (t == 0 && publish) st_i64(&ws->counters[3], need);      // k_eff for the later passes and kernels
    } else {
        need = ld_i64(&ws->state[p].need);       // written by the previous launch, or by another block of this one
        prefix = (uint64_t)ld_i64((const int64_t *)&ws->state[p].prefix);
    }
    {
        uint32_t acc = 0;
        for (int r = 0; r < nrep; ++r) acc += plain ? row[r][t] : ld_u32(&row[r][t]);      // plain: written by an earlier launch
        suf[t] = acc;
    }
    {   // suffix sums S[t] = sum_{b >= t} hist[b]: within each wave by shuffles, then the totals of the waves above
        __shared__ uint32_t wtot[TK_THREADS / 64];
        const int ln = t & 63, wv = t >> 6;
        uint32_t v = suf[t];
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_down((int)v, off);
            if (ln + off < 64) v += o;
        }
        if (ln == 0) wtot[wv] = v;
        __syncthreads();
        for (int w = wv + 1; w < TK_THREADS / 64; ++w) v += wtot[w];
        suf[t] = v;
        __syncthreads();
    }
    const int64_t here = suf[t], above = (t < 255) ? suf[t + 1] : 0;
    if (need >= 1) {
        if (here >= need && above < need) {       // exactly one bin
            const uint64_t pre = prefix | ((uint64_t)t << (8 * (7 - p)));
            // Early stop: the keys above this bin (k_eff - (need - above) of them) plus the WHOLE bin
            // fit the sort buffers, so take that superset -- every key >= the bin's lowest value --
            // and let the final sort put the wanted k_eff first.  With 10^6 spread-out scores this
            // happens after 2-3 digits; masses of equal keys keep the passes going to the last digit,
            // where ties are cut by index as before.
            const int64_t k_eff = (p == 0) ? need : ld_i64(&ws->counters[3]);
            const int64_t in_bin = here - above;
            const int64_t maxk = k <= TK_LDSK ? TK_LDSK : TK_MAXK;     // what the sort buffers of this head hold
            const int64_t superset = (p == 0 ? need : k_eff) - (need - above) + in_bin;
            // Mode COMBALL cannot cut a group of equal keys by index (their order is by obj_improve):
            // at the last digit it either takes the whole group too, or declares the selection void
            // (counters[4]) and the host sorts the full list.
            const bool comball = (mode_known ? (int64_t)mode_known : ld_i64(&ws->mode)) == TK_MODE_COMBALL;
            if (p == 7 && comball && in_bin > need - above && superset > maxk) st_i64(&ws->counters[4], 2);
            if ((p < 7 || comball) && superset <= maxk) {
                if (mine) { mine->prefix = pre; mine->need = in_bin; mine->stop = 1; }
                for (int qq = p + 1; qq <= 8 && publish; ++qq) {
                    st_i64((int64_t *)&ws->state[qq].prefix, (int64_t)pre);
                    st_i64(&ws->state[qq].need, in_bin);   // every key equal to the bin's lowest value, if any
                    st_i64(&ws->state[qq].stop, 1);
                }
            } else {
                if (mine) { mine->prefix = pre; mine->need = need - above; mine->stop = 0; }
                if (publish) {
                    st_i64((int64_t *)&ws->state[p + 1].prefix, (int64_t)pre);
                    st_i64(&ws->state[p + 1].need, need - above);
                }
            }
        }
    } else if (t == 0) {
        if (mine) { mine->prefix = 0; mine->need = 0; mine->stop = 0; }
        if (publish) {
            st_i64((int64_t *)&ws->state[p + 1].prefix, 0);
            st_i64(&ws->state[p + 1].need, 0);
        }
    }
}

// publish this block's LDS histogram; the last block to arrive resolves the digit.
// Everything the resolving block reads (histogram cells, class counters, states) is written with
// device-scope atomics and read with device-scope atomic loads, so the hand-off needs no release
// FENCE: every wave drains its atomics (vmcnt(0) = performed at the coherence point), workgroup
// barrier, one lane takes the ticket.  (A release fence writes the XCD's dirty L2 lines back --
// with the score kernel's 24 MB of output in flight that cost ~25 us over the 2048 blocks; a
// __threadfence() by all threads ~30 us per pass.)  The last block acquires once.
static __device__ void finish_pass(TopkWs *ws, int p, int64_t k, const uint32_t *hist, uint32_t total_blocks,
                                   bool publish = false, uint32_t (*row)[256] = nullptr)
{
    __shared__ uint32_t ticket;
    if (!row) row = ws->hist[p];
    if (hist[threadIdx.x]) atomicAdd(&row[blockIdx.x % TK_HREP][threadIdx.x], hist[threadIdx.x]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        ticket = __hip_atomic_fetch_add(&ws->done[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket == total_blocks - 1) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        resolve_digit(ws, p, k, row);
        if (publish) {      // grid barrier of tk_refine_kernel: state[p+1] is complete
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(&ws->ready[p + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__device__ __forceinline__ int64_t strong_total(const TopkWs *ws)
{
    int64_t t = 0;
#pragma unroll
    for (int r = 0; r < TK_SREP; ++r) t += ld_i64(&ws->strong_rep[r]);
    return t;
}

// the mode a selection runs in: COMBAUTO is resolved from the strong count the score kernels left
__device__ __forceinline__ int resolve_mode(int mode, const TopkWs *ws, int64_t sel)
{
    if (mode != TK_MODE_COMBAUTO) return mode;
    return strong_total(ws) >= sel ? TK_MODE_STRONG : TK_MODE_COMBALL;
}

__device__ __forceinline__ uint64_t masked_key(int mode, double eig, double obj)
{
    if (mode == TK_MODE_OPT) return key_of(obj);
    const bool viol = eig < SDPCUT_NEG_EIGVAL;
    if (mode == TK_MODE_COMBALL) {      // same arithmetic as comb_keys_kernel (rank.hip) for a visited entry
        double s = obj;
        if (s > 0.0) s = viol ? s + SDPCUT_BIG_M : s - SDPCUT_BIG_M;
        else if (viol) s = -eig;
        return key_of(s);
    }
    if (mode == TK_MODE_FEAS) return viol ? key_of(-eig) : 0ull;
    return (obj > 0.0 && viol) ? key_of(obj) : 0ull;
}

