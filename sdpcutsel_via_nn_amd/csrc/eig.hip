// Eigenvalue-only scoring kernel -- what a pure-feasibility round runs
// (cut_select_qp.py:639-654: for every candidate eigvalsh(...)[0], violated iff < -1e-15).
//
// Fifteen of the twenty recorded rounds of BASELINE configs[2] (the strategy switches 4 -> 1 in round 5)
// and every QCQP round over the constraints-only cover (cut_select_qcqp.py:75-77) are this scan.  It has
// no MLP: no weights, no LDS strips, no MFMA accumulators.  score_mfma_kernel serves it with a uniform
// branch but is compiled for the MLP's 225-245 registers (2 waves per SIMD), and a feasibility round is a
// chain of dependent v_rsq / v_rcp / fma sequences behind two dependent memory round trips (index set,
// then the gathers it addresses) -- exactly the kind of code that needs many waves to hide latency.
//
// Here: lane = candidate, nothing but the gather, the register Jacobi (the SAME template as every other
// kernel of the library: bit-equal lambda_min) and the leading-digit histogram of the selection that
// follows.  Register budget by the largest size class present (KMAX): the kernel is instantiated for
// KMAX = 2..5, so that a dim-3 cover is not compiled for the 6x6 matrices of dim 5.
//
// ONE launch for all size classes: the tiles of all classes form one list, largest size first (its
// strips are the longest: the short ones fill the tail), and workgroups take tiles round-robin.  The
// per-size launches of the scoring kernels each end in their own tail; on a mixed cover (spar125-075-1
// dim 4: 2/3/4-variable sets) that was three tails per round.
#include <hip/hip_ext.h>

#include "common.h"
#include "gather.h"
#include "topk_dev.h"

struct EigArgs {
    const int32_t *set[SDPCUT_MAX_K + 1];    // SoA [k][n_k] per size class
    const int32_t *orig[SDPCUT_MAX_K + 1];
    int64_t n[SDPCUT_MAX_K + 1];
    int64_t tile_end[SDPCUT_MAX_K + 1];      // tiles of class k are [tile_end[k + 1], tile_end[k]), class 5 first (from 0)
    const double *vars;
    int32_t nv;
    int64_t L;
    double *eig_out;
    TopkWs *tk;      // FUSE: leading-digit histogram + violated count of the feasibility selection that follows
    int32_t spread;  // the four waves of a tile take their strips from four distant quarters of a class's list (see eig_class)
    int32_t pf_mloc; // FUSE: fine histogram of the class (topk_dev.h): a workgroup reports its table down to its pf_mloc-th largest member; 0: off
};

// ---- re-packing the lanes that have not converged ------------------------------------------------------------
// Cyclic Jacobi needs 3-5 sweeps on these matrices, and a wave runs as many as its SLOWEST lane: on the c2 list 72 %
// of the 4x4 matrices are done after 4 sweeps and 1 % need a fifth, which 47 % of the waves then run for everybody
// (mean over lanes 3.74 sweeps, mean over waves 4.47); 77 % of the 5x5 matrices need 4 sweeps, 23 % need 5, and 96 %
// of the waves run 5.  So a tile stops after R sweeps, the lanes that have not converged put their matrices into LDS
// -- packed, in any order -- and only as many waves as they fill continue.  Every matrix goes through exactly the
// rotations it would have gone through alone: lambda_min is bit-for-bit what jacobi_eig returns.
// Measured on 1e6 candidates per size (tools/gpu_eig_abl.sh, profiles/r03_eig_kernel_variants.txt): 5x5 matrices 85.7 -> 77.6 us
// (-9.5 %), 6x6 141.6 -> 136.9 (-3 %); 4x4 matrices gain nothing either way (re-pack after 3 sweeps: 73 % of the lanes
// go through LDS, 48.6 vs 47.0 us; after 4: 1 % do, 47.7 -- the barrier that couples the four waves costs what the
// fifth sweep of half the waves does), so 2- and 3-variable tiles run the plain loop.
constexpr int eig_pack_sweeps(int k) { return k <= 3 ? JACOBI_MAX_SWEEPS : (k == 4 ? 4 : 5); }      // sweeps before the re-pack
constexpr int eig_pack_cap(int k) { return k <= 3 ? 0 : (k == 4 ? 128 : 64); }      // packed lanes LDS holds (the rest goes on in place)
template <int K> struct EigPack {
    static constexpr int D = K + 1;
    static constexpr int NS = D * (D + 1) / 2 + 1;       // upper triangle + the tolerance
    static constexpr int R = eig_pack_sweeps(K);
    static constexpr int CAP = eig_pack_cap(K);
};
constexpr int eig_pack_doubles(int kmax)
{
    int m = 0;
    for (int k = 2; k <= kmax; ++k) {
        const int d = k + 1, ns = d * (d + 1) / 2 + 1;
        const int cap = eig_pack_cap(k);
        m = ns * cap > m ? ns * cap : m;
    }
    return m > 0 ? m : 1;
}
#ifndef SDPCUT_EIG_REPACK
#define SDPCUT_EIG_REPACK 1
#endif

template <int K, bool FUSE>
__device__ __forceinline__ void eig_emit(const EigArgs &A, double lam, int32_t out_idx, bool live, uint32_t *tk_hist, uint32_t &c_viol)
{
    if (live) A.eig_out[out_idx] = lam;
    if constexpr (FUSE) {
        const bool viol = live && lam < SDPCUT_NEG_EIGVAL;
        const uint64_t key = key_of(-lam);
        hist_add_few(tk_hist, (uint32_t)(key >> 56), viol);
        if (viol && A.pf_mloc > 0) {      // (r5) fine histogram of the class: the workgroup's table sits behind the leading-digit histogram
            const int f = pf_code(key, true);
            atomicAdd(&tk_hist[256 + (f >> 1)], (f & 1) ? 0x10000u : 1u);
        }
        c_viol += viol;
    }
}

// one tile of 256 candidates of size K (index set s, output slot out_idx of this lane's candidate already loaded);
// cnt = this tile's packed-lane counter (zero on entry)
template <int K, bool FUSE>
__device__ __forceinline__ void eig_tile(const EigArgs &A, const int32_t (&s)[K], int32_t out_idx, bool valid, double *s_state,
                                         int32_t *s_out, uint32_t *cnt, uint32_t *tk_hist, uint32_t &c_viol, int par)
{
    using P = EigPack<K>;
    constexpr int D = P::D;
    Cand<K> cd;
    gather_candidate<K>(cd, s, A.vars, nullptr, A.nv, A.L, false);
#if SDPCUT_LMIN
    // (r4) Householder + Laguerre, Jacobi for the lanes it hands back: no re-packing (the lanes Jacobi is left with are a few
    // per cent at structured vertices, none at generic points)
    eig_emit<K, FUSE>(A, candidate_eigmin<K>(cd, s, A.vars, A.nv, A.L), out_idx, valid, tk_hist, c_viol);
    (void)s_state; (void)s_out; (void)cnt;
    return;
#endif
    double a[D][D], v[D][D];
    fill_lifted<K>(a, cd.x, cd.X);
    const double tol = jacobi_tol<D, false>(a);
    if constexpr (SDPCUT_EIG_REPACK && P::CAP > 0) {
    jacobi_sweeps<D, false>(a, v, tol, P::R);
    const bool more = valid && !jacobi_converged<D>(a, tol);
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(more);
    uint32_t base = 0;
    if (lane == 0 && m) base = atomicAdd(cnt, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, 0);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const bool packed = more && slot < (uint32_t)P::CAP;
    if (packed) {
        int j = 0;
#pragma unroll
        for (int p = 0; p < D; ++p)
#pragma unroll
            for (int q = p; q < D; ++q) s_state[(j++) * P::CAP + slot] = a[p][q];
        s_state[j * P::CAP + slot] = tol;
        s_out[slot] = out_idx;
    }
    if (more && !packed) jacobi_sweeps<D, false>(a, v, tol, JACOBI_MAX_SWEEPS - P::R);      // LDS full: this lane goes on in place
    eig_emit<K, FUSE>(A, diag_min<D>(a), out_idx, valid && !packed, tk_hist, c_viol);
    __syncthreads();
    const uint32_t total = *cnt < (uint32_t)P::CAP ? *cnt : (uint32_t)P::CAP;
    if ((threadIdx.x & ~63u) < total) {      // uniform per wave: only the waves the packed lanes fill go on
        const bool mine = threadIdx.x < total;
        const uint32_t sl = mine ? threadIdx.x : 0;
        int j = 0;
#pragma unroll
        for (int p = 0; p < D; ++p)
#pragma unroll
            for (int q = p; q < D; ++q) { a[p][q] = s_state[(j++) * P::CAP + sl]; a[q][p] = a[p][q]; }
        const double tol2 = s_state[j * P::CAP + sl];
        const int32_t out2 = s_out[sl];
        if (mine) jacobi_sweeps<D, false>(a, v, tol2, JACOBI_MAX_SWEEPS - P::R);
        eig_emit<K, FUSE>(A, diag_min<D>(a), out2, mine, tk_hist, c_viol);
    }
    } else {
        jacobi_sweeps<D, false>(a, v, tol, JACOBI_MAX_SWEEPS);
        eig_emit<K, FUSE>(A, diag_min<D>(a), out_idx, valid, tk_hist, c_viol);
    }
}

// this workgroup's tiles of size class K: global tiles [lo, hi) of the launch's list, every gridDim.x-th from blockIdx.x
template <int K, bool FUSE>
__device__ __forceinline__ void eig_class(const EigArgs &A, int64_t lo, int64_t hi, double *s_state, int32_t *s_out, uint32_t *s_packed,
                                          int &par, uint32_t *tk_hist, uint32_t &c_viol)
{
    const int64_t G = gridDim.x, n = A.n[K];
    int64_t t = blockIdx.x;
    if (t < lo) t += (lo - t + G - 1) / G * G;
    if (t >= hi) return;      // uniform
    int32_t s_nxt[K];
    int32_t orig_nxt;
    // (r5) A.spread (one tile per workgroup: most real covers): the four waves of a tile work on four DISTANT quarters of the
    // class's list -- wave w of tile T takes strip w * ntile + T (64 consecutive candidates, coalesced as before).  Real covers
    // are enumerated index set by index set: neighbours share variables and scores, a run rich in members of the head would
    // otherwise fill a whole workgroup (pf_retire_table, topk_dev.h).
    // (32-bit index arithmetic, recomputed per tile: the kernel has no registers to spare -- three 64-bit values kept across the
    // tile loop cost the 3- and 4-variable instantiations 4 and 17 more spilled registers)
    auto cand_of = [&](int64_t tile) -> int {
        const int tl = (int)(tile - lo);
        return A.spread ? ((int)(threadIdx.x >> 6) * (int)(hi - lo) + tl) * 64 + (int)(threadIdx.x & 63) : tl * 256 + (int)threadIdx.x;
    };
    {
        const int c = cand_of(t), cc = c < n ? c : (int)n - 1;
        load_index_set<K>(s_nxt, A.set[K], n, cc);
        orig_nxt = A.orig[K][cc];
    }
    for (; t < hi; t += G, par ^= 1) {
        int32_t s_cur[K];
#pragma unroll
        for (int a = 0; a < K; ++a) s_cur[a] = s_nxt[a];
        const int32_t out_idx = orig_nxt;
        const bool valid = cand_of(t) < n;
        if (t + G < hi) {      // uniform
            const int c = cand_of(t + G), cc = c < n ? c : (int)n - 1;
            load_index_set<K>(s_nxt, A.set[K], n, cc);
            orig_nxt = A.orig[K][cc];
        }
        eig_tile<K, FUSE>(A, s_cur, out_idx, valid, s_state, s_out, &s_packed[par], tk_hist, c_viol, par);
        if constexpr (!SDPCUT_LMIN && SDPCUT_EIG_REPACK && EigPack<K>::CAP > 0) {
            // the next tile counts into the other word (zero since the barrier of the tile before this one); this tile's is
            // cleared behind the barrier that ends its use, in front of the barrier of the next tile
            __syncthreads();
            if (threadIdx.x == 0) s_packed[par] = 0;
        }
    }
}

// waves per SIMD the kernel is compiled for (hipcc -Rpass-analysis=kernel-resource-usage)
#ifndef SDPCUT_EIG_W
#define SDPCUT_EIG_W 6
#endif
#ifndef SDPCUT_EIG_W3
#define SDPCUT_EIG_W3 6
#endif
#ifndef SDPCUT_EIG_W5
#define SDPCUT_EIG_W5 4
#endif
template <int KMAX> struct EigOcc { static constexpr int W = KMAX <= 3 ? SDPCUT_EIG_W3 : (KMAX == 4 ? SDPCUT_EIG_W : SDPCUT_EIG_W5); };

template <int KMAX, bool FUSE>
__global__ __launch_bounds__(256, EigOcc<KMAX>::W) void eig_only_kernel(EigArgs A)
{
    __shared__ uint32_t tk_hist[256 + (FUSE ? PF_BINS / 2 : 0)];      // leading-digit histogram | (r5) the workgroup's table of window codes, 16-bit counters (topk_dev.h)
    __shared__ uint32_t tk_cnt;
    __shared__ double s_state[eig_pack_doubles(KMAX)];
    __shared__ int32_t s_out[256];
    __shared__ uint32_t s_packed[2];      // packed-lane counters of the current / the next tile
    if (threadIdx.x < 2) s_packed[threadIdx.x] = 0;
    if constexpr (FUSE) {
#pragma unroll
        for (int j = 0; j < 1 + PF_BINS / 2 / 256; ++j) tk_hist[threadIdx.x + 256 * j] = 0;
        if (threadIdx.x == 0) tk_cnt = 0;
    }
    __syncthreads();
    uint32_t c_viol = 0;
    int par = 0;
    // tiles of all classes form one list, largest size first; workgroup b takes tiles b, b + G, ... -- class by class, so that
    // the index set of its NEXT tile (HBM) is requested before the current one is worked on (the first of the two dependent
    // memory round trips of a tile off the critical path, as in score_mfma_kernel)
    if constexpr (KMAX >= 5) eig_class<5, FUSE>(A, 0, A.tile_end[5], s_state, s_out, s_packed, par, tk_hist, c_viol);
    if constexpr (KMAX >= 4) eig_class<4, FUSE>(A, A.tile_end[5], A.tile_end[4], s_state, s_out, s_packed, par, tk_hist, c_viol);
    if constexpr (KMAX >= 3) eig_class<3, FUSE>(A, A.tile_end[4], A.tile_end[3], s_state, s_out, s_packed, par, tk_hist, c_viol);
    eig_class<2, FUSE>(A, A.tile_end[3], A.tile_end[2], s_state, s_out, s_packed, par, tk_hist, c_viol);
    if constexpr (FUSE) {
        // (no ticket, nobody waits: the kernel boundary orders the atomics before the selection, see score.hip)
        for (int off = 32; off > 0; off >>= 1) c_viol += __shfl_xor((int)c_viol, off);
        if ((threadIdx.x & 63) == 0 && c_viol) atomicAdd(&tk_cnt, c_viol);
        __syncthreads();
        if (threadIdx.x == 0 && tk_cnt)
            __hip_atomic_fetch_add((unsigned long long *)&A.tk->viol_rep[blockIdx.x % TK_SHREP], (unsigned long long)tk_cnt, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        if (tk_hist[threadIdx.x])
            __hip_atomic_fetch_add(&A.tk->hist_score[blockIdx.x % TK_SHREP][threadIdx.x], tk_hist[threadIdx.x], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        if (A.pf_mloc > 0) pf_retire_table(A.tk, tk_hist + 256, A.pf_mloc);      // (behind the barrier above: the table is complete)
    }
}

// workgroups launched per resident slot: four short-lived ones -- the dispatcher balances tiles whose Jacobi needs a sweep
// more; one long-lived workgroup per slot with the next tile's index set in flight was measured 5 % slower
#ifndef SDPCUT_EIG_BLOCKS_PER_SLOT
#define SDPCUT_EIG_BLOCKS_PER_SLOT 4
#endif

template <int KMAX, bool FUSE>
static void eig_launch(sdpcut_ctx *h, EigArgs &A, hipEvent_t ev_start, hipEvent_t ev_stop, int64_t pf_k)
{
    const int64_t ntiles = A.tile_end[2];
    int64_t cap = (int64_t)h->n_cu * EigOcc<KMAX>::W * SDPCUT_EIG_BLOCKS_PER_SLOT;
    // (r5) with the fine histogram: ONE workgroup per resident slot, two or three tiles each -- a workgroup reports the top of its
    // table once, when it retires, and the table is only worth its scan over a few hundred members
    if (FUSE && pf_k > 0) cap = (int64_t)h->n_cu * EigOcc<KMAX>::W;
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    A.pf_mloc = 0;
#if SDPCUT_LMIN
    if (FUSE && pf_k > 0 && h->N > 0) {      // as pf_mloc_for (score.hip): 24 + eight times the workgroup's expected share of the head
        const int64_t per_wg = (ntiles + grid - 1) / grid * 256;
        const double m = 24.0 + 8.0 * (double)pf_k * (double)per_wg / (double)h->N;
        A.pf_mloc = per_wg >= 60000 ? 0 : (m > 60000.0 ? 60000 : (int)(m + 0.999));
    }
#endif
    A.spread = A.pf_mloc > 0 && ntiles <= grid;
    if (FUSE && A.pf_mloc == 0) h->pf_counted = false;
    if (ev_start || ev_stop)
        hipExtLaunchKernelGGL((eig_only_kernel<KMAX, FUSE>), dim3(grid), dim3(256), 0, h->stream, ev_start, ev_stop, 0, A);
    else
        hipLaunchKernelGGL((eig_only_kernel<KMAX, FUSE>), dim3(grid), dim3(256), 0, h->stream, A);
}

// lambda_min of every candidate of the handle's list at the current point, one launch.
// tk != nullptr: also the leading-digit histogram / violated count of a feasibility selection (TopkWs).
int launch_eig_only(sdpcut_ctx *h, void *tk, hipEvent_t ev_start, hipEvent_t ev_stop, int64_t pf_k)
{
    EigArgs A;
    int kmax = 0;
    int64_t acc = 0;
    for (int k = SDPCUT_MAX_K; k >= 2; --k) {
        const Bucket &b = h->bucket[k];
        A.set[k] = b.d_set; A.orig[k] = b.d_orig; A.n[k] = b.n;
        acc += (b.n + 255) / 256;
        A.tile_end[k] = acc;
        if (b.n > 0 && k > kmax) kmax = k;
    }
    A.set[0] = A.set[1] = nullptr; A.orig[0] = A.orig[1] = nullptr; A.n[0] = A.n[1] = 0; A.tile_end[0] = A.tile_end[1] = acc;
    if (kmax == 0) return 0;
    A.vars = h->d_vars; A.nv = h->nb_vars; A.L = h->L; A.eig_out = h->d_eig; A.tk = (TopkWs *)tk;
    A.pf_mloc = 0;      // (set with the grid: eig_launch)
    A.spread = 0;
#define EIG_LAUNCH(KM)                                                   \
    do {                                                                 \
        if (tk) eig_launch<KM, true>(h, A, ev_start, ev_stop, pf_k);     \
        else eig_launch<KM, false>(h, A, ev_start, ev_stop, 0);          \
    } while (0)
    switch (kmax) {
    case 2: EIG_LAUNCH(2); break;
    case 3: EIG_LAUNCH(3); break;
    case 4: EIG_LAUNCH(4); break;
    default: EIG_LAUNCH(5); break;
    }
#undef EIG_LAUNCH
    HIP_TRY(h, hipGetLastError());
    return 0;
}
