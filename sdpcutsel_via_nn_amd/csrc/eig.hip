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
};

template <int K>
__device__ __forceinline__ double eig_tile(const EigArgs &A, int64_t tile, bool *valid_out, int32_t *out_idx)
{
    const int64_t n = A.n[K];
    const int64_t c = tile * 256 + threadIdx.x;
    const bool valid = c < n;
    const int64_t cc = valid ? c : n - 1;
    Cand<K> cd;
    gather_candidate<K>(cd, A.set[K], n, cc, A.vars, nullptr, A.nv, A.L, false);
    *out_idx = A.orig[K][cc];
    *valid_out = valid;
    return candidate_eigmin<K>(cd);
}

// waves per SIMD the kernel is compiled for (hipcc -Rpass-analysis=kernel-resource-usage: 40 / 54 / 64 / 82 VGPRs for
// KMAX = 2 / 3 / 4 / 5 with the histogram, no scratch): 8 waves up to 5x5 matrices, 5 with the 6x6 ones
template <int KMAX> struct EigOcc { static constexpr int W = KMAX <= 4 ? 8 : 5; };

template <int KMAX, bool FUSE>
__global__ __launch_bounds__(256, EigOcc<KMAX>::W) void eig_only_kernel(EigArgs A)
{
    __shared__ uint32_t tk_hist[256];
    __shared__ uint32_t tk_cnt;
    if constexpr (FUSE) {
        tk_hist[threadIdx.x] = 0;
        if (threadIdx.x == 0) tk_cnt = 0;
        __syncthreads();
    }
    uint32_t c_viol = 0;
    const int64_t ntiles = A.tile_end[2];
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        double lam;
        bool valid;
        int32_t out_idx;
        // uniform per workgroup: which class this tile belongs to
        if (KMAX >= 5 && t < A.tile_end[5]) lam = eig_tile<5>(A, t, &valid, &out_idx);
        else if (KMAX >= 4 && t < A.tile_end[4]) lam = eig_tile<(KMAX >= 4 ? 4 : 2)>(A, t - A.tile_end[5], &valid, &out_idx);
        else if (KMAX >= 3 && t < A.tile_end[3]) lam = eig_tile<(KMAX >= 3 ? 3 : 2)>(A, t - A.tile_end[4], &valid, &out_idx);
        else lam = eig_tile<2>(A, t - A.tile_end[3], &valid, &out_idx);
        if (valid) A.eig_out[out_idx] = lam;
        if constexpr (FUSE) {
            const bool viol = valid && lam < SDPCUT_NEG_EIGVAL;
            hist_add_few(tk_hist, (uint32_t)(key_of(-lam) >> 56), viol);
            c_viol += viol;
        }
    }
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
    }
}

// workgroups launched per resident slot: a few tiles per workgroup keep the flush of the histogram rare,
// more workgroups than slots let the dispatcher balance strips whose Jacobi needs one sweep more
#ifndef SDPCUT_EIG_BLOCKS_PER_SLOT
#define SDPCUT_EIG_BLOCKS_PER_SLOT 4
#endif

template <int KMAX, bool FUSE>
static void eig_launch(sdpcut_ctx *h, const EigArgs &A, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const int64_t ntiles = A.tile_end[2];
    int64_t cap = (int64_t)h->n_cu * EigOcc<KMAX>::W * SDPCUT_EIG_BLOCKS_PER_SLOT;
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    if (ev_start || ev_stop)
        hipExtLaunchKernelGGL((eig_only_kernel<KMAX, FUSE>), dim3(grid), dim3(256), 0, h->stream, ev_start, ev_stop, 0, A);
    else
        hipLaunchKernelGGL((eig_only_kernel<KMAX, FUSE>), dim3(grid), dim3(256), 0, h->stream, A);
}

// lambda_min of every candidate of the handle's list at the current point, one launch.
// tk != nullptr: also the leading-digit histogram / violated count of a feasibility selection (TopkWs).
int launch_eig_only(sdpcut_ctx *h, void *tk, hipEvent_t ev_start, hipEvent_t ev_stop)
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
#define EIG_LAUNCH(KM)                                                   \
    do {                                                                 \
        if (tk) eig_launch<KM, true>(h, A, ev_start, ev_stop);           \
        else eig_launch<KM, false>(h, A, ev_start, ev_stop);             \
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
