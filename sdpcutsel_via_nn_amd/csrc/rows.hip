// Epilogues of a selection round: the eigen-cut rows of the selected candidates
// (cut_select_qp.py:737-750) -- as padded rows (cut_rows_kernel, round_rows_kernel) or assembled into
// one CSR block on the device (round_csr_kernel: SURVEY 8 f row 4, the replacement of the per-cut
// SparsePair loop of :747-754) -- written straight into pinned host memory, and the LP point's way in.
#include <hip/hip_ext.h>

#include "common.h"
#include "gather.h"
#include "topk_dev.h"

// LDS traffic private to one wave needs no workgroup barrier (see score.hip)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------
// Eigen-cut rows of selected candidates (cut_select_qp.py:737-750), one lane per cut.
template <int K>
__device__ __forceinline__ void cut_row_one(const int32_t *s5, const double *vars, int32_t nv, int64_t L,
                                            double *lam_out, double *coef, double *rhs, int64_t *cols, const double *lam_known = nullptr)
{
    constexpr int M = K * (K + 1) / 2;
    constexpr int D = K + 1;
    double x[K], X[M];
    int32_t s[K];
#pragma unroll
    for (int a = 0; a < K; ++a) {
        s[a] = s5[a];
        x[a] = vars[L + s[a]];
        cols[a] = L + s[a];
    }
    {
        int m = 0;
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const int32_t rowbase = nv * s[a] - (s[a] * (s[a] + 1)) / 2;
#pragma unroll
            for (int b = a; b < K; ++b) {
                X[m] = vars[rowbase + s[b]];
                cols[K + m] = rowbase + s[b];
                ++m;
            }
        }
    }
    double a[D][D];
    fill_lifted<K>(a, x, X);
    double lam = 0.0;
    double ev[D];
    bool have = false;
#ifndef SDPCUT_ROWS_INVERSE_ITERATION
#define SDPCUT_ROWS_INVERSE_ITERATION 1
#endif
    if (SDPCUT_ROWS_INVERSE_ITERATION && (lam_known || SDPCUT_LMIN)) {
        // the scoring pass has computed lambda_min of this candidate (the value the selection ranked by): its eigenvector by
        // inverse iteration (jacobi.h: ~700 instead of ~6700 instructions of a wave that has its SIMD to itself).  (r4) A round
        // that ranked without eigenvalues (optimality; explicit sdpcut_cut_rows on an unscored list) computes lambda_min here
        // with the solver the scoring kernels use (lmin.h) instead of running Jacobi with vectors for the whole spectrum.
        bool ok = true;
        if (lam_known) {
            lam = *lam_known;
        } else {
            double b[D][D];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) b[i][j] = a[i][j];
            lam = lmin_laguerre<D>(b, ok);
        }
        have = ok && min_eigvec_known<D>(a, lam, ev) <= 1e-12;
#ifdef SDPCUT_ABL_NOFALLBACK      // timing experiment (tools/build_ablation.sh, ABL_SRC=rows): wrong on multiple eigenvalues
        have = true;
#endif
    }
    if (!have) {      // no lambda_min at hand (optimality rounds, explicit sdpcut_cut_rows) -- or, never observed, a residual too large
        double v[D][D];
        jacobi_eig<D, true>(a, v);
        // eigenvector of the smallest eigenvalue (first minimum, like LAPACK's ascending order)
        lam = a[0][0];
#pragma unroll
        for (int i = 0; i < D; ++i) ev[i] = v[i][0];
#pragma unroll
        for (int j = 1; j < D; ++j) {
            const bool less = a[j][j] < lam;
            lam = less ? a[j][j] : lam;
#pragma unroll
            for (int i = 0; i < D; ++i) ev[i] = less ? v[i][j] : ev[i];
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) ev[i] = (fabs(ev[i]) <= -SDPCUT_NEG_EIGVAL) ? 0.0 : ev[i];  // :744
    {
#pragma clang fp contract(off)
        int m = 0;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = (i > 1 ? i : 1); j < D; ++j) {     // :745-746
                coef[m++] = (i != j) ? ev[i] * ev[j] * 2 : ev[i] * ev[j];
            }
        *rhs = -ev[0] * ev[0];
    }
    *lam_out = lam;
}

// count may be an upper bound: if d_limit != NULL only min(count, *d_limit) rows exist (the
// device-side length of a ranking that the host has not read yet).  coef rows have stride
// coef_ld >= k + k(k+1)/2 of the largest candidate; cols (stride SDPCUT_ROW_LD) is optional.
__global__ __launch_bounds__(64) void cut_rows_kernel(int64_t count, const int64_t *d_limit, const int64_t *idx,
                                                      int64_t idx_base, int64_t n_local, const int32_t *set5,
                                                      const int32_t *ks,
                                                      const double *vars, int32_t nv, int64_t L, double *lam,
                                                      double *coef, int coef_ld, double *rhs, int64_t *cols,
                                                      int32_t *ks_out, const double *eig)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (d_limit && *d_limit < count) count = *d_limit;
    if (i >= count) return;
    const int64_t c = idx[i] - idx_base;
    if (c < 0 || c >= n_local) {   // a candidate of another shard (sdpcut_shard_finish_round): no row here
        ks_out[i] = 0;
        lam[i] = __builtin_nan("");
        rhs[i] = 0.0;
        for (int m = 0; m < coef_ld; ++m) coef[i * coef_ld + m] = 0.0;
        if (cols)
            for (int m = 0; m < SDPCUT_ROW_LD; ++m) cols[i * SDPCUT_ROW_LD + m] = -1;
        return;
    }
    const int k = ks[c];
    const int32_t *s5 = set5 + c * 5;
    double co[SDPCUT_ROW_LD];
    int64_t cl[SDPCUT_ROW_LD];
#pragma unroll
    for (int m = 0; m < SDPCUT_ROW_LD; ++m) { co[m] = 0.0; cl[m] = -1; }
    ks_out[i] = k;
    switch (k) {
    case 2: cut_row_one<2>(s5, vars, nv, L, lam + i, co, rhs + i, cl, eig ? eig + c : nullptr); break;
    case 3: cut_row_one<3>(s5, vars, nv, L, lam + i, co, rhs + i, cl, eig ? eig + c : nullptr); break;
    case 4: cut_row_one<4>(s5, vars, nv, L, lam + i, co, rhs + i, cl, eig ? eig + c : nullptr); break;
    default: cut_row_one<5>(s5, vars, nv, L, lam + i, co, rhs + i, cl, eig ? eig + c : nullptr); break;
    }
#pragma unroll
    for (int m = 0; m < SDPCUT_ROW_LD; ++m) {
        if (m < coef_ld) coef[i * coef_ld + m] = co[m];
        if (cols) cols[i * SDPCUT_ROW_LD + m] = cl[m];
    }
}

// Epilogue of a fused round (sdpcut_select_round): the rows of the ranking head together with
// its ids, scores and the four ranking counters go straight into the caller-visible block --
// pinned host memory mapped into the device, so the stores ARE the device-to-host transfer (no
// SDMA hand-off, no extra copy launch).  Layout (cap entries): 64 B counters | idx | score | lam |
// rhs | coef [cap][coef_ld] | ks.  Coefficient rows are staged in LDS and leave as contiguous
// coalesced stores.  The kernel also zeroes the top-k workspace the NEXT round will use.
// The sharded round (sdpcut_shard_finish_round) uses it with d_c4 == NULL (all cap entries exist,
// those of other shards get ks = 0 / lam = NaN) and a header of world x 64 bytes written elsewhere.
__global__ __launch_bounds__(64) void round_rows_kernel(int64_t cap, const int64_t *d_c4, const int64_t *idx,
                                                        const double *score, int64_t idx_base, int64_t n_local,
                                                        const int32_t *set5, const int32_t *ks, const double *vars,
                                                        int32_t nv, int64_t L, int coef_ld, char *block,
                                                        int64_t hdr_bytes, uint64_t *zero_ptr, int zero_words,
                                                        int64_t done_serial, uint32_t *done_ticket, const double *eig)
{
    __shared__ double tile[64 * SDPCUT_ROW_LD];
    const int lane = threadIdx.x;
    for (int w = blockIdx.x * 64 + lane; w < zero_words; w += gridDim.x * 64) zero_ptr[w] = 0ull;
    int64_t *o_c4 = (int64_t *)block;
    int64_t *o_idx = (int64_t *)(block + hdr_bytes);
    double *o_score = (double *)(o_idx + cap);
    double *o_lam = o_score + cap;
    double *o_rhs = o_lam + cap;
    double *o_coef = o_rhs + cap;
    int32_t *o_ks = (int32_t *)(o_coef + cap * coef_ld);
    const int64_t first = (int64_t)blockIdx.x * 64;
    const int64_t i = first + lane;
    // (requested before the head's length is known: one dependent trip to memory less on the critical path of a
    // kernel that is a chain of them; an entry beyond the head is never dereferenced)
    const int64_t gid_any = i < cap ? idx[i] : 0;
    const double score_any = i < cap ? score[i] : 0.0;
    int64_t limit = cap;
    if (d_c4) {
        if (blockIdx.x == 0 && lane < 7) o_c4[lane] = d_c4[lane];     // counters, strong count, mode (TopkWs::counters)
        limit = d_c4[3];
        if (limit > cap) limit = cap;
    }
    if (i < limit) {
        const int64_t gid = gid_any;
        const int64_t c = gid - idx_base;
        double co[SDPCUT_ROW_LD];
        int64_t cl[SDPCUT_ROW_LD];
#pragma unroll
        for (int m = 0; m < SDPCUT_ROW_LD; ++m) co[m] = 0.0;
        double lam = __builtin_nan(""), rhs = 0.0;
        int k = 0;
        // ids and scores go out first: the kernel ends in a burst of 0.54 MB over PCIe (~8 us at the link's rate, most of
        // what the kernel takes beyond its launch) -- what is known before the eigenvectors travels while they are computed
        o_idx[i] = gid;
        o_score[i] = score_any;
        if (c >= 0 && c < n_local) {
            k = ks[c];
            const int32_t *s5 = set5 + c * 5;
            switch (k) {
            case 2: cut_row_one<2>(s5, vars, nv, L, &lam, co, &rhs, cl, eig ? eig + c : nullptr); break;
            case 3: cut_row_one<3>(s5, vars, nv, L, &lam, co, &rhs, cl, eig ? eig + c : nullptr); break;
            case 4: cut_row_one<4>(s5, vars, nv, L, &lam, co, &rhs, cl, eig ? eig + c : nullptr); break;
            default: cut_row_one<5>(s5, vars, nv, L, &lam, co, &rhs, cl, eig ? eig + c : nullptr); break;
            }
        }
        o_lam[i] = lam;
        o_rhs[i] = rhs;
        o_ks[i] = k;
#pragma unroll
        for (int m = 0; m < SDPCUT_ROW_LD; ++m)
            if (m < coef_ld) tile[lane * coef_ld + m] = co[m];
    }
    wave_lds_sync();
    const int64_t nlive = (limit - first < 64) ? limit - first : 64;
    const int total = nlive > 0 ? (int)nlive * coef_ld : 0;
    for (int w = lane; w < total; w += 64) o_coef[first * coef_ld + w] = tile[w];
    if (done_serial) {
        // completion word for the polling host: every workgroup makes its stores to the host block
        // visible system-wide, then takes a ticket; the last one publishes the round's serial number
        __threadfence_system();
        if (lane == 0) {
            const uint32_t t = __hip_atomic_fetch_add(done_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                __hip_atomic_store(done_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next round
                __threadfence_system();
                __hip_atomic_store(o_c4 + 7, done_serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Epilogue of a fused round in CSR form (sdpcut_round_csr): what _gen_eigcuts_selected appends to the LP,
// cut_select_qp.py:737-754 -- for the ranked head: eigh, skip unless lambda_min < -1e-15, zero the tiny
// components, coefficients on the columns [L + i for i in set_inds] + Xarr_inds, rhs -v0^2 -- as ONE block
// of rows in compressed sparse row form, assembled on the device and stored straight into the pinned host
// block.  The host-side assembly of the padded rows (boolean masks over [5000, 20] arrays, column indices
// from the index sets) cost 1-3 ms per round, more than the whole device side of the round.
//
// One lane per head entry, 64 entries per workgroup.  A cut's position in the block (row number, offset of
// its non-zeros) is the number of cuts / non-zeros in front of it in head order: inside the workgroup a
// wave scan, across workgroups a look-back over the aggregates the workgroups in front have published
// (word = round serial | cuts | non-zeros: the serial makes a word of an earlier round invisible, nothing
// is zeroed).  A workgroup only ever waits for workgroups with a LOWER index, which the dispatcher started
// before it: the wait cannot deadlock whatever else runs on the device.
struct RoundCsrArgs {
    int64_t cap;               // head entries the block is laid out for
    const int64_t *d_c4;       // TopkWs::counters of the selection (k_eff at [3]); NULL: `limit` entries exist
    int64_t limit;
    const int64_t *idx;        // [cap] global candidate ids of the head
    const double *score;       // [cap]
    int64_t idx_base, n_local;
    const int32_t *set5;       // [N][5] index sets in caller order
    const int32_t *ks;         // [N]
    const double *vars;
    int32_t nv;
    int64_t L;
    const double *eig;         // [N] lambda_min of every candidate at this point if the scoring pass computed it, else NULL
    // outputs: device view of the pinned host block
    int64_t *o_hdr;            // [16]: 0..6 selection counters, 7 completion serial, 8 cuts, 9 non-zeros, 10 look-back gave up
                               // (the host zeroes word 10 before the launch)
    int64_t *o_idx;            // [cap]
    double *o_score;           // [cap]
    double *o_lam;             // [cap]
    int32_t *o_ks;             // [cap]
    int32_t *o_sets;           // [cap][5]
    int32_t *o_row_entry;      // [cap]    head position of cut r
    int32_t *o_indptr;         // [cap + 1]
    double *o_rhs;             // [cap]
    int32_t *o_indices;        // [cap * ld]
    double *o_values;          // [cap * ld]
    uint64_t *zero_ptr;        // top-k workspace of the next round (zeroed here, see round_rows_kernel)
    int zero_words;
    int64_t serial;
    uint32_t *done_ticket;
    uint64_t *agg;             // [gridDim.x] look-back words
};

#define CSR_SPIN_LIMIT (1 << 22)
#ifndef ROUND_MIN_BLOCKS
#define ROUND_MIN_BLOCKS 8      // workgroups of an epilogue launch at least (they share the zeroing of the next round's top-k workspace)
#endif

__global__ __launch_bounds__(64) void round_csr_kernel(RoundCsrArgs R)
{
    __shared__ double s_val[64 * SDPCUT_ROW_LD];
    __shared__ int32_t s_ind[64 * SDPCUT_ROW_LD];
    const int lane = threadIdx.x;
    for (int w = blockIdx.x * 64 + lane; w < R.zero_words; w += gridDim.x * 64) R.zero_ptr[w] = 0ull;
    const int64_t first = (int64_t)blockIdx.x * 64;
    const int64_t i = first + lane;
    const int64_t gid_any = i < R.cap ? R.idx[i] : 0;
    const double score_any = i < R.cap ? R.score[i] : 0.0;
    int64_t limit = R.limit;
    if (R.d_c4) {
        if (blockIdx.x == 0 && lane < 7) R.o_hdr[lane] = R.d_c4[lane];
        limit = R.d_c4[3];
    }
    if (limit > R.cap) limit = R.cap;
    double co[SDPCUT_ROW_LD];
    int64_t cl[SDPCUT_ROW_LD];
#pragma unroll
    for (int m = 0; m < SDPCUT_ROW_LD; ++m) { co[m] = 0.0; cl[m] = 0; }
    double lam = __builtin_nan(""), rhs = 0.0;
    int k = 0;
    if (i < limit) {
        R.o_idx[i] = gid_any;
        R.o_score[i] = score_any;
        const int64_t c = gid_any - R.idx_base;
        int32_t s5[5] = {-1, -1, -1, -1, -1};
        if (c >= 0 && c < R.n_local) {
            k = R.ks[c];
            const int32_t *sp = R.set5 + c * 5;
#pragma unroll
            for (int a = 0; a < 5; ++a) s5[a] = sp[a];
            switch (k) {
            case 2: cut_row_one<2>(sp, R.vars, R.nv, R.L, &lam, co, &rhs, cl, R.eig ? R.eig + c : nullptr); break;
            case 3: cut_row_one<3>(sp, R.vars, R.nv, R.L, &lam, co, &rhs, cl, R.eig ? R.eig + c : nullptr); break;
            case 4: cut_row_one<4>(sp, R.vars, R.nv, R.L, &lam, co, &rhs, cl, R.eig ? R.eig + c : nullptr); break;
            default: cut_row_one<5>(sp, R.vars, R.nv, R.L, &lam, co, &rhs, cl, R.eig ? R.eig + c : nullptr); break;
            }
        }
        R.o_lam[i] = lam;
        R.o_ks[i] = k;
#pragma unroll
        for (int a = 0; a < 5; ++a) R.o_sets[i * 5 + a] = s5[a];
    }
    const bool keep = i < limit && k > 0 && lam < SDPCUT_NEG_EIGVAL;       // :739
    const int len = keep ? k * (k + 3) / 2 : 0;
    // position inside the workgroup (= one wave)
    const unsigned long long km = __ballot(keep);
    const int my_row = __popcll(km & ((1ull << lane) - 1ull));
    const int wg_rows = __popcll(km);
    int incl = len;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    const int my_off = incl - len;
    const int wg_nnz = __shfl(incl, 63);
    // publish this workgroup's aggregate, then sum those of the workgroups in front
    const uint32_t tag = (uint32_t)R.serial;
    if (lane == 0)
        __hip_atomic_store(&R.agg[blockIdx.x], ((uint64_t)tag << 32) | ((uint64_t)wg_rows << 16) | (uint64_t)wg_nnz, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    int64_t pre_rows = 0, pre_nnz = 0;
    int gave_up = 0;
    for (int b = lane; b < (int)blockIdx.x && !gave_up; b += 64) {
        uint64_t w = __hip_atomic_load(&R.agg[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t it = 0;
        while ((uint32_t)(w >> 32) != tag) {
            __builtin_amdgcn_s_sleep(2);
            if (++it > CSR_SPIN_LIMIT) { gave_up = 1; break; }
            w = __hip_atomic_load(&R.agg[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!gave_up) {
            pre_rows += (int64_t)((w >> 16) & 0xffffull);
            pre_nnz += (int64_t)(w & 0xffffull);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        pre_rows += __shfl_xor(pre_rows, off);
        pre_nnz += __shfl_xor(pre_nnz, off);
        gave_up |= __shfl_xor(gave_up, off);
    }
    if (gave_up && lane == 0) R.o_hdr[10] = 1;      // the block is void; the host launches the assembly once more (capi.hip: csr_again)
    if (keep) {
        const int64_t r = pre_rows + my_row;
        R.o_row_entry[r] = (int32_t)i;
        R.o_indptr[r] = (int32_t)(pre_nnz + my_off);
        R.o_rhs[r] = rhs;
#pragma unroll
        for (int m = 0; m < SDPCUT_ROW_LD; ++m)
            if (m < len) { s_val[my_off + m] = co[m]; s_ind[my_off + m] = (int32_t)cl[m]; }
    }
    wave_lds_sync();
    for (int w = lane; w < wg_nnz; w += 64) {      // contiguous, coalesced stores over PCIe
        R.o_values[pre_nnz + w] = s_val[w];
        R.o_indices[pre_nnz + w] = s_ind[w];
    }
    if (blockIdx.x == gridDim.x - 1 && lane == 0) {
        R.o_indptr[pre_rows + wg_rows] = (int32_t)(pre_nnz + wg_nnz);
        R.o_hdr[8] = pre_rows + wg_rows;
        R.o_hdr[9] = pre_nnz + wg_nnz;
    }
    // completion word for the polling host (see round_rows_kernel)
    __threadfence_system();
    if (lane == 0) {
        const uint32_t t = __hip_atomic_fetch_add(R.done_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(R.done_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(R.o_hdr + 7, R.serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// LP point: mapped host memory -> device table (sdpcut_set_point)
__global__ __launch_bounds__(256) void point_copy_kernel(const double *src, double *dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------
// host launchers
int launch_cut_rows(sdpcut_ctx *h, int64_t count, const int64_t *d_limit, const int64_t *d_idx, int64_t idx_base,
                    double *d_lam, double *d_coef, int coef_ld, double *d_rhs, int64_t *d_cols, int32_t *d_ks)
{
    if (count == 0) return 0;
    const int grid = (int)((count + 63) / 64);
    hipLaunchKernelGGL(cut_rows_kernel, dim3(grid), dim3(64), 0, h->stream, count, d_limit, d_idx, idx_base,
                       h->N, h->d_set_orig, h->d_k, h->d_vars, h->nb_vars, h->L, d_lam, d_coef, coef_ld, d_rhs, d_cols,
                       d_ks, (h->scored & SDPCUT_EIG) ? (const double *)h->d_eig : (const double *)nullptr);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int launch_point_copy(sdpcut_ctx *h, const double *src_mapped, int64_t n)
{
    int64_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(point_copy_kernel, dim3((unsigned)(g < 1 ? 1 : g)), dim3(256), 0, h->stream, src_mapped, h->d_vars, n);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int launch_round_rows(sdpcut_ctx *h, int64_t cap, const int64_t *d_c4, const int64_t *d_idx, const double *d_score,
                      int coef_ld, void *block, int64_t hdr_bytes, int64_t done_serial)
{
    if (cap <= 0) return 0;
    if (done_serial && !h->d_done_ticket) {
        HIP_TRY(h, hipMalloc((void **)&h->d_done_ticket, 64 + 256 * 8));      // + look-back words of round_csr_kernel
        HIP_TRY(h, hipMemsetAsync(h->d_done_ticket, 0, 64 + 256 * 8, h->stream));
    }
    uint64_t *zp = nullptr;
    int zw = 0;
    int rc = topk_alt_ws(h, &zp, &zw);
    if (rc) return rc;
    int grid = (int)((cap + 63) / 64);
    if (grid < ROUND_MIN_BLOCKS) grid = ROUND_MIN_BLOCKS;      // (the workspace zeroing, see launch_round_csr)
    hipLaunchKernelGGL(round_rows_kernel, dim3(grid), dim3(64), 0, h->stream, cap, d_c4, d_idx, d_score, h->base, h->N,
                       h->d_set_orig, h->d_k, h->d_vars, h->nb_vars, h->L, coef_ld, (char *)block, hdr_bytes, zp, zw, done_serial,
                       h->d_done_ticket, (h->scored & SDPCUT_EIG) ? (const double *)h->d_eig : (const double *)nullptr);
    HIP_TRY(h, hipGetLastError());
    h->topk_alt_clean = true;
    return 0;
}


// Layout of the CSR round block for `cap` head entries and rows of at most `ld` non-zeros (offsets in bytes,
// every array 8-byte aligned); the same arithmetic on the host (sdpcut_round_csr) and for the kernel's pointers.
CsrLayout csr_layout(int64_t cap, int ld)
{
    CsrLayout y;
    const size_t c = (size_t)cap;
    auto al = [](size_t v) { return (v + 7) & ~(size_t)7; };
    size_t o = 128;
    y.idx = o; o += c * 8;
    y.score = o; o += c * 8;
    y.lam = o; o += c * 8;
    y.rhs = o; o += c * 8;
    y.values = o; o += c * (size_t)ld * 8;
    y.ks = o; o = al(o + c * 4);
    y.sets = o; o = al(o + c * 20);
    y.row_entry = o; o = al(o + c * 4);
    y.indptr = o; o = al(o + (c + 1) * 4);
    y.indices = o; o = al(o + c * (size_t)ld * 4);
    y.bytes = o;
    return y;
}

int launch_round_csr(sdpcut_ctx *h, int64_t cap, const int64_t *d_c4, int64_t limit, const int64_t *d_idx, const double *d_score,
                     int ld, void *block, int64_t serial)
{
    if (cap <= 0) return 0;
    int grid = (int)((cap + 63) / 64);
    if (grid > 256) return sdpcut_fail(h, SDPCUT_EINVAL, "round_csr: head too long");
    // (r4) the kernel also zeroes the next round's top-k workspace (124 KB): a head of a few entries is ONE workgroup, whose 64 lanes
    // then spend ~20 us on 242 stores each -- most of the epilogue of a QCQP round with 7 cuts.  Workgroups beyond the head only zero.
    if (grid < ROUND_MIN_BLOCKS) grid = ROUND_MIN_BLOCKS;
    if (!h->d_done_ticket) {
        // completion ticket (64 B) + the look-back words of the CSR epilogue (256 x 8 B)
        HIP_TRY(h, hipMalloc((void **)&h->d_done_ticket, 64 + 256 * 8));
        HIP_TRY(h, hipMemsetAsync(h->d_done_ticket, 0, 64 + 256 * 8, h->stream));
    }
    RoundCsrArgs R;
    R.cap = cap; R.d_c4 = d_c4; R.limit = limit; R.idx = d_idx; R.score = d_score; R.idx_base = h->base; R.n_local = h->N;
    R.set5 = h->d_set_orig; R.ks = h->d_k; R.vars = h->d_vars; R.nv = h->nb_vars; R.L = h->L;
    R.eig = (h->scored & SDPCUT_EIG) ? h->d_eig : nullptr;
    const CsrLayout y = csr_layout(cap, ld);
    char *b = (char *)block;
    R.o_hdr = (int64_t *)b;
    R.o_idx = (int64_t *)(b + y.idx); R.o_score = (double *)(b + y.score); R.o_lam = (double *)(b + y.lam);
    R.o_rhs = (double *)(b + y.rhs); R.o_values = (double *)(b + y.values); R.o_ks = (int32_t *)(b + y.ks);
    R.o_sets = (int32_t *)(b + y.sets); R.o_row_entry = (int32_t *)(b + y.row_entry); R.o_indptr = (int32_t *)(b + y.indptr);
    R.o_indices = (int32_t *)(b + y.indices);
    int rc = topk_alt_ws(h, &R.zero_ptr, &R.zero_words);
    if (rc) return rc;
    R.serial = serial; R.done_ticket = h->d_done_ticket; R.agg = (uint64_t *)((char *)h->d_done_ticket + 64);
    hipLaunchKernelGGL(round_csr_kernel, dim3(grid), dim3(64), 0, h->stream, R);
    HIP_TRY(h, hipGetLastError());
    h->topk_alt_clean = true;
    return 0;
}
