// Gather of one candidate's slices from the instance tables and its lambda_min -- shared by the
// scoring kernels (score.hip) and the eigenvalue-only kernel of the feasibility rounds (eig.hip).
#pragma once
#include "common.h"
#include "jacobi.h"
#include "lmin.h"

// ------------------------------------------------------------------------------------------
// gather of one candidate (cut_select_qp.py:529-540 record + :573-575 slices)
template <int K>
struct Cand {
    static constexpr int M = K * (K + 1) / 2;
    double x[K];
    double X[M];
    double q[M];     // Q_slice (already divided by max_elem)
    double max_elem;
    double negSM;    // (-S) * max_elem
};

template <int K>
__device__ __forceinline__ void load_index_set(int32_t (&s)[K], const int32_t *set, int64_t n, int64_t c)
{
#pragma unroll
    for (int a = 0; a < K; ++a) s[a] = set[(int64_t)a * n + c];
}

template <int K>
__device__ __forceinline__ void gather_candidate(Cand<K> &cd, const int32_t (&s)[K], const double *vars,
                                                 const double *Q, int32_t nv, int64_t L, bool want_q)
{
    constexpr int M = K * (K + 1) / 2;
#pragma unroll
    for (int a = 0; a < K; ++a) cd.x[a] = vars[L + s[a]];
    int32_t pos[M];
    {
        int m = 0;
#pragma unroll
        for (int a = 0; a < K; ++a) {
            // packed row-major upper-triangle position, cut_select_qp.py:531
            const int32_t rowbase = nv * s[a] - (s[a] * (s[a] + 1)) / 2;
#pragma unroll
            for (int b = a; b < K; ++b) pos[m++] = rowbase + s[b];
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) cd.X[m] = vars[pos[m]];
    if (want_q) {
        double amax = 0.0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            cd.q[m] = Q[pos[m]];
            amax = fmax(amax, fabs(cd.q[m]));
        }
        double me = (double)K * amax;       // :536  (exact: K * |integer-ish|, one rounding)
        if (me == 0.0) me += 1.0;           // :537
        cd.max_elem = me;
        // reference operation order, no contraction:  S = ((0 + q0*X0) + q1*X1) + ...
        {
#pragma clang fp contract(off)
            double S = 0.0;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                cd.q[m] = cd.q[m] / me;     // np.divide, :538
                S = S + cd.q[m] * cd.X[m];  // :575
            }
            cd.negSM = (-S) * me;
        }
    }
}

template <int K>
__device__ __forceinline__ void gather_candidate(Cand<K> &cd, const int32_t *set, int64_t n, int64_t c,
                                                 const double *vars, const double *Q, int32_t nv,
                                                 int64_t L, bool want_q)
{
    int32_t s[K];
    load_index_set<K>(s, set, n, c);
    gather_candidate<K>(cd, s, vars, Q, nv, L, want_q);
}

// SDPCUT_LMIN = 1 (r4): lambda_min by Householder + Laguerre (lmin.h), Jacobi only for the lanes it hands back (multiple or
// nearly multiple lambda_min); 0: Jacobi for everybody (rounds 1-3).
#ifndef SDPCUT_LMIN
#define SDPCUT_LMIN 1
#endif

template <int K>
__device__ __forceinline__ double candidate_eigmin(const Cand<K> &cd)
{
    double a[K + 1][K + 1], v[K + 1][K + 1];
    fill_lifted<K>(a, cd.x, cd.X);
#if SDPCUT_LMIN
    bool ok;
    double lam = lmin_laguerre<K + 1>(a, ok);
    if (!ok) {      // (skipped by the whole wave when nobody needs it)
        fill_lifted<K>(a, cd.x, cd.X);
        jacobi_eig<K + 1, false>(a, v);
        lam = diag_min<K + 1>(a);
    }
    return lam;
#else
    jacobi_eig<K + 1, false>(a, v);
    return diag_min<K + 1>(a);
#endif
}

// The same for the two hot kernels (eig_only_kernel, score_mfma_body).  The lanes Jacobi is left with go through a REAL function
// call (noinline): they gather their matrix again from the index set -- same tables, same values, same lambda_min as the inline
// form above -- so that neither x / X (k(k+3)/2 doubles, 40 registers at k = 5) nor Jacobi's own (k+1)^2 working set weigh on
// the register allocation of the path every wave at a generic LP point takes.
template <int K>
__device__ __attribute__((noinline)) double candidate_eigmin_jacobi_cold(const int32_t *sp, const double *vars, int32_t nv, int64_t L)
{
    int32_t s[K];
#pragma unroll
    for (int a = 0; a < K; ++a) s[a] = sp[a];
    Cand<K> again;
    gather_candidate<K>(again, s, vars, nullptr, nv, L, false);
    double a[K + 1][K + 1], v[K + 1][K + 1];
    fill_lifted<K>(a, again.x, again.X);
    jacobi_eig<K + 1, false>(a, v);
    return diag_min<K + 1>(a);
}

template <int K>
struct IndexSetArg { int32_t s[K]; };
template <int K>
__device__ __attribute__((noinline)) double candidate_eigmin_jacobi_cold(IndexSetArg<K> is, const double *vars, int32_t nv, int64_t L)
{
    Cand<K> again;
    gather_candidate<K>(again, is.s, vars, nullptr, nv, L, false);
    double a[K + 1][K + 1], v[K + 1][K + 1];
    fill_lifted<K>(a, again.x, again.X);
    jacobi_eig<K + 1, false>(a, v);
    return diag_min<K + 1>(a);
}

template <int K>
__device__ __forceinline__ double candidate_eigmin(const Cand<K> &cd, const int32_t (&s)[K], const double *vars, int32_t nv, int64_t L)
{
#if SDPCUT_LMIN
    double a[K + 1][K + 1];
    fill_lifted<K>(a, cd.x, cd.X);
    bool ok;
    double lam = lmin_laguerre<K + 1>(a, ok);
    if (!ok) {
        IndexSetArg<K> is;
#pragma unroll
        for (int i = 0; i < K; ++i) is.s[i] = s[i];
        lam = candidate_eigmin_jacobi_cold<K>(is, vars, nv, L);
    }
    return lam;
#else
    return candidate_eigmin<K>(cd);
#endif
}
