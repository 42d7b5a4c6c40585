// lambda_min ONLY of the (k+1)x(k+1) lifted matrices, one matrix per lane, in registers (round 4).
//
// What the scan of a round asks of the eigen-solver is ONE number per candidate -- numpy.linalg.eigvalsh(...)[0] at
// cut_select_qp.py:796 (LAPACK: Householder tridiagonalisation dsytrd + eigenvalues-only QL dsterf).  Cyclic Jacobi
// (jacobi.h) computes the whole spectrum: ~1170 VALU instructions per 4x4 matrix at 4.5 sweeps, ~2500 per 5x5, ~4500 per
// 6x6, on the same fp64 pipe as the MLP's MFMAs.  Here:
//
//  1. Householder tridiagonalisation T = H^T A H (D - 2 reflections, backward stable: what dsytrd does).
//  2. A certified lower bound of lambda_min(T) by block Gershgorin with the blocks {0,1}, {2}, ..., {D-1}: a lifted matrix
//     [[1, x^T], [x, X]] ~ (1, x)(1, x)^T has ONE large eigenvalue, which the first reflection moves into the top 2x2
//     block; the bound is ~0.05 below lambda_min where the trace bounds sit ~1 below.
//  3. Laguerre's iteration on p(l) = det(T - l I) from that bound.  p, p', p'' come from the three-term recurrence
//     p_i = (d_i - l) p_{i-1} - e_{i-1}^2 p_{i-2} (no divisions).  All roots are real, so from the left of the smallest one
//     the iterates increase monotonically, never pass it, and converge cubically.  The recurrence is backward stable in
//     Wilkinson's sense -- the computed p_i are the exact leading minors of a matrix within a few ulp of T - l I -- so left
//     of lambda_min - c eps ||T|| every computed p_i is positive whatever the multiplicities: the limit is lambda_min(T)
//     to a few ulp of ||T||, the accuracy class of LAPACK and of Jacobi (tools/lmin_proto.py: max |d| vs LAPACK
//     1.2-2.0e-15 on the bench lists, against 1.3-2.0e-15 of LAPACK's own QL run on the same T).
//     A step is "the last one" when the NEXT step of a cubically convergent sequence, step^4 / previous^3, is below
//     1e-17, or when it is below a quarter ulp of ||T||: 3.3-3.7 evaluations per matrix, 4.8-5.2 per wave of 64.
//  4. A multiple lambda_min makes the convergence linear.  At structured LP vertices (x = 0.5, X in {0, 0.5}: a quarter of the
//     candidates of rounds 1-3, none at generic LP points) T is numerically REDUCIBLE there -- couplings at rounding level --
//     and is iterated block by block (lmin_blocks_cold): every block has simple eigenvalues.  A coupling counts as zero by
//     LAPACK's relative deflation rule or when it is below the stopping tolerance itself (the exactly singular matrices, whose
//     trailing d_i and e_i are all rounding noise).  What is left is a NEARLY multiple lambda_min with couplings above both
//     thresholds: LP noise of 1e-15 inside a singular block, which couples two zero eigenvalues at the very level that decides
//     on which side of -1e-15 lambda_min falls (round 2 of a run: 0.03 % of the 4-variable candidates of spar125-075-1, a
//     quarter of the 3-variable ones of spar125-075-2; none from round 4 on).  Such a lane gives up after five evaluations
//     (ok = false) and the caller runs Jacobi on it -- per lane: a candidate's lambda_min depends on its matrix alone, never
//     on its neighbours in the wave.
//
// ~100 (Householder) + ~30 (bound) + ~45 per evaluation: ~340 instructions per 4x4 matrix, ~620 per 6x6.
#pragma once
#include <hip/hip_runtime.h>

#include "jacobi.h"

#ifndef LMIN_MAX_EVALS
#define LMIN_MAX_EVALS 8
#endif
#ifndef SDPCUT_LMIN_ABS_SPLIT
#define SDPCUT_LMIN_ABS_SPLIT 1.0      // absolute deflation threshold in units of the stopping tolerance (0: relative rule only)
#endif


// sqrt / reciprocal to ~2^-45 (hardware estimate 2^-23 + one Newton step): what a Laguerre step needs -- the step is a
// correction, its error is multiplied by its own size
__device__ __forceinline__ double lmin_sqrt_fast(double x)       // x > 0
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y;
    return fma(fma(-g, g, x), 0.5 * y, g);
}
__device__ __forceinline__ double lmin_rcp_fast(double x)
{
    const double q = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, q, 1.0), q, q);
}

// Laguerre's step from the left of the smallest root of a real-rooted polynomial of degree n (n1 = n - 1):
//   n p / (-p' + sqrt((n-1) ((n-1) p'^2 - n p p'')));  0 when p <= 0 (at the root to rounding)
__device__ __forceinline__ double lmin_step(double n, double n1, double p, double dp, double sp)
{
#pragma clang fp contract(off)
    const double t = fma(n1 * dp, dp, -(n * p) * sp);
    const double den = lmin_sqrt_fast(fma(n1, fmax(t, 0.0), 1e-300)) - dp;
    const double step = (n * p) * lmin_rcp_fast(den);
    return ((p > 0.0) & (den > 0.0)) ? step : 0.0;
}

// T = H^T A H.  Reads and destroys the LOWER triangle of a (a[i][j], i >= j).  d: diagonal, e2: squared off-diagonals,
// ea: their absolute values.
template <int D>
__device__ __forceinline__ void lmin_tridiagonalise(double (&a)[D][D], double (&d)[D], double (&e2)[D - 1], double (&ea)[D - 1])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int c = 0; c < D - 2; ++c) {
        const int m = D - 1 - c;                    // length of the column below the diagonal
        const double x0 = a[c + 1][c];
        double sig = 0.0;
#pragma unroll
        for (int i = c + 2; i < D; ++i) sig = fma(a[i][c], a[i][c], sig);
        const double nrm2 = fma(x0, x0, sig);
        const double nrm = jac_sqrt(nrm2 + 1e-300);                 // (+1e-300: a zero column stays finite)
        e2[c] = nrm2;
        ea[c] = nrm;
        d[c] = a[c][c];
        // v = x - alpha e_1 with alpha = -sign(x0) ||x|| (no cancellation), H = I - beta v v^T, beta = 2 / v^T v = -1 / (alpha v0)
        const double alpha = -copysign(nrm, x0);
        const double v0 = x0 - alpha;
        // nothing to annihilate (the column is already tridiagonal; entries below 1e-140 are zero against a00 = 1): H = I
        const double beta = sig > 1e-280 ? -jac_rcp(alpha * v0) : 0.0;
        double v[D], p[D], w[D];
        v[0] = v0;
#pragma unroll
        for (int i = 1; i < D; ++i) v[i] = (i < m) ? a[c + 1 + (i < m ? i : 0)][c] : 0.0;
        // p = beta S v,  S = trailing block a[c+1.., c+1..] (symmetric, lower triangle stored)
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if (i < m) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j)
                    if (j < m) {
                        const double sij = (i >= j) ? a[c + 1 + i][c + 1 + j] : a[c + 1 + j][c + 1 + i];
                        s = fma(sij, v[j], s);
                    }
                p[i] = beta * s;
            }
        }
        double vp = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i)
            if (i < m) vp = fma(v[i], p[i], vp);
        const double K = 0.5 * beta * vp;
#pragma unroll
        for (int i = 0; i < D; ++i)
            if (i < m) w[i] = fma(-K, v[i], p[i]);
        // S <- S - v w^T - w v^T
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j)
                if (i < m) {
                    double t = a[c + 1 + i][c + 1 + j];
                    t = fma(-v[i], w[j], t);
                    t = fma(-w[i], v[j], t);
                    a[c + 1 + i][c + 1 + j] = t;
                }
    }
    d[D - 2] = a[D - 2][D - 2];
    d[D - 1] = a[D - 1][D - 1];
    const double off = a[D - 1][D - 2];
    e2[D - 2] = off * off;
    ea[D - 2] = fabs(off);
}

// ---- the cold part, OUT OF LINE: block-by-block iteration for numerically reducible T.  Couplings that are the rounding noise
// of the reflections (relative deflation criterion, see lmin_laguerre) are zero: T is block diagonal there and lambda_min
// is the smallest over the blocks.  Exactly this happens at the structured LP vertices (x = 0.5, X in {0, 0.5}: several per
// cent of the candidates), where two blocks often SHARE their smallest eigenvalue -- a multiple root of det(T - l I), on which
// Laguerre's iteration is only linear.  Block by block every root is simple: the recurrence restarts behind every split, every
// block that closes contributes its own Laguerre step (degree nb), and the iterate moves by the smallest of them: a lower
// bound of the distance to the smallest eigenvalue of ANY block, cubically convergent for the block that attains it,
// indifferent to two blocks sharing it.  A real function call (noinline): the register allocation of the hot loop -- every
// wave at a generic LP point -- does not see this code.  Returns NaN when it does not converge either.
template <int D>
struct LminBlocksIn {
    double d[D], e2[D - 1], lam, tol;
    double abs2;         // couplings with e_i^2 <= abs2 are splits as well (the absolute rule, see lmin_laguerre)
    unsigned smask;      // bit i: the coupling between rows i and i + 1 is treated as zero (the relative rule)
    int max_it;
};

#ifndef SDPCUT_LMIN_BLOCKS_INLINE
#define SDPCUT_LMIN_BLOCKS_INLINE 0
#endif
template <int D>
#if SDPCUT_LMIN_BLOCKS_INLINE
__device__ __forceinline__ double lmin_blocks_cold(const LminBlocksIn<D> &in)
#else
__device__ __attribute__((noinline)) double lmin_blocks_cold(LminBlocksIn<D> in)
#endif
{
#pragma clang fp contract(off)
    double lam = in.lam;
    unsigned smask = in.smask;
#pragma unroll
    for (int i = 0; i < D - 1; ++i) smask |= (in.e2[i] <= in.abs2) ? (1u << i) : 0u;
    bool done = false, bad = smask == 0u;      // (a lane that gave up in the hot loop and has no split by either rule: Jacobi)
    double prev3 = 0.0;
    int prev_row = -1;
#pragma unroll 1
    for (int it = 0; it < in.max_it && !done && !bad; ++it) {
        double pm2 = 0.0, pm1 = 1.0, dm2 = 0.0, dm1 = 0.0, sm2 = 0.0, sm1 = 0.0, nb = 0.0;
        double amin = 1e300;
        int row = -1;      // last row of the block whose step is the smallest
        bool pos = true, fin = true;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double c = 0.0;
            if (i > 0) {
                const bool s = ((smask >> (i - 1)) & 1u) != 0u;
                pm1 = s ? 1.0 : pm1;
                dm1 = s ? 0.0 : dm1;
                sm1 = s ? 0.0 : sm1;
                nb = s ? 0.0 : nb;
                c = s ? 0.0 : in.e2[i > 0 ? i - 1 : 0];
            }
            const double dl = in.d[i] - lam;
            const double p = fma(dl, pm1, -(c * pm2));
            const double dp = fma(-c, dm2, fma(dl, dm1, -pm1));
            const double sp = fma(-c, sm2, fma(dl, sm1, -2.0 * dm1));
            pm2 = pm1; pm1 = p;
            dm2 = dm1; dm1 = dp;
            sm2 = sm1; sm1 = sp;
            nb += 1.0;
            const bool close = (i == D - 1) ? true : (((smask >> i) & 1u) != 0u);
            if (i == D - 1 || __any(close)) {      // (uniform over the lanes in here: a closing formula only where some lane closes a block)
                const double ab = lmin_step(nb, nb - 1.0, p, dp, sp);
                const bool take = close & (ab < amin);
                amin = take ? ab : amin;
                row = take ? i : row;
            }
            pos = pos & ((p > 0.0) | close);       // leading minors inside a block
            fin = fin & ((p > 0.0) | !close);      // the blocks' own determinants
        }
        bad = !pos | ((it == 0) & !fin);
        // the last step: a quarter ulp of ||T||, or -- when the SAME block gave the smallest step twice in a row, i.e. these are
        // consecutive steps of one cubically convergent sequence -- the prediction that the next one would be below 1e-17
        const double s2 = amin * amin;
        done = (amin <= in.tol) | ((row == prev_row) & (s2 * s2 <= prev3));
        prev3 = 1e-17 * (s2 * amin);
        prev_row = row;
        lam += bad ? 0.0 : amin;
    }
    return (done && !bad) ? lam : __builtin_nan("");
}

// lambda_min of the symmetric matrix a (lower triangle read and destroyed).  ok = false: not converged within
// LMIN_MAX_EVALS evaluations (a NEARLY multiple lambda_min; or the start was not left of the spectrum, never observed) -- the
// caller runs Jacobi.
template <int D>
__device__ __forceinline__ double lmin_laguerre(double (&a)[D][D], bool &ok)
{
#pragma clang fp contract(off)
    double d[D], e2[D - 1], ea[D - 1];
    lmin_tridiagonalise<D>(a, d, e2, ea);
    // ---- certified lower bound: block Gershgorin, blocks {0, 1}, {2}, ..., {D - 1}
    double scale = fabs(d[0]);
#pragma unroll
    for (int i = 1; i < D; ++i) scale += fabs(d[i]);
#pragma unroll
    for (int i = 0; i < D - 1; ++i) scale = fma(2.0, ea[i], scale);
    const double h = 0.5 * (d[0] + d[1]), g = 0.5 * (d[0] - d[1]);
    const double rr = fma(g, g, e2[0]) + 1e-300;
    const double r = rr * __builtin_amdgcn_rsq(rr);            // (2^-23 is plenty for a bound: the margin below covers it)
    double lam = h - r;                                         // the smaller eigenvalue of the top 2 x 2 block
    if constexpr (D > 2) {
        lam -= ea[1];
#pragma unroll
        for (int i = 2; i < D; ++i) {
            double gi = d[i] - ea[i - 1];
            if (i < D - 1) gi -= ea[i];
            lam = fmin(lam, gi);
        }
    }
    lam = fma(-1e-6, scale, lam);
    const double tol = 5.551115123125783e-17 * scale;          // a quarter ulp of ||T||
    constexpr double n = (double)D, n1 = (double)(D - 1);
    // A coupling is a split (lmin_blocks_cold) by the RELATIVE criterion of LAPACK's dsterf / dsteqr, e_i^2 <= (c eps)^2 |d_i d_{i+1}|
    // (c = 4 here, 1 there): what the reflections leave of an exact zero between rows with O(1) diagonals goes; a coupling of
    // 1e-15 between rows whose diagonals are themselves ~1e-15 -- the LP noise around an exactly singular block, which decides
    // on which side of -1e-15 lambda_min falls (cut_select_qp.py:24, :647) -- stays, and such a lane ends with Jacobi as before.
    unsigned smask = 0;
#pragma unroll
    for (int i = 0; i < D - 1; ++i) smask |= (e2[i] <= 1.9721522630525295e-31 * fabs(d[i] * d[i + 1])) ? (1u << i) : 0u;
    const bool has_split = smask != 0u;
    // lanes with a split sit the hot loop out (done from the start) and go through the cold function behind it
    bool done = has_split, bad = false;
    double prev3 = 0.0, prev_step = 0.0;
#pragma unroll 1
    for (int it = 0; it < LMIN_MAX_EVALS; ++it) {
        if (!__any(!done & !bad)) break;
        // p, p', p'' of det(T - lam I) by the three-term recurrence (first rows written out: p_0 = 1, p_0' = p_0'' = p_1'' = 0,
        // p_1' = -1); pos = all leading minors p_1 .. p_{D-1} positive (they are, left of lambda_min)
        double pm2 = 1.0, pm1 = d[0] - lam;
        double dm2 = 0.0, dm1 = -1.0;
        double sm2 = 0.0, sm1 = 0.0;
        bool pos = pm1 > 0.0;
#pragma unroll
        for (int i = 1; i < D; ++i) {
            const double dl = d[i] - lam;
            double p, dp, sp;
            if (i == 1) {
                p = fma(dl, pm1, -e2[0]);
                dp = -(dl + pm1);
                sp = 2.0;
            } else if (i == 2) {
                p = fma(dl, pm1, -(e2[1] * pm2));
                dp = fma(dl, dm1, -pm1) + e2[1];
                sp = fma(dl, 2.0, -2.0 * dm1);
            } else {
                p = fma(dl, pm1, -(e2[i - 1] * pm2));
                dp = fma(-e2[i - 1], dm2, fma(dl, dm1, -pm1));
                sp = fma(-e2[i - 1], sm2, fma(dl, sm1, -2.0 * dm1));
            }
            pm2 = pm1; pm1 = p;
            dm2 = dm1; dm1 = dp;
            sm2 = sm1; sm1 = sp;
            if (i < D - 1) pos = pos & (p > 0.0);
        }
        // left of the spectrum every leading minor is positive; the first evaluation must find p itself positive too
        bad = bad | (!done & (!pos | ((it == 0) & !(pm1 > 0.0))));
        const double step = lmin_step(n, n1, pm1, dm1, sm1);
        const double s2 = step * step;
        // the last step: below a quarter ulp of ||T||, or cubic convergence says the NEXT one would be (step^4 / previous^3 <= 1e-17)
        const bool conv = (step <= tol) | (s2 * s2 <= prev3);
        // a lane whose FIFTH step is still more than a quarter of its fourth is converging linearly (a multiple or nearly multiple
        // lambda_min shrinks the step by 0.27-0.37 per evaluation, a simple one by orders of magnitude by then: none of 1.2e6
        // generic matrices trips this, tools/lmin_proto.py): it stops holding its wave and goes to Jacobi three evaluations earlier
        bad = bad | ((it == 4) & !done & !conv & (step > 0.25 * prev_step));
        lam = (done | bad) ? lam : lam + step;      // (a lane that gave up stands still: what it hands on must not depend on how long its wave goes on)
        prev3 = 1e-17 * (s2 * step);
        prev_step = step;
        done = done | conv;
    }
    ok = done & !bad;
    // (r4, late) A lane that gave up gets a second deflation rule before it goes to Jacobi, and so do the lanes of the block
    // iteration: e_i^2 <= (SDPCUT_LMIN_ABS_SPLIT tol)^2 -- a coupling below the stopping tolerance moves no eigenvalue by more than
    // the stopping rule already allows.  That is the exactly singular matrix of a structured vertex, whose trailing d_i, e_i are ALL
    // at rounding level: the relative rule has nothing to compare with there, and the two- or threefold zero eigenvalue made these
    // lanes (5 % of the candidates, 70 % of the WAVES of round 2 of spar125-075-1) converge linearly and end in Jacobi.  Asked
    // inside the cold function: the rule costs the hot path nothing.  The iterate a lane that gave up hands over is still a lower
    // bound of its spectrum (and if it is not -- never observed -- the block iteration notices and returns NaN).
    if (!ok || has_split) {      // (skipped by the whole wave when nobody needs it)
        LminBlocksIn<D> in;
#pragma unroll
        for (int i = 0; i < D; ++i) in.d[i] = d[i];
#pragma unroll
        for (int i = 0; i < D - 1; ++i) in.e2[i] = e2[i];
        in.lam = lam; in.tol = tol; in.smask = smask; in.max_it = LMIN_MAX_EVALS + 2;
        in.abs2 = (SDPCUT_LMIN_ABS_SPLIT * SDPCUT_LMIN_ABS_SPLIT) * (tol * tol);
        lam = lmin_blocks_cold<D>(in);
        ok = lam == lam;
    }
    return lam;
}
