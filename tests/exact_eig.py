"""Exact smallest eigenvalue of a lifted matrix (test infrastructure; moved here from tools/lmin_truth.py, r5).

The entries of a lifted matrix [[1, x^T], [x, X]] are doubles, i.e. exact dyadic rationals.  The characteristic polynomial is
formed in INTEGER arithmetic (entries scaled by a common power of two, Faddeev-LeVerrier: every division is exact) and its smallest
root is refined by Newton's method in 80-digit decimals from a double-precision start.  What the reference ranks by
(cut_select_qp.py:796: numpy.linalg.eigvalsh(M, "U")[0], LAPACK) and what csrc/lmin.h computes are both within a few 1e-16 of this
value; the trajectory replay (tests/test_gpu_config3.py) uses it to decide whether two neighbours of a feasibility ranking can be
told apart by the reference's own arithmetic at all.
"""
import math
from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np

getcontext().prec = 80


def lifted_full(k, x_rho, X_rho):
    """symmetric (k+1) x (k+1) matrix [[1, x^T], [x, X]] from x_rho (k) and X_rho (k(k+1)/2, row-major upper triangle)"""
    M = np.zeros((k + 1, k + 1))
    M[0, 0] = 1.0
    M[0, 1:] = x_rho
    iu = np.triu_indices(k)
    M[iu[0] + 1, iu[1] + 1] = X_rho
    return np.triu(M) + np.triu(M, 1).T


def charpoly_exact(A):
    """coefficients c[0..n] (c[0] = 1) of det(lambda I - A) as Fractions, exact for a matrix of doubles"""
    n = A.shape[0]
    # common scale: every entry = integer * 2^-s
    s = 0
    for v in A.ravel():
        if v != 0.0:
            m, e = math.frexp(float(v))          # v = m 2^e, 0.5 <= |m| < 1: v 2^(53 - e) is an integer
            s = max(s, 53 - e)
    Z = [[int(Fraction(float(A[i, j])) * (1 << s)) if s >= 0 else int(A[i, j]) for j in range(n)] for i in range(n)]

    def mul(X, Y):
        return [[sum(X[i][k] * Y[k][j] for k in range(n)) for j in range(n)] for i in range(n)]
    # Faddeev-LeVerrier on the integer matrix Z = 2^s A: M_1 = I, c_1 = -tr(Z); M_k = Z M_{k-1} + c_{k-1} I, c_k = -tr(Z M_k) / k
    c = [1]
    Mk = [[int(i == j) for j in range(n)] for i in range(n)]
    for k in range(1, n + 1):
        ZM = mul(Z, Mk)
        tr = sum(ZM[i][i] for i in range(n))
        assert tr % k == 0
        ck = -tr // k
        c.append(ck)
        Mk = [[ZM[i][j] + (ck if i == j else 0) for j in range(n)] for i in range(n)]
    # det(mu I - Z) = sum c_k mu^(n-k) with mu = 2^s lambda  ->  coefficients in lambda: c_k 2^(-s k)
    return [Fraction(c[k], 1 << (s * k)) for k in range(n + 1)]


def exact_lambda_min(A, start):
    """smallest eigenvalue of the symmetric matrix of doubles A as an 80-digit Decimal; start: a double within ~1e-10 of it
    (any of the solvers under test).  Newton from a point at or below the smallest root of a polynomial with real roots
    converges monotonically; from slightly above it still lands on the nearest root -- the smallest, the start being 1e-15 off."""
    cf = charpoly_exact(np.asarray(A, dtype=np.float64))
    cd = [Decimal(x.numerator) / Decimal(x.denominator) for x in cf]
    lam = Decimal(float(start))
    for _ in range(200):
        p, dp = Decimal(0), Decimal(0)
        for a in cd:
            dp = dp * lam + p
            p = p * lam + a
        if dp == 0:
            break
        step = p / dp
        lam -= step
        if abs(step) < Decimal(10) ** -60:
            break
    return lam


def exact_lambda_min_of(k, x_rho, X_rho, start):
    return exact_lambda_min(lifted_full(k, x_rho, X_rho), start)
