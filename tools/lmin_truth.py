#!/usr/bin/env python3
"""How far are LAPACK (what the reference calls, cut_select_qp.py:796) and the lambda_min solver of csrc/lmin.h (numpy twin:
tools/lmin_proto.py) from the EXACT smallest eigenvalue?  The entries of a lifted matrix are doubles, i.e. exact rationals: the
characteristic polynomial is formed in rational arithmetic (Faddeev-LeVerrier) and its smallest root refined by Newton's method in
80-digit decimals.  usage: tools/lmin_truth.py rounds_spar125_075_2_d3_s4 18 [count]      (build container, CPU)"""
import os
import sys
from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
getcontext().prec = 80


def exact_lambda_min(A, start):
    n = A.shape[0]
    F = [[Fraction(float(A[i, j])) for j in range(n)] for i in range(n)]
    eye = [[Fraction(int(i == j)) for j in range(n)] for i in range(n)]

    def mul(X, Y):
        return [[sum(X[i][k] * Y[k][j] for k in range(n)) for j in range(n)] for i in range(n)]
    c, Mk = [Fraction(1)], [[Fraction(0)] * n for _ in range(n)]
    for k in range(1, n + 1):
        AM = mul(F, Mk)
        Mk = [[AM[i][j] + c[-1] * eye[i][j] for j in range(n)] for i in range(n)]
        AMk = mul(F, Mk)
        c.append(-sum(AMk[i][i] for i in range(n)) / k)
    cd = [Decimal(x.numerator) / Decimal(x.denominator) for x in c]
    lam = Decimal(float(start))
    for _ in range(80):
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


def main():
    import lmin_proto as P
    from sdpcutsel_via_nn_amd import _capi, harness
    from oracle import cutsel_oracle as oracle
    gold = os.path.join(ROOT, "tests", "golden")
    g = np.load(os.path.join(gold, sys.argv[1] + ".npz"))
    r, count = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    name, dim = str(g["name"]), int(g["dim"])
    inst = harness.parse_boxqp(os.path.join(gold, "instances", name + ".in"))
    n, L = inst["nb_vars"], inst["nb_lifted"]
    S, ks, N = _capi.enumerate_cover(inst["adj"], dim)
    vv = g["r%02d_vars" % r]
    rng = np.random.default_rng(1)
    for k in np.unique(ks):
        m = np.nonzero(ks == k)[0]
        if m.size < 50:
            continue
        m = rng.choice(m, size=min(count, m.size), replace=False)
        A = P.lifted(vv[L:][S[m, :k]], vv[:L][oracle.triu_positions(S[m, :k], n)], int(k))
        lap = np.linalg.eigvalsh(A, UPLO="U")[:, 0]
        lam, ok = P.lambda_min(A)
        el, ep = [], []
        for i in range(m.size):
            t = exact_lambda_min(A[i], lap[i])
            el.append(float(Decimal(float(lap[i])) - t))
            if ok[i]:
                ep.append(float(Decimal(float(lam[i])) - t))
        el, ep = np.abs(el), np.abs(ep)
        print("%s round %d, %d-variable sets (%d sampled): |LAPACK - exact| mean %.2e max %.2e;  |lmin - exact| mean %.2e max %.2e (%d left to Jacobi)"
              % (sys.argv[1], r, k, m.size, el.mean(), el.max(), ep.mean(), ep.max(), int((~ok).sum())), flush=True)


def pairs():
    """the neighbours of the recorded feasibility rankings that the solvers order differently: exact values of both members"""
    import lmin_proto as P
    from sdpcutsel_via_nn_amd import _capi, harness
    from oracle import cutsel_oracle as oracle
    gold = os.path.join(ROOT, "tests", "golden")
    for fn, r, prs in (("rounds_spar125_075_1_d4_s4", 5, [(806014, 803286)]),
                       ("rounds_spar125_075_2_d3_s4", 18, [(26294, 53769), (21539, 126939), (82, 129580)])):
        g = np.load(os.path.join(gold, fn + ".npz"))
        name, dim = str(g["name"]), int(g["dim"])
        inst = harness.parse_boxqp(os.path.join(gold, "instances", name + ".in"))
        n, L = inst["nb_vars"], inst["nb_lifted"]
        S, ks, N = _capi.enumerate_cover(inst["adj"], dim)
        vv = g["r%02d_vars" % r]
        for pr in prs:
            vals = []
            for i in pr:
                k = int(ks[i])
                si = S[i:i + 1, :k]
                A = P.lifted(vv[L:][si], vv[:L][oracle.triu_positions(si, n)], k)[0]
                lapL, lapU = np.linalg.eigvalsh(A, UPLO="L")[0], np.linalg.eigvalsh(A, UPLO="U")[0]
                t = exact_lambda_min(A, lapU)
                vals.append((i, lapU, lapL, t))
                print("%s round %d candidate %7d (%d variables): exact %s   LAPACK 'U' (the reference's call) %.17e (%+.2e)   LAPACK 'L' %+.2e"
                      % (fn, r, i, k, "%.24E" % t, lapU, float(Decimal(float(lapU)) - t), float(Decimal(float(lapL)) - t)))
            print("    exact difference first - second %.3e;  LAPACK 'U' difference %.3e, 'L' difference %.3e"
                  % (float(vals[0][3] - vals[1][3]), vals[0][1] - vals[1][1], vals[0][2] - vals[1][2]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pairs":
        pairs()
    else:
        main()
