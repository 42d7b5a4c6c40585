#!/usr/bin/env python3
"""GPU box: lambda_min of every candidate of a recorded round (tests/golden/rounds_*.npz) from the library against LAPACK
(numpy eigvalsh, UPLO = "U", what the reference calls) -- error statistics and the neighbours of the feasibility ranking whose order
differs.  usage: tools/lmin_diag.py rounds_spar125_075_2_d3_s4 18 [19 ...]        (SDPCUT_LIB=... for a variant build)"""
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import sdpcutsel_via_nn_amd as pkg  # noqa: E402
from sdpcutsel_via_nn_amd import _capi, harness  # noqa: E402
from oracle import cutsel_oracle as oracle  # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(gold, sys.argv[1] + ".npz"))
name, dim, sel = str(g["name"]), int(g["dim"]), int(g["sel_size"])
inst = harness.parse_boxqp(os.path.join(gold, "instances", name + ".in"))
n, L = inst["nb_vars"], inst["nb_lifted"]
sc = pkg.Scorer(0)
sc.set_builtin_networks(dim)
sc.set_instance(n, inst["Q_arr"])
N = sc.set_candidates_cover(inst["adj"], dim)
S, ks = sc.get_candidates(np.arange(N))
for r in [int(a) for a in sys.argv[2:]]:
    vv = g["r%02d_vars" % r]
    sc.set_point(vv)
    sc.score(_capi.EIG)
    lam = sc.get_scores(obj=False)[0][:N]
    ref = np.zeros(N)
    for k in np.unique(ks):
        m = np.nonzero(ks == k)[0]
        si = S[m, :k]
        ref[m] = oracle.eigmin_batch(int(k), vv[L:][si], vv[:L][oracle.triu_positions(si, n)])
    d = np.abs(lam - ref)
    print("%s round %d: N %d  max |d| %.3e  mean %.2e  #(|d| > 3e-15) %d  classification differs %d"
          % (sys.argv[1], r, N, d.max(), d.mean(), int((d > 3e-15).sum()), int(((lam < -1e-15) != (ref < -1e-15)).sum())))
    for i in np.argsort(-d)[:5]:
        k = int(ks[i])
        x, X = vv[L:][S[i, :k]], vv[:L][oracle.triu_positions(S[i:i + 1, :k], n)][0]
        print("   cand %d k=%d  lib %.17e  lapack %.17e  all eigenvalues %s" % (i, k, lam[i], ref[i], oracle.get_eigendecomp(k, x, X, False)))

    def head(l):
        v = np.nonzero(l < -1e-15)[0]
        return v[np.argsort(-l[v], kind="stable")][:sel]
    a, b = head(lam), head(ref)
    w = min(len(a), len(b))
    diff = np.flatnonzero(a[:w] != b[:w])
    print("   feasibility head: lengths %d / %d, positions differing %d %s" % (len(a), len(b), diff.size, diff[:12].tolist()))
    for pos in diff[:6]:
        i, j = a[pos], b[pos]
        print("      pos %d: lib picks %d (lib %.17e lapack %.17e), lapack picks %d (lib %.17e lapack %.17e)"
              % (pos, i, lam[i], ref[i], j, lam[j], ref[j]))
sc.close()
