#!/usr/bin/env python3
"""numpy prototype of the lambda_min-only solver of csrc/lmin.h (round 4) and its replay on the reference's trajectories.

What cut_select_qp.py:796 asks of the eigen-solver is ONE number per candidate, eigvalsh(...)[0].  Cyclic Jacobi computes the whole
spectrum (~1170 VALU instructions per 4x4 matrix).  Here: Householder tridiagonalisation (what LAPACK's dsytrd does in front of its QL
iteration) followed by Laguerre's iteration on the characteristic polynomial of the tridiagonal matrix T, evaluated with the three-term
recurrence p_i = (d_i - l) p_{i-1} - e_{i-1}^2 p_{i-2} (and its first two derivatives), started from a certified lower bound of
lambda_min (block Gershgorin: the top 2x2 block carries the one large eigenvalue of a lifted matrix [[1, x^T], [x, X]] ~ (1, x)(1, x)^T).
From the left of the smallest root Laguerre's iterates increase monotonically and converge cubically; the recurrence is backward
stable in Wilkinson's sense (the computed p_i are the exact minors of a matrix within a few ulp of T), so left of lambda_min - c eps
all computed p_i are positive whatever the multiplicities: the limit is lambda_min of T to a few ulp of ||T||.

    python tools/lmin_proto.py synthetic            accuracy + evaluations per matrix / per wave of 64 on the bench lists, k = 2..5
    python tools/lmin_proto.py clusters             prescribed spectra with gaps 1e-1 .. 0 (exact multiple lambda_min)
    python tools/lmin_proto.py goldens              the structured vertices of tests/golden/inst_boxqp.npz
    python tools/lmin_proto.py replay [file ...]    every recorded round of tests/golden/rounds_*.npz: classification (lambda < -1e-15),
                                                    positions of the feasibility ranking's head that differ from LAPACK's
Build-container tool (numpy only); arithmetic mirrors the device code up to fused multiply-adds.
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
np.seterr(all="ignore")
EPS = 2.220446049250313e-16


def lifted(x, X, k):
    B, D = x.shape[0], k + 1
    A = np.zeros((B, D, D))
    A[:, 0, 0] = 1.0
    A[:, 0, 1:] = x
    A[:, 1:, 0] = x
    m = 0
    for i in range(k):
        for j in range(i, k):
            A[:, i + 1, j + 1] = X[:, m]
            A[:, j + 1, i + 1] = X[:, m]
            m += 1
    return A


def tridiagonalise(A):
    """Householder, column by column -> d [B, n], e2 [B, n-1] (squared off-diagonals), ea [B, n-1] (their absolute values)"""
    A = A.copy()
    B, n, _ = A.shape
    d, e2, ea = np.zeros((B, n)), np.zeros((B, n - 1)), np.zeros((B, n - 1))
    for c in range(n - 2):
        x0 = A[:, c + 1, c]
        sig = (A[:, c + 2:, c] ** 2).sum(axis=1)
        nrm2 = x0 * x0 + sig
        nrm = np.sqrt(nrm2)
        e2[:, c], ea[:, c] = nrm2, nrm
        alpha = -np.copysign(nrm, x0)
        v0 = x0 - alpha
        on = sig > 1e-280                                   # nothing to annihilate: H = I
        beta = np.where(on, -1.0 / np.where(on, alpha * v0, 1.0), 0.0)
        v = A[:, c + 1:, c].copy()
        v[:, 0] = v0
        S = A[:, c + 1:, c + 1:]
        p = beta[:, None] * np.einsum('bij,bj->bi', S, v)
        K = 0.5 * beta * np.einsum('bi,bi->b', p, v)
        w = p - K[:, None] * v
        S -= v[:, :, None] * w[:, None, :] + w[:, :, None] * v[:, None, :]
        d[:, c] = A[:, c, c]
    d[:, n - 2], d[:, n - 1] = A[:, n - 2, n - 2], A[:, n - 1, n - 1]
    off = A[:, n - 1, n - 2]
    e2[:, n - 2], ea[:, n - 2] = off * off, np.abs(off)
    return d, e2, ea


def lower_bound(d, e2, ea):
    """certified lower bound of lambda_min(T): block Gershgorin with the blocks {0, 1}, {2}, ..., {n-1}"""
    B, n = d.shape
    h, g = 0.5 * (d[:, 0] + d[:, 1]), 0.5 * (d[:, 0] - d[:, 1])
    r = np.sqrt(g * g + e2[:, 0])
    big = h + r
    mu = np.where(big > 0, (d[:, 0] * d[:, 1] - e2[:, 0]) / np.where(big > 0, big, 1.0), h - r)
    scale = np.abs(d).sum(axis=1) + 2 * ea.sum(axis=1)
    if n == 2:
        return mu - 1e-6 * scale, scale
    lb = mu - ea[:, 1]
    for i in range(2, n):
        lb = np.minimum(lb, d[:, i] - ea[:, i - 1] - (ea[:, i] if i < n - 1 else 0.0))
    return lb - 1e-6 * scale, scale


def poly(d, e2, lam):
    """p, p', p'' of det(T - lam I) and min_i p_i (positive iff lam < lambda_min)"""
    B, n = d.shape
    pm2, pm1 = np.ones(B), d[:, 0] - lam
    dm2, dm1 = np.zeros(B), -np.ones(B)
    sm2, sm1 = np.zeros(B), np.zeros(B)
    pmin = pm1.copy()
    for i in range(1, n):
        dl = d[:, i] - lam
        p = dl * pm1 - e2[:, i - 1] * pm2
        dp = dl * dm1 - pm1 - e2[:, i - 1] * dm2
        sp = dl * sm1 - 2 * dm1 - e2[:, i - 1] * sm2
        pm2, pm1, dm2, dm1, sm2, sm1 = pm1, p, dm1, dp, sm1, sp
        pmin = np.minimum(pmin, p)
    return pm1, dm1, sm1, pmin


K_MAX = int(os.environ.get("K_MAX", "8"))       # evaluations a lane gets; lanes that have not converged by then are left to Jacobi
SPLIT = os.environ.get("SPLIT", "1") == "1"     # deflate negligible couplings: lambda_min = min over the blocks
ABS_SPLIT = float(os.environ.get("ABS_SPLIT", "1"))   # absolute deflation threshold in units of the stopping tolerance (0 = relative rule only)


def lag_step(nb, p, dp, sp):
    t = (nb - 1) * dp * dp - nb * p * sp
    den = np.sqrt((nb - 1) * np.maximum(t, 0.0) + 1e-300) - dp
    return np.where((p > 0) & (den > 0), nb * p / np.where(den > 0, den, 1.0), 0.0)


def _iterate(d, e2, split, lam, tol, max_it, give_up_early):
    """Laguerre from the left, block by block where `split` says so (a lane without a split is one block).  Returns (lam, done,
    evaluations); give_up_early: the hot loop's rule (a fifth step still above a quarter of the fourth: linear convergence)."""
    B, n = d.shape
    e2s = np.where(split, 0.0, e2)
    done = np.zeros(B, dtype=bool)
    slow = np.zeros(B, dtype=bool)
    prev3 = np.zeros(B)
    nev = np.zeros(B, dtype=int)
    prev_row = np.full(B, -1)
    prev_step = np.zeros(B)
    lam = lam.copy()
    for it in range(max_it):
        if (done | slow).all():
            break
        pm2, pm1 = np.zeros(B), np.ones(B)
        dm2, dm1 = np.zeros(B), np.zeros(B)
        sm2, sm1 = np.zeros(B), np.zeros(B)
        nb = np.zeros(B)
        amin = np.full(B, np.inf)
        row = np.full(B, -1)
        for i in range(n):
            if i > 0:
                s = split[:, i - 1]
                pm1, dm1, sm1, nb = np.where(s, 1.0, pm1), np.where(s, 0.0, dm1), np.where(s, 0.0, sm1), np.where(s, 0.0, nb)
            c = e2s[:, i - 1] if i > 0 else 0.0
            dl = d[:, i] - lam
            p = dl * pm1 - c * pm2
            dp = dl * dm1 - pm1 - c * dm2
            sp = dl * sm1 - 2 * dm1 - c * sm2
            pm2, pm1, dm2, dm1, sm2, sm1 = pm1, p, dm1, dp, sm1, sp
            nb = nb + 1
            close = np.ones(B, dtype=bool) if i == n - 1 else split[:, i]
            ab = lag_step(nb, p, dp, sp)
            take = close & (ab < amin)
            amin, row = np.where(take, ab, amin), np.where(take, i, row)
        act = ~done & ~slow
        nev[act] += 1
        step = amin
        s2 = step * step
        conv = (step <= tol) | ((row == prev_row) & (s2 * s2 <= prev3))      # (prediction: consecutive steps of ONE block)
        if give_up_early and it == 4:
            slow |= act & ~conv & (step > 0.25 * prev_step)
        prev_step = np.where(act, step, prev_step)
        lam = np.where(act & ~slow, lam + step, lam)
        prev3 = np.where(act, 1e-17 * s2 * step, prev3)
        prev_row = np.where(act, row, prev_row)
        done |= act & conv & ~slow
    return lam, done, nev


def lambda_min(A, stats=None):
    """-> (lam, ok): ok False = left to Jacobi (a NEARLY multiple lambda_min: linear convergence).
    A reducible T (couplings at rounding level: the structured LP vertices) is handled block by block: every block has simple
    eigenvalues, the step is the smallest of the blocks' Laguerre steps -- a lower bound of the distance to the smallest eigenvalue
    of ANY block, cubically convergent for the block that attains it, indifferent to two blocks sharing it.  As on the device:
    lanes split by LAPACK's relative rule go through the block loop (K_MAX + 2 evaluations), the others through the hot loop
    (K_MAX, early give-up); for a lane that gave up there, and for the lanes of the block loop, a coupling below ABS_SPLIT tol is a split as
    well (the former go through the block loop too, from the iterate they reached)."""
    d, e2, ea = tridiagonalise(A)
    B, n = d.shape
    lam0, scale = lower_bound(d, e2, ea)
    tol = 0.25 * EPS * scale
    # LAPACK's relative deflation criterion (dsterf: e^2 <= eps^2 |d_i d_{i+1}|), with 4 eps
    split = (e2 <= (4 * 0.5 * EPS) ** 2 * np.abs(d[:, :-1] * d[:, 1:])) if SPLIT else np.zeros_like(e2, dtype=bool)
    has_split = split.any(axis=1)
    nosplit = np.zeros_like(split)
    lam = lam0.copy()
    done = np.zeros(B, dtype=bool)
    nev = np.zeros(B, dtype=int)
    h = ~has_split
    if h.any():
        lam[h], done[h], nev[h] = _iterate(d[h], e2[h], nosplit[h], lam0[h], tol[h], K_MAX, True)
    cold = has_split.copy()
    if SPLIT and ABS_SPLIT > 0:
        # ... or, for a lane that gave up, below the stopping tolerance itself (|e| <= tol moves no eigenvalue by more than tol): the
        # exactly singular matrices of structured vertices, whose trailing d_i, e_i are ALL at rounding level
        extra = (e2 <= ((ABS_SPLIT * tol) ** 2)[:, None]) & ((h & ~done) | has_split)[:, None]
        split = split | extra
        cold |= extra.any(axis=1)
    if cold.any():
        l2, d2, n2 = _iterate(d[cold], e2[cold], split[cold], lam[cold], tol[cold], K_MAX + 2, False)
        lam[cold], done[cold] = l2, d2
        nev[cold] += n2
    if stats is not None:
        stats["evals"] = nev
        stats["notdone"] = int((~done).sum())
        stats["split"] = int(has_split.sum())
    return lam, done


def _workload(k, count=200000, nv=100):
    from sdpcutsel_via_nn_amd import synthetic
    from oracle import cutsel_oracle as oracle
    wl = synthetic.make_workload(nb_vars=nv, k=k, count=count, seed=7)
    L = nv * (nv + 1) // 2
    si, vv = wl["set_inds"][:, :k], wl["vars_values"]
    return lifted(vv[L:][si], vv[:L][oracle.triu_positions(si, nv)], k)


def _report(tag, A):
    ref = np.linalg.eigvalsh(A)[:, 0]
    st = {}
    lam, ok = lambda_min(A, st)
    lam = np.where(ok, lam, ref)          # (lanes left to Jacobi: LAPACK's value stands in)
    nev = st["evals"]
    pad = (-nev.shape[0]) % 64
    w = np.concatenate([nev, np.zeros(pad, dtype=int)]).reshape(-1, 64).max(axis=1)
    cls = int(((lam < -1e-15) != (ref < -1e-15)).sum())
    print("%-34s n %7d  max|d| %.2e  class. differs %d  left to Jacobi %d  evaluations mean %.2f  per wave %.2f  max %d"
          % (tag, A.shape[0], np.abs(lam - ref).max(), cls, st["notdone"], nev.mean(), w.mean(), nev.max()), flush=True)
    return lam, ref


def cmd_synthetic():
    for k in (2, 3, 4, 5):
        _report("c2 list, k = %d" % k, _workload(k))


def cmd_clusters():
    rng = np.random.default_rng(3)
    for n in (3, 4, 5, 6):
        for gap in (1e-1, 1e-3, 1e-6, 1e-9, 1e-12, 1e-15, 0.0):
            for mult in (2, 3):
                if mult >= n:
                    continue
                B = 20000
                ev = np.sort(rng.uniform(0.0, 1.0, size=(B, n)), axis=1)
                ev[:, 0] = rng.uniform(-0.5, 0.1, size=B)
                for j in range(1, mult):
                    ev[:, j] = ev[:, 0] + gap * j
                ev[:, -1] += 2.0
                Qm, _ = np.linalg.qr(rng.normal(size=(B, n, n)))
                A = np.einsum('bij,bj,bkj->bik', Qm, ev, Qm)
                A = 0.5 * (A + np.swapaxes(A, 1, 2))
                _report("n = %d, %d-fold cluster, gap %.0e" % (n, mult, gap), A)


def cmd_goldens():
    from oracle import cutsel_oracle as oracle
    g = np.load(os.path.join(ROOT, "tests", "golden", "inst_boxqp.npz"))
    for tag in ("spar020_100_1_d3", "spar020_100_1_d4", "spar040_030_1_d5", "spar030_060_1_d3"):
        S, ks, n = g[tag + "_set_inds"], g[tag + "_k"], int(g[tag + "_nb_vars"])
        L = n * (n + 1) // 2
        for pt in ("mck", "rnd", "psd"):
            vv = g["%s_%s_vars" % (tag, pt)]
            for k in np.unique(ks):
                si = S[ks == k][:, :k]
                _report("%s %s k=%d" % (tag, pt, k), lifted(vv[L:][si], vv[:L][oracle.triu_positions(si, n)], int(k)))


def cmd_replay(files):
    from sdpcutsel_via_nn_amd import _capi, harness
    from oracle import cutsel_oracle as oracle
    gold = os.path.join(ROOT, "tests", "golden")
    files = files or sorted(glob.glob(os.path.join(gold, "rounds_*.npz")))
    for path in files:
        g = np.load(path)
        name, dim, sel = str(g["name"]), int(g["dim"]), int(g["sel_size"])
        inst = harness.parse_boxqp(os.path.join(gold, "instances", name + ".in"))
        n, L = inst["nb_vars"], inst["nb_lifted"]
        S, ks, N = _capi.enumerate_cover(inst["adj"], dim)
        tot = dict(rounds=0, cls=0, pos_rounds=0, pos=0, maxd=0.0)
        for r in range(1, int(g["rounds_done"]) + 1):
            vv, strat = g["r%02d_vars" % r], int(g["r%02d_strat" % r])
            lam, ref = np.zeros(N), np.zeros(N)
            hard = 0
            for k in np.unique(ks):
                m = np.nonzero(ks == k)[0]
                A = lifted(vv[L:][S[m, :k]], vv[:L][oracle.triu_positions(S[m, :k], n)], int(k))
                ref[m] = np.linalg.eigvalsh(A)[:, 0]
                lk, ok = lambda_min(A)
                lam[m] = np.where(ok, lk, ref[m])
                hard += int((~ok).sum())
            cls = int(((lam < -1e-15) != (ref < -1e-15)).sum())
            # feasibility ranking (cut_select_qp.py:639-654): violated ones by -lambda descending, stable
            def head(l):
                v = np.nonzero(l < -1e-15)[0]
                return v[np.argsort(-l[v], kind="stable")][:sel]
            a, b = head(lam), head(ref)
            w = min(a.shape[0], b.shape[0])
            npos = int((a[:w] != b[:w]).sum()) + abs(a.shape[0] - b.shape[0])
            tot["rounds"] += 1
            tot["cls"] += cls
            tot["pos"] += npos
            tot["pos_rounds"] += npos > 0
            tot["maxd"] = max(tot["maxd"], float(np.abs(lam - ref).max()))
            print("  %s d%d round %2d (strategy %d): max|d| %.2e  classification differs %d  head positions differing %d of %d  left to Jacobi %d of %d"
                  % (name, dim, r, strat, np.abs(lam - ref).max(), cls, npos, w, hard, N), flush=True)
        print("%s d%d: %d rounds, classification differences %d, rounds with a differing head position %d (%d positions), max|d| %.2e"
              % (name, dim, tot["rounds"], tot["cls"], tot["pos_rounds"], tot["pos"], tot["maxd"]), flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
    {"synthetic": cmd_synthetic, "clusters": cmd_clusters, "goldens": cmd_goldens, "replay": lambda: cmd_replay(sys.argv[2:])}[cmd]()
