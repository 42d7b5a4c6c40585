#!/usr/bin/env python3
"""Coefficients of the exp polynomial of the tansig in csrc/score.hip (mpmath, 60 digits).

exp(r) ~ 1 + r (1 + r (c2 + r (c3 + ... + r c_deg)))   on |r| <= ln2/16, then raised to the 8th power:
the two leading coefficients are pinned to 1 (their FMAs then need no constant register) and the rest is
the weighted minimax (Remez on the relative error) of q(r) = (exp(r) - 1 - r) / r^2.
usage: python tools/exp_poly.py [degree]"""
import sys
import mpmath as mp

mp.mp.dps = 60
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 6
R = mp.log(2) / 16
m = deg - 2                                   # degree of q


def q_exact(r):
    return (mp.exp(r) - 1 - r) / r ** 2 if r != 0 else mp.mpf(1) / 2


def fit(nodes):
    """solve for coefficients c[0..m] and level E: r^2 (q_poly - q_exact) / exp(r) = (-1)^i E at the nodes"""
    n = m + 2
    A = mp.matrix(n, n)
    b = mp.matrix(n, 1)
    for i, r in enumerate(nodes):
        w = r ** 2 / mp.exp(r)
        for j in range(m + 1):
            A[i, j] = w * r ** j
        A[i, m + 1] = (-1) ** i
        b[i] = w * q_exact(r)
    sol = mp.lu_solve(A, b)
    return [sol[j] for j in range(m + 1)], sol[m + 1]


def err(c, r):
    p = sum(cj * r ** j for j, cj in enumerate(c))
    return (1 + r + r ** 2 * p - mp.exp(r)) / mp.exp(r)


nodes = [R * mp.cos(mp.pi * (2 * i + 1) / (2 * (m + 2))) for i in range(m + 2)][::-1]
for it in range(30):
    c, E = fit(nodes)
    # new extrema of the error on a fine grid
    grid = [-R + 2 * R * mp.mpf(i) / 4000 for i in range(4001)]
    e = [err(c, r) for r in grid]
    ext = [i for i in range(1, 4000) if (e[i] - e[i - 1]) * (e[i + 1] - e[i]) <= 0 and abs(e[i]) > abs(E) / 4]
    cand = [0] + ext + [4000]
    # keep m + 2 alternating extrema with the largest magnitudes
    best = []
    for i in cand:
        if best and (e[i] > 0) == (e[best[-1]] > 0):
            if abs(e[i]) > abs(e[best[-1]]):
                best[-1] = i
        else:
            best.append(i)
    while len(best) > m + 2:
        best.pop(0 if abs(e[best[0]]) < abs(e[best[-1]]) else -1)
    new = [grid[i] for i in best]
    if len(new) != m + 2:
        break
    if max(abs(a - b) for a, b in zip(new, nodes)) < R * mp.mpf(10) ** -12:
        nodes = new
        break
    nodes = new
c, E = fit(nodes)
# round to double and report the error of the ROUNDED polynomial, of its 8th power, and of tansig
cd = [float(x) for x in c]
worst = max(abs(err([mp.mpf(x) for x in cd], -R + 2 * R * mp.mpf(i) / 20000)) for i in range(20001))
print("degree %d on |r| <= ln2/16: minimax relative error %.3e, with double coefficients %.3e; exp(y) = p^8: %.3e"
      % (deg, float(abs(E)), float(worst), 8 * float(worst)))
for j, x in enumerate(cd):
    print("  c%d = %s   (%.17g)" % (j + 2, float(x).hex(), x))
