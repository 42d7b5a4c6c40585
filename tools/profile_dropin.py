#!/usr/bin/env python3
"""cProfile of the drop-in methods' round (selection + generation) on spar125-075-1, dim 3, at the
McCormick optimum: where the host time of the two-call shape goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg  # noqa: E402
from sdpcutsel_via_nn_amd import _capi, harness  # noqa: E402
from sdpcutsel_via_nn_amd.cut_solver import AggArrays  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "instances")
inst = harness.parse_boxqp(os.path.join(G, "spar125-075-1.in"))
n = inst["nb_vars"]
S, ks, N = _capi.enumerate_cover(inst["adj"], 3)
lp = harness.boxqp_relaxation(inst)
lp.solve()
vv = np.asarray(lp.get_values())
cs = pkg.CutSolver()
cs._sparse_pair = harness.SparsePair
cs.set_instance(n, inst["Q_arr"], AggArrays(S, ks, n, inst["Q_arr"]), dim=3, my_prob=lp)
sel = 5000


def one_round():
    new_strat, rl = cs._sel_eigcut_by_ordering_on_measure(4, vv, 1, sel_size=sel)
    return cs._gen_eigcuts_selected(4, sel, rl, vars_values=vv)


one_round()
t0 = time.perf_counter()
for _ in range(20):
    nb = one_round()
print("N = %d, %d cuts, %.2f ms per round" % (N, nb, (time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    one_round()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
