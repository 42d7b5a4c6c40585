#!/usr/bin/env python3
"""Time of the triangle-inequality separation step of a round (cut_select_qp.py:824-863 = _separate_and_add_triangle: device
separation + ranking of 4 T inequalities, rows handed to the LP) on spar125-075-1 at a recorded LP point: tools/tri_time.py [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import harness  # noqa: E402
from sdpcutsel_via_nn_amd.cut_solver import CutSolver  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = np.load(os.path.join(G, "rounds_spar125_075_1_d4_s4.npz"))
inst = harness.parse_boxqp(os.path.join(G, "instances", "spar125-075-1.in"))
n, L = inst["nb_vars"], inst["nb_lifted"]
cs = CutSolver()
cs._dim, cs._nb_vars, cs._nb_lifted, cs._Q_arr, cs._Q_adj = 4, n, L, inst["Q_arr"], inst["adj"]
cs._my_prob = harness.LinearRelaxation(np.zeros(L + n))
cs._preprocess_triangle_ineq()
for r in (1, 3, 8):
    vv = g["r%02d_vars" % r]
    for _ in range(3):
        cs._my_prob.linear_constraints = harness._RowStore()
        nb = cs._separate_and_add_triangle(0.1, vv)
    t0 = time.perf_counter()
    for _ in range(steps):
        cs._my_prob.linear_constraints = harness._RowStore()
        nb = cs._separate_and_add_triangle(0.1, vv)
    dt = (time.perf_counter() - t0) / steps
    print("spar125-075-1 (%d triples, %d inequalities), LP point of round %d: %d triangle cuts, %.3f ms per separation step"
          % (len(cs._gpu_tri_triples), 4 * len(cs._gpu_tri_triples), r, nb, dt * 1e3))
if len(sys.argv) > 2:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        cs._my_prob.linear_constraints = harness._RowStore()
        cs._separate_and_add_triangle(0.1, vv)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
