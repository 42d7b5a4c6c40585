#!/usr/bin/env python3
"""cProfile of the separation part of CutSolver.cut_select_algo (everything but the LP solves): tools/sep_profile.py <name> <dim> <strat> <rounds>"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "instances")
name, dim, strat, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cs = pkg.CutSolver()
cs.cut_select_algo(os.path.join(G, name + ".in"), dim, 0.1, strat=strat, nb_rounds_cuts=2)
cs = pkg.CutSolver()
pr = cProfile.Profile()
pr.enable()
out = cs.cut_select_algo(os.path.join(G, name + ".in"), dim, 0.1, strat=strat, nb_rounds_cuts=rounds)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
for line in s.getvalue().splitlines():
    if "_solve_incremental" in line or "solve" in line and "harness" in line:
        continue
    print(line[:170])
print("separation per round (ms):", ["%.3f" % (1e3 * t) for t in out[3][1:]])
