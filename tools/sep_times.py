#!/usr/bin/env python3
"""Separation time per round inside CutSolver.cut_select_algo (selection + generation [+ triangle]), round by round:
tools/sep_times.py <name> <dim> <strat> <rounds> [tri]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "instances")
name, dim, strat, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tri = len(sys.argv) > 5
cs = pkg.CutSolver()
out = cs.cut_select_algo(os.path.join(G, name + ".in"), dim, 0.1, strat=strat, nb_rounds_cuts=rounds, triangle_on=tri)
st = out[3]
print("%s dim %d strategy %d%s: separation per round (ms):" % (name, dim, strat, " + triangle" if tri else ""), ["%.3f" % (1e3 * t) for t in st[1:]])
