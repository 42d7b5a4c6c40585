#!/usr/bin/env python3
"""cProfile of a whole CutSolver.cut_select_algo run (parser, cover, LP model, rounds): tools/algo_profile.py <name> <dim> <strat> <rounds> [tri]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "instances")
name, dim, strat, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tri = len(sys.argv) > 5
cs = pkg.CutSolver()
cs.cut_select_algo(os.path.join(G, name + ".in"), dim, 0.1, strat=strat, nb_rounds_cuts=1, triangle_on=tri)      # warm-up (library load)
cs = pkg.CutSolver()
pr = cProfile.Profile()
pr.enable()
out = cs.cut_select_algo(os.path.join(G, name + ".in"), dim, 0.1, strat=strat, nb_rounds_cuts=rounds, triangle_on=tri)
pr.disable()
print("%s dim %d strategy %d%s, %d rounds: total %.2f s" % (name, dim, strat, " + triangle" if tri else "", rounds, out[1]))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
