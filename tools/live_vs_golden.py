#!/usr/bin/env python3
"""Live cutting-plane rounds (CutSolver.cut_select_algo: parser, device cover, HiGHS, GPU separation) against the bounds the
reference's own trajectory recorded round by round: tools/live_vs_golden.py tests/golden/rounds_X.npz [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg  # noqa: E402

g = np.load(sys.argv[1])
name, dim, strat = str(g["name"]), int(g["dim"]), int(g["strat0"])
rounds = min(int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(g["rounds_done"]))
inst = os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "instances", name + ".in")
cs = pkg.CutSolver()
t = time.time()
bounds, t_total, rt, st, cuts, tri, nsub = cs.cut_select_algo(inst, dim, 0.1, strat=strat, nb_rounds_cuts=rounds)
ref = g["bounds"][:rounds + 1]
b = np.array(bounds[:rounds + 1])
if np.sign(b[0]) != np.sign(ref[0]):      # (the fixtures keep the LP's own objective value, the loop returns the reference's sign)
    ref = -ref
rel = np.abs(b - ref) / np.maximum(1.0, np.abs(ref))
ref_cuts = [int(g["r%02d_nb_cuts" % r]) for r in range(1, rounds + 1)]
print("%s dim %d strategy %d: %d candidates, %d rounds in %.1f s (separation %.4f s in total)" % (name, dim, strat, nsub, rounds, time.time() - t, sum(st[1:])))
print("   bound after each round vs the reference's trajectory: max relative difference %.2e (round %d); last round %.2e" % (rel.max(), int(rel.argmax()), rel[-1]))
print("   cuts per round live      ", [int(c) for c in cuts[1:rounds + 1]])
print("   cuts per round reference ", ref_cuts)
closed, closed_ref = b[-1] - b[0], ref[-1] - ref[0]
print("   bound movement live %.6f reference %.6f (ratio %.6f)" % (closed, closed_ref, closed / closed_ref))
