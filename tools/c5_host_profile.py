#!/usr/bin/env python3
"""cProfile of the QCQP round of bench.py's secondary.c5 (host side): tools/c5_host_profile.py [strat=1] [steps=300]"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from sdpcutsel_via_nn_amd import harness  # noqa: E402
from sdpcutsel_via_nn_amd.cut_solver import CutSolverQCQP, DeviceAgg  # noqa: E402

strat = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
gold = os.path.join(bench.ROOT, "tests", "golden")
inst = harness.parse_osil(os.path.join(gold, "instances", bench.C5_INSTANCE + ".osil"))
vv = np.ascontiguousarray(np.load(os.path.join(gold, "inst_qcqp50.npz"))["vars"], dtype=np.float64)
n = inst["nb_vars"]
cs = CutSolverQCQP(0)
cs._dim, cs._nb_vars, cs._nb_lifted, cs._Q_arr = bench.C5_DIM, n, inst["nb_lifted"], inst["Q_arr"]
cs._load_neural_nets()
sc_o, sc_c = cs._gpu_new_scorer(), cs._gpu_new_scorer()
for sc in (sc_o, sc_c):
    sc.set_instance(n, np.asarray(inst["Q_arr"], dtype=np.float64))
n_o, n_c = sc_o.set_candidates_cover_split(sc_c, inst["adj"], inst["adj_cons"], bench.C5_DIM)
cover_obj, cover_cons = DeviceAgg(sc_o, n_o, n, inst["Q_arr"]), DeviceAgg(sc_c, n_c, n, inst["Q_arr"])
cs._agg_list = cover_obj
cs._my_prob = harness.LinearRelaxation(np.zeros(inst["nb_lifted"] + n))


def one():
    cs._my_prob.linear_constraints = harness._RowStore()
    return cs.select_and_generate_round(strat, vv, 1, bench.SEL, cover_obj, cover_cons)


for _ in range(50):
    one()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    one()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print("strategy %d, %d rounds; times below are totals over all rounds (divide by %d)" % (strat, steps, steps))
st.print_stats(22)
