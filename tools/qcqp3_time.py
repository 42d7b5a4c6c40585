#!/usr/bin/env python3
"""Time of a QCQP separation round with 3-variable sub-problems (the paper's QCQP runs, generate_figs_tables.py:266-272) at a
recorded LP point: CutSolverQCQP.select_and_generate_round on the two covers of a qcqp_rounds_* fixture.
tools/qcqp3_time.py tests/golden/qcqp_rounds_q_50_25_75_1_s4.npz [round=2] [steps=300]"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import harness  # noqa: E402
from sdpcutsel_via_nn_amd.cut_solver import AggArrays, CutSolverQCQP  # noqa: E402

g = np.load(sys.argv[1])
r = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
G = os.path.dirname(os.path.abspath(sys.argv[1]))
inst = harness.parse_osil(os.path.join(G, "instances", str(g["name"]) + ".osil"))
n = inst["nb_vars"]
cover_obj = AggArrays(g["obj_set_inds"], g["obj_k"], n, inst["Q_arr"])
cover_cons = AggArrays(g["cons_set_inds"], g["cons_k"], n, inst["Q_arr"])
cs = CutSolverQCQP()
cs.set_instance(n, inst["Q_arr"], cover_obj, dim=int(g["dim"]), my_prob=harness.LinearRelaxation(np.zeros(inst["nb_lifted"] + n)))
sel = int(g["sel_size"])
vv, strat = np.ascontiguousarray(g["r%02d_vars" % r]), int(g["r%02d_strat" % r])
gc.collect()
gc.freeze()
for overlap in (True, False):
    cs._gpu_overlap = overlap
    for b in getattr(cs, "_gpu_bindings", {}).values():
        b.drain()
        b.follower = b.leader = None

    def one():
        cs._my_prob.linear_constraints = harness._RowStore()
        return cs.select_and_generate_round(strat, vv, r, sel, cover_obj, cover_cons)
    for _ in range(50):
        out = one()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    dt = (time.perf_counter() - t0) / steps
    print("%s: objective cover %d, constraints-only cover %d candidates, sel %d, round %d (strategy %d): %.1f us per round, %d cuts%s"
          % (str(g["name"]), len(cover_obj), len(cover_cons), sel, r, strat, dt * 1e6, out[2], "" if overlap else "  [lists one after the other]"))
