#!/usr/bin/env python3
"""GPU box: the separation step (sdpcut_round_csr: LP point in, ranked head and assembled cuts out) of EVERY recorded round of a
reference trajectory (tests/golden/rounds_*.npz), median of `reps` back-to-back repeats per round, and the sum over the run.
usage: tools/trajectory_times.py rounds_spar125_075_1_d4_s4 [reps]          (SDPCUT_LIB=... for a variant build, e.g. -DSDPCUT_LMIN=0)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import sdpcutsel_via_nn_amd as pkg  # noqa: E402
from sdpcutsel_via_nn_amd import harness  # noqa: E402

gold = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(gold, sys.argv[1] + ".npz"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
name, dim, sel = str(g["name"]), int(g["dim"]), int(g["sel_size"])
inst = harness.parse_boxqp(os.path.join(gold, "instances", name + ".in"))
sc = pkg.Scorer(0)
sc.set_builtin_networks(dim)
sc.set_instance(inst["nb_vars"], inst["Q_arr"])
N = sc.set_candidates_cover(inst["adj"], dim)
import gc
gc.collect()
gc.freeze()
tot, line = 0.0, []
for r in range(1, int(g["rounds_done"]) + 1):
    vv, strat = np.ascontiguousarray(g["r%02d_vars" % r]), int(g["r%02d_strat" % r])
    for _ in range(5):
        sc.round_csr(strat, sel, point=vv)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        sc.round_csr(strat, sel, point=vv)
        ts.append(time.perf_counter() - t0)
    ms = 1e3 * float(np.median(ts))
    tot += ms
    line.append("%d:%s%.3f" % (r, "c" if strat == 4 else ("f" if strat == 1 else "o"), ms))
print("%s (%d candidates, head %d): %.2f ms for %d rounds | per round (c combined, f feasibility, o optimality) %s"
      % (sys.argv[1], N, sel, tot, len(line), " ".join(line)))
sc.close()
