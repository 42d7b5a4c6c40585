#!/usr/bin/env python3
"""End-to-end rounds on the BoxQP fixtures (GPU box): bounds, gap closed, separation times."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "instances")
SOL = {"spar020-100-1": 706.5, "spar040-030-1": 839.5, "spar125-075-1": 12330.0}
for name, dim, strat, rounds in (("spar020-100-1", 3, 2, 4), ("spar020-100-1", 3, 1, 4), ("spar020-100-1", 3, 4, 4),
                                 ("spar040-030-1", 5, 4, 4), ("spar125-075-1", 3, 4, 2), ("spar125-075-1", 4, 4, 1)):
    cs = pkg.CutSolver()
    t = time.time()
    bounds, t_total, rt, st, cuts, _, nsub = cs.cut_select_algo(os.path.join(G, name + ".in"), dim, 0.1, strat=strat,
                                                                nb_rounds_cuts=rounds)
    gaps = [(bounds[0] - b) / (bounds[0] - SOL[name]) for b in bounds]
    print("%s dim %d strat %d: N=%d cuts=%s gap=%s sep_s=%s total %.1fs" % (
        name, dim, strat, nsub, cuts, ["%.4f" % g for g in gaps], ["%.4f" % s for s in st], time.time() - t), flush=True)
