import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg
from sdpcutsel_via_nn_amd import _capi, harness
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
g = np.load(os.path.join(G, "qcqp_rounds_q_20_20_100_2_s4.npz"))
inst = harness.parse_osil(os.path.join(G, "instances", "q_20_20_100_2.osil"))
n = inst["nb_vars"]
sc = pkg.Scorer(0)
sc.set_builtin_networks(3)
sc.set_instance(n, inst["Q_arr"])
sc.set_candidates(g["obj_set_inds"], g["obj_k"])
for r in (1, 2, 3, 6):
    vv = np.ascontiguousarray(g["r%02d_vars" % r])
    strat = int(g["r%02d_strat" % r])
    f0 = sc.get_stat(_capi.STAT_SELECT_FALLBACKS)
    res = sc.round_csr(strat, 57, point=vv)
    for _ in range(20):
        sc.round_csr(strat, 57, point=vv)
    t0 = time.perf_counter()
    for _ in range(200):
        sc.round_csr(strat, 57, point=vv)
    dt = (time.perf_counter() - t0) / 200
    print("round", r, "strategy", strat, "->", res["new_strat"], "counters", res["counters"], "n_total", res["n_total"], "head", res["idx"].shape[0],
          "fallbacks +%d" % (sc.get_stat(_capi.STAT_SELECT_FALLBACKS) - f0), "%.1f us per round" % (dt * 1e6))
sc.close()
