#!/usr/bin/env python3
"""Cost of a fused round when the combined strategy finds FEWER strong candidates than sel_size
(the scan visits every entry: full-sort path of rank.hip), config-2 sizes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from sdpcutsel_via_nn_amd import _capi, networks, synthetic  # noqa: E402

n, N = 100, 10 ** 6
wl = synthetic.make_workload(nb_vars=n, k=3, count=N, seed=7)
sc = _capi.Scorer(0)
sc.set_network(3, *networks.load_network(3))
sc.set_instance(n, wl["Q_arr"])
sc.set_candidates(wl["set_inds"], wl["ks"])
rng = np.random.default_rng(1)
x = rng.uniform(0.2, 0.8, n)
iu = np.triu_indices(n)
for noise in (0.0, 0.002, 0.01, 0.05):
    X = x[iu[0]] * x[iu[1]] + noise * rng.normal(size=iu[0].shape[0])     # near rank one: few violated sub-matrices
    X[iu[0] == iu[1]] = x ** 2 + 0.02
    vv = np.concatenate([X, x])
    sc.set_point(vv)
    r = sc.select_round(4, 5000)
    ts = []
    for it in range(20):
        sc.set_point(vv)
        t0 = time.perf_counter()
        r = sc.select_round(4, 5000, copy=False)
        ts.append(time.perf_counter() - t0)
    print("noise %.3f: violated %d positive %d strong %d -> new_strat %d, round %.3f ms" % (
        noise, r["counters"]["nb_violated"], r["counters"]["nb_positive"], r["counters"]["strong"], r["new_strat"],
        1e3 * np.median(ts)), flush=True)
sc.close()
