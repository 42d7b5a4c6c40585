#!/usr/bin/env python3
"""GPU box: a combined round whose threshold tie group overflows the sort buffers (every-entry-visited regime at a structured point:
x = 0.5, X = 0.1 -- ONE lifted matrix shared by all 3-variable candidates; the list keeps the candidates whose obj_improve is not
positive plus 60 positive ones) -- answered by topk_tie_split (r4) -- next to the same list at a generic point (no tie group)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg  # noqa: E402
from sdpcutsel_via_nn_amd import _capi, synthetic  # noqa: E402

n = 100
wl = synthetic.make_workload(nb_vars=n, k=3, count=300000, seed=7)
X = np.full((n, n), 0.1)
vv = np.concatenate([X[np.triu_indices(n)], np.full(n, 0.5)])
sc = pkg.Scorer(0)
sc.set_builtin_networks(3)
sc.set_instance(n, wl["Q_arr"])
sc.set_candidates(wl["set_inds"], wl["ks"])
sc.set_point(vv)
sc.score(_capi.EIG | _capi.NN)
eig0, obj0 = sc.get_scores()
keep = np.sort(np.concatenate([np.flatnonzero(obj0 <= 0), np.flatnonzero(obj0 > 0)[:60]]))
sc.set_candidates(wl["set_inds"][keep], wl["ks"][keep])
for name, point in (("structured point (tie group of the whole list)", vv), ("generic point", wl["vars_values"])):
    for sel in (5000, 16384):
        for _ in range(20):
            r = sc.round_csr(4, sel, point=point)
        s0, f0 = sc.get_stat(_capi.STAT_TIE_SPLITS), sc.get_stat(_capi.STAT_SELECT_FALLBACKS)
        t0 = time.perf_counter()
        for _ in range(100):
            r = sc.round_csr(4, sel, point=point)
        us = (time.perf_counter() - t0) / 100 * 1e6
        print("%d candidates, head %5d, %-46s: %7.1f us per round, %d of 100 by tie split, %d by the full-sort fallback"
              % (keep.size, sel, name, us, sc.get_stat(_capi.STAT_TIE_SPLITS) - s0, sc.get_stat(_capi.STAT_SELECT_FALLBACKS) - f0))
sc.close()
