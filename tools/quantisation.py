#!/usr/bin/env python3
"""Score kernel time against the list length around 10^6 (is the time a staircase in strips per SIMD?): kernel time from the
event pair on the dispatch, combined round.  Usage: tools/quantisation.py [k]"""
import gc
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi, synthetic  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gc.collect()
gc.freeze()
wl = synthetic.make_workload(nb_vars=100, k=k, count=1114112, seed=7)
vv = wl["vars_values"]
sc = _capi.Scorer(0)
sc.set_builtin_networks(5)
sc.set_instance(100, wl["Q_arr"])
for strips_per_simd in (14.0, 14.5, 15.0, 15.26, 15.5, 16.0, 16.5, 17.0):
    count = 10 ** 6 if strips_per_simd == 15.26 else int(strips_per_simd * 1024 * 64)
    sc.set_candidates(wl["set_inds"][:count], wl["ks"][:count])
    sc.set_option(_capi.OPT_TIMING, 0)
    for _ in range(100):
        sc.select_round(4, 5000, copy=False, point=vv)
    sc.set_option(_capi.OPT_TIMING, 1)
    ms = []
    for _ in range(40):
        sc.select_round(4, 5000, copy=False, point=vv)
        ms.append(sc.last_timing()[0])
    med = 1e3 * np.median(ms)
    print("k=%d N=%7d = %5.2f strips per SIMD: kernel median %.1f us min %.1f us -> %.2f us per strip-per-SIMD"
          % (k, count, count / 65536.0, med, 1e3 * min(ms), med / (count / 65536.0)), flush=True)
sc.close()
