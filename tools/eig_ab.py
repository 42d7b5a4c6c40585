#!/usr/bin/env python3
"""Eigenvalue-only kernel inside a feasibility round, with / without the selection's leading-digit histogram
(SDPCUT_OPT_FUSE_KEYS) and through the scoring kernels' eigenvalue branch (SDPCUT_OPT_EIG_KERNEL = 0): kernel time from
the event pair on the dispatch, round time host to host.  Usage: tools/eig_ab.py [k] [count]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi, synthetic  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 6
import gc
gc.collect()
gc.freeze()      # a generation-2 collection of what `import torch` leaves behind stalls one of the timed loops for ~75 ms
wl = synthetic.make_workload(nb_vars=100, k=k, count=count, seed=7)
for name, fuse, eigk in (("eig kernel + histogram", 1, 1), ("eig kernel, selection runs its own key pass", 0, 1),
                         ("scoring kernel's eigenvalue branch + histogram", 1, 0)):
    sc = _capi.Scorer(0)
    sc.set_builtin_networks(5)
    sc.set_option(_capi.OPT_FUSE_KEYS, fuse)
    sc.set_option(_capi.OPT_EIG_KERNEL, eigk)
    sc.set_instance(100, wl["Q_arr"])
    sc.set_candidates(wl["set_inds"], wl["ks"])
    vv = wl["vars_values"]
    for _ in range(200):
        sc.select_round(1, 5000, copy=False, point=vv)
    t0 = time.perf_counter()
    for _ in range(200):
        sc.select_round(1, 5000, copy=False, point=vv)
    dt = (time.perf_counter() - t0) / 200
    sc.set_option(_capi.OPT_TIMING, 1)
    ms = []
    for _ in range(40):
        sc.select_round(1, 5000, copy=False, point=vv)
        ms.append(sc.last_timing()[0])
    print("k=%d N=%d %-52s kernel median %.1f us min %.1f us | round %.1f us" % (k, count, name, 1e3 * np.median(ms), 1e3 * min(ms), dt * 1e6))
    sc.close()
