#!/usr/bin/env python3
"""GPU box: what a fused round (sdpcut_round_csr: LP point in, ranked head and assembled cuts out) costs when the list is a handful of
candidates -- the floor of a round on this runtime (six dependent launches, one host synchronisation, a device that idles between
rounds), against which the short-list rounds of the paper's instances (profiles/r04_small_rounds_ab.txt) are to be read.
usage: tools/round_floor.py [reps] [list length ...]"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg  # noqa: E402
from sdpcutsel_via_nn_amd import synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
only = [int(a) for a in sys.argv[2:]]          # list lengths to run (default: all)
nv, k = 40, 3
Q_arr, vv, _ = synthetic.make_instance(nv, seed=7)
gc.collect()
gc.freeze()
for count, sel in ((64, 3), (1024, 51), (2048, 102), (3072, 153), (4096, 204), (6144, 307), (8192, 409), (12288, 614), (16384, 819), (65536, 3276)):
    if only and count not in only:
        continue
    wl = synthetic.make_workload(nb_vars=nv, k=k, count=count, seed=5)
    sc = pkg.Scorer(0)
    sc.set_builtin_networks(k)
    sc.set_instance(nv, Q_arr)
    sc.set_candidates(wl["set_inds"], wl["ks"])
    vvc = np.ascontiguousarray(wl["vars_values"])
    line = []
    for strat, tag in ((4, "combined"), (1, "feasibility"), (2, "optimality")):
        for _ in range(200):
            sc.round_csr(strat, sel, point=vvc)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            sc.round_csr(strat, sel, point=vvc)
            ts.append(time.perf_counter() - t0)
        ts = np.sort(np.array(ts)) * 1e6
        line.append("%s %.1f us (best decile %.1f)" % (tag, float(np.median(ts)), float(ts[: max(1, reps // 10)].mean())))
    # back to back without the host in between: what the device chain alone takes
    print("%6d three-variable candidates, head %4d: %s" % (count, sel, "; ".join(line)), flush=True)
    sc.close()
