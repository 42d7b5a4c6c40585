#!/usr/bin/env python3
"""Time of the fused round at a recorded LP point of a trajectory golden, with the small size classes on side streams and without: tools/round_time.py tests/golden/rounds_X.npz [round=2] [steps=200]"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg  # noqa: E402
from sdpcutsel_via_nn_amd import harness  # noqa: E402

g = np.load(sys.argv[1])
r = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
name, dim, sel = str(g["name"]), int(g["dim"]), int(g["sel_size"])
inst = harness.parse_boxqp(os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "instances", name + ".in"))
gc.collect()
gc.freeze()
sc = pkg.Scorer(0)
sc.set_builtin_networks(dim)
sc.set_instance(inst["nb_vars"], inst["Q_arr"])
N = sc.set_candidates_cover(inst["adj"], dim)
S, ks = sc.get_candidates(np.arange(min(N, 2000000)))
p = "r%02d_" % r
vv, strat = np.ascontiguousarray(g[p + "vars"]), int(g[p + "strat"])
from sdpcutsel_via_nn_amd import _capi  # noqa: E402
out = []
for side in (2, 0):      # the library's default (one launch over all size classes) / a launch per class, one after the other
    sc.set_option(_capi.OPT_ONE_LAUNCH, 1 if side else 0)
    sc.set_option(_capi.OPT_SIDE_STREAMS, side)
    for _ in range(50):
        sc.round_csr(strat, sel, point=vv)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = sc.round_csr(strat, sel, point=vv)
    out.append((time.perf_counter() - t0) / steps * 1e6)
print("%s dim %d: %d candidates, sizes %s; round %d (strategy %d), sel %d: %.1f us per fused round (sdpcut_round_csr), %d cuts; "
      "size classes one launch after the other: %.1f us"
      % (name, dim, N, np.bincount(ks, minlength=6)[2:].tolist(), r, strat, sel, out[0], res["rhs"].shape[0], out[1]))
sc.close()
