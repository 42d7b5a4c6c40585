#!/usr/bin/env python3
"""Run ONE score launch of a PHASETIME ablation build (tools/build_ablation.sh pt:PHASETIME): a few
waves print the cycles they spend per tile in each phase.  SDPCUT_LIB must point at the build."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi, networks, synthetic  # noqa: E402

_capi.load_library(os.environ["SDPCUT_LIB"])
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = synthetic.make_workload(nb_vars=100, k=k, count=10 ** 6, seed=7)
sc = _capi.Scorer(0)
sc.set_option(_capi.OPT_TIMING, 1)
sc.set_network(k, *networks.load_network(k))
sc.set_instance(100, wl["Q_arr"])
sc.set_candidates(wl["set_inds"], wl["ks"])
sc.set_point(wl["vars_values"])
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    sc.score(_capi.EIG | _capi.NN)
    print("launch", it, "kernel %.1f us" % (1e3 * sc.last_timing()[0]), flush=True)
sc.close()
