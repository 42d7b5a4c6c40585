import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sdpcutsel_via_nn_amd import _capi, networks, synthetic
for count, sel in ((1051, 105), (6000, 600), (20000, 2000)):
    wl = synthetic.make_workload(nb_vars=100, k=3, count=count, seed=7)
    sc = _capi.Scorer(0)
    sc.set_network(3, *networks.load_network(3))
    sc.set_instance(100, wl["Q_arr"])
    sc.set_candidates(wl["set_inds"], wl["ks"])
    for it in range(200):
        sc.set_point(wl["vars_values"]); sc.select_round(4, sel, copy=False)
    n = 1000
    t0 = time.perf_counter()
    for it in range(n):
        sc.set_point(wl["vars_values"]); r = sc.select_round(4, sel, copy=False)
    t1 = time.perf_counter()
    print("N=%d sel=%d: %.1f us per round (host point upload + fused round)" % (count, sel, (t1 - t0) / n * 1e6))
    sc.close()
