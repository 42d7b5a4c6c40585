#!/usr/bin/env python3
"""Score accuracy of the GPU kernels against the CPU oracle on the synthetic workload (GPU box):
max |d lambda_min| and the optimality score's error relative to max(|score|, 1e-3 max_elem), per k."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi, synthetic
from oracle import cutsel_oracle as oracle

n, L, N = 100, 5050, 200000
for k in (2, 3, 4, 5):
    wl = synthetic.make_workload(nb_vars=n, k=k, count=N, seed=7 + k)
    sc = _capi.Scorer(0)
    sc.set_builtin_networks(5)
    sc.set_instance(n, wl["Q_arr"])
    sc.set_candidates(wl["set_inds"], wl["ks"])
    sc.set_point(wl["vars_values"])
    sc.score(_capi.EIG | _capi.NN)
    eig, obj = sc.get_scores()
    si, vv = wl["set_inds"][:, :k], wl["vars_values"]
    ref_obj = oracle.opt_score_batch(k, si, n, vv, wl["Q_arr"])
    ref_eig = oracle.eigmin_batch(k, vv[L:][si], vv[:L][oracle.triu_positions(si, n)])
    me = k * np.abs(wl["Q_arr"][oracle.triu_positions(si, n)]).max(axis=1)
    me[me == 0] = 1.0
    rel = np.abs(obj - ref_obj) / np.maximum(np.abs(ref_obj), 1e-3 * me)
    print("k=%d: max |d eig| %.2e, obj_improve: max abs %.2e, max rel %.2e (median rel %.1e); parity bounds 2e-13 / 1e-9"
          % (k, np.abs(eig - ref_eig).max(), np.abs(obj - ref_obj).max(), rel.max(), np.median(rel)), flush=True)
    sc.close()
