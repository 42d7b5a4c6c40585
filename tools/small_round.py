import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from sdpcutsel_via_nn_amd import _capi, networks, synthetic
wl = synthetic.make_workload(nb_vars=100, k=3, count=6000, seed=7)
sc = _capi.Scorer(0)
sc.set_network(3, *networks.load_network(3))
sc.set_instance(100, wl["Q_arr"])
sc.set_candidates(wl["set_inds"], wl["ks"])
d_vars = torch.from_numpy(wl["vars_values"]).to("cuda:0")
for it in range(300):
    sc.set_point_device(d_vars.data_ptr()); sc.select_round(4, 5000, copy=False)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for it in range(n):
    sc.set_point_device(d_vars.data_ptr()); r = sc.select_round(4, 5000, copy=False)
t1 = time.perf_counter()
print("tiny round (6000 candidates): %.1f us per step" % ((t1 - t0) / n * 1e6))
t0 = time.perf_counter()
for it in range(n):
    sc.set_point_device(d_vars.data_ptr())
sc.synchronize()
t1 = time.perf_counter()
print("set_point_device alone: %.1f us" % ((t1 - t0) / n * 1e6))
