#!/usr/bin/env python3
"""Fixed host-side cost of one round (GPU box): the same calls bench.py makes per step, on a list so small
that the device work is negligible.  usage: python tools/host_overhead.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi, synthetic

wl = synthetic.make_workload(nb_vars=100, k=3, count=256, seed=7)
sc = _capi.Scorer(0)
sc.set_builtin_networks(3)
sc.set_instance(100, wl["Q_arr"])
sc.set_candidates(wl["set_inds"], wl["ks"])
vv = wl["vars_values"]
for _ in range(200):
    sc.set_point(vv); sc.select_round(4, 100, copy=False)
n = 3000
t = time.perf_counter()
for _ in range(n):
    sc.set_point(vv)
t1 = time.perf_counter() - t
sc.synchronize()
t = time.perf_counter()
for _ in range(n):
    sc.set_point(vv); sc.select_round(4, 100, copy=False)
t2 = time.perf_counter() - t
import ctypes
lib, h = sc._lib, sc._h
blk, cap, n_out, n_tot, ns = ctypes.c_void_p(), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0)
cnt = np.zeros(4, dtype=np.int64)
pv = vv.ctypes.data_as(ctypes.POINTER(ctypes.c_double)); pc = cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
t = time.perf_counter()
for _ in range(n):
    lib.sdpcut_set_point(h, pv)
    lib.sdpcut_select_round_view(h, 4, 100, 9, ctypes.byref(blk), ctypes.byref(cap), ctypes.byref(n_out), ctypes.byref(n_tot), ctypes.byref(ns), pc)
t3 = time.perf_counter() - t
print("set_point alone %.1f us; set_point + select_round (256 candidates) %.1f us per round through the Scorer wrapper, %.1f us through bare ctypes calls"
      % (t1 / n * 1e6, t2 / n * 1e6, t3 / n * 1e6))
sc.close()
