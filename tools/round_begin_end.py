import gc, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import sdpcutsel_via_nn_amd as pkg
from sdpcutsel_via_nn_amd import synthetic
nv, k = 40, 3
Q_arr, vv, _ = synthetic.make_instance(nv, seed=7)
gc.collect(); gc.freeze()
for count, sel in ((1140, 57), (2048, 102)):
    wl = synthetic.make_workload(nb_vars=nv, k=k, count=count, seed=5)
    sc = pkg.Scorer(0)
    sc.set_builtin_networks(k); sc.set_instance(nv, Q_arr); sc.set_candidates(wl["set_inds"], wl["ks"])
    vvc = np.ascontiguousarray(wl["vars_values"])
    for strat in (4, 1):
        for idle_us in (0, 60):
            tb, te, tt = [], [], []
            for it in range(1500):
                t0 = time.perf_counter()
                sc.round_csr_begin(strat, sel, point=vvc)
                t1 = time.perf_counter()
                sc.round_csr_end()
                t2 = time.perf_counter()
                if it >= 200:
                    tb.append(t1 - t0); te.append(t2 - t1); tt.append(t2 - t0)
                while time.perf_counter() - t2 < idle_us * 1e-6:
                    pass
            print("n %d strat %d idle %2d us between rounds: begin %.1f us, end %.1f us, total %.1f us" % (count, strat, idle_us, 1e6*np.median(tb), 1e6*np.median(te), 1e6*np.median(tt)), flush=True)
    sc.close()
