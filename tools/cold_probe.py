#!/usr/bin/env python3
"""GPU box: what keeps a round after an idle period (an LP solve in the reference's loop) as fast as a back-to-back one?
Median host-to-host time of one combined and one feasibility round on the c2 list after 100 ms / 1 s of idle, plain, and with a
HEARTBEAT during the idle period: another thread launching an empty kernel on a second handle's stream every `period` ms.
usage: tools/cold_probe.py [repeats=12]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi, networks, synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
wl = synthetic.make_workload(nb_vars=100, k=3, count=10 ** 6, seed=7)
sc = _capi.Scorer(0)
sc.set_network(3, *networks.load_network(3))
sc.set_instance(100, wl["Q_arr"])
sc.set_candidates(wl["set_inds"], wl["ks"])
hb = _capi.Scorer(0)
hb.set_network(3, *networks.load_network(3))
hb.set_instance(100, wl["Q_arr"])
hb.set_candidates(wl["set_inds"][:200000], wl["ks"][:200000])      # a "burn" beat: a real scoring round of ~0.1 ms on the second handle
vv = wl["vars_values"]


class Beat(object):
    def __init__(self, period, sync):
        self.period, self.sync, self.stop = period, sync, False
        self.t = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop:
            if self.sync == "burn":
                hb.select_round(4, 100, copy=False, point=vv)
            elif self.sync == "same":      # on the stream the round will use
                sc.wake()
            else:
                hb.wake()
                if self.sync:
                    hb.synchronize()
            time.sleep(self.period)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.t.join()


def measure(strat, idle, beat):
    ts = []
    for _ in range(reps):
        if beat == "spin":      # the host does not sleep (an LP solve keeps its core busy), the device is idle
            t_e = time.perf_counter() + idle
            while time.perf_counter() < t_e:
                pass
        elif beat is None:
            time.sleep(idle)
        else:
            with Beat(*beat):
                time.sleep(idle)
        t0 = time.perf_counter()
        sc.select_round(strat, 5000, copy=False, point=vv)
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


for strat, name in ((4, "combined"), (1, "feasibility")):
    for _ in range(30):
        sc.select_round(strat, 5000, copy=False, point=vv)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        sc.select_round(strat, 5000, copy=False, point=vv)
        ts.append(time.perf_counter() - t0)
    print("%-12s back to back %.4f ms" % (name, 1e3 * float(np.median(ts))), flush=True)
    for idle in (0.1, 1.0):
        line = ["%-12s idle %4.0f ms: host sleeps %.4f" % (name, idle * 1e3, measure(strat, idle, None))]
        line.append("host spins %.4f" % measure(strat, idle, "spin"))
        for period, sync in ((0.002, False), (0.002, "same"), (0.02, "same"), (0.2, "same")):
            line.append("beat %g ms%s %.4f" % (period * 1e3, " " + sync if isinstance(sync, str) else "", measure(strat, idle, (period, sync))))
        print(" | ".join(line), flush=True)
sc.close()
hb.close()
