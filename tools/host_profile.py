#!/usr/bin/env python3
"""Host-side cost of the calls of one bench step (perf_counter around each), config-2 workload."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from sdpcutsel_via_nn_amd import _capi, networks, synthetic  # noqa: E402


def main():
    if os.environ.get("NOGC"):
        import gc
        gc.disable()
    wl = synthetic.make_workload(nb_vars=100, k=3, count=10 ** 6, seed=7)
    sc = _capi.Scorer(0)
    sc.set_option(_capi.OPT_TIMING, int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    sc.set_network(3, *networks.load_network(3))
    sc.set_instance(100, wl["Q_arr"])
    sc.set_candidates(wl["set_inds"], wl["ks"])
    d_vars = torch.from_numpy(wl["vars_values"]).to("cuda:0")
    acc = np.zeros(4)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    marks = []
    for it in range(n + 20):
        if it == 20:
            acc[:] = 0
            torch.cuda.synchronize()
            t_all = time.perf_counter()
        t0 = time.perf_counter()
        sc.set_point_device(d_vars.data_ptr())
        t1 = time.perf_counter()
        r = sc.select_round(4, 5000, copy=False)
        t2 = time.perf_counter()
        if sc_timing:
            sc.last_timing()
        t3 = time.perf_counter()
        acc += (t1 - t0, t2 - t1, t3 - t2, 0)
        if it >= 20 and (it - 20) % 100 == 99:
            marks.append(time.perf_counter())
    torch.cuda.synchronize()
    total = (time.perf_counter() - t_all) / n * 1e6
    print("us/step per block of 100:", " ".join("%.0f" % ((b - a) * 1e4) for a, b in zip(marks[:-1], marks[1:])))
    print("per step: total %.1f us | set_point_device %.1f | select_round %.1f | last_timing %.1f" % (
        total, acc[0] / n * 1e6, acc[1] / n * 1e6, acc[2] / n * 1e6))


sc_timing = (int(sys.argv[1]) if len(sys.argv) > 1 else 1) > 0
main()
