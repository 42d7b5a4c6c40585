#!/usr/bin/env python3
"""Kernel-time breakdown on the GPU box: score kernel per variant and flag set (hipEvents inside
the library), config-2 workload.  Usage: python tools/ablate.py [k] [count]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi, networks, synthetic  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 6
    nv = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    if os.environ.get("SDPCUT_LIB"):          # a variant built by tools/build_ablation.sh
        _capi.load_library(os.environ["SDPCUT_LIB"])
    wl = synthetic.make_workload(nb_vars=nv, k=k, count=count, seed=7)
    sc = _capi.Scorer(0)
    sc.set_option(_capi.OPT_TIMING, 1)
    sc.set_network(k, *networks.load_network(k))
    sc.set_instance(nv, wl["Q_arr"])
    sc.set_candidates(wl["set_inds"], wl["ks"])
    sc.set_point(wl["vars_values"])
    variants = (("mfma", _capi.KERNEL_MFMA), ("valu", _capi.KERNEL_VALU), ("simple", _capi.KERNEL_SIMPLE))
    if len(sys.argv) > 4:
        variants = [v for v in variants if v[0] in sys.argv[4].split(",")]
    flagsets = (("eig", _capi.EIG), ("nn", _capi.NN), ("eig+nn", _capi.EIG | _capi.NN))
    if len(sys.argv) > 5:
        flagsets = [f for f in flagsets if f[0] in sys.argv[5].split(",")]
    for name, kv in variants:
        sc.set_option(_capi.OPT_KERNEL, kv)
        for fname, flags in flagsets:
            ts = []
            reps = 8 if name == "simple" else max(12, min(200, int(2e8 / count)))   # the clocks need ~50 ms of load
            for it in range(reps):
                sc.score(flags)
                ts.append(sc.last_timing()[0])
            tail = ts[reps // 2:]
            med = 1e3 * np.median(tail)
            print("k=%d n=%d N=%d %-6s %-6s  median %.1f us  min %.1f us  -> %.3g candidates/s" % (
                k, nv, count, name, fname, med, 1e3 * min(tail), count / (med * 1e-6)), flush=True)
    sc.close()


if __name__ == "__main__":
    main()
