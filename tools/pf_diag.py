#!/usr/bin/env python3
"""Which rounds of the bench's workloads does the selection resolve from the fine histogram (SDPCUT_STAT_DIRECT_SELECTIONS)?
For every workload: strategy, direct yes/no, (bin of the k-th largest key, published floor, members at or above that bin).
usage: tools/pf_diag.py        (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from sdpcutsel_via_nn_amd import _capi, harness, networks, synthetic      # noqa: E402

SEL = 5000


def one(sc, strat, vv, tag):
    before = sc.get_stat(_capi.STAT_DIRECT_SELECTIONS)
    r = sc.select_round(strat, SEL, copy=False, point=vv)
    after = sc.get_stat(_capi.STAT_DIRECT_SELECTIONS)
    st = tuple(sc.get_stat(w) for w in (_capi.STAT_PF_BIN, _capi.STAT_PF_FLOOR, _capi.STAT_PF_COUNT))
    print("%-58s strategy %d -> %d  direct %d  bin / floor / members %s  counters %s" % (tag, strat, r["new_strat"], after - before, st, r["counters"]), flush=True)


def main():
    for k in (3, 2, 4, 5):
        wl = synthetic.make_workload(nb_vars=100, k=k, count=10 ** 6, seed=7 if k == 3 else 7 + k)
        Q_arr, vv, _ = synthetic.make_instance(100, seed=7)
        sc = _capi.Scorer(0)
        sc.set_network(k, *networks.load_network(k))
        sc.set_instance(100, Q_arr)
        sc.set_candidates(wl["set_inds"], wl["ks"])
        for strat in (4, 1, 2):
            one(sc, strat, vv, "c2 list, k = %d" % k)
        sc.close()
    gold = os.path.join(ROOT, "tests", "golden")
    for fn in sorted(os.listdir(gold)):
        if not (fn.startswith("rounds_") and fn.endswith(".npz")):
            continue
        g = np.load(os.path.join(gold, fn))
        name, dim = str(g["name"]), int(g["dim"])
        inst = harness.parse_boxqp(os.path.join(gold, "instances", name + ".in"))
        sc = _capi.Scorer(0)
        sc.set_builtin_networks(dim)
        sc.set_instance(inst["nb_vars"], inst["Q_arr"])
        n = sc.set_candidates_cover(inst["adj"], dim)
        if n < 32768:
            sc.close()
            continue
        for r in range(1, int(g["rounds_done"]) + 1):
            one(sc, int(g["r%02d_strat" % r]), np.ascontiguousarray(g["r%02d_vars" % r]), "%s dim %d (%d candidates) round %d" % (name, dim, n, r))
        sc.close()


if __name__ == "__main__":
    main()
