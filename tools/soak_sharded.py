#!/usr/bin/env python3
"""Soak run of the N > 1 round on the GPU box: four handles stand in for four ranks (ragged shards,
the records are concatenated instead of all-gathered); back-to-back rounds over changing points,
strategies and head lengths.  Every repeat must be bit-identical to the first result of its kind,
and the merged head must not depend on which stand-in rank computes it.
Usage: python tools/soak_sharded.py [seconds=120]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi, networks, synthetic  # noqa: E402
from sdpcutsel_via_nn_amd.distributed import DeviceOps  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    dev = torch.device("cuda", 0)
    sizes = [200000, 0, 7, 120000]
    bases = np.concatenate([[0], np.cumsum(sizes)])
    nv = 60
    wl = synthetic.make_workload(nb_vars=nv, k=3, count=int(bases[-1]), seed=23)
    rng = np.random.default_rng(5)
    points = [wl["vars_values"], np.clip(wl["vars_values"] + rng.normal(size=wl["vars_values"].shape) * 0.01, 0, 1)]
    X = np.full((nv, nv), 0.1)
    for v in range(3):
        X[v, :] = X[:, v] = 0.1 + 0.01 * (v + 1)
    points.append(np.concatenate([X[np.triu_indices(nv)], np.full(nv, 0.5)]))     # masses of equal eigenvalues
    opss = []
    for r, n in enumerate(sizes):
        sc = _capi.Scorer(0)
        sc.set_network(3, *networks.load_network(3))
        sc.set_instance(nv, wl["Q_arr"])
        lo = int(bases[r])
        sc.set_candidates(wl["set_inds"][lo:lo + n], wl["ks"][lo:lo + n], global_base=lo)
        opss.append(DeviceOps(sc, dev))
    first, rounds = {}, 0
    t_end = time.time() + budget
    while time.time() < t_end:
        p = int(rng.integers(0, len(points)))
        own = bool(rng.integers(0, 2))      # the heads score their shards themselves (nothing scored when they are asked)
        for ops in opss:
            ops.scorer.set_point(points[p])
            if not own:
                ops.scorer.score(_capi.EIG | _capi.NN)
        for _ in range(10):
            strat, code = [(1, 1), (2, 2), (4, _capi.PART_STRONG)][int(rng.integers(0, 3))]
            sel = int(rng.choice([1, 29, 777, 5000, 8192]))
            if own:
                for ops in opss:
                    ops.scorer.set_point(points[p])
            allrec = torch.cat([ops.shard_head(code, sel) for ops in opss])
            heads = []
            for r, ops in enumerate(opss):
                out = ops.shard_finish(len(sizes), sel, allrec, sel)
                got = {k: np.array(v, copy=True) for k, v in out.items() if isinstance(v, np.ndarray)}
                heads.append((got["idx"], got["score"]))
                key = (p, strat, sel, r, own)
                if key not in first:
                    first[key] = got
                    continue
                for k, v in got.items():
                    if not np.array_equal(v, first[key][k], equal_nan=True):
                        print("MISMATCH round %d kind %s field %s" % (rounds, key, k), flush=True)
                        sys.exit(1)
            for r in range(1, len(opss)):
                if not (np.array_equal(heads[r][0], heads[0][0]) and np.array_equal(heads[r][1], heads[0][1])):
                    print("stand-in ranks disagree on the merged head, round %d" % rounds, flush=True)
                    sys.exit(1)
            rounds += 1
        if rounds % 500 < 10:
            print("%d rounds, %d kinds, all identical so far" % (rounds, len(first)), flush=True)
    print("sharded soak ok: %d rounds x %d stand-in ranks over %d kinds (shards %s), every repeat bit-identical, "
          "all ranks agree on every merged head" % (rounds, len(sizes), len(first), sizes))
    for ops in opss:
        ops.scorer.close()


if __name__ == "__main__":
    main()
