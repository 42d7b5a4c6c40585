#!/usr/bin/env python3
"""Soak run on the GPU box: thousands of back-to-back fused rounds over changing points (random
LP-like points and structured points whose masses of equal eigenvalues drive the radix select
through all eight digits inside the grid barrier), strategies, head lengths and list sizes.
Every result must be bit-identical to the first one of its kind -- and (r5) the first one of a kind on the main and on the
mixed-size handle is computed with SDPCUT_OPT_PREFILTER off: every later round, resolved from the fine histogram wherever that
applies, is thereby compared with what the radix passes return.
Usage: python tools/soak.py [seconds=120] [count=1000000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdpcutsel_via_nn_amd import _capi, networks, synthetic  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 6
    n = 100
    wl = synthetic.make_workload(nb_vars=n, k=3, count=count, seed=7)
    sc = _capi.Scorer(0)
    sc.set_network(3, *networks.load_network(3))
    sc.set_instance(n, wl["Q_arr"])
    rng = np.random.default_rng(11)
    points = [wl["vars_values"], np.clip(wl["vars_values"] + rng.normal(size=wl["vars_values"].shape) * 0.01, 0, 1)]
    for distinct in (0, 3, 12):            # structured: a handful of eigenvalues shared by 1e4..1e6 candidates
        X = np.full((n, n), 0.1)
        for v in range(distinct):
            X[v, :] = X[:, v] = 0.1 + 0.01 * (v + 1)
        points.append(np.concatenate([X[np.triu_indices(n)], np.full(n, 0.5)]))
    sizes = [count, max(count // 7, 1), 8192, 5000, 300]
    # (r3) a second handle whose rounds are begun together with the first one's (sdpcut_round_csr_begin / _end): two streams at once
    sc2 = _capi.Scorer(0)
    sc2.set_network(3, *networks.load_network(3))
    sc2.set_instance(n, wl["Q_arr"])
    sc2.set_candidates(wl["set_inds"][count // 3:count // 3 + min(count // 2, 300000)], wl["ks"][count // 3:count // 3 + min(count // 2, 300000)])
    pairs = 0
    # (r3) a third handle with a list of several size classes (one launch for all of them, score_mfma_all_kernel)
    rng3 = np.random.default_rng(5)
    ks3 = rng3.choice(np.array([5, 5, 5, 5, 5, 5, 4, 3, 2], dtype=np.int32), size=70000)
    S3 = np.full((70000, 5), -1, dtype=np.int32)
    for kk in (2, 3, 4, 5):
        m = np.flatnonzero(ks3 == kk)
        S3[m, :kk] = synthetic.random_index_sets(n, kk, m.size, rng3)
    sc3 = _capi.Scorer(0)
    sc3.set_builtin_networks(5)
    sc3.set_instance(n, wl["Q_arr"])
    sc3.set_candidates(S3, ks3)
    mixed = 0
    first, rounds, per_kind = {}, 0, {}
    t_end = time.time() + budget
    cur_size = None
    while time.time() < t_end:
        size = int(rng.choice(sizes))
        if size != cur_size:
            sc.set_candidates(wl["set_inds"][:size], wl["ks"][:size])
            cur_size = size
        for _ in range(40):
            p = int(rng.integers(0, len(points)))
            strat = int(rng.choice([1, 2, 4]))
            sel = int(rng.choice([1, 64, 777, 5000]))
            csr = bool(rng.integers(0, 2))      # (r3) either epilogue: padded rows or the CSR block assembled on the device
            if csr and rng.integers(0, 3) == 0:
                p2, strat2, sel2 = int(rng.integers(0, len(points))), int(rng.choice([1, 2, 4])), int(rng.choice([64, 5000]))
                sc.round_csr_begin(strat, sel, point=points[p])
                sc2.round_csr_begin(strat2, sel2, point=points[p2])
                r = sc.round_csr_end()
                r2 = sc2.round_csr_end()
                key2 = ("second handle", p2, strat2, sel2)
                got2 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r2.items()}
                pairs += 1
                if key2 not in first:
                    first[key2] = got2
                else:
                    for k, v in got2.items():
                        same = np.array_equal(v, first[key2][k], equal_nan=True) if isinstance(v, np.ndarray) else v == first[key2][k]
                        if not same:
                            print("MISMATCH round %d kind %s field %s" % (rounds, key2, k), flush=True)
                            sys.exit(1)
            elif csr:
                r = sc.round_csr(strat, sel, point=points[p])
            else:
                sc.set_point(points[p])
                r = sc.select_round(strat, sel, copy=False)
            if rng.integers(0, 8) == 0:      # a round on the mixed list in between
                p3, strat3, sel3 = int(rng.integers(0, len(points))), int(rng.choice([1, 2, 4])), int(rng.choice([64, 5000]))
                r3 = sc3.round_csr(strat3, sel3, point=points[p3])
                key3 = ("mixed list", p3, strat3, sel3)
                got3 = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r3.items()}
                mixed += 1
                if key3 not in first:
                    sc3.set_option(_capi.OPT_PREFILTER, 0)
                    r3 = sc3.round_csr(strat3, sel3, point=points[p3])
                    sc3.set_option(_capi.OPT_PREFILTER, 1)
                    first[key3] = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r3.items()}
                for k, v in got3.items():
                    same = np.array_equal(v, first[key3][k], equal_nan=True) if isinstance(v, np.ndarray) else v == first[key3][k]
                    if not same:
                        print("MISMATCH round %d kind %s field %s" % (rounds, key3, k), flush=True)
                        sys.exit(1)
            key = (size, p, strat, sel, csr)
            got = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r.items()}
            rounds += 1
            per_kind[key] = per_kind.get(key, 0) + 1
            if key not in first:      # the reference of this kind: the same round through the radix passes
                sc.set_option(_capi.OPT_PREFILTER, 0)
                if csr:
                    rr = sc.round_csr(strat, sel, point=points[p])
                else:
                    sc.set_point(points[p])
                    rr = sc.select_round(strat, sel, copy=False)
                sc.set_option(_capi.OPT_PREFILTER, 1)
                first[key] = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in rr.items()}
            ref = first[key]
            for k, v in got.items():
                same = np.array_equal(v, ref[k], equal_nan=True) if isinstance(v, np.ndarray) else v == ref[k]
                if not same:
                    print("MISMATCH round %d kind %s field %s" % (rounds, key, k), flush=True)
                    sys.exit(1)
        if rounds % 2000 < 40:
            print("%d rounds, %d kinds, all identical so far" % (rounds, len(first)), flush=True)
    fallbacks = sc.get_stat(_capi.STAT_SELECT_FALLBACKS) + sc2.get_stat(_capi.STAT_SELECT_FALLBACKS) + sc3.get_stat(_capi.STAT_SELECT_FALLBACKS)
    splits = [s_.get_stat(_capi.STAT_TIE_SPLITS) for s_ in (sc, sc2, sc3)]
    direct = [s_.get_stat(_capi.STAT_DIRECT_SELECTIONS) for s_ in (sc, sc2, sc3)]
    print("soak ok: %d rounds over %d kinds (sizes %s, %d points, strategies 1/2/4, both epilogues; %d of them begun together with a round "
          "on a second handle and ended after it; + %d rounds on a list of mixed sizes 2..5), every repeat bit-identical; %d rounds answered by the "
          "full-sort path (SDPCUT_STAT_SELECT_FALLBACKS); %d / %d / %d rounds (first / second handle / mixed list) whose threshold tie group was "
          "cut by its secondary key (SDPCUT_STAT_TIE_SPLITS); %d / %d / %d selections resolved from the fine histogram without a digit "
          "pass (SDPCUT_STAT_DIRECT_SELECTIONS), each compared with the radix passes' result of its kind"
          % (rounds, len(first), sizes, len(points), pairs, mixed, fallbacks, splits[0], splits[1], splits[2], direct[0], direct[1], direct[2]))
    sc.close()
    sc2.close()
    sc3.close()


if __name__ == "__main__":
    main()
