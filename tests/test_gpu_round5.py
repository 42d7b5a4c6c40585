"""Round-5 GPU tests (run with -m gpu on an MI355X).

Selection parity against ORACLE-SCORED rankings at full size: every other full-size test ranks the
device's own scores with the oracle (index work bit-exact given the scores); here the oracle scores the whole list itself
(C restatement of NNs.so + LAPACK eigvalsh, a process pool over the box's cores) and its ranking
(cut_select_qp.py:601-632, :639-654) is compared position by position with what the GPU round returns.  The comparison
code is bench.py's own (`selection_parity`), so the `parity` object of the bench line is what is tested here.
"""
import multiprocessing as mp

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EIG_ATOL = 2e-13      # as tests/test_gpu_parity.py
OBJ_RTOL = 1e-9


def oracle_scores(oracle, k, nb_vars, set_inds, vv, Q):
    """(obj_improve, lambda_min) of every candidate by the oracle, split over a process pool (bench.py's worker)."""
    import bench
    cores = min(bench.host_cores()[0], 16)
    n = set_inds.shape[0]
    si = np.ascontiguousarray(set_inds[:, :k])
    chunks = np.array_split(np.arange(n), max(1, cores * 4))
    if cores == 1 or n < 200000:
        parts = [bench._cpu_score_chunk((k, nb_vars, si[c], vv, Q)) for c in chunks]
    else:
        with mp.get_context("spawn").Pool(cores) as pool:
            parts = pool.map(bench._cpu_score_chunk, [(k, nb_vars, si[c], vv, Q) for c in chunks])
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def check_rounds(oracle, sc, wl, k, sel, base=0):
    import bench
    n, vv, Q = wl["nb_vars"], wl["vars_values"], wl["Q_arr"]
    ref_obj, ref_eig = oracle_scores(oracle, k, n, wl["set_inds"], vv, Q)
    q = np.abs(np.asarray(Q)[oracle.triu_positions(wl["set_inds"][:, :k], n)]).max(axis=1) * k
    max_elem = np.where(q == 0, 1.0, q)
    out = {}
    for strat in (4, 2, 1):
        res = sc.select_round(strat, sel, copy=True, point=vv)
        rec = bench.gpu_round_record(sc, res, strat)
        rec["idx"] = rec["idx"] - base
        par = bench.selection_parity(oracle, strat, None if strat == 1 else ref_obj, None if strat == 2 else ref_eig, max_elem, sel, rec)
        out[strat] = par
        assert par["topk_identical"], par
        assert par["positions_differing"] == 0 and par["ids_one_side_only"] == 0
        assert par["new_strategy_identical"]
        assert par["head"] == par["gpu_head"] == sel
        if strat != 2:
            assert par["max_abs_d_eig"] <= EIG_ATOL, par
        if strat != 1:
            assert par["max_rel_d_obj"] <= OBJ_RTOL, par
        # the scores the round hands back are the oracle's new scores (obj_improve +- BIG_M, -lambda_min) within the same bounds
        assert par["max_abs_d_head_score"] <= (1e-7 if strat != 1 else EIG_ATOL), par
        # ... and the cuts' lambda_min (epilogue) is the oracle's eigenvalue of the same candidates
        assert np.abs(res["lam"] - ref_eig[rec["idx"]]).max() <= EIG_ATOL
    return out


def test_headline_selection_equals_oracle_scored_ranking(oracle):
    """BASELINE configs[1] (the bench list: n = 100, 1e6 random 3-variable sets, seed 7, sel_size 5000): the top-5000 of the
    fused GPU round == the top-5000 of the oracle's ranking of ITS OWN scores, for the combined, optimality and
    feasibility strategies; score deviations over all 1e6 candidates inside the stated tolerances."""
    import sdpcutsel_via_nn_amd as pkg
    from sdpcutsel_via_nn_amd import networks, synthetic
    wl = synthetic.make_workload(nb_vars=100, k=3, count=10 ** 6, seed=7)
    sc = pkg.Scorer(0)
    try:
        sc.set_network(3, *networks.load_network(3))
        sc.set_instance(100, wl["Q_arr"])
        sc.set_candidates(wl["set_inds"], wl["ks"])
        check_rounds(oracle, sc, wl, 3, 5000)
    finally:
        sc.close()


def test_config4_shard_selection_equals_oracle_scored_ranking(oracle):
    """BASELINE configs[3], the share of one of 8 GPUs (n = 1000, 1.25e7 candidates, global ids of rank 5): the same
    comparison -- the oracle scores all 1.25e7 candidates itself."""
    import sdpcutsel_via_nn_amd as pkg
    from sdpcutsel_via_nn_amd import networks, synthetic
    n, N, base = 1000, 12_500_000, 5 * 12_500_000
    wl = synthetic.make_workload(nb_vars=n, k=3, count=N, seed=12)
    sc = pkg.Scorer(0)
    try:
        sc.set_network(3, *networks.load_network(3))
        sc.set_instance(n, wl["Q_arr"])
        sc.set_candidates(wl["set_inds"], wl["ks"], global_base=base)
        check_rounds(oracle, sc, wl, 3, 5000, base=base)
    finally:
        sc.close()


# --------------------------------------------------------------------------- streaming prefilter / direct selection
def _round_ab(sc, strat, sel, vv):
    """the same fused round with the fine histogram on and off -> (result on, result off, direct selections taken)"""
    from sdpcutsel_via_nn_amd import _capi
    out = []
    taken = 0
    for on in (1, 0):
        sc.set_option(_capi.OPT_PREFILTER, on)
        before = sc.get_stat(_capi.STAT_DIRECT_SELECTIONS)
        out.append(sc.select_round(strat, sel, copy=True, point=vv))
        after = sc.get_stat(_capi.STAT_DIRECT_SELECTIONS)
        if on:
            taken = after - before
            _round_ab.last = tuple(sc.get_stat(w) for w in (_capi.STAT_PF_BIN, _capi.STAT_PF_FLOOR, _capi.STAT_PF_COUNT))
        else:
            assert after == before
    sc.set_option(_capi.OPT_PREFILTER, 1)
    return out[0], out[1], taken


def _same_round(a, b):
    for key in ("idx", "score", "lam", "coef", "rhs", "ks"):
        assert np.array_equal(a[key], b[key]), key
    assert a["n_total"] == b["n_total"] and a["new_strat"] == b["new_strat"] and a["counters"] == b["counters"]


@pytest.mark.parametrize("count,sel", [(10 ** 6, 5000), (300000, 5000), (60000, 5000), (10 ** 6, 100), (10 ** 6, 8192), (2 * 10 ** 6, 16000)])
def test_direct_selection_equals_the_radix_passes(count, sel):
    """The selection resolved from the fine histogram (no digit pass, no grid barrier) returns bit for bit what the radix passes
    return -- ids, scores, eigenvalues, rows, counters -- for the three strategies, and it IS taken on generic points."""
    import sdpcutsel_via_nn_amd as pkg
    from sdpcutsel_via_nn_amd import _capi, networks, synthetic
    wl = synthetic.make_workload(nb_vars=100, k=3, count=count, seed=21)
    sc = pkg.Scorer(0)
    try:
        sc.set_network(3, *networks.load_network(3))
        sc.set_instance(100, wl["Q_arr"])
        sc.set_candidates(wl["set_inds"], wl["ks"])
        for strat in (4, 1, 2):
            if strat == 4 and sel > 8192:
                continue                      # (the combined strategy's device-resolved regime takes heads <= 8192)
            a, b, taken = _round_ab(sc, strat, sel, wl["vars_values"])
            _same_round(a, b)
            assert a["idx"].shape[0] == sel
            print("count", count, "sel", sel, "strategy", strat, "direct", taken, "fine bin / floor / members", _round_ab.last, a["counters"])
            # (a head that is a large share of a short list can have its threshold in a fat bin or below the floor, a head of
            # exactly the sort buffers' size cannot have a superset that fits: the passes run)
            if count >= 10 ** 6 and sel <= 5000:
                assert taken == 1, (strat, taken, _round_ab.last)
        assert sc.get_stat(_capi.STAT_SELECT_FALLBACKS) == 0
    finally:
        sc.close()


def test_direct_selection_steps_aside_for_masses_of_equal_keys(golden_boxqp):
    """A structured LP vertex (x = 0.5, X in {0, 0.5}: a handful of distinct eigenvalues over 1e6 candidates): the bin of the
    threshold holds far more members than the sort buffers -- the fine histogram says so and the radix passes run (ties cut by
    index as ever).  Same results with the option off."""
    import sdpcutsel_via_nn_amd as pkg
    from sdpcutsel_via_nn_amd import _capi, networks, synthetic
    wl = synthetic.make_workload(nb_vars=60, k=3, count=400000, seed=5)
    n = 60
    L = n * (n + 1) // 2
    vv = np.zeros(L + n)
    vv[L:] = 0.5
    rng = np.random.default_rng(1)
    X = np.where(rng.random(L) < 0.5, 0.5, 0.0)
    iu = np.triu_indices(n)
    X[iu[0] == iu[1]] = 0.5
    vv[:L] = X
    sc = pkg.Scorer(0)
    try:
        sc.set_network(3, *networks.load_network(3))
        sc.set_instance(n, wl["Q_arr"])
        sc.set_candidates(wl["set_inds"], wl["ks"])
        for strat in (1, 4, 2):
            a, b, taken = _round_ab(sc, strat, 5000, vv)
            _same_round(a, b)
            if strat == 1:
                assert taken == 0 and np.unique(a["score"]).size < 50      # masses of equal eigenvalues: the passes ran
    finally:
        sc.close()


def test_direct_selection_on_a_real_cover(oracle):
    """spar125-075-1 dim 4 (1 700 215 four-variable candidates, enumerated index set by index set: the head clusters) at recorded
    LP points: the feasibility round 8 is resolved from the fine histogram; the combined rounds 4 and 2 are not -- the score
    kernel of 4-variable candidates does not count (csrc/score.hip: it costs that kernel more than the selection saves).  All
    identical to the radix passes."""
    import os
    import sdpcutsel_via_nn_amd as pkg
    from conftest import GOLDEN
    from sdpcutsel_via_nn_amd import harness
    inst = harness.parse_boxqp(os.path.join(GOLDEN, "instances", "spar125-075-1.in"))
    g = np.load(os.path.join(GOLDEN, "rounds_spar125_075_1_d4_s4.npz"))
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(4)
        sc.set_instance(inst["nb_vars"], inst["Q_arr"])
        assert sc.set_candidates_cover(inst["adj"], 4) == 1700215
        for r in (4, 8, 2):
            strat = int(g["r%02d_strat" % r])
            a, b, taken = _round_ab(sc, strat, 5000, np.ascontiguousarray(g["r%02d_vars" % r]))
            _same_round(a, b)
            assert np.array_equal(a["idx"], g["r%02d_ids" % r].astype(np.int64)) or r == 2
            print("round", r, "strategy", strat, "direct", taken, "fine bin / floor / members", _round_ab.last)
            assert taken == (1 if r == 8 else 0), (r, strat, taken, _round_ab.last)
    finally:
        sc.close()
