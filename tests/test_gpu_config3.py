"""BASELINE.json configs[2] at its stated size (top item): spar125-075-*, mixed-size
covers of dimension 4 (1.7e6 candidates) and 5 (12.8e6, past the reference's 4e6 guard), the
combined strategy with the top-10 % / 5000 cap.

  * replay of a 20-round trajectory of the REAL reference (tests/golden/rounds_*.npz, produced by
    tests/golden/make_rounds_golden.py: the reference's own selection + generation driven round after
    round): at every recorded LP point the library must select what the reference selected;
  * live rounds through CutSolver.cut_select_algo (device-side cover, HiGHS);
  * one dim-5 round with the 12.8e6-candidate cover, through size-independent properties."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

PUBLISHED_COUNTS = {                     # data_tables/data_(M+S^E_{3,4,5})_opt_0.1_40.csv:99-101, column nb_subproblems
    "spar125-075-1": {3: 133242, 4: 1700215, 5: 12845805},
    "spar125-075-2": {3: 132145, 4: 1681784, 5: 12739984},
    "spar125-075-3": {3: 129267, 4: 1602598, 5: 11660122},
}


def _instance(name):
    from sdpcutsel_via_nn_amd import harness
    return harness.parse_boxqp(os.path.join(GOLDEN, "instances", name + ".in"))


@pytest.mark.parametrize("name", sorted(PUBLISHED_COUNTS))
@pytest.mark.parametrize("dim", [3, 4, 5])
def test_published_candidate_counts_from_the_device_enumeration(name, dim):
    import sdpcutsel_via_nn_amd as pkg
    inst = _instance(name)
    sc = pkg.Scorer(0)
    try:
        sc.set_instance(inst["nb_vars"], inst["Q_arr"])
        assert sc.set_candidates_cover(inst["adj"], dim) == PUBLISHED_COUNTS[name][dim]
    finally:
        sc.close()


def _trajectories():
    return sorted(glob.glob(os.path.join(GOLDEN, "rounds_*.npz")))


# ---------------------------------------------------------------------------------------------------------------------------
# Where the replay may differ from the reference's recorded ORDER, and what the test proves about it AT RUN TIME (VERDICT r4
# item 2).  199 recorded rounds; everywhere else the head must be the reference's, position by position.
#
# A pair of candidates (a, b) is INVERTED when the reference's list and ours put them in opposite orders (a candidate that only
# one side selected counts as ranked behind the other side's whole head).  For every inverted pair the test computes
#   * the TRUE difference of the two scores -- feasibility measure: the EXACT smallest eigenvalues (tests/exact_eig.py: integer
#     characteristic polynomial, 80-digit root); optimality measure: the reference's own obj_improve (oracle = NNs.so bit for bit);
#   * the measured error of each side on each of the two candidates -- |reference's recorded score - true|, |device score - true|
#     (a candidate outside the reference's recorded head has no recorded score: this box's LAPACK stands in for it);
# and asserts  |true(a) - true(b)| <= err_ref(a) + err_ref(b) + err_dev(a) + err_dev(b),  every single error inside the accuracy
# the parity tests hold everywhere (eigenvalues 2e-15 -- LAPACK's own distance from the exact value reaches 8e-16 on these
# matrices --, obj_improve 1e-9 relative).  I.e. the two solvers' measured errors ON THESE TWO MATRICES are what inverts the
# pair, not an ordering rule.  At generic LP points the inverted pairs are additionally PINNED by id: a third pair fails.
ADMITTED_PAIRS = {
    # (trajectory, round): inverted pairs (ids in the reference's order)
    # permuted copies of ONE matrix: exact eigenvalues EQUAL, the reference's LAPACK puts them 2.8e-16 apart
    ("rounds_spar125_075_1_d4_s4", 5): [(806014, 803286)],
    # obj_improve + 1000 (cut_select_qp.py:611): the reference's own obj_improve of the two members differ by 3.4e-13 / by nothing,
    # the device's MLP (MFMA summation order) is 1e-13 .. 1e-12 from NNs.so on them -- 1e-15 of the score, BASELINE grants 1e-6
    ("rounds_spar125_075_2_d3_s4", 2): [(41980, 110560), (98522, 87399)],
    # exact eigenvalues 2.9e-17 apart; the reference's LAPACK is 1.7e-16 and 2.0e-16 from them, csrc/lmin.h 0.7e-16 and 2.2e-16
    ("rounds_spar125_075_2_d3_s4", 18): [(3688, 42018)],
    # third LP of the pure-feasibility run: six pairs of EXACTLY equal eigenvalues (permuted copies), 1e-16 apart in LAPACK
    ("rounds_spar080_075_1_d4_s1", 3): [(51053, 57454), (80443, 223334), (185697, 126667), (207046, 206215), (223365, 80472),
                                        (234718, 222031)],
}
# the McCormick vertex and the LP after it under PURE feasibility (x = 0.5, X in {0, 0.5}: 2 and 245 distinct eigenvalues in a
# head of 5000): exact tie groups whose members LAPACK's rounding orders and, where a group straddles position 5000, selects --
# 2.8e6 / 1.6e6 inverted pairs, too many to name; every one of them is still proven as above: round 1 true difference 0 for all,
# round 2 at most 2.4e-16 (measured budgets 5.6e-16 / 9.9e-16).  DESIGN.md section 2, stated deviation 1.
STRUCTURED_ROUNDS = {("rounds_spar080_075_1_d4_s1", 1), ("rounds_spar080_075_1_d4_s1", 2)}
EIG_ERR_MAX = 2e-15
OBJ_ERR_REL = 1e-9


def _append_report(lines):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "r05_config3_replay.txt"), "a") as f:
        f.write("\n".join(lines) + "\n")


def _dd(dec):
    """80-digit Decimal -> (hi, lo) doubles"""
    from decimal import Decimal
    hi = float(dec)
    return hi, float(dec - Decimal(hi))


def _explain_inversions(oracle, sc, inst, g, r, strat, res, ref_ids, ref_score):
    """-> (inverted pairs [(a, b)] or None when there are more than 1000, report lines); asserts the per-pair criterion of the
    comment above"""
    import exact_eig
    n, L = inst["nb_vars"], inst["nb_lifted"]
    vv = np.asarray(g["r%02d_vars" % r], dtype=np.float64)
    ours = res["idx"].astype(np.int64)
    w = ref_ids.shape[0]
    diff = np.flatnonzero(ours != ref_ids)
    # everybody between the first and the last differing position: a candidate that kept its place can still be inverted
    # against one that moved across it
    span = slice(int(diff[0]), int(diff[-1]) + 1)
    involved = np.unique(np.concatenate([ours[span], ref_ids[span]]))
    only_ref, only_ours = np.setdiff1d(ref_ids, ours), np.setdiff1d(ours, ref_ids)
    # rank of every involved candidate on both sides (w = behind the whole head)
    pos_ref = {int(c): i for i, c in enumerate(ref_ids)}
    pos_our = {int(c): i for i, c in enumerate(ours)}
    pr = np.array([pos_ref.get(int(c), w) for c in involved])
    po = np.array([pos_our.get(int(c), w) for c in involved])
    eig_dev, obj_dev = sc.get_scores(eig=strat != 2, obj=strat != 1)      # what the round just run scored
    S, ks = sc.get_candidates(involved)
    m = involved.shape[0]
    true_hi, true_lo, e_ref, e_dev, kind = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m), []
    for i, c in enumerate(involved):
        k = int(ks[i])
        si = S[i, :k]
        x, X = vv[L:][si], vv[:L][oracle.triu_positions(si, n)]
        recorded = ref_score[pos_ref[int(c)]] if int(c) in pos_ref else None
        lam_here = float(oracle.get_eigendecomp(k, x, X, False)[0])
        lam_dev = float(eig_dev[c]) if eig_dev is not None else None
        is_eig = strat == 1 or (strat == 4 and recorded is not None and abs(recorded + lam_dev) <= 1e-12)
        if is_eig:      # the score is -lambda_min
            t = exact_eig.exact_lambda_min_of(k, x, X, lam_here)
            hi, lo = _dd(-t)
            from decimal import Decimal
            e_ref[i] = abs(float(Decimal(float(recorded if recorded is not None else -lam_here)) - (-t)))
            e_dev[i] = abs(float(Decimal(-lam_dev) - (-t)))
            assert e_ref[i] <= EIG_ERR_MAX and e_dev[i] <= EIG_ERR_MAX, (r, int(c), e_ref[i], e_dev[i])
        else:           # obj_improve (+- BIG_M): the truth is the reference's own arithmetic
            o_ref = float(oracle.opt_score_batch(k, si[None, :], n, vv, inst["Q_arr"])[0])
            off = 0.0 if recorded is None else round((recorded - o_ref) / 1000.0) * 1000.0
            hi, lo = o_ref + off, 0.0
            e_ref[i] = 0.0 if recorded is None else abs(recorded - (o_ref + off))      # the rounding of obj_improve + 1000
            e_dev[i] = abs(float(obj_dev[c]) - o_ref)
            tol = OBJ_ERR_REL * max(abs(o_ref), 1e-3 * k * max(1.0, float(np.abs(inst["Q_arr"][oracle.triu_positions(si, n)]).max())))
            assert e_dev[i] <= tol and e_ref[i] <= 2.3e-13, (r, int(c), e_dev[i], e_ref[i])
        true_hi[i], true_lo[i] = hi, lo
        kind.append("eig" if is_eig else "opt")
    # inverted pairs: ordered one way by the reference, the other way by us (both outside one head: no statement)
    inv = ((pr[:, None] < pr[None, :]) & (po[:, None] > po[None, :]))
    ia, ib = np.nonzero(inv)
    d_true = np.abs((true_hi[ia] - true_hi[ib]) + (true_lo[ia] - true_lo[ib]))
    budget = e_ref[ia] + e_ref[ib] + e_dev[ia] + e_dev[ib]
    bad = d_true > budget + 1e-30
    assert not bad.any(), (r, [(int(involved[a]), int(involved[b]), float(d), float(q)) for a, b, d, q in
                               zip(ia[bad][:5], ib[bad][:5], d_true[bad][:5], budget[bad][:5])])
    pairs = [(int(involved[a]), int(involved[b])) for a, b in zip(ia, ib)] if ia.size <= 1000 else None
    lines = ["    round %d: %d positions differ, %d / %d ids on one side only, %d inverted pairs, largest true difference %.3e "
             "(its error budget %.3e), kinds %s" % (r, diff.size, only_ref.size, only_ours.size, ia.size,
                                                    d_true.max() if d_true.size else 0.0,
                                                    budget[np.argmax(d_true)] if d_true.size else 0.0, sorted(set(kind)))]
    for d, i, j in list(zip(d_true, ia, ib))[:12]:
        lines.append("        pair (%d, %d): true difference %.3e, errors reference %.2e + %.2e, device %.2e + %.2e"
                     % (int(involved[i]), int(involved[j]), d, e_ref[i], e_ref[j], e_dev[i], e_dev[j]))
    return pairs, lines


@pytest.mark.parametrize("path", _trajectories(), ids=[os.path.basename(p)[:-4] for p in _trajectories()])
def test_replay_of_the_reference_trajectory(path, oracle):
    """Every round the reference ran (its LP point, its rank-list head, its strategy switch, its number
    of cuts): same selection from the library.  Identical selections make the next LP -- and hence the
    whole trajectory -- identical, so this is configs[2] end to end, minus the LP solver.

    Ten recorded trajectories, 199 rounds: spar125-075-{1,2,3} dim 4 combined (20 rounds each, 1.6-1.7e6 candidates),
    spar125-050-1 / spar100-050-1 / spar070-050-1 dim 5 combined (mixed 2..5-variable covers), spar125-075-1 dim 3 and
    spar090-075-1 dim 4 optimality, spar125-075-2 dim 3 combined, spar080-075-1 dim 4 feasibility from the first round.
    EVERY position of EVERY head is the reference's, except in the rounds named by ADMITTED_PAIRS / STRUCTURED_ROUNDS, where each
    inverted pair is proven at run time to lie inside the two solvers' measured errors on those two matrices."""
    import sdpcutsel_via_nn_amd as pkg
    g = np.load(path)
    base = os.path.basename(path)[:-4]
    name, dim, sel = str(g["name"]), int(g["dim"]), int(g["sel_size"])
    inst = _instance(name)
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(dim)
        sc.set_instance(inst["nb_vars"], inst["Q_arr"])
        N = sc.set_candidates_cover(inst["adj"], dim)
        assert N == int(g["nb_subproblems"])
        rounds = int(g["rounds_done"])
        assert rounds >= 3
        report = []
        for r in range(1, rounds + 1):
            p = "r%02d_" % r
            strat = int(g[p + "strat"])
            sc.set_point(g[p + "vars"])
            res = sc.select_round(strat, sel)
            ref_ids, ref_score = g[p + "ids"].astype(np.int64), g[p + "score"]
            w = ref_ids.shape[0]
            assert res["idx"].shape[0] == w and res["n_total"] == int(g[p + "list_len"]), (r, strat)
            assert res["new_strat"] == int(g[p + "new_strat"]), (r, strat)
            tol = 1e-9 * np.maximum(1.0, np.abs(ref_score)) + 2e-13
            assert np.all(np.abs(res["score"] - ref_score) <= tol), (r, np.abs(res["score"] - ref_score).max())
            same = res["idx"] == ref_ids
            n_set = int(np.setdiff1d(res["idx"], ref_ids).size)
            extra = []
            if not same.all():
                key = (base, r)
                pairs, extra = _explain_inversions(oracle, sc, inst, g, r, strat, res, ref_ids, ref_score)
                _append_report(["%s round %d (strategy %d): inverted pairs %s" % (base, r, strat, pairs)] + extra)
                assert key in ADMITTED_PAIRS or key in STRUCTURED_ROUNDS, \
                    "%s round %d must reproduce the reference's head position by position: %d positions differ (%s); pairs %s" % (
                        base, r, int((~same).sum()), np.flatnonzero(~same)[:10].tolist(), pairs)
                if key in ADMITTED_PAIRS:
                    assert n_set == 0, (r, n_set)
                    want = ADMITTED_PAIRS[key]
                    assert want is not None and pairs is not None and sorted(map(tuple, map(sorted, pairs))) == sorted(map(tuple, map(sorted, want))), (key, pairs)
                else:
                    assert strat == 1, (r, strat)
            nb_cuts = int((res["lam"] < -1e-15).sum())
            assert nb_cuts == int(g[p + "nb_cuts"]), (r, nb_cuts)
            report.append("%s dim %d round %2d strategy %d -> %d: %d candidates, head %d, positions with another id %d, "
                          "ids selected by one side only %d, cuts %d" % (name, dim, r, strat, res["new_strat"], N, w,
                                                                        int((~same).sum()), n_set, nb_cuts))
            report.extend(extra)
        _append_report(report)
    finally:
        sc.close()


def test_live_rounds_dim4_device_cover():
    """Three live rounds on spar125-075-1, dim 4 (1 700 215 candidates of sizes 2..4 enumerated on the
    device), combined strategy, 5000 cuts per round (cut_select_qp.py:37)."""
    import sdpcutsel_via_nn_amd as pkg
    from sdpcutsel_via_nn_amd.cut_solver import DeviceAgg
    cs = pkg.CutSolver()
    bounds, t_total, rt, st, cuts, tri, nsub = cs.cut_select_algo(os.path.join(GOLDEN, "instances", "spar125-075-1.in"), 4, 0.1,
                                                                  strat=4, nb_rounds_cuts=3)
    assert nsub == 1700215 and isinstance(cs._agg_list, DeviceAgg)
    assert cuts == [0, 5000, 5000, 5000]
    assert bounds[0] > bounds[1] > bounds[2] > bounds[3] > 12330.0             # best known value, boxqp_instances/filenames.txt:97
    # separation = selection + generation; the reference needs 2.5 s for 133 242 candidates (dim 3, round 1)
    assert all(s < 0.25 for s in st[1:]), st
    e = cs._agg_list[1700214]                                                  # records on demand, from the device
    assert len(e[0]) in (2, 3, 4) and len(e[1]) == len(e[0]) * (len(e[0]) + 1) // 2
    # rank-list entries over a list that only exists on the device: built from fetched index sets
    vv = np.asarray(cs._my_prob.get_values())
    rl = cs._sel_eigcut_by_ordering_on_measure(2, vv, 99)
    head = rl[0:20]
    L = cs._nb_lifted
    for idx, score, curr_pt, X_slice in head:
        rec = cs._agg_list[idx]
        assert isinstance(idx, int) and isinstance(score, float)
        assert curr_pt == tuple(vv[L + i] for i in rec[0]) and X_slice == tuple(vv[i] for i in rec[1])
    assert [x[1] for x in head] == sorted((x[1] for x in head), reverse=True)
    feas = cs._sel_eigcut_by_ordering_on_measure(1, vv, 99)
    if len(feas):
        f = feas[0:10][0]
        assert f[3] == len(f[0]) and f[2] == cs._agg_list[f.agg_idx][1]


@pytest.mark.parametrize("point", ["mccormick_vertex", "generic"])
def test_dim5_round_past_the_reference_cap(oracle, point):
    """One selection round over the dim-5 cover of spar125-075-1: 12 845 805 candidates of sizes 2..5 (the
    reference returns at its 4e6 guard, cut_select_qp.py:117-120).  Size-independent properties: sampled
    scores of every size class equal the oracle's, the head of every strategy is the oracle's ranking of
    the device's scores, the fused round returns the rows of that head."""
    import sdpcutsel_via_nn_amd as pkg
    from sdpcutsel_via_nn_amd import _capi, harness
    inst = _instance("spar125-075-1")
    n, L = inst["nb_vars"], inst["nb_lifted"]
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(5)
        sc.set_instance(n, inst["Q_arr"])
        assert sc.set_candidates_cover(inst["adj"], 5, max_subs=4 * 10 ** 6) == 12845805 and sc.N == 0      # the reference's guard
        N = sc.set_candidates_cover(inst["adj"], 5)                                                        # max_subs = None
        assert N == sc.N == 12845805
        if point == "generic":
            vv = harness.random_mccormick_point(n, np.random.default_rng(15))
        else:
            lp = harness.boxqp_relaxation(inst)
            lp.solve()
            vv = np.asarray(lp.get_values())
            assert np.allclose(vv[L:], 0.5)
        sc.set_point(vv)
        sc.score(_capi.EIG | _capi.NN)
        eig, obj = sc.get_scores()
        assert np.isfinite(eig).all() and np.isfinite(obj).all()
        rng = np.random.default_rng(6)
        pick = np.unique(np.concatenate([np.arange(4096), N - 1 - np.arange(4096), rng.choice(N, 60000, replace=False)]))
        S, ks = sc.get_candidates(pick)
        sizes = np.unique(ks)
        assert set(sizes.tolist()) <= {2, 3, 4, 5} and 5 in sizes
        for k in sizes:
            m = ks == k
            si = S[m, :k]
            ref_obj = oracle.opt_score_batch(int(k), si, n, vv, inst["Q_arr"])
            ref_eig = oracle.eigmin_batch(int(k), vv[L:][si], vv[:L][oracle.triu_positions(si, n)])
            me = int(k) * np.abs(inst["Q_arr"][oracle.triu_positions(si, n)]).max(axis=1)
            me[me == 0] = 1.0
            assert np.abs(eig[pick][m] - ref_eig).max() <= 2e-13, int(k)
            assert np.all(np.abs(obj[pick][m] - ref_obj) <= 1e-9 * np.maximum(np.abs(ref_obj), 1e-3 * me)), int(k)
        for strat in (1, 2, 4):
            ids, score, total, new_strat, cnt = sc.rank(strat, 5000, max_out=5000)
            order, ref_score, ref_strat, ref_cnt = oracle.rank_arrays(strat, obj, eig, 5000)
            assert np.array_equal(ids, order[:5000]), strat
            assert np.array_equal(score, ref_score[:5000] + 0.0) and new_strat == ref_strat and total == order.shape[0]
        sc.set_point(vv)
        r = sc.select_round(4, 5000)
        order, ref_score, ref_strat, _ = oracle.rank_arrays(4, obj, eig, 5000)
        assert np.array_equal(r["idx"], order[:5000]) and np.array_equal(r["score"], ref_score[:5000] + 0.0)
        assert r["new_strat"] == ref_strat
        lam, coef, rhs, cols, ks_r = sc.cut_rows(order[:5000])
        ld = r["coef"].shape[1]
        assert np.array_equal(r["lam"], lam) and np.array_equal(r["coef"], coef[:, :ld]) and np.array_equal(r["rhs"], rhs)
        assert np.array_equal(r["ks"], ks_r) and len(set(ks_r.tolist())) >= 1
    finally:
        sc.close()
