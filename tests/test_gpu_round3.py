"""GPU tests of the round-3 additions (run with -m gpu on an MI355X): the eigenvalue-only kernel of the
feasibility rounds (csrc/eig.hip) against the scoring kernels' eigenvalue branch and the oracle, on single-size
and mixed-size lists, with and without the selection's leading-digit histogram."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import sdpcutsel_via_nn_amd as p
    return p


def _mixed_workload(nb_vars, sizes, count, seed):
    """index sets of mixed sizes in one list (caller order interleaves the classes)"""
    from sdpcutsel_via_nn_amd import synthetic
    Q_arr, vv, rng = synthetic.make_instance(nb_vars, seed)
    ks = rng.choice(np.asarray(sizes, dtype=np.int32), size=count)
    S = np.full((count, 5), -1, dtype=np.int32)
    for k in sizes:
        m = np.flatnonzero(ks == k)
        S[m, :k] = synthetic.random_index_sets(nb_vars, int(k), m.size, rng)
    return Q_arr, vv, S, ks.astype(np.int32)


@pytest.mark.parametrize("sizes,count,nb_vars", [((3,), 10 ** 6, 100), ((2, 3, 4), 300000, 60), ((2, 3, 4, 5), 200000, 40),
                                                 ((5,), 70000, 30), ((2,), 1000, 20), ((4, 2), 257, 25)])
def test_eig_only_kernel_bit_equal_to_the_scoring_kernels(pkg, oracle, sizes, count, nb_vars):
    """lambda_min from eig_only_kernel == lambda_min from score_mfma_kernel (eigenvalue branch and full launch) bit for
    bit -- the same Jacobi template -- and within 2e-13 of LAPACK; every size mix, ragged last tiles included."""
    from sdpcutsel_via_nn_amd import _capi
    Q_arr, vv, S, ks = _mixed_workload(nb_vars, sizes, count, seed=21 + len(sizes))
    L = nb_vars * (nb_vars + 1) // 2
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(5)
        sc.set_instance(nb_vars, Q_arr)
        sc.set_candidates(S, ks)
        out = {}
        for name, opt, flags in (("eig_kernel", 1, _capi.EIG), ("mfma_branch", 0, _capi.EIG), ("mfma_full", 0, _capi.EIG | _capi.NN)):
            sc.set_option(_capi.OPT_EIG_KERNEL, opt)
            sc.set_point(vv)
            sc.score(flags)
            out[name] = sc.get_scores(obj=False)[0]
        assert np.array_equal(out["eig_kernel"], out["mfma_branch"])
        assert np.array_equal(out["eig_kernel"], out["mfma_full"])
        pick = np.unique(np.concatenate([np.arange(min(count, 3000)), count - 1 - np.arange(min(count, 3000))]))
        for k in sizes:
            m = pick[ks[pick] == k]
            si = S[m, :k]
            ref = oracle.eigmin_batch(int(k), vv[L:][si], vv[:L][oracle.triu_positions(si, nb_vars)])
            assert np.abs(out["eig_kernel"][m] - ref).max() <= 2e-13, int(k)
    finally:
        sc.close()


@pytest.mark.parametrize("sizes,count,nb_vars,sel", [((3,), 10 ** 6, 100, 5000), ((2, 3, 4), 300000, 60, 5000),
                                                     ((2, 3, 4, 5), 200000, 40, 700), ((3, 4), 9000, 30, 12000)])
def test_feasibility_round_through_the_eig_kernel(pkg, oracle, sizes, count, nb_vars, sel):
    """A fused feasibility round (sdpcut_round_view, strategy 1): the eigenvalue kernel counts the leading digit of the
    selection keys; ids / scores / rows == the round with the option off == the oracle's ranking of the device's values."""
    from sdpcutsel_via_nn_amd import _capi
    Q_arr, vv, S, ks = _mixed_workload(nb_vars, sizes, count, seed=5)
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(5)
        sc.set_instance(nb_vars, Q_arr)
        sc.set_candidates(S, ks)
        res = {}
        for opt in (1, 0):
            sc.set_option(_capi.OPT_EIG_KERNEL, opt)
            res[opt] = sc.select_round(1, sel, point=vv)
        eig = sc.get_scores(obj=False)[0]
        order, ref_score, _, _ = oracle.rank_arrays(1, None, eig, sel)
        w = min(sel, order.shape[0])
        for opt in (1, 0):
            r = res[opt]
            assert r["n_total"] == order.shape[0] and r["idx"].shape[0] == w
            assert np.array_equal(r["idx"], order[:w]) and np.array_equal(r["score"], ref_score[:w] + 0.0)
            assert r["counters"]["nb_violated"] == order.shape[0]
        for f in ("lam", "coef", "rhs", "ks"):
            assert np.array_equal(res[1][f], res[0][f]), f
        assert sc.get_stat(_capi.STAT_SELECT_FALLBACKS) == 0
    finally:
        sc.close()


def test_feasibility_rounds_of_the_reference_trajectory_through_the_eig_kernel(pkg):
    """The fifteen pure-feasibility rounds the reference ran on spar125-075-1, dim 4 (rounds 6..20 of
    tests/golden/rounds_spar125_075_1_d4_s4.npz): the eigenvalue kernel + selection return the reference's ids in the
    reference's order (the replay test covers all rounds; this one pins WHICH kernel served them)."""
    from sdpcutsel_via_nn_amd import _capi, harness
    g = np.load(os.path.join(GOLDEN, "rounds_spar125_075_1_d4_s4.npz"))
    inst = harness.parse_boxqp(os.path.join(GOLDEN, "instances", "spar125-075-1.in"))
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(4)
        sc.set_instance(inst["nb_vars"], inst["Q_arr"])
        assert sc.set_candidates_cover(inst["adj"], 4) == int(g["nb_subproblems"])
        sc.set_option(_capi.OPT_TIMING, 1)
        seen = 0
        for r in range(1, int(g["rounds_done"]) + 1):
            p = "r%02d_" % r
            if int(g[p + "strat"]) != 1:
                continue
            res = sc.select_round(1, int(g["sel_size"]), point=g[p + "vars"])
            assert np.array_equal(res["idx"], g[p + "ids"].astype(np.int64)), r
            assert np.abs(res["score"] - g[p + "score"]).max() <= 2e-13
            assert sc.last_timing()[0] > 0.0
            seen += 1
        assert seen >= 10
    finally:
        sc.close()


# ----------------------------------------------------------------------------- the round with assembled cuts (sdpcut_round_csr)
def _padded_reference(sc, strat, sel, vv):
    """the same round through sdpcut_round_view + sdpcut_cut_rows (padded rows, columns from the library)"""
    r = sc.select_round(strat, sel, point=vv)
    lam, coef, rhs, cols, ks = sc.cut_rows(r["idx"] - sc.base)
    return r, lam, coef, rhs, cols, ks


@pytest.mark.parametrize("sizes,count,nb_vars,sel", [((3,), 10 ** 6, 100, 5000), ((2, 3, 4), 300000, 60, 5000),
                                                     ((2, 3, 4, 5), 200000, 40, 700), ((3, 4), 9000, 30, 12000),
                                                     ((2, 5), 3000, 30, 5000), ((4,), 130, 12, 64), ((3,), 40000, 50, 16384)])
@pytest.mark.parametrize("strat", [1, 2, 4])
def test_round_csr_equals_the_padded_rows(pkg, sizes, count, nb_vars, sel, strat):
    """sdpcut_round_csr: head == sdpcut_round_view's; the CSR block == the cuts of _gen_eigcuts_selected built from the
    padded rows on the host (kept iff lam < -1e-15, columns [L + i] + Xarr_inds, cut_select_qp.py:739-750), bit for bit."""
    from sdpcutsel_via_nn_amd import _capi
    from sdpcutsel_via_nn_amd.cut_solver import rows_to_csr
    Q_arr, vv, S, ks_all = _mixed_workload(nb_vars, sizes, count, seed=31)
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(5)
        sc.set_instance(nb_vars, Q_arr)
        sc.set_candidates(S, ks_all)
        r, lam, coef, rhs, cols, ks = _padded_reference(sc, strat, sel, vv)
        c = sc.round_csr(strat, sel, point=vv, copy=True)
        assert np.array_equal(c["idx"], r["idx"]) and np.array_equal(c["score"], r["score"])
        assert c["n_total"] == r["n_total"] and c["new_strat"] == r["new_strat"] and c["counters"] == r["counters"]
        assert np.array_equal(c["lam"], lam) and np.array_equal(c["ks"], ks)
        assert np.array_equal(c["set_inds"], S[c["idx"]])
        keep = np.flatnonzero(lam < -1e-15)
        assert np.array_equal(c["row_entry"], keep)
        indptr, ind, val = rows_to_csr(coef[keep], cols[keep], ks[keep])
        assert np.array_equal(c["indptr"], indptr) and np.array_equal(c["indices"], ind) and np.array_equal(c["values"], val)
        assert np.array_equal(c["rhs"], rhs[keep])
        # the same point again without an upload (vars_values = NULL)
        c2 = sc.round_csr(strat, sel, point=None, copy=True)
        for f in ("idx", "score", "indptr", "indices", "values", "rhs", "row_entry"):
            assert np.array_equal(c[f], c2[f]), f
        assert sc.get_stat(_capi.STAT_SELECT_FALLBACKS) == 0
    finally:
        sc.close()


def test_round_csr_general_regime_and_short_lists(pkg, oracle):
    """combined strategy with fewer strong candidates than the quota (every entry visited), a feasibility list shorter
    than the head asked for, and an all-PSD point (no cut at all)"""
    from sdpcutsel_via_nn_amd import synthetic
    from sdpcutsel_via_nn_amd.cut_solver import rows_to_csr
    wl = synthetic.make_workload(nb_vars=40, k=3, count=60000, seed=9)
    n, L = 40, 820
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(3)
        sc.set_instance(n, wl["Q_arr"])
        sc.set_candidates(wl["set_inds"], wl["ks"])
        x = wl["vars_values"][L:]
        iu = np.triu_indices(n)
        psd = np.concatenate([np.minimum(x[iu[0]], x[iu[1]]), x])          # X = min(x_i, x_j) is PSD: nothing violated
        partial = 0
        for vv, sel in ((wl["vars_values"], 5000), (psd, 5000), (0.999 * psd + 0.001 * wl["vars_values"], 3000)):
            for strat in (4, 1, 2):
                r, lam, coef, rhs, cols, ks = _padded_reference(sc, strat, sel, vv)
                c = sc.round_csr(strat, sel, point=vv, copy=True)
                assert np.array_equal(c["idx"], r["idx"]) and np.array_equal(c["score"], r["score"]) and c["new_strat"] == r["new_strat"]
                keep = np.flatnonzero(lam < -1e-15)
                indptr, ind, val = rows_to_csr(coef[keep], cols[keep], ks[keep])
                assert np.array_equal(c["row_entry"], keep) and np.array_equal(c["indptr"], indptr)
                assert np.array_equal(c["indices"], ind) and np.array_equal(c["values"], val) and np.array_equal(c["rhs"], rhs[keep])
                partial += 0 < keep.size < c["idx"].shape[0]
        assert partial >= 1          # heads with violated AND non-violated entries: the compaction skipped rows
        r = sc.round_csr(1, 5000, point=psd)
        assert r["idx"].shape[0] == 0 and r["rhs"].shape[0] == 0 and r["n_total"] == 0 and r["indptr"].tolist() == [0]
    finally:
        sc.close()


@pytest.mark.parametrize("strat", [1, 2, 4])
def test_dropin_pair_hands_over_the_fused_rows(pkg, strat):
    """_sel_eigcut_by_ordering_on_measure + _gen_eigcuts_selected (cut_select_qp.py:165-182) on the mixin: the selection
    runs the fused round and the generation consumes its assembled cuts; the LP's row store ends up with exactly the rows
    the two-call route (rank, then sdpcut_cut_rows + host assembly) produces -- also for a shorter prefix, for strong_only,
    through the reference's per-row objects, and after the block went stale."""
    from sdpcutsel_via_nn_amd import harness, synthetic
    from sdpcutsel_via_nn_amd.cut_solver import CutSolver, AggArrays, RankList
    Q_arr, vv, S, ks = _mixed_workload(50, (2, 3, 4), 120000, seed=77)
    n, L = 50, 1275

    def solver():
        cs = CutSolver()
        cs.set_instance(n, Q_arr, AggArrays(S, ks, n, Q_arr), 4, my_prob=harness.LinearRelaxation(np.zeros(L + n)))
        return cs

    def run(cs, sel, fused, strong_only=False, store=None):
        if store is not None:
            cs._my_prob.linear_constraints = store
        out = cs._sel_eigcut_by_ordering_on_measure(strat, vv, 1, **({"sel_size": sel} if strat == 4 else {}))
        rl = out[1] if strat == 4 else out
        assert isinstance(rl, RankList) and rl._fused is not None
        if not fused:
            rl._fused = None                  # the two-call route: ids -> sdpcut_cut_rows -> rows_to_csr on the host
        nb = cs._gen_eigcuts_selected(strat, sel, rl, strong_only=strong_only, vars_values=vv)
        return nb, cs._my_prob.linear_constraints

    for sel, strong_only in ((5000, False), (1234, False), (5000, True)):
        a, b = solver(), solver()
        nb_a, st_a = run(a, sel, True, strong_only)
        nb_b, st_b = run(b, sel, False, strong_only)
        assert nb_a == nb_b == st_a.get_num() > 0
        da, ca, la = st_a.csr_parts()
        db, cb, lb = st_b.csr_parts()
        assert np.array_equal(da, db) and np.array_equal(ca, cb) and np.array_equal(la, lb)
        assert st_a.rhs == st_b.rhs and st_a.senses == st_b.senses

    class RefStore(object):                   # the reference's LP surface: only add(lin_expr=, rhs=, senses=)
        def __init__(self):
            self.rows, self.rhs = [], []
        def add(self, lin_expr=(), rhs=(), senses=()):
            self.rows.extend(lin_expr); self.rhs.extend(rhs)
            assert len(senses) == len(lin_expr) and all(s == "G" for s in senses)
    c = solver()
    c._sparse_pair = harness.SparsePair
    nb_c, st_c = run(c, 5000, True, store=RefStore())
    a = solver()
    nb_a, st_a = run(a, 5000, True)
    rows_a = st_a.rows
    assert nb_c == nb_a == len(st_c.rows)
    assert all(x.ind == y.ind and x.val == y.val for x, y in zip(st_c.rows, rows_a)) and st_c.rhs == st_a.rhs

    # a later round on the same scorer overwrites the pinned block: the older list must notice
    d = solver()
    out = d._sel_eigcut_by_ordering_on_measure(strat, vv, 1, **({"sel_size": 5000} if strat == 4 else {}))
    rl_old = out[1] if strat == 4 else out
    vv2 = harness.random_mccormick_point(n, np.random.default_rng(3))
    d._sel_eigcut_by_ordering_on_measure(strat, vv2, 2, **({"sel_size": 5000} if strat == 4 else {}))
    assert rl_old.fused_rows(5000) is None
    nb_d = d._gen_eigcuts_selected(strat, 5000, rl_old, vars_values=vv)
    dd, cd, ld_ = d._my_prob.linear_constraints.csr_parts()
    da, ca, la = st_a.csr_parts()
    # (the stale list's rows are generated after a fresh upload of the point, by Jacobi when the round's eigenvalues are gone:
    # the same cuts to the accuracy two eigen-solvers agree to)
    assert nb_d == nb_a and np.array_equal(cd, ca) and np.abs(dd - da).max() <= 1e-9


def test_qcqp_round_keeps_the_fused_rows_through_slices(pkg, oracle, golden_qcqp):
    """CutSolverQCQP.select_and_generate_round slices both rank lists (cut_select_qcqp.py:79-97); the slices still lead
    to the cuts their fused rounds assembled (no second trip to the device: sdpcut_cut_rows is never called), and the
    rows are the oracle's QCQP round's."""
    from conftest import agg_from_arrays
    from sdpcutsel_via_nn_amd import _capi, harness
    from sdpcutsel_via_nn_amd.cut_solver import CutSolverQCQP
    g = golden_qcqp
    n = int(g["nb_vars"])
    L = n * (n + 1) // 2
    agg_o = agg_from_arrays(oracle, g["obj_set_inds"], g["obj_k"], n, g["Q_arr"])
    agg_c = agg_from_arrays(oracle, g["cons_set_inds"], g["cons_k"], n, g["Q_arr"])
    calls = []
    orig = _capi.Scorer.cut_rows
    _capi.Scorer.cut_rows = lambda self, idx: calls.append(len(idx)) or orig(self, idx)
    try:
        for strat, sel in ((4, 40), (2, 7), (1, 40)):
            cs = CutSolverQCQP()
            lp = harness.LinearRelaxation(np.zeros(L + n))
            cs.set_instance(n, g["Q_arr"], agg_o, dim=3, my_prob=lp)
            new_strat, rank_list, nb_cuts, nb_opt = cs.select_and_generate_round(strat, g["vars"], 1, sel, agg_o, agg_c)
            ref = oracle.qcqp_round(agg_o, agg_c, L, strat, g["vars"], sel)
            assert nb_cuts == ref["nb_sdp_cuts"] == lp.linear_constraints.get_num()
            for row, (ind, val) in zip(lp.linear_constraints.rows, ref["rows"]):
                assert row.ind == list(ind)
                assert np.abs(np.array(row.val) - np.array(val, dtype=np.float64)).max() <= 1e-9
            q = "s%d_sel%d" % (strat, sel)
            assert [isinstance(e[0], int) for e in rank_list] == g[q + "_is_obj"].tolist()
        assert calls == []
    finally:
        _capi.Scorer.cut_rows = orig


# ----------------------------------------------------------------------------- selection paths, LP point buffer
@pytest.mark.parametrize("strat", [1, 2, 4])
def test_selection_without_in_kernel_waits_equals_the_fused_one(pkg, strat):
    """SDPCUT_OPT_FUSED_TAIL = 0 (one launch per digit, the path that answers when a bounded wait of the fused kernel
    expires) == the fused selection: spread-out scores, masses of equal keys (duplicated candidates), short lists."""
    from sdpcutsel_via_nn_amd import _capi, synthetic
    for nb_vars, count, sel in ((100, 400000, 5000), (12, 300000, 5000), (30, 6000, 3000), (9, 200000, 16000)):
        wl = synthetic.make_workload(nb_vars=nb_vars, k=3, count=count, seed=41)
        res = {}
        for fused in (1, 0):
            sc = pkg.Scorer(0)
            try:
                sc.set_builtin_networks(3)
                sc.set_option(_capi.OPT_FUSED_TAIL, fused)
                sc.set_instance(nb_vars, wl["Q_arr"])
                sc.set_candidates(wl["set_inds"], wl["ks"])
                res[fused] = sc.select_round(strat, sel, point=wl["vars_values"])
                assert sc.get_stat(_capi.STAT_SELECT_FALLBACKS) == 0
            finally:
                sc.close()
        for f in ("idx", "score", "lam", "coef", "rhs", "ks"):
            assert np.array_equal(res[1][f], res[0][f]), (nb_vars, f)
        assert res[1]["n_total"] == res[0]["n_total"] and res[1]["new_strat"] == res[0]["new_strat"]
        assert res[1]["counters"] == res[0]["counters"]


def test_lp_point_through_the_pinned_buffer(pkg):
    """sdpcut_point_buffer: the point written in place into the handle's staging block gives the round the same results
    as the point handed over in an ordinary array -- also after the instance (and with it the block) changed."""
    from sdpcutsel_via_nn_amd import synthetic
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(3)
        for nb_vars, count in ((40, 50000), (200, 80000), (40, 20000)):
            wl = synthetic.make_workload(nb_vars=nb_vars, k=3, count=count, seed=8)
            sc.set_instance(nb_vars, wl["Q_arr"])
            sc.set_candidates(wl["set_inds"], wl["ks"])
            a = sc.select_round(4, 2000, point=wl["vars_values"])
            buf = sc.point_buffer()
            assert buf.shape == wl["vars_values"].shape
            buf[:] = 0.0
            b0 = sc.select_round(4, 2000, point=buf)
            assert b0["n_total"] == count and not np.array_equal(b0["idx"], a["idx"])      # really another point
            buf[:] = wl["vars_values"]
            b = sc.select_round(4, 2000, point=buf)
            c = sc.round_csr(4, 2000, point=buf, copy=True)
            for f in ("idx", "score", "lam", "coef", "rhs"):
                assert np.array_equal(a[f], b[f]), f
            assert np.array_equal(c["idx"], a["idx"])
    finally:
        sc.close()


def test_triangle_separation_without_in_kernel_waits(pkg):
    from sdpcutsel_via_nn_amd import _capi, harness
    inst = harness.parse_boxqp(os.path.join(GOLDEN, "instances", "spar125-075-1.in"))
    vv = harness.random_mccormick_point(inst["nb_vars"], np.random.default_rng(2))
    out = {}
    for fused in (1, 0):
        sc = pkg.Scorer(0)
        try:
            sc.set_option(_capi.OPT_FUSED_TAIL, fused)
            sc.set_instance(inst["nb_vars"], inst["Q_arr"])
            sc.tri_preprocess(inst["adj"])
            sc.set_point(vv)
            out[fused] = sc.tri_separate(10000)
        finally:
            sc.close()
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1]) and out[1][2] == out[0][2] > 10000


# ----------------------------------------------------------------------------- QCQP covers on the device
@pytest.mark.parametrize("name,dim", [("q_20_4_25_1", 3), ("q_20_4_25_1", 5), ("q_30_6_50_1", 3), ("q_30_6_50_1", 4),
                                      ("q_40_8_25_1", 3), ("q_50_10_25_1", 5)])
def test_qcqp_covers_split_on_the_device(pkg, name, dim):
    """sdpcut_set_candidates_cover_split == the host route (two native enumerations + sorted lookup, harness.qcqp_covers)
    == cut_select_qcqp.py:314-334: same two lists in the same order; q_50_10_25_1 with 5-variable sub-problems against the
    reference's own cover (count and checksum in tests/golden/inst_qcqp50.npz)."""
    import zlib
    from sdpcutsel_via_nn_amd import _capi, harness
    inst = harness.parse_osil(os.path.join(GOLDEN, "instances", name + ".osil"))
    n = inst["nb_vars"]
    (So, ko), (Sc, kc) = harness.qcqp_covers(inst, dim, _capi.enumerate_cover)
    a, b = pkg.Scorer(0), pkg.Scorer(0)
    try:
        for sc in (a, b):
            sc.set_builtin_networks(dim)
            sc.set_instance(n, inst["Q_arr"])
        n_in, n_out = a.set_candidates_cover_split(b, inst["adj"], inst["adj_cons"], dim)
        assert (n_in, n_out) == (len(ko), len(kc))
        for sc, S, k in ((a, So, ko), (b, Sc, kc)):
            if len(k):
                S_d, k_d = sc.get_candidates(np.arange(len(k), dtype=np.int64))
                assert np.array_equal(S_d, S) and np.array_equal(k_d, k)
        if name == "q_50_10_25_1":
            g = np.load(os.path.join(GOLDEN, "inst_qcqp50.npz"))
            assert n_out == int(g["cons_count"]) and n_in == len(g["obj_k"])
            S_d, _ = b.get_candidates(np.arange(n_out, dtype=np.int64))
            assert zlib.crc32(np.ascontiguousarray(S_d).tobytes()) == int(g["cons_crc"])
        # the split lists are live candidate lists: a round on each
        vv = harness.random_mccormick_point(n, np.random.default_rng(4))
        for sc, S, k in ((a, So, ko), (b, Sc, kc)):
            if len(k) == 0:
                continue
            r = sc.round_csr(1, 50, point=vv)
            ref = pkg.Scorer(0)
            try:
                ref.set_builtin_networks(dim)
                ref.set_instance(n, inst["Q_arr"])
                ref.set_candidates(S, k)
                q = ref.round_csr(1, 50, point=vv)
                assert np.array_equal(r["idx"], q["idx"]) and np.array_equal(r["score"], q["score"]) and np.array_equal(r["values"], q["values"])
            finally:
                ref.close()
    finally:
        a.close()
        b.close()


# ----------------------------------------------------------------------------- round in two halves, overlapping lists
def test_round_csr_in_two_halves_on_two_handles(pkg):
    """sdpcut_round_csr_begin / _end: two handles begin before either ends; each ends with exactly what its one-call
    round returns (general-path regimes included); a second begin, or another call, on a pending handle is refused."""
    Q_arr, vv, S, ks = _mixed_workload(50, (2, 3, 4), 150000, seed=5)
    Q2, vv2, S2, ks2 = _mixed_workload(50, (3,), 9000, seed=6)
    x = vv[1275:]
    iu = np.triu_indices(50)
    psd = np.concatenate([np.minimum(x[iu[0]], x[iu[1]]), x])
    a, b = pkg.Scorer(0), pkg.Scorer(0)
    try:
        for sc, (Q, cand, kk) in ((a, (Q_arr, S, ks)), (b, (Q_arr, S2, ks2))):
            sc.set_builtin_networks(5)
            sc.set_instance(50, Q)
            sc.set_candidates(cand, kk)
        fields = ("idx", "score", "lam", "row_entry", "indptr", "indices", "values", "rhs")
        for point, sa, sb, sel_a, sel_b in ((vv, 4, 1, 5000, 5000), (vv, 1, 1, 16384, 100), (vv, 2, 4, 700, 9000),
                                            (0.999 * psd + 0.001 * vv, 4, 1, 3000, 5000), (psd, 1, 4, 5000, 64)):
            one_a = a.round_csr(sa, sel_a, point=point, copy=True)
            one_b = b.round_csr(sb, sel_b, point=point, copy=True)
            a.round_csr_begin(sa, sel_a, point=point)
            b.round_csr_begin(sb, sel_b, point=point)
            with pytest.raises(pkg.SdpCutError):
                a.round_csr_begin(sa, sel_a, point=point)
            with pytest.raises(pkg.SdpCutError):
                b.score(1)
            two_a = a.round_csr_end(copy=True)
            two_b = b.round_csr_end(copy=True)
            for one, two in ((one_a, two_a), (one_b, two_b)):
                for f in fields:
                    assert np.array_equal(one[f], two[f]), f
                assert (one["n_total"], one["new_strat"], one["counters"]) == (two["n_total"], two["new_strat"], two["counters"])
        with pytest.raises(pkg.SdpCutError):
            a.round_csr_end()                  # nothing pending
        # reading the list of a scorer with a begun round drops that round first (lists that live on the device are read
        # through their scorer); the library itself refuses every stateful call on a pending handle
        tok = b.round_csr_begin(1, 100, point=vv)
        assert b.pending is tok
        for call in (lambda: b.set_point(vv), lambda: b.rank(1, 10, max_out=10), lambda: b.cut_rows(np.arange(3)),
                     lambda: b.set_candidates(S2, ks2), lambda: b.get_scores()):
            with pytest.raises(pkg.SdpCutError):
                call()
        got, kk = b.get_candidates(np.arange(7))
        assert b.pending is None and np.array_equal(got[:, :3], S2[:7, :3]) and not b.drop_pending()
        again = b.round_csr(1, 100, point=vv, copy=True)
        assert again["idx"].shape[0] == 100
        # SDPCUT_OPT_STREAM_PRIORITY: the handle's stream re-created at the device's highest priority -- refused while a round
        # is pending, same results afterwards (and back)
        from sdpcutsel_via_nn_amd import _capi
        b.round_csr_begin(1, 100, point=vv)
        with pytest.raises(pkg.SdpCutError):
            b.set_option(_capi.OPT_STREAM_PRIORITY, 1)
        b.round_csr_end()
        for prio in (1, 0):
            b.set_option(_capi.OPT_STREAM_PRIORITY, prio)
            a.round_csr_begin(4, 5000, point=vv)
            b.round_csr_begin(1, 100, point=vv)
            ra, rb = a.round_csr_end(copy=True), b.round_csr_end(copy=True)
            assert all(np.array_equal(rb[f], again[f]) for f in fields) and ra["idx"].shape[0] == 5000
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("strat", [1, 2, 4])
def test_qcqp_loop_begins_both_covers_rounds_together(pkg, oracle, golden_qcqp, strat):
    """The reference's QCQP loop ranks the objective cover, then the constraints cover by feasibility at the same point
    (cut_select_qcqp.py:64-78).  The mixin learns that pair in the first round and from then on begins the second
    list's round together with the first's; the rounds' results are those of a solver that does not overlap, no
    speculative round is wasted, and a break of the pattern (another point for the second list) is noticed."""
    from conftest import agg_from_arrays
    from sdpcutsel_via_nn_amd import harness
    from sdpcutsel_via_nn_amd.cut_solver import CutSolverQCQP
    g = golden_qcqp
    n = int(g["nb_vars"])
    L = n * (n + 1) // 2
    agg_o = agg_from_arrays(oracle, g["obj_set_inds"], g["obj_k"], n, g["Q_arr"])
    agg_c = agg_from_arrays(oracle, g["cons_set_inds"], g["cons_k"], n, g["Q_arr"])
    rng = np.random.default_rng(11)
    points = [g["vars"]] + [harness.random_mccormick_point(n, rng) for _ in range(4)]

    def solver(overlap):
        cs = CutSolverQCQP()
        cs._gpu_overlap = overlap
        cs.set_instance(n, g["Q_arr"], agg_o, dim=3, my_prob=harness.LinearRelaxation(np.zeros(L + n)))
        return cs
    x, y = solver(True), solver(False)
    for r, vv in enumerate(points):
        out = []
        for cs in (x, y):
            cs._my_prob = harness.LinearRelaxation(np.zeros(L + n))
            res = cs.select_and_generate_round(strat, vv, r + 1, 40, agg_o, agg_c)
            st = cs._my_prob.linear_constraints
            out.append((res[0], res[2], res[3], [tuple(e[0]) if not isinstance(e[0], int) else e[0] for e in res[1]],
                        [e[1] for e in res[1]], st.csr_parts(), st.rhs))
        (s1, n1, o1, ids1, sc1, csr1, rhs1), (s2, n2, o2, ids2, sc2, csr2, rhs2) = out
        assert (s1, n1, o1) == (s2, n2, o2) and ids1 == ids2 and sc1 == sc2 and rhs1 == rhs2
        assert all(np.array_equal(p, q) for p, q in zip(csr1, csr2))
    bo, bc = x._gpu_bindings[id(agg_o)], x._gpu_bindings[id(agg_c)]
    assert bo.follower is not None and bo.follower[0] is bc and bo.wasted == bc.wasted == 0 and bc.pending is None
    assert all(b.follower is None for b in y._gpu_bindings.values())
    # the pattern breaks: the objective cover is ranked (the constraints cover's round begins with it), then the
    # constraints cover is asked about ANOTHER point -- the speculative round is dropped, the answer is the right one
    x._agg_list = agg_o
    x._sel_eigcut_by_ordering_on_measure(2, points[1], 9)
    assert bc.pending is not None
    x._agg_list = agg_c
    got = x._sel_eigcut_by_ordering_on_measure(1, points[2], 9)
    y._agg_list = agg_c
    want = y._sel_eigcut_by_ordering_on_measure(1, points[2], 9)
    assert bc.wasted == 1 and bc.pending is None
    assert np.array_equal(got.ids(40), want.ids(40)) and np.array_equal(got.scores(40), want.scores(40))
    x._agg_list = y._agg_list = agg_o


def test_small_size_classes_on_side_streams_give_the_same_round(pkg):
    """A list with one large and several small size classes scored by ONE launch over all classes (SDPCUT_OPT_ONE_LAUNCH, the
    default), by a launch per class one after the other, with the small classes on side streams, and with that form measured at
    first use (SDPCUT_OPT_SIDE_STREAMS 0 / 1 / 2): the same scores and the same round bit for bit, also right after a new list and
    next to a second handle's round."""
    from sdpcutsel_via_nn_amd import _capi
    Q_arr, vv, S, ks = _mixed_workload(60, (5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 3, 2), 33000, seed=41)
    a, b = pkg.Scorer(0), pkg.Scorer(0)
    try:
        for sc in (a, b):
            sc.set_builtin_networks(5)
            sc.set_instance(60, Q_arr)
        a.set_candidates(S, ks)
        b.set_candidates(S[:5000], ks[:5000])
        fields = ("idx", "score", "lam", "row_entry", "indptr", "indices", "values", "rhs")
        ref = {}
        for one, mode in ((0, 0), (0, 1), (0, 2), (0, 2), (1, 2)):      # a launch per class (sequential / side streams / measured); ONE launch
            a.set_option(_capi.OPT_ONE_LAUNCH, one)
            a.set_option(_capi.OPT_SIDE_STREAMS, mode)
            for strat in (4, 2, 1):
                r = a.round_csr(strat, 3000, point=vv, copy=True)
                if (one, mode) == (0, 0):
                    ref[strat] = r
                else:
                    assert all(np.array_equal(r[f], ref[strat][f]) for f in fields) and r["counters"] == ref[strat]["counters"], (one, mode, strat)
            a.score(_capi.EIG | _capi.NN)
            e, o = a.get_scores()
            if (one, mode) == (0, 0):
                ref["scores"] = (e, o)
            else:
                assert np.array_equal(e, ref["scores"][0]) and np.array_equal(o, ref["scores"][1]), (one, mode)
        a.set_option(_capi.OPT_ONE_LAUNCH, 0)
        a.set_candidates(S[::-1].copy(), ks[::-1].copy())          # a new list measures again
        r0 = a.round_csr(4, 3000, point=vv, copy=True)
        a.set_option(_capi.OPT_SIDE_STREAMS, 0)
        r1 = a.round_csr(4, 3000, point=vv, copy=True)
        assert all(np.array_equal(r0[f], r1[f]) for f in fields)
        a.set_option(_capi.OPT_SIDE_STREAMS, 1)
        a.round_csr_begin(4, 3000, point=vv)
        b.round_csr_begin(4, 500, point=vv)
        r2, _ = a.round_csr_end(copy=True), b.round_csr_end()
        assert all(np.array_equal(r2[f], r1[f]) for f in fields)
        with pytest.raises(ValueError):
            a.set_option(_capi.OPT_SIDE_STREAMS, 3)
    finally:
        a.close()
        b.close()


# ----------------------------------------------------------------------------- eigenvector of a known lambda_min
@pytest.mark.parametrize("k", [2, 3, 4, 5])
def test_rows_by_inverse_iteration_agree_with_jacobi(pkg, k):
    """When the scoring pass has computed lambda_min, the cut rows take its eigenvector by inverse iteration (jacobi.h,
    min_eigvec_known) instead of a second Jacobi with vectors.  Against the Jacobi rows of the same candidates: lambda to
    2e-13; coefficients to 1e-9 wherever LAPACK says the smallest eigenvalue is separated by 1e-6 (two eigen-solvers agree to
    eps / gap, no better); for EVERY row the two properties that make it the eigen-cut of lambda_min whatever the solver --
    unit vector, Rayleigh quotient = lambda_min; and at a structured vertex (x = 0.5, X in {0, 0.5}), where lambda_min is
    multiple for many candidates, those candidates are left to Jacobi: bit-equal rows."""
    from sdpcutsel_via_nn_amd import _capi, synthetic
    n, count = 40, 60000
    L = n * (n + 1) // 2
    Q_arr, vv, rng = synthetic.make_instance(n, 5 + k)
    S = np.full((count, 5), -1, dtype=np.int32)
    S[:, :k] = synthetic.random_index_sets(n, k, count, rng)
    ks = np.full(count, k, dtype=np.int32)
    X = np.triu(rng.integers(0, 2, (n, n)), 1) * 0.5
    X = X + X.T + 0.5 * np.eye(n)
    vertex = np.concatenate([X[np.triu_indices(n)], np.full(n, 0.5)])
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(5)
        sc.set_instance(n, Q_arr)
        sc.set_candidates(S, ks)
        ids = np.arange(count, dtype=np.int64)
        for point, structured in ((vv, False), (vertex, True)):
            sc.set_point(point)
            sc.score(_capi.NN)
            lam_j, coef_j, rhs_j, cols, _ = sc.cut_rows(ids)              # no eigenvalues at hand: Jacobi with vectors
            sc.score(_capi.EIG)
            lam_i, coef_i, rhs_i, cols_i, _ = sc.cut_rows(ids)            # lambda_min known: inverse iteration
            assert np.array_equal(cols, cols_i) and np.abs(lam_i - lam_j).max() <= 2e-13
            x, Xp = point[L:], point[:L]
            si = S[:, :k].astype(np.int64)
            M = np.zeros((count, k + 1, k + 1))
            M[:, 0, 0] = 1.0
            M[:, 0, 1:] = M[:, 1:, 0] = x[si]
            for a in range(k):
                for b in range(a, k):
                    M[:, 1 + a, 1 + b] = M[:, 1 + b, 1 + a] = Xp[n * si[:, a] - si[:, a] * (si[:, a] + 1) // 2 + si[:, b]]
            w = np.linalg.eigvalsh(M)
            gap = w[:, 1] - w[:, 0]
            m = k + k * (k + 1) // 2
            sep = gap >= 1e-6
            assert sep.sum() > (0 if structured else count // 2)
            assert np.abs(coef_i[sep, :m] - coef_j[sep, :m]).max() <= 1e-9 and np.abs(rhs_i[sep] - rhs_j[sep]).max() <= 1e-9
            # the cut of a unit vector v is  sum coef * value - rhs = v' M v  (cut_select_qp.py:745-750); v0^2 = -rhs
            vals = point[cols[:, :m].astype(np.int64)]
            for lam, coef, rhs in ((lam_i, coef_i, rhs_i), (lam_j, coef_j, rhs_j)):
                rq = (coef[:, :m] * vals).sum(axis=1) - rhs
                assert np.abs(rq - lam).max() <= 1e-11, np.abs(rq - lam).max()
                diag = [k + sum(k - t for t in range(a)) for a in range(k)]          # positions of the X_aa coefficients = v_a^2
                assert np.abs(coef[:, diag].sum(axis=1) - rhs - 1.0).max() <= 1e-12
            if structured:
                multiple = gap <= 1e-12
                assert multiple.sum() > (100 if k >= 4 else -1), (k, int(multiple.sum()))
                assert np.array_equal(coef_i[multiple], coef_j[multiple]) and np.array_equal(rhs_i[multiple], rhs_j[multiple])
    finally:
        sc.close()


def test_count_above_answers_from_the_head_what_the_whole_list_says(pkg):
    """RankList.count_above (the nb_opt_cuts counter of cut_select_qcqp.py:85-87) from the head of the list == the count over
    the completely ranked list: combined strategy in both regimes, optimality, feasibility, thresholds inside and outside the head."""
    from sdpcutsel_via_nn_amd import harness
    from sdpcutsel_via_nn_amd.cut_solver import CutSolver, AggArrays, RankList, _BIG_M
    Q_arr, vv, S, ks = _mixed_workload(40, (2, 3, 4), 30000, seed=9)
    n, L = 40, 820
    x = vv[L:]
    iu = np.triu_indices(n)
    psd = np.concatenate([np.minimum(x[iu[0]], x[iu[1]]), x])
    for point in (vv, 0.999 * psd + 0.001 * vv):
        for strat, sel in ((4, 300), (4, 5000), (2, 0), (1, 0)):
            cs = CutSolver()
            cs.set_instance(n, Q_arr, AggArrays(S, ks, n, Q_arr), 4, my_prob=harness.LinearRelaxation(np.zeros(L + n)))
            out = cs._sel_eigcut_by_ordering_on_measure(strat, point, 1, **({"sel_size": sel} if strat == 4 else {}))
            rl = out[1] if strat == 4 else out
            if not isinstance(rl, RankList):
                continue                                  # (an empty feasibility list)
            head_scores = np.array(rl._score[:rl._have])
            thresholds = [_BIG_M, 0.0, -_BIG_M, 2 * _BIG_M]
            if head_scores.size:
                thresholds += [float(np.median(head_scores)), float(head_scores.min()), float(head_scores.min()) - 1e-3]
            quick = [rl.count_above(t) for t in thresholds]
            full = rl.scores()                            # the complete ranking
            assert quick == [int(np.count_nonzero(full > t)) for t in thresholds], (strat, sel, quick)
