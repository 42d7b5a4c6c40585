"""GPU tests of the round-2 additions (run with -m gpu on an MI355X): the reference's own FFI on the
product library, built-in networks, the on-device Philox candidate generator vs its numpy twin,
the on-device vertex cover vs the host enumeration, the random strategy and the other host branches
of the loop (strong_only, foreign entry lists, triangle_on, term_on), and the composition of the
mixin with reference-shaped classes."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import sdpcutsel_via_nn_amd as p
    return p


# ----------------------------------------------------------------------------- boundary
@pytest.mark.parametrize("d", [2, 3, 4, 5])
def test_reference_ffi_on_the_product_library(pkg, d):
    """Exactly the reference's binding (cut_select_qp.py:297-303, :579-582), pointed at
    libsdpcut_nns.so (the drop-in the package builds next to libsdpcut_hip.so) instead of neural_nets/NNs.so; values against
    the real NNs.so's (goldens)."""
    from sdpcutsel_via_nn_amd import _capi
    _capi.load_library()      # (a process that also uses PyTorch-ROCm must import torch before the library binds a HIP runtime: INTEGRATION.md section 2)
    nn_library = ctypes.cdll.LoadLibrary(_capi.NNS_LIB_PATH)
    func_dim = getattr(nn_library, "neural_net_%dD" % d)
    func_dim.restype = ctypes.c_double
    input_arr = (ctypes.c_double * (d * (d + 3) // 2))()
    g = golden_nn(d)
    nn_library.NNs_initialize()
    for i in list(range(200)) + [4095]:
        input_arr[:] = g["inputs"][i]
        y = func_dim(input_arr)
        assert abs(y - g["nn_out"][i]) <= 1e-12 * max(1.0, abs(g["nn_out"][i])), (d, i)
    nn_library.NNs_terminate()
    input_arr[:] = g["inputs"][7]
    assert abs(func_dim(input_arr) - g["nn_out"][7]) <= 1e-12            # lazy re-initialisation
    nn_library.NNs_terminate()


def test_builtin_networks_equal_uploaded_ones(pkg):
    from sdpcutsel_via_nn_amd import _capi, networks, synthetic
    wl = synthetic.make_workload(nb_vars=40, k=5, count=3000, seed=3)
    res = []
    for builtin in (False, True):
        sc = pkg.Scorer(0)
        if builtin:
            sc.set_builtin_networks(5)
        else:
            sc.set_network(5, *networks.load_network(5))
        sc.set_instance(40, wl["Q_arr"])
        sc.set_candidates(wl["set_inds"], wl["ks"])
        sc.set_point(wl["vars_values"])
        sc.score(_capi.NN)
        res.append(sc.get_scores(eig=False)[1])
        sc.close()
    assert np.array_equal(res[0], res[1])
    sc = pkg.Scorer(0)
    sc.set_builtin_networks(3)
    with pytest.raises(ValueError):
        sc.set_builtin_networks(6)
    sc.close()


def test_mixin_between_reference_shaped_classes(pkg):
    """The MRO recipe of make_dropin_classes on stand-ins that call the hot path the way the
    reference's two loops do (plain `self.` calls in the QP class, `super().` calls inside the QCQP
    class): every call must land on the GPU mixin, none on the CPU methods of the base."""
    from sdpcutsel_via_nn_amd import harness, synthetic

    class RefQP(object):
        _BIG_M = 1000
        def __init__(self):
            self._agg_list, self._dim = [], 0
        def _load_neural_nets(self): raise AssertionError("CPU loader reached")
        def _sel_eigcut_by_ordering_on_measure(self, *a, **k): raise AssertionError("CPU selection reached")
        def _gen_eigcuts_selected(self, *a, **k): raise AssertionError("CPU generation reached")
        def _get_eigendecomp(self, *a, **k): raise AssertionError("CPU eigen-decomposition reached")
        def round_qp(self, strat, vv, sel):
            rl = self._sel_eigcut_by_ordering_on_measure(strat, vv, 1)
            return self._gen_eigcuts_selected(strat, sel, rl, vars_values=vv)

    class RefQCQP(RefQP):
        def round_qcqp(self, vv, sel, cons):
            super()._load_neural_nets()                                              # cut_select_qcqp.py:41
            a = super()._sel_eigcut_by_ordering_on_measure(2, vv, 1)                 # :73-74
            obj_list, self._agg_list = self._agg_list, cons
            b = super()._sel_eigcut_by_ordering_on_measure(1, vv, 1)                 # :76
            self._agg_list = obj_list
            rl = (a + b)[0:sel]                                                      # :79
            n_comb = sum(isinstance(e[0], int) for e in rl)                          # :90-92
            return (self._gen_eigcuts_selected(1, sel - n_comb, b[0:sel - n_comb], vars_values=vv) +
                    self._gen_eigcuts_selected(2, n_comb, a[0:n_comb], vars_values=vv))

    class qp_mod: CutSolver = RefQP
    class qcqp_mod: CutSolverQCQP = RefQCQP
    G, GQ = pkg.make_dropin_classes(qp_mod, qcqp_mod)
    wl = synthetic.make_workload(nb_vars=30, k=3, count=400, seed=5)
    n, L = 30, 465
    sets = [[int(v) for v in s[:3]] for s in wl["set_inds"]]
    agg = [(s, [n * s[a] - s[a] * (s[a] + 1) // 2 + s[b] for a in range(3) for b in range(a, 3)], None, None) for s in sets]
    for cls, run in ((G, lambda o: o.round_qp(2, wl["vars_values"], 40)),
                     (GQ, lambda o: o.round_qcqp(wl["vars_values"], 60, agg[300:]))):
        o = cls()
        o._sparse_pair = harness.SparsePair
        o._nb_vars, o._nb_lifted, o._Q_arr, o._dim = n, L, wl["Q_arr"], 3
        o._agg_list = agg[:300] if cls is GQ else agg
        o._my_prob = harness.LinearRelaxation(np.zeros(L + n))
        if cls is G:
            o._load_neural_nets()
        nb = run(o)
        assert nb == o._my_prob.linear_constraints.get_num() and nb > 0


# ----------------------------------------------------------------------------- Philox on device
@pytest.mark.parametrize("nv,k,count,first", [(1000, 3, 1250000, 5 * 1250000), (100, 2, 50000, 0), (30, 5, 70000, 2 ** 33),
                                             (64, 4, 100001, 123)])
def test_philox_device_equals_twin(pkg, nv, k, count, first):
    from sdpcutsel_via_nn_amd import _capi, synthetic
    Q_arr, vv, _ = synthetic.make_instance(nv, 7)
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(5)
        sc.set_instance(nv, Q_arr)
        sc.set_candidates_philox(k, count, seed=7, first_id=first)
        pick = np.unique(np.concatenate([np.arange(min(count, 10 ** 6)), np.arange(0, count, 100), [count - 1]]))
        S, ks = sc.get_candidates(pick)
        twin = synthetic.philox_index_sets(nv, k, first + pick, seed=7)
        assert np.array_equal(S, twin) and np.all(ks == k)
        # the generated list scores like the same list uploaded from the host (SoA buckets, global ids)
        sc.set_point(vv)
        sc.score(_capi.EIG | _capi.NN)
        eig, obj = sc.get_scores()
        m = min(count, 200000)
        sc2 = pkg.Scorer(0)
        sc2.set_builtin_networks(5)
        sc2.set_instance(nv, Q_arr)
        sc2.set_candidates(synthetic.philox_index_sets(nv, k, first + np.arange(m), seed=7), np.full(m, k, np.int32), global_base=first)
        sc2.set_point(vv)
        sc2.score(_capi.EIG | _capi.NN)
        eig2, obj2 = sc2.get_scores()
        assert np.array_equal(eig[:m], eig2) and np.array_equal(obj[:m], obj2)
        a = sc.select_round(4, min(5000, m))
        if count == m:
            b = sc2.select_round(4, min(5000, m))
            assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["coef"], b["coef"])
        assert a["idx"].min() >= first and a["idx"].max() < first + count
        sc2.close()
    finally:
        sc.close()


# ----------------------------------------------------------------------------- cover on device
def _adjacency(name):
    from sdpcutsel_via_nn_amd import harness
    if name.endswith(".osil"):
        inst = harness.parse_osil(os.path.join(GOLDEN, "instances", name))
        return inst, inst["adj_cons"]
    inst = harness.parse_boxqp(os.path.join(GOLDEN, "instances", name))
    return inst, inst["adj"]


@pytest.mark.parametrize("name,dim", [("spar020-100-1.in", 3), ("spar020-100-1.in", 4), ("spar020-100-1.in", 5),
                                      ("spar040-030-1.in", 3), ("spar040-030-1.in", 4), ("spar040-030-1.in", 5),
                                      ("q_20_4_25_1.osil", 3), ("q_50_10_25_1.osil", 5), ("spar125-075-1.in", 3),
                                      ("spar125-075-1.in", 4)])
def test_device_cover_equals_host_enumeration(pkg, name, dim):
    """sdpcut_set_candidates_cover: same sets in the same order as the host enumerator (itself pinned
    to the reference's lists and published counts, tests/test_cover.py), and the list it leaves in
    the handle scores exactly like the uploaded one."""
    from sdpcutsel_via_nn_amd import _capi, harness
    inst, adj = _adjacency(name)
    n = inst["nb_vars"]
    S, ks, N = _capi.enumerate_cover(adj, dim)
    sc = pkg.Scorer(0)
    try:
        sc.set_builtin_networks(5)
        sc.set_instance(n, inst["Q_arr"])
        assert sc.set_candidates_cover(adj, dim, max_subs=N) == N and sc.N == 0     # the guard: count only
        assert sc.set_candidates_cover(adj, dim) == N and sc.N == N
        S2, ks2 = sc.get_candidates(np.arange(N))
        assert np.array_equal(ks2, ks) and np.array_equal(S2, S)
        vv = harness.random_mccormick_point(n, np.random.default_rng(4))
        sc.set_point(vv)
        sc.score(_capi.EIG | _capi.NN)
        eig, obj = sc.get_scores()
        sc2 = pkg.Scorer(0)
        sc2.set_builtin_networks(5)
        sc2.set_instance(n, inst["Q_arr"])
        sc2.set_candidates(S, ks)
        sc2.set_point(vv)
        sc2.score(_capi.EIG | _capi.NN)
        eig2, obj2 = sc2.get_scores()
        sc2.close()
        assert np.array_equal(eig, eig2) and np.array_equal(obj, obj2)
    finally:
        sc.close()


def test_device_cover_random_graphs(pkg):
    """random sparsity patterns, all bitset widths (n <= 64, 128, 256, 1024), all dims"""
    from sdpcutsel_via_nn_amd import _capi
    rng = np.random.default_rng(11)
    for n, dens in ((7, 0.9), (33, 0.5), (64, 0.35), (65, 0.3), (128, 0.2), (130, 0.12), (300, 0.04), (70, 0.0), (12, 1.0)):
        A = rng.uniform(size=(n, n)) < dens
        A = np.triu(A, 1)
        A = A | A.T
        for dim in (3, 4, 5):
            S, ks, N = _capi.enumerate_cover(A, dim)
            sc = pkg.Scorer(0)
            sc.set_instance(n, np.zeros(n * (n + 1) // 2))
            assert sc.set_candidates_cover(A, dim) == N, (n, dens, dim)
            if N:
                S2, ks2 = sc.get_candidates(np.arange(N))
                assert np.array_equal(ks2, ks) and np.array_equal(S2, S), (n, dens, dim)
            sc.close()


# ----------------------------------------------------------------------------- host branches of the loop
def _spar020(pkg, dim=3):
    from sdpcutsel_via_nn_amd import _capi, harness
    from sdpcutsel_via_nn_amd.cut_solver import AggArrays
    inst = harness.parse_boxqp(os.path.join(GOLDEN, "instances", "spar020-100-1.in"))
    S, ks, N = _capi.enumerate_cover(inst["adj"], dim)
    cs = pkg.CutSolver()
    cs._sparse_pair = harness.SparsePair
    lp = harness.LinearRelaxation(np.zeros(inst["nb_lifted"] + inst["nb_vars"]))
    cs.set_instance(inst["nb_vars"], inst["Q_arr"], AggArrays(S, ks, inst["nb_vars"], inst["Q_arr"]), dim=dim, my_prob=lp)
    return cs, inst, lp, harness.random_mccormick_point(inst["nb_vars"], np.random.default_rng(9))


@pytest.mark.parametrize("as_list", [False, True])
def test_random_strategy_selection_and_generation(pkg, oracle, as_list):
    """Strategy 5 (cut_select_qp.py:634-637, :729-732): in-place shuffle with numpy's global
    generator, rows of the first sel_size entries of the shuffled list; then a ranking strategy on
    the SAME (now reordered) list must see the new order (stale device twin dropped)."""
    cs, inst, lp, vv = _spar020(pkg)
    n, L = inst["nb_vars"], inst["nb_lifted"]
    if as_list:
        cs._agg_list = [cs._agg_list[i] for i in range(len(cs._agg_list))]
    before = [list(cs._agg_list[i][0]) for i in range(len(cs._agg_list))]
    rl2 = cs._sel_eigcut_by_ordering_on_measure(2, vv, 1)
    first_best = before[rl2[0][0]]
    np.random.seed(7)
    rl = cs._sel_eigcut_by_ordering_on_measure(5, vv, 1)
    assert rl is cs._agg_list
    np.random.seed(7)
    perm = np.random.permutation(len(before))
    after = [list(cs._agg_list[i][0]) for i in range(len(before))]
    assert after == [before[p] for p in perm]
    nb = cs._gen_eigcuts_selected(5, 105, rl, vars_values=vv)
    agg = oracle.build_agg_list(after, n, list(inst["Q_arr"]))
    ref_nb, ref_rows, ref_rhs, _ = oracle.gen_eigcuts_selected(agg, L, 5, 105, agg, vars_values=vv)
    assert nb == ref_nb == lp.linear_constraints.get_num()
    for row, (ind, val), rhs, rr in zip(lp.linear_constraints.rows, ref_rows, lp.linear_constraints.rhs, ref_rhs):
        assert row.ind == list(ind) and np.abs(np.array(row.val) - np.array(val, dtype=float)).max() <= 1e-9
        assert abs(rhs - rr) <= 1e-9
    rl2b = cs._sel_eigcut_by_ordering_on_measure(2, vv, 2)
    assert after[rl2b[0][0]] == first_best and abs(rl2b[0][1] - rl2[0][1]) == 0.0


def test_strong_only_and_foreign_entry_lists(pkg, oracle):
    """strong_only (cut_select_qp.py:725-726) stops at the first non-positive optimality score, through
    the lazy rank list and through a plain list of entry tuples; entry lists that carry no candidate
    index (what the random strategy hands over in the reference) go through _gen_from_entries."""
    from sdpcutsel_via_nn_amd import harness
    cs, inst, lp, vv = _spar020(pkg)
    n, L = inst["nb_vars"], inst["nb_lifted"]
    sets = [list(cs._agg_list[i][0]) for i in range(len(cs._agg_list))]
    agg = oracle.build_agg_list(sets, n, list(inst["Q_arr"]))
    ref_rl = oracle.sel_eigcut_by_ordering_on_measure(agg, L, 2, vv)
    n_pos = next(p for p, e in enumerate(ref_rl) if e[1] <= 0)
    assert 0 < n_pos < len(ref_rl)
    for sel in (n_pos + 40, max(n_pos - 3, 1)):
        ref_nb, ref_rows, ref_rhs, _ = oracle.gen_eigcuts_selected(agg, L, 2, sel, ref_rl, strong_only=True)
        for plain in (False, True):
            lp2 = harness.LinearRelaxation(np.zeros(L + n))
            cs._my_prob = lp2
            rl = cs._sel_eigcut_by_ordering_on_measure(2, vv, 1)
            nb = cs._gen_eigcuts_selected(2, sel, list(rl[0:sel]) if plain else rl, strong_only=True, vars_values=vv)
            assert nb == ref_nb == lp2.linear_constraints.get_num(), (sel, plain)
            for row, (ind, val) in zip(lp2.linear_constraints.rows, ref_rows):
                assert row.ind == list(ind) and np.abs(np.array(row.val) - np.array(val, dtype=float)).max() <= 1e-9
    # entries without a candidate index: (set_inds, Xarr_inds, Q_slice, max_elem) records
    foreign = [agg[i] for i in (5, 900, 17, 333, 4, 1000)] + [([1, 2], [n * 1 - 1 + 1, n * 1 - 1 + 2, 2 * n - 3 + 2], None, None)]
    lp3 = harness.LinearRelaxation(np.zeros(L + n))
    cs._my_prob = lp3
    nb = cs._gen_eigcuts_selected(5, len(foreign), foreign, vars_values=vv)
    ref_nb, ref_rows, ref_rhs, _ = oracle.gen_eigcuts_selected(agg, L, 5, len(foreign), foreign, vars_values=vv)
    assert nb == ref_nb == lp3.linear_constraints.get_num() and nb > 0
    for row, (ind, val), rhs, rr in zip(lp3.linear_constraints.rows, ref_rows, lp3.linear_constraints.rhs, ref_rhs):
        assert row.ind == list(ind) and np.abs(np.array(row.val) - np.array(val, dtype=float)).max() <= 1e-9
        assert abs(rhs - rr) <= 1e-9


@pytest.mark.parametrize("strat", [5, 2])
def test_loop_with_triangles_termination_and_random_strategy(pkg, strat):
    """cut_select_algo with triangle_on / term_on / strategy 5 / strong_only (all previously untested):
    the bound only tightens, triangle cuts are added every round, term_on stops early."""
    path = os.path.join(GOLDEN, "instances", "spar020-100-1.in")
    np.random.seed(7)
    cs = pkg.CutSolver()
    bounds, t_total, rt, st, cuts, tri, nsub = cs.cut_select_algo(path, 3, 0.1, strat=strat, nb_rounds_cuts=4,
                                                                  triangle_on=True, strong_only=(strat == 2))
    assert nsub == 1051 and len(bounds) == 5 and len(cuts) == 5 and len(tri) == 4
    assert all(b >= a - 1e-7 for a, b in zip(bounds[1:], bounds)) and bounds[-1] < bounds[0] - 1.0   # (maximisation bound: decreasing)
    assert tri[0] > 0 and all(t >= 0 for t in tri) and cuts[0] == 0 and all(0 <= c <= 105 for c in cuts[1:])
    cs2 = pkg.CutSolver()
    b2 = cs2.cut_select_algo(path, 3, 0.1, strat=2, nb_rounds_cuts=40, term_on=True)[0]
    assert 4 <= len(b2) < 41
    last = [(b2[i - 1] - b2[i]) / (b2[0] - b2[i]) for i in range(2, len(b2))]
    assert last[-1] < 1e-3 and all(v >= 1e-3 for v in last[:-1])
    assert cs2.cut_select_algo(path, 3, 0.1, strat=2, nb_rounds_cuts=0)[-1] == 1051       # count only
    assert cs2.cut_select_algo(path, 5, 0.1, strat=2, nb_rounds_cuts=3, max_subs=11000)[0] == [0, 0]     # the RAM guard (11701 >= 11000)
    with pytest.raises(AssertionError):
        cs2.cut_select_algo(path, 3, 0.1, strat=3)


def test_qcqp_loop_random_strategy(pkg):
    np.random.seed(3)
    cs = pkg.CutSolverQCQP()
    objs, sel, cuts, opt = cs.cut_select_algo(os.path.join(GOLDEN, "instances", "q_20_4_25_1.osil"), 3, sel_size=0.5, strat=5,
                                              nb_rounds_cuts=3)
    assert sel == 9 and len(objs) == 4 and len(cuts) == 4 and opt == [0, 0, 0, 0]
    assert all(b >= a - 1e-9 for a, b in zip(objs, objs[1:]))


# ----------------------------------------------------------------------------- handle reuse (capi.hip:504)
def test_handle_rebound_to_a_larger_instance_after_a_fused_round(pkg, oracle):
    """Fused round on n = 20, the SAME handle re-bound to n = 100 (the pinned staging block of the LP point
    grows), fused rounds again: results against the oracle both times.  Round 2's sdpcut_set_point freed the
    completion ticket of the round epilogue when the staging block grew and kept the dangling pointer."""
    from sdpcutsel_via_nn_amd import _capi, networks, synthetic
    sc = pkg.Scorer(0)
    try:
        sc.set_network(3, *networks.load_network(3))
        for nb_vars, count, seed in ((20, 3000, 11), (100, 40000, 12), (20, 3000, 13), (100, 40000, 14)):
            wl = synthetic.make_workload(nb_vars=nb_vars, k=3, count=count, seed=seed)
            L = nb_vars * (nb_vars + 1) // 2
            sc.set_instance(nb_vars, wl["Q_arr"])
            sc.set_candidates(wl["set_inds"], wl["ks"])
            vv = wl["vars_values"]
            si = wl["set_inds"][:, :3]
            obj = oracle.opt_score_batch(3, si, nb_vars, vv, wl["Q_arr"])
            eig = oracle.eigmin_batch(3, vv[L:][si], vv[:L][oracle.triu_positions(si, nb_vars)])
            for strat in (4, 1, 2):
                for rep in range(3):                   # several epilogues on the (possibly re-allocated) ticket
                    r = sc.select_round(strat, 500, point=vv)
                sc.score(_capi.EIG | _capi.NN)
                d_eig, d_obj = sc.get_scores()
                assert np.abs(d_eig - eig).max() <= 2e-13
                order, ref_score, ref_strat, _ = oracle.rank_arrays(strat, d_obj, d_eig, 500)
                w = min(500, order.shape[0])
                assert np.array_equal(r["idx"], order[:w]), (nb_vars, strat)
                assert np.array_equal(r["score"], ref_score[:w] + 0.0) and r["new_strat"] == ref_strat
                lam, coef, rhs, cols, ks = sc.cut_rows(order[:w])
                # (bit-equal when both sides had lambda_min at hand or both had not; after an optimality round the fused rows
                # come from Jacobi, the explicit ones -- the eigenvalues now being scored -- from inverse iteration)
                cw = coef[:, :r["coef"].shape[1]]
                if strat == 2:
                    assert np.abs(r["lam"] - lam).max() <= 2e-13 and np.abs(r["coef"] - cw).max() <= 1e-9
                else:
                    assert np.array_equal(r["lam"], lam) and np.array_equal(r["coef"], cw)
        assert sc.get_stat(_capi.STAT_SELECT_FALLBACKS) == 0
    finally:
        sc.close()
