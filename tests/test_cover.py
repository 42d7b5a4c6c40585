"""Vertex-cover enumeration (SURVEY 8 f row 1) vs the reference's own candidate lists (captured in
tests/golden/inst_boxqp.npz), the counts published in data_tables (nb_subproblems) and a
brute-force definition on random graphs."""
import itertools
import os

import numpy as np
import pytest

from conftest import BOXQP_TAGS, GOLDEN


def _adj_from_Q(n, Q_arr):
    A = np.zeros((n, n))
    A[np.triu_indices(n)] = Q_arr
    return (A + A.T) != 0


@pytest.mark.parametrize("tag", BOXQP_TAGS)
def test_cover_equals_reference_agg_list(golden_boxqp, tag):
    from sdpcutsel_via_nn_amd import _capi
    g = golden_boxqp
    n, dim = int(g[tag + "_nb_vars"]), int(tag[-1])
    S, ks, N = _capi.enumerate_cover(_adj_from_Q(n, g[tag + "_Q_arr"]), dim)
    assert N == g[tag + "_set_inds"].shape[0]
    assert np.array_equal(S, g[tag + "_set_inds"]) and np.array_equal(ks, g[tag + "_k"])     # same sets, same order


def test_published_subproblem_counts():
    """data_tables/data_(M+S^E_{3,4,5})_*_0.1_40.csv, column nb_subproblems."""
    from sdpcutsel_via_nn_amd import _capi, harness
    inst = harness.parse_boxqp(os.path.join(GOLDEN, "instances", "spar020-100-1.in"))
    assert [_capi.enumerate_cover(inst["adj"], d)[2] for d in (3, 4, 5)] == [1051, 4106, 11701]
    inst = harness.parse_boxqp(os.path.join(GOLDEN, "instances", "spar125-075-1.in"))
    assert _capi.enumerate_cover(inst["adj"], 3)[2] == 133242
    assert _capi.enumerate_cover(inst["adj"], 4)[2] == 1700215
    S, ks, N = _capi.enumerate_cover(inst["adj"], 5, max_subs=4 * 10 ** 6)      # the reference's RAM guard (:35, :117)
    assert N == 12845805 and S is None


def _brute(adj, dim):
    n = adj.shape[0]
    out = []
    def is_clique(c):
        return all(adj[a, b] for a, b in itertools.combinations(c, 2))
    def extendable(c):
        return any(v not in c and all(adj[v, u] for u in c) for v in range(n))
    cl = {s: [c for c in itertools.combinations(range(n), s) if is_clique(c)] for s in range(2, dim + 1)}
    keep = set(cl[dim]) | {c for s in range(2, dim) for c in cl[s] if not extendable(c)}
    # reference order: DFS over increasing ids, a clique is listed where its last vertex is reached
    def rec(c):
        if len(c) == dim:
            out.append(c); return
        ext = [v for v in range(c[-1] + 1, n) if all(adj[v, u] for u in c)]
        for v in ext:
            rec(c + (v,))
        if not ext and c in keep:
            out.append(c)
    for i in range(n):
        for j in range(i + 1, n):
            if adj[i, j]:
                rec((i, j))
    return out


@pytest.mark.parametrize("seed,n,p,dim", [(0, 12, 0.5, 3), (1, 14, 0.6, 4), (2, 13, 0.7, 5), (3, 70, 0.08, 4), (4, 9, 1.0, 5)])
def test_cover_matches_brute_force(seed, n, p, dim):
    from sdpcutsel_via_nn_amd import _capi
    rng = np.random.default_rng(seed)
    U = rng.uniform(size=(n, n)) < p
    adj = np.triu(U, 1)
    adj = adj | adj.T
    S, ks, N = _capi.enumerate_cover(adj, dim)
    ref = _brute(adj, dim)
    assert N == len(ref)
    assert [tuple(int(v) for v in S[i, :ks[i]]) for i in range(N)] == ref
    assert np.all(S[np.arange(5)[None, :] >= ks[:, None]] == -1) if N else True


def test_cover_argument_errors():
    from sdpcutsel_via_nn_amd import _capi
    with pytest.raises(ValueError):
        _capi.enumerate_cover(np.ones((4, 4)), 2)
    with pytest.raises(ValueError):
        _capi.enumerate_cover(np.ones((4, 5)), 3)
    S, ks, N = _capi.enumerate_cover(np.zeros((6, 6)), 3)
    assert N == 0 and S.shape == (0, 5)


def test_osil_parser_and_qcqp_covers(golden_qcqp):
    """OSiL data side (cut_select_qcqp.py:229-259) and the two covers of :314-334 against what the
    reference itself produced on q_20_4_25_1 (tests/golden/inst_qcqp.npz)."""
    from sdpcutsel_via_nn_amd import _capi, harness
    g = golden_qcqp
    inst = harness.parse_osil(os.path.join(GOLDEN, "instances", "q_20_4_25_1.osil"))
    assert inst["nb_vars"] == int(g["nb_vars"]) and np.array_equal(inst["Q_arr"], g["Q_arr"])
    assert inst["senses"] == ["L"] * 4 and inst["rhs"] == [49.695, 18.723, 56.565, 28.634]
    assert [len(r.ind) for r in inst["rows"]] == [52 + 20, 52 + 20, 52 + 20, 52 + 20]
    (So, ko), (Sc, kc) = harness.qcqp_covers(inst, 3, _capi.enumerate_cover)
    assert np.array_equal(So, g["obj_set_inds"]) and np.array_equal(ko, g["obj_k"])
    assert np.array_equal(Sc, g["cons_set_inds"]) and np.array_equal(kc, g["cons_k"])
    lp = harness.LinearRelaxation(np.concatenate([inst["Q_arr"], inst["c"]]))
    lp.linear_constraints.add(inst["rows"], inst["rhs"], inst["senses"])
    lp.linear_constraints.add(lin_expr=[harness.SparsePair([210], [1.0])], rhs=[0.5], senses=["E"])   # equality rows
    lp.solve()
    assert abs(lp.get_values()[210] - 0.5) < 1e-9
