"""Triangle-inequality separation on the GPU (SURVEY 8 f row 3) against the reference's own
results (tests/golden/inst_tri.npz) and, at scale, against the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TAGS = ["spar020_100_1", "spar040_030_1"]


@pytest.fixture(scope="module")
def golden_tri():
    return np.load(os.path.join(GOLDEN, "inst_tri.npz"))


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("point", ["rnd", "mck"])
@pytest.mark.parametrize("sel", ["0p1", "0p5"])
def test_triangle_rows_match_reference(golden_tri, tag, point, sel):
    import sdpcutsel_via_nn_amd as pkg
    from sdpcutsel_via_nn_amd import harness
    g = golden_tri
    n = int(g[tag + "_nb_vars"])
    L = n * (n + 1) // 2
    cs = pkg.CutSolver()
    cs._sparse_pair = harness.SparsePair
    cs._nb_vars, cs._nb_lifted, cs._Q_arr, cs._Q_adj = n, L, np.zeros(L), g[tag + "_adj"]
    cs._my_prob = lp = harness.LinearRelaxation(np.zeros(L + n))
    cs._preprocess_triangle_ineq()
    assert np.array_equal(cs._gpu_tri_triples, g[tag + "_triples"])
    assert np.array_equal(np.repeat(cs._gpu_tri_density, 4).astype(float), g[tag + "_density"])
    q = "%s_%s_%s" % (tag, point, sel)
    nb = cs._separate_and_add_triangle(float(sel.replace("p", ".")), g[q + "_vars"])
    assert nb == int(g[q + "_nb"]) == lp.linear_constraints.get_num()
    ptr = g[q + "_row_ptr"]
    exact = point == "rnd"      # at the McCormick optimum thousands of violations tie exactly (order = entry id)
    for r, row in enumerate(lp.linear_constraints.rows):
        ref_ind = g[q + "_row_ind"][ptr[r]:ptr[r + 1]].tolist()
        assert row.ind == ref_ind, (r, row.ind, ref_ind)
        assert row.val == g[q + "_row_val"][ptr[r]:ptr[r + 1]].tolist()
    assert lp.linear_constraints.rhs == g[q + "_rhs"].tolist()
    assert exact or nb > 0


def test_triangle_separation_at_scale(oracle):
    """n = 125 dense: 317 750 triples, 1.27e6 inequalities; the device list equals the oracle's
    (bit-exact violations, same order) and the reference's cut-count rule is applied."""
    from sdpcutsel_via_nn_amd import _capi, harness, synthetic
    n = 125
    rng = np.random.default_rng(5)
    adj = rng.uniform(size=(n, n)) < 0.75
    adj = np.triu(adj, 1)
    adj = adj | adj.T
    Q_arr, vv, _ = synthetic.make_instance(n, seed=9)
    sc = _capi.Scorer(0)
    sc.set_instance(n, Q_arr)
    tri, dens = sc.tri_preprocess(adj)
    sc.set_point(vv)
    ent, vio, nv = sc.tri_separate(10000)
    t_ref, d_ref = oracle.preprocess_triangle_ineq(n, adj)
    assert np.array_equal(tri, t_ref) and np.array_equal(dens.astype(float), d_ref)
    # oracle violations vectorised in the reference's operation order
    L = n * (n + 1) // 2
    X, x = vv[:L], vv[L:]
    a, b, c = tri[:, 0].astype(np.int64), tri[:, 1].astype(np.int64), tri[:, 2].astype(np.int64)
    ra, rb = n * a - a * (a + 1) // 2, n * b - b * (b + 1) // 2
    X1, X2, X4 = X[ra + b], X[ra + c], X[rb + c]
    V = np.stack([X1 + X2 - X4 - x[a], X1 - X2 + X4 - x[b], -X1 + X2 + X4 - x[c],
                  -X1 - X2 - X4 + (((0.0 + x[a]) + x[b]) + x[c]) - 1], axis=1).ravel()
    D = np.repeat(d_ref, 4)
    keep = np.nonzero(V >= 1e-7)[0]
    order = keep[np.lexsort((keep, -V[keep], -D[keep]))]
    assert nv == keep.size
    assert np.array_equal(ent, order[:10000])
    assert np.array_equal(vio, V[order[:10000]])
    sc.close()
