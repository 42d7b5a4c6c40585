"""Oracle restatement of the triangle-inequality separation (cut_select_qp.py:799-863) against
the reference's own results (tests/golden/inst_tri.npz, made by make_golden.py `tri`)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

TAGS = ["spar020_100_1", "spar040_030_1"]


@pytest.fixture(scope="module")
def golden_tri():
    return np.load(os.path.join(GOLDEN, "inst_tri.npz"))


@pytest.mark.parametrize("tag", TAGS)
def test_preprocess(oracle, golden_tri, tag):
    g = golden_tri
    triples, dens = oracle.preprocess_triangle_ineq(int(g[tag + "_nb_vars"]), g[tag + "_adj"])
    assert np.array_equal(triples, g[tag + "_triples"])
    assert np.array_equal(np.repeat(dens, 4), g[tag + "_density"])


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("point", ["rnd", "mck"])
@pytest.mark.parametrize("sel", ["0p1", "0p5"])
def test_separation(oracle, golden_tri, tag, point, sel):
    g = golden_tri
    n = int(g[tag + "_nb_vars"])
    triples, dens = oracle.preprocess_triangle_ineq(n, g[tag + "_adj"])
    q = "%s_%s_%s" % (tag, point, sel)
    nb, viol, order, rows, rhs = oracle.separate_triangle(n, triples, dens, float(sel.replace("p", ".")), g[q + "_vars"])
    assert nb == int(g[q + "_nb"])
    assert np.array_equal(viol.ravel(), g[q + "_viol"])
    ptr = np.cumsum([0] + [len(r[0]) for r in rows])
    assert np.array_equal(ptr, g[q + "_row_ptr"])
    if rows:
        assert np.array_equal(np.concatenate([r[0] for r in rows]), g[q + "_row_ind"])
        assert np.array_equal(np.concatenate([r[1] for r in rows]).astype(float), g[q + "_row_val"])
    assert np.array_equal(np.array(rhs, dtype=float), g[q + "_rhs"])
