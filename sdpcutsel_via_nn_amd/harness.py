"""Round harness without CPLEX (SURVEY.md section 8 f, row 2).

The cut-scoring hot path needs (a) the packed objective table ``Q_arr`` and adjacency of an
instance, and (b) an LP point ``vars_values = [X packed | x]``.  The reference obtains both
through CPLEX (cut_select_qp.py:305-350 parser, :352-375 McCormick rows, :134-137 solve).
This module restates only the data side so that real instances can be driven end to end
with scipy's HiGHS:

* :func:`parse_boxqp`      -- ``.in`` file -> Q_arr / adjacency / linear term  (cut_select_qp.py:309-328)
* :func:`mccormick_rows`   -- 2 diagonal + 3 off-diagonal RLT rows per edge   (cut_select_qp.py:352-375)
* :class:`LinearRelaxation` -- row store with the ``linear_constraints.add(lin_expr=, rhs=, senses=)``
  surface that ``_gen_eigcuts_selected`` appends to (cut_select_qp.py:754), solved with HiGHS.

Nothing here is on the GPU path; it is plumbing either side of it.
"""
import numpy as np


class SparsePair(object):
    """Stand-alone twin of ``cplex.SparsePair(ind=, val=)`` (used at cut_select_qp.py:747)."""
    __slots__ = ("ind", "val")

    def __init__(self, ind=None, val=None):
        self.ind = list(ind) if ind is not None else []
        self.val = list(val) if val is not None else []


class _RowStore(object):
    def __init__(self):
        self.rows, self.rhs, self.senses = [], [], []

    def add(self, lin_expr=(), rhs=(), senses=()):
        start = len(self.rows)
        self.rows.extend(lin_expr)
        self.rhs.extend(rhs)
        self.senses.extend(senses)
        return range(start, len(self.rows))

    def get_num(self):
        return len(self.rows)


class LinearRelaxation(object):
    """Minimal LP model: min c^T v, 0 <= v <= 1, rows added through
    ``self.linear_constraints.add`` exactly as the reference does on its CPLEX object."""

    def __init__(self, obj_coeffs):
        self.obj = np.asarray(obj_coeffs, dtype=np.float64)
        self.linear_constraints = _RowStore()
        self._values = None
        self._objval = None

    def solve(self):
        from scipy.optimize import linprog
        from scipy.sparse import csr_matrix
        st = self.linear_constraints
        nv = self.obj.shape[0]
        data, cols, ptr = [], [], [0]
        sign = []
        for row, sense in zip(st.rows, st.senses):
            s = -1.0 if sense == "G" else 1.0
            assert sense in ("G", "L"), "harness supports inequality rows only"
            data.extend(s * np.asarray(row.val, dtype=np.float64))
            cols.extend(row.ind)
            ptr.append(len(cols))
            sign.append(s)
        A = csr_matrix((data, cols, ptr), shape=(len(st.rows), nv))
        b = np.asarray(st.rhs, dtype=np.float64) * np.asarray(sign)
        res = linprog(self.obj, A_ub=A, b_ub=b, bounds=(0, 1), method="highs-ds")
        if res.status != 0:
            raise RuntimeError("HiGHS: " + res.message)
        self._values, self._objval = res.x, res.fun
        return res

    def get_values(self):
        return self._values

    def get_objective_value(self):
        return self._objval


def parse_boxqp(path):
    """BoxQP ``.in`` -> dict(nb_vars, c, Q_arr, adj).  Signs flipped because the model
    minimises; off-diagonal of Q_arr = -q_ij, diagonal = -q_ii/2 (cut_select_qp.py:313-321)."""
    with open(path) as f:
        lines = f.read().split("\n")
    n = int(lines[0].split()[0])
    c = -np.array([int(t) for t in lines[1].split()], dtype=np.float64)
    Q = -np.array([[float(t) for t in lines[2 + r].split()] for r in range(n)])
    assert Q.shape == (n, n)
    half_diag = Q.copy()
    half_diag[np.diag_indices(n)] /= 2.0
    Q_arr = half_diag[np.triu_indices(n)]
    adj = (Q != 0)
    return dict(nb_vars=n, nb_lifted=n * (n + 1) // 2, c=c, Q_arr=Q_arr, adj=adj)


def mccormick_rows(nb_vars, adj):
    """RLT rows over variables [X packed | x] (cut_select_qp.py:352-375)."""
    L = nb_vars * (nb_vars + 1) // 2
    rows, rhs, senses = [], [], []
    for i in range(nb_vars):
        Xii, xi = nb_vars * i - i * (i - 1) // 2, L + i
        rows += [SparsePair([Xii, xi], [1, -1]), SparsePair([Xii, xi], [-1, 2])]
        rhs += [0, 1]
        senses += ["L", "L"]
        for j in range(i + 1, nb_vars):
            if adj[i, j]:
                Xij, xj = Xii + j - i, xi + j - i
                rows += [SparsePair([Xij, xi, xj], [-1, 1, 1]),
                         SparsePair([Xij, xi], [1, -1]),
                         SparsePair([Xij, xj], [1, -1])]
                rhs += [1, 0, 0]
                senses += ["L", "L", "L"]
    return rows, rhs, senses


def boxqp_relaxation(inst):
    """McCormick relaxation M of a parsed BoxQP instance, ready to solve."""
    lp = LinearRelaxation(np.concatenate([inst["Q_arr"], inst["c"]]))
    lp.linear_constraints.add(*mccormick_rows(inst["nb_vars"], inst["adj"]))
    return lp


def random_mccormick_point(nb_vars, rng):
    """Synthetic McCormick-feasible LP point (SURVEY.md section 8 d, C2 generator):
    x ~ U(0,1); X_ij ~ U(max(0, x_i + x_j - 1), min(x_i, x_j)); X_ii ~ U(max(0, 2x_i - 1), x_i).
    Returns vars_values = [X packed | x]."""
    x = rng.uniform(0.0, 1.0, nb_vars)
    iu = np.triu_indices(nb_vars)
    lo = np.maximum(0.0, x[iu[0]] + x[iu[1]] - 1.0)
    hi = np.minimum(x[iu[0]], x[iu[1]])
    X = lo + (hi - lo) * rng.uniform(0.0, 1.0, lo.shape[0])
    return np.concatenate([X, x])
