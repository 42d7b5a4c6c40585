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
    """Rows of the LP.  ``add`` takes the reference's per-row objects (cut_select_qp.py:754);
    ``add_csr`` takes a whole block of rows as CSR arrays (batched cut assembly, SURVEY 8 f
    row 4).  ``rows`` materialises SparsePair objects only if somebody asks for them."""

    def __init__(self):
        self._blocks, self.rhs, self.senses = [], [], []
        self._n = 0

    def add(self, lin_expr=(), rhs=(), senses=()):
        start = self._n
        lin_expr = list(lin_expr)
        self._blocks.append(("rows", lin_expr))
        self._n += len(lin_expr)
        self.rhs.extend(rhs)
        self.senses.extend(senses)
        return range(start, self._n)

    def add_csr(self, indptr, indices, values, rhs, sense):
        start, r = self._n, len(indptr) - 1
        self._blocks.append(("csr", np.asarray(indptr), np.asarray(indices), np.asarray(values)))
        self._n += r
        self.rhs.extend(np.asarray(rhs, dtype=np.float64).tolist())
        self.senses.extend([sense] * r)
        return range(start, self._n)

    def get_num(self):
        return self._n

    @property
    def rows(self):
        out = []
        for b in self._blocks:
            if b[0] == "rows":
                out.extend(b[1])
            else:
                _, ptr, ind, val = b
                out.extend(SparsePair(ind[ptr[r]:ptr[r + 1]].tolist(), val[ptr[r]:ptr[r + 1]].tolist())
                           for r in range(len(ptr) - 1))
        return out

    def csr_parts(self):
        """(data, cols, lengths) of all rows in order, as arrays."""
        data, cols, lens = [], [], []
        for b in self._blocks:
            if b[0] == "rows":
                for row in b[1]:
                    data.append(np.asarray(row.val, dtype=np.float64))
                    cols.append(np.asarray(row.ind, dtype=np.int64))
                    lens.append(len(row.ind))
            else:
                _, ptr, ind, val = b
                data.append(val)
                cols.append(ind)
                lens.extend(np.diff(ptr).tolist())
        cat = (lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dtype=dt))
        return cat(data, np.float64), cat(cols, np.int64), np.asarray(lens, dtype=np.int64)


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
        data, cols, lens = st.csr_parts()
        senses = np.asarray(st.senses)
        assert np.all(np.isin(senses, ["G", "L", "E"]))
        sign = np.where(senses == "G", -1.0, 1.0)
        rhs = np.asarray(st.rhs, dtype=np.float64) * sign
        ptr = np.concatenate([[0], np.cumsum(lens)])
        A = csr_matrix((data * np.repeat(sign, lens), cols, ptr), shape=(lens.shape[0], nv))
        eq = senses == "E"
        A_ub, b_ub = (A[~eq], rhs[~eq]) if (~eq).any() else (None, None)
        A_eq, b_eq = (A[eq], rhs[eq]) if eq.any() else (None, None)
        res = linprog(self.obj, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=b_eq, bounds=(0, 1), method="highs-ds")
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


def parse_osil(path):
    """QCQP instance in OSiL (cut_select_qcqp.py:115-312, data side only).  Returns
    dict(nb_vars, nb_lifted, c, Q_arr, adj, adj_cons, rows, rhs, senses): the objective's packed
    coefficient table and adjacency (:247-256), the adjacency of objective + all constraints, and
    the linearised constraints over [X packed | x] (:283-311; quadratic terms lifted to X_ij)."""
    import xml.etree.ElementTree as ET
    root = ET.parse(path).getroot()
    for el in root.iter():
        el.tag = el.tag.split("}", 1)[-1]
    data = root.find("instanceData")
    n = int(data.find("variables").attrib["numberOfVariables"])
    L = n * (n + 1) // 2
    obj = data.find("objectives/obj")
    c = np.zeros(n)
    if obj.attrib.get("maxOrMin") == "min":
        for co in obj.findall("coef"):
            c[int(co.attrib["idx"])] = float(co.text)

    def expand(parent, tag, as_int, n_out, start_rule=False):
        """<el mult= incr=>v</el> run-length lists of OSiL."""
        out = []
        for el in parent.find(tag).findall("el"):
            mult, incr = int(el.attrib.get("mult", 1)), int(el.attrib.get("incr", 0 if start_rule or not as_int else 1))
            v = int(el.text) if as_int else float(el.text)
            if "incr" not in el.attrib and as_int and not start_rule:
                incr = 1
            out.extend([v + k * incr for k in range(mult)] if as_int else [v] * mult)
        return out[:n_out] if n_out is not None else out

    cons = data.find("constraints")
    nb_cons = int(cons.attrib["numberOfConstraints"]) if cons is not None else 0
    sgn_rhs = []
    if cons is not None:
        for con in cons.findall("con"):
            lb, ub = con.attrib.get("lb"), con.attrib.get("ub")
            sgn_rhs.append(("E", float(lb)) if lb and ub else (("G", float(lb)) if lb else ("L", float(ub))))
    col_idx, col_val, starts = [], [], [0] * (nb_cons + 1)
    lin = data.find("linearConstraintCoefficients")
    if lin is not None and nb_cons:
        nvals = int(lin.attrib["numberOfValues"])
        col_idx = expand(lin, "colIdx", True, nvals)
        col_val = expand(lin, "value", False, nvals)
        starts = expand(lin, "start", True, nb_cons + 1, start_rule=True)
    qterms = []
    quad = data.find("quadraticCoefficients")
    if quad is not None:
        for q in quad.findall("qTerm"):
            qterms.append((int(q.attrib["idx"]), int(q.attrib["idxOne"]), int(q.attrib["idxTwo"]), float(q.attrib["coef"])))
    Q = np.zeros((n, n))
    adj = np.zeros((n, n), dtype=bool)
    adj_cons = np.zeros((n, n), dtype=bool)
    for k, i, j, v in qterms:
        if k == -1:
            adj[i, j] = adj[j, i] = True
            Q[i, j] = Q[j, i] = v
        adj_cons[i, j] = adj_cons[j, i] = True
    rows, rhs, senses = [], [], []
    for ci in range(nb_cons):
        ind = [n * i - i * (i + 1) // 2 + j for k, i, j, v in qterms if k == ci]
        val = [v for k, i, j, v in qterms if k == ci]
        for t in range(starts[ci], starts[ci + 1]):
            ind.append(int(col_idx[t]) + L)
            val.append(col_val[t])
        rows.append(SparsePair(ind, val))
        senses.append(sgn_rhs[ci][0])
        rhs.append(sgn_rhs[ci][1])
    return dict(nb_vars=n, nb_lifted=L, c=c, Q_arr=Q[np.triu_indices(n)], adj=adj, adj_cons=adj_cons, rows=rows, rhs=rhs,
                senses=senses)


def qcqp_covers(inst, dim, enumerate_cover):
    """Objective / constraint covers of cut_select_qcqp.py:314-334: the sub-problems of the
    objective+constraints graph that also belong to the objective-only cover, and the rest
    (both in the order of the objective+constraints enumeration).  -> two (set_inds, ks) pairs."""
    So, ko, _ = enumerate_cover(inst["adj"], dim)
    Sc, kc, _ = enumerate_cover(inst["adj_cons"], dim)
    in_obj = {tuple(int(v) for v in So[i, :ko[i]]) for i in range(ko.shape[0])}
    mask = np.array([tuple(int(v) for v in Sc[i, :kc[i]]) in in_obj for i in range(kc.shape[0])], dtype=bool)
    return (Sc[mask], kc[mask]), (Sc[~mask], kc[~mask])
