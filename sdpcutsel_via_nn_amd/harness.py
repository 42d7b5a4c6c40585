"""Round harness without CPLEX (SURVEY.md section 8 f, row 2).

The cut-scoring hot path needs (a) the packed objective table ``Q_arr`` and adjacency of an
instance, and (b) an LP point ``vars_values = [X packed | x]``.  The reference obtains both
through CPLEX (cut_select_qp.py:305-350 parser, :352-375 McCormick rows, :134-137 solve).
This module restates only the data side so that real instances can be driven end to end
with scipy's HiGHS:

* :func:`parse_boxqp`      -- ``.in`` file -> Q_arr / adjacency / linear term  (cut_select_qp.py:309-328)
* :func:`mccormick_csr`    -- the McCormick / RLT rows of cut_select_qp.py:352-375 as one CSR block
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
        self._blocks = []
        self._rhs, self._senses = [], []       # per block: list / array of right-hand sides; list of senses or (sense, count)
        self._n = 0

    def add(self, lin_expr=(), rhs=(), senses=()):
        start = self._n
        lin_expr = list(lin_expr)
        self._blocks.append(("rows", lin_expr))
        self._n += len(lin_expr)
        self._rhs.append(list(rhs))
        self._senses.append(list(senses))
        return range(start, self._n)

    def add_csr(self, indptr, indices, values, rhs, sense):
        """a block of rows as arrays; the store keeps the arrays it is given (no per-row objects, no lists)"""
        start, r = self._n, len(indptr) - 1
        self._blocks.append(("csr", np.asarray(indptr), np.asarray(indices), np.asarray(values)))
        self._n += r
        self._rhs.append(np.asarray(rhs, dtype=np.float64))
        self._senses.append((sense, r))
        return range(start, self._n)

    @property
    def rhs(self):
        out = []
        for b in self._rhs:
            out.extend(b.tolist() if isinstance(b, np.ndarray) else b)
        return out

    @property
    def senses(self):
        out = []
        for b in self._senses:
            out.extend([b[0]] * b[1] if isinstance(b, tuple) else b)
        return out

    def _tail(self, per_block, start, conv):
        """arrays of the per-row attribute of the rows from ``start`` on: only the blocks behind ``start`` are touched"""
        out, seen = [], 0
        for b in per_block:
            nb = b[1] if isinstance(b, tuple) else len(b)
            skip = min(max(start - seen, 0), nb)
            seen += nb
            if skip < nb:
                out.append(conv(b, skip))
        return out

    def rhs_from(self, start=0):
        """right-hand sides of rows ``start``... as one float64 array (what a solve passes on: the rows added since the last)"""
        parts = self._tail(self._rhs, start, lambda b, s: np.asarray(b[s:], dtype=np.float64))
        return np.concatenate(parts) if parts else np.zeros(0)

    def senses_from(self, start=0):
        """senses of rows ``start``... as an array of one-character strings"""
        parts = self._tail(self._senses, start,
                           lambda b, s: np.full(b[1] - s, b[0], dtype="<U1") if isinstance(b, tuple) else np.asarray(b[s:], dtype="<U1"))
        return np.concatenate(parts) if parts else np.zeros(0, dtype="<U1")

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

    def csr_parts(self, first_row=0):
        """(data, cols, lengths) of the rows from ``first_row`` on, in order, as arrays."""
        data, cols, lens = [], [], []
        seen = 0
        for b in self._blocks:
            nb = len(b[1]) if b[0] == "rows" else len(b[1]) - 1
            skip = min(max(first_row - seen, 0), nb)
            seen += nb
            if skip == nb:
                continue
            if b[0] == "rows":
                for row in b[1][skip:]:
                    data.append(np.asarray(row.val, dtype=np.float64))
                    cols.append(np.asarray(row.ind, dtype=np.int64))
                    lens.append(len(row.ind))
            else:
                _, ptr, ind, val = b
                data.append(val[ptr[skip]:])
                cols.append(ind[ptr[skip]:])
                lens.extend(np.diff(ptr[skip:]).tolist())
        cat = (lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dtype=dt))
        return cat(data, np.float64), cat(cols, np.int64), np.asarray(lens, dtype=np.int64)


class LinearRelaxation(object):
    """Minimal LP model: min c^T v, 0 <= v <= 1, rows added through
    ``self.linear_constraints.add`` exactly as the reference does on its CPLEX object.

    Solved with HiGHS.  When SciPy's bundled HiGHS binding is importable the model is kept inside
    the solver between solves and only the rows added since the last solve are passed on.  Models
    of up to ``IPM_ROWS`` rows use the dual simplex, re-optimising from the previous basis (what
    the reference gets from CPLEX, cut_select_qp.py:106-110, :194); larger ones (spar125-075-* after
    a round of 5000 cuts) the interior-point method with crossover to a vertex, whose time grows far
    more slowly with the number of cuts (dual simplex: 7 / 21 / 43 s for rounds 1-3 of
    spar125-075-1 dim 3, 360 / 630 / 780 s for rounds 3-5 at dim 4; interior point: 2 - 7 s).
    Without the binding every solve starts from scratch through ``scipy.optimize.linprog``."""
    IPM_ROWS = 20000

    def __init__(self, obj_coeffs, incremental=True):
        self.obj = np.asarray(obj_coeffs, dtype=np.float64)
        self.linear_constraints = _RowStore()
        self._values = None
        self._objval = None
        self._core = None
        if incremental:
            try:
                import scipy.optimize._highspy._core as core
                self._core = core
            except ImportError:
                pass
        self._model, self._rows_passed = None, 0

    def _solve_incremental(self):
        core, st, nv = self._core, self.linear_constraints, self.obj.shape[0]
        if self._model is None:
            m = core._Highs()
            m.setOptionValue("output_flag", False)
            m.setOptionValue("solver", "simplex")
            m.setOptionValue("simplex_strategy", 1)          # dual simplex
            m.setOptionValue("presolve", "off")              # keep the basis meaningful across rounds
            m.addVars(nv, np.zeros(nv), np.ones(nv))
            m.changeColsCost(nv, np.arange(nv, dtype=np.int32), self.obj)
            self._model = m
        m = self._model
        data, cols, lens = st.csr_parts(self._rows_passed)
        r = lens.shape[0]
        if r:
            senses = st.senses_from(self._rows_passed)          # (the list properties rebuild ALL rows: quadratic over the rounds)
            rhs = st.rhs_from(self._rows_passed)
            inf = core.kHighsInf
            lower = np.where(senses == "L", -inf, rhs)
            upper = np.where(senses == "G", inf, rhs)
            starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
            m.addRows(r, lower, upper, int(data.shape[0]), starts, cols.astype(np.int32), data)
            self._rows_passed += r
        big = self._rows_passed > self.IPM_ROWS
        m.setOptionValue("solver", "ipm" if big else "simplex")
        m.setOptionValue("run_crossover", "on")
        m.run()
        if m.getModelStatus() != core.HighsModelStatus.kOptimal:
            raise RuntimeError("HiGHS: " + m.modelStatusToString(m.getModelStatus()))
        self._values = np.array(m.getSolution().col_value, dtype=np.float64)
        self._objval = float(m.getInfo().objective_function_value)

    def solve(self):
        st = self.linear_constraints
        assert np.all(np.isin(st.senses_from(self._rows_passed if self._core is not None else 0), ["G", "L", "E"]))
        if self._core is not None:
            return self._solve_incremental()
        from scipy.optimize import linprog
        from scipy.sparse import csr_matrix
        nv = self.obj.shape[0]
        data, cols, lens = st.csr_parts()
        senses = st.senses_from(0)
        sign = np.where(senses == "G", -1.0, 1.0)
        rhs = st.rhs_from(0) * sign
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


def mccormick_csr(nb_vars, adj):
    """McCormick / RLT relaxation of X = x x^T on the edges of ``adj`` as ONE CSR block of "<="
    rows over the variables [X packed | x] (what cut_select_qp.py:352-375 adds row by row):

        per variable i:      X_ii - x_i <= 0,   -X_ii + 2 x_i <= 1
        per edge i < j:     -X_ij + x_i + x_j <= 1,   X_ij - x_i <= 0,   X_ij - x_j <= 0

    Rows are emitted variable by variable, a variable's two diagonal rows first and then its
    edges by ascending j -- the reference's order, which the LP solver's vertex choice can depend
    on.  -> (indptr, indices, values, rhs)."""
    n = int(nb_vars)
    L = n * (n + 1) // 2
    A = np.asarray(adj) != 0
    i_d = np.arange(n, dtype=np.int64)
    X_d = n * i_d - i_d * (i_d - 1) // 2                      # packed position of X_ii
    ei, ej = np.nonzero(np.triu(A, 1))
    ei, ej = ei.astype(np.int64), ej.astype(np.int64)
    X_e = X_d[ei] + (ej - ei)
    # three slots per row, unused slot = -1; sort key = (variable, diagonal first, j, row kind)
    W = 3 * n + 2
    blocks = [
        (i_d * W + 0, np.stack([X_d, L + i_d, -np.ones(n, np.int64)], 1), np.tile([1.0, -1.0, 0.0], (n, 1)), 0.0),
        (i_d * W + 1, np.stack([X_d, L + i_d, -np.ones(n, np.int64)], 1), np.tile([-1.0, 2.0, 0.0], (n, 1)), 1.0),
        (ei * W + 2 + 3 * ej + 0, np.stack([X_e, L + ei, L + ej], 1), np.tile([-1.0, 1.0, 1.0], (ei.size, 1)), 1.0),
        (ei * W + 2 + 3 * ej + 1, np.stack([X_e, L + ei, -np.ones(ei.size, np.int64)], 1),
         np.tile([1.0, -1.0, 0.0], (ei.size, 1)), 0.0),
        (ei * W + 2 + 3 * ej + 2, np.stack([X_e, L + ej, -np.ones(ei.size, np.int64)], 1),
         np.tile([1.0, -1.0, 0.0], (ei.size, 1)), 0.0),
    ]
    key = np.concatenate([b[0] for b in blocks])
    cols = np.concatenate([b[1] for b in blocks])
    vals = np.concatenate([b[2] for b in blocks])
    rhs = np.concatenate([np.full(b[0].shape[0], b[3]) for b in blocks])
    order = np.argsort(key, kind="stable")
    cols, vals, rhs = cols[order], vals[order], rhs[order]
    live = cols >= 0
    indptr = np.concatenate([[0], np.cumsum(live.sum(axis=1))]).astype(np.int64)
    return indptr, cols[live], vals[live], rhs


def boxqp_relaxation(inst):
    """McCormick relaxation M of a parsed BoxQP instance, ready to solve."""
    lp = LinearRelaxation(np.concatenate([inst["Q_arr"], inst["c"]]))
    lp.linear_constraints.add_csr(*mccormick_csr(inst["nb_vars"], inst["adj"]), "L")
    return lp


# ---------------------------------------------------------------------------------------------
# Cutting-plane rounds around the GPU path: "solve, separate, solve, ..." with timers.
class RoundLog(object):
    """What a run of cutting-plane rounds records: the LP bound after every solve (entry 0 = the
    McCormick relaxation), and per round the separation time (selection + generation of cuts, the
    reference's published "separation time"), the LP solve time and the cut counts reported by the
    separator."""

    def __init__(self):
        self.bounds, self.solve_s, self.separation_s = [], [], []
        self.counts = []            # one dict per round, as returned by the separator

    def column(self, name, default=0):
        return [c.get(name, default) for c in self.counts]

    def stalled(self, tol):
        """True when the last round closed less than ``tol`` of the gap closed so far (the
        reference's optional stopping rule, cut_select_qp.py:153-156)."""
        b = self.bounds
        return len(b) >= 3 and b[-1] != b[0] and (b[-1] - b[-2]) / (b[-1] - b[0]) < tol


def run_cut_rounds(lp, separate, max_rounds, setup_s=0.0, stop_tol=None, clock=None, on_round=None, after_solve=None):
    """Drive ``max_rounds`` rounds on ``lp`` (anything with solve / get_values /
    get_objective_value): ``separate(round_no, point) -> dict of counts`` appends cuts to the LP
    between two solves.  ``setup_s`` is added to the first solve's time (model building).
    ``on_round(round_no, log)`` is called after every solve (progress of long runs); ``after_solve()`` the moment a solve
    returns, before the solution vector is extracted (the GPU classes poke the idle device there: sdpcut_wake).
    -> RoundLog."""
    from timeit import default_timer
    clock = clock or default_timer
    log = RoundLog()

    def solve():
        t = clock()
        lp.solve()
        if after_solve:
            after_solve()
        log.solve_s.append(clock() - t)
        log.bounds.append(lp.get_objective_value())
        return np.asarray(lp.get_values(), dtype=np.float64)

    point = solve()
    log.solve_s[0] += setup_s
    if on_round:
        on_round(0, log)
    for round_no in range(1, max_rounds + 1):
        if stop_tol is not None and log.stalled(stop_tol):
            break
        t = clock()
        log.counts.append(separate(round_no, point))
        log.separation_s.append(clock() - t)
        point = solve()
        if on_round:
            on_round(round_no, log)
    return log


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
    (both in the order of the objective+constraints enumeration).  -> two (set_inds, ks) pairs.
    The reference's O(N^2) list-membership tests become one sorted lookup over integer codes of
    the index sets (base nb_vars + 1, padding = 0)."""
    So, ko, _ = enumerate_cover(inst["adj"], dim)
    Sc, kc, _ = enumerate_cover(inst["adj_cons"], dim)
    base = np.int64(inst["nb_vars"] + 1)

    def codes(S):
        c = np.zeros(S.shape[0], dtype=np.int64)
        for a in range(5):
            c = c * base + (S[:, a].astype(np.int64) + 1)      # -1 padding -> digit 0
        return c

    mask = np.isin(codes(Sc), codes(So))
    return (Sc[mask], kc[mask]), (Sc[~mask], kc[~mask])
