"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy + the C restatement oracle/libnnoracle.so) of the reference's
per-round cut scoring / selection / generation path:

    cut_select_qp.py:529-540   candidate records (Xarr_inds, Q_slice, max_elem)
    cut_select_qp.py:543-703   _sel_eigcut_by_ordering_on_measure  (strategies 1, 2, 4)
    cut_select_qp.py:705-755   _gen_eigcuts_selected
    cut_select_qp.py:788-797   _get_eigendecomp   (numpy.linalg eigh/eigvalsh, UPLO="U")
    cut_select_qcqp.py:63-103  QCQP composition of the above

Pinned by tests/golden/ (produced by tests/golden/make_golden.py in the build container
from the real reference: NNs.so called through ctypes, numpy LAPACK, the reference's own
methods run on real instances, and the published data_figures/fig8_data.csv round 1).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (sdpcutsel_via_nn_amd) never does.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

THRES_NEG_EIGVAL = -10 ** (-15)   # cut_select_qp.py:24
THRES_MIN_OPT = 0                 # cut_select_qp.py:22
BIG_M = 1000                      # cut_select_qp.py:26

_lib = None


def nn_lib():
    """ctypes handle to the C restatement (built by oracle/Makefile)."""
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libnnoracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libnnoracle.so missing: run `make -C oracle`")
        lib = ctypes.CDLL(path)
        for k in (2, 3, 4, 5):
            getattr(lib, "neural_net_%dD" % k).restype = ctypes.c_double
        dp = ctypes.POINTER(ctypes.c_double)
        lib.oracle_nn_batch.argtypes = [ctypes.c_int, ctypes.c_int64, dp, dp]
        lib.oracle_opt_score_batch.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_int32), dp, dp, dp, dp]
        _lib = lib
    return _lib


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# --------------------------------------------------------------------------- records
def triu_positions(set_inds, nb_vars):
    """Packed row-major upper-triangle positions of all pairs (a<=b) of set_inds
    (cut_select_qp.py:530-531)."""
    s = np.asarray(set_inds, dtype=np.int64)
    k = s.shape[-1]
    ia, ib = np.triu_indices(k)
    a, b = s[..., ia], s[..., ib]
    return nb_vars * a - a * (a + 1) // 2 + b


def candidate_record(set_inds, nb_vars, Q_arr):
    """(set_inds, Xarr_inds, Q_slice tuple, max_elem) as built at cut_select_qp.py:529-540."""
    pos = [int(p) for p in triu_positions(set_inds, nb_vars)]
    q = [Q_arr[p] for p in pos]
    big = abs(max(q, key=abs))
    max_elem = len(set_inds) * big
    if not max_elem:
        max_elem += 1
    return (list(set_inds), pos, tuple(np.divide(q, max_elem)), max_elem)


def build_agg_list(idx_sets, nb_vars, Q_arr):
    return [candidate_record(s, nb_vars, Q_arr) for s in idx_sets]


# --------------------------------------------------------------------------- eigen
def lifted_matrix(k, curr_pt, X_slice):
    """Upper triangle of [[1, x^T], [x, X]] ((k+1) x (k+1)); lower triangle left zero,
    exactly what the reference hands to LAPACK with UPLO='U' (cut_select_qp.py:64-68, 792-794)."""
    M = np.zeros((k + 1, k + 1))
    M[0, 0] = 1
    M[0, 1:] = curr_pt
    iu = np.triu_indices(k)
    M[iu[0] + 1, iu[1] + 1] = X_slice
    return M


def get_eigendecomp(k, curr_pt, X_slice, ev_yes):
    M = lifted_matrix(k, curr_pt, X_slice)
    return np.linalg.eigh(M, "U") if ev_yes else np.linalg.eigvalsh(M, "U")


def eigmin_batch(k, x_rho, X_rho):
    """lambda_min for a batch: x_rho [N,k], X_rho [N,m].  Batched LAPACK call is
    bit-identical to the per-matrix calls."""
    N = x_rho.shape[0]
    if N == 0:
        return np.zeros(0)
    M = np.zeros((N, k + 1, k + 1))
    M[:, 0, 0] = 1
    M[:, 0, 1:] = x_rho
    iu = np.triu_indices(k)
    M[:, iu[0] + 1, iu[1] + 1] = X_rho
    return np.linalg.eigvalsh(M, "U")[:, 0]


# --------------------------------------------------------------------------- scoring
def nn_scalar(k, vec):
    """One call of neural_net_kD, the reference's own calling pattern (cut_select_qp.py:579-582)."""
    d = k * (k + 3) // 2
    buf = (ctypes.c_double * d)(*vec)
    return getattr(nn_lib(), "neural_net_%dD" % k)(buf)


def nn_batch(k, inputs):
    inputs = np.ascontiguousarray(inputs, dtype=np.float64)
    out = np.empty(inputs.shape[0])
    rc = nn_lib().oracle_nn_batch(k, inputs.shape[0], _dptr(inputs), _dptr(out))
    assert rc == 0
    return out


def opt_score_batch(k, set_inds, nb_vars, vars_values, Q_arr, want_raw=False):
    """obj_improve for N same-size candidates (C restatement, oracle/nn_ref.c)."""
    set_inds = np.ascontiguousarray(set_inds, dtype=np.int32)
    vars_values = np.ascontiguousarray(vars_values, dtype=np.float64)
    Q_arr = np.ascontiguousarray(Q_arr, dtype=np.float64)
    N = set_inds.shape[0]
    out = np.empty(N)
    raw = np.empty(N)
    rc = nn_lib().oracle_opt_score_batch(
        k, N, nb_vars, set_inds.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        _dptr(vars_values), _dptr(Q_arr), _dptr(out), _dptr(raw))
    assert rc == 0
    return (out, raw) if want_raw else out


def gather_point(agg_entry, X_vals, x_vals):
    set_inds, Xarr_inds = agg_entry[0], agg_entry[1]
    curr_pt = tuple(x_vals[i] for i in set_inds)
    X_slice = tuple(X_vals[i] for i in Xarr_inds)
    return curr_pt, X_slice


def opt_score_entry(agg_entry, X_vals, x_vals):
    """obj_improve of one candidate, reference op order (cut_select_qp.py:573-582)."""
    set_inds, Xarr_inds, Q_slice, max_elem = agg_entry
    k = len(set_inds)
    curr_pt, X_slice = gather_point(agg_entry, X_vals, x_vals)
    acc = 0
    for q, X in zip(Q_slice, X_slice):
        acc = acc + q * X
    obj = -acc * max_elem
    obj += nn_scalar(k, list(curr_pt) + list(Q_slice)) * max_elem
    return obj, curr_pt, X_slice


# --------------------------------------------------------------------------- selection
def sel_eigcut_by_ordering_on_measure(agg_list, nb_lifted, strat, vars_values, sel_size=0):
    """Strategies 1 (feasibility), 2 (optimality), 4 (combined) of
    cut_select_qp.py:543-654.  Returns what the reference returns: a list for 1/2,
    (new_strat, list) for 4 (or the bare list when sel_size == 0, mirroring the
    reference's swallowed ZeroDivisionError at :631-632)."""
    X_vals, x_vals = list(vars_values[0:nb_lifted]), list(vars_values[nb_lifted:])
    N = len(agg_list)
    sel_size = min(sel_size, N)
    if strat in (2, 4):
        ranked = []
        for idx, entry in enumerate(agg_list):
            obj, curr_pt, X_slice = opt_score_entry(entry, X_vals, x_vals)
            ranked.append((idx, obj, curr_pt, X_slice))
        ranked.sort(key=lambda e: e[1], reverse=True)
        if strat == 2:
            return ranked
        strong = viol = 0
        for pos, (idx, obj, curr_pt, X_slice) in enumerate(ranked):
            if strong >= sel_size:
                break
            lam = get_eigendecomp(len(curr_pt), curr_pt, X_slice, False)[0]
            if obj > THRES_MIN_OPT:
                if lam < THRES_NEG_EIGVAL:
                    ranked[pos] = (idx, obj + BIG_M, curr_pt, X_slice)
                    strong += 1
                    viol += 1
                else:
                    ranked[pos] = (idx, obj - BIG_M, curr_pt, X_slice)
            elif lam < THRES_NEG_EIGVAL:
                ranked[pos] = (idx, -lam, curr_pt, X_slice)
                viol += 1
        ranked.sort(key=lambda e: e[1], reverse=True)
        if sel_size == 0 or N == 0:
            return ranked
        return (1, ranked) if strong / sel_size < viol / N else (4, ranked)
    if strat == 1:
        ranked = []
        nb_violated = 0
        for entry in agg_list:
            set_inds, Xarr_inds = entry[0], entry[1]
            curr_pt, X_slice = gather_point(entry, X_vals, x_vals)
            lam = get_eigendecomp(len(set_inds), curr_pt, X_slice, False)[0]
            if lam < THRES_NEG_EIGVAL:
                ranked.append((set_inds, -lam, Xarr_inds, len(set_inds)))
                nb_violated += 1
            else:
                ranked.append((0, 0))
        ranked.sort(key=lambda e: e[1], reverse=True)
        return ranked[0:nb_violated]
    raise ValueError("oracle covers strategies 1, 2, 4")


def cut_row(k, set_inds, Xarr_inds, nb_lifted, evect):
    """Row of the eigen-cut v^T [[1,x^T],[x,X]] v >= 0 (cut_select_qp.py:744-750):
    columns, coefficients, rhs."""
    v = np.where(abs(evect) <= -THRES_NEG_EIGVAL, 0, evect)
    coef = []
    for a in range(k + 1):
        for b in range(max(a, 1), k + 1):
            coef.append(v[a] * v[b] * 2 if a != b else v[a] * v[b])
    ind = [i + nb_lifted for i in set_inds] + list(Xarr_inds)
    return ind, coef, -v[0] * v[0]


def gen_eigcuts_selected(agg_list, nb_lifted, strat, sel_size, rank_list,
                         strong_only=False, vars_values=None):
    """cut_select_qp.py:705-755 without the CPLEX call: returns
    (nb_sdp_cuts, rows=[(ind, val)], rhs, senses)."""
    opt_sel, feas_sel = strat in (2, 3, 4, -1), strat == 1
    sel_size = min(sel_size, len(rank_list))
    if not opt_sel:
        X_vals, x_vals = list(vars_values[0:nb_lifted]), list(vars_values[nb_lifted:])
    rows, rhs, senses = [], [], []
    pos = 0
    while len(rows) < sel_size and pos < sel_size:
        if feas_sel:
            set_inds, _, Xarr_inds, k = rank_list[pos]
            curr_pt = tuple(x_vals[i] for i in set_inds)
            X_slice = tuple(X_vals[i] for i in Xarr_inds)
        elif opt_sel:
            idx, diff, curr_pt, X_slice = rank_list[pos]
            if strong_only and diff <= 0:
                break
            set_inds, Xarr_inds = agg_list[idx][0:2]
            k = len(set_inds)
        else:
            set_inds, Xarr_inds = rank_list[pos][0:2]
            curr_pt = tuple(x_vals[i] for i in set_inds)
            X_slice = tuple(X_vals[i] for i in Xarr_inds)
            k = len(set_inds)
        eigvals, evecs = get_eigendecomp(k, curr_pt, X_slice, True)
        if eigvals[0] < THRES_NEG_EIGVAL:
            ind, coef, r = cut_row(k, set_inds, Xarr_inds, nb_lifted, evecs.T[0])
            rows.append((ind, coef))
            rhs.append(r)
            senses.append("G")
        pos += 1
    return len(rows), rows, rhs, senses


def qcqp_round(agg_list_obj, agg_list_cons, nb_lifted, strat, vars_values, sel_size):
    """One round of cut_select_qcqp.py:64-98 (strat in 1, 2, 4).  Returns dict with the
    concatenated rank list, counters and the generated rows of both generation calls."""
    strat_old = strat
    if strat == 4:
        strat, comb_obj = sel_eigcut_by_ordering_on_measure(
            agg_list_obj, nb_lifted, 4, vars_values, sel_size=sel_size)
    else:
        comb_obj = sel_eigcut_by_ordering_on_measure(agg_list_obj, nb_lifted, strat, vars_values)
    feas_cons = sel_eigcut_by_ordering_on_measure(agg_list_cons, nb_lifted, 1, vars_values)
    rank_list = (comb_obj + feas_cons)[0:sel_size]
    if strat_old == 1:
        n, rows, rhs, _ = gen_eigcuts_selected(agg_list_obj, nb_lifted, strat_old, sel_size, rank_list,
                                               vars_values=vars_values)
        return dict(new_strat=strat, rank_list=rank_list, nb_sdp_cuts=n, rows=rows, rhs=rhs,
                    nb_opt_cuts=0, nb_cuts_combined=0)
    nb_opt_cuts = sum(1 for e in comb_obj if e[1] > BIG_M)
    nb_cuts_combined = sum(1 for e in rank_list if isinstance(e[0], int))
    rest = sel_size - nb_cuts_combined
    n1, rows1, rhs1, _ = gen_eigcuts_selected(agg_list_obj, nb_lifted, 1, rest, feas_cons[0:rest],
                                              vars_values=vars_values)
    n2, rows2, rhs2, _ = gen_eigcuts_selected(agg_list_obj, nb_lifted, strat_old, nb_cuts_combined,
                                              comb_obj[0:nb_cuts_combined], vars_values=vars_values)
    return dict(new_strat=strat, rank_list=rank_list, nb_sdp_cuts=n1 + n2, rows=rows1 + rows2,
                rhs=rhs1 + rhs2, nb_opt_cuts=nb_opt_cuts, nb_cuts_combined=nb_cuts_combined)


# --------------------------------------------------------------------------- array form
def rank_arrays(strat, obj, lam, sel_size):
    """Array (closed-form) version of the ranking for large N, same semantics as
    sel_eigcut_by_ordering_on_measure: returns (order, scores_in_order, new_strat, counters).
    `order` lists candidate indices in final rank order (full length for 2/4, violated only
    for 1).  Checked against the literal loop above in tests/test_oracle.py."""
    N = lam.shape[0] if lam is not None else obj.shape[0]
    sel_size = min(sel_size, N)
    if strat == 1:
        viol = lam < THRES_NEG_EIGVAL
        score = np.where(viol, -lam, 0.0)
        order = np.argsort(-score, kind="stable")
        nv = int(viol.sum())
        return order[:nv], score[order[:nv]], 1, dict(nb_violated=nv)
    first = np.argsort(-obj, kind="stable")
    if strat == 2:
        return first, obj[first], 2, {}
    s = obj[first]
    viol = lam[first] < THRES_NEG_EIGVAL
    pos = s > THRES_MIN_OPT
    strong_flag = pos & viol
    before = np.cumsum(strong_flag) - strong_flag          # strong ones strictly earlier
    visited = before < sel_size
    new = s.copy()
    up = visited & strong_flag
    down = visited & pos & ~viol
    low = visited & ~pos & viol
    new[up] = s[up] + BIG_M
    new[down] = s[down] - BIG_M
    new[low] = -lam[first][low]
    strong = int(up.sum())
    nviol = strong + int(low.sum())
    second = np.argsort(-new, kind="stable")
    order = first[second]
    new_strat = 4
    if sel_size > 0 and N > 0 and strong / sel_size < nviol / N:
        new_strat = 1
    return order, new[second], new_strat, dict(strong=strong, violated=nviol)


# --------------------------------------------------------------------------- triangle inequalities
THRES_TRI_DENSE = 2             # cut_select_qp.py:31
THRES_TRI_VIOL = 10 ** (-7)     # cut_select_qp.py:33
TRI_CUTS_PER_ROUND_MIN = 5000   # cut_select_qp.py:39
TRI_CUTS_PER_ROUND_MAX = 10000  # cut_select_qp.py:41


def preprocess_triangle_ineq(nb_vars, adj):
    """cut_select_qp.py:799-822: triples i1<i2<i3 with at least two of their three edges in the
    sparsity graph, lexicographic.  -> (triples [T,3], density [T])."""
    triples, dens = [], []
    for i1 in range(nb_vars):
        for i2 in range(i1 + 1, nb_vars):
            for i3 in range(i2 + 1, nb_vars):
                d = float(adj[i1, i2]) + float(adj[i1, i3]) + float(adj[i2, i3])
                if d >= THRES_TRI_DENSE:
                    triples.append((i1, i2, i3))
                    dens.append(d)
    return np.array(triples, dtype=np.int32).reshape(-1, 3), np.array(dens)


def separate_triangle(nb_vars, triples, density, sel_size, vars_values):
    """cut_select_qp.py:824-863 without the CPLEX call.  Returns (nb_tri_cuts, viol [T,4],
    order = entry ids 4*t+type of the selected cuts, rows [(ind, val)], rhs)."""
    L = nb_vars * (nb_vars + 1) // 2
    X_vals, x_vals = list(vars_values[0:L]), list(vars_values[L:])
    T = triples.shape[0]
    viol = np.zeros((T, 4))
    pos = []
    for t in range(T):
        s = [int(v) for v in triples[t]]
        Xi = [nb_vars * a - a * (a + 1) // 2 + b for a, b in ((s[0], s[0]), (s[0], s[1]), (s[0], s[2]),
                                                             (s[1], s[1]), (s[1], s[2]), (s[2], s[2]))]
        pos.append(Xi)
        X1, X2, X4 = X_vals[Xi[1]], X_vals[Xi[2]], X_vals[Xi[4]]
        pt = [x_vals[i] for i in s]
        viol[t, 0] = X1 + X2 - X4 - pt[0]
        viol[t, 1] = X1 - X2 + X4 - pt[1]
        viol[t, 2] = -X1 + X2 + X4 - pt[2]
        viol[t, 3] = -X1 - X2 - X4 + sum(pt) - 1
    entries = [(4 * t + c, density[t], viol[t, c]) for t in range(T) for c in range(4) if viol[t, c] >= THRES_TRI_VIOL]
    entries.sort(key=lambda e: (e[1], e[2]), reverse=True)
    nb = max(min(TRI_CUTS_PER_ROUND_MIN, int(np.floor(sel_size * len(entries)))),
             min(TRI_CUTS_PER_ROUND_MAX, len(entries)))
    coeffs = {0: [-1, -1, 1, 1], 1: [-1, 1, -1, 1], 2: [1, -1, -1, 1], 3: [1, 1, 1, -1, -1, -1]}
    rows, rhs = [], []
    for e in entries[:nb]:
        t, c = divmod(e[0], 4)
        Xi, s = pos[t], [int(v) for v in triples[t]]
        if c == 3:
            rows.append(([Xi[1], Xi[2], Xi[4], s[0] + L, s[1] + L, s[2] + L], coeffs[3]))
            rhs.append(-1)
        else:
            rows.append(([Xi[1], Xi[2], Xi[4], s[c] + L], coeffs[c]))
            rhs.append(0)
    return nb, viol, np.array([e[0] for e in entries[:nb]], dtype=np.int64), rows, rhs
