#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference); nothing here runs on the GPU
box.  Three independent sources pin the oracle (SURVEY.md section 8 c):

 1. neural_nets/NNs.so called through ctypes on random inputs           -> nn_k{k}.npz
    numpy.linalg.eigvalsh / eigh (the LAPACK the reference calls)        -> same files
 2. the reference's own methods (_get_sdp_vertex_cover,
    _sel_eigcut_by_ordering_on_measure, _gen_eigcuts_selected, QCQP round) run
    unmodified on real instances; third-party solver modules that are not
    installed (cplex, mosek, cvxopt, chompack, lxml) are replaced by inert
    recording objects, because the hot path never calls into them         -> inst_*.npz
 3. the published known answers data_figures/fig8_data.csv rows 7-1057   -> fig8_round1.csv

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import ctypes
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from sdpcutsel_via_nn_amd import harness  # noqa: E402


# ----------------------------------------------------------------------------- stand-ins
class _Recorder(object):
    """Accepts any attribute access / call; records linear_constraints.add rows."""

    def __init__(self):
        self.rows, self.rhs, self.senses = [], [], []

    def __getattr__(self, name):
        return self

    def __call__(self, *a, **k):
        return self

    def add(self, *a, **k):
        if "lin_expr" in k:
            self.rows.extend(k["lin_expr"])
            self.rhs.extend(k["rhs"])
            self.senses.extend(k["senses"])
        return self


class _Dense(object):
    """cvxopt.spmatrix look-alike backed by an ndarray ([i, j] get/set, [:] copy)."""

    def __init__(self, v, I, J, size):
        self.a = np.zeros(size)
        self.a[np.asarray(I, dtype=int), np.asarray(J, dtype=int)] = v

    def __getitem__(self, ij):
        if isinstance(ij, slice):
            out = _Dense(0, [], [], self.a.shape)
            out.a = self.a.copy()
            return out
        return self.a[ij]

    def __setitem__(self, ij, v):
        self.a[ij] = v


def install_standins():
    cplex = types.ModuleType("cplex")
    cplex.Cplex = _Recorder
    cplex.SparsePair = harness.SparsePair
    mosek = types.ModuleType("mosek")
    fusion = types.ModuleType("mosek.fusion")
    for n in ("Model", "Domain", "ObjectiveSense", "Expr"):
        setattr(fusion, n, _Recorder())
    mosek.fusion = fusion
    cvxopt = types.ModuleType("cvxopt")
    cvxopt.spmatrix = _Dense
    cvxopt.amd = _Recorder()
    chompack = types.ModuleType("chompack")
    lxml = types.ModuleType("lxml")
    import xml.etree.ElementTree as ET
    lxml.etree = ET
    sys.modules.update({"cplex": cplex, "mosek": mosek, "mosek.fusion": fusion, "cvxopt": cvxopt,
                        "chompack": chompack, "lxml": lxml, "lxml.etree": ET})


def import_reference():
    install_standins()
    sys.path.insert(0, REF)
    os.chdir(REF)                      # NNs.so path is cwd-relative (cut_select_qp.py:293)
    import cut_select_qp
    import cut_select_qcqp
    warnings.resetwarnings()           # the modules set process-wide filters on import
    warnings.simplefilter("ignore")
    return cut_select_qp, cut_select_qcqp


# ----------------------------------------------------------------------------- helpers
def pack_agg(agg_list):
    """agg_list -> arrays: set_inds padded to 5 with -1, k per candidate."""
    N = len(agg_list)
    ks = np.array([len(e[0]) for e in agg_list], dtype=np.int32)
    S = -np.ones((N, 5), dtype=np.int32)
    for i, e in enumerate(agg_list):
        S[i, :ks[i]] = e[0]
    return S, ks


def pack_rows(rec):
    """recorded SparsePair rows -> flat arrays."""
    ptr = np.cumsum([0] + [len(r.ind) for r in rec.rows]).astype(np.int64)
    ind = np.array([i for r in rec.rows for i in r.ind], dtype=np.int64)
    val = np.array([v for r in rec.rows for v in r.val], dtype=np.float64)
    return ptr, ind, val, np.array(rec.rhs, dtype=np.float64)


def rank_ids(rank_list, agg_list, strat):
    """candidate index of every rank-list entry."""
    if strat == 1:
        key = {tuple(e[0]): i for i, e in enumerate(agg_list)}
        return np.array([key[tuple(e[0])] for e in rank_list], dtype=np.int64)
    return np.array([e[0] for e in rank_list], dtype=np.int64)


def run_instance(cs, tag, agg_list, nb_vars, Q_arr, points, sel_size, out):
    S, ks = pack_agg(agg_list)
    out[tag + "_nb_vars"] = np.int64(nb_vars)
    out[tag + "_Q_arr"] = np.asarray(Q_arr, dtype=np.float64)
    out[tag + "_set_inds"] = S
    out[tag + "_k"] = ks
    out[tag + "_sel_size"] = np.int64(sel_size)
    out[tag + "_max_elem"] = np.array([e[3] for e in agg_list], dtype=np.float64)
    for pname, vv in points.items():
        p = "%s_%s" % (tag, pname)
        out[p + "_vars"] = vv
        for strat in (1, 2, 4):
            cs._my_prob = _Recorder()
            if strat == 4:
                new_strat, rl = cs._sel_eigcut_by_ordering_on_measure(4, vv, 1, sel_size=sel_size)
            else:
                new_strat, rl = strat, cs._sel_eigcut_by_ordering_on_measure(strat, vv, 1)
            q = "%s_s%d" % (p, strat)
            out[q + "_order"] = rank_ids(rl, agg_list, strat)
            out[q + "_score"] = np.array([e[1] for e in rl], dtype=np.float64)
            out[q + "_new_strat"] = np.int64(new_strat)
            nb = cs._gen_eigcuts_selected(strat, sel_size, rl, vars_values=vv)
            ptr, ind, val, rhs = pack_rows(cs._my_prob)
            out[q + "_nb_cuts"] = np.int64(nb)
            out[q + "_row_ptr"], out[q + "_row_ind"] = ptr, ind
            out[q + "_row_val"], out[q + "_rhs"] = val, rhs


def mccormick_optimum(nb_vars, adj, Q_arr, c):
    lp = harness.LinearRelaxation(np.concatenate([Q_arr, c]))
    lp.linear_constraints.add_csr(*harness.mccormick_csr(nb_vars, adj), "L")
    lp.solve()
    return np.asarray(lp.get_values())


# ----------------------------------------------------------------------------- main
def golden_random(rng_seed=11):
    """Source 1: NNs.so + LAPACK on random candidates."""
    lib = ctypes.CDLL(os.path.join(REF, "neural_nets", "NNs.so"))
    rng = np.random.default_rng(rng_seed)
    for k in (2, 3, 4, 5):
        m, d = k * (k + 1) // 2, k * (k + 3) // 2
        fn = getattr(lib, "neural_net_%dD" % k)
        fn.restype = ctypes.c_double
        N = 4096
        x = rng.uniform(0, 1, (N, k))
        iu = np.triu_indices(k)
        lo = np.maximum(0, x[:, iu[0]] + x[:, iu[1]] - 1)
        hi = np.minimum(x[:, iu[0]], x[:, iu[1]])
        X = lo + (hi - lo) * rng.uniform(0, 1, (N, m))
        # a few exactly singular / PSD / degenerate points
        x[:64] = 0.5
        X[:64] = rng.integers(0, 2, (64, m)) * 0.5
        X[64:128] = (x[64:128, iu[0]] * x[64:128, iu[1]])          # rank one: lambda_min = 0
        X[128:192] = np.minimum(x[128:192, iu[0]], x[128:192, iu[1]])  # PSD
        q = rng.integers(-50, 51, (N, m)).astype(np.float64)
        q[192:200] = 0.0
        big = k * np.abs(q).max(axis=1)
        big[big == 0] = 1.0
        qs = q / big[:, None]
        inp = np.concatenate([x, qs], axis=1)
        y = np.empty(N)
        buf = (ctypes.c_double * d)()
        for i in range(N):
            buf[:] = inp[i]
            y[i] = fn(buf)
        M = np.zeros((N, k + 1, k + 1))
        M[:, 0, 0] = 1
        M[:, 0, 1:] = x
        M[:, iu[0] + 1, iu[1] + 1] = X
        evals = np.array([np.linalg.eigvalsh(M[i], "U") for i in range(N)])
        vecs = np.array([np.linalg.eigh(M[i], "U")[1][:, 0] for i in range(N)])
        np.savez_compressed(os.path.join(HERE, "nn_k%d.npz" % k), inputs=inp, nn_out=y, x=x, X=X,
                            eigvals=evals, evec_min=vecs,
                            numpy_version=np.array(np.__version__))
        print("nn_k%d.npz" % k, N)


def golden_instances(qp, qcqp):
    out = {}
    rng = np.random.default_rng(7)
    for name, dim in (("spar020-100-1", 3), ("spar020-100-1", 4), ("spar040-030-1", 5),
                      ("spar030-060-1", 3)):
        cs = qp.CutSolver()
        cs._dim = dim
        cs._load_neural_nets()
        cs._CutSolver__parse_boxqp_into_cplex(name)
        nb = cs._get_sdp_vertex_cover(dim)
        agg = cs._agg_list
        n = cs._nb_vars
        inst = harness.parse_boxqp(os.path.join(REF, "boxqp_instances", name + ".in"))
        assert np.array_equal(inst["Q_arr"], np.asarray(cs._Q_arr)), "harness parser differs"
        adj = cs._Q_adj.a != 0
        assert np.array_equal(adj, inst["adj"])
        vv0 = mccormick_optimum(n, adj, inst["Q_arr"], inst["c"])
        vv1 = harness.random_mccormick_point(n, rng)
        sel = min(int(np.floor(0.1 * nb)), 5000)
        tag = "%s_d%d" % (name.replace("-", "_"), dim)
        run_instance(cs, tag, agg, n, inst["Q_arr"], {"mck": vv0, "rnd": vv1}, sel, out)
        # all-PSD point: X = min(x_i, x_j)  -> feasibility list is empty
        x = rng.uniform(0, 1, n)
        iu = np.triu_indices(n)
        vv2 = np.concatenate([np.minimum(x[iu[0]], x[iu[1]]), x])
        run_instance(cs, tag, agg, n, inst["Q_arr"], {"psd": vv2}, sel, out)
        print(tag, "N =", nb, "sel =", sel)
    np.savez_compressed(os.path.join(HERE, "inst_boxqp.npz"), **out)

    # QCQP composition (cut_select_qcqp.py:63-103) on q_20_4_25_1, dim 3
    out = {}
    cs = qcqp.CutSolverQCQP()
    cs._dim = 3
    cs._CutSolverQCQP__parse_qcqp_osil_into_cplex("q_20_4_25_1")
    cs._load_neural_nets()
    agg_cons = cs._CutSolverQCQP__get_vertex_cover(3)
    agg_obj = cs._agg_list[:]
    n = cs._nb_vars
    vv = harness.random_mccormick_point(n, rng)
    So, ko = pack_agg(agg_obj)
    Sc, kc = pack_agg(agg_cons)
    out.update(nb_vars=np.int64(n), Q_arr=np.asarray(cs._Q_arr, dtype=np.float64),
               obj_set_inds=So, obj_k=ko, cons_set_inds=Sc, cons_k=kc, vars=vv)
    for strat in (1, 2, 4):
        for sel in (1, 7, 40):
            cs._agg_list = agg_obj
            if strat == 4:
                new_strat, comb = cs._sel_eigcut_by_ordering_on_measure(4, vv, 1, sel_size=sel)
            else:
                new_strat, comb = strat, cs._sel_eigcut_by_ordering_on_measure(strat, vv, 1)
            cs._agg_list = agg_cons
            feas = cs._sel_eigcut_by_ordering_on_measure(1, vv, 1)
            cs._agg_list = agg_obj
            rl = (comb + feas)[0:sel]
            q = "s%d_sel%d" % (strat, sel)
            out[q + "_new_strat"] = np.int64(new_strat)
            out[q + "_is_obj"] = np.array([isinstance(e[0], int) for e in rl])
            out[q + "_score"] = np.array([e[1] for e in rl], dtype=np.float64)
            ids = []
            key_o = {tuple(e[0]): i for i, e in enumerate(agg_obj)}
            key_c = {tuple(e[0]): i for i, e in enumerate(agg_cons)}
            n_obj_entries = len(comb)
            for pos, e in enumerate(rl):
                if isinstance(e[0], int):
                    ids.append(e[0])
                else:   # feasibility entry: from obj cover if strat 1 and pos < len(comb)
                    src = key_o if (strat == 1 and pos < n_obj_entries) else key_c
                    ids.append(src[tuple(e[0])])
            out[q + "_ids"] = np.array(ids, dtype=np.int64)
            out[q + "_nb_opt_cuts"] = np.int64(sum(1 for e in comb if e[1] > 1000))
    np.savez_compressed(os.path.join(HERE, "inst_qcqp.npz"), **out)
    print("qcqp: obj cover", len(agg_obj), "cons-only cover", len(agg_cons))


def golden_fig8():
    src = open(os.path.join(REF, "data_figures", "fig8_data.csv")).read().split("\n")
    rows = src[5:1057]      # header line 6 + round-1 rows 7..1057 (1-based)
    assert rows[0].startswith("cuts_round,cut_number") and rows[1].startswith("1,")
    assert rows[-1].startswith("1,") and src[1057].startswith("2,")
    with open(os.path.join(HERE, "fig8_round1.csv"), "w") as f:
        f.write("\n".join(rows) + "\n")
    print("fig8_round1.csv", len(rows) - 1, "rows")




def golden_triangles(qp):
    """Source 2, triangle inequalities (cut_select_qp.py:799-863): the reference's own
    pre-processing and separation on two instances at a random McCormick-feasible point and at
    the McCormick optimum."""
    out = {}
    rng = np.random.default_rng(17)
    for name in ("spar020-100-1", "spar040-030-1"):
        cs = qp.CutSolver()
        cs._CutSolver__parse_boxqp_into_cplex(name)
        n = cs._nb_vars
        inst = harness.parse_boxqp(os.path.join(REF, "boxqp_instances", name + ".in"))
        cs._CutSolver__preprocess_triangle_ineq()
        tag = name.replace("-", "_")
        out[tag + "_nb_vars"] = np.int64(n)
        out[tag + "_adj"] = (cs._Q_adj.a != 0)
        out[tag + "_triples"] = np.array([t[1] for t in cs._idx_list_tri], dtype=np.int32).reshape(-1, 3)
        out[tag + "_density"] = np.array([e[2] for e in cs._rank_list_tri], dtype=np.float64)
        pts = {"rnd": harness.random_mccormick_point(n, rng),
               "mck": mccormick_optimum(n, cs._Q_adj.a != 0, inst["Q_arr"], inst["c"])}
        for pname, vv in pts.items():
            for sel in (0.1, 0.5):
                cs._my_prob = _Recorder()
                nb = cs._CutSolver__separate_and_add_triangle(sel, vv)
                viol = np.array([e[3] for e in cs._rank_list_tri], dtype=np.float64)
                ptr, ind, val, rhs = pack_rows(cs._my_prob)
                q = "%s_%s_%s" % (tag, pname, str(sel).replace(".", "p"))
                out[q + "_vars"] = vv
                out[q + "_viol"] = viol
                out[q + "_nb"] = np.int64(nb)
                out[q + "_row_ptr"], out[q + "_row_ind"], out[q + "_row_val"], out[q + "_rhs"] = ptr, ind, val, rhs
        print(tag, "triples", len(cs._idx_list_tri))
    np.savez_compressed(os.path.join(HERE, "inst_tri.npz"), **out)


def golden_qcqp50(qcqp):
    """BASELINE.json config 5 at full size: q_50_10_25_1 with 5-variable sub-problems.  The
    objective cover has 4 of them, the constraints-only cover 1 377 077; one selection round of
    the reference (cut_select_qcqp.py:63-103) at a random McCormick-feasible point.  Stored: the
    point, a checksum of the reference's covers (the test re-enumerates them natively) and the
    first 5000 entries of every strategy's combined list."""
    import zlib
    rng = np.random.default_rng(23)
    name = "q_50_10_25_1"
    cs = qcqp.CutSolverQCQP()
    cs._dim = 5
    cs._CutSolverQCQP__parse_qcqp_osil_into_cplex(name)
    cs._load_neural_nets()
    agg_cons = cs._CutSolverQCQP__get_vertex_cover(5)
    agg_obj = cs._agg_list[:]
    n = cs._nb_vars
    vv = harness.random_mccormick_point(n, rng)
    So, ko = pack_agg(agg_obj)
    Sc, kc = pack_agg(agg_cons)
    out = dict(nb_vars=np.int64(n), vars=vv, obj_set_inds=So, obj_k=ko, cons_count=np.int64(len(agg_cons)),
               cons_crc=np.int64(zlib.crc32(np.ascontiguousarray(Sc).tobytes())), cons_k_all5=np.bool_((kc == 5).all()))
    print("qcqp50: obj cover", len(agg_obj), "cons-only cover", len(agg_cons), flush=True)
    cs._agg_list = agg_cons
    feas = cs._sel_eigcut_by_ordering_on_measure(1, vv, 1)      # 1.4e6 eigvalsh calls
    out["cons_nb_violated"] = np.int64(len(feas))
    key_o = {tuple(e[0]): i for i, e in enumerate(agg_obj)}
    key_c = {tuple(e[0]): i for i, e in enumerate(agg_cons)}
    sel = 5000
    for strat in (1, 2, 4):
        cs._agg_list = agg_obj
        if strat == 4:
            new_strat, comb = cs._sel_eigcut_by_ordering_on_measure(4, vv, 1, sel_size=sel)
        else:
            new_strat, comb = strat, cs._sel_eigcut_by_ordering_on_measure(strat, vv, 1)
        rl = (comb + feas)[0:sel]
        q = "s%d" % strat
        out[q + "_new_strat"] = np.int64(new_strat)
        out[q + "_is_obj"] = np.array([isinstance(e[0], int) for e in rl])
        out[q + "_score"] = np.array([e[1] for e in rl], dtype=np.float64)
        ids = []
        for pos, e in enumerate(rl):
            if isinstance(e[0], int):
                ids.append(e[0])
            else:
                src = key_o if (strat == 1 and pos < len(comb)) else key_c
                ids.append(src[tuple(e[0])])
        out[q + "_ids"] = np.array(ids, dtype=np.int64)
        out[q + "_nb_opt_cuts"] = np.int64(sum(1 for e in comb if e[1] > 1000))
        print("qcqp50 strat", strat, "comb", len(comb), "head", len(rl), flush=True)
    np.savez_compressed(os.path.join(HERE, "inst_qcqp50.npz"), **out)


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else "all"
    if only == "qcqp50":          # minutes of reference Python: not part of "all"
        golden_qcqp50(import_reference()[1])
        sys.exit(0)
    if only in ("all", "fig8"):
        golden_fig8()
    if only in ("all", "random"):
        golden_random()
    qp, qcqp = import_reference()
    if only in ("all", "instances"):
        golden_instances(qp, qcqp)
    if only in ("all", "tri"):
        golden_triangles(qp)
