#!/usr/bin/env python3
"""Golden cutting-plane TRAJECTORIES of the QCQP path (cut_select_qcqp.py:63-103) as the paper runs it
(generate_figs_tables.py:266-272, :616: 3-variable sub-problems, sel_size 5 %, strategies 1 / 4 / 5, 10 rounds):
the REAL reference's selection and cut generation (`_sel_eigcut_by_ordering_on_measure`, `_gen_eigcuts_selected`,
the parser and `__get_vertex_cover`, all unmodified, NNs.so through ctypes, numpy LAPACK) composed per round exactly
as the reference's loop composes them, with this repo's HiGHS relaxation standing in for the CPLEX object (cplex is
not installed; the hot path never calls into it, see make_golden.py).

Per round the fixture keeps the LP point, the strategy in force and the one switched to, the composed head
((A + B)[0:sel_size], :79) as (is_obj, candidate id in its cover, score), the cut counts of :85-97 and the bound
after the re-solve; for the random strategy (5) the index sets of the first sel_size entries of the shuffled list.
A GPU test replays every recorded point through the library and re-runs the whole loop live.

Runs only in the build container (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_qcqp_rounds_golden.py q_30_6_50_1 4 10
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
from make_golden import REF, _Recorder, harness, pack_agg  # noqa: E402

SEED = 7          # generate_figs_tables.py:61 (seed_nb), reset in front of every instance's strategies (:267)


def main(name, strat0, rounds, frac=0.05, dim=3):
    _, qcqp = make_golden.import_reference()
    np.random.seed(SEED)
    cs = qcqp.CutSolverQCQP()
    cs._dim = dim
    cs._CutSolverQCQP__parse_qcqp_osil_into_cplex(name)
    cs._load_neural_nets()
    agg_cons = cs._CutSolverQCQP__get_vertex_cover(dim)         # :50
    agg_obj = cs._agg_list[:]                                   # :51
    n = cs._nb_vars
    inst = harness.parse_osil(os.path.join(REF, "qcqp_instances", name + ".osil"))
    assert np.array_equal(inst["Q_arr"], np.asarray(cs._Q_arr, dtype=np.float64))
    lp = harness.LinearRelaxation(np.concatenate([inst["Q_arr"], inst["c"]]))
    lp.linear_constraints.add(inst["rows"], inst["rhs"], inst["senses"])
    lp.linear_constraints.add_csr(*harness.mccormick_csr(n, inst["adj"]), "L")
    lp.solve()
    nb = len(agg_obj)
    sel = min(int(np.floor(frac * nb)) if frac < 1 else min(frac, nb), 5000)      # :55-56
    if sel == 0:
        sel = 1                                                 # :58
    So, ko = pack_agg(agg_obj)
    Sc, kc = pack_agg(agg_cons)
    out = dict(name=np.array(name), dim=np.int64(dim), strat0=np.int64(strat0), sel_size=np.int64(sel), sel_frac=np.float64(frac),
               seed=np.int64(SEED), obj_set_inds=So, obj_k=ko, cons_set_inds=Sc, cons_k=kc, bounds=[lp.get_objective_value()])
    key_o = {tuple(e[0]): i for i, e in enumerate(agg_obj)}
    key_c = {tuple(e[0]): i for i, e in enumerate(agg_cons)}
    strat = strat_old = strat0
    for r in range(1, rounds + 1):
        vv = np.array(lp.get_values())
        p = "r%02d_" % r
        out[p + "vars"], out[p + "strat"] = vv, np.int64(strat)
        cs._my_prob = _Recorder()
        nb_opt = 0
        if strat == 5:                                          # :64-66, :80-82
            rank_list = cs._sel_eigcut_by_ordering_on_measure(strat, vv, r)
            nb_cuts = cs._gen_eigcuts_selected(strat, sel, rank_list, vars_values=vv)
            S, k = pack_agg(rank_list[0:sel])
            out[p + "rand_set_inds"], out[p + "rand_k"] = S, k
        else:
            cs._agg_list = agg_obj
            if strat == 4:
                strat, comb = cs._sel_eigcut_by_ordering_on_measure(4, vv, r, sel_size=sel)      # :68-70
            else:
                comb = cs._sel_eigcut_by_ordering_on_measure(strat, vv, r)                       # :72-73
            cs._agg_list = agg_cons                                                              # :75
            feas = cs._sel_eigcut_by_ordering_on_measure(1, vv, r)
            cs._agg_list = agg_obj                                                               # :78
            rank_list = (comb + feas)[0:sel]                                                     # :79
            if strat_old == 1:
                nb_cuts = cs._gen_eigcuts_selected(strat_old, sel, rank_list, vars_values=vv)    # :80-82 (feas_sel)
            else:
                nb_opt = sum(1 for e in comb if e[1] > 1000)                                     # :85-88
                nb_comb = sum(1 for e in rank_list if isinstance(e[0], int))                     # :90-92
                a = cs._gen_eigcuts_selected(1, sel - nb_comb, feas[0:(sel - nb_comb)], vars_values=vv)      # :94-95
                b = cs._gen_eigcuts_selected(strat_old, nb_comb, comb[0:nb_comb], vars_values=vv)             # :96-97
                nb_cuts = a + b
            ids = []
            for pos, e in enumerate(rank_list):
                if isinstance(e[0], int):
                    ids.append(e[0])
                else:       # feasibility entry: of the objective cover if it sits in the A part of a strategy-1 list
                    ids.append((key_o if (strat_old == 1 and pos < len(comb)) else key_c)[tuple(e[0])])
            out[p + "is_obj"] = np.array([isinstance(e[0], int) for e in rank_list])
            out[p + "from_obj"] = np.array([pos < min(len(comb), sel) for pos in range(len(rank_list))])
            out[p + "ids"] = np.array(ids, dtype=np.int64)
            out[p + "score"] = np.array([e[1] for e in rank_list], dtype=np.float64)
        out[p + "new_strat"] = np.int64(strat)
        out[p + "nb_cuts"], out[p + "nb_opt_cuts"] = np.int64(nb_cuts), np.int64(nb_opt)
        strat_old = strat                                       # :100-102
        rec = cs._my_prob
        lp.linear_constraints.add(rec.rows, rec.rhs, rec.senses)
        lp.solve()
        out["bounds"].append(lp.get_objective_value())
        print("%s round %d: strategy -> %d, %d cuts (%d by the optimality measure), bound %.6f"
              % (name, r, strat, nb_cuts, nb_opt, out["bounds"][-1]), flush=True)
        out["rounds_done"] = np.int64(r)
    np.savez_compressed(os.path.join(HERE, "qcqp_rounds_%s_s%d.npz" % (name, strat0)),
                        **{k: (np.array(v) if isinstance(v, list) else v) for k, v in out.items()})


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
