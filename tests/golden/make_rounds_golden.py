#!/usr/bin/env python3
"""Golden cutting-plane TRAJECTORIES for BASELINE.json configs[2] (spar125-075-*, mixed-size covers):
the REAL reference's selection and cut generation (`_sel_eigcut_by_ordering_on_measure`,
`_gen_eigcuts_selected`, unmodified, NNs.so through ctypes, numpy LAPACK) driven round after
round, with this repo's HiGHS relaxation standing in for the CPLEX object the reference needs
(cplex is not installed; the hot path never calls into it, see make_golden.py).

Per round the fixture keeps the LP point the reference ranked at, the first `sel` entries of the
rank list it produced (candidate ids and scores), the strategy it switched to, the number of cuts it
generated and the bound after the re-solve.  A GPU test replays every recorded point through the
library and must reproduce the reference's selection -- identical selections imply, by induction,
the identical trajectory.

Runs only in the build container (needs /root/reference).  Hours of CPU for the large covers:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_rounds_golden.py spar125-075-1 4 4 20
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
from make_golden import REF, ROOT, _Recorder, harness  # noqa: E402


def main(name, dim, strat, rounds, frac=0.1, out_dir=HERE):
    qp, _ = make_golden.import_reference()
    cs = qp.CutSolver()
    cs._dim = dim
    cs._load_neural_nets()
    cs._CutSolver__parse_boxqp_into_cplex(name)
    t = time.time()
    nb = cs._get_sdp_vertex_cover(dim)
    print("cover: %d candidates in %.1f s" % (nb, time.time() - t), flush=True)
    n = cs._nb_vars
    inst = harness.parse_boxqp(os.path.join(REF, "boxqp_instances", name + ".in"))
    assert np.array_equal(inst["Q_arr"], np.asarray(cs._Q_arr))
    lp = harness.boxqp_relaxation(inst)
    lp.solve()
    sel = min(int(np.floor(frac * nb)), 5000)                    # cut_select_qp.py:123-125
    out = dict(name=np.array(name), dim=np.int64(dim), strat0=np.int64(strat), sel_size=np.int64(sel),
               nb_subproblems=np.int64(nb), bounds=[lp.get_objective_value()])
    cur = strat
    for r in range(1, rounds + 1):
        vv = np.array(lp.get_values())
        t = time.time()
        if cur == 4:
            nxt, rl = cs._sel_eigcut_by_ordering_on_measure(4, vv, r, sel_size=sel)
        else:
            nxt, rl = cur, cs._sel_eigcut_by_ordering_on_measure(cur, vv, r)
        cs._my_prob = _Recorder()
        nb_cuts = cs._gen_eigcuts_selected(cur, sel, rl, vars_values=vv)
        t_sep = time.time() - t
        head = rl[0:sel]
        if cur == 1:
            key = getattr(main, "_key", None)
            if key is None:
                key = main._key = {tuple(e[0]): i for i, e in enumerate(cs._agg_list)}
            ids = np.array([key[tuple(e[0])] for e in head], dtype=np.int64)
        else:
            ids = np.array([e[0] for e in head], dtype=np.int64)
        p = "r%02d_" % r
        out[p + "vars"] = vv
        out[p + "strat"] = np.int64(cur)
        out[p + "new_strat"] = np.int64(nxt)
        out[p + "ids"] = ids.astype(np.int32)
        out[p + "score"] = np.array([e[1] for e in head], dtype=np.float64)
        out[p + "list_len"] = np.int64(len(rl))
        out[p + "nb_cuts"] = np.int64(nb_cuts)
        del rl, head
        rec = cs._my_prob
        lp.linear_constraints.add(rec.rows, rec.rhs, rec.senses)
        t = time.time()
        lp.solve()
        out["bounds"].append(lp.get_objective_value())
        print("round %d: strat %d -> %d, %d cuts, separation (reference, CPU) %.1f s, LP %.1f s, bound %.4f"
              % (r, cur, nxt, nb_cuts, t_sep, time.time() - t, -out["bounds"][-1]), flush=True)
        cur = nxt
        out["rounds_done"] = np.int64(r)
        np.savez_compressed(os.path.join(out_dir, "rounds_%s_d%d_s%d.npz" % (name.replace("-", "_"), dim, strat)),
                            **{k: (np.array(v) if isinstance(v, list) else v) for k, v in out.items()})


if __name__ == "__main__":
    # optional 5th argument: directory to write to (long captures write aside and are moved in when complete)
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), out_dir=sys.argv[5] if len(sys.argv) > 5 else HERE)
