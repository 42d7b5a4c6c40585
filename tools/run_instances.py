#!/usr/bin/env python3
"""End-to-end cutting-plane rounds on the BoxQP fixtures (GPU box): bounds, gap closed, separation and
LP times per round.

    python tools/run_instances.py                      # the small instances (seconds)
    python tools/run_instances.py config3 [rounds]     # BASELINE.json configs[2]: spar125-075-1/-2/-3, dim 4, combined
                                                       # strategy, top-10 % / 5000 cap, 20 rounds
    python tools/run_instances.py <name> <dim> <strat> <rounds> [triangle]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdpcutsel_via_nn_amd as pkg  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "instances")
SOL = {"spar020-100-1": 706.5, "spar040-030-1": 839.5, "spar125-075-1": 12330.0, "spar125-075-2": 10382.4694,
       "spar125-075-3": 9635.5}          # best known values, boxqp_instances/filenames.txt


def run(name, dim, strat, rounds, triangle=False, term_on=False):
    cs = pkg.CutSolver()
    t = time.time()

    def progress(r, log):       # long runs: one line per solve, so that a run cut short still leaves its rounds behind
        if rounds > 6:
            c = log.counts[-1] if log.counts else {}
            print("      %s dim %d round %2d: bound %.4f gap %.4f, %s cuts, separation %.4f s, LP %.2f s" % (
                name, dim, r, -log.bounds[-1], (log.bounds[-1] - log.bounds[0]) / (-SOL[name] - log.bounds[0]), c.get("sdp", 0),
                log.separation_s[-1] if log.separation_s else 0.0, log.solve_s[-1]), flush=True)

    bounds, t_total, rt, st, cuts, tri, nsub = cs.cut_select_algo(os.path.join(G, name + ".in"), dim, 0.1, strat=strat,
                                                                  nb_rounds_cuts=rounds, triangle_on=triangle, term_on=term_on,
                                                                  on_round=progress)
    gaps = [(bounds[0] - b) / (bounds[0] - SOL[name]) for b in bounds]
    print("%s dim %d strat %d%s: N=%d, %d rounds, total %.1f s (separation %.3f s, LP + model %.1f s)"
          % (name, dim, strat, " +tri" if triangle else "", nsub, len(cuts) - 1, time.time() - t, sum(st[1:]),
             sum(rt) - sum(st[1:])), flush=True)
    print("   round  bound        gap_closed  psd_cuts tri_cuts  separation_s  lp_s")
    for r in range(len(bounds)):
        print("   %3d   %12.4f  %8.4f   %6d  %6d   %10.4f  %8.2f"
              % (r, bounds[r], gaps[r], cuts[r], (tri[r - 1] if (r and tri) else 0), st[r], rt[r] - (st[r] if r else 0)), flush=True)
    return gaps


if __name__ == "__main__":
    a = sys.argv[1:]
    if not a:
        for name, dim, strat, rounds in (("spar020-100-1", 3, 2, 4), ("spar020-100-1", 3, 1, 4), ("spar020-100-1", 3, 4, 4),
                                         ("spar040-030-1", 5, 4, 4), ("spar125-075-1", 3, 4, 2)):
            run(name, dim, strat, rounds)
    elif a[0] == "config3":
        rounds = int(a[1]) if len(a) > 1 else 20
        for name in ("spar125-075-1", "spar125-075-2", "spar125-075-3"):
            run(name, 4, 4, rounds)
    else:
        run(a[0], int(a[1]), int(a[2]), int(a[3]), triangle=len(a) > 4 and a[4] == "triangle")
