#!/usr/bin/env python3
"""Rounds on the c3 cover (spar125-075-1 dim 4, bench.py's setup) through ONE route and ONE strategy, for profiling:

    tools/c3_rounds.py <fused_rows|fused_csr|dropin_pair> <4|1> [steps] [--cprofile] [--legacy] [--round R]

prints ms per round; --cprofile adds the host-side profile of the route; --legacy makes the drop-in pair take the
round-2 route (rank + sdpcut_cut_rows + host assembly of the CSR block) for before/after comparisons."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    route, strat = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 100
    import torch
    cs, sc, pts, n = bench.c3_setup(0)
    f = bench.c3_steps(cs, sc)[route]
    if "--legacy" in sys.argv:
        from sdpcutsel_via_nn_amd import cut_solver
        cut_solver._FUSED_HEAD_MAX = 0
    vv = pts[strat]
    if "--round" in sys.argv:      # the LP point the reference's trajectory recorded for round R (tests/golden) instead of bench.py's choice
        import numpy as np
        r = int(sys.argv[sys.argv.index("--round") + 1])
        g = np.load(os.path.join(bench.ROOT, "tests", "golden", "rounds_%s_d%d_s4.npz" % (bench.C3_INSTANCE.replace("-", "_"), bench.C3_DIM)))
        assert int(g["r%02d_strat" % r]) == strat, "round %d of the recorded trajectory ran strategy %d" % (r, int(g["r%02d_strat" % r]))
        vv = np.ascontiguousarray(g["r%02d_vars" % r], dtype=np.float64)
    for _ in range(30):
        cuts = f(strat, vv)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        f(strat, vv)
    torch.cuda.synchronize()
    print("%s strategy %d: %.3f ms per round, %d cuts, %d candidates" % (route, strat, (time.perf_counter() - t0) / steps * 1e3, cuts, n))
    if "--cprofile" in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(steps):
            f(strat, vv)
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(14)


main()
